#!/usr/bin/env python3
"""python tools/timeline.py <rocprofv3 output dir>: kernels and memory copies (> 30 us) of the LAST call of the profiled process, ms from its first event.
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o t -- python tools/files_bench.py jpeg   (profiles/r04_jpeg_timeline.txt)"""
import csv, glob, sys, os
d=sys.argv[1]
ev=[]
for f in glob.glob(os.path.join(d,"**","*kernel_trace.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K "+r["Kernel_Name"].replace("gamut::(anonymous namespace)::","")[:60], r.get("Stream_Id", r.get("Queue_Id",""))))
for f in glob.glob(os.path.join(d,"**","*memory_copy_trace.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"C "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size","")), ""))
ev.sort()
# the last call: events after the last gap > 20 ms
cut=0
for i in range(1,len(ev)):
    if ev[i][0]-max(e[1] for e in ev[max(0,i-50):i])>20_000_000: cut=i
last=ev[cut:]
t0=last[0][0]
for s,e,n,q in last:
    if (e-s)>int(os.environ.get("TL_MIN_US","30"))*1000: print(f"{(s-t0)/1e6:8.3f} -> {(e-t0)/1e6:8.3f} ms ({(e-s)/1e6:6.3f})  {n} {q}")
