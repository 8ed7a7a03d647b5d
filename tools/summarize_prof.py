#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small files fit for profiles/:

   <out>/kernel_stats.csv      per-kernel summary over the TIMED dispatches of the profiled bench.py run only: the last `steps`
                               dispatches of every gamut kernel (the cold first call of check() and the warm-up launches of a
                               profiled process are not what bench.py times, and they made round 3's averages exceed the bench
                               line's ms_per_step).  Same columns as rocprofv3's own --stats file, plus `AllCalls` /
                               `AllAverageNs` = the figures over every dispatch of the process, for reference.
   <out>/traffic.json          per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes

   python tools/summarize_prof.py <rocprof-dir> <out-dir> [pmc-batch] [timed-steps]

FETCH_SIZE / WRITE_SIZE: rocprofv3 reports them in KB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) on gfx950
FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced read stream, so the corrected read bytes are
2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as reported (uncalibrated there)."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def counter_avg(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in find(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def timed_stats(trace_dir, steps, launches_per_step=None):
    """Per kernel: durations of its dispatches in start order; the timed ones are the last steps * (launches per step) of them.
    launches per step is inferred when not given: a kernel launched m times per step has (1 + warmup + steps) * m dispatches, so
    m = the largest divisor pattern is ambiguous -- callers pass it for multi-launch steps; default 1."""
    disp = defaultdict(list)
    for f in find(trace_dir, "*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "gamut" not in k:
                continue
            disp[k].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    rows = []
    for k, v in disp.items():
        v.sort()
        d_all = [d for _, d in v]
        m = launches_per_step or 1
        d = d_all[-steps * m:] if steps and len(d_all) > steps * m else d_all
        rows.append({"Name": k, "Calls": len(d), "TotalDurationNs": sum(d), "AverageNs": round(sum(d) / len(d), 1), "MinNs": min(d), "MaxNs": max(d),
                     "StdDev": round(statistics.pstdev(d), 1) if len(d) > 1 else 0.0, "AllCalls": len(d_all), "AllAverageNs": round(sum(d_all) / len(d_all), 1)})
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    return rows


def main():
    base, out = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    per_step = int(sys.argv[5]) if len(sys.argv) > 5 else None
    os.makedirs(out, exist_ok=True)
    rows = timed_stats(os.path.join(base, "trace"), steps, per_step)
    if rows:
        with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as f:
            f.write(f"# gamut kernels of the profiled bench.py run; Calls / AverageNs / Min / Max over the LAST {steps or 'all'} x {per_step or 1} dispatches (the timed steps), "
                    "AllCalls / AllAverageNs over every dispatch of the process (check() + warm-up + timed)\n")
            wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
            wr.writeheader()
            wr.writerows(rows)
    fetch = counter_avg(os.path.join(base, "pmc_fetch"), "FETCH_SIZE")
    write = counter_avg(os.path.join(base, "pmc_write"), "WRITE_SIZE")
    kernels = []
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (None, 0))
        w, nw = write.get(k, (None, 0))
        if "gamut" not in k:
            continue
        row = {"kernel": k, "batch_in_pmc_passes": batch, "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
               "FETCH_SIZE_raw_avg": f, "WRITE_SIZE_raw_avg": w}
        if f is not None and w is not None:
            row["read_bytes_corrected"] = 2 * f * 1024
            row["write_bytes"] = w * 1024
            row["hbm_bytes_per_launch"] = 2 * f * 1024 + w * 1024
            if batch:
                row["hbm_bytes_per_image"] = row["hbm_bytes_per_launch"] / batch
        kernels.append(row)
    if kernels:
        json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); units KB->B",
                   "kernels": kernels}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    for r in rows[:8]:
        print(f"{r['Name'][:90]:90s} timed calls {r['Calls']:4d} avg {r['AverageNs'] / 1e6:9.4f} ms  min {r['MinNs'] / 1e6:9.4f}  (all {r['AllCalls']} calls: {r['AllAverageNs'] / 1e6:.4f} ms)")
    print(json.dumps(kernels, indent=1)[:3000])


if __name__ == "__main__":
    main()
