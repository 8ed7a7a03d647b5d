import ctypes as C, os, sys, io, time
import numpy as np, torch
from PIL import Image
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from gamut_amd import _capi
L=_capi.lib(); _capi.check(L.gamut_hip_init(0))
prof = C.CDLL(_capi.LIB_PATH).gamut_hip_jpeg_sync_profile
w,h,B,distinct=1920,1080,1024,8
files=[]
for i in range(distinct):
    bio=io.BytesIO(); Image.fromarray(gen.synth_rgb(w,h,100+i)).save(bio,"JPEG",quality=90,subsampling=2); files.append(np.frombuffer(bio.getvalue(),np.uint8))
bufs=[files[i%distinct] for i in range(B)]
ptrs=(C.c_void_p*B)(*[b.ctypes.data for b in bufs]); lens=(C.c_size_t*B)(*[b.size for b in bufs])
out=torch.empty((B,h,w*4),dtype=torch.uint8,device="cuda"); off=(np.arange(B,dtype=np.int64)*h*w*4)
st=torch.cuda.current_stream().cuda_stream
def run():
    info=(_capi.JpegFrame*B)(); _capi.check(L.gamut_hip_jpeg_decode_batch_device(ptrs,lens,B,4,off.ctypes.data_as(C.POINTER(C.c_int64)),out.data_ptr(),info,None,None,st)); torch.cuda.synchronize()
run(); buf=(C.c_ulonglong*16)(); prof(buf,1)
t0=time.perf_counter(); run(); dt=time.perf_counter()-t0
prof(buf,0); v=list(buf); n=v[7] or 1
names=["sweep0","resync","scan","write","first+clear","?","#sweeps","#segments"]
print(f"{dt*1e3:.2f} ms; per segment (cycles): "+", ".join(f"{names[k]} {v[k]/n:.0f}" for k in range(7)))
if v[8]:
    print(f"re-decodes per segment: {v[8]/n:.1f} lanes decoded again; {v[9]/n:.1f} stopped at a checkpoint (mean checkpoint number {v[10]/max(v[9],1):.2f} of 0..4), {v[11]/n:.1f} ran to the end of their sub-sequence, {v[12]/n:.1f} of those left in a new state")
