import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, gen
from gamut_amd import _capi
L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
w, h = 1920, 1080
a = gen.synth_rgb(w, h, 3)
t0 = time.perf_counter(); data = gen.qoi_encode(a) if False else None
# fast numpy-free encoder is too slow in python for 2 Mpx; build a stream of RGB ops + runs instead
px = a.reshape(-1, 3)
body = bytearray()
body += b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([3, 0])
ops = np.empty((px.shape[0], 4), np.uint8); ops[:, 0] = 0xFE; ops[:, 1:] = px
body += ops.tobytes() + bytes([0, 0, 0, 0, 0, 0, 0, 1])
buf = np.frombuffer(bytes(body), np.uint8)
for B in (64, 1024):
    ptrs = (C.c_void_p * B)(*[buf.ctypes.data] * B); sizes = (C.c_int * B)(*[buf.size] * B)
    offs = (np.arange(B, dtype=np.int64) * w * h * 4)
    dout = L.gamut_hip_device_malloc(B * w * h * 4)
    descs = (_capi.QoiDesc * B)()
    for rep in range(2):
        t0 = time.perf_counter()
        _capi.check(L.gamut_hip_qoi_decode_batch_device(ptrs, sizes, B, 4, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, descs, None, None))
        dt = time.perf_counter() - t0
    print(f"QOI (all RGB ops, {buf.size/1e6:.1f} MB/file) x {B}: {dt*1e3:.1f} ms  {B*w*h/dt/1e6:.0f} Mpx/s")
    L.gamut_hip_device_free(dout)
