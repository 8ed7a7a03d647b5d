"""QOI lane decoder, kernel only: streams resident in HBM (gamut_hip_qoi_decode_resident_device), wall time of the call
(it returns when the decode has finished).  8 distinct 1080p streams rotated over the lanes, so neighbouring lanes diverge."""
import sys, os, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from gamut_amd import _capi, synth
L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
w, h = 1920, 1080
dev = torch.device("cuda", 0)
ND = 8
rgb = synth.synth_rgb_batch(ND, w, h, dev, seed=5).permute(0, 2, 3, 1).to(torch.uint8).cpu().numpy()
rgb[:, 200:400, 300:900] = rgb[:, 200:201, 300:301]                     # a flat patch: RUN ops
files = [synth.qoi_encode(np.ascontiguousarray(rgb[i])) for i in range(ND)]
files.append(b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([3, 0]) +
             np.concatenate([np.full((w * h, 1), 0xFE, np.uint8), rgb[0].reshape(-1, 3)], axis=1).tobytes() + bytes(7) + b"\x01")   # all QOI_OP_RGB
SL = 160
for name, pool in (("spec-encoder streams", files[:ND]), ("all-RGB-op stream", files[ND:])):
    print(f"{name}: {np.mean([len(f) for f in pool]) / 1e6:.2f} MB per 1080p file")
    for B in [int(x) for x in os.environ.get("QOI_BENCH_B", "64,1024,4096").split(",")]:
        begin, size, parts, pos = [], [], [], 0
        for i in range(B):
            f = pool[i % len(pool)]
            begin.append(pos); size.append(len(f)); parts.append(f); parts.append(bytes(SL)); pos += len(f) + SL
        blob = torch.from_numpy(np.frombuffer(b"".join(parts), np.uint8).copy()).to(dev)
        descs = (_capi.QoiDesc * B)()
        for i in range(B):
            f = pool[i % len(pool)]
            _capi.check(L.gamut_hip_qoi_read_header(f, len(f), C.byref(descs[i])))
        out = torch.empty((B, h * w * 4), dtype=torch.uint8, device=dev)
        offs = (np.arange(B, dtype=np.int64) * w * h * 4)
        b_arr = np.array(begin, np.int64); s_arr = np.array(size, np.int32)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _capi.check(L.gamut_hip_qoi_decode_resident_device(blob.data_ptr(), blob.numel(), b_arr.ctypes.data_as(C.POINTER(C.c_int64)), s_arr.ctypes.data_as(C.POINTER(C.c_int)),
                                                               descs, B, 4, offs.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), None))
            torch.cuda.synchronize()                                   # the call is asynchronous
            dt = time.perf_counter() - t0
        prof = getattr(C.CDLL(_capi.LIB_PATH), "gamut_hip_qoi_profile", None)
        if prof:
            buf = (C.c_ulonglong * 8)(); prof(buf); v = list(buf); tot = sum(v[:7]) or 1
            print("    wave 0 cycles: " + ", ".join(f"{n} {100 * v[k] / tot:.0f}%" for k, n in enumerate(["boundaries + chain", "barriers", "compaction", "op scans", "pixels + table + output (W=4) / rest", "W=1 resolve", "W=1 emit"])) + f"; {tot / B / 2 / 1e6:.1f} Mcycles per stream")
        ok = all(np.array_equal(out[i].view(h, w, 4)[:, :, :3].cpu().numpy(), rgb[i % len(pool) if len(pool) > 1 else 0]) for i in (0, B - 1))
        print(f"  x {B}: {dt * 1e3:.1f} ms  {B * w * h / dt / 1e6:.0f} Mpx/s  ({dt / (w * h) * 1e9:.0f} ns per pixel per lane)  parity {'ok' if ok else 'FAIL'}")
        del blob, out
