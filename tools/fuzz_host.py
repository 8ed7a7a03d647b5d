#!/usr/bin/env python3
"""Mutation fuzzing of the host-side parsers (JPEG markers + baseline / progressive entropy decode, PNG chunk walk, QOI
header) under AddressSanitizer.  Build + run (CPU only; GPU AddressSanitizer is not available on this pool):
    bash tools/fuzz_host.sh [iterations]
The driver loads a build of the library whose host code is ASan-instrumented and feeds it mutated fixtures; without a
GPU the calls stop before any kernel launch, which is exactly the untrusted-input surface."""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GAMUT_HIP_LIB", "/tmp/asan/libgamut_hip_asan.so")
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    L = C.CDLL(os.environ["GAMUT_HIP_LIB"])
    L.gamut_hip_jpeg_decode_coeffs.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_capi.JpegFrame)]
    L.gamut_hip_jpeg_read_header.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_capi.JpegFrame)]
    L.gamut_hip_jpeg_scan_layout.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_capi.JpegFrame), C.c_void_p, C.c_void_p]
    L.gamut_hip_jpeg_frame_free.argtypes = [C.POINTER(_capi.JpegFrame)]
    L.gamut_hip_png_is16.argtypes = [C.c_void_p, C.c_size_t]
    L.gamut_hip_qoi_read_header.argtypes = [C.c_void_p, C.c_int, C.POINTER(_capi.QoiDesc)]
    L.gamut_hip_stbi_load_from_memory.restype = C.c_void_p
    L.gamut_hip_stbi_load_from_memory.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
    L.gamut_hip_decompress_jpeg_image_from_stream.restype = C.c_void_p
    L.gamut_hip_decompress_jpeg_image_from_stream.argtypes = [_capi.JPEG_STREAM_READ_FUNC, C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
    L.gamut_hip_stbi_load_from_callbacks.restype = C.c_void_p
    L.gamut_hip_stbi_load_from_callbacks.argtypes = [C.POINTER(_capi.StbiIoCallbacks), C.c_void_p] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
    L.gamut_hip_stbi_png_is16_from_callbacks.argtypes = [C.POINTER(_capi.StbiIoCallbacks), C.c_void_p]
    G = os.path.join(ROOT, "tests", "golden")
    seeds = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(G, "jpeg", "*.jpg")) + glob.glob(os.path.join(G, "ref_images", "*")))]
    seeds.append(gen.qoi_encode(gen.synth_rgb(33, 9, 1)))
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
    ok = bad = 0
    for it in range(iters):
        data = bytearray(seeds[it % len(seeds)])
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 5))
            if not data:
                break
            if kind == 0:
                data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
            elif kind == 1:
                data = data[:int(rng.integers(0, len(data)))]
            elif kind == 2:
                i = int(rng.integers(0, len(data))); data[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            elif kind == 3:
                i = int(rng.integers(0, len(data))); data[i] = 0xFF
                if i + 1 < len(data): data[i + 1] = int(rng.choice([0xC0, 0xC2, 0xC4, 0xDA, 0xDB, 0xDD, 0xD0, 0xD9, 0x00, 0xFF]))
            else:
                i = int(rng.integers(0, max(1, len(data) - 4))); data[i:i + 2] = bytes([int(rng.integers(0, 256)), int(rng.integers(0, 256))])
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) if data else b"\0")
        n = len(data)
        fr = _capi.JpegFrame()
        rc = L.gamut_hip_jpeg_decode_coeffs(buf, n, C.byref(fr))
        ok += rc == 0; bad += rc != 0
        L.gamut_hip_jpeg_frame_free(C.byref(fr))
        L.gamut_hip_jpeg_read_header(buf, n, C.byref(fr))
        L.gamut_hip_jpeg_scan_layout(buf, n, C.byref(fr), C.byref(C.c_int32()), C.byref(C.c_uint64()))
        L.gamut_hip_png_is16(buf, n)
        x, y, c = C.c_int(), C.c_int(), C.c_int(); f = C.c_float()
        L.gamut_hip_stbi_load_from_memory(buf, n, C.byref(x), C.byref(y), C.byref(c), 0, C.byref(f), C.byref(f), C.byref(f))
        L.gamut_hip_qoi_read_header(buf, min(n, 2**31 - 1), C.byref(_capi.QoiDesc()))
        # the stream gatherers (stream_host.hip): the same bytes through callbacks that hand out odd-sized pieces, lie about skips, or fail
        raw = bytes(data)
        piece = int(rng.choice([1, 3, 7, 100, 4096, 1 << 20]))
        st = {"pos": 0}

        def rd_jpeg(pbuf, max_bytes, peof, user):
            k = max(0, min(max_bytes, piece, len(raw) - st["pos"]))
            C.memmove(pbuf, raw[st["pos"]:st["pos"] + k], k); st["pos"] += k
            if st["pos"] >= len(raw):
                peof[0] = 1
            return k
        w_, h_, c_ = C.c_int(), C.c_int(), C.c_int()
        L.gamut_hip_decompress_jpeg_image_from_stream(_capi.JPEG_STREAM_READ_FUNC(rd_jpeg), None, C.byref(w_), C.byref(h_), C.byref(c_), None, None, 4)
        st["pos"] = 0

        def rd_stb(user, buf_, size):
            k = max(0, min(size, piece, len(raw) - st["pos"]))
            C.memmove(buf_, raw[st["pos"]:st["pos"] + k], k); st["pos"] += k
            return k

        def skip_stb(user, k):
            st["pos"] = max(0, min(len(raw) + 5, st["pos"] + k))        # may run past the end, as a file seek would

        def eof_stb(user):
            return int(st["pos"] >= len(raw))
        cb = _capi.StbiIoCallbacks()
        keep = (type(cb.read)(rd_stb), type(cb.skip)(skip_stb), type(cb.eof)(eof_stb))
        cb.read, cb.skip, cb.eof = keep
        L.gamut_hip_stbi_load_from_callbacks(C.byref(cb), None, C.byref(x), C.byref(y), C.byref(c), 0, None, None, None)
        st["pos"] = 0
        L.gamut_hip_stbi_png_is16_from_callbacks(C.byref(cb), None)
    print(f"fuzz_host: {iters} mutated inputs, {ok} decoded, {bad} rejected, no sanitizer report")


if __name__ == "__main__":
    main()
