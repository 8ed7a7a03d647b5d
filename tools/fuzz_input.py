#!/usr/bin/env python3
"""fuzz_input.py -- differential fuzzing of the JPEG input layer on the CPU (no GPU minutes): mutated baseline / progressive / restart /
JFIF / EXIF files through
    R  tools/ref_literal_input.py     the kept-state second reading of jpegload.d (Python)
    O  oracle/liboracle.so            orc_jpeg_decode_coeffs (oracle_jpeg_input.c)
    P  gamut_amd/lib/libgamut_hip.so  gamut_hip_jpeg_decode_coeffs, the product's host feeder  (--product)
and compares verdict, width / height / components, every coefficient, every max_zag, pixelAspectRatio and dotsPerInchY (NaN == NaN).
Files on which the second reading raises Undefined (the reference has no defined result) are counted apart: there the repo's decoders must
reject (O returns -2, P fails).

    python tools/fuzz_input.py --files 200000 --procs 8 --out /tmp/fuzz_input [--product] [--seed 1]

Every disagreement is written to <out>/<kind>_<hash>.jpg; minimised representatives are committed under tests/golden/jpeg_fuzz/ by
tools/make_jpeg_fuzz_fixtures.py together with what the second reading says about them.
Needs /root/reference only in so far as ref_literal_input.py was written from it; reads nothing from it at run time.
"""
import argparse
import ctypes as C
import hashlib
import io
import math
import multiprocessing as mp
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]


# ------------------------------------------------------------------------------------------------------------ seeds
def exif_segment(order=b"II", version=42, xres=(300, 1), yres=(150, 1), unit=2, ifd_offset=8, next_ifd=0, extra_tags=(), pad=0, second_ifd=None):
    """an APP1 segment: "Exif\\0\\0" + a TIFF header + one IFD with XResolution / YResolution / ResolutionUnit (+ a second IFD)"""
    le = order == b"II"
    e16 = (lambda v: struct.pack("<H", v)) if le else (lambda v: struct.pack(">H", v))
    e32 = (lambda v: struct.pack("<I", v)) if le else (lambda v: struct.pack(">I", v))
    tags = []
    if xres is not None:
        tags.append((282, 5, 1, "x"))
    if yres is not None:
        tags.append((283, 5, 1, "y"))
    if unit is not None:
        tags.append((296, 3, 1, unit))
    tags += list(extra_tags)
    n = len(tags)
    data_at = ifd_offset + 2 + 12 * n + 4
    body = bytearray()
    values = bytearray()
    for tag, typ, cnt, val in tags:
        if val == "x":
            off = data_at + len(values); values += e32(xres[0]) + e32(xres[1]); v = e32(off)
        elif val == "y":
            off = data_at + len(values); values += e32(yres[0]) + e32(yres[1]); v = e32(off)
        else:
            v = e32(val)
        body += e16(tag) + e16(typ) + e32(cnt) + v
    if second_ifd is not None:
        next_ifd = data_at + len(values)
    tiff = order + e16(version) + e32(ifd_offset) + bytes(ifd_offset - 8) + e16(n) + bytes(body) + e32(next_ifd) + bytes(values)
    if second_ifd is not None:
        tiff += e16(len(second_ifd)) + b"".join(e16(t) + e16(3) + e32(1) + e32(v) for t, v in second_ifd) + e32(0)
    payload = b"Exif\0\0" + tiff + bytes(pad)
    return b"\xFF\xE1" + struct.pack(">H", len(payload) + 2) + payload


def insert_after_soi(data, seg, after_app0=True):
    pos = 2
    if after_app0 and data[2:4] == b"\xFF\xE0":
        pos = 4 + struct.unpack(">H", data[4:6])[0]
    return data[:pos] + seg + data[pos:]


def strip_app0(data):
    if data[2:4] == b"\xFF\xE0":
        return data[:2] + data[4 + struct.unpack(">H", data[4:6])[0]:]
    return data


def seeds(seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    out = []

    def picture(w, h, kind):
        yy, xx = np.mgrid[0:h, 0:w]
        if kind == "noise":
            a = rng.integers(0, 256, (h, w, 3))
        elif kind == "smooth":
            a = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1)
        else:
            a = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1) + rng.integers(-12, 12, (h, w, 3))
        return np.clip(a, 0, 255).astype(np.uint8)

    def enc(a, **kw):
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", **kw)
        return b.getvalue()

    for (w, h) in ((8, 8), (17, 13), (40, 56), (33, 70)):
        for kind in ("noise", "smooth", "mixed"):
            a = picture(w, h, kind)
            for ss in (0, 1, 2):
                out.append(("base_%dx%d_%s_s%d" % (w, h, kind, ss), enc(a, quality=int(rng.integers(30, 96)), subsampling=ss)))
            out.append(("gray_%dx%d_%s" % (w, h, kind), enc(a[:, :, 0], quality=80)))
            out.append(("opt_%dx%d_%s" % (w, h, kind), enc(a, quality=75, subsampling=2, optimize=True)))
            out.append(("prog_%dx%d_%s" % (w, h, kind), enc(a, quality=80, subsampling=int(rng.integers(0, 3)), progressive=True)))
            out.append(("proggray_%dx%d_%s" % (w, h, kind), enc(a[:, :, 1], quality=70, progressive=True)))
            for rb in (1, 2, 5):
                out.append(("rst%d_%dx%d_%s" % (rb, w, h, kind), enc(a, quality=80, subsampling=int(rng.integers(0, 3)), restart_marker_blocks=rb)))
            out.append(("progrst_%dx%d_%s" % (w, h, kind), enc(a, quality=80, subsampling=2, progressive=True, restart_marker_blocks=3)))
    # files longer than the 8 KiB input buffer (prep_in_buffer boundaries inside markers, scans, restart searches)
    big = picture(150, 130, "noise")
    out.append(("big_base", enc(big, quality=92, subsampling=2)))
    out.append(("big_rst", enc(big, quality=92, subsampling=0, restart_marker_blocks=4)))
    out.append(("big_prog", enc(big, quality=90, subsampling=2, progressive=True)))
    out.append(("big_gray", enc(big[:, :, 0], quality=95)))
    # density: JFIF units 0 / 1 / 2, EXIF in both byte orders and units, both, EXIF only
    a = picture(24, 16, "mixed")
    plain = enc(a, quality=80, subsampling=2)
    out.append(("jfif_dpi", enc(a, quality=80, subsampling=2, dpi=(300, 150))))
    jf = bytearray(plain); jf[13] = 2; jf[14:18] = struct.pack(">HH", 118, 59); out.append(("jfif_cm", bytes(jf)))
    jf = bytearray(plain); jf[13] = 0; jf[14:18] = struct.pack(">HH", 4, 3); out.append(("jfif_ratio", bytes(jf)))
    jf = bytearray(plain); jf[13] = 7; out.append(("jfif_unit7", bytes(jf)))
    out.append(("nojfif", strip_app0(plain)))
    for name, kw in (("exif_ii", {}), ("exif_mm", dict(order=b"MM")), ("exif_cm", dict(unit=3, xres=(1181, 10), yres=(590, 10))),
                     ("exif_nounit", dict(unit=None)), ("exif_unit1", dict(unit=1)), ("exif_frac", dict(xres=(7200, 100), yres=(0, 0))),
                     ("exif_ifd16", dict(ifd_offset=16)), ("exif_two", dict(second_ifd=[(296, 3), (40000, 1)])), ("exif_pad", dict(pad=40)),
                     ("exif_badorder", dict(order=b"XX")), ("exif_v43", dict(version=43)), ("exif_oob", dict(next_ifd=60000)),
                     ("exif_noxy", dict(xres=None, yres=None))):
        seg = exif_segment(**kw)
        out.append((name, insert_after_soi(plain, seg)))
        out.append((name + "_only", insert_after_soi(strip_app0(plain), seg, after_app0=False)))
    out.append(("exif_then_jfif", plain[:2] + exif_segment() + plain[2:]))
    out.append(("exif_prog", insert_after_soi(enc(a, quality=80, progressive=True), exif_segment(order=b"MM", unit=3))))
    out.append(("app1_xmp", insert_after_soi(plain, b"\xFF\xE1" + struct.pack(">H", 2 + 40) + b"http://ns.adobe.com/xap/1.0/\0" + bytes(11))))
    out.append(("com", insert_after_soi(plain, b"\xFF\xFE" + struct.pack(">H", 7) + b"hello")))
    return out


# ------------------------------------------------------------------------------------------------------------ mutations
MARKERS = [0x00, 0x01, 0xC0, 0xC1, 0xC2, 0xC3, 0xC4, 0xC8, 0xC9, 0xCC, 0xD0, 0xD1, 0xD3, 0xD7, 0xD8, 0xD9, 0xDA, 0xDB, 0xDD, 0xE0, 0xE1, 0xE2, 0xFE, 0xFF, 0x02, 0xF0]


def sos_offset(d):
    i = d.find(b"\xFF\xDA")
    return i if i > 0 else len(d) // 2


def i_seg(d, q):
    """start of the DHT segment that holds offset q"""
    return d.rfind(b"\xFF\xC4", 0, q)


def mutate(data, rng):
    for _ in range(8):
        try:
            return _mutate(data, rng)
        except (IndexError, ValueError, struct.error):      # an operation that does not fit this (already shortened) file: draw again
            continue
    return bytes(data)


def _mutate(data, rng):
    d = bytearray(data)
    n = int(rng.integers(1, 4))
    for _ in range(n):
        if not d:
            break
        op = int(rng.integers(0, 22))
        sos = sos_offset(d)
        head = int(rng.integers(0, max(1, sos)))
        body = int(rng.integers(min(sos, len(d) - 1), len(d)))
        anyw = int(rng.integers(0, len(d)))
        at = (head, body, anyw)[int(rng.integers(0, 3))]
        if op == 0:
            d[at] ^= 1 << int(rng.integers(0, 8))
        elif op == 1:
            d[at] = int(rng.integers(0, 256))
        elif op == 2:
            d[at] = 0xFF
        elif op == 3:                                       # a marker dropped in
            d[at:at] = bytes([0xFF, MARKERS[int(rng.integers(0, len(MARKERS)))]])
        elif op == 4:                                       # a marker written over two bytes
            d[at:at + 2] = bytes([0xFF, MARKERS[int(rng.integers(0, len(MARKERS)))]])
        elif op == 5:
            k = int(rng.integers(1, 1 + min(40, len(d) - at)))
            del d[at:at + k]
        elif op == 6:
            k = int(rng.integers(1, 40))
            d[at:at] = d[at:at + k]
        elif op == 7:                                       # truncate
            del d[int(rng.integers(2, len(d) + 1)):]
        elif op == 8:                                       # bytes in front of SOI
            k = int(rng.choice([1, 2, 3, 17, 500, 4093, 4094, 4095, 4096, 5000]))
            pre = bytes(rng.integers(0, 255, k).astype(np.uint8)) if rng.integers(0, 2) else bytes(k)
            d[0:0] = pre
        elif op == 9:                                       # something behind the scan: garbage, a marker segment, a bad one
            tail = [b"\xFF\x01", b"\xFF\xC8", b"\xFF\xD3", b"\xFF\xC4\x00\x03\x00", b"\xFF\xDB\x00\x02", b"\xFF\xFE\x00\x10abc", b"\x12\x34\xFF\x00\x99",
                    b"\xFF\xE1\x00\x10Exif\0\0XX\0\x2A", b"\xFF\xE0\x00\x10JFIF\0\x01\x01\x01\x00\x60\x00\x30\0\0", b"\xFF\xDD\x00\x04\x00\x02", b"\xFF\xC0"][int(rng.integers(0, 11))]
            eoi = d.rfind(b"\xFF\xD9")
            if eoi > 0 and rng.integers(0, 2):
                d[eoi:eoi] = tail
            else:
                d += tail
        elif op == 10:                                      # a segment length nudged
            i = d.find(b"\xFF", head)
            if 0 <= i < len(d) - 3:
                v = (struct.unpack(">H", d[i + 2:i + 4])[0] + int(rng.choice([-3, -2, -1, 1, 2, 3, 16, 300, 60000]))) & 0xFFFF
                d[i + 2:i + 4] = struct.pack(">H", v)
        elif op == 11:                                      # FF 00 pair in the scan / a stuffed byte unstuffed
            if rng.integers(0, 2):
                d[body:body] = b"\xFF\x00"
            else:
                i = d.find(b"\xFF\x00", sos)
                if i > 0:
                    del d[i + 1]
        elif op == 12:                                      # RST juggling
            i = d.find(b"\xFF\xD0", sos)
            if i > 0:
                d[i + 1] = 0xD0 + int(rng.integers(0, 8))
            else:
                d[body:body] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
        elif op == 13:                                      # fill bytes in front of a marker
            i = d.find(b"\xFF", at)
            if i >= 0:
                d[i:i] = b"\xFF" * int(rng.integers(1, 6))
        elif op == 14:                                      # drop EOI / cut the tail
            if d[-2:] == b"\xFF\xD9":
                del d[-2:]
            del d[len(d) - int(rng.integers(0, min(8, len(d)))):]
        elif 16 <= op <= 19:                                # a Huffman table bent: a count changed, a symbol value repeated / raised / zeroed
            tabs = []
            i = d.find(b"\xFF\xC4")
            while 0 <= i < len(d) - 4:
                L = struct.unpack(">H", d[i + 2:i + 4])[0]
                q = i + 4
                while q + 17 <= min(len(d), i + 2 + L):
                    cnt = sum(d[q + 1:q + 17])
                    tabs.append((q, cnt))
                    q += 17 + cnt
                i = d.find(b"\xFF\xC4", i + 2 + max(L, 2))
            if tabs:
                q, cnt = tabs[int(rng.integers(0, len(tabs)))]
                if op == 16:                                # one code word fewer / more of some length (the segment length is NOT fixed up: some are refused, some shift)
                    l = int(rng.integers(1, 17))
                    if q + 17 + cnt > len(d):
                        pass
                    elif rng.integers(0, 2) and d[q + l] > 0 and cnt > 0:
                        d[q + l] -= 1
                        del d[q + 17 + cnt - 1]
                        L = struct.unpack(">H", d[i_seg(d, q) + 2:i_seg(d, q) + 4])[0]
                        d[i_seg(d, q) + 2:i_seg(d, q) + 4] = struct.pack(">H", max(2, L - 1))
                    elif d[q + l] < 255:
                        d[q + l] += 1
                        d[q + 17 + cnt:q + 17 + cnt] = bytes([int(rng.integers(0, 256))])
                        L = struct.unpack(">H", d[i_seg(d, q) + 2:i_seg(d, q) + 4])[0]
                        d[i_seg(d, q) + 2:i_seg(d, q) + 4] = struct.pack(">H", min(65535, L + 1))
                elif cnt and q + 17 + cnt <= len(d):
                    v = q + 17 + int(rng.integers(0, cnt))
                    if op == 17:
                        d[v] = d[q + 17 + int(rng.integers(0, cnt))]          # a symbol value twice
                    elif op == 18:
                        d[v] = int(rng.choice([0, 0x10, 0x1F, 0xF0, 0xFF, 16, 17]))
                    else:
                        d[q] = int(rng.choice([0x00, 0x01, 0x04, 0x10, 0x11, 0x13, 0x14, 0x20]))      # the table's class / number
        elif op == 20:                                      # SOS: selectors, component ids, spectral bytes
            if sos + 5 < len(d):
                ns = d[sos + 4]
                k = sos + 5 + int(rng.integers(0, max(1, 2 * ns + 3)))
                if k < len(d):
                    d[k] = int(rng.choice([0x00, 0x01, 0x02, 0x03, 0x10, 0x11, 0x40, 0x44, 0x80, 0x0F, 0x3F, 0xFF]))
        else:                                               # swap two ranges of the header (segment order)
            k = int(rng.integers(2, 30))
            a, b = sorted((int(rng.integers(2, max(3, sos))), int(rng.integers(2, max(3, sos)))))
            if a + k <= b and b + k <= len(d):
                d[a:a + k], d[b:b + k] = d[b:b + k], d[a:a + k]
    return bytes(d)


# ------------------------------------------------------------------------------------------------------------ the three readers
def feq(a, b):
    a, b = np.float32(a), np.float32(b)
    return bool((np.isnan(a) and np.isnan(b)) or a == b)


def read_R(data):
    import ref_literal_input as R
    try:
        r = R.decompress_jpeg_image_from_stream(data)
    except R.Undefined as e:
        return ("undefined", str(e))
    except RecursionError:
        return ("undefined", "recursion")
    if r is None:
        return ("null",)
    co = np.array([m[0] for m in r["mcus"]], np.int16).reshape(-1, 64)
    mz = np.array([z for m in r["mcus"] for z in m[1]], np.uint8)
    return ("image", r["width"], r["height"], r["actual_comps"], co, mz, r["pixelAspectRatio"], r["dotsPerInchY"])


def read_O(data):
    import oracle_lib as O
    f = O.JpegFrame()
    buf = np.frombuffer(bytes(data), np.uint8)
    rc = O.lib().orc_jpeg_decode_coeffs(O._ptr(buf) if buf.size else None, buf.size, C.byref(f))
    if rc == -2:
        return ("undefined", "")
    if rc != 0:
        return ("null",)
    n = f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu
    out = ("image", f.width, f.height, f.comps, np.ctypeslib.as_array(f.coeffs, (n, 64)).copy(), np.ctypeslib.as_array(f.max_zag, (n,)).copy(),
           f.pixel_aspect_ratio, f.dpi_y)
    O.lib().orc_jpeg_frame_free(C.byref(f))
    return out


def read_P(data):
    from gamut_amd import _capi
    L = _capi.lib()
    f = _capi.JpegFrame()
    buf = np.frombuffer(bytes(data), np.uint8)
    rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data if buf.size else None, buf.size, C.byref(f))
    if rc != 0:
        return ("null",)
    n = f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu
    out = ("image", f.width, f.height, f.comps, np.ctypeslib.as_array(f.coeffs, (n, 64)).copy(), np.ctypeslib.as_array(f.max_zag, (n,)).copy(),
           f.pixel_aspect_ratio, f.dpi_y)
    L.gamut_hip_jpeg_frame_free(C.byref(f))
    return out


def compare(a, b):
    """'' when equal, else what differs"""
    if a[0] != b[0]:
        return "verdict %s/%s" % (a[0], b[0])
    if a[0] != "image":
        return ""
    if a[1:4] != b[1:4]:
        return "geometry %s/%s" % (a[1:4], b[1:4])
    if a[4].shape != b[4].shape or not np.array_equal(a[4], b[4]):
        return "coefficients"
    if not np.array_equal(a[5], b[5]):
        return "max_zag"
    if not feq(a[6], b[6]):
        return "par %r/%r" % (a[6], b[6])
    if not feq(a[7], b[7]):
        return "dpi %r/%r" % (a[7], b[7])
    return ""


def check(data, product):
    r = read_R(data)
    o = read_O(data)
    res = []
    if r[0] == "undefined":
        if r[1].startswith("EXIF: read of 6 bytes at 0"):       # an APP1 shorter than "Exif\0\0": the repo takes it as not EXIF (oracle_jpeg_input.c)
            return "undefined", res
        if o[0] == "image":
            res.append(("undef-O-accepts", r[1]))
        if product and read_P(data)[0] == "image":
            res.append(("undef-P-accepts", r[1]))
        return "undefined", res
    if o[0] == "undefined":
        res.append(("O-undefined-R-" + r[0], ""))
    else:
        c = compare(r, o)
        if c:
            res.append(("R-O " + c, ""))
    if product:
        c = compare(r, read_P(data))
        if c:
            res.append(("R-P " + c, ""))
    return r[0], res


def worker(args):
    wid, nfiles, seed, product, outdir = args
    rng = np.random.default_rng([seed, wid])
    pool = seeds(0)
    stats = {"image": 0, "null": 0, "undefined": 0}
    found = {}
    t0 = time.time()
    for k in range(nfiles):
        name, data = pool[int(rng.integers(0, len(pool)))]
        m = mutate(data, rng) if k >= len(pool) or wid else data     # worker 0 starts with the unmutated seeds
        try:
            verdict, res = check(m, product)
        except Exception as e:                                      # a crash of a reader is a finding too
            verdict, res = "image", [("EXCEPTION %s: %s" % (type(e).__name__, e), "")]
        stats[verdict] += 1
        for kind, why in res:
            key = kind.split(" (")[0]
            h = hashlib.sha1(m).hexdigest()[:12]
            found.setdefault(key, [])
            if len(found[key]) < 40:
                found[key].append((name, h, why))
                fn = os.path.join(outdir, "%s_%s_%s.jpg" % (key.replace(" ", "_").replace("/", "-")[:60], name, h))
                with open(fn, "wb") as fh:
                    fh.write(m)
    import ref_literal_input as R
    return stats, found, dict(R.EVENTS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=20000)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="/tmp/fuzz_input")
    ap.add_argument("--product", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    per = (a.files + a.procs - 1) // a.procs
    t0 = time.time()
    with mp.Pool(a.procs) as pool:
        results = pool.map(worker, [(w, per, a.seed, a.product, a.out) for w in range(a.procs)])
    stats = {"image": 0, "null": 0, "undefined": 0}
    found = {}
    events = {}
    for s, f, ev in results:
        for k, v in ev.items():
            events[k] = events.get(k, 0) + v
        for k in stats:
            stats[k] += s[k]
        for k, v in f.items():
            found.setdefault(k, []).extend(v)
    total = sum(stats.values())
    print("fuzz_input: %d files (seed %d, %d processes, %.0f s): second reading says image %d / null %d / undefined %d; compared with %s"
          % (total, a.seed, a.procs, time.time() - t0, stats["image"], stats["null"], stats["undefined"], "oracle + product host feeder" if a.product else "oracle"))
    print("exercised (events counted inside the second reading):")
    for k in sorted(events):
        print("  %-90s %d" % (k, events[k]))
    if not found:
        print("no disagreement")
    for k in sorted(found):
        print("  %-60s %d (e.g. %s)" % (k, len(found[k]), ", ".join("%s:%s" % (n, h) for n, h, _ in found[k][:3])))
        whys = sorted({w for _, _, w in found[k] if w})
        for w in whys[:6]:
            print("      ", w[:150])
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
