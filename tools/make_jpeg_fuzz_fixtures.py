#!/usr/bin/env python3
"""make_jpeg_fuzz_fixtures.py -- writes the round-5 files of tests/golden/jpeg_fuzz/ (small JPEG files, each built to stand on one habit of jpgd's
input layer that tools/fuzz_input.py exercises at random) and tests/golden/jpeg_fuzz/expected.json: what the SECOND READING of the reference
(tools/ref_literal_input.py, build-container only) says about EVERY file of the directory -- verdict ("image" / "null" / "undefined"), geometry,
pixelAspectRatio / dotsPerInchY (NaN as the string "nan"), SHA-256 of the coefficients and of max_zag in transform order.

The tests compare the oracle (CPU), the product's host feeder (CPU) and the device decoders (-m gpu) with that file: the GPU box has neither
/root/reference nor any use for this script.  "undefined" = the reference reads outside an array / memory nobody wrote / never returns: the repo's
decoders must refuse the file.

    python tools/make_jpeg_fuzz_fixtures.py            (re)write the round-5 files and expected.json
    python tools/make_jpeg_fuzz_fixtures.py --check    exit 1 if expected.json differs from a fresh run
"""
import hashlib
import io
import json
import math
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import fuzz_input as F            # noqa: E402
import ref_literal_input as R     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "jpeg_fuzz")


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(xx * 5 + yy * 2) % 256, (xx + yy * 3) % 256, (xx * yy // 5) % 256], -1) + rng.integers(-10, 10, (h, w, 3))
    return np.clip(a, 0, 255).astype(np.uint8)


def enc(a, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", **kw)
    return b.getvalue()


def seg(marker, payload):
    return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload


def before_eoi(d, extra):
    assert d[-2:] == b"\xFF\xD9"
    return d[:-2] + extra + d[-2:]


def build():
    files = {}
    a = picture(40, 24, 1)
    base = enc(a, quality=80, subsampling=2)                     # 4:2:0, JFIF 1:1 no unit
    nojfif = F.strip_app0(base)
    sos = base.index(b"\xFF\xDA")
    # --- density: APP1 / EXIF (jpegload.d:1704-1816), APP0 / JFIF (:1634-1702), neither (the members stay NaN)
    files["exif_ii_300x150_r05"] = F.insert_after_soi(nojfif, F.exif_segment(), after_app0=False)                         # the judge's probe: par 2.0, dpi_y 150
    files["exif_mm_300x150_r05"] = F.insert_after_soi(nojfif, F.exif_segment(order=b"MM"), after_app0=False)
    files["exif_cm_r05"] = F.insert_after_soi(base, F.exif_segment(unit=3, xres=(1181, 10), yres=(590, 10)))              # behind a JFIF header: the later segment wins
    files["exif_bad_byte_order_r05"] = F.insert_after_soi(nojfif, F.exif_segment(order=b"XX"), after_app0=False)          # null
    files["exif_version_43_r05"] = F.insert_after_soi(nojfif, F.exif_segment(version=43), after_app0=False)               # null
    files["exif_ifd_offset_behind_segment_r05"] = F.insert_after_soi(nojfif, F.exif_segment(next_ifd=60000), after_app0=False)   # null
    files["exif_value_offset_outside_r05"] = F.insert_after_soi(nojfif, F.exif_segment(extra_tags=[(282, 5, 1, 5000)]), after_app0=False)   # undefined (a read behind the malloc block)
    files["exif_second_ifd_unit_r05"] = F.insert_after_soi(nojfif, F.exif_segment(second_ifd=[(296, 3)]), after_app0=False)
    files["exif_then_jfif_r05"] = base[:2] + F.exif_segment() + base[2:]                                                  # JFIF behind EXIF: JFIF wins
    files["app1_not_exif_r05"] = F.insert_after_soi(base, seg(0xE1, b"http://ns.adobe.com/xap/1.0/\0<x/>"))
    files["no_density_segment_r05"] = nojfif                                                                               # NaN / NaN
    jf = bytearray(base); jf[13] = 2; jf[14:18] = struct.pack(">HH", 118, 59); files["jfif_dots_per_cm_r05"] = bytes(jf)
    jf = bytearray(base); jf[13] = 9; files["jfif_unit_9_r05"] = bytes(jf)                                                 # par set, dpi stays NaN
    files["jfif_header_of_14_bytes_r05"] = base[:4] + struct.pack(">H", 14) + base[6:18] + base[20:]                       # the shortest APP0 whose density counts
    files["app0_of_5_bytes_r05"] = nojfif[:2] + seg(0xE0, b"JFI") + nojfif[2:]                                             # null (JPGD_BAD_VARIABLE_MARKER)
    # --- the end of the stream: FF D9 padding (:631-652), 1-bits at a marker (:683-696), symbol 0 for a bit pattern no code word begins (:746-813)
    files["file_ends_inside_its_scan_r05"] = base[:sos + 14 + (len(base) - sos - 14) // 2]                                 # image: the rest of the picture is flat
    files["file_ends_inside_its_sos_r05"] = base[:sos + 11]                                                                # image: Se / AhAl and a data byte 0xD9 come out of the padding
    files["file_ends_inside_a_dqt_r05"] = base[:base.index(b"\xFF\xDB") + 30]                                              # null: EOI where the frame header should be
    # --- find_eoi (:2826-2848): the markers behind the last MCU row are processed like those in front of the frame
    files["tem_behind_the_scan_r05"] = before_eoi(base, b"\xFF\x01")                                                       # null
    files["jpg_behind_the_scan_r05"] = before_eoi(base, b"\xFF\xC8")                                                       # null
    files["rst_behind_the_scan_r05"] = before_eoi(base, b"\xFF\xD3\xFF\xD0")                                               # image (Issue #93)
    files["bad_dht_behind_the_scan_r05"] = before_eoi(base, b"\xFF\xC4\x00\x03\x00")                                       # null
    files["bad_exif_behind_the_scan_r05"] = before_eoi(base, F.exif_segment(order=b"XX"))                                  # null
    files["exif_behind_the_scan_r05"] = before_eoi(base, F.exif_segment())                                                 # image; the density is the trailer's
    files["com_longer_than_the_file_behind_the_scan_r05"] = base[:-2] + b"\xFF\xFE\x7F\xFF" + b"abc"                       # image: the segment is read out of the padding
    files["sof_behind_the_scan_r05"] = before_eoi(base, b"\xFF\xC0")                                                       # image: any of SOFn / SOI / EOI / SOS ends the search
    files["stuffed_bytes_behind_the_scan_r05"] = before_eoi(base, b"\x12\x34\xFF\x00\x56")                                 # image
    # --- process_markers' verdicts in front of the frame / the scan (:1818-1848, :1911-1967)
    dqt = base.index(b"\xFF\xDB")
    files["rst_in_front_of_the_frame_r05"] = base[:dqt] + b"\xFF\xD0" + base[dqt:]                                         # null
    files["tem_in_front_of_the_frame_r05"] = base[:dqt] + b"\xFF\x01" + base[dqt:]                                         # null
    files["second_soi_r05"] = base[:dqt] + b"\xFF\xD8" + base[dqt:]                                                        # null (JPGD_UNSUPPORTED_MARKER)
    files["sof3_r05"] = base.replace(b"\xFF\xC0", b"\xFF\xC3", 1)                                                          # null
    files["rst_between_frame_and_scan_r05"] = base[:sos] + b"\xFF\xD2" + base[sos:]                                        # null (JPGD_UNEXPECTED_MARKER by way of init_sequential)
    files["fill_bytes_and_ff00_between_segments_r05"] = base[:dqt] + b"\x00\x11\xFF\x00\xFF\xFF\xFF" + base[dqt + 1:]     # image: next_marker skips all of it
    # --- locate_soi_marker (:1854-1908)
    files["17_bytes_in_front_of_soi_r05"] = bytes(range(1, 18)) + base                                                     # image
    files["4095_bytes_in_front_of_soi_r05"] = bytes(4095) + base                                                           # image: the last place the search reaches (2 + 4095 bytes read)
    files["4096_bytes_in_front_of_soi_r05"] = bytes(4096) + base                                                           # null
    files["soi_found_but_no_marker_behind_it_r05"] = b"\x00\x00\xFF\xD8\x00" + base[2:]                                    # null
    files["eoi_in_front_of_soi_r05"] = b"\x00\xFF\xD9" + base                                                              # null
    # --- tables
    s0 = bytearray(base); s0[sos + 6] = 0x40 | (s0[sos + 6] & 15); files["dc_selector_names_an_ac_table_r05"] = bytes(s0)   # image: m_pHuff_tabs[4] decodes the DC differences
    s0 = bytearray(base); s0[sos + 6] = 0x80; files["dc_selector_8_r05"] = bytes(s0)                                        # undefined: m_huff_num[8]
    dht = base.index(b"\xFF\xC4")
    over = bytearray(base)                                         # two more code words of 1 bit, two fewer of the most frequent length: same symbols, 4 one-bit codes
    kmax = max(range(2, 17), key=lambda l: over[dht + 4 + l]); over[dht + 5] += 2; over[dht + 4 + kmax] -= 2
    files["dht_over_subscribed_r05"] = bytes(over)                                                                         # undefined: look_up[] written behind its end
    files["dht_over_subscribed_behind_the_scan_r05"] = before_eoi(base, bytes(over[dht:dht + 2 + struct.unpack(">H", base[dht + 2:dht + 4])[0]]))   # image: read, never built
    # a progressive file whose AC table carries a symbol value twice: the two-argument huff_decode takes code_size[symbol] bits (:761)
    prog = enc(a, quality=80, subsampling=2, progressive=True)
    files["progressive_without_a_scan_r05"] = prog[:prog.index(b"\xFF\xDA")] + b"\xFF\xD9"                                  # image: every coefficient 0
    psos = [i for i in range(len(prog) - 1) if prog[i] == 0xFF and prog[i + 1] == 0xDA]
    files["progressive_tem_between_scans_r05"] = prog[:psos[2]] + b"\xFF\x01" + prog[psos[2]:]                              # undefined: decode() goes on without an error code
    rng = np.random.default_rng(11)
    R.EVENTS.clear()
    for _ in range(4000):                                          # a DHT symbol repeated until the quirk is exercised and the file still decodes
        d = bytearray(prog)
        tabs = []
        i = d.find(b"\xFF\xC4")
        while 0 <= i < len(d) - 4:
            L = struct.unpack(">H", d[i + 2:i + 4])[0]
            q = i + 4
            while q + 17 <= i + 2 + L:
                cnt = sum(d[q + 1:q + 17]); tabs.append((q, cnt)); q += 17 + cnt
            i = d.find(b"\xFF\xC4", i + 2 + L)
        q, cnt = tabs[int(rng.integers(0, len(tabs)))]
        d[q + 17 + int(rng.integers(0, cnt))] = d[q + 17 + int(rng.integers(0, cnt))]
        before = R.EVENTS["two-argument huff_decode: code_size[symbol] is another code word's length"]
        try:
            r = R.decompress_jpeg_image_from_stream(bytes(d))
        except R.Undefined:
            continue
        if r is not None and R.EVENTS["two-argument huff_decode: code_size[symbol] is another code word's length"] > before:
            files["progressive_symbol_value_twice_r05"] = bytes(d)
            break
    assert "progressive_symbol_value_twice_r05" in files
    # an AC table with a code word missing (one count lowered, its symbol dropped): patterns under the hole decode as symbol 0
    for l in range(16, 1, -1):
        d = bytearray(base)
        q = base.index(b"\xFF\xC4") + 4
        tabs = []
        i = base.index(b"\xFF\xC4")
        while 0 <= i < len(d) - 4:
            L = struct.unpack(">H", d[i + 2:i + 4])[0]
            qq = i + 4
            while qq + 17 <= i + 2 + L:
                cnt = sum(d[qq + 1:qq + 17]); tabs.append((i, qq, cnt)); qq += 17 + cnt
            i = d.find(b"\xFF\xC4", i + 2 + L)
        i, q, cnt = [t for t in tabs if d[t[1]] == 0x10][0]         # AC table 0
        if d[q + l] == 0:
            continue
        first = q + 17 + sum(d[q + 1:q + l])                         # the first code word of length l: drop it
        d[q + l] -= 1
        del d[first]
        L = struct.unpack(">H", d[i + 2:i + 4])[0]
        d[i + 2:i + 4] = struct.pack(">H", L - 1)
        R.EVENTS.clear()
        try:
            r = R.decompress_jpeg_image_from_stream(bytes(d))
        except R.Undefined:
            continue
        if r is not None and R.EVENTS["three-argument huff_decode: empty tree slot"] + R.EVENTS["three-argument huff_decode: empty look_up2 entry"] > 0:
            files["ac_table_with_a_hole_r05"] = bytes(d)
            break
    assert "ac_table_with_a_hole_r05" in files
    # restart intervals: the marker comes early (the rest of the interval is decoded from 1-bits), a marker missing at the end of the data
    rst = enc(picture(48, 32, 3), quality=85, subsampling=2, restart_marker_blocks=2)
    pos = [i for i in range(rst.index(b"\xFF\xDA"), len(rst) - 1) if rst[i] == 0xFF and 0xD0 <= rst[i + 1] <= 0xD7]
    files["restart_marker_comes_early_r05"] = rst[:pos[1] - 9] + rst[pos[1]:]                                              # image
    files["restart_file_ends_inside_the_last_interval_r05"] = rst[:pos[1] + 6]                                             # image
    files["restart_file_ends_inside_the_first_interval_r05"] = rst[:pos[0] - 3]                                            # null: the marker process_restart finds is the padding's EOI
    files["restart_markers_without_dri_r05"] = rst[:rst.index(b"\xFF\xDD")] + rst[rst.index(b"\xFF\xDD") + 6:]             # image: the input stops at the first RSTn, the rest is 1-bits; find_eoi skips the markers
    return files


def describe(data):
    try:
        r = R.decompress_jpeg_image_from_stream(data)
    except R.Undefined as e:
        return {"verdict": "undefined", "why": str(e)}
    if r is None:
        return {"verdict": "null"}
    co = np.array([m[0] for m in r["mcus"]], np.int16).reshape(-1, 64)
    mz = np.array([z for m in r["mcus"] for z in m[1]], np.uint8)

    def fl(v):
        return "nan" if math.isnan(v) else "inf" if math.isinf(v) and v > 0 else "-inf" if math.isinf(v) else float(np.float32(v))
    return {"verdict": "image", "width": r["width"], "height": r["height"], "comps": r["actual_comps"], "scan_type": r["scan_type"],
            "pixel_aspect_ratio": fl(r["pixelAspectRatio"]), "dpi_y": fl(r["dotsPerInchY"]),
            "coeffs_sha256": hashlib.sha256(np.ascontiguousarray(co).tobytes()).hexdigest(), "max_zag_sha256": hashlib.sha256(mz.tobytes()).hexdigest(),
            "blocks": int(co.shape[0])}


def main():
    check = "--check" in sys.argv
    os.makedirs(OUT, exist_ok=True)
    files = build()
    if not check:
        for name, data in files.items():
            with open(os.path.join(OUT, name + ".jpg"), "wb") as fh:
                fh.write(data)
    expected = {}
    for n in sorted(os.listdir(OUT)):
        if n.endswith(".jpg"):
            expected[n] = describe(open(os.path.join(OUT, n), "rb").read())
    text = json.dumps(expected, indent=1, sort_keys=True) + "\n"
    path = os.path.join(OUT, "expected.json")
    if check:
        same = os.path.exists(path) and open(path).read() == text
        print("expected.json", "is up to date" if same else "DIFFERS from a fresh run")
        return 0 if same else 1
    open(path, "w").write(text)
    kinds = {}
    for v in expected.values():
        kinds[v["verdict"]] = kinds.get(v["verdict"], 0) + 1
    print("%d files (%d new), %d bytes: %s" % (len(expected), len(files), sum(len(d) for d in files.values()), kinds))
    return 0


if __name__ == "__main__":
    sys.exit(main())
