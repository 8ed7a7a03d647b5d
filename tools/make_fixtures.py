#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the BUILD container (needs Pillow; the reference tree is only used as the
source of the data files already copied to tests/golden/ref_images/).  What it
writes is data only:
  tests/golden/jpeg/*.jpg     small baseline JPEGs written by Pillow (libjpeg-turbo)
                              from the seeded synthetic image of tests/gen.py
  tests/golden/golden.json    sha256 of expected outputs:
      "pillow"  -- decodes by an INDEPENDENT decoder (Pillow): PNG pixels (lossless,
                   so any correct decoder is an oracle) and H1V1 / grey JPEG pixels
                   (equal to the reference's arithmetic with the IDCT pass order
                   swapped -- the oracle's test-only `colfirst` mode)
      "frozen"  -- outputs of the CPU oracle in the reference's own pass order,
                   frozen once the independent checks passed (regression pins)
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image, ImageFile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import oracle_lib as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
ImageFile.LOAD_TRUNCATED_IMAGES = True


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(os.path.join(G, "jpeg"), exist_ok=True)
    golden = {"pillow": {}, "frozen": {}, "meta": {}}

    # ---- JPEG fixtures -----------------------------------------------------
    specs = [
        ("cfg1_640x480_420_q90", 640, 480, dict(quality=90, subsampling=2), 0),      # BASELINE.json config 1
        ("s_131x97_444", 131, 97, dict(quality=90, subsampling=0), 1),
        ("s_131x97_422", 131, 97, dict(quality=85, subsampling=1), 2),
        ("s_131x97_420", 131, 97, dict(quality=75, subsampling=2), 3),
        ("s_131x97_420_rst", 131, 97, dict(quality=92, subsampling=2, restart_marker_blocks=3), 4),
        ("s_131x97_420_opt", 131, 97, dict(quality=60, subsampling=2, optimize=True), 5),
        # H1V2 (4:4:0): Pillow cannot write it; a 4:2:2 stream of a square image has the same MCU count and
        # block order, so patching the SOF sampling byte 0x21 -> 0x12 gives a valid (scrambled-looking) H1V2 file
        ("s_128x128_440_patched", 128, 128, dict(quality=88, subsampling=1), 6),
        ("s_97x131_gray", 97, 131, dict(quality=85), 7),
        ("s_16x16_420", 16, 16, dict(quality=95, subsampling=2), 8),
        ("s_1x1_444", 1, 1, dict(quality=95, subsampling=0), 9),
        # progressive twins (SOF2; libjpeg's default script: DC first, AC first, AC refine, DC refine scans) of the files
        # above: same source pixels and quantisation => the SAME quantised coefficients, only the entropy coding differs
        ("p_131x97_444", 131, 97, dict(quality=90, subsampling=0, progressive=True), 1),
        ("p_131x97_422", 131, 97, dict(quality=85, subsampling=1, progressive=True), 2),
        ("p_131x97_420", 131, 97, dict(quality=75, subsampling=2, progressive=True), 3),
        ("p_131x97_420_rst", 131, 97, dict(quality=92, subsampling=2, restart_marker_blocks=3, progressive=True), 4),
        ("p_97x131_gray", 97, 131, dict(quality=85, progressive=True), 7),
        ("p_16x16_420", 16, 16, dict(quality=95, subsampling=2, progressive=True), 8),
        ("p_1x1_444", 1, 1, dict(quality=95, subsampling=0, progressive=True), 9),
    ]
    for name, w, h, kw, idx in specs:
        img = gen.synth_rgb(w, h, idx)
        bio = io.BytesIO()
        try:
            if "gray" in name:
                Image.fromarray(img[:, :, 1]).save(bio, "JPEG", **kw)
            else:
                Image.fromarray(img).save(bio, "JPEG", **kw)
        except Exception as e:      # e.g. a Pillow without "4:4:0"
            print("skip", name, e)
            continue
        data = bio.getvalue()
        if "440_patched" in name:
            i = data.index(b"\xff\xc0")
            assert data[i + 11] == 0x21
            data = data[:i + 11] + b"\x12" + data[i + 12:]
        with open(os.path.join(G, "jpeg", name + ".jpg"), "wb") as f:
            f.write(data)
        d = O.DecodedJpeg(data)
        golden["meta"][name] = dict(width=d.width, height=d.height, comps=d.comps, scan_type=d.scan_type,
                                    bytes=len(data), coeff_sha=sha(d.coeffs), max_zag_sha=sha(d.max_zag))
        if name.startswith("p_"):
            assert b"\xff\xc2" in data
            twin = "s_" + name[2:]
            assert golden["meta"][twin]["coeff_sha"] == golden["meta"][name]["coeff_sha"], name
            golden["meta"][name]["baseline_twin"] = twin
        pil = np.array(Image.open(io.BytesIO(data))) if "patched" not in name else None
        if d.scan_type in (O.JPGD_GRAYSCALE, O.JPGD_YH1V1):
            rc = 1 if d.comps == 1 else 3
            cf = O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, d.max_zag, rc, colfirst=True)
            assert np.array_equal(cf.reshape(pil.shape), pil), name
            golden["pillow"][name + ":colfirst"] = sha(pil)
        for rc in (1, 3, 4):
            out = O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, d.max_zag, rc)
            golden["frozen"][f"{name}:comps{rc}"] = sha(out)
        print(name, len(data), "bytes", d.scan_type)

    # the reference's own JPEG fixture
    data = open(os.path.join(G, "ref_images", "issue35.jpg"), "rb").read()
    d = O.DecodedJpeg(data)
    pil = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    cf = O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, d.max_zag, 3, colfirst=True)
    assert np.array_equal(cf.reshape(pil.shape), pil)
    golden["pillow"]["issue35:colfirst"] = sha(pil)
    for rc in (1, 3, 4):
        golden["frozen"][f"issue35:comps{rc}"] = sha(O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, d.max_zag, rc))

    # ---- PNG fixtures (the reference's test images) --------------------------
    for f in ["issue65.png", "vst3-compatible.png", "issue76.png", "issue92-no-IEND.png", "issue92-truncated-in-CRC.png"]:
        data = open(os.path.join(G, "ref_images", f), "rb").read()
        im = Image.open(io.BytesIO(data))
        im.load()
        pa = np.array(im)
        if f == "issue76.png":
            pa = pa.astype(np.uint16)
        golden["pillow"][f] = dict(sha=sha(pa), shape=list(pa.shape), dtype=str(pa.dtype))
    for f in ["issue51cgbi.png", "issue51cgbi2.png"]:       # Pillow rejects CgBI: frozen oracle output only
        arr, n = O.stbi_load(open(os.path.join(G, "ref_images", f), "rb").read())
        golden["frozen"][f] = dict(sha=sha(arr), shape=list(arr.shape), comps=n)

    with open(os.path.join(G, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote golden.json")


if __name__ == "__main__":
    main()
