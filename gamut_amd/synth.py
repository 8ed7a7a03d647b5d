"""Synthetic, seeded workloads for bench.py / smoke(), built on the GPU with torch.

torch is plumbing here (device memory + fast input synthesis); nothing in this file is
on the measured path.  The JPEG batch is produced the way an encoder would: smooth-plus-
noise RGB -> YCbCr -> 4:2:0 -> forward DCT -> quantise with the Annex-K tables at q=90
-> de-quantise -> int16, laid out in MCU order, i.e. exactly the dense coefficient form
decode_next_row (jpegload.d:2405-2525) hands to transform_mcu_expand.
"""
import math

import torch

# ITU T.81 Annex K.1 / K.2 (natural order)
_LUMA_Q = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
           18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]
_CHROMA_Q = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99]


def _qtable(base, quality, device):
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality      # libjpeg jpeg_quality_scaling
    q = (torch.tensor(base, dtype=torch.float32, device=device) * scale + 50).div(100, rounding_mode="floor").clamp(1, 255)
    return q.reshape(8, 8)


def _dct_matrix(device):
    k = torch.arange(8, dtype=torch.float64, device=device)
    c = torch.cos((2 * k[None, :] + 1) * k[:, None] * math.pi / 16) * 0.5
    c[0, :] *= 1 / math.sqrt(2)
    return c.to(torch.float32)


def synth_rgb_batch(n, width, height, device, seed):
    """(n, 3, height, width) float32 in [0,255]: 3 low-frequency sinusoids per channel (amplitude <= 100 in total) + noise +-4."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    yy = torch.arange(height, dtype=torch.float32, device=device)[None, None, :, None]
    xx = torch.arange(width, dtype=torch.float32, device=device)[None, None, None, :]
    img = torch.full((n, 3, height, width), 128.0, device=device)
    for _ in range(3):
        fx = (torch.rand((n, 3, 1, 1), device=device, generator=g) * 5.5 + 0.5) * (2 * math.pi / max(width, height))
        fy = (torch.rand((n, 3, 1, 1), device=device, generator=g) * 5.5 + 0.5) * (2 * math.pi / max(width, height))
        ph = torch.rand((n, 3, 1, 1), device=device, generator=g) * (2 * math.pi)
        amp = torch.rand((n, 3, 1, 1), device=device, generator=g) * 23.3 + 10.0
        img += amp * torch.sin(xx * fx + yy * fy + ph)
    img += torch.randint(-4, 5, img.shape, device=device, generator=g).to(torch.float32)
    return img.clamp_(0, 255).floor_()


def jpeg_coeff_batch(n, width, height, device, seed=0, quality=90, chunk=16, scan_type=4):
    """Dense de-quantised coefficients of n synthetic images: int16 tensor (n, MY*MX*blocks_per_mcu, 64), blocks in MCU
    order.  scan_type (jpgd's): 4 = 4:2:0 (default), 3 = 4:4:0 (H1V2), 2 = 4:2:2 (H2V1), 1 = 4:4:4 (H1V1), 0 = grey."""
    hs, vs = {0: (1, 1), 1: (1, 1), 2: (2, 1), 3: (1, 2), 4: (2, 2)}[scan_type]
    my, mx = (height + 8 * vs - 1) // (8 * vs), (width + 8 * hs - 1) // (8 * hs)
    hp, wp = my * 8 * vs, mx * 8 * hs
    nb = hs * vs + (2 if scan_type else 0)
    d = _dct_matrix(device)
    ql, qc = _qtable(_LUMA_Q, quality, device), _qtable(_CHROMA_Q, quality, device)
    out = torch.empty((n, my * mx * nb, 64), dtype=torch.int16, device=device)
    for i0 in range(0, n, chunk):
        c = min(chunk, n - i0)
        rgb = synth_rgb_batch(c, width, height, device, seed * 1000003 + i0)
        rgb = torch.nn.functional.pad(rgb, (0, wp - width, 0, hp - height), mode="replicate")
        r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
        y = 0.299 * r + 0.587 * g + 0.114 * b - 128.0
        cb = -0.168736 * r - 0.331264 * g + 0.5 * b
        cr = 0.5 * r - 0.418688 * g - 0.081312 * b
        if hs * vs > 1:
            cb = torch.nn.functional.avg_pool2d(cb[:, None], (vs, hs))[:, 0]
            cr = torch.nn.functional.avg_pool2d(cr[:, None], (vs, hs))[:, 0]

        def fdct_quant(p, q):
            cc, h, w = p.shape
            blk = p.reshape(cc, h // 8, 8, w // 8, 8).permute(0, 1, 3, 2, 4)            # (c, by, bx, 8, 8)
            f = d @ blk @ d.T
            return (torch.round(f / q) * q).clamp_(-32768, 32767).to(torch.int16)

        yb = fdct_quant(y, ql).reshape(c, my, vs, mx, hs, 64).permute(0, 1, 3, 2, 4, 5).reshape(c, my, mx, hs * vs, 64)
        parts = [yb]
        if scan_type:
            parts += [fdct_quant(cb, qc).reshape(c, my, mx, 1, 64), fdct_quant(cr, qc).reshape(c, my, mx, 1, 64)]
        out[i0:i0 + c] = torch.cat(parts, dim=3).reshape(c, my * mx * nb, 64)
    return out


# zig-zag scan (ITU T.81 figure 5): _ZAG[k] = natural index of the k-th coefficient of the scan
_ZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
        49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def jpeg_max_zag(coeffs, chunk=16):
    """m_mcu_block_max_zag of a dense coefficient batch, as decode_next_row leaves it for a file an encoder wrote
    (jpegload.d:2432-2512): the scan position behind a block's last non-zero coefficient (EOB follows it at once), 1 for a
    DC-only or empty block, 64 when coefficient 63 is set.  uint8 tensor (n, blocks)."""
    n, nblk, _ = coeffs.shape
    rank = torch.empty(64, dtype=torch.uint8, device=coeffs.device)
    rank[torch.tensor(_ZAG, device=coeffs.device)] = torch.arange(1, 65, dtype=torch.uint8, device=coeffs.device)
    out = torch.empty((n, nblk), dtype=torch.uint8, device=coeffs.device)
    for i0 in range(0, n, chunk):
        nz = coeffs[i0:i0 + chunk] != 0
        out[i0:i0 + chunk] = (nz.to(torch.uint8) * rank).amax(dim=2).clamp_(min=1)
    return out


def photo_rgb(width, height, seed):
    """One synthetic photograph, uint8 (height, width, 3) on the host: a 1/f amplitude spectrum (the statistics of natural
    scenes) shaped into a few soft-edged regions of different tint and brightness (sky / ground / objects: flat or gently
    graded areas next to textured ones), defocused areas, mild sensor noise.  What matters for the decoder is the population
    of coefficient blocks a JPEG encoder makes of it at q 75-90: most smooth-area blocks end within the first ten zig-zag
    positions, textured ones run to the end -- unlike the smooth-plus-noise generator, whose noise fills every block."""
    import numpy as np
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(height)[:, None]; fx = np.fft.rfftfreq(width)[None, :]
    f = np.sqrt(fx * fx + fy * fy); f[0, 0] = 1.0

    def field(alpha, sigma):
        spec = (rng.standard_normal((height, fx.shape[1])) + 1j * rng.standard_normal((height, fx.shape[1]))) / f ** alpha
        spec[0, 0] = 0
        v = np.fft.irfft2(spec, s=(height, width))
        return v / (v.std() + 1e-9) * sigma
    base = field(2.6, 1.0)                                              # large-scale layout: sky / ground / defocused background
    texture = field(1.0, 1.0) + 0.6 * field(0.4, 1.0)                   # fine detail of the in-focus parts, 1/f + grain
    m = field(2.2, 1.0)
    detail_mask = 1.0 / (1.0 + np.exp(-(m - np.quantile(m, 0.55)) * 14.0))   # ~ 45 % of the frame is in focus / textured
    yy = np.linspace(0, 1, height)[:, None]
    lum = 128 + 50 * base + 30 * (0.5 - yy) + 20 * texture * detail_mask
    # a few hard-edged objects (occlusion boundaries): constant offsets inside random ellipses
    gy, gx = np.mgrid[0:height, 0:width]
    for _ in range(6):
        cy, cx = rng.uniform(0, height), rng.uniform(0, width)
        ry, rx = rng.uniform(40, height / 3), rng.uniform(40, width / 4)
        lum += rng.uniform(-35, 35) * ((((gy - cy) / ry) ** 2 + ((gx - cx) / rx) ** 2) < 1.0)
    img = np.empty((height, width, 3), np.float64)
    tint = field(2.4, 1.0)
    img[:, :, 0] = lum + 18 * tint
    img[:, :, 1] = lum - 4 * tint
    img[:, :, 2] = lum - 22 * tint + 10 * (0.5 - yy)
    img += rng.standard_normal(img.shape) * 0.6                         # sensor noise after in-camera denoising
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def png_raw_batch(n, width, height, device, seed=0, policy="heuristic", chunk=4, channels=4):
    """Inflated (post-zlib) filtered streams of n synthetic 8-bit images (channels: 4 RGBA, 3 RGB, 2 grey+alpha, 1 grey):
    uint8 tensor (n, height*(width*channels+1)).
    Pixels: the smooth-plus-noise generator + a smooth alpha ramp.  Per-row filter type:
    "heuristic" = stb_image_write's min sum |residual| choice (stb_image_write.d:387-406), "random" = uniform 0..4,
    or an int 0..4 for a single filter type.  Also returns the original pixels' checksums for a round-trip check."""
    ch = int(channels)
    wb = width * ch
    out = torch.empty((n, height, wb + 1), dtype=torch.uint8, device=device)
    sums = torch.empty((n,), dtype=torch.int64, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + 17)
    yy = torch.arange(height, dtype=torch.float32, device=device)[None, :, None]
    xx = torch.arange(width, dtype=torch.float32, device=device)[None, None, :]
    alpha = (255.0 * (xx + yy) / max(1, width + height - 2)).clamp(0, 255).floor()
    for i0 in range(0, n, chunk):
        c = min(chunk, n - i0)
        rgb = synth_rgb_batch(c, width, height, device, seed * 1000003 + i0)            # (c,3,h,w)
        planes = torch.cat([rgb, alpha.expand(c, height, width)[:, None]], 1)                        # (c,4,h,w)
        planes = planes[:, {4: [0, 1, 2, 3], 3: [0, 1, 2], 2: [1, 3], 1: [1]}[ch]]
        px = planes.permute(0, 2, 3, 1).reshape(c, height, wb).to(torch.int16)
        sums[i0:i0 + c] = px.to(torch.int64).sum(dim=(1, 2))
        a = torch.zeros_like(px); a[:, :, ch:] = px[:, :, :-ch]
        b = torch.zeros_like(px); b[:, 1:] = px[:, :-1]
        cc = torch.zeros_like(px); cc[:, 1:, ch:] = px[:, :-1, :-ch]
        p = a + b - cc
        pa, pb, pc = (p - a).abs(), (p - b).abs(), (p - cc).abs()
        paeth = torch.where((pa <= pb) & (pa <= pc), a, torch.where(pb <= pc, b, cc))
        del p, pa, pb, pc
        preds = [torch.zeros_like(px), a, b, (a + b) >> 1, paeth]
        res = [((px - q) & 255).to(torch.uint8) for q in preds]
        del preds, a, b, cc, paeth
        if policy == "heuristic":
            cost = torch.stack([r.view(torch.int8).to(torch.int32).abs().sum(dim=2) for r in res], 0)     # (5,c,h)
            f = cost.argmin(dim=0)
        elif policy == "random":
            f = torch.randint(0, 5, (c, height), device=device, generator=g)
        else:
            f = torch.full((c, height), int(policy), device=device, dtype=torch.int64)
        sel = torch.stack(res, 0).gather(0, f[None, :, :, None].expand(1, c, height, wb))[0]
        out[i0:i0 + c, :, 0] = f.to(torch.uint8)
        out[i0:i0 + c, :, 1:] = sel
        del res, sel
    return out.reshape(n, height * (wb + 1)), sums


def qoi_encode(px, colorspace=0):
    """QOI stream of a (h, w, 3|4) uint8 numpy image, vectorised (numpy) so that 1080p inputs for bench.py take a fraction
    of a second.  Written from the public format specification (qoiformat.org); the op choice is the specification
    encoder's (RUN, then INDEX, DIFF, LUMA, RGB / RGBA).  Every pixel refreshes the 64-entry table whatever op codes it, so
    "the table holds this value" is a property of the pixel sequence alone: the latest earlier pixel with the same hash
    must have the same value."""
    import numpy as np
    h, w, ch = px.shape
    v = px.reshape(-1, ch)
    n = v.shape[0]
    rgba = np.empty((n, 4), np.uint8)
    rgba[:, :3] = v[:, :3]
    rgba[:, 3] = v[:, 3] if ch == 4 else 255
    val = rgba.view("<u4").reshape(-1)
    prev = np.empty_like(rgba)
    prev[0] = (0, 0, 0, 255)
    prev[1:] = rgba[:-1]
    eq = val == prev.view("<u4").reshape(-1)
    c = rgba.astype(np.int32)
    hsh = ((c[:, 0] * 3 + c[:, 1] * 5 + c[:, 2] * 7 + c[:, 3] * 11) & 63).astype(np.uint8)
    order = np.argsort(hsh, kind="stable")
    sh = hsh[order]
    same = sh[1:] == sh[:-1]
    last = np.full(n, -1, np.int64)                      # latest earlier pixel with the same hash
    last[order[1:][same]] = order[:-1][same]
    in_table = np.where(last >= 0, val[np.maximum(last, 0)] == val, val == 0)      # the table starts out all zero
    d = ((c[:, :3] - prev[:, :3].astype(np.int32) + 128) & 255) - 128
    dr, dg, db = d[:, 0], d[:, 1], d[:, 2]
    same_a = rgba[:, 3] == prev[:, 3]
    small = same_a & (np.abs(d + 0.5) < 2).all(axis=1)                               # -2..1
    luma = same_a & (dg >= -32) & (dg <= 31) & (dr - dg >= -8) & (dr - dg <= 7) & (db - dg >= -8) & (db - dg <= 7)
    ops = np.zeros((n, 5), np.uint8)
    lens = np.zeros(n, np.int64)
    # runs: pixel number o (0-based) of a group of L repeats carries a RUN op if it completes 62 repeats or ends the group
    idx = np.arange(n)
    start = eq & ~np.concatenate([[False], eq[:-1]])
    gstart = np.maximum.accumulate(np.where(start, idx, 0))
    o = idx - gstart
    ends = eq & ~np.concatenate([eq[1:], [False]])
    full = eq & ((o + 1) % 62 == 0)
    ops[full, 0] = 0xC0 | 61
    part = ends & ~full
    ops[part, 0] = 0xC0 | (o[part] % 62)
    lens[full | part] = 1
    ne = ~eq
    m = ne & in_table
    ops[m, 0] = hsh[m]; lens[m] = 1
    rest = ne & ~in_table
    m = rest & small
    ops[m, 0] = (0x40 | (dr[m] + 2) << 4 | (dg[m] + 2) << 2 | (db[m] + 2)).astype(np.uint8); lens[m] = 1
    m = rest & ~small & luma
    ops[m, 0] = (0x80 | (dg[m] + 32)).astype(np.uint8); ops[m, 1] = ((dr[m] - dg[m] + 8) << 4 | (db[m] - dg[m] + 8)).astype(np.uint8); lens[m] = 2
    m = rest & ~small & ~luma & same_a
    ops[m, 0] = 0xFE; ops[m, 1:4] = rgba[m, :3]; lens[m] = 4
    m = rest & ~same_a
    ops[m, 0] = 0xFF; ops[m, 1:5] = rgba[m]; lens[m] = 5
    body = ops[np.arange(5)[None, :] < lens[:, None]]
    return (b"qoif" + int(w).to_bytes(4, "big") + int(h).to_bytes(4, "big") + bytes([ch, colorspace]) + body.tobytes() + bytes([0, 0, 0, 0, 0, 0, 0, 1]))
