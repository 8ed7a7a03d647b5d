"""Seeded synthetic inputs shared by the parity tests and bench.py."""
import numpy as np

from oracle_lib import PT, PT_CHANNELS, PT_DTYPE, PT_SIZE, PIXEL_TYPES  # noqa: F401


def f32_edge_values():
    """Values around every rounding / truncation boundary of the f32 -> u8/u16 row kernels."""
    one = np.float32(1.0)
    vals = [0.0, 1.0, 0.5, 1e-40, -1e-40, 1e-30, 0.999999, 1.0000001, 1.5, 2.0, -0.25, 3.4e38, -3.4e38, 1e10, -1e10,
            256.0, 65536.0, 8421504.0, 8421505.0, 32768.5, 3.0e9]
    out = [np.float32(v) for v in vals]
    for m in (255.0, 65535.0):
        ks = np.arange(0, int(m) + 1, 1 if m == 255.0 else 257, dtype=np.float64)
        for off in (0.0, 0.5, -0.5):
            base = ((ks + off) / m).astype(np.float32)
            out += list(base)
            out += list(np.nextafter(base, np.float32(2.0)))
            out += list(np.nextafter(base, np.float32(-1.0)))
    return np.array(out, np.float32)


def make_pixels(ptype, npx, rng, edge=True):
    """Return a (npx, channels) array of the type's dtype with adversarial + random content."""
    t = PT[ptype] if isinstance(ptype, str) else ptype
    ch, dt = PT_CHANNELS[t], PT_DTYPE[t]
    n = npx * ch
    if dt == np.uint8:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        if edge:
            k = min(n, 256)
            a[:k] = np.arange(k, dtype=np.uint8)
    elif dt == np.uint16:
        a = rng.integers(0, 65536, n, dtype=np.uint16)
        if edge:
            e = np.array([0, 1, 127, 128, 255, 256, 257, 32767, 32768, 65534, 65535], np.uint16)
            a[:min(n, e.size)] = e[:min(n, e.size)]
    else:
        a = rng.random(n, dtype=np.float32)
        if edge:
            e = f32_edge_values()
            k = min(n // 2, e.size)
            pos = rng.choice(n, k, replace=False)
            a[pos] = rng.permutation(e)[:k]
    a = a.reshape(npx, ch)
    if ch in (2, 4) and npx >= 8:        # alpha == 0 / alpha == max rows for the (un)premultiply branches
        a[1, ch - 1] = 0
        a[3, ch - 1] = np.array(1.0 if dt == np.float32 else np.iinfo(dt).max, dt)
        a[5, :] = 0
    return a


def pack_rows(pixels, width, height, pitch, flipped=False):
    """Lay (height*width, ch) pixels into a byte buffer with the given positive pitch.
    Returns (buffer uint8, first-scanline offset, signed pitch)."""
    row = pixels.reshape(height, -1).view(np.uint8).reshape(height, -1)
    rb = row.shape[1]
    buf = np.full(pitch * height + 32, 0xA5, np.uint8)
    for y in range(height):
        pos = (height - 1 - y) if flipped else y
        buf[pos * pitch: pos * pitch + rb] = row[y]
    if flipped:
        return buf, (height - 1) * pitch, -pitch
    return buf, 0, pitch


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def synth_rgb(width, height, index=0, alpha=False):
    """Deterministic smooth-plus-noise image: three low-frequency
    sinusoids per channel (amplitude <= 100) + uniform noise +-4, clamped to u8."""
    seed = splitmix64(0x9E3779B97F4A7C15 ^ index)
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    chans = []
    for c in range(3):
        v = np.full((height, width), 128.0, np.float32)
        for _ in range(3):
            fx, fy = rng.uniform(0.5, 6.0, 2) * 2 * np.pi / max(width, height)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(10, 33.3)
            v += amp * np.sin(xx * fx + yy * fy + ph)
        v += rng.integers(-4, 5, (height, width))
        chans.append(np.clip(v, 0, 255))
    if alpha:
        chans.append(np.clip(255.0 * (xx + yy) / max(1, width + height - 2), 0, 255))
    return np.stack(chans, -1).astype(np.uint8)


# ---------------------------------------------------------------- PNG stream builders (tests + bench)
def png_forward_filter(rows, fb, filters):
    """rows: (h, wb) uint8 packed sample bytes; fb: filter unit in bytes; filters: (h,) values 0..4.
    Returns the inflated-stream layout the de-filter kernels consume: per row 1 filter byte + wb bytes."""
    rows = np.ascontiguousarray(rows, np.uint8)
    h, wb = rows.shape
    cur = rows.astype(np.int16)
    a = np.zeros_like(cur); a[:, fb:] = cur[:, :-fb] if wb > fb else 0
    b = np.zeros_like(cur); b[1:] = cur[:-1]
    c = np.zeros_like(cur); c[1:, fb:] = cur[:-1, :-fb] if wb > fb else 0
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
    preds = [np.zeros_like(cur), a, b, (a + b) >> 1, paeth]
    f = np.asarray(filters, np.uint8).reshape(h)
    out = np.empty((h, wb + 1), np.uint8)
    out[:, 0] = f
    for t in range(5):
        m = f == t
        if m.any():
            out[m, 1:] = ((cur[m] - preds[t][m]) & 255).astype(np.uint8)
    return out.reshape(-1)


def png_heuristic_filters(rows, fb):
    """stb_image_write's choice (stb_image_write.d:387-406): per row the filter with the smallest sum of |signed residual|."""
    rows = np.ascontiguousarray(rows, np.uint8)
    h, wb = rows.shape
    best = np.zeros(h, np.uint8)
    best_cost = np.full(h, np.iinfo(np.int64).max, np.int64)
    for t in range(5):
        res = png_forward_filter(rows, fb, np.full(h, t, np.uint8)).reshape(h, wb + 1)[:, 1:].view(np.int8)
        cost = np.abs(res.astype(np.int64)).sum(axis=1)
        upd = cost < best_cost
        best[upd] = t
        best_cost[upd] = cost[upd]
    return best


def pack_samples(samples, depth):
    """samples: (h, n) integer sample values (n = width*channels) -> (h, ceil(n*depth/8)) packed bytes, PNG bit order / big-endian."""
    samples = np.asarray(samples)
    h, n = samples.shape
    if depth == 8:
        return samples.astype(np.uint8)
    if depth == 16:
        s = samples.astype(np.uint16)
        return np.stack([(s >> 8).astype(np.uint8), (s & 255).astype(np.uint8)], -1).reshape(h, n * 2)
    per = 8 // depth
    pad = (-n) % per
    s = np.concatenate([samples.astype(np.uint8), np.zeros((h, pad), np.uint8)], 1).reshape(h, -1, per)
    out = np.zeros(s.shape[:2], np.uint8)
    for k in range(per):
        out |= (s[:, :, k] & ((1 << depth) - 1)) << (8 - depth * (k + 1))
    return out


def write_png(samples, width, height, color, depth, filters=None, interlace=0, palette=None, trns=None, iphone=False,
              idat_split=3, extra_chunks=(), no_iend=False):
    """Minimal PNG writer for tests: samples (height, width*channels) ints; returns file bytes."""
    import struct
    import zlib
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    fb = 1 if depth < 8 else ch * (2 if depth == 16 else 1)
    rng = np.random.default_rng(width * 7919 + height)

    def stream(smp, w, h):
        rows = pack_samples(smp.reshape(h, w * ch), depth)
        f = filters if filters is not None and np.ndim(filters) == 0 else None
        ff = np.full(h, f, np.uint8) if f is not None else (rng.integers(0, 5, h).astype(np.uint8) if filters is None else np.asarray(filters)[:h])
        return png_forward_filter(rows, fb, ff).tobytes()

    smp = np.asarray(samples).reshape(height, width, ch)
    if not interlace:
        raw = stream(smp, width, height)
    else:
        xo, yo, xs, ys = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
        raw = b""
        for p in range(7):
            sub = smp[yo[p]::ys[p], xo[p]::xs[p]]
            if sub.shape[0] and sub.shape[1]:
                raw += stream(sub, sub.shape[1], sub.shape[0])
    if iphone:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z = co.compress(raw) + co.flush()
    else:
        z = zlib.compress(raw, 6)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n"
    if iphone:
        out += chunk(b"CgBI", b"\x50\x00\x20\x02")
    out += chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color, 0, 0, interlace))
    for t, d in extra_chunks:
        out += chunk(t, d)
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    n = max(1, len(z) // idat_split)
    for i in range(0, len(z), n):
        out += chunk(b"IDAT", z[i:i + n])
    if not no_iend:
        out += chunk(b"IEND", b"")
    return out


def qoi_encode(px, colorspace=0, force_ops=True):
    """QOI encoder written from the public format specification (qoiformat.org: 14-byte header, QOI_OP_RGB / RGBA / INDEX /
    DIFF / LUMA / RUN, 8-byte end marker) -- an independent producer of test streams, NOT taken from the reference.
    px: (h, w, 3|4) uint8."""
    h, w, ch = px.shape
    out = bytearray(b"qoif" + int(w).to_bytes(4, "big") + int(h).to_bytes(4, "big") + bytes([ch, colorspace]))
    index = [(0, 0, 0, 0)] * 64
    prev = (0, 0, 0, 255)
    run = 0
    flat = px.reshape(-1, ch).tolist()
    n = len(flat)
    for i, v in enumerate(flat):
        cur = (v[0], v[1], v[2], v[3] if ch == 4 else 255)
        if cur == prev:
            run += 1
            if run == 62 or i == n - 1:
                out.append(0xC0 | (run - 1)); run = 0
            continue
        if run:
            out.append(0xC0 | (run - 1)); run = 0
        hp = (cur[0] * 3 + cur[1] * 5 + cur[2] * 7 + cur[3] * 11) % 64
        if index[hp] == cur:
            out.append(hp)
        else:
            index[hp] = cur
            if cur[3] == prev[3]:
                dr, dg, db = ((cur[0] - prev[0] + 128) & 255) - 128, ((cur[1] - prev[1] + 128) & 255) - 128, ((cur[2] - prev[2] + 128) & 255) - 128
                dgr, dgb = dr - dg, db - dg
                if -2 <= dr <= 1 and -2 <= dg <= 1 and -2 <= db <= 1:
                    out.append(0x40 | (dr + 2) << 4 | (dg + 2) << 2 | (db + 2))
                elif -8 <= dgr <= 7 and -32 <= dg <= 31 and -8 <= dgb <= 7:
                    out += bytes([0x80 | (dg + 32), (dgr + 8) << 4 | (dgb + 8)])
                else:
                    out += bytes([0xFE, cur[0], cur[1], cur[2]])
            else:
                out += bytes([0xFF, cur[0], cur[1], cur[2], cur[3]])
        prev = cur
    out += bytes([0, 0, 0, 0, 0, 0, 0, 1])
    return bytes(out)


def sos_component_lists(data, lists=((1, 3, 2), (2, 1, 3), (3, 2, 1), (1, 2, 2), (1, 1, 3), (2, 2, 2), (1, 3, 3))):
    """A three-component JPEG with the component ids of its interleaved SOS headers rewritten (the table selectors stay where they are): files no
    encoder writes -- the scan header must list the components in frame order, each once -- but the reference reads them without a check
    (read_sos_marker jpegload.d:1466-1540, calc_mcu_block_order :3068-3088).  `lists` are positions in the frame (1-based).  -> [bytes]"""
    i = 2
    ids = None
    sos = []
    while i + 4 <= len(data):
        assert data[i] == 0xFF
        m = data[i + 1]
        seglen = (data[i + 2] << 8) | data[i + 3]
        if m in (0xC0, 0xC1, 0xC2):
            ids = [data[i + 10 + 3 * k] for k in range(data[i + 9])]
        if m == 0xDA:
            if data[i + 4] == 3:
                sos.append(i)
            j = i + 2 + seglen                                   # over the entropy-coded data to the next marker that is not RSTn / a stuffed byte
            while j + 1 < len(data) and not (data[j] == 0xFF and data[j + 1] not in (0x00, 0xFF) and not 0xD0 <= data[j + 1] <= 0xD7):
                j += 1
            i = j
            continue
        if m == 0xD9:
            break
        i += 2 + seglen
    assert ids is not None and len(ids) == 3 and sos
    out = []
    for lst in lists:
        for at in sos:
            b = bytearray(data)
            for k, c in enumerate(lst):
                b[at + 5 + 2 * k] = ids[c - 1]
            out.append(bytes(b))
    return out


def rst_positions(d):
    sos = d.index(b"\xff\xda")
    return [i for i in range(sos, len(d) - 1) if d[i] == 0xFF and 0xD0 <= d[i + 1] <= 0xD7]


def leftover_variants(d, which):
    """octets between the end of a restart interval's data and its RSTn marker (`which`: the marker's number in the file)"""
    p = rst_positions(d)[which]
    ins = [b"\x55" * 8, b"\x55" * 6 + b"\xff\x00\x55", b"\x11" * 1600, b"\x11" * 1400, b"\xff" * 3, b"\x11" * 1400 + b"\xff" * 200, b"\x00" * 3 + b"\xff" * 1600]
    ins += [b"\x11" * n for n in range(1526, 1542)]                       # around the 1536 reads process_restart allows itself
    return [d[:p] + v + d[p:] for v in ins]
