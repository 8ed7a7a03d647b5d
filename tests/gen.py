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
