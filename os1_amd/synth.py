"""Seeded synthetic gray frames.  Pure numpy, bit-reproducible everywhere.

synth(seed, W, H): value noise at 3 octaves (lattice periods 64/16/4 px) scaled to [40,215],
plus W*H/2000 axis-aligned rectangles and half as many discs with random gray levels (hard edges
-> FAST corners at every pyramid level), plus +-4 uniform pixel noise.  RNG = counter-based
SplitMix64 with an explicit seed (no libc rand, no numpy Generator-version dependence).
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, stream=0):
    """n 64-bit values: output i is SplitMix64's (i+1)-th value for state0 = hash(seed, stream)."""
    with np.errstate(over='ignore'):
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream) * np.uint64(0xA0761D6478BD642F)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * _G)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(seed, n, stream):
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _value_noise(seed, stream, W, H, period):
    gw, gh = W // period + 2, H // period + 2
    lat = _uniform(seed, gw * gh, stream).reshape(gh, gw)
    ys, xs = np.arange(H) / period, np.arange(W) / period
    y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = lat[y0][:, x0]
    b = lat[y0][:, x0 + 1]
    c = lat[y0 + 1][:, x0]
    d = lat[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def synth(seed, W, H):
    """Return an (H, W) uint8 C-contiguous frame."""
    base = (0.5 * _value_noise(seed, 1, W, H, 64) + 0.3 * _value_noise(seed, 2, W, H, 16)
            + 0.2 * _value_noise(seed, 3, W, H, 4))
    img = 40.0 + 175.0 * base
    K = max(1, W * H // 2000)
    r = _uniform(seed, K * 5, 4).reshape(K, 5)
    for i in range(K):
        w = int(8 + r[i, 2] * 88)
        h = int(8 + r[i, 3] * 88)
        x = int(r[i, 0] * (W - 4))
        y = int(r[i, 1] * (H - 4))
        img[y:y + h, x:x + w] = np.floor(20 + r[i, 4] * 215)
    D = max(1, K // 2)
    r = _uniform(seed, D * 4, 5).reshape(D, 4)
    for i in range(D):
        rad = int(5 + r[i, 2] * 30)
        cx = int(r[i, 0] * W)
        cy = int(r[i, 1] * H)
        x0, x1, y0, y1 = max(cx - rad, 0), min(cx + rad + 1, W), max(cy - rad, 0), min(cy + rad + 1, H)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.ogrid[y0:y1, x0:x1]
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad
        img[y0:y1, x0:x1][m] = np.floor(20 + r[i, 3] * 215)
    noise = (splitmix64(seed, W * H, 6) % np.uint64(9)).astype(np.int64).reshape(H, W) - 4
    return np.ascontiguousarray(np.clip(np.floor(img).astype(np.int64) + noise, 0, 255).astype(np.uint8))


def shifted(frame, dx, dy, seed):
    """frame translated by (dx, dy) px with reflect fill and fresh +-4 noise (configs 3/4)."""
    H, W = frame.shape
    pad = max(abs(dx), abs(dy)) + 1
    p = np.pad(frame, pad, mode='reflect')
    out = p[pad - dy:pad - dy + H, pad - dx:pad - dx + W].astype(np.int64)
    noise = (splitmix64(seed, W * H, 7) % np.uint64(9)).astype(np.int64).reshape(H, W) - 4
    return np.ascontiguousarray(np.clip(out + noise, 0, 255).astype(np.uint8))
