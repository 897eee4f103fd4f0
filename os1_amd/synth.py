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


def synth_vocabulary(seed, k=10, L=6, scoring=0, weighting=0, stop_fraction=0.02):
    """A synthetic ORB vocabulary in the fork's binary file layout (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
    1563-1640): 4 header bytes (k, L, scoring, weighting) followed by one 45-byte record per node in id order
    {int32 parent, u8 isLeaf, u8 descriptor[32], f64 weight}.  The tree is complete (k^l nodes at level l, written
    level by level); a child's descriptor is its parent's with a level-dependent number of random bit flips, so a
    descent is meaningful; leaf weights look like idf values (log(N/Ni)) and a few are 0 ("stopped" words)."""
    rng = np.random.default_rng(seed)
    level_desc = [rng.integers(0, 256, (k, 32), dtype=np.uint8)]
    parents = [np.zeros(k, np.int32)]
    first_id = 1
    for lvl in range(2, L + 1):
        prev = level_desc[-1]
        n = prev.shape[0] * k
        flips = max(4, 128 >> (lvl - 1))
        bits = np.unpackbits(np.repeat(prev, k, axis=0), axis=1)
        pos = rng.integers(0, 256, (n, flips))
        np.put_along_axis(bits, pos, 1 - np.take_along_axis(bits, pos, axis=1), axis=1)
        level_desc.append(np.packbits(bits, axis=1))
        parents.append(first_id + np.repeat(np.arange(prev.shape[0], dtype=np.int32), k))
        first_id += prev.shape[0]
    desc = np.concatenate(level_desc)
    parent = np.concatenate(parents)
    n_nodes = desc.shape[0]
    n_leaves = level_desc[-1].shape[0]
    leaf = np.zeros(n_nodes, np.uint8)
    leaf[n_nodes - n_leaves:] = 1
    weight = np.zeros(n_nodes, np.float64)
    w = np.log(1000.0 / rng.integers(1, 900, n_leaves))
    w[rng.random(n_leaves) < stop_fraction] = 0.0
    weight[n_nodes - n_leaves:] = w
    rec = np.zeros((n_nodes, 45), np.uint8)
    rec[:, 0:4] = parent.astype('<i4').view(np.uint8).reshape(-1, 4)
    rec[:, 4] = leaf
    rec[:, 5:37] = desc
    rec[:, 37:45] = weight.astype('<f8').view(np.uint8).reshape(-1, 8)
    return bytes([k, L, scoring, weighting]) + rec.tobytes()
