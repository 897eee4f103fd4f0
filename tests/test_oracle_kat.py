"""First-principles known-answer tests that pin the CPU oracle's restated primitives
(the reference ships no tests, so these are the only pins -- 'parity unpinned')."""
import hashlib
import math

import numpy as np
import pytest

from oracle.pyoracle import KP_DTYPE, OracleExtractor


def test_brief_pattern_hash():
    for f in ('oracle/brief_pattern.inc', 'os1_amd/csrc/brief_pattern.inc'):
        import os, re
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        txt = open(os.path.join(root, f)).read()
        txt = re.sub(r'//.*', '', txt)
        v = np.array([int(x) for x in re.findall(r'-?\d+', txt)], np.int8)
        assert v.size == 1024
        assert hashlib.sha256(v.tobytes()).hexdigest() == \
            '2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023'
        # first and last rows (reference src/ORBextractor.cc:184,439)
        assert v[:4].tolist() == [8, -3, 9, 5] and v[-4:].tolist() == [-1, -6, 0, -11]
        assert np.abs(v).max() == 13


def test_tables(oracle):
    # SURVEY.md s4 / Appendix A.1 (derived from ORBextractor.cc:447-501)
    exp = {1000: [217, 181, 151, 126, 105, 87, 73, 60],
           2000: [434, 362, 302, 251, 209, 175, 145, 122],
           4000: [869, 724, 603, 503, 419, 349, 291, 242]}
    for n, want in exp.items():
        t = OracleExtractor(n, 1.2, 8, 20, 7, oracle).tables()
        assert t['nfeat'].tolist() == want
        assert t['umax'].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
        sf = np.float32(1.0)
        for i in range(8):
            assert t['sf'][i] == sf
            assert t['isf'][i] == np.float32(1.0) / sf
            assert t['s2'][i] == sf * sf
            sf = np.float32(sf * np.float32(1.2))
    # patch pixel count of the circular IC-angle patch
    um = t['umax']
    assert (2 * um[0] + 1) + 2 * sum(2 * int(u) + 1 for u in um[1:]) == 749


def test_level_sizes(oracle):
    want = {(640, 480): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
            (1920, 1080): [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362),
                           (536, 301)]}
    for (w, h), sizes in want.items():
        ex = OracleExtractor(100, 1.2, 8, 20, 7, oracle)
        ex.extract(np.zeros((h, w), np.uint8))
        for l, (lw, lh) in enumerate(sizes):
            assert ex.level(l).shape == (lh, lw)


def test_cv_round_half_even(oracle):
    f = oracle.L.orc_cv_round_f
    assert [f(x) for x in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_gauss_kernel(oracle):
    k = oracle.gauss_kernel(7, 2.0)
    assert k.tolist() == [18, 34, 48, 56, 48, 34, 18] and k.sum() == 256


def test_blur_constant_and_impulse(oracle):
    img = np.full((40, 50), 137, np.uint8)
    assert (oracle.gauss7(img) == 137).all()          # kernel sums to 256 => identity on constants
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = oracle.gauss7(imp)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    want = ((np.outer(k, k) * 255 + 32768) >> 16).astype(np.uint8)
    assert (out[7:14, 7:14] == want).all() and out.sum() == want.sum()
    # reflect-101 at the border: impulse at column 0 mirrors onto columns 1..3
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 0] = 255
    out = oracle.gauss7(imp)
    row = (k[3] * 255 * k[3:] + 32768) >> 16
    assert out[10, :4].tolist() == row.tolist()
    imp[10, 0] = 0
    imp[10, 1] = 255   # column 1 is its own reflection partner for x=-1 -> weights k[2]+k[4] at x=0
    out = oracle.gauss7(imp)
    assert out[10, 0] == ((k[3] * 255 * (k[2] + k[4]) + 32768) >> 16)  # x=0: taps -3..3 reflect to 3,2,1,0,1,2,3


def test_blur_random_vs_numpy(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    p = np.pad(img.astype(np.int64), 3, mode='reflect')   # numpy 'reflect' == BORDER_REFLECT_101
    h = sum(k[t] * p[:, t:t + 53] for t in range(7))
    v = sum(k[t] * h[t:t + 37, :] for t in range(7))
    want = ((v + 32768) >> 16).astype(np.uint8)
    assert (oracle.gauss7(img) == want).all()


def test_legacy_gaussian_variant(oracle):
    """Variant 1 = GaussianBlur of OpenCV 4.0.0 - 4.1.0 (the era the reference dates from, README.md:18): every 8.8 tap rounded on its
    own, [18,34,49,55,49,34,18] = cvRound(k_i * 256), sum 257; exact integer sums (257 * 255 = 65 535 still fits the 16-bit row pass)
    and ONE saturation at the final cast.  Against an independent numpy restatement, on noise, on a bright image where the cast
    saturates, and on constants (a constant c maps to min(255, (c * 257 * 257 + 32768) >> 16), not to c)."""
    k = oracle.gauss_kernel(7, 2.0, 1)
    x = np.arange(7) - 3.0
    g = np.exp(-x * x / 8.0)
    assert k.tolist() == [18, 34, 49, 55, 49, 34, 18] == np.rint(g / g.sum() * 256).astype(int).tolist() and k.sum() == 257
    assert oracle.gauss_kernel(7, 2.0, 0).tolist() == [18, 34, 48, 56, 48, 34, 18]
    k = k.astype(np.int64)

    def ref(img):
        h_, w_ = img.shape
        p = np.pad(img.astype(np.int64), 3, mode='reflect')
        h = sum(k[t] * p[:, t:t + w_] for t in range(7))
        assert h.max() <= 65535
        v = sum(k[t] * h[t:t + h_, :] for t in range(7))
        return np.minimum((v + 32768) >> 16, 255).astype(np.uint8)
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (41, 57), dtype=np.uint8)
    assert (oracle.gauss7(img, 1) == ref(img)).all()
    assert (oracle.gauss7(img, 1) != oracle.gauss7(img, 0)).mean() > 0.3            # the two releases really differ
    bright = np.clip(rng.integers(235, 300, (41, 57)), 0, 255).astype(np.uint8)        # mostly 255: (sum + 32768) >> 16 reaches 256, 257
    out = oracle.gauss7(bright, 1)
    assert (out == ref(bright)).all() and (out == 255).any()
    p = np.pad(bright.astype(np.int64), 3, mode='reflect')
    h = sum(k[t] * p[:, t:t + 57] for t in range(7))
    v = sum(k[t] * h[t:t + 41, :] for t in range(7))
    assert ((v + 32768) >> 16).max() >= 256                                          # the saturation is exercised, not just present
    for c in (0, 1, 100, 137, 254, 255):
        assert (oracle.gauss7(np.full((20, 20), c, np.uint8), 1) == min(255, (c * 257 * 257 + 32768) >> 16)).all()


def test_resize_constant_ramp_and_numpy(oracle):
    img = np.full((60, 72), 91, np.uint8)
    assert (oracle.resize(img, 60, 50) == 91).all()
    assert (oracle.resize(img, 72, 60) == img).all()      # identity size => exact copy
    # independent numpy restatement of the fixed-point bilinear
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    for (dw, dh) in [(53, 40), (64, 48), (33, 17), (100, 70)]:
        sw, sh = 64, 48

        def coeffs(d, s):
            scale = 1.0 / (d / s)
            f = ((np.arange(d) + 0.5) * scale - 0.5).astype(np.float32)
            i = np.floor(f).astype(np.int64)
            f = (f - i.astype(np.float32)).astype(np.float32)
            return i, f
        ix, fx = coeffs(dw, sw)
        fx[ix < 0] = 0
        ix[ix < 0] = 0
        fx[ix >= sw - 1] = 0
        ix[ix >= sw - 1] = sw - 1
        a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        iy, fy = coeffs(dh, sh)
        b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
        b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
        y0 = np.clip(iy, 0, sh - 1)
        y1 = np.clip(iy + 1, 0, sh - 1)
        S = src.astype(np.int64)
        x1 = np.minimum(ix + 1, sw - 1)
        H = S[:, ix] * a0 + S[:, x1] * a1
        want = ((((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
        assert (oracle.resize(src, dw, dh) == want).all(), (dw, dh)


def _ring_img(center, ring_vals):
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    img = np.full((7, 7), center, np.uint8)
    for k in range(16):
        img[3 + dy[k], 3 + dx[k]] = ring_vals[k]
    return img


def test_fast_arcs(oracle):
    c = 100
    for start in range(16):
        # exactly 9 contiguous brighter pixels (by 30) => corner at t=20, score 29; 8 => not a corner
        for n, is_corner in ((9, True), (8, False)):
            ring = [c] * 16
            for j in range(n):
                ring[(start + j) % 16] = c + 30
            k = oracle.fast9(_ring_img(c, ring), 20, nms=True)
            if is_corner:
                assert k.tolist() == [[3, 3, 29]], (start, n, k)
            else:
                assert len(k) == 0
        # darker arc
        ring = [c] * 16
        for j in range(9):
            ring[(start + j) % 16] = c - 41
        assert oracle.fast9(_ring_img(c, ring), 20).tolist() == [[3, 3, 40]]
    # threshold edge: diff == t is NOT a corner, diff == t+1 is (score t)
    ring = [c + 20] * 9 + [c] * 7
    assert len(oracle.fast9(_ring_img(c, ring), 20)) == 0
    ring = [c + 21] * 9 + [c] * 7
    assert oracle.fast9(_ring_img(c, ring), 20).tolist() == [[3, 3, 20]]
    # score = min over the best arc - 1, arcs may be longer than 9
    ring = [c + 50, c + 60, c + 33, c + 70, c + 80, c + 90, c + 35, c + 40, c + 45, c + 47, c, c, c, c, c, c]
    assert oracle.fast9(_ring_img(c, ring), 20).tolist() == [[3, 3, 32]]


def test_fast_score_is_max_threshold(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 40), dtype=np.uint8)
    img[10:30, 10:30] = (img[10:30, 10:30] // 4) + 150      # a contrasty block => real corners
    k = oracle.fast9(img, 7, nms=False)
    assert len(k) > 20
    got = {(int(x), int(y)): int(s) for x, y, s in k}
    for y in range(3, 37):
        for x in range(3, 37):
            bf = oracle.fast_score_bruteforce(img, x, y)
            if bf >= 7:
                assert got.get((x, y)) == bf, (x, y)
            else:
                assert (x, y) not in got
    # nms keeps strict local maxima of the score map (zero outside the 3-px inset), row-major order
    smap = np.zeros((40, 40), np.int32)
    for (x, y), s in got.items():
        smap[y, x] = s
    want = []
    for y in range(3, 37):
        for x in range(3, 37):
            s = smap[y, x]
            nb = smap[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (x, y) in got and (s > nb).all():
                want.append([x, y, s])
    assert oracle.fast9(img, 7, nms=True).tolist() == want


def test_fast_atan2(oracle):
    for deg in np.linspace(0, 359.9, 721):
        y, x = math.sin(math.radians(deg)) * 1000, math.cos(math.radians(deg)) * 1000
        a = oracle.fast_atan2(y, x)
        d = abs(a - deg)
        assert min(d, 360 - d) < 0.02, (deg, a)   # polynomial's documented accuracy ~0.01 deg
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 5.0) == 0.0
    assert oracle.fast_atan2(5.0, 0.0) == 90.0
    assert oracle.fast_atan2(0.0, -5.0) == 180.0
    assert oracle.fast_atan2(-5.0, 0.0) == 270.0
    # restated in numpy float32 without FMA
    p = [np.float32(c) * np.float32(180 / math.pi) for c in
         (0.9997878412794807, -0.3258083974640975, 0.1555786518463281, -0.04432655554792128)]
    rng = np.random.default_rng(5)
    for _ in range(2000):
        y, x = (np.float32(v) for v in rng.integers(-200000, 200000, 2))
        ax, ay = abs(x), abs(y)
        eps = np.float32(2.220446049250313e-16)
        if ax >= ay:
            c = ay / (ax + eps)
            c2 = c * c
            a = (((p[3] * c2 + p[2]) * c2 + p[1]) * c2 + p[0]) * c
        else:
            c = ax / (ay + eps)
            c2 = c * c
            a = np.float32(90) - (((p[3] * c2 + p[2]) * c2 + p[1]) * c2 + p[0]) * c
        if x < 0:
            a = np.float32(180) - a
        if y < 0:
            a = np.float32(360) - a
        assert oracle.fast_atan2(float(y), float(x)) == float(a)


def test_hamming(oracle):
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, o) == 256
    for bit in (0, 7, 8, 100, 255):
        b = z.copy()
        b[bit // 8] = 1 << (bit % 8)
        assert oracle.hamming(z, b) == 1 and oracle.hamming(o, b) == 255
    rng = np.random.default_rng(7)
    for _ in range(200):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
        assert oracle.hamming(a[::-1].copy(), b[::-1].copy()) == oracle.hamming(a, b)   # byte order irrelevant


def test_octtree_small(oracle):
    # 4 well separated points in a 100x100 box, N large => all kept, one per node; list order is
    # push_front order: children n1..n4 pushed front => reversed quadrant order.
    pts = [(10, 10, 5), (80, 10, 6), (10, 80, 7), (80, 80, 8)]
    k = np.zeros(4, KP_DTYPE)
    for i, (x, y, r) in enumerate(pts):
        k[i] = (x, y, 7, -1, r, 0, -1)
    out = oracle.distribute_octtree(k, 0, 100, 0, 100, 10)
    assert [(int(p['x']), int(p['y'])) for p in out] == [(80, 80), (10, 80), (80, 10), (10, 10)]
    # N=1: the root already satisfies size>=N after the first (mandatory) split pass... the root
    # holds 4 points, is split once (4 nodes >= 1) and the best of each node is returned.
    out = oracle.distribute_octtree(k, 0, 100, 0, 100, 1)
    assert len(out) == 4
    # two points in the same final cell: the higher response wins, first wins ties
    k2 = np.zeros(3, KP_DTYPE)
    k2[0] = (10, 10, 7, -1, 5, 0, -1)
    k2[1] = (11, 10, 7, -1, 9, 0, -1)
    k2[2] = (80, 80, 7, -1, 1, 0, -1)
    out = oracle.distribute_octtree(k2, 0, 100, 0, 100, 2)
    assert sorted((int(p['x']), int(p['response'])) for p in out) == [(11, 9), (80, 1)]


def test_extract_invariants(oracle):
    from os1_amd.synth import synth
    img = synth(1, 640, 480)
    ex = OracleExtractor(1000, 1.2, 8, 20, 7, oracle)
    kps, desc = ex.extract(img)
    t = ex.tables()
    assert 900 <= len(kps) <= 1000 + 2 * 8
    assert (np.diff(kps['octave']) >= 0).all()                       # level-major output order
    for l in range(8):
        m = kps['octave'] == l
        assert m.sum() <= t['nfeat'][l] + 2
        lw, lh = ex.level(l).shape[::-1]
        x = kps['x'][m] / t['sf'][l]
        y = kps['y'][m] / t['sf'][l]
        assert x.min() >= 19 - 1e-3 and x.max() <= lw - 20 + 1e-3
        assert y.min() >= 19 - 1e-3 and y.max() <= lh - 20 + 1e-3
        assert (kps['size'][m] == np.float32(int(31 * t['sf'][l]))).all()
    assert ((kps['angle'] >= 0) & (kps['angle'] < 360)).all()
    # determinism
    kps2, desc2 = ex.extract(img)
    assert kps.tobytes() == kps2.tobytes() and desc.tobytes() == desc2.tobytes()
    # empty image => no output
    k0, d0 = ex.extract(np.zeros((0, 0), np.uint8))
    assert len(k0) == 0


def test_distinctive_descriptor_definition(oracle):
    # independent numpy restatement of MapPoint.cc:258-286
    rng = np.random.default_rng(4)
    for N in (1, 2, 3, 6, 11, 40):
        d = rng.integers(0, 256, (N, 32), dtype=np.uint8)
        D = np.array([[int(np.unpackbits(d[i] ^ d[j]).sum()) for j in range(N)] for i in range(N)])
        med = [sorted(D[i])[int(0.5 * (N - 1))] for i in range(N)]
        assert oracle.distinctive_descriptor(d) == int(np.argmin(med))


def test_cvt_gray_known_answers(oracle):
    """cv::cvtColor RGB2GRAY 8u: primaries, white, and an independent numpy restatement of both fixed-point variants."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [1, 1, 1], [128, 64, 32]]], np.uint8)
    # 0.299 / 0.587 / 0.114 of 255, rounded by the fixed-point formula
    assert oracle.cvt_gray(px, True, 0)[0].tolist() == [76, 150, 29, 255, 0, 1, 79]
    assert oracle.cvt_gray(px, False, 0)[0].tolist() == [29, 150, 76, 255, 0, 1, 62]
    assert oracle.cvt_gray(px, True, 1)[0].tolist() == [76, 150, 29, 255, 0, 1, 79]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    for variant, (cr, cg, cb, sh) in enumerate([(9798, 19235, 3735, 15), (4899, 9617, 1868, 14)]):
        assert cr + cg + cb == 1 << sh
        i = img.astype(np.int64)
        want = ((i[..., 0] * cr + i[..., 1] * cg + i[..., 2] * cb + (1 << (sh - 1))) >> sh).astype(np.uint8)
        assert (oracle.cvt_gray(img, True, variant) == want).all()
        assert (oracle.cvt_gray(img[..., :3], True, variant) == want).all()
        assert (oracle.cvt_gray(img[..., [2, 1, 0, 3]], False, variant) == want).all()
    # the two variants do differ somewhere (so the choice is observable)
    assert (oracle.cvt_gray(img, True, 0) != oracle.cvt_gray(img, True, 1)).any()
