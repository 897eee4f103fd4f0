"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle -- bit-exact."""
import numpy as np
import pytest

from oracle.pyoracle import OracleExtractor
from os1_amd.synth import shifted, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def api():
    from os1_amd import api as a
    assert a.device_count() >= 1, 'no GPU visible: the product has no CPU fallback'
    return a


def _cmp_extract(got, want):
    (gk, gd), (wk, wd) = got, want
    assert len(gk) == len(wk), (len(gk), len(wk))
    for f in gk.dtype.names:
        assert (gk[f] == wk[f]).all(), f
    assert gk.tobytes() == wk.tobytes()
    assert gd.tobytes() == wd.tobytes()


@pytest.mark.parametrize('cfg', [(1, 640, 480, 1000), (11, 752, 480, 500), (2, 1920, 1080, 2000)])
def test_stages_and_extract_bit_exact(api, oracle, cfg):
    seed, W, H, N = cfg
    img = synth(seed, W, H)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
    got = ex(img)
    want = ox.extract(img)
    t, ot = ex.tables(), ox.tables()
    for k in ('sf', 'isf', 's2', 'is2', 'nfeat'):
        assert (t[k] == ot[k]).all()
    total = 0
    for l in range(8):
        assert (ex.level(l) == ox.level(l)).all(), 'pyramid level %d' % l
        c, oc = ex.candidates(l), ox.candidates(l)
        assert len(c) == len(oc), 'level %d candidate count' % l
        assert (c[:, 0] == oc['x']).all() and (c[:, 1] == oc['y']).all() and (c[:, 2] == oc['response']).all()
        total += len(c)
    assert total >= 5 * N             # the quadtree actually saturates
    _cmp_extract(got, want)
    assert len(got[0]) >= N * 0.95


def test_batch_device_input_and_strides(api, oracle):
    imgs = [synth(20 + i, 800, 600) for i in range(3)]
    ex = api.Extractor(800, 1.2, 8, 20, 7)
    ox = OracleExtractor(800, 1.2, 8, 20, 7, oracle)
    want = [ox.extract(im) for im in imgs]
    for g, w in zip(ex.extract_batch(imgs), want):
        _cmp_extract(g, w)
    # frames already resident in HBM, with a row stride larger than the width
    dev = api.DeviceFrames(imgs, 0, stride=832)
    kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, 600, 800, 832, True)
    for i in range(3):
        _cmp_extract((kps[i, :n[i]], desc[i, :n[i]]), want[i])
    # frames in page-locked host memory (orbfe_host_alloc)
    pin = api.PinnedFrames(imgs)
    kps, desc, n = ex.extract_batch_ptrs(pin.ptrs, 600, 800, 800, False)
    for i in range(3):
        _cmp_extract((kps[i, :n[i]], desc[i, :n[i]]), want[i])
    pin.free()
    # host image with a padded stride (a ROI view)
    big = np.zeros((600, 900), np.uint8)
    big[:, 50:850] = imgs[0]
    _cmp_extract(ex(big[:, 50:850]), want[0])
    # changing the image size on the same handle
    img2 = synth(5, 512, 384)
    _cmp_extract(ex(img2), ox.extract(img2))


def test_colour_input_parity(api, oracle):
    """RGB / BGR / RGBA / BGRA frames (Tracking.cc:96-109): GPU colour->gray in front of the pyramid == oracle cvtColor
    restatement followed by the gray extractor; host and device input, odd widths (row tails, unaligned rows)."""
    rng = np.random.default_rng(3)
    for W, H in [(640, 480), (611, 403)]:
        g = synth(40 + W, W, H).astype(np.int32)
        col = np.stack([np.clip(g + rng.integers(-30, 31, g.shape), 0, 255),
                        np.clip(g + rng.integers(-10, 11, g.shape), 0, 255),
                        np.clip(g + rng.integers(-40, 41, g.shape), 0, 255),
                        rng.integers(0, 256, g.shape)], axis=-1).astype(np.uint8)
        ex = api.Extractor(700, 1.2, 8, 20, 7)
        ox = OracleExtractor(700, 1.2, 8, 20, 7, oracle)
        for fmt, ch, rgb in [('rgb', 3, True), ('bgr', 3, False), ('rgba', 4, True), ('bgra', 4, False)]:
            for variant in (0, 1):
                img = np.ascontiguousarray(col[..., :ch])
                gray = oracle.cvt_gray(img, rgb, variant)
                ex.set_input_format(fmt, variant)
                want = ox.extract(gray)
                _cmp_extract(ex.extract_color(img), want)
                assert (ex.level(0) == gray).all()
                if variant == 0:
                    dev = api.DeviceFrames([img.reshape(H, W * ch)], 0)
                    kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, H, W, W * ch, True)
                    _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want)
        ex.set_input_format('gray')
        _cmp_extract(ex(gray), ox.extract(gray))


def test_legacy_gaussian_variant_parity(api, oracle):
    """orbfe_extractor_set_blur_variant(ORBFE_GAUSS_ROUNDED): GaussianBlur of OpenCV 4.0.0 - 4.1.0 (taps [18,34,49,55,49,34,18], sum 257, one
    saturating cast; ORBextractor.cc:949-950 decides every descriptor bit).  Both variants against the oracle's, on every route a frame
    can take -- one-frame call (four waves per keypoint, results written to host memory), batch (one wave per keypoint), stream runner --
    and on an over-exposed frame where (sum + 32768) >> 16 reaches 256 / 257 and must clip to 255.  Keypoints and angles do not depend
    on the blur; the descriptors of the two variants differ."""
    imgs = [synth(61, 800, 600), synth(62, 800, 600)]
    hot = np.clip(synth(63, 800, 600).astype(np.int32) * 2 + 30, 0, 255).astype(np.uint8)       # large areas at 255, textured rims
    assert (hot == 255).mean() > 0.15
    imgs.append(hot)
    ex = api.Extractor(1200, 1.2, 8, 20, 7)
    ox = OracleExtractor(1200, 1.2, 8, 20, 7, oracle)
    per_variant = {}
    for variant in (1, 0, 1):
        ex.set_blur_variant(variant)
        ox.set_gauss_variant(variant)
        want = [ox.extract(im) for im in imgs]
        if variant == 1:      # the saturation is reached inside a descriptor's reach on the over-exposed frame
            lv0 = ox.level(0)
            k7 = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
            p = np.pad(lv0.astype(np.int64), 3, mode='reflect')
            v = sum(k7[t] * sum(k7[u] * p[:, u:u + lv0.shape[1]] for u in range(7))[t:t + lv0.shape[0], :] for t in range(7))
            assert ((v + 32768) >> 16).max() >= 256
            assert (ox.level(0, blurred=True) == np.minimum((v + 32768) >> 16, 255)).all()
        for im, w in zip(imgs, want):
            _cmp_extract(ex(im), w)                                   # one-frame route
        for g, w in zip(ex.extract_batch(imgs + imgs[:1]), want + want[:1]):
            _cmp_extract(g, w)                                        # batch route
        per_variant[variant] = want
    for (k0, d0), (k1, d1) in zip(per_variant[0], per_variant[1]):
        assert k0.tobytes() == k1.tobytes() and (d0 != d1).any()
    # stream runner: every extractor handle of the runner switches
    dev = api.DeviceFrames(imgs + imgs[:1], 0)
    st = api.Stream(1200, 1.2, 8, 20, 7, 0, 2, 2)
    st.set_blur_variant(1)
    st.set_matching((0.0, 800.0, 0.0, 600.0), 0, 0.9, True)
    for b in range(2):
        st.push_ptrs(dev.ptrs[2 * b:2 * b + 2], 600, 800, dev.stride, True)
    for b in range(2):
        kps, desc, n, _, _ = st.pop(copy=True)
        for i in range(2):
            wk, wd = per_variant[1][(2 * b + i) % 3]
            assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes()
    st.close()
    with pytest.raises(Exception):
        ex.set_blur_variant(2)
    # the HOST-quadtree route (strips with more than 4 roots per level, more than 2 044 features on a level) describes through
    # launch_describe: the variant must reach it too (tools/sweep_debug.py found that it did not, round 5)
    for (W, H, N, nl) in [(1266, 290, 56, 2), (604, 514, 2759, 1)]:
        img = synth(64, W, H)
        for variant in (1, 0):
            ex = api.Extractor(N, 1.25, nl, 20, 7)
            ox = OracleExtractor(N, 1.25, nl, 20, 7, oracle)
            ex.set_blur_variant(variant)
            ox.set_gauss_variant(variant)
            want = ox.extract(img)
            _cmp_extract(ex(img), want)
            for g in ex.extract_batch([img, img, img]):
                _cmp_extract(g, want)


def test_other_parameters(api, oracle):
    img = synth(7, 960, 540)
    for (N, sf, nl, ini, mn) in [(1500, 1.2, 8, 20, 7), (300, 1.5, 4, 30, 10), (1000, 1.1, 6, 12, 5), (50, 1.2, 3, 40, 40)]:
        _cmp_extract(api.Extractor(N, sf, nl, ini, mn)(img), OracleExtractor(N, sf, nl, ini, mn, oracle).extract(img))


def test_tiny_quotas_and_wide_strips(api, oracle):
    """DistributeOctTree divides every root once before it looks at N (ORBextractor.cc:620-700): with a quota of 1-3
    features a level still returns up to 4 * roots keypoints -- more than nfeatures in total; strips wider than 4.5 : 1
    (more than 4 roots) take the host-quadtree path, also with a tiny quota."""
    for (W, H, N, nl) in [(1920, 1080, 20, 8), (640, 480, 9, 8), (1600, 500, 16, 8), (1167, 252, 72, 5), (900, 120, 500, 2)]:
        img = synth(5, W, H)
        ex = api.Extractor(N, 1.2, nl, 20, 7)
        want = OracleExtractor(N, 1.2, nl, 20, 7, oracle).extract(img)
        got = ex(img)
        _cmp_extract(got, want)
        assert len(want[0]) <= ex.L.orbfe_extractor_max_keypoints_for_size(ex.h, H, W)
    assert len(want[0]) > 0


@pytest.mark.parametrize('cfg', [(225, 300, 300, 8), (110, 150, 120, 4), (229, 320, 250, 8)])
def test_batch_with_a_level_one_tile_wide(api, oracle, cfg):
    """Portrait images whose top pyramid level is at most 64 px wide but taller than 64 px: ONE tile column, several tile
    rows (tilesX == 1) -- in batches larger than ORBFE_CONE_MAX_FRAMES, i.e. through the per-level resize launches.  The
    XCD-consecutive tile mapping of k_resize_fixed divides by tilesX with a multiply-high that is exact only for
    tilesX >= 2; such levels must take the generic kernel.  Every level pixel-exact for every frame of the batch."""
    W, H, N, nl = cfg
    imgs = [synth(300 + i, W, H) for i in range(4)]
    ex = api.Extractor(N, 1.2, nl, 20, 7)
    ox = OracleExtractor(N, 1.2, nl, 20, 7, oracle)
    got = ex.extract_batch(imgs)
    lh, lw = ex.level(nl - 1).shape
    assert lw <= 64 < lh, (lw, lh)
    for i, im in enumerate(imgs):
        want = ox.extract(im)
        for l in range(nl):
            assert (ex.level(l, frame=i) == ox.level(l)).all(), 'frame %d pyramid level %d' % (i, l)
        _cmp_extract(got[i], want)


def test_edge_cases(api, oracle):
    ex = api.Extractor(500, 1.2, 8, 20, 7)
    k, d = ex(np.zeros((0, 0), np.uint8))                       # empty image: silent, no output
    assert len(k) == 0 and d.shape == (0, 32)
    flat = np.full((480, 640), 77, np.uint8)                     # no corners anywhere
    k, d = ex(flat)
    assert len(k) == 0
    ko, _ = OracleExtractor(500, 1.2, 8, 20, 7, oracle).extract(flat)
    assert len(ko) == 0
    with pytest.raises(api.OrbfeError) as e:                     # a level without a FAST cell
        ex(np.zeros((120, 160), np.uint8))
    assert e.value.code == -4
    # low-contrast frame: every cell falls back to minThFAST
    rng = np.random.default_rng(3)
    low = (120 + rng.integers(0, 12, (480, 640))).astype(np.uint8)
    _cmp_extract(ex(low), OracleExtractor(500, 1.2, 8, 20, 7, oracle).extract(low))
    # hard-edged checkerboard: dense, highly tied scores
    yy, xx = np.mgrid[0:480, 0:640]
    chk = (((yy // 9) + (xx // 7)) % 2 * 200 + 20).astype(np.uint8)
    _cmp_extract(ex(chk), OracleExtractor(500, 1.2, 8, 20, 7, oracle).extract(chk))
    # saturated noise: maximum candidate density
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    _cmp_extract(ex(noise), OracleExtractor(500, 1.2, 8, 20, 7, oracle).extract(noise))


def test_maximum_frame_size(api, oracle):
    """The largest frame the coordinate packing admits (4095 x 4095) is extracted bit-exactly -- keypoints up to
    x, y = 4075 on level 0, 18 000 FAST cells on it -- and one pixel more in either direction is refused, not truncated."""
    W = H = 4095
    img = synth(17, W, H)
    ex = api.Extractor(6000, 1.2, 8, 20, 7)
    k, d = ex(img)
    _cmp_extract((k, d), OracleExtractor(6000, 1.2, 8, 20, 7, oracle).extract(img))
    assert len(k) >= 5900 and k['x'].max() > 4000 and k['y'].max() > 4000
    for shape in ((100, 4096), (4096, 100)):
        with pytest.raises(api.OrbfeError) as e:
            ex(np.zeros(shape, np.uint8))
        assert e.value.code == -1 and '4095' in str(e.value)
    _cmp_extract(ex(img[:600, :800].copy()), OracleExtractor(6000, 1.2, 8, 20, 7, oracle).extract(img[:600, :800].copy()))


def test_device_sincos_matches_host_libm(api, oracle):
    ex = api.Extractor(100, 1.2, 8, 20, 7)
    rng = np.random.default_rng(0)
    ang = np.concatenate([rng.uniform(0, 360, 2_000_000), np.arange(0, 360, 0.25), [0.0, 1e-4, 359.99997]]).astype(np.float32)
    c, s = ex.sincos(ang)
    rad = (ang * np.float32(np.pi / 180.0)).astype(np.float32)
    import ctypes
    libm = ctypes.CDLL('libm.so.6')
    libm.cosf.restype = libm.sinf.restype = ctypes.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    idx = rng.choice(len(ang), 20000, replace=False)
    for i in idx:
        assert c[i] == np.float32(libm.cosf(float(rad[i]))) and s[i] == np.float32(libm.sinf(float(rad[i])))
    # and the whole array against the oracle's host evaluation of the reference expression on a subset
    for i in idx[:2000]:
        a, b = oracle.sincos(float(ang[i]))
        assert c[i] == np.float32(a) and s[i] == np.float32(b)


def _frames(api, oracle, W=1920, H=1080, N=2000, seed=3):
    A = synth(seed, W, H)
    B = shifted(A, -24, 3, seed)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    return ex, ex(A), ex(B), (0.0, float(W), 0.0, float(H))


def test_search_for_initialization_parity(api, oracle):
    ex, (k1, d1), (k2, d2), bounds = _frames(api, oracle)
    m = api.Matcher()
    prev = np.stack([k1['x'], k1['y']], 1)
    for window, ratio, ori in [(100, 0.9, True), (100, 0.9, False), (10, 0.6, True), (300, 0.95, True)]:
        n, m12, p = m.search_for_initialization(k1, d1, k2, d2, bounds, prev, window, ratio, ori)
        on, om12, op = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev, window, ratio, ori)
        assert n == on and (m12 == om12).all() and p.tobytes() == op.tobytes()
    n, m12, p = m.search_for_initialization(k1, d1, k2, d2, bounds, prev, 100, 0.9, True)
    assert n > 100                                     # the shifted pair really matches
    good = m12 >= 0
    dx = k2['x'][m12[good]] - k1['x'][good]
    assert abs(np.median(dx) + 24) <= 1.0
    # second round with the updated vbPrevMatched (Tracking.cc:384 keeps calling with it)
    n2, m12b, p2 = m.search_for_initialization(k1, d1, k2, d2, bounds, p, 100, 0.9, True)
    on2, om12b, op2 = oracle.search_for_initialization(k1, d1, k2, d2, bounds, p, 100, 0.9, True)
    assert n2 == on2 and (m12b == om12b).all() and p2.tobytes() == op2.tobytes()
    # empty / degenerate inputs
    n0, m0, _ = m.search_for_initialization(k1[:0], d1[:0], k2, d2, bounds, prev[:0])
    assert n0 == 0 and len(m0) == 0
    n0, m0, _ = m.search_for_initialization(k1, d1, k2[:0], d2[:0], bounds, prev)
    assert n0 == 0 and (m0 == -1).all()


def test_get_features_in_area_parity(api, oracle):
    ex, (k1, d1), _, bounds = _frames(api, oracle, 1280, 720, 1500, 8)
    m = api.Matcher()
    rng = np.random.default_rng(1)
    for _ in range(60):
        x, y = float(rng.uniform(-50, 1330)), float(rng.uniform(-50, 770))
        r = float(rng.choice([1.0, 2.5, 4.0, 15.0, 100.0, 5000.0]))
        lo, hi = [(-1, -1), (0, 0), (0, 3), (2, 3), (3, -1), (7, 7)][rng.integers(6)]
        got = m.get_features_in_area(k1, bounds, x, y, r, lo, hi)
        want = oracle.get_features_in_area(k1, bounds, x, y, r, lo, hi)
        assert got.tolist() == want.tolist()
    # undistorted (fisheye) bounds that do not start at zero
    b2 = (-211.5, 1500.25, -80.0, 799.0)
    for _ in range(20):
        x, y, r = float(rng.uniform(0, 1280)), float(rng.uniform(0, 720)), float(rng.uniform(1, 60))
        assert m.get_features_in_area(k1, b2, x, y, r, -1, -1).tolist() == \
            oracle.get_features_in_area(k1, b2, x, y, r, -1, -1).tolist()


def _mappoints(k, d, n_mp, rng):
    src = rng.integers(0, len(k), n_mp)
    desc = d[src].copy()
    for i in range(n_mp):                                   # 0-40 random bit flips
        for b in rng.integers(0, 256, rng.integers(0, 41)):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    xy = np.stack([k['x'][src], k['y'][src]], 1) + rng.uniform(-3, 3, (n_mp, 2)).astype(np.float32)
    level = np.minimum(k['octave'][src] + rng.integers(0, 2, n_mp), 7).astype(np.int32)
    viewcos = rng.uniform(0.9, 1.0, n_mp).astype(np.float32)
    flags = np.full(n_mp, 1 | 8, np.uint8)
    flags[rng.random(n_mp) < 0.02] |= 2                      # 2 % bad
    flags[rng.random(n_mp) < 0.05] &= ~np.uint8(1)           # some not in view
    flags[rng.random(n_mp) < 0.05] |= 4                      # plCandidato
    flags[rng.random(n_mp) < 0.1] &= ~np.uint8(8)            # no observations yet
    return xy.astype(np.float32), level, viewcos, flags, desc


def test_search_by_projection_parity(api, oracle):
    ex, (k, d), _, bounds = _frames(api, oracle, 1920, 1080, 2000, 12)
    sf = ex.tables()['sf']
    m = api.Matcher()
    rng = np.random.default_rng(2)
    xy, level, viewcos, flags, mdesc = _mappoints(k, d, 5000, rng)
    occ = (rng.random(len(k)) < 0.1).astype(np.uint8)
    for th, ratio in [(1.0, 0.8), (5.0, 0.8), (3.0, 0.6)]:
        n, a = m.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, th, ratio)
        on, oa = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, th, ratio)
        assert n == on and (a == oa).all()
        assert n > 500
    n, a = m.search_by_projection(k, d, bounds, sf, occ, xy[:0], level[:0], viewcos[:0], flags[:0], mdesc[:0], 1.0, 0.8)
    assert n == 0 and (a == -1).all()


def test_search_by_projection_uv_parity(api, oracle):
    ex, (k1, d1), (k2, d2), bounds = _frames(api, oracle, 1280, 720, 1500, 21)
    sf = ex.tables()['sf']
    m = api.Matcher()
    rng = np.random.default_rng(4)
    # "last frame" = frame 1: every keypoint with a MapPoint projects to its position shifted by (-24,+3)
    uv = np.stack([k1['x'] - 24 + rng.uniform(-2, 2, len(k1)), k1['y'] + 3 + rng.uniform(-2, 2, len(k1))], 1).astype(np.float32)
    valid = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    sflags = np.where(rng.random(len(k1)) < 0.9, 8, 0).astype(np.uint8)
    occ = (rng.random(len(k2)) < 0.05).astype(np.uint8)
    for th, maxd, skip_any, ori in [(15.0, 100, 0, True), (30.0, 100, 0, False), (10.0, 64, 1, True), (3.0, 100, 1, True)]:
        n, a = m.search_by_projection_uv(k2, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid, d1,
                                         th, maxd, skip_any, ori)
        on, oa = oracle.search_by_projection_uv(k2, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid,
                                                d1, th, maxd, skip_any, ori)
        assert n == on and (a == oa).all()
    assert n > 50


def test_host_and_gpu_quadtree_paths_agree(api, oracle, monkeypatch):
    """The extractor has two thinning back ends (GPU-resident k_quadtree, default; host quadtree.h);
    both must reproduce the oracle on awkward inputs."""
    rng = np.random.default_rng(17)
    imgs = [synth(40, 1280, 720), rng.integers(0, 256, (600, 800), dtype=np.uint8),
            (((np.mgrid[0:480, 0:640][0] // 9) + (np.mgrid[0:480, 0:640][1] // 7)) % 2 * 200 + 20).astype(np.uint8)]
    for N in (2000, 137, 6000):          # 6000 -> 1300 features on level 0: the 2048-node quadtree variant
        ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
        monkeypatch.setenv('ORBFE_HOST_QUADTREE', '1')
        ex_host = api.Extractor(N, 1.2, 8, 20, 7)
        monkeypatch.setenv('ORBFE_HOST_QUADTREE', '0')
        ex_gpu = api.Extractor(N, 1.2, 8, 20, 7)
        for im in imgs:
            want = ox.extract(im)
            _cmp_extract(ex_host(im), want)
            _cmp_extract(ex_gpu(im), want)
    monkeypatch.delenv('ORBFE_HOST_QUADTREE')


def test_search_for_initialization_batch_parity(api, oracle):
    ex = api.Extractor(1000, 1.2, 8, 20, 7)
    frames = [synth(50, 960, 540)]
    for i in range(1, 5):
        frames.append(shifted(frames[-1], 3, -2, 50 + i))
    feats = [ex(f) for f in frames]
    bounds = (0.0, 960.0, 0.0, 540.0)
    pairs = []
    for i in range(1, 5):
        (k1, d1), (k2, d2) = feats[i - 1], feats[i]
        pairs.append((k1, d1, k2, d2, np.stack([k1['x'], k1['y']], 1)))
    pairs.append((feats[0][0][:0], feats[0][1][:0], feats[1][0], feats[1][1], np.zeros((0, 2), np.float32)))   # empty F1
    pairs.append((feats[0][0], feats[0][1], feats[1][0][:0], feats[1][1][:0], np.stack([feats[0][0]['x'], feats[0][0]['y']], 1)))  # empty F2
    got = api.Matcher().search_for_initialization_batch(pairs, bounds, 100, 0.9, True)
    assert len(got) == len(pairs)
    for (k1, d1, k2, d2, prev), (n, m12, p) in zip(pairs, got):
        on, om12, op = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev, 100, 0.9, True)
        assert n == on and (m12 == om12).all() and p.tobytes() == op.tobytes()
    assert got[0][0] > 50


@pytest.mark.parametrize('mode', ['gpu', 'gpu-noori', 'gpu-tight', 'host'])
def test_stream_runner_parity(api, oracle, mode, monkeypatch):
    """orbfe_stream_*: pushed batches come back in order with the same keypoints / descriptors / matches as
    the oracle computes frame by frame (frame i matched against frame i-1 of the stream).  Modes: GPU-resident
    SearchForInitialization (k_sfi_*, default) with several parameter sets, and the host-side match workers."""
    W, H, N, B = 800, 600, 800, 3
    window, ratio, ori = {'gpu': (100, 0.9, True), 'gpu-noori': (100, 0.9, False), 'gpu-tight': (25, 0.7, True),
                          'host': (100, 0.9, True)}[mode]
    monkeypatch.setenv('ORBFE_STREAM_HOST_MATCH', '1' if mode == 'host' else '0')
    base = synth(60, W, H)
    frames = [base] + [shifted(base, 3 * i, -2 * i, 600 + i) for i in range(1, 3 * B)]
    frames[4] = np.full((H, W), 90, np.uint8)          # a frame without any keypoint in the middle of the stream
    dev = api.DeviceFrames(frames, 0)
    st = api.Stream(N, 1.2, 8, 20, 7, 0, B, 2)
    bounds = (0.0, float(W), 0.0, float(H))
    st.set_matching(bounds, window, ratio, ori)
    ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
    want = [ox.extract(f) for f in frames]
    for b in range(3):
        st.push_ptrs(dev.ptrs[b * B:(b + 1) * B], H, W, dev.stride, True)
    total = 0
    for b in range(3):
        kps, desc, n, m12, nm = st.pop(copy=True)
        for i in range(B):
            g = b * B + i
            wk, wd = want[g]
            assert n[i] == len(wk)
            assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes()
            if g == 0:
                assert nm[i] == 0
                continue
            pk, pd = want[g - 1]
            on, om12, _ = oracle.search_for_initialization(pk, pd, wk, wd, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2),
                                                           window, ratio, ori)
            assert nm[i] == on and (m12[i, :len(pk)] == om12).all()
            total += on
    assert total > 100
    st.close()


@pytest.mark.parametrize('cfg', [(640, 480, 25, 8, 2), (1000, 300, 60, 4, 3), (333, 251, 300, 3, 2), (1500, 260, 200, 3, 2),
                                 (1280, 720, 12000, 4, 2)])
def test_stream_runner_small_quotas_and_odd_sizes(api, oracle, cfg):
    """The streamed path (GPU quadtree, slot-mode descriptors, GPU SearchForInitialization chain) with tiny per-level
    quotas (a level returns 4 x roots keypoints, more than its quota), 3-4 quadtree roots and odd image sizes -- and the two
    geometries OUTSIDE the GPU quadtree's limits, which the asynchronous entry points route through the host quadtree and
    a matcher handle by themselves: a strip with 6 root nodes per level (1500 x 260) and 12 000 features (more than 2 044
    on a level)."""
    W, H, N, nl, B = cfg
    base = synth(70 + W, W, H)
    frames = [base] + [shifted(base, 2 * i, i, 700 + i) for i in range(1, 3 * B)]
    dev = api.DeviceFrames(frames, 0)
    st = api.Stream(N, 1.2, nl, 20, 7, 0, B, 2)
    bounds = (0.0, float(W), 0.0, float(H))
    st.set_matching(bounds, 60, 0.9, True)
    ox = OracleExtractor(N, 1.2, nl, 20, 7, oracle)
    want = [ox.extract(f) for f in frames]
    for b in range(3):
        st.push_ptrs(dev.ptrs[b * B:(b + 1) * B], H, W, dev.stride, True)
    for b in range(3):
        kps, desc, n, m12, nm = st.pop(copy=True)
        for i in range(B):
            g = b * B + i
            wk, wd = want[g]
            assert n[i] == len(wk)
            assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes()
            if g == 0:
                continue
            pk, pd = want[g - 1]
            on, om12, _ = oracle.search_for_initialization(pk, pd, wk, wd, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2),
                                                           60, 0.9, True)
            assert nm[i] == on and (m12[i, :len(pk)] == om12).all()
    st.close()


def test_window_candidates_primitive(api, oracle):
    """orbfe_window_candidates: per query the GetFeaturesInArea list in reference order with Hamming distances
    (the GPU half of Fuse / SearchBySim3 / SearchByProjection(KeyFrame*, Scw, ...))."""
    ex, (k, d), _, bounds = _frames(api, oracle, 1280, 720, 1500, 33)
    m = api.Matcher()
    rng = np.random.default_rng(9)
    nq = 300
    qx = rng.uniform(-20, 1300, nq).astype(np.float32)
    qy = rng.uniform(-20, 740, nq).astype(np.float32)
    qr = rng.choice([-1.0, 3.0, 7.5, 25.0, 90.0, 400.0], nq).astype(np.float32)
    lv = rng.integers(0, 8, nq)
    qmin = np.where(rng.random(nq) < 0.2, -1, lv - 1).astype(np.int32)
    qmax = np.where(qmin < 0, -1, lv + rng.integers(0, 2, nq)).astype(np.int32)
    qdesc = d[rng.integers(0, len(k), nq)].copy()
    res = m.window_candidates(k, d, bounds, qx, qy, qr, qmin, qmax, qdesc)
    total = 0
    for q in range(nq):
        idx, dist = res[q]
        if qr[q] < 0:
            assert len(idx) == 0
            continue
        want = oracle.get_features_in_area(k, bounds, float(qx[q]), float(qy[q]), float(qr[q]), int(qmin[q]), int(qmax[q]))
        assert idx.tolist() == want.tolist()
        assert dist.tolist() == [oracle.hamming(qdesc[q], d[i]) for i in want]
        total += len(want)
    assert total > 2000


def test_distinctive_descriptors_parity(api, oracle):
    """MapPoint::ComputeDistinctiveDescriptors, batched: all-pairs Hamming + row medians on the GPU."""
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    lists = []
    for N in [1, 2, 3, 4, 5, 7, 8, 20, 33, 64, 65, 100, 257] + rng.integers(1, 60, 200).tolist():
        proto = base[rng.integers(0, 64)]
        d = np.repeat(proto[None], N, 0).copy()
        for i in range(N):                                    # noisy copies of one descriptor: realistic and tie-rich
            for b in rng.integers(0, 256, rng.integers(0, 60)):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        if N > 3 and rng.random() < 0.3:
            d[rng.integers(0, N)] = d[rng.integers(0, N)]      # exact duplicates
        lists.append(d)
    lists.append(np.zeros((0, 32), np.uint8))                  # MapPoint without usable observations
    got = api.Matcher().distinctive_descriptors(lists)
    for d, g in zip(lists, got):
        assert g == oracle.distinctive_descriptor(d), len(d)


def test_config5_4k_fisheye_search_by_projection(api, oracle):
    """BASELINE.json configs[4]: 3840x2160, nFeatures=4000, camera modo 1 (equidistant fisheye: keypoints are
    undistorted on the host, the image is not warped), SearchByProjection against 10 000 MapPoints, th 1 and 5."""
    W, H, N = 3840, 2160, 4000
    fx = fy = 2196.0
    cx, cy = 1839.0, 1155.0
    img = synth(5, W, H)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
    k, d = ex(img)
    _cmp_extract((k, d), ox.extract(img))
    assert len(k) >= 3900
    # Frame::UndistortKeyPoints (modo 1) and ComputeImageBounds: product helper == oracle
    kun = k.copy()
    xy = api.undistort_equidistant(np.stack([k['x'], k['y']], 1), fx, fy, cx, cy)
    assert xy.tobytes() == oracle.undistort_equidistant(np.stack([k['x'], k['y']], 1), fx, fy, cx, cy).tobytes()
    kun['x'], kun['y'] = xy[:, 0], xy[:, 1]
    corners = api.undistort_equidistant(np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32), fx, fy, cx, cy)
    bounds = (float(min(corners[0, 0], corners[2, 0])), float(max(corners[1, 0], corners[3, 0])),
              float(min(corners[0, 1], corners[1, 1])), float(max(corners[2, 1], corners[3, 1])))
    sf = ex.tables()['sf']
    rng = np.random.default_rng(55)
    mxy, level, viewcos, flags, mdesc = _mappoints(kun, d, 10000, rng)
    occ = np.zeros(len(k), np.uint8)
    m = api.Matcher()
    for th in (1.0, 5.0):
        n, a = m.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8)
        on, oa = oracle.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8)
        assert n == on and (a == oa).all()
        assert n > 1000


def test_search_projected_parity(api, oracle):
    """The projected best-match loop of SearchByProjection(KeyFrame*, Scw, ...), Fuse x2 and SearchBySim3:
    claim / skip semantics, Fuse's chi-square gate, TH_LOW / TH_HIGH."""
    W, H, N = 1280, 720, 1500
    img = synth(21, W, H)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    tab = ex.tables()
    m = api.Matcher()
    rng = np.random.default_rng(8)
    NS = 4000
    src = rng.integers(0, len(kps), NS)
    sdesc = desc[src].copy()
    for i in range(NS):
        for b in rng.integers(0, 256, rng.integers(0, 45)):
            sdesc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    uv = (np.stack([kps['x'][src], kps['y'][src]], 1) + rng.normal(0, 2.5, (NS, 2))).astype(np.float32)
    level = np.clip(kps['octave'][src] + rng.integers(-1, 2, NS), -1, 7).astype(np.int32)
    valid = (rng.random(NS) < 0.93).astype(np.uint8)
    bounds = (0.0, float(W), 0.0, float(H))
    total = 0
    for th, claim, skip, gate, maxd in [(4.0, True, True, False, 50), (3.0, False, False, True, 50), (2.5, False, False, False, 50),
                                        (7.5, False, False, False, 100), (10.0, True, False, True, 100)]:
        radius = (th * tab['sf'][np.clip(level, 0, 7)]).astype(np.float32)
        kp_skip = (rng.random(len(kps)) < 0.2).astype(np.uint8) if skip else None
        inv = tab['is2'] if gate else None
        got = m.search_projected(kps, desc, bounds, uv, radius, level, valid, sdesc, kp_skip, claim, inv, 5.99, maxd)
        want = oracle.search_projected(kps, desc, bounds, uv, radius, level, valid, sdesc, kp_skip, claim, inv, 5.99, maxd)
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes() and got[2].tobytes() == want[2].tobytes()
        if claim:
            acc = got[1][got[1] >= 0]
            assert len(np.unique(acc)) == len(acc)          # a claimed keypoint is taken once
        if kp_skip is not None:
            assert not kp_skip[got[1][got[1] >= 0]].any()
        total += got[0]
    assert total > 3000
    n, bi, bd = m.search_projected(kps, desc, bounds, uv[:0], uv[:0, 0], level[:0], valid[:0], sdesc[:0])
    assert n == 0 and len(bi) == 0


def test_randomised_sizes_and_parameters(api, oracle):
    """Seeded sweep over image sizes (odd widths, narrow strips, level-0 strides that are not multiples of 4), pyramid
    shapes (wide strips with more than 4 quadtree roots take the host-quadtree path) and thresholds -- including
    iniThFAST < minThFAST, where the reference's second cv::FAST call sees FEWER
    corners than the first (ORBextractor.cc:846-856) -- and low-texture images where most cells need the second pass."""
    rng = np.random.default_rng(2024)
    for it in range(14):
        W = int(rng.integers(97, 900))
        H = min(int(rng.integers(97, 700)), int(1.7 * W))     # taller than 2:1 has no quadtree root (reference: division by zero)
        N = int(rng.integers(30, 1500))
        sf = float(rng.choice([1.1, 1.2, 1.3, 1.5]))
        nl = int(rng.integers(1, 9))
        ini, mn = [(20, 7), (7, 20), (12, 12), (40, 5), (9, 3)][it % 5]
        while min(W, H) / sf ** (nl - 1) < 66:           # every level must hold at least one 30-px FAST cell inside the border
            nl -= 1
        img = synth(500 + it, W, H)
        if it % 3 == 1:                                    # low texture: flatten contrast so iniTh finds little
            img = (128 + (img.astype(np.int32) - 128) // 6).astype(np.uint8)
        if it % 4 == 2:                                    # a constant block: cells with no corner at any threshold
            img[: H // 2, : W // 2] = 90
        ex = api.Extractor(N, sf, nl, ini, mn)
        ox = OracleExtractor(N, sf, nl, ini, mn, oracle)
        want = ox.extract(img)
        _cmp_extract(ex(img), want)
        # the same frame through the batch path with a padded device stride
        stride = W + int(rng.integers(0, 3))
        dev = api.DeviceFrames([img, img[::-1].copy()], 0, stride=stride)
        kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, H, W, stride, True)
        _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want)
        _cmp_extract((kps[1, :n[1]], desc[1, :n[1]]), ox.extract(img[::-1].copy()))


def test_fast_cell_pairs_variant_bit_exact(api, oracle, monkeypatch):
    """ORBFE_FAST_PAIRS=1: two horizontally adjacent FAST cells per wave (fewer instructions, kept as an option,
    DESIGN.md s5) -- same keypoints and descriptors, including the per-cell minThFAST fallback and the NMS cut at the
    boundary between the two cells of a wave."""
    monkeypatch.setenv('ORBFE_FAST_PAIRS', '1')
    for seed, W, H, N in ((2, 1920, 1080, 2000), (12, 752, 480, 900), (13, 801, 333, 600)):
        img = synth(seed, W, H)
        if seed == 13:
            img[:, W // 2:] = (img[:, W // 2:] // 8 + 100).astype(np.uint8)     # low contrast half: cells that need the minTh pass
        ex = api.Extractor(N, 1.2, 8, 20, 7)
        ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
        _cmp_extract(ex(img), ox.extract(img))
        for l in range(8):
            c, oc = ex.candidates(l), ox.candidates(l)
            assert len(c) == len(oc), (seed, l)
            assert (c[:, 0] == oc['x']).all() and (c[:, 1] == oc['y']).all() and (c[:, 2] == oc['response']).all(), (seed, l)


def test_latency_path_variants_agree(api, oracle, monkeypatch):
    """One- and two-frame calls take their own kernels (k_pyramid_cone, LDS-resident quadtree candidates, results written
    straight to the host arena, level-0 pointers in the kernel arguments, level-local candidate lists from k_compact_local).  Every switch of that path, alone and
    together, reproduces the oracle: the per-level pyramid kernels, candidates in HBM, the copied
    result arena, page-locked host frames fetched by the compute stream's own kernel (k_ingest: widths that are multiples of 16, of 4,
    and odd ones that go through the copy engine) or by a copy command (ORBFE_INGEST_KERNEL=0)."""
    cases = [(31, 1920, 1080, 2000, 1.2, 8), (32, 641, 479, 700, 1.2, 8), (33, 1280, 720, 1200, 1.3, 5), (34, 500, 400, 300, 1.5, 3),
             (35, 900, 500, 400, 2.0, 3)]      # scale 2.0: outside the cone kernel's range, per-level kernels by themselves
    settings = [{}, {'ORBFE_CONE_MAX_FRAMES': '0'}, {'ORBFE_QT_LDS_BYTES': '0'}, {'ORBFE_ZERO_COPY': '0'}, {'ORBFE_QT_JUMP': '0'},
                {'ORBFE_QT_LDS_BYTES': '20000'}, {'ORBFE_INGEST_KERNEL': '0'},
                {'ORBFE_CONE_MAX_FRAMES': '0', 'ORBFE_QT_LDS_BYTES': '0', 'ORBFE_ZERO_COPY': '0'}]
    for seed, W, H, N, sf, nl in cases:
        img = synth(seed, W, H)
        img2 = shifted(img, 5, 3, seed + 100)
        ox = OracleExtractor(N, sf, nl, 20, 7, oracle)
        want, want2 = ox.extract(img), ox.extract(img2)
        for env in settings:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ex = api.Extractor(N, sf, nl, 20, 7)
            _cmp_extract(ex(img), want)
            dev = api.DeviceFrames([img, img2], 0)
            kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, H, W, dev.stride, True)
            _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want)
            _cmp_extract((kps[1, :n[1]], desc[1, :n[1]]), want2)
            # host input, as Frame.cc:133 hands it over: pageable (ex(img) above) and page-locked, one and two frames
            pin = api.PinnedFrames([img, img2])
            kps, desc, n = ex.extract_batch_ptrs(pin.ptrs[1:], H, W, W, False)
            _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want2)
            kps, desc, n = ex.extract_batch_ptrs(pin.ptrs, H, W, W, False)
            _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want)
            _cmp_extract((kps[1, :n[1]], desc[1, :n[1]]), want2)
            pin.free()
            for k in env:
                monkeypatch.delenv(k)


def test_mixed_pinned_and_pageable_frames(api, oracle):
    """A two-frame host call (the stereo shape of Frame.cc:131-134) whose frames live in different kinds of memory: one page-locked,
    one pageable, in either order -- and a page-locked frame whose rows run past the end of the allocation's mapping would be the
    same case.  The ingest kernel reads a frame in place only when EVERY frame of the call is mapped for the GPU over its whole
    extent; otherwise the whole call goes through the copy route (round-5 advice: the check used to look at frame 0 only and the
    kernel then read unmapped pageable memory)."""
    W, H, N = 1280, 720, 1200
    a = synth(61, W, H)
    b = shifted(a, 4, 2, 62)
    ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
    wa, wb = ox.extract(a), ox.extract(b)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    pin = api.PinnedFrames([a, b])
    pa, pb = np.ascontiguousarray(a.copy()), np.ascontiguousarray(b.copy())     # pageable copies
    for ptrs, want in (([pin.ptrs[0], pb.ctypes.data], (wa, wb)), ([pa.ctypes.data, pin.ptrs[1]], (wa, wb)),
                       ([pb.ctypes.data, pin.ptrs[0]], (wb, wa)), ([pin.ptrs[0], pin.ptrs[1]], (wa, wb))):
        for rep in range(2):
            kps, desc, n = ex.extract_batch_ptrs(ptrs, H, W, W, False)
            _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want[0])
            _cmp_extract((kps[1, :n[1]], desc[1, :n[1]]), want[1])
    pin.free()


def test_submit_wait_collect_on_two_handles(api, oracle):
    """orbfe_extract_batch_submit / _wait / _collect: two handles pipelined the way a throughput caller does it -- the second batch is
    submitted between the first one's wait and its collect; a wait without a pending batch and a second wait are no-ops; the blocking
    host-quadtree route (a strip with more than four roots) goes through the same calls."""
    imgs = [synth(91 + i, 960, 540) for i in range(6)]
    ox = OracleExtractor(900, 1.2, 8, 20, 7, oracle)
    want = [ox.extract(im) for im in imgs]
    dev = api.DeviceFrames(imgs, 0)
    a, b = api.Extractor(900, 1.2, 8, 20, 7), api.Extractor(900, 1.2, 8, 20, 7)
    a.wait()                                             # nothing pending: returns at once
    a.submit_ptrs(dev.ptrs[0:3], 540, 960, dev.stride, True)
    a.wait()
    b.submit_ptrs(dev.ptrs[3:6], 540, 960, dev.stride, True)
    a.wait()
    ka, da, na = a.collect()
    b.wait()
    kb, db, nb = b.collect()
    for i in range(3):
        _cmp_extract((ka[i, :na[i]], da[i, :na[i]]), want[i])
        _cmp_extract((kb[i, :nb[i]], db[i, :nb[i]]), want[3 + i])
    strip = synth(97, 1500, 260)
    sx = api.Extractor(200, 1.2, 3, 20, 7)
    sdev = api.DeviceFrames([strip, strip, strip], 0)
    sx.submit_ptrs(sdev.ptrs, 260, 1500, sdev.stride, True)
    sx.wait()
    cap = sx.L.orbfe_extractor_max_keypoints_for_size(sx.h, 260, 1500)
    kk, dd = np.zeros((3, max(cap, sx.cap)), api.KP_DTYPE), np.zeros((3, max(cap, sx.cap), 32), np.uint8)
    sx.cap = max(cap, sx.cap)
    ks, ds, ns = sx.collect(kk, dd)
    ws = OracleExtractor(200, 1.2, 3, 20, 7, oracle).extract(strip)
    for i in range(3):
        _cmp_extract((ks[i, :ns[i]], ds[i, :ns[i]]), ws)


def test_registered_host_buffers_take_the_page_locked_route(api, oracle):
    """orbfe_host_register: a buffer the CALLER owns (a capture ring, a long-lived cv::Mat) page-locked and mapped in place; frames inside it
    are fetched by the compute stream like orbfe_host_alloc memory (one- and two-frame calls: k_ingest reads them through the mapping;
    batches: copy commands).  Frames at different offsets of one registered block, aligned and not; after unregistering the same memory is
    ordinary pageable memory again."""
    import time
    W, H, N = 1280, 720, 1200
    imgs = [synth(81, W, H), shifted(synth(81, W, H), 4, 2, 82), synth(83, W, H)]
    ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
    want = [ox.extract(im) for im in imgs]
    ring = np.zeros(3 * W * H + 64, np.uint8)                      # the caller's buffer: three frames back to back, then 64 spare bytes
    reg = api.RegisteredArray(ring)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    for off in (0, 16, 4, 1):                                      # 16-byte / dword aligned frames (k_ingest), an odd address (copy engine)
        view = ring[off:off + 3 * W * H].reshape(3, H, W)
        view[:] = np.stack(imgs)
        ptrs = [view[i].ctypes.data for i in range(3)]
        for i in range(3):
            kps, desc, n = ex.extract_batch_ptrs(ptrs[i:i + 1], H, W, W, False)
            _cmp_extract((kps[0, :n[0]], desc[0, :n[0]]), want[i])
        kps, desc, n = ex.extract_batch_ptrs(ptrs[:2], H, W, W, False)
        for i in range(2):
            _cmp_extract((kps[i, :n[i]], desc[i, :n[i]]), want[i])
        kps, desc, n = ex.extract_batch_ptrs(ptrs, H, W, W, False)      # three frames: the batch route
        for i in range(3):
            _cmp_extract((kps[i, :n[i]], desc[i, :n[i]]), want[i])
    # the registered route is the fast one (k_ingest instead of the runtime's staging copy)
    view = ring[:3 * W * H].reshape(3, H, W)
    view[:] = np.stack(imgs)
    kb, db = np.zeros((1, ex.cap), api.KP_DTYPE), np.zeros((1, ex.cap, 32), np.uint8)

    def med(ptr):
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            ex.extract_batch_ptrs([ptr], H, W, W, False, kb, db)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[8:]))
    t_reg = med(view[1].ctypes.data)
    reg.close()
    t_page = med(view[1].ctypes.data)
    _cmp_extract(ex(view[2]), want[2])                               # ... and still correct as pageable memory
    print('one 720p frame: registered %.3f ms, pageable %.3f ms' % (t_reg * 1e3, t_page * 1e3))
    assert t_reg < t_page


@pytest.mark.parametrize('cfg', [(640, 480, 25, 8, 2, [0, 0]), (1500, 260, 200, 3, 2, [0, 0, 0]), (1280, 720, 12000, 4, 2, [0, 0]), (800, 600, 800, 8, 3, None)])
def test_isolated_batches_and_multi_runner_odd_geometries(api, oracle, cfg):
    """orbfe_stream_multi_* (one stream dealt to several device runners, results in push order, the batch-boundary predecessor over the
    host) on geometries that take different routes inside a runner -- tiny quotas, a strip with 6 quadtree roots and 12 000 features
    (both OUTSIDE the GPU quadtree's limits: host quadtree + matcher handle, where the chain's isolated mode drops the host-side
    predecessor) -- against the oracle frame by frame; and (devices None) a single runner with isolated batches: frame 0 of EVERY batch
    reports no match, the other frames are matched as always."""
    W, H, N, nl, B, devices = cfg
    base = synth(70 + W, W, H)
    frames = [base] + [shifted(base, 2 * i, i, 700 + i) for i in range(1, 4 * B)]
    dev = api.DeviceFrames(frames, 0)
    bounds = (0.0, float(W), 0.0, float(H))
    ox = OracleExtractor(N, 1.2, nl, 20, 7, oracle)
    want = [ox.extract(f) for f in frames]
    if devices is None:
        st = api.Stream(N, 1.2, nl, 20, 7, 0, B, 2)
        st.set_isolated_batches(True)
    else:
        st = api.MultiStream(N, 1.2, nl, 20, 7, devices, B, 2)
    st.set_matching(bounds, 60, 0.9, True)
    for b in range(4):
        st.push_ptrs(dev.ptrs[b * B:(b + 1) * B], H, W, dev.stride, True)
    total = 0
    for b in range(4):
        kps, desc, n, m12, nm = st.pop(copy=True)
        for i in range(B):
            g = b * B + i
            wk, wd = want[g]
            assert n[i] == len(wk), (b, i)
            assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes(), (b, i)
            if g == 0 or (devices is None and i == 0):
                assert nm[i] == 0 and (m12[i] == -1).all(), (b, i)
                continue
            pk, pd = want[g - 1]
            on, om12, _ = oracle.search_for_initialization(pk, pd, wk, wd, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2), 60, 0.9, True)
            assert nm[i] == on and (m12[i, :len(pk)] == om12).all() and (m12[i, len(pk):] == -1).all(), (b, i)
            total += on
    assert total > 20
    st.close()


@pytest.mark.parametrize('seq', ['0', '1', 'cap1', 'cap3'])
def test_stream_matching_dense_clusters(api, oracle, seq, monkeypatch):
    """SearchForInitialization bookkeeping under stress: many level-0 keypoints in a small image, a window that covers a
    large part of it and nnratio 1.0, so that almost every query finds a match, keypoints are taken away from earlier
    queries all the time (ORBmatcher.cc:455-466) and the dependency chains between queries are long.  k_sfi_resolve
    iterates the bookkeeping to its fixed point; ORBFE_SFI_SEQUENTIAL=1 replays it serially; ORBFE_SFI_MAX_ROUNDS=1 / 3
    stops the fixed point early, so the kernel's serial finish (the bound on adversarial steal chains) does the work.
    All equal the oracle."""
    if seq == '1':
        monkeypatch.setenv('ORBFE_SFI_SEQUENTIAL', '1')
    if seq.startswith('cap'):
        monkeypatch.setenv('ORBFE_SFI_MAX_ROUNDS', seq[3:])
    W, H, N, nl, B = 420, 300, 1500, 2, 3
    base = synth(81, W, H)
    frames = [base] + [shifted(base, (3 * i) % 7 - 3, (5 * i) % 5 - 2, 800 + i) for i in range(1, 2 * B)]
    dev = api.DeviceFrames(frames, 0)
    ox = OracleExtractor(N, 1.2, nl, 20, 7, oracle)
    want = [ox.extract(f) for f in frames]
    bounds = (0.0, float(W), 0.0, float(H))
    total = 0
    for window, ratio, ori in ((150, 1.0, True), (60, 0.95, False), (400, 1.0, True)):
        st = api.Stream(N, 1.2, nl, 20, 7, 0, B, 2)
        st.set_matching(bounds, window, ratio, ori)
        for b in range(2):
            st.push_ptrs(dev.ptrs[b * B:(b + 1) * B], H, W, dev.stride, True)
        for b in range(2):
            kps, desc, n, m12, nm = st.pop(copy=True)
            for i in range(B):
                g = b * B + i
                wk, wd = want[g]
                assert n[i] == len(wk) and kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes()
                if g == 0:
                    continue
                pk, pd = want[g - 1]
                on, om12, _ = oracle.search_for_initialization(pk, pd, wk, wd, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2),
                                                               window, ratio, ori)
                assert nm[i] == on and (m12[i, :len(pk)] == om12).all(), (window, g)
                total += on
        st.close()
    assert total > 1000
    if seq == '1':
        monkeypatch.delenv('ORBFE_SFI_SEQUENTIAL')
    if seq.startswith('cap'):
        monkeypatch.delenv('ORBFE_SFI_MAX_ROUNDS')


_STAGING_VARIANT_SCRIPT = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from os1_amd import api
from os1_amd.synth import synth, shifted
from oracle.pyoracle import Oracle, OracleExtractor
o = Oracle()
for seed, W, H, N in ((71, 1280, 720, 1500), (72, 803, 601, 700)):
    img = synth(seed, W, H)
    imgs = [img, shifted(img, 3, 2, seed + 1), np.ascontiguousarray(img[::-1]), shifted(img, 7, 5, seed + 2)]
    ox = OracleExtractor(N, 1.2, 8, 20, 7, o)
    want = [ox.extract(im) for im in imgs]
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    for stride_pad in (0, 4, 3):            # a row stride that is / is not a multiple of 4 (the byte-wise staging paths)
        dev = api.DeviceFrames(imgs, 0, stride=W + stride_pad)
        kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, H, W, W + stride_pad, True)
        for i, (wk, wd) in enumerate(want):
            assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes(), (seed, stride_pad, i)
        dev.free()
    gk, gd = ex(img)
    assert gk.tobytes() == want[0][0].tobytes() and gd.tobytes() == want[0][1].tobytes()
print('OK')
'''


_ENV_PRESET_SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from os1_amd import api
from os1_amd.synth import synth, shifted
from oracle.pyoracle import Oracle, OracleExtractor
o = Oracle()
variant = int(os.environ.get('ORBFE_GAUSS_VARIANT', '0'))
img = synth(81, 960, 540)
imgs = [img, shifted(img, 3, 2, 82), shifted(img, 6, 4, 83)]
ox = OracleExtractor(900, 1.2, 8, 20, 7, o)
ox.set_gauss_variant(variant)
want = [ox.extract(im) for im in imgs]
ex = api.Extractor(900, 1.2, 8, 20, 7)          # reads ORBFE_GAUSS_VARIANT / ORBFE_POLL_WAIT_US when it is created
for im, (wk, wd) in zip(imgs, want):
    gk, gd = ex(im)
    assert gk.tobytes() == wk.tobytes() and gd.tobytes() == wd.tobytes()
for (gk, gd), (wk, wd) in zip(ex.extract_batch(imgs), want):
    assert gk.tobytes() == wk.tobytes() and gd.tobytes() == wd.tobytes()
dev = api.DeviceFrames(imgs, 0)
st = api.Stream(900, 1.2, 8, 20, 7, 0, 3, 2)    # the runner leaves ORBFE_POLL_WAIT_US alone when it is set
bounds = (0.0, 960.0, 0.0, 540.0)
st.set_matching(bounds, 100, 0.9, True)
for rep in range(3):
    st.push_ptrs(dev.ptrs, 540, 960, dev.stride, True)
prev = None
for rep in range(3):
    kps, desc, n, m12, nm = st.pop(copy=True)
    for i, (wk, wd) in enumerate(want):
        assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes()
        if prev is not None:       # SearchForInitialization of every frame against its predecessor in the stream (Tracking.cc:355-357, 384)
            pk, pd = prev
            on, om12, _ = o.search_for_initialization(pk, pd, wk, wd, bounds, np.stack([pk['x'], pk['y']], 1), 100, 0.9, True)
            assert nm[i] == on and (m12[i, :len(pk)] == om12).all(), (rep, i)
        prev = (wk, wd)
st.close()
print('OK')
'''


@pytest.mark.parametrize('env', [{'ORBFE_GAUSS_VARIANT': '1'}, {'ORBFE_GAUSS_VARIANT': '0'}, {'ORBFE_POLL_WAIT_US': '20'}, {'ORBFE_POLL_WAIT_US': '0'},
                                 {'ORBFE_ZERO_COPY': '0'}, {'ORBFE_ZERO_COPY': '2'}])
def test_environment_presets(env):
    """ORBFE_GAUSS_VARIANT presets the GaussianBlur variant of every extractor of the process (what an integrator without access to
    the facade's constructor sets), ORBFE_POLL_WAIT_US how a handle waits for the GPU (sleep-poll / spin), ORBFE_ZERO_COPY=0 sends every
    result through one copy command behind the call and =2 has the kernels store the results of batches into the page-locked arena too
    (default: one- and two-frame calls only): read when a handle
    is created, so each setting runs in a process of its own -- one-frame calls, a batch and the stream runner (with its matching chain)
    against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', _ENV_PRESET_SCRIPT % root], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), (env, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize('env', [{'ORBFE_LDS_DMA': '0'}, {'ORBFE_FAST_LEAN': '0'}, {'ORBFE_LDS_DMA': '0', 'ORBFE_FAST_LEAN': '0'}])
def test_staging_variants_agree(env):
    """How the tiles reach LDS: by LDS-DMA with 16 bytes per lane (default) or through registers (ORBFE_LDS_DMA=0: the pyramid footprints
    and the descriptor patches everywhere, not only on a frame's last row), and the FAST kernel's generic prologue instead of the LEAN one
    (ORBFE_FAST_LEAN=0; the script's odd row stride takes the generic one anyway).  Each switch is read once
    per process, so every setting runs in a process of its own; four-frame batches at two sizes and three row strides, and the one-frame
    route, against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', _STAGING_VARIANT_SCRIPT % root], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), (env, r.stdout[-500:], r.stderr[-1500:])
