"""-m gpu: device-resident frames (orbfe_frame) and the searches that run on them, against the CPU oracle -- bit-exact.

Every windowed search is driven through all of its call forms:
  host     host arrays; the library builds a transient resident frame and keeps the bookkeeping on the GPU (default)
  hostres  host arrays, ORBFE_MATCH_HOST_RESOLVE=1: candidate lists to the host, bookkeeping there (the round-2 route)
  frame    an orbfe_frame created from host arrays once, searched repeatedly
  extract  an orbfe_frame created from the extractor's result arena (nothing but the queries is uploaded)
Reference: Frame::AssignFeaturesToGrid src/Frame.cc:114-129, GetFeaturesInArea :209-262, ORBmatcher::SearchByProjection
src/ORBmatcher.cc:45-132, :1292-1552, the projected loops :357-392 / :872-936, and the call sequence of
Tracking::TrackWithMotionModel + SearchLocalPoints, src/Tracking.cc:608-614, 824."""
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


def _grid_order(k, bounds):
    """Frame::AssignFeaturesToGrid restated with numpy: cell = ix * 48 + iy, insertion order inside a cell."""
    minx, maxx, miny, maxy = (np.float32(b) for b in bounds)
    invw = np.float32(64) / np.float32(maxx - minx)
    invh = np.float32(48) / np.float32(maxy - miny)
    fx = (k['x'] - minx) * invw
    fy = (k['y'] - miny) * invh
    px = np.where(fx >= 0, np.floor(fx + np.float32(0.5)), np.ceil(fx - np.float32(0.5))).astype(np.int64)   # round(): half away from zero
    py = np.where(fy >= 0, np.floor(fy + np.float32(0.5)), np.ceil(fy - np.float32(0.5))).astype(np.int64)
    ok = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    cell = px * 48 + py
    idx = np.nonzero(ok)[0]
    order = idx[np.argsort(cell[idx], kind='stable')]
    start = np.zeros(64 * 48 + 1, np.int64)
    np.add.at(start, cell[idx] + 1, 1)
    return order, np.cumsum(start)


def _routes(api, m, ex, k, d, bounds, monkeypatch, xy_un=None):
    """(name, first-argument factory) per call form; `ex` has just extracted (k, d) as frame 0 of its last batch."""
    fr_host = api.Frame.from_host(m, k, d, bounds)
    fr_ex = api.Frame.from_extract(ex, 0, bounds, xy_un)
    return [('host', k, None), ('hostres', k, ('ORBFE_MATCH_HOST_RESOLVE', '1')), ('frame', fr_host, None), ('extract', fr_ex, None),
            # marshalled query arrays + an upload instead of raw arrays read in place by the window kernel
            ('upload', fr_host, ('ORBFE_FRAME_ZEROCOPY', '0'))]


def test_frame_content_and_grid(api, oracle):
    W, H, N = 1280, 720, 1500
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k, d = ex(synth(12, W, H))
    m = api.Matcher()
    for bounds in [(0.0, float(W), 0.0, float(H)), (-211.5, 1500.25, -80.0, 799.0), (100.0, 900.0, 50.0, 500.0)]:
        worder, wstart = _grid_order(k, bounds)
        for fr in (api.Frame.from_host(m, k, d, bounds), api.Frame.from_extract(ex, 0, bounds)):
            gk, gd, order, cs = fr.download()
            assert len(fr) == len(k)
            for f in ('x', 'y', 'angle', 'octave'):
                assert gk[f].tobytes() == k[f].tobytes(), f
            assert gd.tobytes() == d.tobytes()
            assert cs.tolist() == wstart.tolist()
            assert order.tolist() == worder.tolist()
    # undistorted coordinates handed to the from-extract form replace pt
    xy = (np.stack([k['x'], k['y']], 1) * np.float32(1.01) + np.float32(3.5)).astype(np.float32)
    fr = api.Frame.from_extract(ex, 0, (0.0, 1400.0, 0.0, 800.0), xy)
    gk, gd, order, cs = fr.download()
    assert gk['x'].tobytes() == xy[:, 0].tobytes() and gk['y'].tobytes() == xy[:, 1].tobytes()
    k2 = k.copy()
    k2['x'], k2['y'] = xy[:, 0], xy[:, 1]
    worder, wstart = _grid_order(k2, (0.0, 1400.0, 0.0, 800.0))
    assert order.tolist() == worder.tolist() and cs.tolist() == wstart.tolist()
    # an empty frame
    fr = api.Frame.from_host(m, k[:0], d[:0], (0.0, 10.0, 0.0, 10.0))
    assert len(fr) == 0 and fr.download()[3][-1] == 0
    # frames of a batch: every index of the last collected batch is addressable, later calls invalidate nothing already built
    imgs = [synth(30 + i, 800, 600) for i in range(3)]
    ex2 = api.Extractor(700, 1.2, 8, 20, 7)
    res = ex2.extract_batch(imgs)
    frs = [api.Frame.from_extract(ex2, i, (0.0, 800.0, 0.0, 600.0)) for i in range(3)]
    ex2(imgs[0])                                              # the arena is reused; the frames live on
    for (kk, dd), fr in zip(res, frs):
        gk, gd, _, _ = fr.download()
        assert gk['x'].tobytes() == kk['x'].tobytes() and gd.tobytes() == dd.tobytes()
    with pytest.raises(api.OrbfeError):
        api.Frame.from_extract(ex2, 1, (0.0, 800.0, 0.0, 600.0))   # the last batch has one frame
    # a frame outlives the extractor it was taken from (its build was recorded on that extractor's stream)
    ex3 = api.Extractor(400, 1.2, 4, 20, 7)
    k3, d3 = ex3(imgs[1])
    fr3 = api.Frame.from_extract(ex3, 0, (0.0, 800.0, 0.0, 600.0))
    ex3.close()
    del ex3
    sf3 = api.Extractor(400, 1.2, 4, 20, 7).tables()['sf']
    occ3 = np.zeros(len(k3), np.uint8)
    q = np.stack([k3['x'], k3['y']], 1)
    got = m.search_by_projection(fr3, None, None, sf3, occ3, q, k3['octave'], np.ones(len(k3), np.float32),
                                 np.full(len(k3), 9, np.uint8), d3, 1.0, 0.8)
    want = oracle.search_by_projection(k3, d3, (0.0, 800.0, 0.0, 600.0), sf3, occ3, q, k3['octave'], np.ones(len(k3), np.float32),
                                       np.full(len(k3), 9, np.uint8), d3, 1.0, 0.8)
    assert got[0] == want[0] and (got[1] == want[1]).all() and got[0] > 300
    fr3.close()
    fr4 = api.Frame.from_host(m, k3, d3, (0.0, 800.0, 0.0, 600.0))    # no stale error is left behind for the next call
    assert len(fr4) == len(k3)


def test_search_by_projection_all_call_forms(api, oracle, monkeypatch):
    W, H, N = 1920, 1080, 2000
    img = synth(12, W, H)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k, d = ex(img)
    bounds = (0.0, float(W), 0.0, float(H))
    sf = ex.tables()['sf']
    m = api.Matcher()
    rng = np.random.default_rng(2)
    xy, level, viewcos, flags, mdesc = _mappoints(k, d, 5000, rng)
    occ = (rng.random(len(k)) < 0.1).astype(np.uint8)
    for name, first, env in _routes(api, m, ex, k, d, bounds, monkeypatch):
        if env:
            monkeypatch.setenv(*env)
        for th, ratio in [(1.0, 0.8), (5.0, 0.8), (3.0, 0.6), (12.0, 0.9)]:
            n, a = m.search_by_projection(first, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, th, ratio)
            on, oa = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, th, ratio)
            assert n == on and (a == oa).all(), (name, th)
            assert n > 500
        n, a = m.search_by_projection(first, d, bounds, sf, occ, xy[:0], level[:0], viewcos[:0], flags[:0], mdesc[:0], 1.0, 0.8)
        assert n == 0 and (a == -1).all()
        if env:
            monkeypatch.delenv(env[0])
    # every input array page-locked: the window kernel reads them where they are, nothing is staged
    arrays = (occ, xy, level.astype(np.int32), viewcos.astype(np.float32), flags, mdesc)
    pins = [api.PinnedArray(a.shape, a.dtype) for a in arrays]
    for p_, a_ in zip(pins, arrays):
        p_.a[...] = a_
    fr = api.Frame.from_host(m, k, d, bounds)
    for th in (1.0, 5.0):
        n, a = m.search_by_projection(fr, d, bounds, sf, *[p_.a for p_ in pins], th, 0.8)
        on, oa = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, th, 0.8)
        assert n == on and (a == oa).all(), th
    # an out-of-range level is an error only on an active MapPoint (ORBmatcher.cc:55-61 skips the others before using it)
    lv = level.astype(np.int32).copy()
    lv[7] = 99
    fl = flags.copy()
    fl[7] = 0
    n, a = m.search_by_projection(fr, d, bounds, sf, occ, xy, lv, viewcos, fl, mdesc, 1.0, 0.8)
    on, oa = oracle.search_by_projection(k, d, bounds, sf, occ, xy, np.where(np.arange(len(lv)) == 7, 0, lv), viewcos, fl, mdesc, 1.0, 0.8)
    assert n == on and (a == oa).all()
    fl[7] = 1
    with pytest.raises(api.OrbfeError):
        m.search_by_projection(fr, d, bounds, sf, occ, xy, lv, viewcos, fl, mdesc, 1.0, 0.8)


def test_search_by_projection_uv_and_projected_all_call_forms(api, oracle, monkeypatch):
    W, H, N = 1280, 720, 1500
    A = synth(21, W, H)
    B = shifted(A, -24, 3, 21)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k1, d1 = ex(A)
    k2, d2 = ex(B)                                            # frame 0 of the extractor's last batch = the searched frame
    bounds = (0.0, float(W), 0.0, float(H))
    tab = ex.tables()
    sf = tab['sf']
    m = api.Matcher()
    rng = np.random.default_rng(4)
    uv = np.stack([k1['x'] - 24 + rng.uniform(-2, 2, len(k1)), k1['y'] + 3 + rng.uniform(-2, 2, len(k1))], 1).astype(np.float32)
    valid = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    sflags = np.where(rng.random(len(k1)) < 0.9, 8, 0).astype(np.uint8)
    occ = (rng.random(len(k2)) < 0.05).astype(np.uint8)
    NS = 4000
    src = rng.integers(0, len(k2), NS)
    sdesc = d2[src].copy()
    for i in range(NS):
        for b in rng.integers(0, 256, rng.integers(0, 45)):
            sdesc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    puv = (np.stack([k2['x'][src], k2['y'][src]], 1) + rng.normal(0, 2.5, (NS, 2))).astype(np.float32)
    plevel = np.clip(k2['octave'][src] + rng.integers(-1, 2, NS), -1, 7).astype(np.int32)
    pvalid = (rng.random(NS) < 0.93).astype(np.uint8)
    kp_skip = (rng.random(len(k2)) < 0.2).astype(np.uint8)
    total = 0
    for name, first, env in _routes(api, m, ex, k2, d2, bounds, monkeypatch):
        if env:
            monkeypatch.setenv(*env)
        for th, maxd, skip_any, ori in [(15.0, 100, 0, True), (30.0, 100, 0, False), (10.0, 64, 1, True), (3.0, 100, 1, True),
                                        (60.0, 100, 0, True)]:
            n, a = m.search_by_projection_uv(first, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid, d1,
                                             th, maxd, skip_any, ori)
            on, oa = oracle.search_by_projection_uv(k2, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid,
                                                    d1, th, maxd, skip_any, ori)
            assert n == on and (a == oa).all(), (name, th)
            total += n
        for th, claim, skip, gate, maxd in [(4.0, True, True, False, 50), (3.0, False, False, True, 50), (2.5, False, False, False, 50),
                                            (7.5, False, False, False, 100), (10.0, True, False, True, 100), (25.0, True, True, True, 100)]:
            radius = (th * sf[np.clip(plevel, 0, 7)]).astype(np.float32)
            inv = tab['is2'] if gate else None
            sk = kp_skip if skip else None
            got = m.search_projected(first, d2, bounds, puv, radius, plevel, pvalid, sdesc, sk, claim, inv, 5.99, maxd)
            want = oracle.search_projected(k2, d2, bounds, puv, radius, plevel, pvalid, sdesc, sk, claim, inv, 5.99, maxd)
            assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes() and got[2].tobytes() == want[2].tobytes(), (name, th)
            total += got[0]
        if env:
            monkeypatch.delenv(env[0])
    assert total > 20000


def test_bookkeeping_bound_and_serial_finish(api, oracle, monkeypatch):
    """Many MapPoints per keypoint and wide windows: long chains of keypoints taken from later queries.  With the bound on
    the rounds of k_resolve at 1 / 2 the serial pass on the device finishes the job; results stay those of the oracle."""
    W, H, N = 640, 480, 600
    ex = api.Extractor(N, 1.2, 4, 20, 7)
    k, d = ex(synth(33, W, H))
    bounds = (0.0, float(W), 0.0, float(H))
    sf = ex.tables()['sf']
    m = api.Matcher()
    rng = np.random.default_rng(9)
    xy, level, viewcos, flags, mdesc = _mappoints(k, d, 6000, rng)
    level = np.minimum(level, 3)
    occ = np.zeros(len(k), np.uint8)
    fr = api.Frame.from_extract(ex, 0, bounds)
    want = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 8.0, 0.95)
    got = m.search_by_projection(fr, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 8.0, 0.95)
    assert got[0] == want[0] and (got[1] == want[1]).all()
    free_rounds = m.resolve_rounds()
    assert free_rounds >= 3                                   # the input does produce chains
    assert m.resolve_route() in (1, 2)                        # tables (and, when they fit, the entries) in LDS
    # the same kernel with its tables in global scratch (the route of problems too large for LDS)
    monkeypatch.setenv('ORBFE_RESOLVE_GENERIC', '1')
    got = m.search_by_projection(fr, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 8.0, 0.95)
    assert got[0] == want[0] and (got[1] == want[1]).all() and m.resolve_route() == 0 and m.resolve_rounds() == free_rounds
    monkeypatch.delenv('ORBFE_RESOLVE_GENERIC')
    for cap in ('1', '2'):
        monkeypatch.setenv('ORBFE_RESOLVE_MAX_ROUNDS', cap)
        got = m.search_by_projection(fr, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 8.0, 0.95)
        assert got[0] == want[0] and (got[1] == want[1]).all()
        assert m.resolve_rounds() == -int(cap)
        # the claiming projected loop through the same bound
        radius = (6.0 * sf[level]).astype(np.float32)
        ok = np.ones(len(level), np.uint8)
        g = m.search_projected(fr, d, bounds, xy, radius, level, ok, mdesc, None, True, None, 5.99, 100)
        w = oracle.search_projected(k, d, bounds, xy, radius, level, ok, mdesc, None, True, None, 5.99, 100)
        assert g[0] == w[0] and g[1].tobytes() == w[1].tobytes() and g[2].tobytes() == w[2].tobytes()
    monkeypatch.delenv('ORBFE_RESOLVE_MAX_ROUNDS')
    # the round tags of the claim table run out (here after a handful of rounds instead of four thousand): unsettled claims
    # are forgotten and the tags start over, chunk after chunk
    monkeypatch.setenv('ORBFE_RESOLVE_TAG_MAX', '57')
    got = m.search_by_projection(fr, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 8.0, 0.95)
    assert got[0] == want[0] and (got[1] == want[1]).all() and m.resolve_rounds() == free_rounds
    monkeypatch.delenv('ORBFE_RESOLVE_TAG_MAX')


def test_tracking_shaped_sequence_on_one_resident_frame(api, oracle):
    """Tracking::TrackWithMotionModel + SearchLocalPoints on one frame (src/Tracking.cc:608-614, 824): extract ->
    SearchByProjection(F, LastFrame, th) -> (too few matches) the same with 2*th -> SearchByProjection(F, MapPoints, th),
    each step seeing the mvpMapPoints the previous one left.  The frame is built once from the extractor's arena."""
    W, H, N = 1280, 720, 1500
    A = synth(61, W, H)
    B = shifted(A, -9, 4, 61)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k1, d1 = ex(A)
    k2, d2 = ex(B)
    bounds = (0.0, float(W), 0.0, float(H))
    sf = ex.tables()['sf']
    m = api.Matcher()
    fr = api.Frame.from_extract(ex, 0, bounds)
    rng = np.random.default_rng(7)
    uv = np.stack([k1['x'] - 9 + rng.uniform(-4, 4, len(k1)), k1['y'] + 4 + rng.uniform(-4, 4, len(k1))], 1).astype(np.float32)
    valid = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    sflags = np.where(rng.random(len(k1)) < 0.85, 8, 0).astype(np.uint8)

    def run(first):
        occ = np.zeros(len(k2), np.uint8)
        out = []
        for th in (7.0, 14.0):                                # Tracking.cc:608 (th), :614 (2*th, after fill(mvpMapPoints, NULL))
            n, a = (m if first is not None else oracle).search_by_projection_uv(
                first if first is not None else k2, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid, d1,
                th, 100, 0, True)
            out.append((n, a.copy()))
        # the frame's mvpMapPoints after the second search: assigned keypoints whose MapPoint has observations are occupied
        occ = ((a >= 0) & (sflags[np.clip(a, 0, len(k1) - 1)] & 8 > 0)).astype(np.uint8)
        mrng = np.random.default_rng(70)
        xy, level, viewcos, flags, mdesc = _mappoints(k2, d2, 3000, mrng)
        n, a3 = (m if first is not None else oracle).search_by_projection(
            first if first is not None else k2, d2, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 3.0, 0.8)   # Tracking.cc:824
        out.append((n, a3.copy()))
        return out
    got, want = run(fr), run(None)
    for (gn, ga), (wn, wa) in zip(got, want):
        assert gn == wn and (ga == wa).all()
    assert got[0][0] > 200 and got[2][0] > 300
    # LastFrame's descriptors are compared where they already are: the resident copy of the last frame (device memory)
    ex(A)
    last = api.Frame.from_extract(ex, 0, bounds)
    rows = last.descriptors_device()
    assert len(rows) == len(k1) and rows.ptr % 16 == 0
    occ = np.zeros(len(k2), np.uint8)
    for th in (7.0, 14.0):
        n, a = m.search_by_projection_uv(fr, None, None, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid, rows, th, 100, 0, True)
        wn, wa = oracle.search_by_projection_uv(k2, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid, d1, th, 100, 0, True)
        assert n == wn and (a == wa).all()
    # ... and so may any other per-query array (here: all of them page-locked)
    arrays = [occ, uv, k1['octave'].astype(np.int32), k1['angle'].astype(np.float32), sflags, valid]
    pins = [api.PinnedArray(x.shape, x.dtype) for x in arrays]
    for p_, x in zip(pins, arrays):
        p_.a[...] = x
    n, a = m.search_by_projection_uv(fr, None, None, sf, *[p_.a for p_ in pins], rows, 7.0, 100, 1, False)
    wn, wa = oracle.search_by_projection_uv(k2, d2, bounds, sf, occ, uv, k1['octave'], k1['angle'], sflags, valid, d1, 7.0, 100, 1, False)
    assert n == wn and (a == wa).all()
    last.close()


def test_config5_4k_fisheye_resident_frame(api, oracle):
    """BASELINE.json configs[4] on a resident frame: 3840x2160 / 4000 features extracted, fisheye undistortion of the
    keypoints on the host (8 bytes per keypoint go up), SearchByProjection against 10 000 MapPoints whose descriptor rows
    lie in page-locked memory."""
    W, H, N = 3840, 2160, 4000
    fx = fy = 2196.0
    cx, cy = 1839.0, 1155.0
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k, d = ex(synth(5, W, H))
    xy = api.undistort_equidistant(np.stack([k['x'], k['y']], 1), fx, fy, cx, cy)
    kun = k.copy()
    kun['x'], kun['y'] = xy[:, 0], xy[:, 1]
    bounds = api.compute_image_bounds(W, H, 1, fx, fy, cx, cy)
    fr = api.Frame.from_extract(ex, 0, bounds, xy)
    sf = ex.tables()['sf']
    rng = np.random.default_rng(55)
    mxy, level, viewcos, flags, mdesc = _mappoints(kun, d, 10000, rng)
    pin = api.PinnedArray((10000, 32), np.uint8)
    pin.a[:] = mdesc
    occ = np.zeros(len(k), np.uint8)
    m = api.Matcher()
    for th in (1.0, 5.0):
        on, oa = oracle.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8)
        for desc_rows in (mdesc, pin.a):
            n, a = m.search_by_projection(fr, None, None, sf, occ, mxy, level, viewcos, flags, desc_rows, th, 0.8)
            assert n == on and (a == oa).all()
        assert n > 1000


def test_frame_searches_from_three_threads(api, oracle):
    """One matcher + resident frame per host thread (a handle is single-threaded, handles are independent): every call's result
    is the oracle's while three threads search at once; the calls return on the kernels' completion word, not on the stream."""
    import threading
    W, H, N = 960, 540, 1000
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k, d = ex(synth(77, W, H))
    bounds = (0.0, float(W), 0.0, float(H))
    sf = ex.tables()['sf']
    occ = np.zeros(len(k), np.uint8)
    cases = []
    for t in range(3):
        rng = np.random.default_rng(100 + t)
        xy, level, viewcos, flags, mdesc = _mappoints(k, d, 2500 + 700 * t, rng)
        want = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 2.0 + t, 0.8)
        cases.append((xy, level, viewcos, flags, mdesc, 2.0 + t, want))
    bad = [0, 0, 0]

    def work(t):
        m = api.Matcher()
        fr = api.Frame.from_host(m, k, d, bounds)
        xy, level, viewcos, flags, mdesc, th, want = cases[t]
        for _ in range(150):
            n, a = m.search_by_projection(fr, None, None, sf, occ, xy, level, viewcos, flags, mdesc, th, 0.8)
            if n != want[0] or not (a == want[1]).all():
                bad[t] += 1
        fr.close()
    ths = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert bad == [0, 0, 0]


def test_candidate_pool_grows_on_demand(api, oracle):
    """Image-sized windows: every source lists hundreds of candidates, far beyond a fresh matcher's candidate pool.  The
    bookkeeping kernel reports the demand through its header, the call submits again with a pool of that size (the kernels'
    running counters are back at zero), and the lists -- all of them walked from global memory: they do not fit LDS -- give
    the oracle's result."""
    W, H, N = 800, 600, 900
    ex = api.Extractor(N, 1.2, 6, 20, 7)
    k, d = ex(synth(91, W, H))
    bounds = (0.0, float(W), 0.0, float(H))
    m = api.Matcher()                                          # fresh: the pool has its initial size
    fr = api.Frame.from_extract(ex, 0, bounds)
    rng = np.random.default_rng(5)
    NS = 2500
    src = rng.integers(0, len(k), NS)
    sdesc = d[src].copy()
    puv = np.stack([k['x'][src], k['y'][src]], 1).astype(np.float32)
    plevel = np.clip(k['octave'][src] + rng.integers(0, 2, NS), 0, 5).astype(np.int32)
    radius = np.full(NS, 2000.0, np.float32)
    ok = np.ones(NS, np.uint8)
    for claim in (False, True):
        got = m.search_projected(fr, None, None, puv, radius, plevel, ok, sdesc, None, claim, None, 5.99, 256)
        want = oracle.search_projected(k, d, bounds, puv, radius, plevel, ok, sdesc, None, claim, None, 5.99, 256)
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes() and got[2].tobytes() == want[2].tobytes()
        assert m.resolve_route() == 1                          # tables in LDS, lists in the pool
    assert got[0] > 800


def test_search_by_projection_indexed_descriptor_table(api, oracle, monkeypatch):
    """orbfe_search_by_projection_frame_rows: the MapPoints' descriptors are rows of a table the caller keeps on the device
    (the local map across frames, Tracking.cc:818-824).  Rows referenced in a permuted order, some read from the device
    copy and some -- changed since the last upload -- from the page-locked mirror (bit 31); then a descriptor is mutated
    between two searches (MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:227-292) and must be seen.  Results equal the
    oracle's on the gathered rows; the marshalled route (ORBFE_FRAME_ZEROCOPY=0) gathers from the mirror."""
    W, H, N = 1280, 720, 1500
    img = synth(14, W, H)
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k, d = ex(img)
    bounds = (0.0, float(W), 0.0, float(H))
    sf = ex.tables()['sf']
    m = api.Matcher()
    rng = np.random.default_rng(21)
    n_mp = 4000
    xy, level, viewcos, flags, mdesc = _mappoints(k, d, n_mp, rng)
    occ = (rng.random(len(k)) < 0.1).astype(np.uint8)
    fr = api.Frame.from_host(m, k, d, bounds)
    tab = api.DescTable(6000)
    slot = rng.permutation(6000)[:n_mp].astype(np.int64)          # MapPoint i lives in row slot[i]: order unrelated to the query order
    tab.host.a[slot] = mdesc
    tab.upload(m, 0, 6000)
    want = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 1.0, 0.8)
    rows = slot.astype(np.int32)
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, rows, 1.0, 0.8)
    assert n == want[0] and (a == want[1]).all() and n > 400
    # 300 descriptors change: the mirror has them, the device copy does not -> those rows are flagged for the mirror
    changed = rng.choice(n_mp, 300, replace=False)
    mdesc2 = mdesc.copy()
    mdesc2[changed] ^= rng.integers(0, 256, (300, 32), dtype=np.uint8)
    tab.host.a[slot[changed]] = mdesc2[changed]
    rows2 = rows.copy()
    rows2[changed] = (slot[changed] | 0x80000000).astype(np.uint32).view(np.int32)
    want2 = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc2, 3.0, 0.8)
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, rows2, 3.0, 0.8)
    assert n == want2[0] and (a == want2[1]).all()
    assert not (want2[1] == oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 3.0, 0.8)[1]).all()
    # stale device rows really are stale: the same search WITHOUT the flags gives the old descriptors' result
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, rows, 3.0, 0.8)
    old = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 3.0, 0.8)
    assert n == old[0] and (a == old[1]).all()
    # after the asynchronous upload of the changed rows the plain indices see the new bytes
    for s_ in np.sort(slot[changed]):
        tab.upload(m, int(s_), int(s_) + 1)
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, rows, 3.0, 0.8)
    assert n == want2[0] and (a == want2[1]).all()
    # page-locked index array, read in place
    pr = api.PinnedArray(rows.shape, np.int32)
    pr.a[:] = rows2
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, pr.a, 3.0, 0.8)
    assert n == want2[0] and (a == want2[1]).all()
    # the marshalled route gathers the rows from the mirror
    monkeypatch.setenv('ORBFE_FRAME_ZEROCOPY', '0')
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, rows2, 3.0, 0.8)
    monkeypatch.delenv('ORBFE_FRAME_ZEROCOPY')
    assert n == want2[0] and (a == want2[1]).all()
    # a row index outside the table is refused for a MapPoint that is searched (both routes), ignored for one that is not
    bad = rows2.copy()
    victim = int(np.nonzero((flags & 1) != 0)[0][7])
    for r in (6000, 6000 | -0x80000000, 0x7fffffff):
        bad[victim] = np.array([r], np.int64).astype(np.int32)[0] if r >= 0 else np.int32(r)
        for zc in ('1', '0'):
            monkeypatch.setenv('ORBFE_FRAME_ZEROCOPY', zc)
            with pytest.raises(api.OrbfeError) as e:
                m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, flags, tab, bad, 3.0, 0.8)
            assert e.value.code == -1 and 'outside the table' in str(e.value)
        monkeypatch.delenv('ORBFE_FRAME_ZEROCOPY')
    fl2 = flags.copy()
    fl2[victim] = 0                                     # not in view: its row index is never used
    w3 = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, fl2, mdesc2, 3.0, 0.8)
    n, a = m.search_by_projection_rows(fr, sf, occ, xy, level, viewcos, fl2, tab, bad, 3.0, 0.8)
    assert n == w3[0] and (a == w3[1]).all()
    m.synchronize()
    tab.free()


def test_device_pointers_are_refused_for_host_read_arrays(api, oracle):
    """include/orbfe.h: of a `_frame` search's input arrays only the descriptor rows may live in device memory -- the
    library reads the others on the host too (level range, largest radius).  A device pointer for one of them must come back
    as ORBFE_ERR_INVALID before anything dereferences it (it used to be dereferenced: a segfault); device-resident
    descriptor rows of the bag-of-words search that the kernels cannot take (misaligned) are refused, not memcpy'd."""
    import ctypes as C
    W, H, N = 960, 540, 800
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    k, d = ex(synth(15, W, H))
    bounds = (0.0, float(W), 0.0, float(H))
    sf = ex.tables()['sf']
    m = api.Matcher()
    rng = np.random.default_rng(3)
    xy, level, viewcos, flags, mdesc = _mappoints(k, d, 500, rng)
    occ = np.zeros(len(k), np.uint8)
    fr = api.Frame.from_host(m, k, d, bounds)
    L = m.L
    dev = C.c_void_p()
    api._check(L.orbfe_device_malloc(0, 1 << 16, C.byref(dev)))
    assigned = np.full(len(k), -1, np.int32)
    n = C.c_int(0)
    args = [api._p(occ), api._p(xy), api._p(level.astype(np.int32)), api._p(viewcos), api._p(flags)]
    keep = [level.astype(np.int32)]
    args[2] = api._p(keep[0])
    for bad in range(5):
        a = list(args)
        a[bad] = dev
        rc = L.orbfe_search_by_projection_frame(m.h, fr.h, api._p(sf), len(sf), a[0], a[1], a[2], a[3], a[4], api._p(mdesc), 500,
                                                C.c_float(1.0), C.c_float(0.8), api._p(assigned), C.byref(n))
        assert rc != 0 and b'device memory' in L.orbfe_last_error(), bad
    # the untouched call still works and equals the oracle
    rc = L.orbfe_search_by_projection_frame(m.h, fr.h, api._p(sf), len(sf), *args, api._p(mdesc), 500, C.c_float(1.0), C.c_float(0.8),
                                            api._p(assigned), C.byref(n))
    on, oa = oracle.search_by_projection(k, d, bounds, sf, occ, xy, level, viewcos, flags, mdesc, 1.0, 0.8)
    assert rc == 0 and n.value == on and (assigned == oa).all()
    # bag of words: misaligned device rows
    fv = (np.zeros(1, np.uint32), np.array([0, 1], np.uint32), np.zeros(1, np.uint32))
    m12 = np.full(len(k), -1, np.int32)
    ang = np.zeros(len(k), np.float32)
    valid = np.ones(len(k), np.uint8)
    rc = L.orbfe_search_by_bow(m.h, C.c_void_p(dev.value + 4), api._p(ang), api._p(valid), 8, api._p(fv[0]), api._p(fv[1]), api._p(fv[2]), 1,
                               api._p(d), api._p(ang), None, 8, api._p(fv[0]), api._p(fv[1]), api._p(fv[2]), 1, C.c_float(0.7), 0, 0,
                               api._p(m12), C.byref(n))
    assert rc != 0 and b'16-byte aligned' in L.orbfe_last_error()
    L.orbfe_device_free(0, dev)
