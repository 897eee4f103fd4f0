"""Host-side product logic that needs no GPU, checked against the oracle: the quadtree and the
(cosf,sinf) restatement the rBRIEF kernel uses."""
import numpy as np
import pytest

from oracle.pyoracle import KP_DTYPE, OracleExtractor
from os1_amd import api
from os1_amd.synth import synth


def _as_kps(x, y, s):
    k = np.zeros(len(x), KP_DTYPE)
    k['x'], k['y'], k['response'], k['size'], k['angle'], k['class_id'] = x, y, s, 7, -1, -1
    return k


def _same(oracle, x, y, s, box, N):
    want = oracle.distribute_octtree(_as_kps(x, y, s), box[0], box[1], box[2], box[3], N)
    got = api.quadtree(x, y, s, box[0], box[1], box[2], box[3], N)
    assert len(got) == len(want)
    assert (x[got] == want['x']).all() and (y[got] == want['y']).all() and (s[got] == want['response']).all()
    return len(got)


def test_quadtree_on_real_candidates(oracle):
    img = synth(1, 640, 480)
    ex = OracleExtractor(1000, 1.2, 8, 20, 7, oracle)
    ex.extract(img)
    nf = ex.tables()['nfeat']
    for l in range(8):
        c = ex.candidates(l)
        h, w = ex.level(l).shape
        x, y, s = c['x'].astype(np.int16), c['y'].astype(np.int16), c['response'].astype(np.uint8)
        for N in (int(nf[l]), 1, 7, 50, 100000):
            _same(oracle, x, y, s, (16, w - 16, 16, h - 16), N)


@pytest.mark.parametrize('seed', range(12))
def test_quadtree_random(oracle, seed):
    rng = np.random.default_rng(seed)
    W, H = [(608, 448), (1888, 1048), (300, 900), (200, 130)][seed % 4]
    n = int(rng.integers(1, 6000))
    # distinct integer positions, reference order is irrelevant for the tree itself
    pos = rng.choice(W * H, size=min(n, W * H), replace=False)
    x, y = (pos % W).astype(np.int16), (pos // W).astype(np.int16)
    s = rng.integers(7, 40 if seed % 2 else 255, len(x)).astype(np.uint8)    # many response ties
    if W < H and round(W / H) == 0:
        pytest.skip('nIni == 0: the reference divides by zero here')
    for N in (1, 13, 200, 1000, 10 ** 6):
        k = _same(oracle, x, y, s, (16, 16 + W, 16, 16 + H), N)
        assert k <= max(N + 3, 4) or k <= len(x)


def test_quadtree_edge_cases(oracle):
    e = np.zeros(0, np.int16)
    assert len(api.quadtree(e, e, e.astype(np.uint8), 16, 624, 16, 464, 100)) == 0
    one = np.array([5], np.int16)
    assert api.quadtree(one, one, np.array([9], np.uint8), 16, 624, 16, 464, 100).tolist() == [0]
    # clustered points (all in one corner), forcing deep subdivision
    rng = np.random.default_rng(99)
    pos = rng.choice(40 * 40, 500, replace=False)
    x, y = (pos % 40).astype(np.int16), (pos // 40).astype(np.int16)
    s = rng.integers(7, 255, 500).astype(np.uint8)
    _same(oracle, x, y, s, (16, 1904, 16, 1064), 434)


def test_sincos_restatement_matches_libm_exhaustively():
    # every float in [0, 6.3] (the rBRIEF argument is angle*pi/180 in [0, 2*pi)): bitwise equal to libm
    hi = int(np.float32(6.3).view(np.uint32))
    assert api.sincos_host_mismatches(0, hi, 1) == 0


def test_fisheye_undistortion_matches_oracle(oracle):
    # Sony AS-20 720p fisheye calibration scaled x3
    fx = fy = 2196.0
    cx, cy = 1839.0, 1155.0
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(0, 3840, 5000), rng.uniform(0, 2160, 5000)], 1).astype(np.float32)
    pts[0] = (cx, cy)                                   # theta = 0 branch
    got = api.undistort_equidistant(pts, fx, fy, cx, cy)
    want = oracle.undistort_equidistant(pts, fx, fy, cx, cy)
    assert got.tobytes() == want.tobytes()
    assert got[0].tolist() == [cx, cy]
    r_in = np.hypot(pts[:, 0] - cx, pts[:, 1] - cy)
    r_out = np.hypot(got[:, 0] - cx, got[:, 1] - cy)
    assert (r_out >= r_in - 1e-2).all()                 # tan(theta)/theta >= 1: points move outwards


def test_pinhole_undistortion_and_image_bounds_match_oracle(oracle):
    """Frame::UndistortKeyPoints (pinhole branch, cv::undistortPoints) and Frame::ComputeImageBounds (Frame.cc:286-353):
    host double math of the C ABI against the oracle's restatement, all coefficient counts, both camera models."""
    from os1_amd import api
    rng = np.random.default_rng(3)
    xy = rng.uniform(-20, 780, (4000, 2)).astype(np.float32)
    K = (517.3, 516.5, 318.6, 255.3)
    for dist in ([], [0.0, 0.0, 0.0, 0.0], [0.262383, -0.953104, -0.005358, 0.002628, 1.163314], [-0.28, 0.07, 0.0002, 0.00002],
                 [-0.3, 0.1, 0.001, -0.002, 0.01, 0.02, -0.01, 0.001], [3.0, 9.0, 0.0, 0.0]):
        got = api.undistort_pinhole(xy, *K, dist)
        want = oracle.undistort_pinhole(xy, *K, dist)
        assert got.tobytes() == want.tobytes(), dist
        assert api.compute_image_bounds(752, 480, 0, *K, dist).tobytes() == oracle.image_bounds(752, 480, 0, *K, dist).tobytes()
    assert (api.undistort_pinhole(xy, *K, []) == xy).all()                     # no coefficients: identity up to rounding of K math
    assert list(api.compute_image_bounds(640, 480, 0, *K, [0.0, 0.1, 0, 0])) == [0.0, 640.0, 0.0, 480.0]   # k1 == 0: the image rectangle
    fish = (2196.0, 2196.0, 1839.0, 1155.0)
    assert api.compute_image_bounds(3840, 2160, 1, *fish).tobytes() == oracle.image_bounds(3840, 2160, 1, *fish).tobytes()


def test_bench_stream_definition():
    """os1_amd/stream_workload.py: frame i of stream `seed` is synth.shifted(base, 2i, i, seed*1000+i) (config 4: each
    frame = its predecessor shifted by (2,1) px), generated from one padded copy; the endless stream walks the pool
    forwards and backwards so consecutive frames always differ by one step; the committed digests cover 8 ranks."""
    import json
    import os
    from os1_amd import stream_workload as wl
    from os1_amd.synth import shifted
    sf = wl.StreamFrames(103, 320, 200, pool=40)
    for i in (1, 2, 17, 39):
        assert (sf.frame(i) == shifted(sf.base, 2 * i, i, 103 * 1000 + i)).all()
    assert sf.frame(0) is sf.base
    seq = [wl.pool_index(p, 5) for p in range(12)]
    assert seq == [0, 1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3]
    assert all(abs(a - b) == 1 for a, b in zip(seq, seq[1:]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, 'tests', 'golden', 'stream1080_digests.json')))
    assert sorted(d['streams']) == [str(wl.stream_seed(g)) for g in range(8)] and d['format'] == 2
    for v in d['streams'].values():      # per-position table: every pool frame, every forward and every backward pair
        assert len(v['frames']) == wl.POOL == len(v['fwd']) == len(v['nm_fwd']) and len(v['bwd']) == wl.POOL - 1 == len(v['nm_bwd'])
        assert len(set(v['frames'])) == wl.POOL and min(v['nm_fwd'][1:]) > 100 and min(v['nm_bwd']) > 100 and v['nm_fwd'][0] == 0
    assert d['workload']['batch'] == wl.BATCH and d['workload']['image'] == [wl.W, wl.H] and d['workload']['period'] == wl.PERIOD == 510
    # position -> (frame, predecessor): forwards, the turn-around at 255, backwards, the second period starts WITH a predecessor
    t = d['streams']['100']
    assert wl.expected_digests(t, 0) == (t['frames'][0], t['fwd'][0], 0)
    assert wl.expected_digests(t, 7)[:2] == (t['frames'][7], t['fwd'][7])
    assert wl.expected_digests(t, 255)[:2] == (t['frames'][255], t['fwd'][255])
    assert wl.expected_digests(t, 256)[:2] == (t['frames'][254], t['bwd'][254])
    assert wl.expected_digests(t, 509)[:2] == (t['frames'][1], t['bwd'][1])
    assert wl.expected_digests(t, 510)[:2] == (t['frames'][0], t['bwd'][0])
    assert wl.expected_digests(t, 511)[:2] == (t['frames'][1], t['fwd'][1])
    assert wl.expected_digests(t, 300, first_of_runner=True)[:2] == (t['frames'][wl.pool_index(300)], t['fwd'][0])
    steps, total = wl.expected_steps(t, 16)
    assert len(set(steps)) == 16 and total > 100000
    # digest helpers: order and content sensitive
    k = np.zeros(3, dtype=[('x', 'f4'), ('y', 'f4')])
    a = wl.frame_digest(k, np.zeros((3, 32), np.uint8), 3)
    k['x'][1] = 1
    assert a != wl.frame_digest(k, np.zeros((3, 32), np.uint8), 3)
    assert wl.match_digest(2, np.array([1, -1, 0]), 3) != wl.match_digest(2, np.array([1, 0, -1]), 3)


def test_shim_restated_ops_equal_the_oracle_restatement(oracle):
    """orbfe::detail::RestatedOps (include/orbfe/orb_shim.hpp: what the cv-free shim computes for `Rcw*x3Dw+tcw`,
    `-Rcw.t()*tcw`, cv::norm, Mat::dot, ORBmatcher.cc:293-298, 322-348, 1326-1343) and the oracle's cvGemm3 / cvGemmT3 /
    cvNorm3 / cvDot3 are two statements of the same recalled OpenCV arithmetic: they must agree bit for bit on random inputs
    over six decades (tests/test_opencv_live.py compares both with a real OpenCV where one exists)."""
    from test_opencv_live import _restated_ops
    shim = _restated_ops()
    rng = np.random.default_rng(11)
    for it in range(3000):
        sc = 10.0 ** rng.integers(-3, 4)
        A = (rng.standard_normal((3, 3)) * sc).astype(np.float32)
        b = (rng.standard_normal(3) * sc).astype(np.float32)
        c = (rng.standard_normal(3) * sc).astype(np.float32)
        al, be = float(rng.choice([1.0, -1.0, 0.5, 1.37])), float(rng.choice([1.0, 0.0, -1.0]))
        assert shim('gemm', A, b, al, c, be).tobytes() == oracle.cv_small('gemm', A, b, al, c, be).tobytes()
        assert shim('gemm', A, b, al, None, 0.0).tobytes() == oracle.cv_small('gemm', A, b, al, None, 0.0).tobytes()
        assert shim('gemmT', A, b, al).tobytes() == oracle.cv_small('gemmT', A, b, al).tobytes()
        assert shim('norm', A, b) == oracle.cv_small('norm', A, b)
        assert shim('dot', A.ravel()[:3], b) == oracle.cv_small('dot', A.ravel()[:3], b)
        s = float(rng.uniform(0.3, 3.0))
        assert (shim('scale', A, b, s) == A.ravel() * np.float32(s)).all()
        assert (shim('divide', A, b, s) == A.ravel() * np.float32(1.0 / s)).all()


def test_step_hasher_folds_any_submission_size():
    """The committed stream digests are per 32 frames; the bench submits 64 (stream_workload.SUBMIT).  StepHasher must give the same
    steps however the frames arrive."""
    from os1_amd import stream_workload as wl
    from oracle.pyoracle import KP_DTYPE
    rng = np.random.default_rng(5)
    nfr = 96
    n = rng.integers(3, 9, nfr)
    kps = np.zeros((nfr, 8), KP_DTYPE)
    kps['x'] = rng.random((nfr, 8))
    desc = rng.integers(0, 256, (nfr, 8, 32), dtype=np.uint8)
    nm = rng.integers(0, 3, nfr)
    m12 = rng.integers(-1, 5, (nfr, 8)).astype(np.int32)
    want = None
    for sub in (32, 64, 16, 96, 24):
        h = wl.StepHasher()
        for a in range(0, nfr, sub):
            h.add(kps[a:a + sub], desc[a:a + sub], n[a:a + sub], m12[a:a + sub], nm[a:a + sub])
        assert len(h.steps) == nfr // wl.BATCH
        want = want or h.steps
        assert h.steps == want and h.nmatches == int(nm.sum())


def test_position_checker_finds_what_step_digests_cannot_place():
    """PositionChecker (what bench.py uses for the period check and for the batches popped inside its timed region): a synthetic
    stream over a 6-frame pool whose outputs depend on (frame, predecessor) the way the runner's do.  Any submission size, a batch
    picked from the middle of the stream (with the predecessor's keypoint count handed in), a fresh runner in mid-stream; a flipped
    descriptor bit or match entry is reported at its position."""
    from os1_amd import stream_workload as wl
    from oracle.pyoracle import KP_DTYPE
    pool, cap = 6, 12
    rng = np.random.default_rng(9)
    nk = rng.integers(4, cap, pool)
    K = np.zeros((pool, cap), KP_DTYPE)
    K['x'] = rng.random((pool, cap))
    D = rng.integers(0, 256, (pool, cap, 32), dtype=np.uint8)

    def match(pred, cur):
        if pred is None:
            return 0, np.zeros(cap, np.int32)
        r = np.random.default_rng(pred * 16 + cur)
        return int(r.integers(1, 4)), r.integers(-1, 5, cap).astype(np.int32)
    table = {'frames': [wl.frame_digest(K[i], D[i], nk[i]) for i in range(pool)],
             'fwd': [wl.match_digest(*match(None, 0), 0)] + [wl.match_digest(*match(i - 1, i), nk[i - 1]) for i in range(1, pool)],
             'nm_fwd': [0] + [match(i - 1, i)[0] for i in range(1, pool)],
             'bwd': [wl.match_digest(*match(i + 1, i), nk[i + 1]) for i in range(pool - 1)], 'nm_bwd': [match(i + 1, i)[0] for i in range(pool - 1)]}

    def batch(p0, nb, fresh=False):
        idx = [wl.pool_index(p, pool) for p in range(p0, p0 + nb)]
        pred = [None if (p == 0 or (fresh and p == p0)) else wl.pool_index(p - 1, pool) for p in range(p0, p0 + nb)]
        mm = [match(a, b) for a, b in zip(pred, idx)]
        return K[idx].copy(), D[idx].copy(), nk[idx].astype(np.int32), np.stack([m for _, m in mm]), np.array([n for n, _ in mm], np.int32)
    for sub in (1, 3, 4, 10):
        c = wl.PositionChecker(table, pool)
        for p0 in range(0, 40, sub):
            c.check(p0, *batch(p0, sub), first_of_runner=(p0 == 0))
        assert not c.bad and c.frames >= 40 and len(c.positions) == 2 * pool - 2
    c = wl.PositionChecker(table, pool)
    assert c.check(23, *batch(23, 5), prev_n=int(nk[wl.pool_index(22, pool)])) and c.frames == 5
    assert wl.PositionChecker(table, pool).check(17, *batch(17, 4, fresh=True), first_of_runner=True)
    assert not wl.PositionChecker(table, pool).check(17, *batch(17, 4), first_of_runner=True)       # has a predecessor, checker told otherwise
    k, d, n, m, nm = batch(8, 4)
    d[2, 1, 5] ^= 4
    m[3, 0] += 1
    c = wl.PositionChecker(table, pool)
    c.check(8, k, d, n, m, nm, prev_n=int(nk[wl.pool_index(7, pool)]))
    assert c.bad == [(10, 'frame'), (11, 'match')]
    a = wl.PositionChecker(table, pool); b = wl.PositionChecker(table, pool)
    for p0 in range(0, 12, 3):
        a.check(p0, *batch(p0, 3), first_of_runner=(p0 == 0))
    for p0 in range(0, 12, 4):
        b.check(p0, *batch(p0, 4), first_of_runner=(p0 == 0))
    assert a.outputs_sha256(10) == b.outputs_sha256(10) and a.outputs_sha256(10) != a.outputs_sha256(9)


def test_inorder_gate_under_shuffled_completion(tmp_path):
    """The in-order completion queue of the multi-device stream runner (os1_amd/csrc/inorder_gate.h, SURVEY.md s8(e): "round-robin
    frames over GPUs with an in-order completion queue"): producers finish in shuffled order, the consumer sees 0, 1, 2, ...; nothing is
    recycled while the consumer holds it; closing the gate releases every waiter.  Host only."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'inorder_gate_test')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-Wall', '-Werror', '-pthread', '-I' + os.path.join(ROOT, 'os1_amd', 'csrc'),
                           os.path.join(ROOT, 'tests', 'cpp', 'inorder_gate_test.cpp'), '-o', exe])
    for lanes, total in ((1, 60), (2, 300), (4, 400), (8, 800)):
        r = subprocess.run([exe, str(lanes), str(total)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (lanes, r.stdout, r.stderr)
