"""-m gpu: bag-of-words on the GPU (orbfe_bow_transform, orbfe_search_by_bow) against the CPU oracle -- bit-exact,
including the double-precision BowVector values."""
import numpy as np
import pytest

from bow_util import ragged_vocabulary, with_header
from os1_amd.synth import shifted, synth, synth_vocabulary
from test_bow_oracle import _descs, make_bow_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def api():
    from os1_amd import api as a
    assert a.device_count() >= 1, 'no GPU visible: the product has no CPU fallback'
    return a


def _same_transform(got, want):
    gi, gv, gfv, gw, gn = got
    wi, wv, wfv, ww, wn = want
    assert gi.tobytes() == wi.tobytes()
    assert gv.tobytes() == wv.tobytes()            # doubles, bit for bit
    for a, b in zip(gfv, wfv):
        assert a.tobytes() == b.tobytes()
    assert gw.tobytes() == ww.tobytes() and gn.tobytes() == wn.tobytes()


@pytest.mark.parametrize('scoring,weighting', [(0, 0), (1, 1), (5, 0), (3, 2), (0, 3)])
def test_bow_transform_parity(api, oracle, scoring, weighting):
    for image, levelsups in [(synth_vocabulary(3, 10, 4), (0, 2, 4, 6)), (ragged_vocabulary(4), (0, 1, 2, 4)),
                             (synth_vocabulary(8, 19, 2), (0, 1))]:
        image = with_header(image, scoring, weighting)
        v = api.Vocabulary(image)
        ov = oracle.vocabulary(image)
        assert v.info()['n_nodes'] == (len(image) - 4) // 45 + 1
        for n in (1, 17, 1500):
            d = _descs(n, image, n)
            for lu in levelsups:
                _same_transform(v.transform(d, lu), ov.transform(d, lu))
        got = v.transform(np.zeros((0, 32), np.uint8), 2)
        assert len(got[0]) == 0 and len(got[2][0]) == 0
        v.close()


def test_compute_bow_of_extracted_frames_full_size_vocabulary(api, oracle):
    """Frame::ComputeBoW on real descriptors with a k=10, L=6 vocabulary (1.1 M nodes, the size of ORBvoc), levelsup 4;
    then both SearchByBoW overloads between a frame and its shifted successor."""
    image = synth_vocabulary(1, 10, 6)
    v = api.Vocabulary(image)
    ov = oracle.vocabulary(image)
    ex = api.Extractor(2000, 1.2, 8, 20, 7)
    A = synth(3, 1920, 1080)
    B = shifted(A, -24, 3, 33)
    (k1, d1), (k2, d2) = ex(A), ex(B)
    t1, t2 = v.transform(d1, 4), v.transform(d2, 4)
    _same_transform(t1, ov.transform(d1, 4))
    _same_transform(t2, ov.transform(d2, 4))
    assert len(t1[0]) > 1000 and abs(t1[1].sum() - 1.0) < 1e-9
    m = api.Matcher()
    rng = np.random.default_rng(0)
    v1 = (rng.random(len(k1)) < 0.85).astype(np.uint8)
    v2 = (rng.random(len(k2)) < 0.85).astype(np.uint8)
    for ratio, ori in [(0.7, True), (0.75, False)]:
        for valid2, strict in [(None, False), (v2, True)]:
            nm, m12 = m.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], valid2, t2[2], ratio, ori, strict)
            wn, w12 = oracle.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], valid2, t2[2], ratio, ori)
            assert nm == wn and m12.tobytes() == w12.tobytes()
    v.close()


@pytest.mark.parametrize('levelsup', [0, 1, 2, 3])
def test_search_by_bow_parity(api, oracle, levelsup, monkeypatch):
    """Synthetic descriptor sets with many true correspondences; levelsup 3 on an L=3 tree puts every feature under
    the root (one group of several hundred features per side: the kernel's walk computes its distances itself, inputs go up by
    a copy command); levelsup 2 with 2 000 features gives ten groups of about 200 x 220 (lists and descriptors in LDS, the
    distance matrix in tiles of rows); lower levels: small groups, everything in LDS, no copy command at all."""
    image = synth_vocabulary(6, 10, 3)
    ov = oracle.vocabulary(image)
    m = api.Matcher()
    if levelsup == 1:
        monkeypatch.setenv('ORBFE_BOW_ZEROCOPY', '0')          # small groups through the upload / download route too
    for seed, n1, n2 in [(7, 400, 450), (9, 1, 300), (10, 900, 70), (11, 2000, 2200)]:
        d1, a1, v1, d2, a2, v2 = make_bow_pair(seed, image, n1, n2)
        fv1, fv2 = ov.transform(d1, levelsup)[2], ov.transform(d2, levelsup)[2]
        total = 0
        for ratio, ori in [(0.7, True), (0.9, False), (0.6, True)]:
            for valid2, strict in [(None, False), (v2, True)]:
                nm, m12 = m.search_by_bow(d1, a1, v1, fv1, d2, a2, valid2, fv2, ratio, ori, strict)
                wn, w12 = oracle.search_by_bow(d1, a1, v1, fv1, d2, a2, valid2, fv2, ratio, ori)
                assert nm == wn and m12.tobytes() == w12.tobytes()
                total += nm
        assert total > 0 or n1 == 1
    # empty sides
    e = (np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros(0, np.uint32))
    nm, m12 = m.search_by_bow(d1, a1, v1, fv1, np.zeros((0, 32), np.uint8), np.zeros(0, np.float32), None, e)
    assert nm == 0 and (m12 == -1).all()


def test_search_by_bow_with_resident_descriptor_rows(api, oracle):
    """SearchByBoW(KeyFrame, Frame) with the descriptor rows of one side, the other, or both read from resident frames (device
    memory, orbfe_frame_descriptors_device): the same matches as with host rows and as the oracle."""
    from os1_amd.synth import synth, shifted
    W, H = 1280, 720
    image = synth_vocabulary(6, 10, 4)
    v = api.Vocabulary(image)
    ex = api.Extractor(1200, 1.2, 8, 20, 7)
    A = synth(44, W, H)
    k1, d1 = ex(A)
    bounds = (0.0, float(W), 0.0, float(H))
    fr1 = api.Frame.from_extract(ex, 0, bounds)
    k2, d2 = ex(shifted(A, -7, 3, 44))
    fr2 = api.Frame.from_extract(ex, 0, bounds)
    m = api.Matcher()
    fv1, fv2 = v.transform(d1, 2)[2], v.transform(d2, 2)[2]
    v1 = np.ones(len(k1), np.uint8)
    want = oracle.search_by_bow(d1, k1['angle'], v1, fv1, d2, k2['angle'], None, fv2, 0.7, True)
    assert want[0] > 100
    for a, b in [(d1, d2), (fr1.descriptors_device(), d2), (d1, fr2.descriptors_device()), (fr1.descriptors_device(), fr2.descriptors_device())]:
        got = m.search_by_bow(a, k1['angle'], v1, fv1, b, k2['angle'], None, fv2, 0.7, True)
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes()
    v.close()


def test_search_by_bow_batch_is_the_relocalisation_loop(api, oracle):
    """Tracking::Relocalization (Tracking.cc:1005-1030): SearchByBoW of several candidate keyframes against the current frame,
    one GPU submission; every keyframe's result equals its own oracle call (and the single-keyframe entry point)."""
    image = synth_vocabulary(6, 10, 3)
    ov = oracle.vocabulary(image)
    m = api.Matcher()
    sides, frame = [], None
    for seed, n1 in [(21, 500), (22, 1), (23, 800), (24, 0), (25, 350)]:
        d1, a1, v1, d2, a2, v2 = make_bow_pair(seed, image, max(n1, 1), 600)
        if n1 == 0:
            d1, a1, v1 = d1[:0], a1[:0], v1[:0]
        if frame is None:
            frame = (d2, a2, ov.transform(d2, 1)[2])
        sides.append((d1, a1, v1, ov.transform(d1, 1)[2] if len(d1) else (np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros(0, np.uint32))))
    d2, a2, fv2 = frame
    total = 0
    for ratio, ori in [(0.75, True), (0.9, False)]:
        got = m.search_by_bow_batch(sides, d2, a2, None, fv2, ratio, ori, False)
        assert len(got) == len(sides)
        for (d1, a1, v1, fv1), (nm, m12) in zip(sides, got):
            wn, w12 = oracle.search_by_bow(d1, a1, v1, fv1, d2, a2, None, fv2, ratio, ori) if len(d1) else (0, np.zeros(0, np.int32))
            assert nm == wn and m12.tobytes() == w12.tobytes()
            if len(d1):
                sn, s12 = m.search_by_bow(d1, a1, v1, fv1, d2, a2, None, fv2, ratio, ori, False)
                assert sn == nm and s12.tobytes() == m12.tobytes()
            total += nm
    assert total > 100
    assert m.search_by_bow_batch([], d2, a2, None, fv2) == []


def test_search_for_triangulation_parity(api, oracle, monkeypatch):
    """LocalMapping::CreateNewMapPoints' matcher: two extracted frames related by an image translation t, F12 = [t]x
    (true correspondences lie on their epipolar lines), an epipole placed inside the image so the epipole-distance
    test rejects some, random 'already has a MapPoint' flags."""
    image = synth_vocabulary(2, 10, 4)
    ov = oracle.vocabulary(image)
    ex = api.Extractor(1500, 1.2, 8, 20, 7)
    A = synth(12, 1280, 720)
    B = shifted(A, -18, 5, 77)
    (k1, d1), (k2, d2) = ex(A), ex(B)
    tab = ex.tables()
    m = api.Matcher()
    rng = np.random.default_rng(4)
    total = 0
    # levelsup 2 / 3: groups of about 15 / 150 features (lists, descriptors and coordinates in LDS, four waves per group, inputs read
    # from the page-locked arena); levelsup 4 on this L = 4 tree: ONE group of 1 500 x 1 500 (four waves, everything from global memory,
    # inputs uploaded); the last case runs the small groups through the upload route as well
    for levelsup, t, epi, zc in [(2, (18.0, -5.0), (600.0, 300.0), None), (3, (-18.0, 5.0), (-1e4, 50.0), None),
                                 (4, (18.0, -5.0), (600.0, 300.0), None), (2, (3.0, 40.0), (640.0, 360.0), '0')]:
        if zc:
            monkeypatch.setenv('ORBFE_BOW_ZEROCOPY', zc)
        fv1, fv2 = ov.transform(d1, levelsup)[2], ov.transform(d2, levelsup)[2]
        tx, ty = t
        F12 = np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float32) * np.float32(1e-3)
        h1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
        h2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
        for ori in (True, False):
            nm, pairs = m.search_for_triangulation(k1, d1, h1, fv1, k2, d2, h2, fv2, F12, epi[0], epi[1], tab['sf'],
                                                   tab['s2'], ori)
            wn, wp = oracle.search_for_triangulation(k1, d1, h1, fv1, k2, d2, h2, fv2, F12, epi[0], epi[1], tab['sf'],
                                                     tab['s2'], ori)
            assert nm == wn and pairs.tobytes() == wp.tobytes()
            total += nm
        if zc:
            monkeypatch.delenv('ORBFE_BOW_ZEROCOPY')
    assert total > 200
    # zero fundamental matrix: den == 0 everywhere -> nothing matches
    nm, _ = m.search_for_triangulation(k1, d1, h1, fv1, k2, d2, h2, fv2, np.zeros(9, np.float32), 0.0, 0.0, tab['sf'], tab['s2'])
    assert nm == 0


def test_compute_bow_fused_into_the_extractor(api, oracle):
    """orbfe_extractor_set_vocabulary: the descent runs behind the descriptor kernel on descriptors still in HBM; every
    frame of a batch gets the same BowVector / FeatureVector as the stand-alone transform and the oracle."""
    image = synth_vocabulary(4, 10, 5)
    v = api.Vocabulary(image)
    ov = oracle.vocabulary(image)
    ex = api.Extractor(900, 1.2, 8, 20, 7)
    ex.set_vocabulary(v, 3)
    frames = [synth(30 + i, 800, 600) for i in range(3)] + [np.full((600, 800), 77, np.uint8)]   # the last has no keypoint
    dev = api.DeviceFrames(frames, 0)
    kps, desc, n = ex.extract_batch_ptrs(dev.ptrs, 600, 800, dev.stride, True)
    for f in range(4):
        got = ex.bow(f, int(n[f]))
        want = ov.transform(desc[f, :n[f]], 3)
        _same_transform(got, want)
        if n[f]:
            _same_transform(got, v.transform(desc[f, :n[f]], 3))
    assert n[3] == 0 and n[0] > 500
    # single-frame call, other levelsup, then switched off again
    ex.set_vocabulary(v, 1)
    k, d = ex(frames[1])
    _same_transform(ex.bow(0, len(k)), ov.transform(d, 1))
    ex.set_vocabulary(None)
    ex(frames[1])
    with pytest.raises(api.OrbfeError):
        ex.bow(0, len(k))
    v.close()


def test_stream_runner_with_vocabulary(api, oracle):
    """orbfe_stream_set_vocabulary: every popped frame carries its (leaf, node) pairs; assembling them gives the oracle's
    BowVector / FeatureVector of that frame's descriptors."""
    image = synth_vocabulary(9, 10, 4)
    v = api.Vocabulary(image)
    ov = oracle.vocabulary(image)
    W, H, N, B = 800, 600, 700, 3
    base = synth(61, W, H)
    frames = [base] + [shifted(base, 3 * i, -2 * i, 900 + i) for i in range(1, 2 * B)]
    dev = api.DeviceFrames(frames, 0)
    st = api.Stream(N, 1.2, 8, 20, 7, 0, B, 2)
    st.set_matching((0.0, float(W), 0.0, float(H)), 100, 0.9, True)
    st.set_vocabulary(v, 2)
    for b in range(2):
        st.push_ptrs(dev.ptrs[b * B:(b + 1) * B], H, W, dev.stride, True)
    for b in range(2):
        kps, desc, n, m12, nm = st.pop(copy=True)
        for i in range(B):
            leaf, node = st.bow_raw(i)
            assert len(leaf) == n[i]
            _same_transform(v.assemble(leaf, node), ov.transform(desc[i, :n[i]], 2))
    st.close()
    v.close()


def test_search_by_bow_large_nodes_and_exhausted_candidate_lists(api, oracle):
    """Nodes beyond 256 features per side take k_bow_topk + the single-wave walk (round 4): every frame-1 feature's eight least
    keys are computed in parallel, the in-order walk (ORBmatcher.cc:196-222) then decides from them and the matched bitmap.
    (a) 2 000 x 2 000 features under ONE node -- the degenerate grouping of a vocabulary with L <= levelsup, which used to cost
    6.8 ms; (b) heavy competition: 40 distinct descriptors in ~15 near-copies each on both sides, so that most features find
    seven or eight of their eight nearest already taken and rescan the node (the fallback); (c) several large nodes and small
    ones in one call, an invalid-flag mix, nodes larger than the 4 096 features k_bow_topk stages in LDS."""
    import time
    m = api.Matcher()
    rng = np.random.default_rng(77)

    def one_node(n):
        return (np.array([3], np.uint32), np.array([0, n], np.uint32), np.arange(n, dtype=np.uint32))

    def noisy(base, flips):
        d = base.copy()
        for i in range(len(d)):
            for b in rng.integers(0, 256, rng.integers(0, flips + 1)):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        return d

    # (a) 2000 x 2000, mostly true correspondences with 0-30 flipped bits
    n = 2000
    d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d2 = noisy(d1[rng.permutation(n)], 30)
    a1 = rng.uniform(0, 360, n).astype(np.float32)
    a2 = rng.uniform(0, 360, n).astype(np.float32)
    v1 = (rng.random(n) < 0.9).astype(np.uint8)
    v2 = (rng.random(n) < 0.9).astype(np.uint8)
    for ratio, ori, valid2, strict in [(0.7, False, None, False), (0.9, True, v2, True)]:
        got = m.search_by_bow(d1, a1, v1, one_node(n), d2, a2, valid2, one_node(n), ratio, ori, strict)
        want = oracle.search_by_bow(d1, a1, v1, one_node(n), d2, a2, valid2, one_node(n), ratio, ori)
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes() and got[0] > 300
    t = []
    for _ in range(12):
        t0 = time.perf_counter()
        m.search_by_bow(d1, a1, v1, one_node(n), d2, a2, None, one_node(n), 0.7, False, False)
        t.append(time.perf_counter() - t0)
    ms = float(np.median(t[2:])) * 1e3
    print('SearchByBoW, 2000 x 2000 features under one node: %.3f ms per call' % ms)
    assert ms < 1.0          # 6.8 ms before round 4; measured 0.2 - 0.3 ms
    # (b) competition: few distinct descriptors, many near-copies -> exhausted lists
    proto = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    n1, n2 = 620, 580
    d1 = noisy(proto[rng.integers(0, 40, n1)], 3)
    d2 = noisy(proto[rng.integers(0, 40, n2)], 3)
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = rng.uniform(0, 360, n2).astype(np.float32)
    v1 = np.ones(n1, np.uint8)
    for ratio in (0.95, 1.0, 0.6):
        got = m.search_by_bow(d1, a1, v1, one_node(n1), d2, a2, None, one_node(n2), ratio, False, False)
        want = oracle.search_by_bow(d1, a1, v1, one_node(n1), d2, a2, None, one_node(n2), ratio, False)
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes(), ratio
    # (c) three large nodes (one beyond the LDS staging limit) and forty small ones in one call
    sizes1 = [300, 5000, 700] + [int(x) for x in rng.integers(1, 40, 40)]
    sizes2 = [900, 4500, 260] + [int(x) for x in rng.integers(1, 40, 40)]
    n1, n2 = sum(sizes1), sum(sizes2)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    src = rng.integers(0, n1, n2)
    d2 = noisy(d1[src], 25)
    p1, p2 = rng.permutation(n1).astype(np.uint32), rng.permutation(n2).astype(np.uint32)
    fv1 = (np.arange(len(sizes1), dtype=np.uint32) * 2 + 1, np.concatenate([[0], np.cumsum(sizes1)]).astype(np.uint32), p1)
    fv2 = (np.arange(len(sizes2), dtype=np.uint32) * 2 + 1, np.concatenate([[0], np.cumsum(sizes2)]).astype(np.uint32), p2)
    # the reference's lists hold ascending feature indices inside a node
    for fv, sizes in ((fv1, sizes1), (fv2, sizes2)):
        o = fv[1]
        for i in range(len(sizes)):
            fv[2][o[i]:o[i + 1]].sort()
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = rng.uniform(0, 360, n2).astype(np.float32)
    v1 = (rng.random(n1) < 0.8).astype(np.uint8)
    v2 = (rng.random(n2) < 0.8).astype(np.uint8)
    for valid2, strict in [(None, False), (v2, True)]:
        got = m.search_by_bow(d1, a1, v1, fv1, d2, a2, valid2, fv2, 0.8, True, strict)
        want = oracle.search_by_bow(d1, a1, v1, fv1, d2, a2, valid2, fv2, 0.8, True)
        assert got[0] == want[0] and got[1].tobytes() == want[1].tobytes()


def test_vocabulary_image_with_the_reference_loaders_trailing_duplicate(api, oracle):
    """orbfe_vocabulary_create_from_image on a file image as the reference's loadFromBinaryFile effectively sees it -- the last
    record twice (its `while(!eof)` loop appends a copy of the last node, TemplatedVocabulary.h:1604-1640): the descent can never
    choose the copy (strict `<`), so the GPU transform equals the transform of the plain image and the oracle's."""
    from test_bow_oracle import _with_trailing_duplicate
    image = synth_vocabulary(9, 10, 4)
    dup = _with_trailing_duplicate(image)
    v, vd, ov = api.Vocabulary(image), api.Vocabulary(dup), oracle.vocabulary(image)
    assert vd.info()['n_nodes'] == v.info()['n_nodes'] + 1
    d = _descs(5, image, 1500)
    d[:30] = np.frombuffer(image[-45 + 5:-45 + 37], np.uint8)
    for lu in (0, 2, 4):
        want = ov.transform(d, lu)
        _same_transform(v.transform(d, lu), want)
        _same_transform(vd.transform(d, lu), want)
    v.close(); vd.close()
