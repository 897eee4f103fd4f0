"""CPU: the oracle's DBoW2 transform / SearchByBoW restatements against independent pure-Python versions."""
import numpy as np
import pytest

from bow_util import fv_to_dict, py_transform, ragged_vocabulary, with_header
from os1_amd.synth import synth_vocabulary


def _descs(seed, image, n):
    """Descriptors near vocabulary nodes (random node descriptors with a few bit flips) plus pure noise."""
    rng = np.random.default_rng(seed)
    rec = np.frombuffer(image, np.uint8, offset=4).reshape(-1, 45)
    pick = rec[rng.integers(0, len(rec), n), 5:37].copy()
    bits = np.unpackbits(pick, axis=1)
    flip = rng.random(bits.shape) < 0.05
    out = np.packbits(bits ^ flip, axis=1)
    out[::7] = rng.integers(0, 256, (len(out[::7]), 32), dtype=np.uint8)
    return out


@pytest.mark.parametrize('scoring,weighting', [(0, 0), (1, 0), (5, 1), (0, 2), (2, 3), (5, 0)])
def test_transform_matches_python_restatement(oracle, scoring, weighting):
    for image, levelsups in [(synth_vocabulary(3, 10, 3), (0, 1, 2, 3, 5)), (ragged_vocabulary(4), (0, 1, 2, 4))]:
        image = with_header(image, scoring, weighting)
        v = oracle.vocabulary(image)
        d = _descs(scoring * 10 + weighting, image, 300)
        for lu in levelsups:
            ids, vals, fv, wof, nof = v.transform(d, lu)
            pids, pvals, pfv, pw, pn = py_transform(image, d, lu)
            assert ids.tolist() == pids
            assert vals.tolist() == pvals          # doubles, bit for bit
            assert fv_to_dict(fv) == pfv
            assert wof.tolist() == pw and nof.tolist() == pn


def test_restated_accumulation_equals_reference_dbow2(oracle):
    """The oracle's restated BowVector / FeatureVector accumulation against the REFERENCE'S OWN BowVector.cpp +
    FeatureVector.cpp (oracle/_ref/libdbow2_vec.so, built from /root/reference by oracle/Makefile): every weighting /
    scoring combination, doubles bit for bit.  Skipped only where the reference-built object is absent."""
    if not oracle.have_dbow2_ref():
        pytest.skip('oracle/_ref/libdbow2_vec.so not built (no /root/reference on this box)')
    try:
        for scoring in range(6):
            for weighting in range(4):
                for image, lu in [(synth_vocabulary(3, 10, 3, stop_fraction=0.1), 1), (ragged_vocabulary(4), 2)]:
                    image = with_header(image, scoring, weighting)
                    v = oracle.vocabulary(image)
                    d = _descs(scoring * 10 + weighting + 100, image, 400)
                    assert not oracle.use_dbow2_ref(False) and not oracle.bow_accumulator_is_reference()
                    a = v.transform(d, lu)
                    assert oracle.use_dbow2_ref(True) and oracle.bow_accumulator_is_reference()
                    b = v.transform(d, lu)
                    assert a[0].tolist() == b[0].tolist()
                    assert a[1].tobytes() == b[1].tobytes()
                    assert all(x.tolist() == y.tolist() for x, y in zip(a[2], b[2]))
    finally:
        oracle.use_dbow2_ref(False)


def test_golden_bow_vector_was_accumulated_by_reference_code(oracle):
    """tests/golden/vga_seed1_bow.npz carries BowVector / FeatureVector produced through the reference's DBoW2 code
    (flag in the fixture); the restatement must reproduce them here, wherever the test runs."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vga_seed1_bow.npz'))
    assert int(g['bow_accumulated_by_reference']) == 1
    v0 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vga_seed1.npz'))
    ov = oracle.vocabulary(synth_vocabulary(3, 10, 4))
    oracle.use_dbow2_ref(False)
    t1 = ov.transform(v0['desc1'], 2)
    assert t1[0].tolist() == g['bow1_ids'].tolist() and t1[1].tobytes() == g['bow1_vals'].tobytes()
    assert t1[2][0].tolist() == g['fv1_nodes'].tolist() and t1[2][2].tolist() == g['fv1_feat'].tolist()


def test_transform_empty_and_stopped(oracle):
    image = synth_vocabulary(5, 4, 2, stop_fraction=1.0)      # every word stopped
    v = oracle.vocabulary(image)
    ids, vals, fv, wof, nof = v.transform(_descs(1, image, 20), 1)
    assert len(ids) == 0 and len(fv[0]) == 0 and len(wof) == 20
    ids, vals, fv, _, _ = v.transform(np.zeros((0, 32), np.uint8), 1)
    assert len(ids) == 0 and len(fv[0]) == 0


def _py_search_by_bow(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori, strict):
    """Both SearchByBoW overloads, written from the reference's description with python containers."""
    def dist(x, y):
        return int(np.unpackbits(x ^ y).sum())
    f1, f2 = fv_to_dict(fv1), fv_to_dict(fv2)
    m12 = {}
    taken = set()
    hist = [[] for _ in range(30)]
    for node in sorted(set(f1) & set(f2)):
        for i1 in f1[node]:
            if not v1[i1]:
                continue
            b1, b2, bi = 256, 256, -1
            for i2 in f2[node]:
                if i2 in taken or (v2 is not None and not v2[i2]):
                    continue
                dd = dist(d1[i1], d2[i2])
                if dd < b1:
                    b2, b1, bi = b1, dd, i2
                elif dd < b2:
                    b2 = dd
            ok = b1 < 50 if strict else b1 <= 50
            if ok and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                m12[i1] = bi
                taken.add(bi)
                if ori:
                    rot = np.float32(a1[i1]) - np.float32(a2[bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    hist[_bin(rot)].append(i1)
    if ori:
        cnt = [len(h) for h in hist]
        order = _three_maxima(cnt)
        for b in range(30):
            if b in order:
                continue
            for i1 in hist[b]:
                del m12[i1]
    return m12


def _round_half_away(x):
    import math
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def _bin(rot):
    b = _round_half_away(float(np.float32(rot) * np.float32(1.0 / 30)))
    return 0 if b == 30 else b


def _three_maxima(cnt):
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(cnt):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s
            i3, i2 = i2, i
        elif s > max3:
            max3, i3 = s, i
    if max2 < np.float32(0.1) * np.float32(max1):
        i2 = i3 = -1
    elif max3 < np.float32(0.1) * np.float32(max1):
        i3 = -1
    return {i1, i2, i3}


def make_bow_pair(seed, image, n1=400, n2=450):
    """Two descriptor sets sharing many (noisy) descriptors, with angles and validity flags."""
    rng = np.random.default_rng(seed)
    d1 = _descs(seed, image, n1)
    d2 = _descs(seed + 1, image, n2)
    m = min(n1, n2) * 2 // 3
    src = rng.permutation(n1)[:m]
    dst = rng.permutation(n2)[:m]
    bits = np.unpackbits(d1[src], axis=1)
    d2[dst] = np.packbits(bits ^ (rng.random(bits.shape) < 0.04), axis=1)
    a1 = (rng.random(n1) * 360).astype(np.float32)
    a2 = (rng.random(n2) * 360).astype(np.float32)
    a2[dst] = np.mod(a1[src] - np.float32(40.0) + rng.normal(0, 6, m).astype(np.float32), np.float32(360.0)).astype(np.float32)
    a2[a2 >= 360] = 0
    v1 = (rng.random(n1) < 0.8).astype(np.uint8)
    v2 = (rng.random(n2) < 0.9).astype(np.uint8)
    return d1, a1, v1, d2, a2, v2


@pytest.mark.parametrize('levelsup', [1, 3])
def test_search_by_bow_matches_python_restatement(oracle, levelsup):
    image = synth_vocabulary(6, 10, 3)
    v = oracle.vocabulary(image)
    d1, a1, v1, d2, a2, v2 = make_bow_pair(7, image)
    fv1, fv2 = v.transform(d1, levelsup)[2], v.transform(d2, levelsup)[2]
    for ratio, ori in [(0.7, True), (0.9, False), (0.75, True)]:
        nm, m12 = oracle.search_by_bow(d1, a1, v1, fv1, d2, a2, None, fv2, ratio, ori)
        want = _py_search_by_bow(d1, a1, v1, fv1, d2, a2, None, fv2, ratio, ori, False)
        assert {i: int(j) for i, j in enumerate(m12) if j >= 0} == want and nm == len(want)
        assert nm > 20
        nm, m12 = oracle.search_by_bow(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori)
        want = _py_search_by_bow(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori, True)
        assert {i: int(j) for i, j in enumerate(m12) if j >= 0} == want and nm == len(want)


def _with_trailing_duplicate(image):
    """What the reference's loadFromBinaryFile BUILDS from a vocabulary file (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
    1604-1640): it loops `while(!f.eof())`, and eof is only raised by the read AFTER the last record -- that read fails, leaves
    the 45-byte buffer as it was and one more node is appended: a copy of the last record (same parent, leaf flag, descriptor,
    weight).  As a file image: the last record twice."""
    return image + image[-45:]


@pytest.mark.parametrize('k,L', [(10, 3), (4, 4), (7, 2)])
def test_reference_loader_trailing_duplicate_changes_nothing(oracle, k, L):
    """The duplicate node sits behind its original among the children of their parent, and the descent takes a child only on a
    STRICTLY smaller distance (TemplatedVocabulary.h:1328-1336): it can never be chosen -- BowVector, FeatureVector and the
    per-feature (word, node) pairs of a vocabulary loaded the reference's way equal those of the file as written, also for
    descriptors that ARE the last word's descriptor (distance 0 to both copies)."""
    image = synth_vocabulary(5 + k, k, L)
    dup = _with_trailing_duplicate(image)
    assert len(dup) == len(image) + 45
    a, b = oracle.vocabulary(image), oracle.vocabulary(dup)
    d = _descs(3, image, 1200)
    last_word = np.frombuffer(image[-45 + 5:-45 + 37], np.uint8)
    d[:40] = last_word                                   # exact hits of the duplicated word
    for i in range(40, 80):                              # ... and near misses
        d[i] = last_word
        d[i, i % 32] ^= np.uint8(1 << (i % 8))
    for lu in (0, 1, L, L + 2):
        ta, tb = a.transform(d, lu), b.transform(d, lu)
        assert ta[0].tobytes() == tb[0].tobytes() and ta[1].tobytes() == tb[1].tobytes()
        for x, y in zip(ta[2], tb[2]):
            assert x.tobytes() == y.tobytes()
        assert ta[3].tobytes() == tb[3].tobytes() and ta[4].tobytes() == tb[4].tobytes()
