"""Committed regression vectors (tests/golden, made by tools/gen_golden.py from the oracle)."""
import hashlib
import os

import numpy as np
import pytest

from oracle.pyoracle import OracleExtractor
from os1_amd.synth import shifted, synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vga_seed1.npz')


def _frames():
    A = synth(1, 640, 480)
    return A, shifted(A, -24, 3, 1)


def test_oracle_reproduces_golden(oracle):
    g = np.load(G)
    A, B = _frames()
    assert hashlib.sha256(A.tobytes()).digest() == g['frame_sha'].tobytes()       # generator unchanged
    ox = OracleExtractor(1000, 1.2, 8, 20, 7, oracle)
    k1, d1 = ox.extract(A)
    assert [len(ox.candidates(l)) for l in range(8)] == g['cand_counts'].tolist()
    k2, d2 = ox.extract(B)
    assert k1.tobytes() == g['kps1'].tobytes() and d1.tobytes() == g['desc1'].tobytes()
    assert k2.tobytes() == g['kps2'].tobytes() and d2.tobytes() == g['desc2'].tobytes()
    prev = np.stack([k1['x'], k1['y']], 1)
    n, m12, p = oracle.search_for_initialization(k1, d1, k2, d2, (0, 640, 0, 480), prev, 100, 0.9, True)
    assert n == int(g['nmatches']) and (m12 == g['matches12']).all() and p.tobytes() == g['prev_out'].tobytes()


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    from os1_amd import api
    g = np.load(G)
    A, B = _frames()
    ex = api.Extractor(1000, 1.2, 8, 20, 7)
    k1, d1 = ex(A)
    k2, d2 = ex(B)
    assert k1.tobytes() == g['kps1'].tobytes() and d1.tobytes() == g['desc1'].tobytes()
    assert k2.tobytes() == g['kps2'].tobytes() and d2.tobytes() == g['desc2'].tobytes()
    prev = np.stack([k1['x'], k1['y']], 1)
    n, m12, p = api.Matcher().search_for_initialization(k1, d1, k2, d2, (0, 640, 0, 480), prev, 100, 0.9, True)
    assert n == int(g['nmatches']) and (m12 == g['matches12']).all() and p.tobytes() == g['prev_out'].tobytes()


def _sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.mark.gpu
def test_gpu_reproduces_1080p_digests():
    """BASELINE.json configs[1] and configs[2] at full size against committed digests (tools/gen_golden.py)."""
    import json
    from os1_amd import api
    g = json.load(open(os.path.join(os.path.dirname(G), 'hd1080_digests.json')))
    ex = api.Extractor(2000, 1.2, 8, 20, 7)
    A = synth(2, 1920, 1080)
    assert _sha(A) == g['config2']['frame_sha256']
    k, d = ex(A)
    assert len(k) == g['config2']['n'] and _sha(k, d) == g['config2']['kps_desc_sha256']
    assert [len(ex.candidates(l)) for l in range(8)] == g['config2']['cand_counts']
    A3 = synth(3, 1920, 1080)
    B3 = shifted(A3, -24, 3, 3)
    (k1, d1), (k2, d2) = ex(A3), ex(B3)
    assert _sha(k1, d1, k2, d2) == g['config3']['extract_sha256']
    prev = np.stack([k1['x'], k1['y']], 1)
    n, m12, p = api.Matcher().search_for_initialization(k1, d1, k2, d2, (0, 1920, 0, 1080), prev, 100, 0.9, True)
    assert n == g['config3']['nmatches'] and _sha(m12, p) == g['config3']['match_sha256']
