#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (regression pins for oracle AND product).

NOTE: these are outputs of THIS repository's oracle, not of the reference -- the reference cannot be
built here (no OpenCV) and ships no vectors, so parity stays "unpinned" (DESIGN.md s2).  The fixtures
freeze the current agreed behaviour so that an accidental change to the oracle or to the synthetic
generator is caught on CPU, and so the GPU path can be checked against committed numbers.
Run:  python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, OracleExtractor  # noqa: E402
from os1_amd.synth import shifted, synth  # noqa: E402

o = Oracle()
out = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(out, exist_ok=True)

# config 1 (BASELINE.json configs[0]): 640x480, N=1000, seed 1 -- plus the SearchForInitialization pair
A = synth(1, 640, 480)
B = shifted(A, -24, 3, 1)
ox = OracleExtractor(1000, 1.2, 8, 20, 7, o)
k1, d1 = ox.extract(A)
cand_counts = np.array([len(ox.candidates(l)) for l in range(8)], np.int32)
k2, d2 = ox.extract(B)
prev = np.stack([k1['x'], k1['y']], 1)
n, m12, p = o.search_for_initialization(k1, d1, k2, d2, (0, 640, 0, 480), prev, 100, 0.9, True)
np.savez_compressed(os.path.join(out, 'vga_seed1.npz'), kps1=k1, desc1=d1, kps2=k2, desc2=d2, cand_counts=cand_counts,
                    nmatches=np.int32(n), matches12=m12, prev_out=p,
                    frame_sha=np.frombuffer(__import__('hashlib').sha256(A.tobytes()).digest(), np.uint8))
print('vga_seed1: %d/%d kps, %d matches' % (len(k1), len(k2), n))

# configs 2 and 3 (BASELINE.json configs[1], configs[2]): 1080p, N=2000 -- digests only (the arrays are large)
import hashlib, json
def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
A = synth(2, 1920, 1080)
ox = OracleExtractor(2000, 1.2, 8, 20, 7, o)
k, d = ox.extract(A)
g = {'config2': {'seed': 2, 'n': int(len(k)), 'kps_desc_sha256': sha(k, d), 'frame_sha256': sha(A),
                 'cand_counts': [int(len(ox.candidates(l))) for l in range(8)]}}
A3 = synth(3, 1920, 1080)
B3 = shifted(A3, -24, 3, 3)
k1, d1 = ox.extract(A3)
k2, d2 = ox.extract(B3)
prev = np.stack([k1['x'], k1['y']], 1)
n, m12, p = o.search_for_initialization(k1, d1, k2, d2, (0, 1920, 0, 1080), prev, 100, 0.9, True)
g['config3'] = {'seed': 3, 'shift': [-24, 3], 'n1': int(len(k1)), 'n2': int(len(k2)), 'nmatches': int(n),
                'extract_sha256': sha(k1, d1, k2, d2), 'match_sha256': sha(m12, p)}
json.dump(g, open(os.path.join(out, 'hd1080_digests.json'), 'w'), indent=1)
print('hd1080 digests:', g['config2']['n'], g['config3']['nmatches'])

# bag of words and the vocabulary-grouped / projected searches on the VGA pair above (synthetic vocabulary, seed 3)
from os1_amd.synth import synth_vocabulary  # noqa: E402
voc = synth_vocabulary(3, 10, 4)
ov = o.vocabulary(voc)
_v = np.load(os.path.join(out, 'vga_seed1.npz'))
k1, d1, k2, d2 = _v['kps1'], _v['desc1'], _v['kps2'], _v['desc2']
ox = OracleExtractor(1000, 1.2, 8, 20, 7, o)
# BowVector / FeatureVector accumulate through the REFERENCE'S OWN DBoW2 BowVector.cpp / FeatureVector.cpp
# (oracle/_ref/libdbow2_vec.so, see oracle/Makefile); refuse to write the fixture otherwise.
assert o.use_dbow2_ref(True), 'oracle/_ref/libdbow2_vec.so missing: run make -C oracle where /root/reference exists'
t1, t2 = ov.transform(d1, 2), ov.transform(d2, 2)
o.use_dbow2_ref(False)
r1 = ov.transform(d1, 2)
assert r1[1].tobytes() == t1[1].tobytes() and r1[0].tolist() == t1[0].tolist()
rng = np.random.default_rng(17)
v1 = (rng.random(len(k1)) < 0.85).astype(np.uint8)
v2 = (rng.random(len(k2)) < 0.85).astype(np.uint8)
nb1, mb1 = o.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True)
nb2, mb2 = o.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], v2, t2[2], 0.75, True)
F12 = np.array([0, 0, 3e-3, 0, 0, 24e-3, -3e-3, -24e-3, 0], np.float32)      # [t]x of the (-24, +3) px shift
tab = ox.tables()
nt, tp = o.search_for_triangulation(k1, d1, v1, t1[2], k2, d2, v2, t2[2], F12, 320.0, 240.0, tab['sf'], tab['s2'], True)
src = rng.integers(0, len(k2), 1500)
uv = (np.stack([k2['x'][src], k2['y'][src]], 1) + rng.normal(0, 2.0, (1500, 2))).astype(np.float32)
lvl = np.clip(k2['octave'][src] + rng.integers(-1, 2, 1500), 0, 7).astype(np.int32)
rad = (4.0 * tab['sf'][lvl]).astype(np.float32)
sd = d2[src].copy()
for i in range(1500):
    for b in rng.integers(0, 256, rng.integers(0, 40)):
        sd[i, b >> 3] ^= np.uint8(1 << (b & 7))
ok = np.ones(1500, np.uint8)
np_, bi, bd = o.search_projected(k2, d2, (0, 640, 0, 480), uv, rad, lvl, ok, sd, None, True, tab['is2'], 5.99, 50)
np.savez_compressed(os.path.join(out, 'vga_seed1_bow.npz'),
                    bow1_ids=t1[0], bow1_vals=t1[1], fv1_nodes=t1[2][0], fv1_off=t1[2][1], fv1_feat=t1[2][2],
                    bow2_ids=t2[0], bow2_vals=t2[1], valid1=v1, valid2=v2,
                    sbb_kf_f=mb1, sbb_kf_f_n=np.int32(nb1), sbb_kf_kf=mb2, sbb_kf_kf_n=np.int32(nb2),
                    tri_pairs=tp, tri_n=np.int32(nt), F12=F12,
                    proj_uv=uv, proj_level=lvl, proj_radius=rad, proj_desc=sd, proj_best=bi, proj_dist=bd, proj_n=np.int32(np_),
                    bow_accumulated_by_reference=np.int32(1),
                    voc_sha=np.frombuffer(hashlib.sha256(voc).digest(), np.uint8))
print('vga_seed1_bow: %d words, SearchByBoW %d / %d, triangulation %d, projected %d' % (len(t1[0]), nb1, nb2, nt, np_))
