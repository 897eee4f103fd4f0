"""Developer timing helper (GPU box): batched SearchForInitialization cost."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd.synth import synth, shifted
B = 16
W, H, N = 1920, 1080, 2000
base = synth(100, W, H)
frames = [base] + [shifted(base, 2 * i, i, 100000 + i) for i in range(1, B)]
ex = api.Extractor(N, 1.2, 8, 20, 7)
feats = ex.extract_batch(frames)
mt = api.Matcher()
bounds = (0.0, float(W), 0.0, float(H))
pairs = []
for i in range(B):
    k1, d1 = feats[i - 1]; k2, d2 = feats[i]
    pairs.append((k1, d1, k2, d2, np.stack([k1['x'], k1['y']], 1)))
for _ in range(3): r = mt.search_for_initialization_batch(pairs, bounds)
t = time.time(); R = 20
for _ in range(R): r = mt.search_for_initialization_batch(pairs, bounds)
dt = (time.time() - t) / R
print('stages ms (arena, gpu, resolve):', mt.stage_ms())
print('batch of %d pairs: %.3f ms  (%.1f us/pair)  matches %s' % (B, dt * 1e3, dt / B * 1e6, [x[0] for x in r][:6]))
t = time.time()
for _ in range(R):
    for p in pairs[:4]: mt.search_for_initialization(p[0], p[1], p[2], p[3], bounds, p[4])
print('single pair: %.1f us' % ((time.time() - t) / R / 4 * 1e6))
