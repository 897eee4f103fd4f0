"""Random sweep of the searches on device-resident frames against the oracle (GPU box): image sizes, feature counts, query
counts, radii from tracking-sized to image-sized (long candidate lists, many rounds of the bookkeeping), occupancy densities,
claim / skip / chi-square settings, all three search forms.  usage: sweep_matcher.py <seed> <configs>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd.synth import synth
from oracle.pyoracle import Oracle

seed, ncfg = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
o = Oracle()
m = api.Matcher()
bad = 0
rounds = []
for c in range(ncfg):
    W, H = int(rng.integers(160, 1400)), int(rng.integers(120, 900))
    N = int(rng.choice([60, 300, 1000, 2500]))
    nl = int(rng.integers(2, 9))
    ex = api.Extractor(N, 1.2, nl, 20, 7)
    try:
        k, d = ex(synth(int(rng.integers(1, 10**6)), W, H))
    except api.OrbfeError:
        continue                                  # image too small for the pyramid
    if len(k) == 0:
        continue
    b = (float(rng.uniform(-40, 0)), float(W + rng.uniform(0, 40)), float(rng.uniform(-30, 0)), float(H + rng.uniform(0, 30)))
    fr = None
    if rng.random() < 0.5:
        try:
            fr = api.Frame.from_extract(ex, 0, b)
        except api.OrbfeError:
            fr = None                             # a geometry outside the GPU quadtree's limits: its selections are not kept per slot
    if fr is None:
        fr = api.Frame.from_host(m, k, d, b)
    tab = ex.tables()
    sf = tab['sf']
    nq = int(rng.choice([1, 17, 400, 3000, 9000]))
    src = rng.integers(0, len(k), nq)
    qd = d[src].copy()
    flip = rng.integers(0, 60, nq)
    for i in range(nq):
        for bit in rng.integers(0, 256, flip[i]):
            qd[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
    spread = float(rng.choice([1.0, 4.0, 30.0]))
    xy = (np.stack([k['x'][src], k['y'][src]], 1) + rng.normal(0, spread, (nq, 2))).astype(np.float32)
    lvl = np.clip(k['octave'][src] + rng.integers(-1, 2, nq), 0, nl - 1).astype(np.int32)
    occ = (rng.random(len(k)) < rng.choice([0.0, 0.05, 0.5])).astype(np.uint8)
    th = float(rng.choice([1.0, 3.0, 8.0, 40.0]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        vc = rng.uniform(0.9, 1.0, nq).astype(np.float32)
        fl = np.full(nq, 1 | 8, np.uint8)
        fl[rng.random(nq) < 0.1] &= ~np.uint8(8)
        fl[rng.random(nq) < 0.05] |= 2
        fl[rng.random(nq) < 0.05] |= 4
        ratio = float(rng.choice([0.6, 0.8, 0.95]))
        g = m.search_by_projection(fr, None, None, sf, occ, xy, lvl, vc, fl, qd, th, ratio)
        w = o.search_by_projection(k, d, b, sf, occ, xy, lvl, vc, fl, qd, th, ratio)
        ok = g[0] == w[0] and (g[1] == w[1]).all()
    elif kind == 1:
        ang = rng.uniform(0, 360, nq).astype(np.float32)
        fl = np.where(rng.random(nq) < 0.8, 8, 0).astype(np.uint8)
        va = (rng.random(nq) < 0.9).astype(np.uint8)
        maxd, skip_any, ori = int(rng.choice([50, 100])), int(rng.integers(0, 2)), bool(rng.integers(0, 2))
        g = m.search_by_projection_uv(fr, None, None, sf, occ, xy, lvl, ang, fl, va, qd, th, maxd, skip_any, ori)
        w = o.search_by_projection_uv(k, d, b, sf, occ, xy, lvl, ang, fl, va, qd, th, maxd, skip_any, ori)
        ok = g[0] == w[0] and (g[1] == w[1]).all()
    else:
        va = (rng.random(nq) < 0.9).astype(np.uint8)
        rad = (th * sf[lvl]).astype(np.float32)
        claim, gate = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        skip = occ if rng.random() < 0.5 else None
        maxd = int(rng.choice([50, 100]))
        g = m.search_projected(fr, None, None, xy, rad, lvl, va, qd, skip, claim, tab['is2'] if gate else None, 5.99, maxd)
        w = o.search_projected(k, d, b, xy, rad, lvl, va, qd, skip, claim, tab['is2'] if gate else None, 5.99, maxd)
        ok = g[0] == w[0] and g[1].tobytes() == w[1].tobytes() and g[2].tobytes() == w[2].tobytes()
    rounds.append(m.resolve_rounds())
    if not ok:
        bad += 1
        print('MISMATCH cfg %d: %dx%d N=%d nl=%d n=%d nq=%d th=%g kind=%d' % (c, W, H, N, nl, len(k), nq, th, kind))
print('seed %d: %d configurations, %d mismatches; rounds of the bookkeeping: max %d, serial finishes %d' % (
    seed, len(rounds), bad, max(rounds) if rounds else 0, sum(1 for r in rounds if r < 0)))
sys.exit(1 if bad else 0)
