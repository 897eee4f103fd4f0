"""Tracking-sized searches on a resident 1080p frame (GPU box): SearchByProjection(F, LastFrame, th = 15) with 2 000 sources and
SearchByProjection(F, MapPoints, th = 3) with 3 000 MapPoints through prepared C calls (what orb_shim.hpp pays, without Python
marshalling), median / p90 of 200 calls, the library's own stage clocks and the bookkeeping kernel's phase clocks."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from os1_amd import api
from os1_amd.synth import synth, shifted

W, H, N = 1920, 1080, 2000
A = synth(100, W, H)
B = shifted(A, 2, 1, 100)
ex = api.Extractor(N, 1.2, 8, 20, 7)
k1, d1 = ex(A)
k2, d2 = ex(B)
bounds = (0.0, float(W), 0.0, float(H))
fr = api.Frame.from_extract(ex, 0, bounds)
m = api.Matcher()
sf = np.ascontiguousarray(ex.tables()['sf'], np.float32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
occ = np.zeros(len(k2), np.uint8)
assigned = np.full(len(k2), -1, np.int32)
nm = C.c_int(0)
uv = np.stack([k1['x'] + 2, k1['y'] + 1], 1).astype(np.float32)
lvl1 = np.ascontiguousarray(k1['octave'], np.int32)
ang1 = np.ascontiguousarray(k1['angle'], np.float32)
valid = np.ones(len(k1), np.uint8)
sflags = np.full(len(k1), 8, np.uint8)
rng = np.random.default_rng(3)
src = rng.integers(0, len(k2), 3000)
mxy = (np.stack([k2['x'][src], k2['y'][src]], 1) + rng.uniform(-2, 2, (3000, 2))).astype(np.float32)
lvl = np.ascontiguousarray(k2['octave'][src], np.int32)
vcos = np.full(3000, 0.95, np.float32)
fl = np.full(3000, 9, np.uint8)
md = np.ascontiguousarray(d2[src])
a_uv = (m.h, fr.h, P(sf), len(sf), P(occ), P(uv), P(lvl1), P(ang1), P(sflags), P(valid), P(d1), len(k1), C.c_float(15.0), 100, 0, 1,
        P(assigned), C.byref(nm))
a_mp = (m.h, fr.h, P(sf), len(sf), P(occ), P(mxy), P(lvl), P(vcos), P(fl), P(md), 3000, C.c_float(3.0), C.c_float(0.8), P(assigned),
        C.byref(nm))
for name, fn, args in (('SearchByProjection(F, LastFrame, 15), %d sources' % len(k1), m.L.orbfe_search_by_projection_uv_frame, a_uv),
                       ('SearchByProjection(F, MapPoints, 3), 3000 MapPoints', m.L.orbfe_search_by_projection_frame, a_mp)):
    lat = []
    for _ in range(220):
        t0 = time.perf_counter()
        rc = fn(*args)
        lat.append(time.perf_counter() - t0)
        assert rc == 0
    lat = np.array(lat[20:]) * 1e3
    print('%s: median %.4f ms  p90 %.4f ms  (%d matches, %d rounds)  stages %s  phases %s' %
          (name, np.median(lat), np.percentile(lat, 90), nm.value, m.resolve_rounds(), np.round(m.stage_ms(), 4), m.resolve_phases()))
