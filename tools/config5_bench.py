"""BASELINE.json configs[4] timing (GPU box): 3840x2160 / 4000 features, equidistant-fisheye keypoint undistortion,
SearchByProjection against 10 000 MapPoints (th 1 and 5); CPU oracle beside it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from os1_amd import api
from os1_amd.synth import synth
from oracle.pyoracle import Oracle, OracleExtractor

W, H, N = 3840, 2160, 4000
fx = fy = 2196.0
cx, cy = 1839.0, 1155.0
oracle = Oracle()
img = synth(5, W, H)
ex = api.Extractor(N, 1.2, 8, 20, 7)
ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)


def t(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


k, d = ex(img)
print('extract 4K/4000: gpu %.2f ms (host frame, blocking call)   oracle %.0f ms' % (t(lambda: ex(img), 20), t(lambda: ox.extract(img), 1)))
B = 8
dev = api.DeviceFrames([img] * B, 0)
ms = t(lambda: ex.extract_batch_ptrs(dev.ptrs, H, W, dev.stride, True), 10)
print('extract batch of %d resident frames: %.2f ms  (%.0f frames/s)' % (B, ms, B / ms * 1e3))
xy0 = np.stack([k['x'], k['y']], 1)
print('undistort 4000 keypoints: %.3f ms' % t(lambda: api.undistort_equidistant(xy0, fx, fy, cx, cy), 50))
kun = k.copy()
xy = api.undistort_equidistant(xy0, fx, fy, cx, cy)
kun['x'], kun['y'] = xy[:, 0], xy[:, 1]
c = api.undistort_equidistant(np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32), fx, fy, cx, cy)
bounds = (float(min(c[0, 0], c[2, 0])), float(max(c[1, 0], c[3, 0])), float(min(c[0, 1], c[1, 1])), float(max(c[2, 1], c[3, 1])))
sf = ex.tables()['sf']
rng = np.random.default_rng(55)
n_mp = 10000
src = rng.integers(0, len(k), n_mp)
mdesc = d[src].copy()
for i in range(n_mp):
    for b in rng.integers(0, 256, rng.integers(0, 41)):
        mdesc[i, b >> 3] ^= np.uint8(1 << (b & 7))
mxy = (np.stack([kun['x'][src], kun['y'][src]], 1) + rng.uniform(-3, 3, (n_mp, 2))).astype(np.float32)
level = np.minimum(k['octave'][src] + rng.integers(0, 2, n_mp), 7).astype(np.int32)
viewcos = rng.uniform(0.9, 1.0, n_mp).astype(np.float32)
flags = np.full(n_mp, 1 | 8, np.uint8)
occ = np.zeros(len(k), np.uint8)
m = api.Matcher()
# resident frame (orbfe_frame_create_from_extract: nothing but 8 bytes per keypoint of undistorted coordinates goes up) and
# a prepared C call -- what a C++ caller (orb_shim.hpp) pays: no numpy marshalling inside the timed region
k, d = ex(img)
fr = api.Frame.from_extract(ex, 0, bounds, xy)
pin = api.PinnedArray((n_mp, 32), np.uint8)
pin.a[:] = mdesc
import ctypes as C
assigned = np.full(len(k), -1, np.int32)
nmat = C.c_int(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
sfa = np.ascontiguousarray(sf, np.float32)
th_c = {1.0: C.c_float(1.0), 5.0: C.c_float(5.0)}


# the argument list is converted once: a C++ caller passes plain pointers, ctypes' per-call conversions (about 1 us per array) are
# not part of the library
_fixed = {}


# the MapPoints' descriptors kept on the device by the caller (here: as the rows of a resident table built once)
_kp = np.zeros(n_mp, api.KP_DTYPE)
_table = api.Frame.from_host(m, _kp, mdesc, bounds)
dev_rows = _table.descriptors_device()


def c_call(th, rows):
    key = id(rows)
    if key not in _fixed:
        rp = C.c_void_p(rows.ptr) if isinstance(rows, api.DeviceRows) else P(rows)
        _fixed[key] = (m.h, fr.h, P(sfa), len(sfa), P(occ), P(mxy), P(level), P(viewcos), P(flags), rp, n_mp)
    rc = m.L.orbfe_search_by_projection_frame(*_fixed[key], th, 0.8, _out[0], _out[1])
    assert rc == 0


_out = (P(assigned), C.byref(nmat))


for th in (1.0, 5.0):
    want = oracle.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8)
    for name, rows in (('pageable descriptor rows', mdesc), ('page-locked descriptor rows', pin.a), ('device-resident descriptor rows', dev_rows)):
        c_call(th, rows)
        assert nmat.value == want[0] and (assigned == want[1]).all()
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            c_call(th, rows)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[20:]) * 1e3
        print('SearchByProjection on a RESIDENT frame, 10k MapPoints th=%g, %s: median %.4f ms  p90 %.4f ms  (%d matches, %d rounds)  stages %s'
              % (th, name, np.median(lat), np.percentile(lat, 90), nmat.value, m.resolve_rounds(), np.round(m.stage_ms(), 4)))
        print('   resolve route %d, phase clocks %s' % (m.resolve_route(), m.resolve_phases()))
for th in (1.0, 5.0):
    g = t(lambda: m.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8), 20)
    o = t(lambda: oracle.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8), 3)
    n = m.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8)[0]
    print('SearchByProjection 10k MapPoints th=%g: gpu %.3f ms  oracle %.3f ms  (%d matches)  stages %s' % (th, g, o, n, np.round(m.stage_ms(), 3)))
