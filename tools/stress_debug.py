"""Exploratory robustness run (GPU box): handle churn, size changes, pending work at destroy, two threads."""
import sys
import threading
import time
sys.path.insert(0, '.')
import numpy as np
from os1_amd import api
from os1_amd.synth import synth
from oracle.pyoracle import Oracle, OracleExtractor

oracle = Oracle()
imgs = {(W, H): synth(W + H, W, H) for (W, H) in [(640, 480), (800, 600), (1280, 720), (333, 251)]}
want = {}
ox = OracleExtractor(600, 1.2, 8, 20, 7, oracle)
for k, im in imgs.items():
    want[k] = ox.extract(im)


def same(got, w):
    return got[0].tobytes() == w[0].tobytes() and got[1].tobytes() == w[1].tobytes()


# 1. one handle, sizes alternate; many calls
ex = api.Extractor(600, 1.2, 8, 20, 7)
ok = True
for it in range(40):
    k = list(imgs)[it % 4]
    ok &= same(ex(imgs[k]), want[k])
print('alternating sizes on one handle:', ok)

# 2. handle churn
t0 = time.time()
for it in range(60):
    e = api.Extractor(600, 1.2, 8, 20, 7)
    ok &= same(e(imgs[(640, 480)]), want[(640, 480)])
    e.close()
print('60 create/extract/destroy cycles: %s  %.2f s' % (ok, time.time() - t0))

# 3. destroy with a submitted, uncollected batch
dev = api.DeviceFrames([imgs[(800, 600)]] * 4, 0)
for it in range(5):
    e = api.Extractor(600, 1.2, 8, 20, 7)
    e.submit_ptrs(dev.ptrs, 600, 800, dev.stride, True)
    e.close()
print('destroy with a batch in flight: ok')

# 4. stream runner destroyed with batches in flight / never popped
for it in range(5):
    st = api.Stream(600, 1.2, 8, 20, 7, 0, 4, 3)
    st.set_matching((0.0, 800.0, 0.0, 600.0), 100, 0.9, True)
    for b in range(3):
        st.push_ptrs(dev.ptrs, 600, 800, dev.stride, True)
    if it % 2:
        st.pop()
    st.close()
print('stream destroyed with batches in flight: ok')

# 5. two threads, each with its own extractor + matcher, concurrently
res = {}


def worker(tid):
    e = api.Extractor(600, 1.2, 8, 20, 7)
    m = api.Matcher()
    good = True
    for it in range(30):
        k = list(imgs)[(it + tid) % 4]
        g = e(imgs[k])
        good &= same(g, want[k])
        n, m12, _ = m.search_for_initialization(g[0], g[1], g[0], g[1], (0.0, float(k[0]), 0.0, float(k[1])),
                                                np.stack([g[0]['x'], g[0]['y']], 1), 100, 0.9, True)
        good &= n > 100
        # a resident frame per iteration, two searches on it, destroyed while other threads run theirs
        b = (0.0, float(k[0]), 0.0, float(k[1]))
        fr = api.Frame.from_extract(e, 0, b) if it % 2 else api.Frame.from_host(m, g[0], g[1], b)
        sf = e.tables()['sf']
        nk = len(g[0])
        q = np.stack([g[0]['x'], g[0]['y']], 1)
        occ = np.zeros(nk, np.uint8)
        a = m.search_by_projection(fr, None, None, sf, occ, q, g[0]['octave'], np.ones(nk, np.float32), np.full(nk, 9, np.uint8), g[1], 1.0, 0.8)
        bres = m.search_by_projection_uv(fr, None, None, sf, occ, q, g[0]['octave'], g[0]['angle'], np.full(nk, 8, np.uint8),
                                         np.ones(nk, np.uint8), g[1], 7.0, 100, 0, True)
        good &= a[0] > nk * 0.8 and bres[0] > nk * 0.7
        fr.close()
    res[tid] = good


ths = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
[t.start() for t in ths]
[t.join() for t in ths]
print('three threads with own handles:', res)

# 6. error paths
try:
    ex.extract_batch_ptrs([0], 480, 640, 640, True)
    print('NULL frame accepted?!')
except api.OrbfeError as e:
    print('NULL frame ->', e)
try:
    ex(np.zeros((40, 40), np.uint8))
    print('tiny image accepted?!')
except api.OrbfeError as e:
    print('tiny image ->', e)
print('empty image ->', len(ex(np.zeros((0, 0), np.uint8))[0]))
ok &= same(ex(imgs[(640, 480)]), want[(640, 480)])
print('handle still healthy after errors:', ok)
