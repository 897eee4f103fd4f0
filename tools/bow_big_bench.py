"""GPU box: SearchByBoW with every feature under ONE vocabulary node (2000 x 2000), timing + optional kernel trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
m = api.Matcher()
rng = np.random.default_rng(77)
n = 2000
one = (np.array([3], np.uint32), np.array([0, n], np.uint32), np.arange(n, dtype=np.uint32))
d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
d2 = d1[rng.permutation(n)].copy()
for i in range(n):
    for b in rng.integers(0, 256, rng.integers(0, 31)):
        d2[i, b >> 3] ^= np.uint8(1 << (b & 7))
a = np.zeros(n, np.float32)
v1 = np.ones(n, np.uint8)
t = []
for _ in range(30):
    t0 = time.perf_counter()
    nm, _ = m.search_by_bow(d1, a, v1, one, d2, a, None, one, 0.7, False, False)
    t.append(time.perf_counter() - t0)
print('matches %d  median %.3f ms' % (nm, float(np.median(t[5:])) * 1e3))
