"""Timing probe for the bag-of-words calls (single-frame C-ABI calls, host descriptors in, host vectors out)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from os1_amd import api
from os1_amd.synth import shifted, synth, synth_vocabulary
from oracle.pyoracle import Oracle

image = synth_vocabulary(1, 10, 6)
v = api.Vocabulary(image)
o = Oracle()
ov = o.vocabulary(image)
ex = api.Extractor(2000, 1.2, 8, 20, 7)
A = synth(3, 1920, 1080)
B = shifted(A, -24, 3, 33)
(k1, d1), (k2, d2) = ex(A), ex(B)
m = api.Matcher()


def t(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


print('ComputeBoW  gpu %.3f ms   oracle(1 core) %.3f ms' % (t(lambda: v.transform(d1, 4), 200), t(lambda: ov.transform(d1, 4), 20)))
t1, t2 = v.transform(d1, 4), v.transform(d2, 4)
v1 = np.ones(len(k1), np.uint8)
print('SearchByBoW gpu %.3f ms   oracle(1 core) %.3f ms' % (
    t(lambda: m.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True), 200),
    t(lambda: o.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True), 20)))
print('nmatches', m.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True)[0], 'groups', len(t1[2][0]))
