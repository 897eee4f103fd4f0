"""Blocking single-frame extraction from a page-locked host frame, 40 calls: for rocprofv3 --kernel-trace --memory-copy-trace
(tools/trace_calls.py prints the timeline of the last call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd.synth import synth, shifted
W, H, N = 1920, 1080, 2000
base = synth(100, W, H)
frames = [base] + [shifted(base, 2 * i, i, 1000 + i) for i in range(1, 8)]
pin = api.PinnedFrames(frames)
ex = api.Extractor(N, 1.2, 8, 20, 7)
kb = np.zeros((1, ex.cap), api.KP_DTYPE); db = np.zeros((1, ex.cap, 32), np.uint8)
lat = []
for i in range(40):
    t0 = time.perf_counter()
    ex.extract_batch_ptrs([pin.ptrs[i % 8]], H, W, W, False, kb, db)
    lat.append(time.perf_counter() - t0)
    time.sleep(0.002)
print('median %.4f ms' % (np.median(lat[5:]) * 1e3))
