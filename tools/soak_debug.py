"""Soak run (GPU box): many batches through the stream runner, resident-set size before / after."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from os1_amd import api
from os1_amd.synth import synth, shifted


def rss_mb():
    for l in open('/proc/self/status'):
        if l.startswith('VmRSS'):
            return int(l.split()[1]) / 1024.0


W, H, B = 1920, 1080, 32
base = synth(100, W, H)
frames = [base] + [shifted(base, 2 * i, i, 1000 + i) for i in range(1, B)]
dev = api.DeviceFrames(frames, 0)
st = api.Stream(2000, 1.2, 8, 20, 7, 0, B, 3)
st.set_matching((0.0, float(W), 0.0, float(H)), 100, 0.9, True)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for _ in range(5):
    st.push_ptrs(dev.ptrs, H, W, dev.stride, True)
r0 = None
t0 = time.time()
tot = 0
for i in range(nsteps):
    kps, desc, n, m12, nm = st.pop()
    tot += int(n.sum())
    st.push_ptrs(dev.ptrs, H, W, dev.stride, True)
    if i == 200:
        r0 = rss_mb()
dt = time.time() - t0
print('%d batches in %.1f s = %.0f frames/s; keypoints/frame %.1f; RSS %.0f -> %.0f MB' % (nsteps, dt, nsteps * B / dt, tot / nsteps / B, r0, rss_mb()))
