"""GPU box: how the GPU-resident SearchForInitialization's bookkeeping kernel (k_sfi_resolve) spends its time, pair by pair: rounds of the
fixed point, candidate entries, whether the serial finish ran, shader cycles of the block -- over the whole forwards-and-backwards period of
bench stream 100 (512 pairs), 64-frame submissions, one batch in flight (blocks uncontended) unless argv[1] gives another depth.
ORBFE_SFI_DEBUG=1 makes the kernel write one record per pair (orbfe_debug_sfi_records)."""
import ctypes as C
import os
import sys
os.environ['ORBFE_SFI_DEBUG'] = '1'
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd import stream_workload as wl

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = wl.SUBMIT
sf = wl.StreamFrames(100)
frames = sf.frames()
dev = api.DeviceFrames(frames, 0)
st = api.Stream(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, 0, B, depth)
st.set_matching(wl.BOUNDS, wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
ex = api.Extractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH)
ex.L.orbfe_debug_sfi_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
nsub = -(-wl.PERIOD // B) * 2          # two periods: the first submission has no predecessor for its first frame
pos = 0
pushed = 0
while pushed < min(depth + 2, nsub):
    st.push_ptrs([dev.ptrs[wl.pool_index(pos + i)] for i in range(B)], wl.H, wl.W, dev.stride, True); pos += B; pushed += 1
for k in range(nsub):
    st.pop()
    if pushed < nsub:
        st.push_ptrs([dev.ptrs[wl.pool_index(pos + i)] for i in range(B)], wl.H, wl.W, dev.stride, True); pos += B; pushed += 1
rec = np.zeros((1 << 16, 8), np.int32)
n = C.c_int(0)
api._check(ex.L.orbfe_debug_sfi_records(ex.h, rec.ctypes.data_as(C.c_void_p), len(rec), C.byref(n), 1))
rec = rec[:n.value]
rounds, total, serial, cyc = rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4].astype(np.int64) & 0xffffffff
print('k_sfi_resolve, %d pairs of stream 100 (depth %d, %d-frame submissions); cycles = s_memtime of the block (100 MHz reference x 24 on this part if the numbers look like microseconds x 100)' % (len(rec), depth, B))
print('rounds of the fixed point: min %d  median %d  mean %.2f  p90 %d  max %d;  serial finish in %d pairs' % (rounds.min(), np.median(rounds), rounds.mean(), np.percentile(rounds, 90), rounds.max(), int(serial.sum())))
print('histogram of rounds:', {int(k): int(v) for k, v in zip(*np.unique(rounds, return_counts=True))})
print('candidate entries per pair: mean %.0f  max %d;  level-0 queries n1: mean %.0f  max %d;  pool in LDS for %d pairs' % (total.mean(), total.max(), rec[:, 5].mean(), rec[:, 5].max(), int(rec[:, 7].sum())))
print('block cycles: mean %.0f  median %.0f  p90 %.0f  max %.0f   (max / mean = %.2f)' % (cyc.mean(), np.median(cyc), np.percentile(cyc, 90), cyc.max(), cyc.max() / cyc.mean()))
for r in np.unique(rounds):
    print('  rounds %2d: %4d pairs, mean cycles %.0f' % (r, (rounds == r).sum(), cyc[rounds == r].mean()))
order = np.argsort(-cyc)[:6]
print('slowest pairs: (frame in batch, rounds, candidates, serial, cycles)', [tuple(int(x) for x in rec[i, :5]) for i in order])
