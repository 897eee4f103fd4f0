"""GPU box: where a FAST cell-wave's LIFETIME goes (not its instruction count): ORBFE_FAST_ABLATE=4 builds k_fast_tasks with
s_memtime stamps between its phases; per-wave averages over blocking 32-frame 1080p batches, the kernel alone."""
import ctypes as C
import os
import sys
os.environ['ORBFE_FAST_ABLATE'] = '4'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from os1_amd import api
from os1_amd.synth import synth
B, W, H, N = 32, 1920, 1080, 2000
frames = [synth(100 + i, W, H) for i in range(4)]
dev = torch.stack([torch.from_numpy(frames[i % 4]) for i in range(B)]).cuda()
torch.cuda.synchronize()
ex = api.Extractor(N, 1.2, 8, 20, 7)
ptrs = [dev[i].data_ptr() for i in range(B)]
kps = np.zeros((B, ex.cap), api.KP_DTYPE); desc = np.zeros((B, ex.cap, 32), np.uint8)
out = (C.c_ulonglong * 8)()
ex.L.orbfe_debug_fast_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for it in range(3):
    ex.extract_batch_ptrs(ptrs, H, W, W, True, kps, desc)
ex.L.orbfe_debug_fast_stamps(ex.h, out, 0)     # the records of the last launch
n = out[5]
names = ['entry -> task + level geometry known (scalar loads)', 'ROI issue -> landed in LDS (+ score-tile zeroing)', 'stage 1: pre-test + compaction',
         'stage 2: arc score', 'stage 3 + second pass + exit']
tot = sum(out[k] for k in range(5))
print('k_fast_tasks phase latency per cell-wave, shader cycles (s_memtime), %d waves:' % n)
for k in range(5):
    print('  %-58s %8.0f  %5.1f %%' % (names[k], out[k] / n, 100.0 * out[k] / tot))
print('  %-58s %8.0f  (= %.2f us at 2.4 GHz)' % ('stamped lifetime', tot / n, tot / n / 2400.0))

