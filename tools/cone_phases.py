"""Phase timeline of k_pyramid_cone (one block in the middle of the frame) for ONE blocking 1080p extraction: experiments build
(tools/build_exp.sh, ORBFE_LIB=os1_amd/liborbfe_exp.so).  Stamps are s_memrealtime (100 MHz)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd import stream_workload as wl
sf = wl.StreamFrames(100)
frames = [sf.frame(i) for i in range(8)]
ex = api.Extractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH)
dev = api.DeviceFrames(frames, 0)
L = api.load_library()
for rep in range(4):
    ex.extract_batch_ptrs([dev.ptrs[rep]], wl.H, wl.W, dev.stride, True)
    out = np.zeros(32, np.uint64)
    L.orbfe_exp_cone_stamps(out.ctypes.data_as(C.c_void_p))
o = out.astype(np.int64)
print('prologue (ranges, coefficient + region loads, coefficient slices) %.2f us' % ((o[1] - o[0]) / 100.0))
prev = o[1]
for l in range(1, wl.NLEVELS):
    print('level %d: row pass %.2f  column pass %.2f us' % (l, (o[16 + l] - prev) / 100.0, (o[1 + l] - o[16 + l]) / 100.0))
    prev = o[1 + l]
print('TOTAL in block %.2f us' % ((o[wl.NLEVELS] - o[0]) / 100.0))
