"""Phase timeline of k_quadtree3 for ONE blocking 1080p extraction (experiments build: make -C os1_amd/csrc EXPERIMENTS=1 OUT=../liborbfe_exp.so,
ORBFE_LIB=os1_amd/liborbfe_exp.so).  Stamps are s_memrealtime (100 MHz) of thread 0 of every level's block."""
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
names = {0: 'entry', 1: 'setup', 2: 'jump sweep', 3: 'jump fold+vote', 4: 'jump nodes+record', 5: 'it0 order/sort', 6: 'it0 ninfo+zero', 7: 'it0 sweep',
         8: 'it0 fold', 9: 'it0 node pass', 10: 'it1 order/sort', 11: 'it1 ninfo+zero', 12: 'it1 sweep', 13: 'it1 fold', 14: 'it1 node pass',
         15: 'loop exit', 16: 'pick sweep', 17: 'outputs'}
for rep in range(4):
    out = np.zeros((12, 32), np.uint64)
    L.orbfe_exp_qt_stamps(out.ctypes.data_as(C.c_void_p), 1)
    k, d, n = ex.extract_batch_ptrs([dev.ptrs[rep]], wl.H, wl.W, dev.stride, True)
    L.orbfe_exp_qt_stamps(out.ctypes.data_as(C.c_void_p), 0)
for l in range(wl.NLEVELS):
    row = out[l].astype(np.int64)
    ncand = len(ex.candidates(l))
    t0 = row[0]
    line = 'level %d (%5d candidates): ' % (l, ncand)
    prev = t0
    for i in range(1, 18):
        if row[i]:
            line += '%s %.2f | ' % (names[i], (row[i] - prev) / 100.0)
            prev = row[i]
    print(line + 'TOTAL %.2f us' % ((row[17] - t0) / 100.0))
