import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd import stream_workload as wl
sf = wl.StreamFrames(100); frames = [sf.frame(i) for i in range(64)]
ex = api.Extractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH)
kbuf = np.zeros((1, ex.cap), api.KP_DTYPE); dbuf = np.zeros((1, ex.cap, 32), np.uint8)
pin = api.PinnedFrames(frames)
dev = api.DeviceFrames(frames, 0)
ref = [ex(f) for f in frames[:4]]
for name, ptr_of, ondev, stride in (('resident', lambda i: dev.ptrs[i], True, dev.stride), ('page_locked', lambda i: pin.ptrs[i], False, wl.W), ('pageable', lambda i: frames[i].ctypes.data, False, wl.W)):
    lat = []
    for i in range(74):
        t = time.perf_counter(); _, _, n = ex.extract_batch_ptrs([ptr_of(i % 64)], wl.H, wl.W, stride, ondev, kbuf, dbuf); lat.append(time.perf_counter() - t)
        if i < 4: assert kbuf[0, :n[0]].tobytes() == ref[i][0].tobytes() and dbuf[0, :n[0]].tobytes() == ref[i][1].tobytes(), (name, i)
    lat = np.array(lat[10:]) * 1e3
    print('%-12s median %.4f ms  p90 %.4f' % (name, np.median(lat), np.percentile(lat, 90)))
# where the host's share goes (orbfe_debug_stage_ms of the last call form: enqueue, GPU wait inside collect, -, output assembly, total)
st = []
for i in range(40):
    ex.extract_batch_ptrs([pin.ptrs[i % 64]], wl.H, wl.W, wl.W, False, kbuf, dbuf)
    st.append(ex.stage_ms())
st = np.array(st[5:])
print('page_locked stage ms (median): enqueue %.4f  gpu wait %.4f  assembly %.4f  total in library %.4f' % tuple(np.median(st, 0)[[0, 1, 3, 4]]))
