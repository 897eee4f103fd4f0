"""Resident rate of the stream runner in a process that has NOT loaded torch (what a C++ integration looks like to the HIP runtime:
no foreign streams): frames/s over a few seconds, the batches in flight the runner settled on, the hardware queues it was given."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from os1_amd import api
from os1_amd import stream_workload as wl
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl.SUBMIT
frames = wl.StreamFrames(100).frames()
dev = api.DeviceFrames(frames, 0)
st = api.Stream(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, 0, B, depth)
st.set_matching(wl.BOUNDS, wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
st.set_queue_slots(depth + 20)
pos = 0
def push():
    global pos
    st.push_ptrs([dev.ptrs[wl.pool_index(pos + i)] for i in range(B)], wl.H, wl.W, dev.stride, True)
    pos += B
for _ in range(depth + 12):
    push()
def run(seconds):
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        st.pop()
        push()
        n += B
    return n / (time.perf_counter() - t0)
run(1.5)
rates = [run(2.0) for _ in range(3)]
print('queues %s depth %d batch %d: in flight %d, %.0f frames/s (runs %s)' % (os.environ.get('GPU_MAX_HW_QUEUES'), depth, B, st.batches_in_flight(),
      float(np.median(rates)), ' '.join('%.0f' % r for r in rates)))
