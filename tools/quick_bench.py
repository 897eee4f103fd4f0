"""Developer timing helper (GPU box): per-stage ms of the extractor at 1080p/2000."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from os1_amd import api
from os1_amd.synth import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H, N = 1920, 1080, 2000
frames = [synth(100 + i, W, H) for i in range(min(B, 4))]
dev = torch.stack([torch.from_numpy(frames[i % len(frames)]) for i in range(B)]).cuda()
torch.cuda.synchronize()
ex = api.Extractor(N, 1.2, 8, 20, 7)
ex.set_profiling(len(sys.argv) > 2)
ptrs = [dev[i].data_ptr() for i in range(B)]
kps = np.zeros((B, ex.cap), api.KP_DTYPE); desc = np.zeros((B, ex.cap, 32), np.uint8)
for it in range(3):
    ex.extract_batch_ptrs(ptrs, H, W, W, True, kps, desc)
t = time.time(); R = 10
acc = np.zeros(5)
for it in range(R):
    _, _, n = ex.extract_batch_ptrs(ptrs, H, W, W, True, kps, desc)
    acc += ex.stage_ms()
dt = time.time() - t
km, kb, kf = ex.kernel_ms()
if kf and len(sys.argv) > 2:   # per-kernel events are recorded with profiling on and batches of more than two frames
    print('gpu us/frame: pyramid %.1f fast %.1f compact %.1f describe %.1f quadtree %.1f' % tuple(km / kf * 1e3))
print('B=%d  %.2f ms/batch  %.1f fps  host(ms) wait1=%.2f quadtree=%.2f wait2=%.2f assemble=%.2f total=%.2f  n=%s' % (
    B, dt / R * 1e3, B * R / dt, *(acc / R), n[:4]))
