#!/usr/bin/env python3
"""One worker PROCESS of bench.py's cpu_baseline_all_cores leg: `threads` oracle extractors, each over `per` consecutive
frames of the stream (extract + SearchForInitialization against the predecessor), all starting at wall-clock `t0`.
Separate processes because 256 threads inside one process serialise on its address-space lock (every extract call
allocates pyramid buffers): 256 threads x 1 process ran at 38 frames/s where 64 threads reached 65.
usage: cpu_oracle_worker.py <frames.npy> <first_frame> <threads> <per> <t0> <do_match>
       -> prints "<t_start> <t_end> <frames>" (t_start = when this worker's threads really started: t0, or later if its
          interpreter was not ready by then) """
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle.pyoracle import Oracle, OracleExtractor  # noqa: E402
from os1_amd import stream_workload as wl  # noqa: E402

path, first, threads, per, t0, do_match = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
frames = np.load(path, mmap_mode='r')
o = Oracle()
exs = [OracleExtractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, o) for _ in range(threads)]


def work(i):
    prev = None
    for k in range(per):
        img = np.ascontiguousarray(frames[(first + i * per + k) % len(frames)])
        kp, d = exs[i].extract(img)
        if do_match and prev is not None:
            o.search_for_initialization(prev[0], prev[1], kp, d, wl.BOUNDS, np.stack([prev[0]['x'], prev[0]['y']], 1),
                                        wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
        prev = (kp, d)


ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
while time.time() < t0:
    time.sleep(0.002)
t_start = time.time()
for t in ths:
    t.start()
for t in ths:
    t.join()
print('%.6f %.6f %d' % (t_start, time.time(), threads * per), flush=True)
