"""What an integrated build pays per call THROUGH the C++ facade (include/orbfe/orb_shim.hpp over the C ABI), GPU box: builds
tests/cpp/facade_test.cpp, feeds it a 1080p pair (2 000 features), 3 000 MapPoints and a k = 10, L = 6 vocabulary, and prints its
timing line (median of 200 blocking calls each: ORBextractor on a host frame in ordinary memory, SearchByProjection(F, MapPoints),
SearchByBoW(KF, F))."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from os1_amd import api
from os1_amd.synth import shifted, synth, synth_vocabulary

W, H, N, NMP = 1920, 1080, 2000, 3000
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, 'facade_test')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'cpp', 'facade_test.cpp'),
                           '-o', exe, '-L' + os.path.join(ROOT, 'os1_amd'), '-lorbfe', '-Wl,-rpath,' + os.path.join(ROOT, 'os1_amd'),
                           '-Wl,-rpath-link,/opt/rocm/lib'])
    A = synth(31, W, H)
    B = shifted(A, -10, 4, 31)
    A.tofile(os.path.join(d, 'A.gray'))
    B.tofile(os.path.join(d, 'B.gray'))
    ex = api.Extractor(N, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = ex(A), ex(B)
    rng = np.random.default_rng(5)
    src = rng.integers(0, len(k2), NMP)
    rec = np.zeros(NMP, np.dtype([('x', 'f4'), ('y', 'f4'), ('c', 'f4'), ('l', 'i4'), ('f', 'u1'), ('d', 'u1', 32)]))
    xy = (np.stack([k2['x'][src], k2['y'][src]], 1) + rng.uniform(-2, 2, (NMP, 2))).astype(np.float32)
    rec['x'], rec['y'], rec['c'], rec['l'], rec['f'], rec['d'] = xy[:, 0], xy[:, 1], 0.95, k2['octave'][src], 9, d2[src]
    rec.tofile(os.path.join(d, 'mp.bin'))
    open(os.path.join(d, 'meta.txt'), 'w').write('%d %d %d %d\n' % (W, H, N, NMP))
    open(os.path.join(d, 'voc.bin'), 'wb').write(synth_vocabulary(2, 10, 6))   # ORBvoc's shape: levelsup 4 = 100 groups
    rng.choice(np.array([0, 1, 2], np.uint8), len(k1) + len(k2), p=[0.15, 0.8, 0.05]).tofile(os.path.join(d, 'bow.valid'))
    out = subprocess.run([exe, d, 'time'], capture_output=True, text=True, timeout=600)
    print(out.stdout.strip())
    if out.returncode:
        print(out.stderr[-2000:])
        sys.exit(out.returncode)
