#!/usr/bin/env python3
"""tests/golden/stream1080_digests.json: per-step sha256 digests of the bench workload (BASELINE.json configs[3]:
stream g = seed 100+g, 1920x1080, 2000 features, 32-frame steps, extract + SearchForInitialization against the
predecessor), computed by the CPU oracle.  bench.py hashes what the GPU path returns for the same steps and prints
"verified": true/false; tests/test_gpu_stream_bench.py does the same under pytest.

These are outputs of THIS repository's oracle (parity unpinned), not of the reference.
Run:  python tools/gen_stream_digests.py [nsteps=4] [nstreams=8]      (about 2 minutes on 8 cores)"""
import json
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(args):
    seed, nsteps = args
    from oracle.stream_ref import oracle_stream_steps
    steps, total = oracle_stream_steps(seed, nsteps)
    return seed, steps, total


if __name__ == '__main__':
    from os1_amd import stream_workload as wl
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    with Pool(min(nstreams, os.cpu_count() or 1)) as p:
        res = p.map(one, [(wl.stream_seed(g), nsteps) for g in range(nstreams)])
    out = {'workload': {'image': [wl.W, wl.H], 'nfeatures': wl.NFEAT, 'nlevels': wl.NLEVELS, 'scale': wl.SCALE,
                        'fast': [wl.INI_TH, wl.MIN_TH], 'batch': wl.BATCH, 'window': wl.WINDOW, 'nnratio': wl.NNRATIO,
                        'check_orientation': wl.CHECK_ORI, 'pool': wl.POOL},
           'source': 'oracle/orb_oracle.cpp via oracle/stream_ref.py (not reference output)',
           'streams': {str(seed): {'steps': steps, 'nmatches': total} for seed, steps, total in res}}
    path = os.path.join(ROOT, 'tests', 'golden', 'stream1080_digests.json')
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path, {s: t for s, _, t in res})
