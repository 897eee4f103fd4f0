#!/usr/bin/env python3
"""tests/golden/stream1080_digests.json (format 2): per-POSITION sha256 digests of the bench workload (BASELINE.json configs[3]:
stream g = seed 100+g, 1920x1080, 2000 features, extract + SearchForInitialization against the predecessor), computed by the CPU
oracle.  Per stream: the extraction digest of each of the 256 pool frames, the match digest of every frame against its predecessor
walking the pool forwards (fwd) and backwards (bwd), and the match counts -- which covers all 510 positions of the forwards-and-backwards
period and every later position (os1_amd.stream_workload.expected_digests).  bench.py checks the whole period before it times anything
and the batches it pops INSIDE the timed region afterwards; tests/test_gpu_stream_bench.py does the same under pytest.

These are outputs of THIS repository's oracle (parity unpinned), not of the reference.
Run:  python tools/gen_stream_digests.py [nstreams=8]      (about 2 minutes on 8 cores)"""
import json
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(seed):
    from oracle.stream_ref import oracle_stream_table
    return seed, oracle_stream_table(seed)


if __name__ == '__main__':
    from os1_amd import stream_workload as wl
    nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    with Pool(min(nstreams, os.cpu_count() or 1)) as p:
        res = p.map(one, [wl.stream_seed(g) for g in range(nstreams)])
    out = {'format': 2,
           'workload': {'image': [wl.W, wl.H], 'nfeatures': wl.NFEAT, 'nlevels': wl.NLEVELS, 'scale': wl.SCALE,
                        'fast': [wl.INI_TH, wl.MIN_TH], 'batch': wl.BATCH, 'window': wl.WINDOW, 'nnratio': wl.NNRATIO,
                        'check_orientation': wl.CHECK_ORI, 'pool': wl.POOL, 'period': wl.PERIOD},
           'source': 'oracle/orb_oracle.cpp via oracle/stream_ref.py::oracle_stream_table (not reference output)',
           'layout': 'frames[i]: extraction of pool frame i; fwd[i]: SearchForInitialization(pred = i-1, cur = i), fwd[0] = no predecessor; '
                     'bwd[i]: (pred = i+1, cur = i); nm_*: the return values; nkeys[i]: keypoints of frame i',
           'streams': {str(seed): t for seed, t in res}}
    path = os.path.join(ROOT, 'tests', 'golden', 'stream1080_digests.json')
    with open(path, 'w') as f:
        json.dump(out, f, separators=(',', ':'))
        f.write('\n')
    print('wrote', path, {s: sum(t['nm_fwd']) + sum(t['nm_bwd']) for s, t in res})
