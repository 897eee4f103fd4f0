"""-m gpu: the workload bench.py measures (BASELINE.json configs[3]: stream seed 100+g, 1920x1080, 2000 features,
32-frame submissions, 4 batches in flight, SearchForInitialization chained frame to frame and across submissions) against the
CPU oracle -- live, frame by frame, and through the committed digests bench.py itself checks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from os1_amd import stream_workload as wl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIGESTS = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'stream1080_digests.json')))


@pytest.fixture(scope='module')
def api():
    from os1_amd import api as a
    assert a.device_count() >= 1, 'no GPU visible: the product has no CPU fallback'
    return a


def _run_stream(api, seed, nsub, depth=4, source='hbm'):
    sf = wl.StreamFrames(seed)
    idx = [wl.pool_index(p) for p in range(nsub * wl.BATCH)]
    frames = {i: sf.frame(i) for i in sorted(set(idx))}
    order = sorted(frames)
    stack = [frames[i] for i in order]
    dev = api.DeviceFrames(stack, 0) if source == 'hbm' else api.PinnedFrames(stack)
    at = {i: dev.ptrs[k] for k, i in enumerate(order)}
    st = api.Stream(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, 0, wl.BATCH, depth)
    st.set_matching(wl.BOUNDS, wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
    pushed = 0

    def push():
        nonlocal pushed
        ptrs = [at[i] for i in idx[pushed * wl.BATCH:(pushed + 1) * wl.BATCH]]
        st.push_ptrs(ptrs, wl.H, wl.W, wl.W, source == 'hbm')
        pushed += 1
    while pushed < min(depth + 2, nsub):
        push()
    out = []
    for _ in range(nsub):
        out.append(st.pop(copy=True))
        if pushed < nsub:
            push()
    st.close()
    return out


def test_bench_stream_matches_oracle_frame_by_frame(api, oracle):
    """4 submissions of stream 100 at the bench's exact configuration (all 4 handles used, the SearchForInitialization
    chain crosses 3 submission boundaries): every keypoint field, descriptor byte and vnMatches12 entry of the first 64
    frames against the live oracle, and all 128 frames against the committed digests."""
    from oracle.stream_ref import oracle_stream_steps
    nlive = 2
    steps, total, kept = oracle_stream_steps(100, nlive, keep=True)
    assert steps == DIGESTS['streams']['100']['steps'][:nlive], 'committed digests are stale: run tools/gen_stream_digests.py'
    got = _run_stream(api, 100, 4)
    hasher = wl.StepHasher()
    prev_n = 0
    for s, (kps, desc, n, m12, nm) in enumerate(got):
        if s < nlive:
            for i in range(wl.BATCH):
                wk, wd, wnm, wm12 = kept[s * wl.BATCH + i]
                assert n[i] == len(wk), (s, i)
                for f in wk.dtype.names:
                    assert (kps[i, :n[i]][f] == wk[f]).all(), (s, i, f)
                assert desc[i, :n[i]].tobytes() == wd.tobytes(), (s, i)
                assert nm[i] == wnm, (s, i)
                assert (m12[i, :prev_n] == wm12).all(), (s, i)
                prev_n = int(n[i])
        hasher.add(kps, desc, n, m12, nm)
    assert hasher.steps == DIGESTS['streams']['100']['steps']
    assert hasher.nmatches == DIGESTS['streams']['100']['nmatches'] and hasher.nmatches > 20000


@pytest.mark.parametrize('seed', [101, 107])
def test_other_ranks_streams_against_digests(api, seed):
    got = _run_stream(api, seed, 4)
    hasher = wl.StepHasher()
    for res in got:
        hasher.add(*res)
    assert hasher.steps == DIGESTS['streams'][str(seed)]['steps']


def test_host_input_stream_against_digests(api):
    """The PCIe-inclusive leg (page-locked host frames, shared upload lane) returns the same bytes."""
    got = _run_stream(api, 100, 4, source='pinned')
    hasher = wl.StepHasher()
    for res in got:
        hasher.add(*res)
    assert hasher.steps == DIGESTS['streams']['100']['steps']


def _bench(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])


def test_bench_line_is_self_verified(api):
    res = _bench(['--steps', '2', '--warmup', '1', '--cpu-frames', '0'])
    assert res['n_gpus'] == 1 and res['verified'] is True and len(res['outputs_sha256']) == 64
    assert res['verify']['frames_checked'] == 128
    assert res['config']['distinct_frames_per_gpu'] == 256 and res['config']['frames_per_step_per_gpu'] == 256
    assert res['value'] > 1000 and res['pcie_inclusive']['value'] > 1000
    assert res['roofline']['frac'] > 0 and res['roofline']['launch_ms'] > 0


def test_bench_two_ranks_on_one_gpu(api):
    """bench.py --gpus 2 launches its two ranks itself; ORBFE_BENCH_DEVICE=0 puts both on the one GPU of this box.
    Each rank runs its own stream (seeds 100, 101) through the product and checks it against that seed's digests."""
    res = _bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-pcie'], {'ORBFE_BENCH_DEVICE': '0'})
    assert res['n_gpus'] == 2
    assert res['verified'] is True and res['verify']['ranks_verified'] == 2
    assert res['value'] > 1000 and res['config']['parallelism'].startswith('independent streams')


def test_bench_eight_ranks_on_one_gpu(api):
    """The launcher at N = 8: bench.py --gpus 8 spawns eight fresh child ranks before any GPU call (never a re-exec of an
    initialised process), ORBFE_BENCH_DEVICE=0 puts all of them on the one GPU of this box.  Every rank runs its own camera
    stream (seeds 100..107) through the product and checks it against that seed's committed digests
    (tests/golden/stream1080_digests.json); the line reports the host cores all ranks used together."""
    res = _bench(['--gpus', '8', '--steps', '1', '--warmup', '1', '--no-pcie', '--no-latency', '--cpu-frames', '0', '--prewarm-seconds', '0'],
                 {'ORBFE_BENCH_DEVICE': '0'})
    assert res['n_gpus'] == 8
    assert res['verified'] is True and res['verify']['ranks_verified'] == 8
    assert res['value'] > 1000 and res['config']['parallelism'].startswith('independent streams')
    assert 0 < res['host_cpu_cores_used_all_ranks'] < 64


def test_stream_runner_4k_4000_features(api, oracle):
    """BASELINE configs[4] geometry through the streaming path: 3840x2160, 4000 features (the 1 024-node quadtree variant,
    869 level-0 queries per SearchForInitialization), 2-frame submissions on 2 handles, against the live oracle."""
    from oracle.pyoracle import OracleExtractor
    from os1_amd.synth import shifted, synth
    W, H, N, B = 3840, 2160, 4000, 2
    base = synth(5, W, H)
    frames = [base] + [shifted(base, 2 * i, i, 5000 + i) for i in range(1, 3 * B)]
    dev = api.DeviceFrames(frames, 0)
    st = api.Stream(N, 1.2, 8, 20, 7, 0, B, 2)
    bounds = (0.0, float(W), 0.0, float(H))
    st.set_matching(bounds, 100, 0.9, True)
    ox = OracleExtractor(N, 1.2, 8, 20, 7, oracle)
    want = [ox.extract(f) for f in frames]
    for b in range(3):
        st.push_ptrs(dev.ptrs[b * B:(b + 1) * B], H, W, dev.stride, True)
    total = 0
    for b in range(3):
        kps, desc, n, m12, nm = st.pop(copy=True)
        for i in range(B):
            g = b * B + i
            wk, wd = want[g]
            assert n[i] == len(wk)
            assert kps[i, :n[i]].tobytes() == wk.tobytes() and desc[i, :n[i]].tobytes() == wd.tobytes()
            if g == 0:
                continue
            pk, pd = want[g - 1]
            on, om12, _ = oracle.search_for_initialization(pk, pd, wk, wd, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2), 100, 0.9, True)
            assert nm[i] == on and (m12[i, :len(pk)] == om12).all()
            total += on
    assert total > 1000
    st.close()
