"""N>1 path on CPU: world_size 2 over gloo.  The data path has no exchange step (independent streams,
one per rank); what N>1 adds is stream sharding (seed 100+rank), the barrier and the max-over-ranks
of the elapsed time.  Each rank runs the CPU oracle on its own small stream (the checker standing in
for the GPU worker, which cannot run here) and rank 0 aggregates exactly as bench.py does."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import hashlib, json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    from os1_amd.synth import shifted, synth
    from oracle.pyoracle import Oracle, OracleExtractor
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    seed = 100 + rank                                    # bench.py: stream g -> rank g
    frames = [synth(seed, 320, 240)]
    frames.append(shifted(frames[0], 2, 1, seed * 1000 + 1))
    o = Oracle(); ox = OracleExtractor(300, 1.2, 8, 20, 7, o)
    dist.barrier()
    t0 = time.perf_counter()
    h = hashlib.sha256()
    nk = 0
    for f in frames:
        k, d = ox.extract(f)
        h.update(k.tobytes()); h.update(d.tobytes()); nk += len(k)
    dist.barrier()
    el = time.perf_counter() - t0 + 0.01 * rank          # make the ranks' times differ
    t = torch.tensor([el], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    digests = [None] * world
    dist.all_gather_object(digests, (rank, seed, h.hexdigest(), nk, el))
    if rank == 0:
        print(json.dumps({'max_elapsed': float(t[0]), 'ranks': digests, 'frames': world * len(frames)}))
    dist.destroy_process_group()
''')


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    ranks = sorted(res['ranks'])
    assert [r[0] for r in ranks] == [0, 1] and [r[1] for r in ranks] == [100, 101]
    assert ranks[0][2] != ranks[1][2]                     # different streams -> different outputs
    assert all(r[3] > 100 for r in ranks)
    assert abs(res['max_elapsed'] - max(r[4] for r in ranks)) < 1e-9   # MAX over ranks
    assert res['frames'] == 4


def _json_line(out):
    import json
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])


def test_bench_self_launch_two_ranks_control_plane():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts the two ranks itself (fresh child processes, gloo on
    127.0.0.1): rank g gets stream seed 100+g, the line reports n_gpus = 2 and the MAX over ranks of the elapsed
    time.  --plumbing-only keeps the GPU hot path out so this runs on the CPU box; the product run of the same
    launcher is tests/test_gpu_stream_bench.py::test_bench_two_ranks_on_one_gpu."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--plumbing-only'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = _json_line(out)
    assert res['n_gpus'] == 2 and res['seeds'] == [100, 101] and res['plumbing_only'] is True and res['value'] is None
    assert abs(res['max_elapsed'] - 0.02) < 1e-12          # rank 1's (larger) time wins


def test_bench_eight_ranks_report_every_rank():
    """`--gpus 8`: the line carries one `per_rank` entry per rank (rank, device, NUMA node, its own rate over its own elapsed time, p50,
    verified, host cores, H2D link) gathered with ONE all_gather_object, the slowest rank by name and what waiting for it costs
    (`scaling_efficiency_vs_min_rank` = N x slowest / sum of the ranks' own rates).  Plumbing mode gives rank r the fake duration
    0.01 (r + 1) s, so rank 7 must come out slowest and the efficiency is 8 / (8 * H_8) = 0.3679."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--plumbing-only'], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = _json_line(out)
    assert res['n_gpus'] == 8 and res['seeds'] == list(range(100, 108))
    pr = res['per_rank']
    assert [p['rank'] for p in pr] == list(range(8)) == [p['device'] for p in pr] and [p['stream_seed'] for p in pr] == res['seeds']
    for k in ('numa_node', 'value', 'p50', 'verified', 'host_cores', 'h2d_link_gbs', 'own_elapsed_s'):
        assert all(k in p for p in pr), k
    assert res['slowest_rank']['rank'] == 7 and res['rank_value_min_max'] == [12500.0, 100000.0]
    assert abs(res['scaling_efficiency_vs_min_rank'] - 8 / (8 * sum(1 / (r + 1) for r in range(8)))) < 1e-3
    assert abs(res['aggregate_over_sum_of_rank_rates'] - res['scaling_efficiency_vs_min_rank']) < 1e-3


def test_bench_under_torchrun_two_ranks_control_plane():
    """The driver's launch line (python -m torch.distributed.run ... bench.py --gpus 2) takes the same path."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--plumbing-only']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS='1'))
    assert out.returncode == 0, out.stderr[-2000:]
    res = _json_line(out)
    assert res['n_gpus'] == 2 and res['seeds'] == [100, 101]


def test_bench_refuses_to_run_without_gpu():
    from os1_amd import api
    if api.device_count() > 0:
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert 'no CPU fallback' in (out.stderr + out.stdout)
