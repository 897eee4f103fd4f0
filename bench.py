#!/usr/bin/env python3
"""bench.py -- frames/sec of the ORB front end (extract + match) on N MI355X GPUs of one node.

Metric (BASELINE.json): frames/sec extract+match, 1920x1080 @ 2000 ORB features, 8 levels, 1.2.
Workload = BASELINE.json configs[3] per GPU (SURVEY.md s8(d) "Config 4", os1_amd/stream_workload.py): camera stream
g (seed 100+g) -> GPU g, a pool of 256 DISTINCT 1080p frames (530 MB, larger than the 256 MB Infinity Cache), each
frame = its predecessor shifted by (2,1) px; the stream walks the pool forwards and backwards.
A "step" is one pass of the hot path over one batch of 2 048 frames of the stream (--passes 8 walks over the 256-frame pool, 32
submissions of 64 frames to the native stream runner; a 256-frame step had become a 3 ms measurement): ORBextractor::operator()
on every frame and ORBmatcher::SearchForInitialization of every frame against its predecessor (window 100, nnratio 0.9,
checkOrientation).  Keypoints / descriptors / match indices
come back to host memory inside the timed region (they are the path's outputs).

`value`: frames resident in HBM when the timed region starts (the task's bench contract).  The same run also times
the stream with the frames starting in page-locked HOST memory (`pcie_inclusive`: H2D of every frame inside the timed
region, the figure SURVEY.md s8(d) defines) and prints it next to the measured link rate.

Self-check against the oracle's per-position digests (tests/golden/stream1080_digests.json), twice: (1) before anything is
timed the WHOLE forwards-and-backwards period of the stream (510 positions: every pool frame, the turn-around, every backward
pair) goes through the same runner; (2) INSIDE the timed region the batch popped last in every step is copied, and after the
clock has stopped every frame of those copies is checked at its stream position (`verify.timed_verified`,
`verify.timed_frames_checked`).  "verified": true means every keypoint, descriptor and vnMatches12 entry of both sets is
bit-identical to the CPU oracle's.

Multi-GPU: independent streams, one process + one stream runner per GPU, no data-path collective; the only cross-rank
traffic is the barrier and the max-over-ranks of the elapsed time (gloo, CPU tensors).  Started either by the driver
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) or, when WORLD_SIZE is unset, by
bench.py itself: `python bench.py --gpus N` spawns the N ranks as fresh child processes before it touches any GPU.

One JSON line on rank 0 (see the task's bench contract) with extra objects:
  roofline       -- dominant kernel (k_fast_tasks): algorithmic bytes per launch / HIP-event duration vs HBM peak,
                    plus its vector-instruction-issue roof (the bound that actually explains its time)
  cpu_baseline   -- the CPU oracle (scalar port, 1 core) on a bounded sample of the same frames
  pcie_inclusive -- the same stream from page-locked host frames
"""
import argparse
import collections
import json
import os

# Four extraction batches in flight need more than the HIP runtime's default of 4 hardware queues (two streams folded onto one
# queue serialise; every rank also has its upload lane): must be in the environment BEFORE the process's first HIP call -- i.e.
# before torch / liborbfe are loaded.  A value the caller exported wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz = 6.1e11 /s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='frames per submission to the stream runner (default: stream_workload.SUBMIT = 64; the digests are per 32 frames whatever this is)')
    ap.add_argument('--pool', type=int, default=256, help='distinct frames per stream')
    ap.add_argument('--passes', type=int, default=None, help='walks over the pool in one step (default stream_workload.PASSES = 8: a step = 2 048 frames)')
    ap.add_argument('--cpu-frames', type=int, default=64, help='frames of the CPU-oracle baseline sample (0 = skip)')
    ap.add_argument('--bow', action='store_true', help='also run Frame::ComputeBoW (k=10, L=6 synthetic vocabulary) behind the descriptor kernel (not the headline value)')
    ap.add_argument('--no-match', action='store_true', help='extract only (configs[1])')
    ap.add_argument('--no-pcie', action='store_true', help='skip the PCIe-inclusive leg')
    ap.add_argument('--no-verify', action='store_true', help='skip the digest check of the first steps')
    ap.add_argument('--no-latency', action='store_true', help='skip the blocking single-frame latency leg (profiling runs: keeps every launch a full batch)')
    ap.add_argument('--host-input', nargs='?', const='pageable', default=None, choices=['pageable', 'pinned'],
                    help='headline leg from HOST frames instead (developer aid; never the contract value)')
    ap.add_argument('--depth', type=int, default=4, help='extraction batches in flight inside the stream runner (4 with 8 hardware queues: '
                    '66.7 k frames/s against 63.6 k at 3; 5 and 6 are slower -- DESIGN.md s5)')
    ap.add_argument('--prewarm-seconds', type=float, default=1.5,
                    help='untimed stream work in front of the W warm-up steps (a box that has just been started runs its first second slower: clocks, first-touch pages)')
    ap.add_argument('--single-stream', action='store_true',
                    help='SURVEY.md s8(e), the single-stream shape: ONE camera stream (seed 100) dealt batch by batch to --gpus N devices by '
                         'one process (orbfe_stream_multi_*: in-order completion queue, host bounce of the batch-boundary predecessor); '
                         'same JSON contract, scaling "strong" (the stream is the same whatever N is)')
    ap.add_argument('--plumbing-only', action='store_true',
                    help='CPU test aid for the N>1 control plane: rendezvous, barrier, max-over-ranks and the JSON line with NO '
                         'hot-path work and no value (tests/test_multiproc.py); never a measurement')
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  Nothing in this
    (parent) process has touched a GPU -- no HIP call, no torch.cuda call -- and the children are new interpreters,
    not exec()s of an initialised one."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    if 'ORBFE_BENCH_DEVICE' not in os.environ and not args.plumbing_only:
        import torch                                 # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < args.gpus:
            print('bench.py --gpus %d: only %d GPU(s) visible (the product has no CPU fallback)' % (args.gpus, have), file=sys.stderr)
            return 2
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies would leave the others waiting in a gloo collective: watch all of them, stop the rest on the first failure
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                for q in live:
                    q.kill()                         # exact children of this process
    out0 = procs[0].stdout.read() if procs[0].stdout else ''
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main():
    args = parse_args()
    if args.single_stream:
        run_single_stream(args)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


def run_single_stream(args):
    """`--single-stream --gpus N`: one process, N devices, ONE stream (seed 100).  Batch k of the stream is extracted on device k mod N
    (frames resident in that device's HBM when the timed region starts: the pool is replicated), every frame is matched against its
    predecessor -- inside a batch on its GPU, across a batch boundary after a host bounce -- and the batches come back in stream order.
    A step is the same 2 048 frames as in the per-GPU-stream mode, so `value` is directly comparable; `scaling` is "strong"."""
    import numpy as np
    if 'WORLD_SIZE' in os.environ and int(os.environ.get('RANK', '0')) != 0:
        return                                       # under a launcher: rank 0 drives every device, the others have nothing to do
    os.environ.setdefault('ORBFE_POLL_WAIT_US', '50')
    from os1_amd import api
    from os1_amd import stream_workload as wl
    n = args.gpus
    if 'ORBFE_BENCH_DEVICE' in os.environ:           # testing aid: N runners on one GPU
        devices = [int(os.environ['ORBFE_BENCH_DEVICE'])] * n
    else:
        devices = list(range(n))
        if api.device_count() < n:
            raise SystemExit('bench.py --single-stream --gpus %d: only %d GPU(s) visible (the product has no CPU fallback)' % (n, api.device_count()))
    W, H, B = wl.W, wl.H, (args.batch or wl.SUBMIT)
    assert args.pool % B == 0
    passes = args.passes or wl.PASSES
    step_frames = passes * args.pool
    subs = step_frames // B
    seed = wl.stream_seed(0)
    frames = wl.StreamFrames(seed, W, H, args.pool).frames()
    pools = {d: api.DeviceFrames(frames, d) for d in sorted(set(devices))}
    depth = max(1, min(args.depth, 4))
    st = api.MultiStream(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, devices, B, depth)
    st.set_matching(wl.BOUNDS, 0 if args.no_match else wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
    pos = [0]
    pending = collections.deque()
    first = [True]
    ahead = n * depth + 2

    def push():
        idx = [wl.pool_index(pos[0] + i, args.pool) for i in range(B)]
        pending.append(pos[0])
        pos[0] += B
        pool = pools[st.device_of_next_push()]
        st.push_ptrs([pool.ptrs[i] for i in idx], H, W, pool.stride, True)

    nmatch = [0]
    last_n = [0]
    pop_times = []

    def run(nb, on_pop=None):
        pushed = 0
        while pushed < min(ahead, nb):
            push()
            pushed += 1
        for _ in range(nb):
            res = st.pop()
            pop_times.append(time.perf_counter())
            p0 = pending.popleft()
            if on_pop:
                on_pop(p0, res, first[0])
            first[0] = False
            last_n[0] = int(res[2][-1])     # vnMatches12 of the next batch's first frame has this many entries
            nmatch[0] += int(res[4].sum())
            if pushed < nb:
                push()
                pushed += 1

    def sync():
        for d in sorted(set(devices)):
            api.device_synchronize(d)

    table = None if (args.no_verify or args.no_match or args.pool != wl.POOL) else load_digest_table(seed)
    verify = {'verified': None}
    if table:
        verify = verify_period(wl, table, seed, B, lambda nb, cb: run(nb, cb))
    t_end = time.perf_counter() + args.prewarm_seconds
    while time.perf_counter() < t_end:
        run(subs)
    run(args.warmup * subs)
    samples = []
    every = -(-args.steps // 32)
    popped = [0]

    def sample(p0, res, fst):
        popped[0] += 1
        k, last = divmod(popped[0], subs)
        if last == 0 and (k - 1) % every == 0 and table is not None:
            samples.append((p0, tuple(a.copy() for a in res), last_n[0]))
    del pop_times[:]
    nmatch[0] = 0
    sync()
    t0 = time.perf_counter()
    run(args.steps * subs, sample)
    sync()
    el = time.perf_counter() - t0
    bad = []
    checked = 0
    if table is not None:
        # a batch in the middle of the stream: its first frame's predecessor is the previous batch's last frame (the host bounce)
        for p0, res, prev_n in samples:
            chk = wl.PositionChecker(table)
            chk.check(p0, *res, prev_n=prev_n, first_of_runner=False)
            bad += chk.bad
            checked += chk.frames
        verify['timed_verified'] = not bad
        verify['timed_frames_checked'] = checked
        verify['verified'] = bool(verify.get('verified')) and not bad
    fps = args.steps * step_frames / el
    step_ms = np.diff(np.array(pop_times)[subs - 1::subs]) * 1e3 if len(pop_times) >= 3 * subs else None
    out = {'metric': 'frames/sec extract+match, 1920x1080 @ 2000 ORB feats', 'value': round(fps, 2),
           'value_p50': round(step_frames / (float(np.percentile(step_ms, 50)) * 1e-3), 2) if step_ms is not None and len(step_ms) > 1 else None,
           'unit': 'frames/s', 'n_gpus': n, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(el / args.steps * 1e3, 4),
           'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
           'config': {'workload': '1080p_2000feat_8lv_1.2_extract%s_ONE_stream_over_%d_devices' % ('' if args.no_match else '+SearchForInitialization', n),
                      'frames_per_step': step_frames, 'frames_per_submission': B, 'distinct_frames': args.pool, 'devices': devices,
                      'batches_in_flight_per_device': depth, 'parallelism': 'one stream, batch k -> device k mod N, in-order completion queue, '
                      'host bounce of the batch-boundary predecessor (no collective)',
                      'input': 'frames resident in HBM of the device their batch goes to (pool replicated per device)',
                      'matches_per_frame': round(nmatch[0] / max(args.steps * step_frames, 1), 1), 'hip_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')},
           'verified': verify.get('verified'), 'verify': verify,
           'roofline': None, 'cpu_baseline': None,
           'note': 'single-stream mode (SURVEY.md s8(e)); the kernels are the per-GPU-stream mode\'s: its line carries roofline and cpu_baseline'}
    st.close()
    print(json.dumps(out), flush=True)


def run_rank(args):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'ORBFE_BENCH_DEVICE' in os.environ:          # testing aid: run several ranks on one GPU
        local_rank = int(os.environ['ORBFE_BENCH_DEVICE'])
    world = int(os.environ.get('WORLD_SIZE', '1'))

    # torch first (it carries its own HIP runtime; loading it after liborbfe has initialised HIP
    # leaves torch without devices), then the product library.
    import numpy as np
    import torch
    have_torch_gpu = torch.cuda.is_available()
    if have_torch_gpu:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # control plane only (barrier + max of a scalar): gloo; the data path has no exchange step
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    from os1_amd import stream_workload as wl
    if args.plumbing_only:
        # control plane only: what N>1 adds to the data path is stream sharding (seed 100+rank), the barrier and the
        # max-over-ranks of the elapsed time -- exercised here without a GPU, with a rank-dependent fake duration
        if dist is not None:
            dist.barrier()
        own = 0.01 * (rank + 1)
        el = own
        seeds = [wl.stream_seed(rank)]
        mine = rank_record(rank, local_rank, None, 1000.0 / own, 1000.0 / own, None, 0.0, own, None, wl.stream_seed(rank))
        per_rank = [mine]
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
            seeds = [None] * world
            dist.all_gather_object(seeds, wl.stream_seed(rank))
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        if rank == 0:
            print(json.dumps({'plumbing_only': True, 'value': None, 'n_gpus': world, 'max_elapsed': el, 'seeds': seeds,
                              'steps': args.steps, 'warmup': args.warmup, **per_rank_summary(per_rank, world * 1000.0 / el)}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    from os1_amd import api
    if api.device_count() <= local_rank:
        raise SystemExit('bench.py needs GPU %d (found %d): the product has no CPU fallback' % (local_rank, api.device_count()))
    # this rank's threads (stream-runner workers are created below) and page-locked buffers next to its GPU
    numa_node = api.device_numa_node(local_rank)
    numa_cpus = api.bind_thread_to_device(local_rank)

    # How the runner's worker waits for the GPU (orbfe_extractor_set_wait_mode): sleeping 50 us between polls saves 0.4 host core per
    # rank, spinning is worth +1 % and steadier steps (DESIGN_NOTES E.5).  Spin where the rank has cores to spare -- at least four of its
    # own, counting the cgroup's quota and the ranks sharing the node; ORBFE_POLL_WAIT_US in the environment wins.
    def cores_for_this_rank():
        n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        total = os.cpu_count() or n
        try:
            q, period = open('/sys/fs/cgroup/cpu.max').read().split()
            if q != 'max':
                total = min(total, max(1, int(float(q) / float(period) + 0.5)))
        except Exception:
            pass
        return min(n, total / max(int(os.environ.get('LOCAL_WORLD_SIZE', world)), 1))
    wait_mode = os.environ.get('ORBFE_POLL_WAIT_US')
    if wait_mode is None:
        wait_mode = '0' if cores_for_this_rank() >= 4 else '50'
        os.environ['ORBFE_POLL_WAIT_US'] = wait_mode

    W, H, B = wl.W, wl.H, (args.batch or wl.SUBMIT)
    assert args.pool % B == 0, '--pool must be a multiple of --batch'
    passes = args.passes or wl.PASSES
    step_frames = passes * args.pool            # frames of one step (per GPU)
    subs = step_frames // B                     # submissions per step
    cur = {'B': B, 'subs': subs, 'first': True}  # (the PCIe-inclusive leg runs its own runner with 32-frame submissions)
    seed = wl.stream_seed(rank)                 # config 4: stream g -> GPU g, seeds 100+g
    sf = wl.StreamFrames(seed, W, H, args.pool)
    frames = sf.frames()
    dev = api.DeviceFrames(frames, local_rank)
    want_pinned = (not args.no_pcie) or args.host_input == 'pinned'
    pinned = api.PinnedFrames(frames) if want_pinned else None

    # native stream runner: `depth` extractor handles, GPU-resident matching, C++ worker thread
    def make_runner(depth, batch=None):
        r = api.Stream(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, local_rank, batch or B, max(1, depth))
        r.set_matching(wl.BOUNDS, 0 if args.no_match else wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
        if args.bow:
            r.set_vocabulary(voc, 4)
        return r

    voc = None
    if args.bow:
        from os1_amd.synth import synth_vocabulary
        voc = api.Vocabulary(synth_vocabulary(1, 10, 6), local_rank)
    st = make_runner(args.depth)
    head_in_flight = st.batches_in_flight()   # what the runner settled on (fewer than --depth when the process has too few hardware queues)

    pos = [0]                                   # stream position of the next pushed frame
    pending = collections.deque()               # first stream position of every batch pushed and not yet popped

    def push(source):
        idx = [wl.pool_index(pos[0] + i, args.pool) for i in range(cur['B'])]
        pending.append(pos[0])
        pos[0] += cur['B']
        if source == 'hbm':
            st.push_ptrs([dev.ptrs[i] for i in idx], H, W, dev.stride, True)
        elif source == 'pinned':
            st.push_ptrs([pinned.ptrs[i] for i in idx], H, W, W, False)
        else:
            st.push_ptrs([frames[i].ctypes.data for i in idx], H, W, W, False)

    nmatch_total = [0]
    pop_times = []

    # batches pushed ahead of the pops: `depth` are on the GPU, the rest wait in the runner's queue, so that a hiccup of this
    # (Python) thread does not drain the GPU -- such hiccups of 4-12 ms happen about once a second on some test boxes
    lookahead = int(os.environ.get('ORBFE_BENCH_LOOKAHEAD', args.depth + 27))
    st.set_queue_slots(lookahead + 2)          # the library's default queue is short (depth + 4 slots)
    lookahead = min(lookahead, st.queue_slots() - 2)   # never more ahead than the runner can hold (push would block forever)
    want_lookahead = lookahead

    def run(nbatches, source, on_pop=None):
        """nbatches submissions through the runner: push (async extraction + SearchForInitialization of every frame
        against its predecessor on the GPU), pop keypoints / descriptors / matches in host memory.  Up to `lookahead`
        batches are in the pipeline; every push and pop of the nbatches batches is inside this call."""
        pushed = 0
        while pushed < min(lookahead, nbatches):
            push(source)
            pushed += 1
        for _ in range(nbatches):
            res = st.pop()
            pop_times.append(time.perf_counter())
            p0 = pending.popleft()
            if on_pop:
                on_pop(p0, res, cur['first'])
            cur['first'] = False                # the runner has seen a frame: from now on every frame has a predecessor
            cur['last_n'] = int(res[2][-1])     # vnMatches12 of the next batch's first frame has this many entries
            nmatch_total[0] += int(res[4].sum())
            if pushed < nbatches:
                push(source)
                pushed += 1

    def sync():
        api.device_synchronize(local_rank)
        if have_torch_gpu:
            torch.cuda.synchronize()

    host_cpu = [0.0]
    own_elapsed = [0.0]

    def prewarm(source, seconds):
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            run(cur['subs'], source)

    samples = []                                # copies of batches popped inside the timed region, checked after the clock stops
    MAX_SAMPLES = 32

    def timed(nsteps, nwarm, source):
        run(nwarm * cur['subs'], source)
        del pop_times[:]
        del samples[:]
        every = -(-nsteps // MAX_SAMPLES)           # the last submission of every step (of every 2nd ... when there are more than 32 steps)
        popped = [0]

        def sample(p0, res, first):
            popped[0] += 1
            k, last = divmod(popped[0], cur['subs'])
            if last == 0 and (k - 1) % every == 0 and table is not None:
                samples.append((p0, tuple(a.copy() for a in res), cur.get('last_n', 0)))
        st.kernel_ms(reset=True)
        st.stats(reset=True)
        nmatch_total[0] = 0
        sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        c0 = time.process_time()
        run(nsteps * cur['subs'], source, sample)
        sync()
        t_own = time.perf_counter()
        host_cpu[0] = time.process_time() - c0      # CPU seconds of this rank's threads inside the timed region
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        own_elapsed[0] = t_own - t0                # this rank's own time (before it waited for the others)
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        return el

    # ---- self-check 1: the whole forwards-and-backwards period of the stream against the committed oracle digests (before the
    # timed region; self-check 2 -- the batches popped INSIDE the timed region -- follows it, see check_samples)
    verify = {'verified': None, 'outputs_sha256': None}
    table = None
    if not args.no_verify and not args.no_match and not args.bow and args.pool == wl.POOL:
        table = load_digest_table(seed)
        verify = verify_period(wl, table, seed, B, lambda n, cb: run(n, 'hbm', cb))
    verify['verified_own'] = verify['verified']
    if dist is not None:                        # every rank checks its own stream; rank 0 reports the conjunction
        flag = torch.tensor([1 if verify['verified'] else 0, 1 if verify['verified'] is None else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        verify['ranks_verified'] = int(flag[0])
        if verify['verified'] is not None:
            verify['verified'] = int(flag[0]) == world

    def check_samples():
        """Self-check 2: every frame of the batches copied inside the timed region against the digests of its stream position."""
        if table is None:
            return {'timed_verified': None, 'timed_frames_checked': 0}
        chk = wl.PositionChecker(table, args.pool)
        for p0, res, prev_n in samples:
            chk.check(p0, *res, prev_n=prev_n)
        verify['verified_own'] = bool(verify.get('verified_own') and not chk.bad and chk.frames > 0) if verify.get('verified_own') is not None else None
        out = {'timed_verified': bool(not chk.bad and chk.frames > 0), 'timed_frames_checked': chk.frames,
               'timed_batches_checked': len(samples), 'timed_distinct_period_positions': len(chk.positions),
               'timed_stream_positions': [int(samples[0][0]), int(samples[-1][0]) + cur['B'] - 1] if samples else None,
               'timed_mismatches': chk.bad[:8]}
        del samples[:]
        if dist is not None:
            flag = torch.tensor([1 if out['timed_verified'] else 0, chk.frames], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            out['timed_ranks_verified'], out['timed_frames_checked_all_ranks'] = int(flag[0]), int(flag[1])
            out['timed_verified'] = int(flag[0]) == world
        return out

    head_source = {'pinned': 'pinned', 'pageable': 'pageable', None: 'hbm'}[args.host_input]
    if args.prewarm_seconds > 0:
        prewarm(head_source, args.prewarm_seconds)
    elapsed = timed(args.steps, args.warmup, head_source)
    verify.update(check_samples())
    if verify.get('verified') is not None and verify.get('timed_verified') is not None:
        verify['verified_before_timing'] = verify['verified']
        verify['verified'] = bool(verify['verified'] and verify['timed_verified'])
    head_cpu = host_cpu[0]
    cpu_all = head_cpu                            # CPU seconds of every rank's threads inside the timed region, summed
    if dist is not None:
        t = torch.tensor([head_cpu], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        cpu_all = float(t[0])
    kms, kbatches, kframes = st.kernel_ms()
    wstats = st.stats()
    head_pops = np.array(pop_times)
    head_matches = nmatch_total[0]
    head_own = own_elapsed[0]
    own_step_ms = np.diff(head_pops[subs - 1::subs]) * 1e3 if len(head_pops) >= 3 * subs else None

    pcie_own = {}

    def pcie_leg():
        """The same stream from page-locked HOST frames, H2D of every frame inside the timed region (SURVEY.md s8(d)'s figure).
        Host frames: a batch computes only after its whole upload, so one batch fewer in flight serves the link better (26.2 k
        frames/s with 3, 25.3 k with 4) -- a runner of its own for this leg."""
        nonlocal st, lookahead
        if args.no_pcie or head_source != 'hbm':
            return None
        pdepth = min(3, max(1, args.depth))
        pB = min(B, wl.BATCH)                     # a batch computes only after its whole upload: 32-frame submissions serve the link better than 64
        if pdepth != max(1, args.depth) or pB != B:
            st.close()
            st = make_runner(pdepth, pB)
            cur['B'], cur['subs'], cur['first'] = pB, step_frames // pB, True
            pending.clear()
            st.set_queue_slots(want_lookahead + 2)
            lookahead = min(want_lookahead, st.queue_slots() - 2)
        psteps, pwarm = max(2, min(args.steps, 12)), max(1, min(args.warmup, 2))
        if dist is not None:
            dist.barrier()                        # all ranks measure their link at the same time: root-complex contention is the expected limiter
        link = api.h2d_rate_gbs(local_rank, pinned.base, pinned.frame_bytes * pB)
        pel = timed(psteps, pwarm, 'pinned')
        pcie_own['link'], pcie_own['fps'] = link, step_frames * psteps / max(own_elapsed[0], 1e-9)
        pchk = check_samples()
        pfps = world * step_frames * psteps / pel
        return {'value': round(pfps, 2), 'unit': 'frames/s', 'steps': psteps, 'warmup': pwarm,
                'timed_verified': pchk['timed_verified'], 'timed_frames_checked': pchk['timed_frames_checked'],
                'ms_per_step': round(pel / psteps * 1e3, 4),
                'input': 'the same %d-frame pool in page-locked host memory (orbfe_host_alloc); H2D of every frame inside the timed region' % args.pool,
                'batches_in_flight': pdepth, 'frames_per_submission': pB,
                'h2d_link_gbs_rank0': round(link, 2), 'h2d_link_frames_per_s_rank0': round(link * 1e9 / (W * H), 1),
                'frac_of_link_rank0': round(pfps / world * W * H / (link * 1e9), 4) if link > 0 else None}

    # N > 1: every rank runs the leg now (it has a barrier inside).  N = 1: after the blocking single-call legs below -- a second
    # runner adds streams to the process and shifts the stream -> hardware-queue mapping of whatever is created after it (the
    # page-locked single-frame call read 0.195 instead of 0.170 ms behind it: its upload lane and its compute stream on different queues)
    pcie = pcie_leg() if world > 1 else None

    # ---- every rank's own figures (N > 1: one all_gather_object): which rank / NUMA node lagged, and by how much
    mine = rank_record(rank, local_rank, numa_node, step_frames * args.steps / max(head_own, 1e-9),
                       (step_frames / (float(np.percentile(own_step_ms, 50)) * 1e-3)) if own_step_ms is not None and len(own_step_ms) > 1 else None,
                       verify.get('verified_own'), head_cpu / max(head_own, 1e-9), head_own, pcie_own.get('link'), seed,
                       pcie_own.get('fps'), numa_cpus)
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        frames_done = world * step_frames * args.steps
        fps = frames_done / elapsed
        # roofline of the dominant kernel: k_fast_tasks reads every pyramid pixel once (sum of level
        # sizes) and writes 4 B per surviving candidate; algorithmic bytes per launch = that x B.
        ex = api.Extractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, device=local_rank)   # geometry + candidate count only
        t = ex.tables()
        px = 0
        for l in range(wl.NLEVELS):
            lw = int(np.rint(np.float32(W) * t['isf'][l]))
            lh = int(np.rint(np.float32(H) * t['isf'][l]))
            px += lw * lh
        ex.extract_batch_ptrs(dev.ptrs[:1], H, W, dev.stride, True)
        ncand = sum(len(ex.candidates(l, 0)) for l in range(wl.NLEVELS))
        # the call pattern of the reference's Tracking thread: one blocking ORBextractor::operator() per frame
        # (Frame.cc:133), frame already in HBM; median wall time of 60 calls on distinct frames, outside the timed region
        single = None
        if world == 1 and not args.no_latency:
            kbuf = np.zeros((1, ex.cap), api.KP_DTYPE)
            dbuf = np.zeros((1, ex.cap, 32), np.uint8)
            lat = []
            for i in range(70):
                t_a = time.perf_counter()
                ex.extract_batch_ptrs(dev.ptrs[i % args.pool:i % args.pool + 1], H, W, dev.stride, True, kbuf, dbuf)
                lat.append(time.perf_counter() - t_a)
            lat = np.array(lat[10:]) * 1e3
            single = {'ms_median': round(float(np.median(lat)), 4), 'ms_p90': round(float(np.percentile(lat, 90)), 4),
                      'call': 'blocking orbfe_extract_batch of ONE resident 1080p frame (extract only), 60 calls'}
        # ... and what Frame.cc:133 really hands over: a HOST cv::Mat.  One blocking orbfe_extract per frame, host pointer in
        # -> keypoints / descriptors in host memory out, from (a) pageable and (b) page-locked memory; 60 calls on distinct frames
        single_host = None
        if world == 1 and not args.no_latency:
            single_host = {}
            kbuf = np.zeros((1, ex.cap), api.KP_DTYPE)
            dbuf = np.zeros((1, ex.cap, 32), np.uint8)
            nlat = 64
            pin_lat = api.PinnedFrames([frames[i % len(frames)] for i in range(nlat)])
            for name, ptr_of in (('pageable', lambda i: frames[i % len(frames)].ctypes.data), ('page_locked', lambda i: pin_lat.ptrs[i])):
                lat = []
                for i in range(nlat + 10):
                    p = ptr_of(i % nlat)
                    t_a = time.perf_counter()
                    ex.extract_batch_ptrs([p], H, W, W, False, kbuf, dbuf)
                    lat.append(time.perf_counter() - t_a)
                lat = np.array(lat[10:]) * 1e3
                single_host[name] = {'ms_median': round(float(np.median(lat)), 4), 'ms_p90': round(float(np.percentile(lat, 90)), 4)}
            # (b') the same page-locked call as a C caller issues it: arguments prepared once, the C entry point called directly -- what
            # include/orbfe/orb_shim.hpp pays; the figures above include os1_amd.api's per-call ctypes / numpy marshalling (about 5 us),
            # and the library's own clocks split the call into enqueue / GPU wait / output assembly (orbfe_debug_stage_ms)
            import ctypes as C
            nout = np.zeros(1, np.int32)
            arrs = [(C.c_void_p * 1)(pin_lat.ptrs[i]) for i in range(nlat)]
            cargs = lambda i: (ex.h, 1, arrs[i], 0, H, W, C.c_size_t(W), kbuf.ctypes.data_as(C.c_void_p), dbuf.ctypes.data_as(C.c_void_p),
                               int(kbuf.shape[1]), nout.ctypes.data_as(C.c_void_p))
            prepared = [cargs(i) for i in range(nlat)]
            lat, stages = [], []
            for i in range(nlat + 10):
                a_ = prepared[i % nlat]
                t_a = time.perf_counter()
                rc_ = ex.L.orbfe_extract_batch(*a_)
                lat.append(time.perf_counter() - t_a)
                stages.append(ex.stage_ms())
                if rc_:
                    raise SystemExit('bench: orbfe_extract_batch failed')
            lat = np.array(lat[10:]) * 1e3
            stg = np.median(np.array(stages[10:]), 0)
            single_host['page_locked_prepared_c_call'] = {'ms_median': round(float(np.median(lat)), 4), 'ms_p90': round(float(np.percentile(lat, 90)), 4),
                                                          'library_stages_ms': {'enqueue': round(float(stg[0]), 4), 'gpu_wait': round(float(stg[1]), 4),
                                                                                'output_assembly': round(float(stg[3]), 4), 'total': round(float(stg[4]), 4)}}
            # (c) the caller's OWN pageable buffer registered once (orbfe_host_register: a capture ring, a long-lived cv::Mat): page-locked route
            ring = np.stack([frames[i % len(frames)] for i in range(nlat)])
            reg = api.RegisteredArray(ring)
            lat = []
            for i in range(nlat + 10):
                p = ring[i % nlat].ctypes.data
                t_a = time.perf_counter()
                ex.extract_batch_ptrs([p], H, W, W, False, kbuf, dbuf)
                lat.append(time.perf_counter() - t_a)
            reg.close()
            lat = np.array(lat[10:]) * 1e3
            single_host['registered'] = {'ms_median': round(float(np.median(lat)), 4), 'ms_p90': round(float(np.percentile(lat, 90)), 4)}
            single_host['call'] = ('blocking orbfe_extract_batch of ONE 1080p frame in HOST memory (the cv::Mat of Frame.cc:133), '
                                   'keypoints + descriptors returned to host memory, %d calls on distinct frames' % nlat)
            pin_lat.free()
        fast_bytes_per_frame = px + 4 * ncand
        fast_ms_per_launch = kms[1] / max(kbatches, 1)
        achieved = fast_bytes_per_frame * B / (fast_ms_per_launch * 1e-3) / 1e9 if fast_ms_per_launch > 0 else 0.0
        prof = load_profile_counters(B)
        traffic = prof.get('traffic_bytes_per_launch')
        valu = None
        if prof.get('valu_insts_per_launch') and fast_ms_per_launch > 0:
            # the binding roof of this kernel: vector-instruction issue.  Cycles per wave64 instruction are MEASURED, twice:
            # tools/ubench/valu_rate.hip (profiles/r03_valu_rate.txt) -- 2 cycles for v_fma/add/mul_f32, v_add/sub_u32, and/or/xor,
            # right shifts and the unpacked 16-bit ops once two waves share a SIMD (the guide's figure), 4 cycles for every packed
            # 16-bit op, v_alignbyte, v_perm, v_lshlrev_b32, min/max_u32, 24-bit multiplies, DPP / SDWA forms, compares and any
            # op with a scalar source -- and for the kernel's own mix by the SQ counters (4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU)
            cpi = prof.get('valu_cycles_per_inst') or 4.0
            peak = 1024 * 2.4e9 / cpi
            rate = prof['valu_insts_per_launch'] / (fast_ms_per_launch * 1e-3)
            valu = {'insts_per_launch': prof['valu_insts_per_launch'], 'insts_per_cell_wave': prof.get('valu_insts_per_cell_wave'),
                    'cycles_per_inst': cpi, 'cycles_per_inst_source': 'SQ counters of this kernel (' + str(prof.get('source', ''))[:28] + '...) and profiles/r03_valu_rate.txt per opcode',
                    'issue_peak_per_s': round(peak, 1), 'achieved_per_s': round(rate, 1), 'frac': round(rate / peak, 4),
                    'frac_if_all_ops_were_2_cycle': round(rate / (1024 * 2.4e9 / 2.0), 4),
                    'note': 'wave64 VALU instructions (SQ_INSTS_VALU, separate rocprofv3 --pmc pass); peak = 1024 SIMDs x 2.4 GHz / cycles_per_inst'}
        # ... and the scalar port beside it (round 5): a CU has ONE scalar ALU, visited round-robin by its four SIMDs -- a SIMD's scalar port
        # takes one SALU instruction per ~4.25 cycles (tools/ubench/salu_rate.hip, profiles/r05_salu_rate.txt) and issues side by side
        # with the vector port when the two instructions come from different waves (same file, "mix" rows)
        salu = None
        if prof.get('salu_insts_per_launch') and fast_ms_per_launch > 0:
            scpi = prof.get('salu_cycles_per_inst') or 4.25
            speak = 1024 * 2.4e9 / scpi
            srate = (prof['salu_insts_per_launch'] + prof.get('smem_insts_per_launch', 0)) / (fast_ms_per_launch * 1e-3)
            salu = {'insts_per_launch': prof['salu_insts_per_launch'], 'smem_insts_per_launch': prof.get('smem_insts_per_launch'),
                    'insts_per_cell_wave': prof.get('salu_insts_per_cell_wave'), 'cycles_per_inst': scpi,
                    'issue_peak_per_s': round(speak, 1), 'achieved_per_s': round(srate, 1), 'frac': round(srate / speak, 4),
                    'scalar_over_vector_port_time': round((prof['salu_insts_per_launch'] + prof.get('smem_insts_per_launch', 0)) * scpi /
                                                          (prof['valu_insts_per_launch'] * (prof.get('valu_cycles_per_inst') or 4.0)), 4) if prof.get('valu_insts_per_launch') else None,
                    'note': 'SQ_INSTS_SALU + SQ_INSTS_SMEM (separate --pmc pass); peak = 1024 SIMDs x 2.4 GHz / 4.25 cycles per scalar instruction per SIMD '
                            '(measured, profiles/r05_salu_rate.txt); the scalar and the vector port issue side by side (different waves), so the two fractions do not add'}
        step_ms = np.diff(head_pops[subs - 1::subs]) * 1e3 if len(head_pops) >= 3 * subs else None
        value_p50 = round(world * step_frames / (float(np.percentile(step_ms, 50)) * 1e-3), 2) if step_ms is not None and len(step_ms) > 1 else None
        # the WHOLE path against both roofs (not only its dominant kernel): SURVEY.md s8(d)'s algorithmic 30.03 MB per frame x
        # frames/s against HBM, and the vector instructions of all kernels of a batch (SQ_INSTS_VALU summed over every dispatch of
        # the committed --pmc pass, profiles/counters.json) / (batch period x issue roof at the pipeline's own measured cycles
        # per instruction)
        batch_period_s = elapsed / max(args.steps * subs, 1)
        pipeline = {'hbm': {'algorithmic_MB_per_frame': 30.03, 'achieved_GBps_per_gpu': round(30.03e6 * fps / world / 1e9, 2),
                            'frac': round(30.03e6 * fps / world / (HBM_PEAK_GBS * 1e9), 5),
                            'note': 'SURVEY.md s8(d): pyramid build + FAST + full-level blur + descriptor patches per 1080p frame, resident input'}}
        if prof.get('pipeline_valu_insts_per_batch'):
            pcpi = prof.get('pipeline_valu_cycles_per_inst') or 4.0
            ppeak = 1024 * 2.4e9 / pcpi
            prate = prof['pipeline_valu_insts_per_batch'] / batch_period_s
            pipeline['valu'] = {'insts_per_batch': prof['pipeline_valu_insts_per_batch'], 'batch_period_ms': round(batch_period_s * 1e3, 4),
                                'cycles_per_inst': pcpi, 'issue_peak_per_s': round(ppeak, 1), 'achieved_per_s': round(prate, 1),
                                'frac': round(prate / ppeak, 4), 'by_kernel': prof.get('pipeline_valu_insts_per_batch_by_kernel'),
                                'note': 'all kernels of one %d-frame submission (extract + SearchForInitialization), counters from %s' % (B, str(prof.get('source', ''))[:40])}
        if prof.get('pipeline_salu_insts_per_batch'):
            scpi = prof.get('salu_cycles_per_inst') or 4.25
            n_s = prof['pipeline_salu_insts_per_batch'] + prof.get('pipeline_smem_insts_per_batch', 0)
            pipeline['salu'] = {'insts_per_batch': n_s, 'cycles_per_inst': scpi, 'issue_peak_per_s': round(1024 * 2.4e9 / scpi, 1),
                                'achieved_per_s': round(n_s / batch_period_s, 1), 'frac': round(n_s / batch_period_s / (1024 * 2.4e9 / scpi), 4)}
        out = {
            'metric': 'frames/sec extract+match, 1920x1080 @ 2000 ORB feats',
            'value': round(fps, 2), 'value_p50': value_p50,
            # SURVEY.md s8(d) puts the H2D of every frame inside its metric: the same stream from page-locked host frames, timed in this
            # run (detail: pcie_inclusive).  The pair travels together: `value` is the bench contract's resident rate, this the survey's.
            'value_pcie_inclusive': (pcie or {}).get('value'),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': '1080p_2000feat_8lv_1.2_extract%s%s_stream' % ('' if args.no_match else '+SearchForInitialization', '+ComputeBoW' if args.bow else ''),
                       'frames_per_step_per_gpu': step_frames, 'passes_over_the_pool_per_step': passes, 'frames_per_submission': B, 'distinct_frames_per_gpu': args.pool,
                       'image': '%dx%d' % (W, H), 'nfeatures': wl.NFEAT, 'nlevels': wl.NLEVELS,
                       'parallelism': 'independent streams, 1 per GPU (no collective)' if world > 1 else 'single GPU',
                       'matches_per_frame': round(head_matches / max(step_frames * args.steps, 1), 1),
                       'input': {'hbm': 'frames resident in HBM (pool of %d distinct frames, %d MB, walked forwards and backwards)' % (args.pool, args.pool * W * H >> 20),
                                 'pinned': 'frames in page-locked host memory (PCIe-inclusive)',
                                 'pageable': 'frames in pageable host memory (PCIe-inclusive)'}[head_source] + '; keypoints/descriptors/matches returned to host',
                       'value_is': ('the RESIDENT rate (frames in HBM when the timed region starts, the bench contract); SURVEY.md s8(d) puts the H2D of every '
                                    'frame inside its metric: that figure is pcie_inclusive.value' if head_source == 'hbm' else 'a PCIe-inclusive rate (developer run)'),
                       'batches_in_flight': head_in_flight, 'batches_in_flight_asked': max(1, args.depth), 'hip_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                       'worker_wait': 'spin' if wait_mode == '0' else 'sleep-poll %s us' % wait_mode,
                       'numa': {'node_of_gpu': numa_node, 'cpus_bound': numa_cpus}},
            'verified': verify['verified'], 'outputs_sha256': verify['outputs_sha256'], 'verify': verify,
            **per_rank_summary(per_rank, fps),
            'pcie_inclusive': pcie,
            'single_frame_latency': single,
            'single_frame_latency_host': single_host,
            # HIP-event time of the kernels in the pipeline (they overlap other batches' kernels); only k_fast_tasks is
            # always timed (roofline), the others appear with ORBFE_PROFILE_KERNELS=1 (costs about 1 % of the rate)
            'gpu_kernel_ms_per_frame': {k: round(v / max(kframes, 1), 5) for k, v in
                                        zip(('pyramid', 'fast_cells', 'compaction', 'describe', 'quadtree'), kms) if v > 0},
            'ms_per_step_series': [round(float(v), 3) for v in step_ms] if (step_ms is not None and os.environ.get('ORBFE_BENCH_SERIES')) else None,
            'ms_per_step_percentiles': (lambda d: {'p10': round(float(np.percentile(d, 10)), 4), 'p50': round(float(np.percentile(d, 50)), 4),
                                                    'p90': round(float(np.percentile(d, 90)), 4), 'max': round(float(d.max()), 4)})(step_ms) if step_ms is not None and len(step_ms) > 1 else None,
            # CPU time of rank 0's threads over the timed region / its wall time: what a rank needs from the host when
            # N ranks share the box's cores (Python driver + the stream runner's worker threads)
            'host_cpu_cores_used_rank0': round(head_cpu / max(elapsed, 1e-9), 2),
            'host_cpu_cores_used_all_ranks': round(cpu_all / max(elapsed, 1e-9), 2),
            'host_worker_ms_per_submission': {'submit': round(wstats[0] / max(wstats[3], 1), 4), 'collect_incl_gpu_wait': round(wstats[1] / max(wstats[3], 1), 4)},
            'roofline': {'kernel': 'k_fast_tasks', 'bound': 'hbm', 'binding_roof': 'valu-issue (see roofline.valu)', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                         'traffic_source': prof.get('source'),
                         'algorithmic_bytes_per_launch': fast_bytes_per_frame * B,
                         'launch_ms': round(fast_ms_per_launch, 4), 'valu': valu, 'salu': salu, 'pipeline': pipeline},
        }
        if world == 1 and not args.no_latency:
            out['config5_search_by_projection'] = config5_leg(api, local_rank, args.cpu_frames > 0)
            out['tracking_step'] = tracking_step_leg(api, local_rank, frames, W, H, wl, args.cpu_frames > 0)
            out['bow'] = bow_leg(api, local_rank, frames, W, H, wl, args.cpu_frames > 0)
        if world == 1:
            out['pcie_inclusive'] = pcie_leg()
            out['value_pcie_inclusive'] = (out['pcie_inclusive'] or {}).get('value')
        if world == 1 and args.cpu_frames > 0:
            out['cpu_baseline'] = cpu_baseline(frames, args.cpu_frames, not args.no_match)
            out['cpu_baseline_all_cores'] = cpu_baseline_all_cores(frames, not args.no_match)
        print(json.dumps(out), flush=True)
    st.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def rank_record(rank, device, numa_node, value, p50, verified, host_cores, own_elapsed_s, h2d_link_gbs, seed, pcie_value=None, cpus_bound=None):
    """One rank's own figures for the `per_rank` list of the bench line."""
    r = lambda v, n=2: None if v is None else round(float(v), n)
    return {'rank': rank, 'device': device, 'numa_node': numa_node, 'cpus_bound': cpus_bound, 'stream_seed': seed, 'value': r(value), 'p50': r(p50),
            'verified': verified, 'host_cores': r(host_cores), 'own_elapsed_s': r(own_elapsed_s, 4), 'h2d_link_gbs': r(h2d_link_gbs),
            'pcie_inclusive_value': r(pcie_value), 'host': socket.gethostname()}


def per_rank_summary(per_rank, aggregate):
    """`per_rank` (every rank's own rate over its own elapsed time, its NUMA node, its link) and what waiting for the slowest rank
    costs: the whole-job value is N x the slowest rank's rate (max-over-ranks timing), so value / sum of the ranks' own rates = 1 for a
    balanced node and drops when one rank (one NUMA node, one root complex) lags."""
    per_rank = sorted([p for p in per_rank if p], key=lambda p: p['rank'])
    vals = [p['value'] for p in per_rank if p.get('value')]
    out = {'per_rank': per_rank}
    if vals:
        slow = min(per_rank, key=lambda p: p['value'] if p.get('value') else float('inf'))
        out['scaling_efficiency_vs_min_rank'] = round(len(vals) * min(vals) / sum(vals), 4)
        out['rank_value_min_max'] = [round(min(vals), 2), round(max(vals), 2)]
        out['slowest_rank'] = {'rank': slow['rank'], 'device': slow['device'], 'numa_node': slow['numa_node']}
        out['aggregate_over_sum_of_rank_rates'] = round(aggregate / sum(vals), 4) if aggregate else None
    return out


def load_digest_table(seed):
    path = os.path.join(ROOT, 'tests', 'golden', 'stream1080_digests.json')
    try:
        d = json.load(open(path))
        return d['streams'].get(str(seed)) if d.get('format') == 2 else None
    except Exception:
        return None


def verify_period(wl, table, seed, B, run):
    """Push the first period of the stream (510 positions: the pool forwards, the turn-around, the pool backwards) through the runner
    and compare every frame's outputs with the oracle's digests of its position (tests/golden/stream1080_digests.json <-
    tools/gen_stream_digests.py).  The runner is fresh: position 0 has no predecessor."""
    if not table:
        return {'verified': None, 'outputs_sha256': None, 'note': 'no committed digest table for seed %d' % seed}
    chk = wl.PositionChecker(table)
    nsub = -(-wl.PERIOD // B)                        # submissions of B frames that cover the period
    run(nsub, lambda p0, res, first: chk.check(p0, *res, first_of_runner=first))
    want = sum(wl.expected_digests(table, p)[2] for p in range(nsub * B))
    ok = not chk.bad and chk.frames >= wl.PERIOD and len(chk.positions) == wl.PERIOD
    return {'verified': bool(ok), 'outputs_sha256': chk.outputs_sha256(wl.PERIOD),
            'submissions_checked': nsub, 'frames_per_submission': B, 'frames_checked': chk.frames,
            'distinct_period_positions': len(chk.positions), 'period': wl.PERIOD,
            'matches_in_checked_frames': chk.nmatches, 'oracle_matches': want, 'mismatches': chk.bad[:8],
            'digest_file': 'tests/golden/stream1080_digests.json', 'seed': seed}


def load_profile_counters(B):
    """Counters of k_fast_tasks from the committed rocprofv3 --pmc passes (profiles/counters.json names the
    profile tag and the commit they were taken at; they are offline measurements of the same command, not of this run)."""
    path = os.path.join(ROOT, 'profiles', 'counters.json')
    try:
        c = json.load(open(path))
    except Exception:
        return {}
    if c.get('batch') != B:
        return {}
    return c


def tracking_step_leg(api, device, frames, W, H, wl, with_oracle_check=True):
    """The front end of ONE tracked frame as Tracking.cc drives it (Frame.cc:133, Tracking.cc:608, 824), blocking calls, outside
    the timed region: ORBextractor::operator() on a page-locked host frame -> the frame's features stay on the GPU
    (orbfe_frame_create_from_extract) -> SearchByProjection(CurrentFrame, LastFrame, th) with the last frame's keypoints as
    sources (their descriptors stand in for the MapPoints' descriptors; handed over from host memory, as include/orbfe/orb_shim.hpp
    does, and -- second figure -- read from device memory, the rows of the last frame's resident copy) -> SearchByProjection(
    CurrentFrame, local MapPoints, th) with 3 000 MapPoints.  Median ms per stage over 60 frames."""
    import ctypes as C
    import numpy as np
    P = lambda a_: a_.ctypes.data_as(C.c_void_p)
    nfr = 61
    pin = api.PinnedFrames([frames[i] for i in range(nfr)])
    ex = api.Extractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, device=device)
    m = api.Matcher(device)
    sf = np.ascontiguousarray(ex.tables()['sf'], np.float32)
    bounds = (0.0, float(W), 0.0, float(H))
    kbuf = np.zeros((1, ex.cap), api.KP_DTYPE)
    dbuf = np.zeros((1, ex.cap, 32), np.uint8)
    rng = np.random.default_rng(3)
    t_ex, t_fr, t_ff, t_ffd, t_mp, t_mpt, nm1, nm2 = [], [], [], [], [], [], [], []
    tab = api.DescTable(4096, device)
    tperm = np.random.default_rng(4).permutation(4096)[:3000]
    trow = api.PinnedArray((3000,), np.int32)
    trow.a[:] = tperm
    prev = None
    prev_fr = None
    for i in range(nfr):
        t0 = time.perf_counter()
        _, _, n = ex.extract_batch_ptrs([pin.ptrs[i]], H, W, W, False, kbuf, dbuf)
        t1 = time.perf_counter()
        fr = api.Frame.from_extract(ex, 0, bounds)
        t2 = time.perf_counter()
        k, d = kbuf[0, :n[0]].copy(), dbuf[0, :n[0]].copy()
        if prev is not None:
            pk, pd = prev
            uv = np.stack([pk['x'] + 2, pk['y'] + 1], 1).astype(np.float32)       # the stream moves by (2, 1) px per frame
            valid = np.ones(len(pk), np.uint8)
            sflags = np.full(len(pk), 8, np.uint8)
            occ = np.zeros(len(k), np.uint8)
            rows = prev_fr.descriptors_device()
            # the searches are timed as C calls with their arguments prepared, as orb_shim.hpp issues them (the numpy marshalling
            # of os1_amd.api costs about 0.02 ms per call and is not the library's)
            lv1 = np.ascontiguousarray(pk['octave'], np.int32)
            an1 = np.ascontiguousarray(pk['angle'], np.float32)
            asg = np.full(max(len(k), 1), -1, np.int32)
            asg2 = np.full(max(len(k), 1), -1, np.int32)
            nmc = C.c_int(0)
            uv_args = [m.h, fr.h, P(sf), len(sf), P(occ), P(uv), P(lv1), P(an1), P(sflags), P(valid), P(pd), len(pk), C.c_float(15.0),
                       100, 0, 1, P(asg), C.byref(nmc)]
            t3 = time.perf_counter()
            rc = m.L.orbfe_search_by_projection_uv_frame(*uv_args)
            t4 = time.perf_counter()
            a = (nmc.value, asg[:len(k)].copy())
            uv_args[10] = C.c_void_p(rows.ptr)
            uv_args[16] = P(asg2)
            t4b = time.perf_counter()
            rc |= m.L.orbfe_search_by_projection_uv_frame(*uv_args)
            t_ffd.append(time.perf_counter() - t4b)
            if rc or nmc.value != a[0] or not (asg2[:len(k)] == a[1]).all():
                raise SystemExit('bench: device-resident and host descriptor rows disagree')
            src = rng.integers(0, len(k), 3000)
            mxy = (np.stack([k['x'][src], k['y'][src]], 1) + rng.uniform(-2, 2, (3000, 2))).astype(np.float32)
            lvl = np.ascontiguousarray(k['octave'][src], np.int32)
            vcos = np.full(3000, 0.95, np.float32)
            fl = np.full(3000, 1 | 8, np.uint8)
            occ2 = (a[1] >= 0).astype(np.uint8)
            md = np.ascontiguousarray(d[src])
            mp_args = (m.h, fr.h, P(sf), len(sf), P(occ2), P(mxy), P(lvl), P(vcos), P(fl), P(md), 3000, C.c_float(3.0), C.c_float(0.8),
                       P(asg), C.byref(nmc))
            t5 = time.perf_counter()
            rc = m.L.orbfe_search_by_projection_frame(*mp_args)
            t6 = time.perf_counter()
            if rc:
                raise SystemExit('bench: SearchByProjection failed')
            b = (nmc.value, asg[:len(k)].copy())
            # ... and with the local map's descriptors in the device table orb_shim.hpp keeps across frames (steady state: no
            # MapPoint's descriptor changed since the last frame, every row is read from device memory)
            tab.host.a[tperm] = md
            tab.upload(m, 0, 4096)
            m.synchronize()
            tb_args = (m.h, fr.h, P(sf), len(sf), P(occ2), P(mxy), P(lvl), P(vcos), P(fl), C.c_void_p(tab.dev), C.c_void_p(tab.host.base),
                       P(trow.a), 4096, 3000, C.c_float(3.0), C.c_float(0.8), P(asg2), C.byref(nmc))
            t7 = time.perf_counter()
            rc = m.L.orbfe_search_by_projection_frame_rows(*tb_args)
            t_mpt.append(time.perf_counter() - t7)
            if rc or nmc.value != b[0] or not (asg2[:len(k)] == b[1]).all():
                raise SystemExit('bench: SearchByProjection with the descriptor table differs from the plain rows')
            if i == 1 and with_oracle_check:   # one frame of the sequence against the oracle (the parity tests cover the rest)
                from oracle.pyoracle import Oracle
                o = Oracle()
                ou = o.search_by_projection_uv(k, d, bounds, sf, occ, uv, lv1, an1, sflags, valid, pd, 15.0, 100, 0, True)
                om = o.search_by_projection(k, d, bounds, sf, occ2, mxy, lvl, vcos, fl, md, 3.0, 0.8)
                if ou[0] != a[0] or not (ou[1] == a[1]).all() or om[0] != b[0] or not (om[1] == b[1]).all():
                    raise SystemExit('bench: tracking-step searches differ from the oracle')
            t_ex.append(t1 - t0); t_fr.append(t2 - t1); t_ff.append(t4 - t3); t_mp.append(t6 - t5)
            nm1.append(a[0]); nm2.append(b[0])
        prev = (k, d)
        if prev_fr is not None:
            prev_fr.close()
        prev_fr = fr
    prev_fr.close()
    pin.free()
    m.synchronize()
    tab.free()
    med = lambda v: round(float(np.median(v)) * 1e3, 4)
    return {'extract_host_frame_ms': med(t_ex), 'resident_frame_from_extract_ms': med(t_fr), 'search_by_projection_last_frame_ms': med(t_ff), 'search_by_projection_last_frame_device_rows_ms': med(t_ffd),
            'search_by_projection_mappoints_ms': med(t_mpt), 'search_by_projection_mappoints_host_rows_ms': med(t_mp),
            'front_end_total_ms': round(med(t_ex) + med(t_fr) + med(t_ff) + med(t_mpt), 4),
            'matches_last_frame_median': int(np.median(nm1)), 'matches_mappoints_median': int(np.median(nm2)),
            'note': '1080p / 2000 features, 60 frames; blocking C calls (extract from a page-locked host frame; searches with prepared arguments); '
                    'search_by_projection_mappoints_ms = the 3 000 MapPoints\' descriptors in the device table include/orbfe/orb_shim.hpp keeps across '
                    'frames (orbfe_search_by_projection_frame_rows, no row changed), _host_rows_ms = 32-byte rows handed over from host memory'}


def bow_leg(api, device, frames, W, H, wl, with_oracle):
    """The bag-of-words calls of the keyframe / relocalisation path, outside the timed region: Frame::ComputeBoW (DBoW2 transform
    at levelsup 4 on a k = 10, L = 6 vocabulary of synthetic words -- ORBvoc's shape; the real file is not on the box) and
    ORBmatcher::SearchByBoW between two consecutive frames of the stream, blocking calls through os1_amd.api, median ms of 60."""
    import numpy as np
    from os1_amd.synth import synth_vocabulary
    image = synth_vocabulary(1, 10, 6)
    v = api.Vocabulary(image, device)
    ex = api.Extractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, device=device)
    (k1, d1), (k2, d2) = ex(frames[0]), ex(frames[1])
    m = api.Matcher(device)

    def med(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return round(float(np.median(ts)) * 1e3, 4)
    t1, t2 = v.transform(d1, 4), v.transform(d2, 4)
    v1 = np.ones(len(k1), np.uint8)
    out = {'compute_bow_ms': med(lambda: v.transform(d1, 4), 60),
           'search_by_bow_ms': med(lambda: m.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True), 60),
           'features': [int(len(k1)), int(len(k2))], 'common_nodes': int(len(np.intersect1d(t1[2][0], t2[2][0])))}
    got = m.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True)
    out['matches'] = int(got[0])
    # both sides as resident frames (what orb_shim.hpp's MatcherContext passes): no descriptor row is copied
    bounds = (0.0, float(W), 0.0, float(H))
    f1, f2 = api.Frame.from_host(m, k1, d1, bounds), api.Frame.from_host(m, k2, d2, bounds)
    r1, r2 = f1.descriptors_device(), f2.descriptors_device()
    out['search_by_bow_resident_rows_ms'] = med(lambda: m.search_by_bow(r1, k1['angle'], v1, t1[2], r2, k2['angle'], None, t2[2], 0.7, True), 60)
    got2 = m.search_by_bow(r1, k1['angle'], v1, t1[2], r2, k2['angle'], None, t2[2], 0.7, True)
    if got2[0] != got[0] or not (got2[1] == got[1]).all():
        raise SystemExit('bench: SearchByBoW with resident rows differs')
    f1.close(); f2.close()
    if with_oracle:
        from oracle.pyoracle import Oracle
        o = Oracle()
        ov = o.vocabulary(image)
        out['oracle_1core_compute_bow_ms'] = med(lambda: ov.transform(d1, 4), 3)
        out['oracle_1core_search_by_bow_ms'] = med(lambda: o.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True), 3)
        want = o.search_by_bow(d1, k1['angle'], v1, t1[2], d2, k2['angle'], None, t2[2], 0.7, True)
        out['equal_to_oracle'] = bool(want[0] == got[0] and (want[1] == got[1]).all())
    v.close()
    return out


def config5_leg(api, device, with_oracle):
    """BASELINE.json configs[4], outside the timed region: one 3840x2160 frame, 4000 features, camera modo 1 (keypoints
    undistorted on the host), ORBmatcher::SearchByProjection(Frame&, MapPoints, th) against 10 000 MapPoints on the frame's
    device-resident features (orbfe_frame_create_from_extract): blocking C calls, median of 100, host arrays in -> kp_assigned
    out.  The oracle (1 core) runs the same search beside it and must agree."""
    import ctypes as C
    import numpy as np
    from os1_amd.synth import synth
    W, H, N, n_mp = 3840, 2160, 4000, 10000
    fx = fy = 2196.0
    cx, cy = 1839.0, 1155.0
    ex = api.Extractor(N, 1.2, 8, 20, 7, device=device)
    img = synth(5, W, H)
    k, d = ex(img)
    xy = api.undistort_equidistant(np.stack([k['x'], k['y']], 1), fx, fy, cx, cy)
    kun = k.copy()
    kun['x'], kun['y'] = xy[:, 0], xy[:, 1]
    bounds = api.compute_image_bounds(W, H, 1, fx, fy, cx, cy)
    fr = api.Frame.from_extract(ex, 0, bounds, xy)
    sf = np.ascontiguousarray(ex.tables()['sf'], np.float32)
    rng = np.random.default_rng(55)
    src = rng.integers(0, len(k), n_mp)
    mdesc = d[src].copy()
    flips = rng.integers(0, 41, n_mp)
    for i in range(n_mp):
        for b in rng.integers(0, 256, flips[i]):
            mdesc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    mxy = (np.stack([kun['x'][src], kun['y'][src]], 1) + rng.uniform(-3, 3, (n_mp, 2))).astype(np.float32)
    level = np.minimum(k['octave'][src] + rng.integers(0, 2, n_mp), 7).astype(np.int32)
    viewcos = rng.uniform(0.9, 1.0, n_mp).astype(np.float32)
    flags = np.full(n_mp, 1 | 8, np.uint8)
    occ = np.zeros(len(k), np.uint8)
    m = api.Matcher(device)
    assigned = np.full(len(k), -1, np.int32)
    nmat = C.c_int(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {'frame': '3840x2160, %d keypoints, fisheye-undistorted on the host' % len(k), 'mappoints': n_mp,
           'call': 'blocking orbfe_search_by_projection_frame (resident frame built by orbfe_frame_create_from_extract), 100 calls'}
    if with_oracle:
        from oracle.pyoracle import Oracle
        oracle = Oracle()
    # the pointer arguments are converted once, as a C++ caller would hold them (ctypes' per-call conversion is not the library's)
    args = (m.h, fr.h, P(sf), len(sf), P(occ), P(mxy), P(level), P(viewcos), P(flags), P(mdesc), n_mp)
    outs = (P(assigned), C.byref(nmat))
    tab = api.DescTable(n_mp + 2048, device)
    perm = rng.permutation(n_mp + 2048)[:n_mp]           # MapPoint i's row: unrelated to its position in the query vector
    tab.host.a[perm] = mdesc
    tab.upload(m, 0, n_mp + 2048)
    m.synchronize()
    trow = api.PinnedArray((n_mp,), np.int32)
    trow.a[:] = perm
    for th in (1.0, 5.0):
        lat = []
        for _ in range(110):
            t0 = time.perf_counter()
            rc = m.L.orbfe_search_by_projection_frame(*args, th, 0.8, *outs)
            lat.append(time.perf_counter() - t0)
            assert rc == 0
        lat = np.array(lat[10:]) * 1e3
        row = {'gpu_ms_median': round(float(np.median(lat)), 4), 'gpu_ms_p90': round(float(np.percentile(lat, 90)), 4), 'matches': int(nmat.value)}
        # the local map's descriptors in a device table (orbfe_search_by_projection_frame_rows, what orb_shim.hpp's MatcherContext
        # maintains across frames): 4 bytes of row index per MapPoint cross PCIe instead of 32 of descriptor
        ref_assigned, ref_n = assigned.copy(), nmat.value
        lat = []
        for _ in range(110):
            t0 = time.perf_counter()
            rc = m.L.orbfe_search_by_projection_frame_rows(*args[:9], C.c_void_p(tab.dev), C.c_void_p(tab.host.base), P(trow.a), n_mp + 2048, n_mp, th, 0.8, *outs)
            lat.append(time.perf_counter() - t0)
            assert rc == 0
        if nmat.value != ref_n or not (assigned == ref_assigned).all():
            raise SystemExit('bench: config 5 with the descriptor table differs from the plain rows')
        lat = np.array(lat[10:]) * 1e3
        row['gpu_ms_median_descriptor_table'] = round(float(np.median(lat)), 4)
        if with_oracle:
            t0 = time.perf_counter()
            for _ in range(3):
                on, oa = oracle.search_by_projection(kun, d, bounds, sf, occ, mxy, level, viewcos, flags, mdesc, th, 0.8)
            row['oracle_1core_ms'] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            row['equal_to_oracle'] = bool(on == nmat.value and (oa == assigned).all())
            row['speedup_vs_oracle_1core'] = round(row['oracle_1core_ms'] / row['gpu_ms_median'], 1)
        out['th_%g' % th] = row
    return out


def cpu_baseline(frames, nframes, do_match):
    """The CPU oracle (scalar C++ port of the reference path, g++ -O3, ONE core) on the same frames."""
    import numpy as np
    from oracle.pyoracle import Oracle, OracleExtractor
    from os1_amd import stream_workload as wl
    o = Oracle()
    ox = OracleExtractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, o)
    ox.extract(frames[0])                                           # warm-up
    prev = None
    t0 = time.perf_counter()
    for i in range(nframes):
        k, d = ox.extract(frames[i % len(frames)])
        if do_match and prev is not None:
            pxy = np.stack([prev[0]['x'], prev[0]['y']], 1)
            o.search_for_initialization(prev[0], prev[1], k, d, wl.BOUNDS, pxy, wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
        prev = (k, d)
    dt = time.perf_counter() - t0
    return {'value': round(nframes / dt, 3), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'host_cores_available': os.cpu_count(),
            'sample': '%d frames of the same 1920x1080 stream, extract%s, oracle/liborb_oracle.so (scalar C++ '
                      'restatement, g++ -O3 -ffp-contract=off), %.1f s' % (nframes, '+SearchForInitialization' if do_match else '', dt)}


def cpu_baseline_all_cores(frames, do_match):
    """The same oracle on EVERY host core: one extractor instance per core over independent
    pieces of the stream, 16 consecutive frames each, every frame matched against its predecessor.  Worker processes
    of 4 threads (tools/cpu_oracle_worker.py) start together at a wall-clock instant; every worker reports its own
    start and end, `value` = sum of the workers' own rates (a worker that was ready late does not stretch the others'
    clocks), `wall_value` = all frames over first start .. last end.  A failed worker fails the leg (reported)."""
    import tempfile
    import numpy as np
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))          # undo this rank's NUMA binding for this leg
    except Exception:
        pass
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:                                                             # a container may see 256 CPUs and own 16 of them
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = max(1, int(float(q) / float(period) + 0.5))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = max(1, int(q / period + 0.5))
        except Exception:
            pass
    visible = cores
    if quota:
        cores = min(cores, quota)
    tpp, per = 4, 16
    nproc = max(1, cores // tpp)
    shm = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    path = os.path.join(shm, 'orbfe_cpu_frames_%d.npy' % os.getpid())
    np.save(path, np.stack(frames[:64]))
    try:
        t0 = time.time() + (4.0 if nproc <= 16 else 10.0)            # interpreter + numpy start-up of every worker
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'cpu_oracle_worker.py'), path, str(p * tpp * per),
                                   str(tpp), str(per), repr(t0), str(int(do_match))], stdout=subprocess.PIPE, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS='1')) for p in range(nproc)]
        starts, ends, total, rate, failed = [], [], 0, 0.0, 0
        for p in procs:
            out, _ = p.communicate()
            if p.returncode == 0 and len(out.split()) == 3:
                ts, te, nf = out.split()
                starts.append(float(ts))
                ends.append(float(te))
                total += int(nf)
                rate += int(nf) / max(float(te) - float(ts), 1e-9)
            else:
                failed += 1
        if failed or not ends:
            return {'error': '%d of %d oracle workers failed' % (failed, nproc), 'cores': nproc * tpp, 'kind': 'port'}
        dt = max(ends) - min(starts)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    return {'value': round(rate, 2), 'wall_value': round(total / dt, 2), 'unit': 'frames/s', 'cores': nproc * tpp,
            'kind': 'port', 'cpus_visible': visible, 'cgroup_cpu_quota': quota,
            'start_skew_s': round(max(starts) - min(starts), 3),
            'sample': '%d processes x %d threads x %d consecutive frames, extract%s, %.1f s' % (
                nproc, tpp, per, '+SearchForInitialization (15 of 16 frames have a predecessor)' if do_match else '', dt)}


if __name__ == '__main__':
    main()
