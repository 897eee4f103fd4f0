#!/usr/bin/env python3
"""bench.py -- frames/sec of the ORB front end (extract + match) on N MI355X GPUs of one node.

Metric (BASELINE.json): frames/sec extract+match, 1920x1080 @ 2000 ORB features, 8 levels, 1.2.
A "step" is one pass of the hot path over one batch of B synthetic frames of a seeded stream:
ORBextractor::operator() on every frame (orbfe_extract_batch, frames already resident in HBM) and
ORBmatcher::SearchForInitialization of every frame against its predecessor in the stream
(window 100, nnratio 0.9, checkOrientation) -- BASELINE.json configs[1]+[2] on the stream of
configs[3].  Keypoints / descriptors / match indices come back to host memory inside the timed
region (they are the path's outputs).  Independent streams shard across GPUs (one process and one
stream per GPU, no data-path collective; the only cross-rank traffic is the barrier and the
max-over-ranks of the elapsed time, done over gloo).

One JSON line on rank 0 (see the task's bench contract) with two extra objects:
  roofline     -- dominant kernel (k_fast_cells): algorithmic bytes per launch / HIP-event duration
  cpu_baseline -- the CPU oracle (scalar port, 1 core) on a bounded sample of the same frames
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS, SCALE, INI_TH, MIN_TH = 1920, 1080, 2000, 8, 1.2, 20, 7
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='frames per step and GPU')
    ap.add_argument('--cpu-frames', type=int, default=64, help='frames of the CPU-oracle baseline sample (0 = skip)')
    ap.add_argument('--bow', action='store_true', help='also run Frame::ComputeBoW (k=10, L=6 synthetic vocabulary) behind the descriptor kernel (not the headline value)')
    ap.add_argument('--no-match', action='store_true', help='extract only (configs[1])')
    ap.add_argument('--host-input', nargs='?', const='pageable', default=None, choices=['pageable', 'pinned'],
                    help='frames start in HOST memory, pageable or page-locked (PCIe-inclusive rate; never the headline value)')
    ap.add_argument('--depth', type=int, default=3, help='extraction batches in flight inside the stream runner')
    args = ap.parse_args()

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

    from os1_amd import api
    from os1_amd.synth import shifted, synth
    if api.device_count() <= local_rank:
        raise SystemExit('bench.py needs GPU %d (found %d): the product has no CPU fallback' % (local_rank, api.device_count()))

    B = args.batch
    seed = 100 + rank                       # config 4: stream g -> GPU g, seeds 100+g
    base = synth(seed, W, H)
    # frame i = the scene translated by (2i, i) px with fresh +-4 sensor noise ("each frame = previous shifted
    # by (2,1) px"; derived from the base frame so that noise does not accumulate along the stream)
    frames = [base] + [shifted(base, 2 * i, i, seed * 1000 + i) for i in range(1, B)]
    dev = api.DeviceFrames(frames, local_rank)
    pinned = api.PinnedFrames(frames) if args.host_input == 'pinned' else None
    host_ptrs = pinned.ptrs if pinned else [f.ctypes.data for f in frames]

    def push():
        if args.host_input:
            st.push_ptrs(host_ptrs, H, W, W, False)
        else:
            st.push_ptrs(dev.ptrs, H, W, dev.stride, True)
    # native stream runner: `depth` extractor handles + one matcher + two worker threads, all C++
    st = api.Stream(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, local_rank, B, max(1, args.depth))
    bounds = (0.0, float(W), 0.0, float(H))
    st.set_matching(bounds, 0 if args.no_match else 100, 0.9, True)   # window 100, nnratio 0.9, checkOrientation
    if args.bow:
        from os1_amd.synth import synth_vocabulary
        voc = api.Vocabulary(synth_vocabulary(1, 10, 6), local_rank)
        st.set_vocabulary(voc, 4)
    nmatch_total = [0]
    pop_times = []

    def run(nsteps):
        """nsteps passes of the hot path: push a batch (async extraction on the GPU, then SearchForInitialization
        of every frame against its predecessor), pop its keypoints / descriptors / matches in host memory.
        Up to depth+2 batches are in the pipeline; every push and pop of the nsteps batches is inside this call."""
        pushed = 0
        while pushed < min(args.depth + 2, nsteps):
            push()
            pushed += 1
        for _ in range(nsteps):
            _, _, n, _, nm = st.pop()
            pop_times.append(time.perf_counter())
            nmatch_total[0] += int(nm.sum())
            if pushed < nsteps:
                push()
                pushed += 1

    def sync():
        api.device_synchronize(local_rank)
        if have_torch_gpu:
            torch.cuda.synchronize()

    run(args.warmup)
    del pop_times[:]
    st.kernel_ms(reset=True)
    st.stats(reset=True)
    nmatch_total[0] = 0
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kms, kbatches, kframes = st.kernel_ms()
    wstats = st.stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank == 0:
        frames_done = world * B * args.steps
        fps = frames_done / elapsed
        # roofline of the dominant kernel (DESIGN.md "Roofline"): k_fast_cells reads every pyramid pixel once
        # (sum of level sizes) and writes 4 B per surviving candidate; algorithmic bytes per launch = that x B.
        ex = api.Extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank)   # geometry + candidate count only
        t = ex.tables()
        px = 0
        for l in range(NLEVELS):
            lw = int(np.rint(np.float32(W) * t['isf'][l]))
            lh = int(np.rint(np.float32(H) * t['isf'][l]))
            px += lw * lh
        ex.extract_batch_ptrs(dev.ptrs[:1], H, W, dev.stride, True)
        ncand = sum(len(ex.candidates(l, 0)) for l in range(NLEVELS))
        fast_bytes_per_frame = px + 4 * ncand
        fast_ms_per_launch = kms[1] / max(kbatches, 1)
        achieved = fast_bytes_per_frame * B / (fast_ms_per_launch * 1e-3) / 1e9 if fast_ms_per_launch > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')     # written from rocprofv3 --pmc passes (offline)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get('k_fast_cells_bytes_per_launch_b%d' % B)
            except Exception:
                traffic = None
        out = {
            'metric': 'frames/sec extract+match, 1920x1080 @ 2000 ORB feats',
            'value': round(fps, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': '1080p_2000feat_8lv_1.2_extract%s%s_stream' % ('' if args.no_match else '+SearchForInitialization', '+ComputeBoW' if args.bow else ''),
                       'frames_per_step_per_gpu': B, 'image': '%dx%d' % (W, H), 'nfeatures': NFEAT, 'nlevels': NLEVELS,
                       'parallelism': 'independent streams, 1 per GPU' if world > 1 else 'single GPU',
                       'matches_per_frame': round(nmatch_total[0] / max(B * args.steps, 1), 1),
                       'input': ('frames in %s host memory (PCIe-inclusive)' % args.host_input if args.host_input else 'frames resident in HBM') + '; keypoints/descriptors/matches returned to host'},
            # HIP-event time of the kernels in the pipeline (they overlap other batches' kernels); only k_fast_cells is
            # always timed (roofline), the others appear with ORBFE_PROFILE_KERNELS=1 (costs about 1 % of the rate)
            'gpu_kernel_ms_per_frame': {k: round(v / max(kframes, 1), 5) for k, v in
                                        zip(('pyramid', 'fast_cells', 'compaction', 'describe', 'quadtree'), kms) if v > 0},
            'ms_per_step_percentiles': (lambda d: {'p10': round(float(np.percentile(d, 10)), 4), 'p50': round(float(np.percentile(d, 50)), 4),
                                                    'p90': round(float(np.percentile(d, 90)), 4), 'max': round(float(d.max()), 4)})(np.diff(np.array(pop_times)) * 1e3) if len(pop_times) > 2 else None,
            'host_worker_ms_per_step': {'submit': round(wstats[0] / max(wstats[3], 1), 4), 'collect_incl_gpu_wait': round(wstats[1] / max(wstats[3], 1), 4),
                                        'match': round(wstats[2] / max(wstats[3], 1), 4)},
            'roofline': {'kernel': 'k_fast_cells', 'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                         'algorithmic_bytes_per_launch': fast_bytes_per_frame * B,
                         'launch_ms': round(fast_ms_per_launch, 4)},
        }
        if world == 1 and args.cpu_frames > 0:
            out['cpu_baseline'] = cpu_baseline(frames, args.cpu_frames, not args.no_match)
            out['cpu_baseline_all_cores'] = cpu_baseline_all_cores(frames)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(frames, nframes, do_match):
    """The CPU oracle (scalar C++ port of the reference path, g++ -O3, ONE core) on the same frames."""
    import numpy as np
    from oracle.pyoracle import Oracle, OracleExtractor
    o = Oracle()
    ox = OracleExtractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, o)
    bounds = (0.0, float(W), 0.0, float(H))
    ox.extract(frames[0])                                           # warm-up
    prev = None
    t0 = time.perf_counter()
    for i in range(nframes):
        k, d = ox.extract(frames[i % len(frames)])
        if do_match and prev is not None:
            pxy = np.stack([prev[0]['x'], prev[0]['y']], 1)
            o.search_for_initialization(prev[0], prev[1], k, d, bounds, pxy, 100, 0.9, True)
        prev = (k, d)
    dt = time.perf_counter() - t0
    return {'value': round(nframes / dt, 3), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'host_cores_available': os.cpu_count(),
            'sample': '%d frames of the same 1920x1080 stream, extract%s, oracle/liborb_oracle.so (scalar C++ '
                      'restatement, g++ -O3 -ffp-contract=off), %.1f s' % (nframes, '+SearchForInitialization' if do_match else '', dt)}


def cpu_baseline_all_cores(frames):
    """The same oracle with one extractor instance per host thread over independent frames;
    extraction only, 4 frames per thread, at most 64 threads so the run stays bounded."""
    import threading
    from oracle.pyoracle import Oracle, OracleExtractor
    nthreads = max(1, min(os.cpu_count() or 1, 64))
    o = Oracle()
    exs = [OracleExtractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, o) for _ in range(nthreads)]
    per = 4

    def work(i):
        for k in range(per):
            exs[i].extract(frames[(i * per + k) % len(frames)])

    ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {'value': round(nthreads * per / dt, 2), 'unit': 'frames/s', 'cores': nthreads, 'kind': 'port',
            'sample': '%d threads x %d frames, extract only, %.1f s' % (nthreads, per, dt)}


if __name__ == '__main__':
    main()
