"""Oracle-side walk over the bench stream (os1_amd/stream_workload.py): per-step digests of what the stream runner must
return -- ORBextractor::operator() on every frame (ORBextractor.cc:907-969) and SearchForInitialization of every frame
against its predecessor with vbPrevMatched := predecessor keypoints (ORBmatcher.cc:400-515, Tracking.cc:355-357,383-384).
TEST INFRASTRUCTURE ONLY (tests/, tools/gen_stream_digests.py)."""
import numpy as np

from os1_amd import stream_workload as wl
from oracle.pyoracle import Oracle, OracleExtractor


def oracle_stream_steps(seed, nsteps, batch=wl.BATCH, w=wl.W, h=wl.H, nfeat=wl.NFEAT, keep=False):
    """-> (step digests, total matches[, per-frame (kps, desc, nm, m12)])"""
    o = Oracle()
    ox = OracleExtractor(nfeat, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, o)
    sf = wl.StreamFrames(seed, w, h)
    bounds = (0.0, float(w), 0.0, float(h))
    prev = None
    steps, total, kept = [], 0, []
    for s in range(nsteps):
        fh, mh = [], []
        for i in range(batch):
            k, d = ox.extract(sf.frame(wl.pool_index(s * batch + i)))
            if prev is None:
                nm, m12 = 0, np.zeros(0, np.int32)
            else:
                pk, pd = prev
                nm, m12, _ = o.search_for_initialization(pk, pd, k, d, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2),
                                                         wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
            fh.append(wl.frame_digest(k, d, len(k)))
            mh.append(wl.match_digest(nm, m12, 0 if prev is None else len(prev[0])))
            total += int(nm)
            if keep:
                kept.append((k, d, int(nm), m12))
            prev = (k, d)
        steps.append(wl.step_digest(fh, mh))
    return (steps, total, kept) if keep else (steps, total)


def oracle_stream_table(seed, pool=wl.POOL, w=wl.W, h=wl.H, nfeat=wl.NFEAT, keep=False):
    """Per-position table of stream `seed` (os1_amd.stream_workload.expected_digests): extraction digests of the `pool` frames, match
    digests of every frame against its predecessor walking forwards (fwd[i]: pred = i-1; fwd[0] = no predecessor) and backwards
    (bwd[i]: pred = i+1), and the match counts.  About 256 oracle extractions + 510 searches = a minute per stream on one core."""
    o = Oracle()
    ox = OracleExtractor(nfeat, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, o)
    sf = wl.StreamFrames(seed, w, h, pool)
    bounds = (0.0, float(w), 0.0, float(h))
    ext = [ox.extract(sf.frame(i)) for i in range(pool)]

    def sfi(pred, cur):
        pk, pd = ext[pred]
        k, d = ext[cur]
        nm, m12, _ = o.search_for_initialization(pk, pd, k, d, bounds, np.stack([pk['x'], pk['y']], 1).reshape(-1, 2),
                                                 wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
        return int(nm), wl.match_digest(nm, m12, len(pk))
    fwd = [(0, wl.match_digest(0, np.zeros(0, np.int32), 0))] + [sfi(i - 1, i) for i in range(1, pool)]
    bwd = [sfi(i + 1, i) for i in range(pool - 1)]
    table = {'frames': [wl.frame_digest(k, d, len(k)) for k, d in ext], 'nkeys': [int(len(k)) for k, _ in ext],
             'fwd': [m for _, m in fwd], 'nm_fwd': [n for n, _ in fwd], 'bwd': [m for _, m in bwd], 'nm_bwd': [n for n, _ in bwd]}
    return (table, ext) if keep else table
