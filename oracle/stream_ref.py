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
