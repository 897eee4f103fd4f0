#!/usr/bin/env python3
"""Hazard H1 in numbers: how much of ORBextractor's output hangs on the order in which the reference's quadtree takes EQUAL-SIZED nodes.

The reference sorts (node size, node ADDRESS) pairs (`sort(vPrevSizeAndPointerToNode)`, /root/reference/src/ORBextractor.cc:716), walks them
from the back and stops dividing as soon as the level's quota is reached (:762-763).  Between nodes of equal size the heap address decides
which are divided before the break and in which order their children are pushed to the front of the list -- i.e. which keypoints a level
returns AND in which order.  The address is a function of the allocator's history, not of the image.  The oracle and the product pin
"creation order, later node = larger address" (rule 0).  This script runs the oracle with the other rules
    1  reversed creation order          2  the real addresses of this process's list nodes (glibc malloc, the reference's node size and
    3..6  seeded random orders             allocation pattern: what the reference's own binary would do in THIS heap state)
on BASELINE config 2 (seed 2, 1080p / 2000) and on 64 frames of bench stream 100, and reports per rule: keypoints that are in one set but not
in the other (identified by (x, y, octave); angle, response and descriptor are functions of those), output positions holding a different
keypoint, and what it does to SearchForInitialization between consecutive frames (matched point pairs in one result but not in the other).

CPU only (test infrastructure: uses the oracle).  Run:  python tools/h1_sensitivity.py [nframes=64] > profiles/r05_h1_sensitivity.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, OracleExtractor   # noqa: E402
from os1_amd import stream_workload as wl             # noqa: E402
from os1_amd.synth import synth                       # noqa: E402

RULES = [(1, 'reversed creation order'), (2, 'real heap addresses (this process)'), (3, 'random order, seed 3'), (4, 'random order, seed 4'),
         (5, 'random order, seed 5'), (6, 'random order, seed 6')]


def keyset(k):
    return set(zip(k['x'].tolist(), k['y'].tolist(), k['octave'].tolist()))


def compare(base, other):
    (bk, bd), (k, d) = base, other
    A, B = keyset(bk), keyset(k)
    npos = sum(1 for i in range(min(len(k), len(bk))) if k[i].tobytes() != bk[i].tobytes() or d[i].tobytes() != bd[i].tobytes()) + abs(len(k) - len(bk))
    # a keypoint present in both sets carries the same angle / response / descriptor (functions of position and level only)
    bmap = {(x, y, o): i for i, (x, y, o) in enumerate(zip(bk['x'].tolist(), bk['y'].tolist(), bk['octave'].tolist()))}
    same = all(bk[bmap[key]].tobytes() == k[i].tobytes() and bd[bmap[key]].tobytes() == d[i].tobytes()
               for i, key in enumerate(zip(k['x'].tolist(), k['y'].tolist(), k['octave'].tolist())) if key in bmap)
    return len(A - B), len(B - A), npos, same, len(bk), len(k)


def match_pairs(o, f1, f2):
    (k1, d1), (k2, d2) = f1, f2
    nm, m12, _ = o.search_for_initialization(k1, d1, k2, d2, wl.BOUNDS, np.stack([k1['x'], k1['y']], 1).reshape(-1, 2), wl.WINDOW, wl.NNRATIO, wl.CHECK_ORI)
    idx = np.nonzero(m12 >= 0)[0]
    return set((float(k1['x'][i]), float(k1['y'][i]), float(k2['x'][m12[i]]), float(k2['y'][m12[i]])) for i in idx)


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    o = Oracle()

    def extractor(rule):
        ox = OracleExtractor(wl.NFEAT, wl.SCALE, wl.NLEVELS, wl.INI_TH, wl.MIN_TH, o)
        ox.set_tie_rule(rule)
        return ox
    print('H1 sensitivity: quadtree tie-break between equal-sized nodes (ORBextractor.cc:716, 762-763); baseline = rule 0 (creation order)')
    print('inputs: BASELINE config 2 = synth(2, 1920, 1080), 2000 features, 8 levels; stream = %d frames of bench stream 100' % nframes)
    img = synth(2, wl.W, wl.H)
    o.tie_stats()
    base = extractor(0).extract(img)
    st = o.tie_stats()
    print('\nconfig 2, rule 0: %d keypoints; final-phase sorts %d (one per level), nodes sorted %d, of which %d had an equal-sized neighbour; '
          'early breaks %d, of which %d left an equal-sized node undivided' % (len(base[0]), st['sorts'], st['sorted_nodes'], st['nodes_in_ties'],
                                                                              st['breaks'], st['breaks_inside_a_tie']))
    print('%-40s %10s %10s %12s %10s' % ('rule', 'only in 0', 'only in r', 'positions !=', 'common =='))
    for rule, name in RULES:
        a, b, npos, same, n0, n1 = compare(base, extractor(rule).extract(img))
        print('%-40s %10d %10d %12d %10s' % ('%d %s' % (rule, name), a, b, npos, same))

    sf = wl.StreamFrames(100)
    frames = [sf.frame(i) for i in range(nframes)]
    ox0 = extractor(0)
    base_all = [ox0.extract(f) for f in frames]
    base_pairs = [match_pairs(o, base_all[i - 1], base_all[i]) for i in range(1, nframes)]
    nk = sum(len(k) for k, _ in base_all)
    nmt = sum(len(p) for p in base_pairs)
    print('\nstream 100, %d frames, rule 0: %d keypoints, %d SearchForInitialization matches over %d consecutive pairs' % (nframes, nk, nmt, nframes - 1))
    print('%-40s %10s %10s %12s %12s %12s %12s' % ('rule', 'only in 0', 'only in r', 'positions !=', 'frames !=', 'matches -', 'matches +'))
    for rule, name in RULES[:3]:
        ox = extractor(rule)
        allr = [ox.extract(f) for f in frames]
        a = b = npos = nfr = 0
        for x, y in zip(base_all, allr):
            da, db, dp, same, _, _ = compare(x, y)
            assert same
            a += da; b += db; npos += dp; nfr += int(dp > 0)
        pairs = [match_pairs(o, allr[i - 1], allr[i]) for i in range(1, nframes)]
        lost = sum(len(p0 - p1) for p0, p1 in zip(base_pairs, pairs))
        gained = sum(len(p1 - p0) for p0, p1 in zip(base_pairs, pairs))
        print('%-40s %10d %10d %12d %12d %12d %12d' % ('%d %s' % (rule, name), a, b, npos, nfr, lost, gained))
    print('\n(per cent of the stream\'s keypoints: divide "only in" by %d; of its matches: by %d)' % (nk, nmt))


if __name__ == '__main__':
    main()
