"""Helpers shared by the bag-of-words tests: vocabulary images and an independent pure-Python DBoW2 transform."""
import math

import numpy as np


def ragged_vocabulary(seed, k=6, L=4, scoring=0, weighting=0):
    """Random tree whose nodes have 0..k children (leaves at different depths), in the binary file layout."""
    rng = np.random.default_rng(seed)
    recs = []       # (parent, leaf, desc, weight)
    frontier = [(0, 0)]
    while frontier:
        nxt = []
        for pid, depth in frontier:
            nc = k if depth == 0 else int(rng.integers(0, k + 1))
            if depth >= L:
                nc = 0
            for _ in range(nc):
                recs.append([pid, 0, rng.integers(0, 256, 32, dtype=np.uint8), 0.0])
                nxt.append((len(recs), depth + 1))
        frontier = nxt
    has_child = set(r[0] for r in recs)
    for i, r in enumerate(recs):
        if (i + 1) not in has_child:
            r[1] = 1
            r[3] = 0.0 if rng.random() < 0.1 else float(rng.random() * 5 + 0.01)
    out = bytearray([k, L, scoring, weighting])
    for pid, leaf, desc, w in recs:
        out += np.int32(pid).tobytes() + bytes([leaf]) + desc.tobytes() + np.float64(w).tobytes()
    return bytes(out)


def with_header(image, scoring, weighting):
    return image[:2] + bytes([scoring, weighting]) + image[4:]


def py_transform(image, desc, levelsup):
    """TemplatedVocabulary::transform(features, v, fv, levelsup) written independently of the oracle (dicts, python
    floats = doubles, popcount via numpy)."""
    k, L, scoring, weighting = image[0], image[1], image[2], image[3]
    rec = np.frombuffer(image, np.uint8, offset=4).reshape(-1, 45)
    parent = rec[:, 0:4].copy().view('<i4').ravel()
    nodes_desc = np.vstack([np.zeros((1, 32), np.uint8), rec[:, 5:37]])
    weight = np.concatenate([[0.0], rec[:, 37:45].copy().view('<f8').ravel()])
    children = [[] for _ in range(len(rec) + 1)]
    word = np.zeros(len(rec) + 1, np.int64)
    nw = 0
    for i in range(len(rec)):
        children[parent[i]].append(i + 1)
        if rec[i, 4] > 0:
            word[i + 1] = nw
            nw += 1
    v, fv = {}, {}
    word_of, node_of = [], []
    for i, d in enumerate(desc):
        cur, level, nid = 0, 0, 0
        while True:
            level += 1
            ch = children[cur]
            dist = np.unpackbits(nodes_desc[ch] ^ d[None, :], axis=1).sum(axis=1)
            cur = ch[int(np.argmin(dist))]       # argmin = first minimum
            if level == L - levelsup:
                nid = cur
            if not children[cur]:
                break
        w = float(weight[cur])
        word_of.append(int(word[cur]))
        node_of.append(nid)
        if w > 0:
            wid = int(word[cur])
            if weighting in (0, 1):
                v[wid] = v.get(wid, 0.0) + w
            else:
                v.setdefault(wid, w)
            fv.setdefault(nid, []).append(i)
    ids = sorted(v)
    vals = [v[i] for i in ids]
    must = scoring != 5
    if weighting in (0, 1) and ids and not must:
        vals = [x / float(len(ids)) for x in vals]
    if must:
        if scoring == 1:
            norm = 0.0
            for x in vals:
                norm += x * x
            norm = math.sqrt(norm)
        else:
            norm = 0.0
            for x in vals:
                norm += abs(x)
        if norm > 0.0:
            vals = [x / norm for x in vals]
    return ids, vals, {n: fv[n] for n in sorted(fv)}, word_of, node_of


def fv_to_dict(fv):
    nodes, off, feat = fv
    return {int(nodes[i]): [int(x) for x in feat[off[i]:off[i + 1]]] for i in range(len(nodes))}
