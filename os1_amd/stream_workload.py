"""The synthetic camera stream of BASELINE.json configs[3] (SURVEY.md s8(d) "Config 4"): stream g has seed 100+g and
256 frames, each frame = the previous one shifted by (2,1) px.  Shared by bench.py, tools/gen_golden.py and the tests
so that the workload the bench measures is exactly the workload the parity tests and the committed digests cover.

Frame i of stream `seed` = os1_amd.synth.shifted(base, 2*i, i, seed*1000 + i) with base = synth(seed, W, H) (frame 0 is
the base itself): the scene translated by (2i, i) px with reflect fill and fresh +-4 sensor noise -- derived from the
base frame so that noise does not accumulate along the stream.  An endless stream walks the 256 frames forwards and
backwards (0,1,...,255,254,...,1,0,1,...), so that consecutive frames always differ by one (+-2,+-1) px step and the
pool (530 MB at 1080p) is larger than the 256 MB Infinity Cache.
"""
import hashlib

import numpy as np

from .synth import splitmix64, synth

W, H, NFEAT, NLEVELS, SCALE, INI_TH, MIN_TH = 1920, 1080, 2000, 8, 1.2, 20, 7
WINDOW, NNRATIO, CHECK_ORI = 100, 0.9, True          # Tracking.cc:383-384
POOL = 256
PERIOD = 2 * POOL - 2   # stream positions until the forwards-and-backwards walk repeats (510)
PASSES = 8      # passes over the pool in one bench step: a step = PASSES * POOL = 2 048 frames (23 ms at 87 k frames/s)
BATCH = 32      # frames per DIGEST step: what one entry of tests/golden/stream1080_digests.json covers (and the submission size of the parity tests)
SUBMIT = 64     # frames per submission the bench hands to the stream runner (any size: StepHasher folds the frames into 32-frame steps)
BOUNDS = (0.0, float(W), 0.0, float(H))


def stream_seed(rank):
    return 100 + rank


def pool_index(pos, pool=POOL):
    """Frame of the pool shown at stream position `pos` (forwards, then backwards, ...)."""
    if pool <= 1:
        return 0
    period = 2 * pool - 2
    r = pos % period
    return r if r < pool else period - r


class StreamFrames:
    """Generates frames of one stream; bit-identical to synth.shifted(base, 2i, i, seed*1000+i) but pads the base once."""

    def __init__(self, seed, w=W, h=H, pool=POOL):
        self.seed, self.w, self.h, self.pool = seed, w, h, pool
        self.base = synth(seed, w, h)
        self.pad = 2 * (pool - 1) + 1
        self._padded = None

    def frame(self, i):
        if i == 0:
            return self.base
        if self._padded is None:
            self._padded = np.pad(self.base, self.pad, mode='reflect')
        dx, dy, pad = 2 * i, i, self.pad
        out = self._padded[pad - dy:pad - dy + self.h, pad - dx:pad - dx + self.w].astype(np.int16)
        noise = (splitmix64(self.seed * 1000 + i, self.w * self.h, 7) % np.uint64(9)).astype(np.int16).reshape(self.h, self.w) - 4
        return np.ascontiguousarray(np.clip(out + noise, 0, 255).astype(np.uint8))

    def frames(self, n=None):
        return [self.frame(i) for i in range(self.pool if n is None else n)]


def frame_digest(kps, desc, n):
    """sha256 over one frame's outputs: count, cv::KeyPoint-layout records, descriptor rows."""
    h = hashlib.sha256()
    h.update(np.int32(n).tobytes())
    h.update(np.ascontiguousarray(kps[:n]).tobytes())
    h.update(np.ascontiguousarray(desc[:n]).tobytes())
    return h.hexdigest()


def match_digest(nmatches, m12, n_prev):
    """sha256 over one SearchForInitialization result: return value and vnMatches12 (size = predecessor's keypoints)."""
    h = hashlib.sha256()
    h.update(np.int32(nmatches).tobytes())
    h.update(np.ascontiguousarray(m12[:max(n_prev, 0)], dtype=np.int32).tobytes())
    return h.hexdigest()


def step_digest(frame_hex, match_hex):
    """One digest per step (= batch): the per-frame digests in stream order."""
    h = hashlib.sha256()
    for a, b in zip(frame_hex, match_hex):
        h.update(bytes.fromhex(a))
        h.update(bytes.fromhex(b))
    return h.hexdigest()


class StepHasher:
    """Folds popped batches (Stream.pop() tuples) of ANY size into digests of BATCH-frame steps (frames in stream order); carries the
    predecessor's keypoint count across batches (vnMatches12 of frame i has as many entries as frame i-1 has keypoints)."""

    def __init__(self, step=BATCH):
        self.prev_n = 0
        self.steps = []
        self.nmatches = 0
        self.step = step
        self._fh, self._mh = [], []

    def add(self, kps, desc, n, m12, nm):
        for i in range(len(n)):
            self._fh.append(frame_digest(kps[i], desc[i], int(n[i])))
            self._mh.append(match_digest(int(nm[i]), m12[i], self.prev_n))
            self.prev_n = int(n[i])
            self.nmatches += int(nm[i])
            if len(self._fh) == self.step:
                self.steps.append(step_digest(self._fh, self._mh))
                self._fh, self._mh = [], []
        return self.steps[-1] if self.steps else None


# ---- per-position digests (tests/golden/stream1080_digests.json, format 2) ---------------------------------------------------
# What the stream runner returns for stream position p depends on two pool frames only: cur = pool_index(p) (the extraction) and
# pred = pool_index(p - 1) (SearchForInitialization of cur against pred with vbPrevMatched := pred's keypoints, Tracking.cc:355-357,
# 383-384).  pred is cur - 1 (walking forwards), cur + 1 (walking backwards) or absent (the first frame a runner sees).  The file
# therefore holds, per stream, 256 extraction digests, 256 forward match digests (fwd[0] = "no predecessor") and 255 backward ones:
# enough to check ANY stream position -- the whole 510-position period and whatever the bench pops inside its timed region.

def expected_digests(table, pos, first_of_runner=False, pool=POOL):
    """(frame digest, match digest, oracle nmatches) the oracle holds for stream position `pos`; `first_of_runner`: the runner has not
    seen a frame before this one (no predecessor, whatever the position)."""
    cur = pool_index(pos, pool)
    if first_of_runner or pos == 0:
        return table['frames'][cur], table['fwd'][0], 0
    pred = pool_index(pos - 1, pool)
    if pred == cur - 1:
        return table['frames'][cur], table['fwd'][cur], table['nm_fwd'][cur]
    assert pred == cur + 1
    return table['frames'][cur], table['bwd'][cur], table['nm_bwd'][cur]


def expected_steps(table, nsteps, step=BATCH, pool=POOL):
    """Digests of the first `nsteps` step-sized pieces of the stream (positions 0 .. nsteps*step-1), derived from the per-position table:
    what StepHasher.steps must equal for a runner that started at position 0.  -> (steps, total matches)"""
    out, total = [], 0
    for s in range(nsteps):
        fh, mh = [], []
        for i in range(step):
            f, m, nm = expected_digests(table, s * step + i, pool=pool)
            fh.append(f); mh.append(m); total += nm
        out.append(step_digest(fh, mh))
    return out, total


class PositionChecker:
    """Checks popped batches (Stream.pop() tuples) frame by frame against the per-position table.  check() takes the stream position of
    the batch's first frame; the keypoint count of the frame before the batch (vnMatches12 has that many entries) comes either from the
    previous check() call (consecutive batches) or from `prev_n` (a batch picked out of the middle of the stream: the count of the last
    frame of the batch popped before it)."""

    def __init__(self, table, pool=POOL):
        self.table, self.pool = table, pool
        self.frames = 0
        self.bad = []           # (position, 'frame' | 'match') of every mismatch
        self.nmatches = 0
        self.positions = set()  # position mod period of every frame checked
        self.got = []           # (position, frame digest, match digest) of what was returned, in the order checked
        self._next_pos, self._prev_n = None, 0

    def outputs_sha256(self, npos=None):
        """One digest over the returned outputs of the first `npos` frames checked (independent of the submission size)."""
        h = hashlib.sha256()
        for _, f, m in self.got[:npos]:
            h.update(bytes.fromhex(f))
            h.update(bytes.fromhex(m))
        return h.hexdigest()

    def check(self, pos, kps, desc, n, m12, nm, prev_n=None, first_of_runner=False):
        if prev_n is None:
            prev_n = self._prev_n if self._next_pos == pos else 0
        for i in range(len(n)):
            p = pos + i
            ef, em, _ = expected_digests(self.table, p, first_of_runner and i == 0, self.pool)
            gf = frame_digest(kps[i], desc[i], int(n[i]))
            gm = match_digest(int(nm[i]), m12[i], 0 if (first_of_runner and i == 0) else prev_n)
            self.got.append((p, gf, gm))
            if gf != ef:
                self.bad.append((p, 'frame'))
            if gm != em:
                self.bad.append((p, 'match'))
            prev_n = int(n[i])
            self.nmatches += int(nm[i])
            self.frames += 1
            self.positions.add(p % max(2 * self.pool - 2, 1))
        self._next_pos, self._prev_n = pos + len(n), prev_n
        return not self.bad
