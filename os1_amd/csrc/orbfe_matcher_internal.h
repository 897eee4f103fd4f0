// orbfe_matcher_internal.h -- shared by orbfe_matcher.hip (host-array searches), orbfe_frame.hip (device-resident frames
// + GPU-side match bookkeeping) and orbfe_bow.hip: the window kernel, its parameter block, buffer helpers and the matcher
// handle.  Reference semantics: Frame::GetFeaturesInArea (src/Frame.cc:209-262) over Frame::AssignFeaturesToGrid's
// 64x48 grid (Frame.cc:114-129, 264-274), ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1605-1621).
#pragma once
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"
#include "host_pool.h"

namespace orbfe {
void set_err(const char* fmt, ...);
}
using orbfe::set_err;

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return ORBFE_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)


namespace orbfe_match {

constexpr int kGridCols = 64, kGridRows = 48;  // FRAME_GRID_COLS / FRAME_GRID_ROWS (Frame.h:36-37)
constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // ORBmatcher.cc:37-39

// One "pair" = one train frame (grid-sorted keypoints + descriptors) and a run of queries against it.
struct PairInfo {
  int trainOff;    // first entry of the pair in sx/sy/soct/sidx
  int cellOff;     // first entry of the pair's cellStart table ([64*48+1] ints)
  int tdescOff;    // first descriptor row of the pair's train frame in tdesc
  float minX, minY, invW, invH;
};

struct MatchParams {
  // train frames, keypoints permuted into grid order (cell = ix*48+iy ascending, insertion order inside)
  const float* sx;
  const float* sy;
  const int* soct;
  const int* sidx;        // original keypoint index inside its frame
  const int* cellStart;   // per pair [64*48+1], values relative to the pair's trainOff
  const uint8_t* tdesc;   // descriptor rows in the same grid-sorted order as sx/sy/soct/sidx
  const PairInfo* pairs;
  // queries (all pairs concatenated)
  const int* qpair;       // pair of each query (nullptr: pair 0)
  const float* qx;
  const float* qy;
  const float* qr;        // < 0 : inactive query
  const int* qminL;
  const int* qmaxL;
  const uint8_t* qdesc;   // [nq][32]
  // indexed descriptor rows (a caller-maintained table, orbfe_search_by_projection_frame_rows): query q's descriptor is row
  // (qdescRow[q] & 0x7fffffff) of qdesc (bit 31 clear) or of qdescAlt (bit 31 set: the table's page-locked host mirror, for
  // rows the device copy has not received yet); nullptr: row q of qdesc
  const int* qdescRow = nullptr;
  const uint8_t* qdescAlt = nullptr;
  int nq;
  // outputs
  uint32_t* qcount;       // [nq]
  uint32_t* qoff;         // [nq] offset into pool
  uint32_t* pool;         // entries: idx | dist << 16, reference candidate order per query
  uint32_t poolCap;
  uint32_t* total;        // [1] pool entries claimed (may exceed poolCap: host retries with a larger pool)
  uint32_t* wpool = nullptr;   // wide records (resident-frame searches; sized for the worst case by the host)
  uint32_t* wtotal = nullptr;  // [1] words of it claimed
  // searches on a resident frame (orbfe_frame.hip) filter and annotate the candidates where they are produced, so the
  // bookkeeping kernel needs nothing but the entries themselves (per-candidate tests, independent of order):
  uint32_t* rec = nullptr;            // [nq][4]: {count | decision code << 16, entry 0, entry 1, entry 2}; lists of <= kRecEntries
                                      // live only here.  The decision code (decision_code below) tabulates the query's outcome for
                                      // every subset of its candidates being available, so the bookkeeping kernel's rounds are a
                                      // table lookup
  int codeMode = 0;                   // acceptance rule of the search: 0 SearchByProjection(MapPoints), 1 / 2 best <= maxDist
  float nnratio = 0.f;
  int maxDist = 0;
  const float* invSigma2 = nullptr;   // Fuse's gate e2 * mvInvLevelSigma2[octave] > chi2 (ORBmatcher.cc:896-903): dropped
  double chi2 = 0.0;
  int packOctave = 0;                 // entries carry the keypoint's octave in bits 25..30
  uint2* qword = nullptr;             // [nq] the record as the bookkeeping kernel keeps it in LDS: {idx0 | idx1 << 16, idx2 | code << 16};
                                      // lists beyond the record: {offset in the pool (of the wide record for four candidates),
                                      // count | kCodeWide / kCodeLongList << 16}
  // two small byte arrays of the page-locked query arena (claim flags, occupancy) reach the single-block bookkeeping kernel
  // as bit masks in device memory, packed by this kernel's many blocks on the way (64 flags per ballot)
  const uint8_t* bitSrc[2] = {nullptr, nullptr};
  unsigned long long* bitDst[2] = {nullptr, nullptr};
  int bitN[2] = {0, 0};
  uint8_t bitMask[2] = {0xff, 0xff};  // flag = (byte & mask) != 0
  int bitConst[2] = {-1, -1};         // >= 0: no source array, every flag has this value
  // raw queries: the kernel derives window and level range from the caller's own arrays (read in place when they are
  // page-locked) instead of from marshalled qx / qy / qr / qminL / qmaxL
  int rawKind = 0;                    // 0 marshalled; 1 SearchByProjection(F, MapPoints, th) (ORBmatcher.cc:55-73, 126-132);
                                      // 2 SearchByProjection(F, LastFrame / KeyFrame) from the projection on (:1353-1356, :1483-1485);
                                      // 3 the projected loops (radius given)
  const float* rawXY = nullptr;       // [nq][2]
  const int* rawLevel = nullptr;      // [nq]
  const float* rawAux = nullptr;      // kind 1: mTrackViewCos; kind 3: radius
  const uint8_t* rawFlags = nullptr;  // kind 1: ORBFE_MP_* bits; kinds 2, 3: valid
  const float* rawSf = nullptr;       // [32] mvScaleFactors in device memory
  float rawTh = 1.f;
  int rawFactor = 0;                  // kind 1: th != 1.0 (ORBmatcher.cc:49, 67-68)
};

__device__ __forceinline__ int hamming256(const uint32_t* __restrict__ a, const uint32_t q[8]) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a[i] ^ q[i]);
  return d;
}

// Outcome of a query restricted to the candidates whose bit is set in `avail`: 0 = no match, c + 1 = candidate c.
// mode 0: ORBmatcher::SearchByProjection(Frame&, MapPoints, th), src/ORBmatcher.cc:95-120 -- best / second best with their
// levels, TH_HIGH, the ratio test only when both lie on the same level; modes 1, 2: best <= maxDist (:1373-1383, :376-392).
// Entries: index | distance << 16 | octave << 25.
template <int NC>
__device__ __forceinline__ int outcome_of(int mode, int cnt, const uint32_t e[NC], unsigned avail, float nnratio, int maxDist) {
  int bestDist = mode == 2 ? INT_MAX : 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, best = -1;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    if (c >= cnt || !((avail >> c) & 1u)) continue;
    const int dist = (int)((e[c] >> 16) & 0x1ff), oct = (int)(e[c] >> 25);
    if (mode == 0) {
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct; best = c; }
      else if (dist < bestDist2) { bestLevel2 = oct; bestDist2 = dist; }
    } else if (dist < bestDist) {
      bestDist = dist; best = c;
    }
  }
  if (mode == 0) {
    if (bestDist > TH_HIGH) return 0;
    if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) return 0;
  } else if (bestDist > maxDist) {
    return 0;
  }
  return best + 1;
}
// Decision codes.  Record (up to three candidates): 8 x 2 bits, field p = outcome when exactly the candidates of bit pattern p
// are available (bits beyond the count do not matter).  Wide record (N = four to six candidates): 2^N x 3 bits, eight
// patterns per word (word p >> 3, field p & 7).  Field 0 is always 0 (nothing available: no match).
constexpr uint32_t kCodeLongList = 0xffffu;   // (field 0 of a real code is always 0) the list is longer than the record: read the pool
constexpr uint32_t kCodeWide = 0xfffeu;       // four to six candidates: entries in the pool, and in the wide pool a wide record
                                              // {idx0 | idx1 << 16, idx2 | idx3 << 16, idx4 | idx5 << 16, 2^N / 8 code words}
__host__ __device__ constexpr uint32_t wide_words(uint32_t n) { return 3u + (1u << (n - 3u)); }   // 5, 7, 11

// LPQ lanes per query (64/LPQ queries per wave), one LANE per grid column of the window.  Window
// semantics: Frame::GetFeaturesInArea, Frame.cc:209-262: columns ix ascending, rows iy ascending inside a
// column, insertion order inside a cell -- i.e. for column ix the contiguous run
// [cellStart[ix*48+cy0], cellStart[ix*48+cy1+1]) of the cell-sorted keypoint table.  Lane l of a query's
// group walks the run of column cx0+l (a handful of entries), a prefix sum over the group's hit counts
// gives every lane its output offset, so the candidate list comes out in exactly the reference order in
// a single pass with all columns in flight at once.  The host picks LPQ >= the widest window in columns.
// A block is 4 waves (kWinThreads / LPQ queries); the pool space of ALL its queries is claimed with ONE atomic: ten thousand
// same-address atomics with return (one per query) serialise in the L2 and took longer than the search itself.
constexpr int kWinThreads = 256;
constexpr int kRecEntries = 3;
constexpr int kTmpEntries = 6;   // lists gathered in LDS: those of the record, and the lists of four to six candidates for their wide record
constexpr int kWideMax = 6;
template <int LPQ>
__global__ __launch_bounds__(kWinThreads) void k_window_match(MatchParams M) {
  __shared__ uint32_t wtot[kWinThreads / 64];
  __shared__ uint32_t blockBase, blockBaseW;
  __shared__ uint32_t recTmp[(kWinThreads / LPQ) * kTmpEntries];   // short lists gathered per query (resident-frame searches)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPQ - 1);
  const int q = blockIdx.x * (kWinThreads / LPQ) + threadIdx.x / LPQ;
  const bool live = q < M.nq;
#pragma unroll
  for (int c = 0; c < 2; c++)
    for (int g0 = blockIdx.x * kWinThreads + wave * 64; g0 < M.bitN[c]; g0 += gridDim.x * kWinThreads) {
      const int g = g0 + lane;
      const unsigned long long bits =
          M.bitConst[c] >= 0 ? (M.bitConst[c] ? ~0ull : 0ull) : __ballot(g < M.bitN[c] && (M.bitSrc[c][g] & M.bitMask[c]) != 0);
      if (lane == 0) M.bitDst[c][g0 >> 6] = bits;
    }
  float r = -1.f, x = 0.f, y = 0.f;
  int minL = 0, maxL = -1;
  PairInfo pi = M.pairs[0];
  int drow = 0;   // row of the query's descriptor in an indexed table: requested together with the other query fields
  if (live && M.qdescRow) drow = M.qdescRow[q];
  if (live && M.rawKind) {
    const float2 p = reinterpret_cast<const float2*>(M.rawXY)[q];
    const int lvl = M.rawLevel[q];
    const unsigned fl = M.rawFlags[q];
    x = p.x; y = p.y;
    minL = lvl - 1;
    if (M.rawKind == 1) {
      maxL = lvl;                                      // GetFeaturesInArea(x, y, r * scale, nPredictedLevel - 1, nPredictedLevel)
      float rr = (fl & ORBFE_MP_CANDIDATO) ? 4.0f : ((double)M.rawAux[q] > 0.998 ? 2.5f : 4.0f);   // RadiusByViewingCos
      if (M.rawFactor) rr *= M.rawTh;
      r = ((fl & ORBFE_MP_IN_VIEW) && !(fl & ORBFE_MP_BAD)) ? rr * M.rawSf[lvl & 31] : -1.f;
    } else if (M.rawKind == 2) {
      maxL = lvl + 1;                                  // GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1)
      r = fl ? M.rawTh * M.rawSf[lvl & 31] : -1.f;
    } else {
      maxL = lvl;
      r = (fl && lvl >= 0) ? M.rawAux[q] : -1.f;       // no octave lies in [level - 1, level] for level < 0
    }
  } else if (live) {
    r = M.qr[q]; x = M.qx[q]; y = M.qy[q]; minL = M.qminL[q]; maxL = M.qmaxL[q];
    pi = M.pairs[M.qpair ? M.qpair[q] : 0];   // qpair == nullptr: every query searches pair 0 (a resident frame)
  }
  const float* sx = M.sx + pi.trainOff;
  const float* sy = M.sy + pi.trainOff;
  const int* soct = M.soct + pi.trainOff;
  const int* sidx = M.sidx + pi.trainOff;
  const int* cellStart = M.cellStart + pi.cellOff;
  const uint8_t* tdesc = M.tdesc + (size_t)pi.tdescOff * 32;
  int cx0 = 0, cx1 = -1, cy0 = 0, cy1 = -1;
  if (live && r >= 0.f) {
    cx0 = max(0, (int)floorf((x - pi.minX - r) * pi.invW));
    cx1 = min(kGridCols - 1, (int)ceilf((x - pi.minX + r) * pi.invW));
    cy0 = max(0, (int)floorf((y - pi.minY - r) * pi.invH));
    cy1 = min(kGridRows - 1, (int)ceilf((y - pi.minY + r) * pi.invH));
    if (cx0 >= kGridCols || cx1 < 0 || cy0 >= kGridRows || cy1 < 0) cx1 = cx0 - 1;  // empty window
  }
  // the query's descriptor is fetched now, with the window still unknown: it may come over PCIe (page-locked rows read in
  // place), and that round trip then overlaps the walk of the grid instead of following it
  uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live && r >= 0.f) {
    const uint4* qp = reinterpret_cast<const uint4*>(M.qdesc + (size_t)q * 32);
    if (M.qdescRow) {
      const unsigned row = (unsigned)drow;
      qp = reinterpret_cast<const uint4*>(((row >> 31) ? M.qdescAlt : M.qdesc) + (size_t)(row & 0x7fffffffu) * 32);
    }
    const uint4 a = qp[0], b4 = qp[1];
    qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b4.x; qd[5] = b4.y; qd[6] = b4.z; qd[7] = b4.w;
  }
  const bool checkLevels = (minL > 0) || (maxL >= 0);
  const int ix = cx0 + sub;
  int b = 0, e1 = 0;
  if (ix <= cx1) {
    b = cellStart[ix * kGridRows + cy0];
    e1 = cellStart[ix * kGridRows + cy1 + 1];
  }
  auto inWindow = [&](int e) -> bool {
    if (checkLevels) {
      const int o = soct[e];
      if (o < minL) return false;
      if (maxL >= 0 && o > maxL) return false;
    }
    const float dx = sx[e] - x, dy = sy[e] - y;
    if (!(fabsf(dx) < r && fabsf(dy) < r)) return false;
    if (M.invSigma2) {   // (u - kp.x)^2 + (v - kp.y)^2: a float difference and its negation have the same square
      const float e2 = dx * dx + dy * dy;
      if ((double)(e2 * M.invSigma2[soct[e]]) > M.chi2) return false;
    }
    return true;
  };
  // pass 1: hits per column (the first 64 entries of a run remember their verdict in a bit mask, so the second pass
  // does not fetch their coordinates again), prefix sum inside the query's lane group.  The run is walked four entries at
  // a time with all their loads issued before the first test: a lane's run is a handful of entries, and one dependent
  // memory round trip per ENTRY (what a plain loop costs) was the kernel's whole duration.
  int hits = 0;
  unsigned long long hm = 0;
  for (int e = b; e < e1; e += 4) {
    float kx[4], ky[4];
    int ko[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int ee = min(e + u, e1 - 1);
      kx[u] = sx[ee]; ky[u] = sy[ee]; ko[u] = soct[ee];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (e + u >= e1) break;
      bool h = true;
      if (checkLevels) h = !(ko[u] < minL) && !(maxL >= 0 && ko[u] > maxL);
      const float dx = kx[u] - x, dy = ky[u] - y;
      h = h && fabsf(dx) < r && fabsf(dy) < r;
      if (h && M.invSigma2) {   // (u - kp.x)^2 + (v - kp.y)^2: a float difference and its negation have the same square
        const float e2 = dx * dx + dy * dy;
        if ((double)(e2 * M.invSigma2[ko[u]]) > M.chi2) h = false;
      }
      hits += h ? 1 : 0;
      if (h && e + u - b < 64) hm |= 1ull << (e + u - b);
    }
  }
  // pass 1b (searches on a resident frame): candidates whose distance rules them out whatever the other queries do never
  // enter the list -- shorter lists mean records instead of walked lists and fewer rounds of the bookkeeping, with the
  // same outcome.  Modes 1, 2 (best <= maxDist decides alone, ORBmatcher.cc:1373-1383, :376-392): dist > maxDist.  Mode 0
  // (:95-120): a candidate with dist > TH_HIGH and nnratio * dist >= TH_HIGH can neither be accepted nor, as second best, make
  // the ratio test reject an acceptable best (bestDist <= TH_HIGH <= nnratio * dist); dropping it leaves, in its place, a
  // second best that is at least as far, i.e. as irrelevant.  Distances are computed here for the verdict only; the
  // survivors' entries are produced by pass 2 as before.  Runs beyond the 64-entry mask are left as they are.
  const uint4* td4 = reinterpret_cast<const uint4*>(tdesc);   // descriptors are grid-sorted, 32-byte rows
  if (M.rec && hm) {
    unsigned long long left = hm;
    while (left) {
      const int a0 = __builtin_ctzll(left);
      left &= left - 1;
      const int a1 = left ? __builtin_ctzll(left) : a0;
      left &= left - 1;   // (no-op when a1 == a0 was the last bit: left is 0)
      const uint4 da0 = td4[(size_t)(b + a0) * 2], da1 = td4[(size_t)(b + a0) * 2 + 1], db0 = td4[(size_t)(b + a1) * 2],
                  db1 = td4[(size_t)(b + a1) * 2 + 1];
      const int d0 = __popc(da0.x ^ qd[0]) + __popc(da0.y ^ qd[1]) + __popc(da0.z ^ qd[2]) + __popc(da0.w ^ qd[3]) + __popc(da1.x ^ qd[4]) +
                     __popc(da1.y ^ qd[5]) + __popc(da1.z ^ qd[6]) + __popc(da1.w ^ qd[7]);
      const int d1 = __popc(db0.x ^ qd[0]) + __popc(db0.y ^ qd[1]) + __popc(db0.z ^ qd[2]) + __popc(db0.w ^ qd[3]) + __popc(db1.x ^ qd[4]) +
                     __popc(db1.y ^ qd[5]) + __popc(db1.z ^ qd[6]) + __popc(db1.w ^ qd[7]);
      // (d > TH_HIGH spelled out: for nnratio >= 1 the product alone would drop an acceptable candidate, e.g. d == 100)
      const bool drop0 = M.codeMode == 0 ? (d0 > TH_HIGH && M.nnratio * (float)d0 >= (float)TH_HIGH) : d0 > M.maxDist;
      const bool drop1 = M.codeMode == 0 ? (d1 > TH_HIGH && M.nnratio * (float)d1 >= (float)TH_HIGH) : d1 > M.maxDist;
      if (drop0) { hm &= ~(1ull << a0); hits--; }
      if (drop1 && a1 != a0) { hm &= ~(1ull << a1); hits--; }
    }
  }
  int incl = hits;
#pragma unroll
  for (int o = 1; o < LPQ; o <<= 1) {
    const int t = __shfl_up(incl, o, LPQ);
    if (sub >= o) incl += t;
  }
  const uint32_t count = (uint32_t)__shfl(incl, LPQ - 1, LPQ);
  // a resident-frame search keeps lists of up to kRecEntries candidates in the query's fixed 16-byte record (no pool space,
  // no offset to chase); longer lists go to the pool like every list of the host-resolved searches
  const bool inRec = M.rec != nullptr && count <= (uint32_t)kRecEntries;
  const bool wide = M.rec != nullptr && count >= 4u && count <= (uint32_t)kWideMax;   // entries to the pool AND (via LDS) a wide record
  // offsets: exclusive scan of the queries' counts inside the wave, of the wave totals inside the block, one atomic per pool.
  // Both sums travel in one word: pool entries in the low 24 bits, wide-record words in the high 8 (a wave has at most
  // eight queries: 8 x 65 535 entries, 8 x 11 words).
  const uint32_t mine = (live && sub == 0 && !inRec) ? (count | (wide ? wide_words(count) << 24 : 0u)) : 0u;
  uint32_t wincl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(wincl, o, 64);
    if (lane >= o) wincl += t;
  }
  if (lane == 63) wtot[wave] = wincl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0, sw = 0;
    for (int w = 0; w < kWinThreads / 64; w++) { s += wtot[w] & 0xffffffu; sw += wtot[w] >> 24; }
    blockBase = s ? atomicAdd(M.total, s) : 0u;
    blockBaseW = sw ? atomicAdd(M.wtotal, sw) : 0u;
  }
  __syncthreads();
  uint32_t off = blockBase + ((wincl - mine) & 0xffffffu), offW = blockBaseW + ((wincl - mine) >> 24);
  for (int w = 0; w < wave; w++) { off += wtot[w] & 0xffffffu; offW += wtot[w] >> 24; }
  if (live && sub == 0) {
    M.qcount[q] = count;
    M.qoff[q] = off;
  }
  off = __shfl(off, 0, LPQ);
  if (hits == 0 && !M.rec) return;
  const int qlocal = threadIdx.x / LPQ;
  // pass 2: distances, written at the lane's position in column order; the hits of the mask are fetched two at a time
  uint32_t pos = (uint32_t)(incl - hits);   // position inside the query's list
  auto emit = [&](int e, int idx, int oct, const uint4& d0, const uint4& d1) {
    const int d = __popc(d0.x ^ qd[0]) + __popc(d0.y ^ qd[1]) + __popc(d0.z ^ qd[2]) + __popc(d0.w ^ qd[3]) + __popc(d1.x ^ qd[4]) +
                  __popc(d1.y ^ qd[5]) + __popc(d1.z ^ qd[6]) + __popc(d1.w ^ qd[7]);
    uint32_t entry = (uint32_t)idx | ((uint32_t)d << 16);
    if (M.packOctave) entry |= (uint32_t)oct << 25;
    if (inRec || wide) recTmp[qlocal * kTmpEntries + pos] = entry;
    if (!inRec && off + pos < M.poolCap) M.pool[off + pos] = entry;
    pos++;
  };
  while (hm) {
    const int a0 = __builtin_ctzll(hm);
    hm &= hm - 1;
    const int a1 = hm ? __builtin_ctzll(hm) : a0;
    const bool two = hm != 0;
    if (two) hm &= hm - 1;
    const int ea = b + a0, eb = b + a1;
    const int ia = sidx[ea], ib = sidx[eb], oa = soct[ea], ob = soct[eb];
    const uint4 da0 = td4[(size_t)ea * 2], da1 = td4[(size_t)ea * 2 + 1], db0 = td4[(size_t)eb * 2], db1 = td4[(size_t)eb * 2 + 1];
    emit(ea, ia, oa, da0, da1);
    if (two) emit(eb, ib, ob, db0, db1);
  }
  for (int e = b + 64; e < e1; e++) {   // runs beyond the mask (very wide windows)
    if (!inWindow(e)) continue;
    emit(e, sidx[e], soct[e], td4[(size_t)e * 2], td4[(size_t)e * 2 + 1]);
  }
  if (M.rec) {
    // the query's record: its lanes' entries meet in LDS (a query's lanes are one wave's: a wave-level fence is enough).
    // The outcomes are tabulated by the group's first eight lanes, ONE availability pattern each (one per code word for a
    // wide record), and OR-ed together by three shuffles; lane 0 writes the 16 bytes at once.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (live && inRec) {
      uint32_t e3[3];
#pragma unroll
      for (int c = 0; c < 3; c++) e3[c] = (uint32_t)c < count ? recTmp[qlocal * kTmpEntries + c] : 0u;
      const unsigned p = (unsigned)sub & 7u;
      uint32_t code = 0u;
      if (sub < 8 && p) code = (uint32_t)outcome_of<3>(M.codeMode, (int)count, e3, p & ((1u << count) - 1u), M.nnratio, M.maxDist) << (2u * p);
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) code |= (uint32_t)__shfl_xor((int)code, o, LPQ);
      if (sub == 0) {
        reinterpret_cast<uint4*>(M.rec)[q] = make_uint4(count | (code << 16), e3[0], e3[1], e3[2]);
        if (M.qword) M.qword[q] = make_uint2((e3[0] & 0xffffu) | (e3[1] << 16), (e3[2] & 0xffffu) | (code << 16));
      }
    } else if (live && wide) {
      uint32_t e6[kWideMax];
#pragma unroll
      for (int c = 0; c < kWideMax; c++) e6[c] = (uint32_t)c < count ? recTmp[qlocal * kTmpEntries + c] : 0u;
      const unsigned p = (unsigned)sub & 7u;
      const int nW = 1 << ((int)count - 3);     // code words: 2, 4, 8
      uint32_t* wr = M.wpool + offW;
#pragma unroll
      for (int w = 0; w < 8; w++) {
        if (w >= nW) break;                      // (uniform inside the group)
        uint32_t word = 0u;
        if (sub < 8 && (w | p)) word = (uint32_t)outcome_of<kWideMax>(M.codeMode, (int)count, e6, 8u * w + p, M.nnratio, M.maxDist) << (3u * p);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) word |= (uint32_t)__shfl_xor((int)word, o, LPQ);
        if (sub == 0) wr[3 + w] = word;
      }
      if (sub == 0) {
        wr[0] = (e6[0] & 0xffffu) | (e6[1] << 16);
        wr[1] = (e6[2] & 0xffffu) | (e6[3] << 16);
        wr[2] = (e6[4] & 0xffffu) | (e6[5] << 16);
        reinterpret_cast<uint4*>(M.rec)[q] = make_uint4(count | (kCodeWide << 16), 0u, 0u, 0u);
        if (M.qword) M.qword[q] = make_uint2(offW, count | (kCodeWide << 16));
      }
    } else if (live && sub == 0) {
      reinterpret_cast<uint4*>(M.rec)[q] = make_uint4(count | (kCodeLongList << 16), 0u, 0u, 0u);
      if (M.qword) M.qword[q] = make_uint2(off, count | (kCodeLongList << 16));
    }
  }
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n) return ORBFE_OK;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
    return ORBFE_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};
template <class T>
struct PinBuf {
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n) return ORBFE_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
    // Coherent (fine-grained, uncached on the GPU side) EXPLICITLY: kernels store results and completion words here and the
    // host polls them while the kernel runs; with hipHostMallocDefault that property would hang on HIP_HOST_COHERENT.
    HIP_TRY(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocCoherent));
    n = count;
    return ORBFE_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace orbfe_match
using namespace orbfe_match;

struct orbfe_frame;
struct orbfe_matcher {
  int device = 0;
  hipStream_t stream = nullptr;
  std::shared_ptr<void> bow;   // scratch of orbfe_search_by_bow (orbfe_bow.hip)
  DevBuf<uint8_t> d_in;    // packed upload arena
  PinBuf<uint8_t> h_in;
  DevBuf<uint32_t> d_out;  // [total(1) pad][qcount nq][qoff nq]
  PinBuf<uint32_t> h_out;
  DevBuf<uint32_t> d_pool;
  PinBuf<uint32_t> h_pool;
  // searches on device-resident frames (orbfe_frame.hip): query arena, result / scratch words, the transient frame of
  // the host-array call forms, rounds the last k_resolve needed (< 0: finished by its serial pass)
  DevBuf<uint8_t> d_q;
  PinBuf<uint8_t> h_q;
  DevBuf<int> d_r;
  PinBuf<int> h_r;
  orbfe_frame* scratch = nullptr;
  int lastRounds = 0, lastResolveRoute = 0;   // route: 2 = tables + entries in LDS, 1 = tables in LDS, 0 = global scratch
  bool resolveAttr[3] = {false, false, false};   // dynamic-LDS attribute set (per matcher: one host thread per handle)
  ~orbfe_matcher() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    orbfe_frame_destroy(scratch);
    d_q.release(); h_q.release(); d_r.release(); h_r.release(); d_sf.release(); d_wpool.release();
    d_in.release(); h_in.release(); d_out.release(); h_out.release(); d_pool.release(); h_pool.release();
    if (stream) (void)hipStreamDestroy(stream);
  }

  // One search job: a train frame and a run of queries against it (host pointers).
  struct Job {
    const OrbfeKeyPoint* kps; const uint8_t* desc; int n; const float* bounds;
    const float* qx; const float* qy; const float* qr; const int* qminL; const int* qmaxL; const uint8_t* qdesc; int nq;
  };
  // Results of the last candidates() call: per job the first query index; per query count/offset; pool.
  std::vector<int> jobQ0;
  const uint32_t* qcount = nullptr;
  const uint32_t* qoff = nullptr;

  DevBuf<uint32_t> d_wpool;      // wide records of the frame searches
  DevBuf<float> d_sf;            // raw queries: mvScaleFactors in device memory (and the host copy they were made from)
  float sfHost[32] = {};
  int sfN = -1;
  int seq = 0;                   // number of the last frame search (k_resolve reports it back through page-locked memory)
  double tEntry = 0, tSynced = 0;   // frame searches: clock at the entry of the C call / when the results were seen
  double stageMs[4] = {0, 0, 0, 0};  // arena build, upload+kernel+download, (resolve: filled by callers), total
  static double nowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  // Results of candidates(), indexed by ORIGINAL query number (all jobs concatenated)
  std::vector<uint32_t> qcountFull, qoffFull;
  std::vector<int> qmap;        // compact (launched) query -> original query
  struct JobPlan { int lo, hi; size_t trainOff, q0c; int nTrain, nqc; };
  std::vector<JobPlan> plan;

  // Runs the window kernel for all jobs in ONE upload + ONE launch.  Only what can influence a result is
  // uploaded: active queries (r >= 0) and the train keypoints whose octave some active query of the job
  // accepts (e.g. SearchForInitialization touches level-0 keypoints only, ORBmatcher.cc:416-420).
  // Dropping the others cannot change any candidate list: Frame::GetFeaturesInArea would skip them.
  int candidates(const Job* jobs, int njobs) {
    const double tA = nowMs();
    HIP_TRY(hipSetDevice(device));
    int rc;
    const int ncell = kGridCols * kGridRows;
    jobQ0.assign(njobs + 1, 0);
    plan.assign(njobs, JobPlan{});
    size_t nTrain = 0, nqOrig = 0, nq = 0;
    for (int j = 0; j < njobs; j++) {
      const Job& J = jobs[j];
      jobQ0[j] = (int)nqOrig;
      nqOrig += J.nq;
      JobPlan& pl = plan[j];
      pl.lo = INT_MAX; pl.hi = INT_MIN; pl.nqc = 0;
      for (int q = 0; q < J.nq; q++) {
        if (!(J.qr[q] >= 0.f)) continue;
        pl.nqc++;
        const bool check = (J.qminL[q] > 0) || (J.qmaxL[q] >= 0);
        const int lo = check ? J.qminL[q] : INT_MIN, hi = (check && J.qmaxL[q] >= 0) ? J.qmaxL[q] : INT_MAX;
        pl.lo = std::min(pl.lo, lo);
        pl.hi = std::max(pl.hi, hi);
      }
      pl.trainOff = nTrain;
      pl.q0c = nq;
      nq += pl.nqc;
      int cnt = 0;
      if (pl.nqc)
        for (int i = 0; i < J.n; i++) cnt += (J.kps[i].octave >= pl.lo && J.kps[i].octave <= pl.hi) ? 1 : 0;
      pl.nTrain = cnt;   // upper bound (keypoints outside the grid are dropped below)
      nTrain += cnt;
    }
    int maxCols = 1;
    for (int j = 0; j < njobs; j++) {
      const Job& J = jobs[j];
      const float invW = static_cast<float>(kGridCols) / static_cast<float>(J.bounds[1] - J.bounds[0]);
      float rmax = 0.f;
      for (int q = 0; q < J.nq; q++) rmax = std::max(rmax, J.qr[q]);
      const float cols = 2.f * rmax * invW + 3.f;
      maxCols = std::max(maxCols, cols >= 64.f ? 64 : (int)std::ceil(cols));
    }
    jobQ0[njobs] = (int)nqOrig;
    qcountFull.assign(nqOrig, 0);
    qoffFull.assign(nqOrig, 0);
    qmap.resize(nq);
    if (nq == 0) {
      qcount = qcountFull.data();
      qoff = qoffFull.data();
      stageMs[0] = nowMs() - tA;
      stageMs[1] = 0;
      return ORBFE_OK;
    }
    // arena layout (one H2D copy); train descriptors are stored in grid-sorted order
    const size_t oSx = 0, oSy = oSx + al(4 * nTrain), oOct = oSy + al(4 * nTrain), oIdx = oOct + al(4 * nTrain),
                 oCell = oIdx + al(4 * nTrain), oTd = oCell + al(4 * (size_t)(ncell + 1) * njobs),
                 oPair = oTd + al(32 * nTrain), oQp = oPair + al(sizeof(PairInfo) * (size_t)njobs),
                 oQx = oQp + al(4 * nq), oQy = oQx + al(4 * nq), oQr = oQy + al(4 * nq), oQa = oQr + al(4 * nq),
                 oQb = oQa + al(4 * nq), oQd = oQb + al(4 * nq), total = oQd + al(32 * nq);
    if ((rc = h_in.ensure(total))) return rc;
    if ((rc = d_in.ensure(total))) return rc;
    uint8_t* H = h_in.p;
    pool->parallelFor(njobs, [&](int j, int) {
      const Job& J = jobs[j];
      const JobPlan& pl = plan[j];
      const float minX = J.bounds[0], maxX = J.bounds[1], minY = J.bounds[2], maxY = J.bounds[3];
      const float invW = static_cast<float>(kGridCols) / static_cast<float>(maxX - minX);   // Frame.cc:98
      const float invH = static_cast<float>(kGridRows) / static_cast<float>(maxY - minY);   // Frame.cc:99
      int* cellCnt = (int*)(H + oCell) + (size_t)(ncell + 1) * j;
      for (int c = 0; c <= ncell; c++) cellCnt[c] = 0;
      PairInfo pi;
      pi.trainOff = (int)pl.trainOff; pi.cellOff = (ncell + 1) * j; pi.tdescOff = (int)pl.trainOff;
      pi.minX = minX; pi.minY = minY; pi.invW = invW; pi.invH = invH;
      ((PairInfo*)(H + oPair))[j] = pi;
      if (pl.nqc == 0) return;
      // AssignFeaturesToGrid / PosInGrid (Frame.cc:114-129, 264-274) as a stable counting sort by cell
      std::vector<int> cellOf(J.n);
      for (int i = 0; i < J.n; i++) {
        cellOf[i] = -1;
        if (J.kps[i].octave < pl.lo || J.kps[i].octave > pl.hi) continue;
        const int px = (int)roundf((J.kps[i].x - minX) * invW);
        const int py = (int)roundf((J.kps[i].y - minY) * invH);
        if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
        cellOf[i] = px * kGridRows + py;
        cellCnt[cellOf[i] + 1]++;
      }
      for (int c = 0; c < ncell; c++) cellCnt[c + 1] += cellCnt[c];
      std::vector<int> order(cellCnt, cellCnt + ncell);
      float* sx = (float*)(H + oSx) + pl.trainOff;
      float* sy = (float*)(H + oSy) + pl.trainOff;
      int* so = (int*)(H + oOct) + pl.trainOff;
      int* si = (int*)(H + oIdx) + pl.trainOff;
      uint8_t* td = H + oTd + 32 * pl.trainOff;
      for (int i = 0; i < J.n; i++) {
        if (cellOf[i] < 0) continue;
        const int p = order[cellOf[i]]++;
        sx[p] = J.kps[i].x; sy[p] = J.kps[i].y; so[p] = J.kps[i].octave; si[p] = i;
        memcpy(td + 32 * (size_t)p, J.desc + 32 * (size_t)i, 32);
      }
      // active queries, compacted
      int* qp = (int*)(H + oQp) + pl.q0c;
      float* qxo = (float*)(H + oQx) + pl.q0c;
      float* qyo = (float*)(H + oQy) + pl.q0c;
      float* qro = (float*)(H + oQr) + pl.q0c;
      int* qao = (int*)(H + oQa) + pl.q0c;
      int* qbo = (int*)(H + oQb) + pl.q0c;
      uint8_t* qdo = H + oQd + 32 * pl.q0c;
      int c = 0;
      for (int q = 0; q < J.nq; q++) {
        if (!(J.qr[q] >= 0.f)) continue;
        qp[c] = j; qxo[c] = J.qx[q]; qyo[c] = J.qy[q]; qro[c] = J.qr[q]; qao[c] = J.qminL[q]; qbo[c] = J.qmaxL[q];
        memcpy(qdo + 32 * (size_t)c, J.qdesc + 32 * (size_t)q, 32);
        qmap[pl.q0c + c] = jobQ0[j] + q;
        c++;
      }
    });
    const double tB = nowMs();
    stageMs[0] = tB - tA;
    // (experiments build, ORBFE_MATCH_ZEROCOPY=1: the kernel reads the pinned host arena directly over PCIe instead of a DMA upload)
    static const bool zeroCopy = ORBFE_EXP_ENV("ORBFE_MATCH_ZEROCOPY") && atoi(ORBFE_EXP_ENV("ORBFE_MATCH_ZEROCOPY")) != 0;
    if (!zeroCopy) HIP_TRY(hipMemcpyAsync(d_in.p, H, total, hipMemcpyHostToDevice, stream));

    const size_t outWords = 64 + 2 * nq;
    if ((rc = d_out.ensure(outWords))) return rc;
    if ((rc = h_out.ensure(outWords))) return rc;
    size_t poolCap = d_pool.n ? d_pool.n : std::max<size_t>(1 << 16, nq * 32);
    for (int attempt = 0; attempt < 2; attempt++) {
      if ((rc = d_pool.ensure(poolCap))) return rc;
      HIP_TRY(hipMemsetAsync(d_out.p, 0, 64 * sizeof(uint32_t), stream));
      MatchParams M;
      uint8_t* D = zeroCopy ? H : d_in.p;
      M.sx = (const float*)(D + oSx); M.sy = (const float*)(D + oSy); M.soct = (const int*)(D + oOct);
      M.sidx = (const int*)(D + oIdx); M.cellStart = (const int*)(D + oCell); M.tdesc = D + oTd;
      M.pairs = (const PairInfo*)(D + oPair); M.qpair = (const int*)(D + oQp);
      M.qx = (const float*)(D + oQx); M.qy = (const float*)(D + oQy); M.qr = (const float*)(D + oQr);
      M.qminL = (const int*)(D + oQa); M.qmaxL = (const int*)(D + oQb); M.qdesc = D + oQd;
      M.nq = (int)nq;
      M.total = d_out.p; M.qcount = d_out.p + 64; M.qoff = d_out.p + 64 + nq;
      M.pool = d_pool.p; M.poolCap = (uint32_t)d_pool.n;
      {
        // widest window in grid columns over all active queries (+3: floor/ceil slack of the cell range)
        int lpq = 8;
        while (lpq < 64 && lpq < maxCols) lpq <<= 1;
        const unsigned nblk = (unsigned)((nq + (kWinThreads / lpq) - 1) / (kWinThreads / lpq));
        if (lpq == 8) hipLaunchKernelGGL(k_window_match<8>, dim3(nblk), dim3(kWinThreads), 0, stream, M);
        else if (lpq == 16) hipLaunchKernelGGL(k_window_match<16>, dim3(nblk), dim3(kWinThreads), 0, stream, M);
        else if (lpq == 32) hipLaunchKernelGGL(k_window_match<32>, dim3(nblk), dim3(kWinThreads), 0, stream, M);
        else hipLaunchKernelGGL(k_window_match<64>, dim3(nblk), dim3(kWinThreads), 0, stream, M);
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(h_out.p, d_out.p, outWords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      // optimistic: fetch a generous prefix of the pool in the same round trip
      const size_t guess = std::min<size_t>(d_pool.n, std::max<size_t>(lastTotal + lastTotal / 4 + 1024, 4096));
      if ((rc = h_pool.ensure(guess + 1))) return rc;
      HIP_TRY(hipMemcpyAsync(h_pool.p, d_pool.p, guess * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      const size_t tot = h_out.p[0];
      if (tot <= d_pool.n) {
        if (tot > guess) {
          if ((rc = h_pool.ensure(tot + 1))) return rc;
          HIP_TRY(hipMemcpyAsync(h_pool.p, d_pool.p, tot * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
          HIP_TRY(hipStreamSynchronize(stream));
        }
        lastTotal = tot;
        stageMs[1] = nowMs() - tB;
        const uint32_t* qc = h_out.p + 64;
        const uint32_t* qo = h_out.p + 64 + nq;
        for (size_t c = 0; c < nq; c++) {
          qcountFull[qmap[c]] = qc[c];
          qoffFull[qmap[c]] = qo[c];
        }
        qcount = qcountFull.data();
        qoff = qoffFull.data();
        return ORBFE_OK;
      }
      poolCap = tot;  // pool too small: grow to the exact demand and rerun once
    }
    set_err("candidate pool sizing failed");
    return ORBFE_ERR_HIP;
  }

  int candidates(const OrbfeKeyPoint* kps, const uint8_t* desc, int n, const float bounds[4], const float* qx,
                 const float* qy, const float* qr, const int* qminL, const int* qmaxL, const uint8_t* qdesc, int nq) {
    Job j{kps, desc, n, bounds, qx, qy, qr, qminL, qmaxL, qdesc, nq};
    return candidates(&j, 1);
  }
  size_t lastTotal = 0;
  std::unique_ptr<orbfe::HostPool> pool;
};
