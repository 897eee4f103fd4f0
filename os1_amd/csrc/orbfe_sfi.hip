// orbfe_sfi.hip -- GPU-resident ORBmatcher::SearchForInitialization for consecutive frames of a stream.
//
// Behaviour contract: ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:400-515) with
// vbPrevMatched initialised to F1's keypoints (Tracking.cc:355-357), F1 = the frame's predecessor in the
// stream, F2 = the frame; Frame grid semantics of Frame.cc:98-99,114-129,209-274.  Everything it reads is
// already in HBM (the extractor's result arena: level-0 selection slots, angles, descriptors), so a
// streamed frame is matched without its descriptors ever being re-uploaded.
//
// Only level-0 keypoints take part (queries must have octave 0, ORBmatcher.cc:416-418; the window query
// is restricted to levels [0,0], :420), and level 0 is the first region of the output order, so
// "keypoint index" == "level-0 slot index" for every index this search produces.
//
// Three kernels per batch, all on the extractor's stream:
//   k_sfi_sort        per frame (as F2): grid cell of every level-0 keypoint (PosInGrid, round()), stable rank
//                     by (cell, index)  == the order GetFeaturesInArea enumerates candidates in
//   k_sfi_candidates  one wave per (pair, query): scan F2 in that order, box test, Hamming -> ordered list
//   k_sfi_resolve     one block per pair: the reference's sequential bookkeeping (vMatchedDistance, vnMatches21,
//                     steal-back, rotation histogram, ComputeThreeMaxima) as a fixed point -- see the kernel
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "orbfe_internal.h"

namespace orbfe {

constexpr int kSfiGridCols = 64, kSfiGridRows = 48;   // FRAME_GRID_COLS / ROWS (Frame.h:36-37)
constexpr int kSfiThLow = 50, kSfiHisto = 30;        // TH_LOW, HISTO_LENGTH (ORBmatcher.cc:38-39)

// level-0 keypoint data of frame f (or of the carried predecessor)
struct SfiFrame {
  const SelKp* sel;
  const float* angle;
  const uint8_t* desc;
  int n;
};

__device__ __forceinline__ SfiFrame sfi_frame(const SfiParams& S, int f) {
  SfiFrame F;
  if (f < 0) {
    F.sel = S.carrySel; F.angle = S.carryAngle; F.desc = S.carryDesc;
    F.n = S.carryCount ? (int)*S.carryCount : -1;
    if (F.n > S.n0cap) F.n = -1;   // 0xffffffff = no predecessor yet
  } else {
    const long long base = (long long)f * S.selPerFrame;   // level 0 is the first region of a frame's slots
    F.sel = S.sel + base; F.angle = S.angle + base; F.desc = S.desc + base * 32;
    F.n = (int)S.selCount[(long long)f * kMaxLevels];
  }
  return F;
}

// ---- per frame: candidate enumeration order -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sfi_sort(SfiParams S) {
  // counting sort by grid cell; inside a cell by keypoint index (the order mGrid's vectors were filled in)
  constexpr int kCells = kSfiGridCols * kSfiGridRows, kPer = kCells / 256;
  static_assert(kCells % 256 == 0, "cells per thread");
  extern __shared__ int dyn[];
  int* cellOf = dyn;                 // [n0cap]
  int* arrival = dyn + S.n0cap;      // [n0cap] position among the keypoints of the same cell, in arrival order
  int* bucket = dyn + 2 * S.n0cap;   // [n0cap] keypoints grouped by cell
  __shared__ int start[kCells + 1];
  __shared__ int wsum[4];
  const int f = S.frameBase + blockIdx.x, tid = threadIdx.x;
  const SfiFrame F = sfi_frame(S, f);
  for (int c = tid; c <= kCells; c += 256) start[c] = 0;
  __syncthreads();
  for (int k = tid; k < F.n; k += 256) {
    const uint32_t xy = F.sel[k].xy;
    const float x = (float)(xy & 0xffff), y = (float)(xy >> 16);
    const int px = (int)roundf((x - S.minX) * S.invW), py = (int)roundf((y - S.minY) * S.invH);   // Frame.cc:266-267
    const int c = (px < 0 || px >= kSfiGridCols || py < 0 || py >= kSfiGridRows) ? -1 : px * kSfiGridRows + py;
    cellOf[k] = c;
    if (c >= 0) arrival[k] = atomicAdd(&start[c], 1);
  }
  __syncthreads();
  // exclusive prefix over the cells: kPer consecutive cells per thread, then a block scan of the thread sums
  int loc[kPer], sum = 0;
#pragma unroll
  for (int i = 0; i < kPer; i++) { loc[i] = start[tid * kPer + i]; sum += loc[i]; }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if ((tid & 63) >= o) incl += t;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
#pragma unroll
  for (int i = 0; i < kPer; i++) { start[tid * kPer + i] = base; base += loc[i]; }
  if (tid == 255) start[kCells] = base;
  __syncthreads();
  for (int k = tid; k < F.n; k += 256) {
    const int c = cellOf[k];
    if (c >= 0) bucket[start[c] + arrival[k]] = k;
  }
  __syncthreads();
  uint16_t* order = S.order + (long long)f * S.n0cap;
  for (int k = tid; k < F.n; k += 256) {
    const int c = cellOf[k];
    if (c < 0) continue;
    const int b0 = start[c], b1 = start[c + 1];
    int rank = 0;
    for (int j = b0; j < b1; j++) rank += bucket[j] < k;
    order[b0 + rank] = (uint16_t)k;
  }
  if (tid == 0) S.orderCount[f] = start[kCells];
}

// ---- per (pair, query): ordered candidate list with distances -------------------------------------------------
__global__ __launch_bounds__(64) void k_sfi_candidates(SfiParams S) {
  const int i1 = blockIdx.x, fr = blockIdx.y, lane = threadIdx.x;
  const int f = S.frameBase + fr;
  const SfiFrame F2 = sfi_frame(S, f);
  const SfiFrame F1 = sfi_frame(S, fr == 0 ? -1 : f - 1);
  uint32_t* cnt = S.pcount + (long long)f * S.n0cap + i1;
  if (F1.n < 0 || i1 >= F1.n) {
    if (lane == 0 && i1 < S.n0cap) *cnt = 0;
    return;
  }
  const uint32_t qxy = F1.sel[i1].xy;
  const float x = (float)(qxy & 0xffff), y = (float)(qxy >> 16), r = S.window;
  uint32_t qd[8];
  const uint32_t* qp = reinterpret_cast<const uint32_t*>(F1.desc + (long long)i1 * 32);
#pragma unroll
  for (int w = 0; w < 8; w++) qd[w] = qp[w];
  const uint16_t* order = S.order + (long long)f * S.n0cap;
  const int nin = S.orderCount[f];
  uint32_t* pool = S.pool + ((long long)f * S.n0cap + i1) * S.n0cap;
  const unsigned long long below = (1ull << lane) - 1ull;
  int run = 0;
  for (int e0 = 0; e0 < nin; e0 += 64) {
    const int e = e0 + lane;
    bool ok = false;
    int i2 = 0;
    if (e < nin) {
      i2 = order[e];
      const uint32_t xy = F2.sel[i2].xy;
      const float dx = (float)(xy & 0xffff) - x, dy = (float)(xy >> 16) - y;
      ok = fabsf(dx) < r && fabsf(dy) < r;   // Frame.cc:252-256 (levels are 0 by construction)
    }
    const unsigned long long m = __ballot(ok);
    if (ok) {
      const uint32_t* dp = reinterpret_cast<const uint32_t*>(F2.desc + (long long)i2 * 32);
      int d = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) d += __popc(dp[w] ^ qd[w]);
      pool[run + __popcll(m & below)] = (uint32_t)i2 | ((uint32_t)d << 16);
    }
    run += __popcll(m);
  }
  if (lane == 0) *cnt = (uint32_t)run;
}

// ---- per pair: the sequential bookkeeping ---------------------------------------------------------------------
__device__ __forceinline__ int sfi_rot_bin(float a1, float a2) {   // ORBmatcher.cc:470-475
  const float factor = 1.0f / kSfiHisto;
  float rot = a1 - a2;
  if (rot < 0.0f) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == kSfiHisto) bin = 0;
  return bin;
}

__global__ __launch_bounds__(64) void k_sfi_resolve_seq(SfiParams S) {
  extern __shared__ int sm[];
  const int fr = blockIdx.x, lane = threadIdx.x;
  const int f = S.frameBase + fr;
  const SfiFrame F2 = sfi_frame(S, f);
  const SfiFrame F1 = sfi_frame(S, fr == 0 ? -1 : f - 1);
  int32_t* out = S.matches12 + (long long)f * S.n0cap;
  if (F1.n < 0) {   // no predecessor: first frame of the stream
    for (int i = lane; i < S.n0cap; i += 64) out[i] = -1;
    if (lane == 0) S.nmatches[f] = 0;
    return;
  }
  const int n1 = F1.n, n2 = F2.n, cap = S.n0cap;
  int* vMatchedDistance = sm;            // [cap]
  int* vnMatches21 = sm + cap;           // [cap]
  int* vnMatches12 = sm + 2 * cap;       // [cap]
  int* binOf = sm + 3 * cap;             // [cap] rotation bin of a pushed query, -1 = never pushed
  int* pcnt = sm + 4 * cap;              // [cap] candidate counts of the queries
  int* hist = sm + 5 * cap;              // [32]
  for (int i = lane; i < cap; i += 64) {
    vMatchedDistance[i] = 0x7fffffff;
    vnMatches21[i] = -1;
    vnMatches12[i] = -1;
    binOf[i] = -1;
    pcnt[i] = i < n1 ? (int)S.pcount[(long long)f * cap + i] : 0;
  }
  if (lane < 32) hist[lane] = 0;
  __syncthreads();
  const uint32_t* pool = S.pool + (long long)f * cap * cap;
  int nm = 0;
  // software pipeline: the first chunk of query i1+1 is fetched while query i1 is resolved
  uint32_t nextEntry = 0;
  if (n1 > 0 && lane < pcnt[0]) nextEntry = pool[lane];
  for (int i1 = 0; i1 < n1; i1++) {
    const int cnt = pcnt[i1];
    uint32_t entry = nextEntry;
    if (i1 + 1 < n1 && lane < pcnt[i1 + 1]) nextEntry = pool[(long long)(i1 + 1) * cap + lane];
    if (cnt == 0) continue;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      if (c0 > 0) entry = (c0 + lane < cnt) ? pool[(long long)i1 * cap + c0 + lane] : 0u;
      int d = 0x7fffffff, i2 = -1;
      if (c0 + lane < cnt) {
        i2 = (int)(entry & 0xffff);
        const int dist = (int)(entry >> 16);
        if (!(vMatchedDistance[i2] <= dist)) d = dist;   // :439 candidate owned by an equal-or-better match
      }
      // Chunk minimum / second minimum by bitwise bisection with ballots (distances fit 9 bits; 511 = skipped):
      // after the loop `c1` holds exactly the lanes with the smallest distance; its lowest lane is the first
      // position (candidate order = lane order).  Scalar mask arithmetic instead of 19 dependent cross-lane
      // permutes per query.
      const unsigned dd = (d == 0x7fffffff) ? 511u : (unsigned)d;
      unsigned long long c1 = ~0ull;
#pragma unroll
      for (int b = 8; b >= 0; b--) {
        const unsigned long long z = __ballot(((dd >> b) & 1u) == 0u) & c1;
        if (z) c1 = z;
      }
      const int ml = (int)__builtin_ctzll(c1);
      const unsigned d1 = (unsigned)__shfl((int)dd, ml, 64);
      unsigned long long c2 = ~(1ull << ml);   // everyone but the winner
#pragma unroll
      for (int b = 8; b >= 0; b--) {
        const unsigned long long z = __ballot(((dd >> b) & 1u) == 0u) & c2;
        if (z) c2 = z;
      }
      const unsigned d2 = (unsigned)__shfl((int)dd, (int)__builtin_ctzll(c2), 64);
      const int md = d1 >= 511u ? 0x7fffffff : (int)d1;
      const int sd = d2 >= 511u ? 0x7fffffff : (int)d2;
      const int mi2 = __shfl(i2, ml, 64);
      // merge with the running (best, second) -- sequential semantics of :441-450, earlier chunk wins ties
      if (md < bestDist) {
        bestDist2 = min(bestDist, sd);
        bestDist = md;
        bestIdx2 = mi2;
      } else {
        bestDist2 = min(bestDist2, md);
      }
    }
    if (bestDist <= kSfiThLow && (float)bestDist < (float)bestDist2 * S.nnratio) {
      if (lane == 0) {
        const int old = vnMatches21[bestIdx2];
        if (old >= 0) vnMatches12[old] = -1;
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        if (S.checkOri) {
          const int bin = sfi_rot_bin(F1.angle[i1], F2.angle[bestIdx2]);
          binOf[i1] = bin;
          hist[bin]++;
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (S.checkOri) {
    // ComputeThreeMaxima, ORBmatcher.cc:1554-1595 (every lane evaluates it identically)
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < kSfiHisto; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int i = lane; i < n1; i += 64) {
      const int b = binOf[i];
      if (b >= 0 && b != ind1 && b != ind2 && b != ind3 && vnMatches12[i] >= 0) vnMatches12[i] = -1;
    }
    __syncthreads();
  }
  for (int i = lane; i < cap; i += 64) {
    const int v = i < n1 ? vnMatches12[i] : -1;
    out[i] = v;
    nm += v >= 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nm += __shfl_xor(nm, o, 64);
  if (lane == 0) S.nmatches[f] = nm;
  (void)n2;
}

// The same bookkeeping without the serial walk.  Query i1's outcome (its accepted match, or none) is a function of
// its own candidate list and of vMatchedDistance at the moment the reference reaches i1, and vMatchedDistance[i2] at
// that moment is the smallest distance among the matches accepted for i2 by queries j < i1 (:439, :455-466: every
// accepted match lowers it).  So the vector of outcomes M satisfies M[i1] = g(i1, {M[j] : j < i1}); that recurrence
// has exactly one solution (induction over i1) and the reference's loop computes it.  The kernel iterates
// M <- g(M) from "no matches" with one thread per query until nothing changes: round k fixes at least queries
// 0..k-1, in practice the dependency chains (a match taken away from a later query whose second choice then takes a
// keypoint from a still later one ...) are a handful long.  Afterwards vnMatches12[j] survives iff j is the last
// query that took its keypoint (steal-back, :457-461) and the rotation histogram holds every accepted match, stolen
// or not (:468-478 push without ever removing).
// (threads per pair: a template parameter of k_sfi_resolve -- 512 = one query per thread at 1080p / 2000 features, launch_sfi)

// measurement only (ORBFE_SFI_DEBUG=1, tools/sfi_rounds.py): one record per resolved pair -- {frame, rounds of the fixed point,
// candidate entries, 1 if the serial finish ran, shader cycles of the block}; [0] of the buffer counts the records.  One buffer per
// process (a device symbol), like the FAST phase stamps.
__device__ int* g_sfiDbg;
constexpr int kSfiDbgRecords = 1 << 16;

// Serial replay of :430-466 by ONE wave, lanes over a query's candidates: the outcome M[i1] (i2 | dist << 16, -1 = none)
// of every query in order, vMatchedDistance in `vmd`.  k_sfi_resolve falls back to it when the fixed point has not
// settled after `maxRounds` rounds (adversarial inputs can need up to n1 + 1), so the worst case is one serial pass, as
// it was before the fixed point existed.
template <class PoolAt>
__device__ void sfi_serial_outcomes(int lane, int n1, const int* pcnt, PoolAt poolAt, float nnratio, int* vmd, int* M) {
  for (int i1 = 0; i1 < n1; i1++) {
    const int cnt = pcnt[i1];
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      int d = 0x7fffffff, i2 = -1;
      if (c0 + lane < cnt) {
        const uint32_t entry = poolAt(i1, c0 + lane);
        i2 = (int)(entry & 0xffff);
        const int dist = (int)(entry >> 16);
        if (!(vmd[i2] <= dist)) d = dist;   // :439
      }
      const unsigned dd = (d == 0x7fffffff) ? 511u : (unsigned)d;
      unsigned long long c1 = ~0ull;
#pragma unroll
      for (int b = 8; b >= 0; b--) {
        const unsigned long long z = __ballot(((dd >> b) & 1u) == 0u) & c1;
        if (z) c1 = z;
      }
      const int ml = (int)__builtin_ctzll(c1);
      const unsigned d1 = (unsigned)__shfl((int)dd, ml, 64);
      unsigned long long c2 = ~(1ull << ml);
#pragma unroll
      for (int b = 8; b >= 0; b--) {
        const unsigned long long z = __ballot(((dd >> b) & 1u) == 0u) & c2;
        if (z) c2 = z;
      }
      const unsigned d2 = (unsigned)__shfl((int)dd, (int)__builtin_ctzll(c2), 64);
      const int md = d1 >= 511u ? 0x7fffffff : (int)d1;
      const int sd = d2 >= 511u ? 0x7fffffff : (int)d2;
      const int mi2 = __shfl(i2, ml, 64);
      if (md < bestDist) { bestDist2 = min(bestDist, sd); bestDist = md; bestIdx2 = mi2; }   // :441-450, earlier chunk wins ties
      else bestDist2 = min(bestDist2, md);
    }
    int m = -1;
    if (bestDist <= kSfiThLow && (float)bestDist < (float)bestDist2 * nnratio) {
      m = bestIdx2 | (bestDist << 16);
      if (lane == 0) vmd[bestIdx2] = bestDist;
    }
    if (lane == 0) M[i1] = m;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // lane 0's LDS stores land before the wave's next reads
  }
}

template <int kSfiThreads>
__global__ __launch_bounds__(kSfiThreads) void k_sfi_resolve(SfiParams S, int ldsPool, int maxRounds) {
  extern __shared__ int sm[];
  const int fr = blockIdx.x, tid = threadIdx.x;
  const unsigned long long tStart = g_sfiDbg ? __builtin_amdgcn_s_memtime() : 0ull;
  int dbgSerial = 0;
  const int f = S.frameBase + fr;
  const SfiFrame F2 = sfi_frame(S, f);
  const SfiFrame F1 = sfi_frame(S, fr == 0 ? -1 : f - 1);
  int32_t* out = S.matches12 + (long long)f * S.n0cap;
  if (F1.n < 0) {   // no predecessor: first frame of the stream
    for (int i = tid; i < S.n0cap; i += kSfiThreads) out[i] = -1;
    if (tid == 0) S.nmatches[f] = 0;
    return;
  }
  const int n1 = F1.n, n2 = F2.n, cap = S.n0cap;
  int* Mbuf[2] = {sm, sm + cap};         // outcome of a query: i2 | dist << 16, -1 = none
  int* head = sm + 2 * cap;              // [cap] per F2 keypoint: first query of its list of takers
  int* next = sm + 3 * cap;              // [cap] per query: next taker of the same keypoint
  int* off = sm + 4 * cap;               // [cap + 1] start of a query's candidates in the compacted pool
  int* pcnt = sm + 5 * cap + 1;          // [cap]
  int* hist = sm + 6 * cap + 1;          // [32]
  int* misc = hist + 32;                 // [2] nmatches, unused
  uint32_t* lpool = reinterpret_cast<uint32_t*>(misc + 2);   // [ldsPool]
  for (int i = tid; i < cap; i += kSfiThreads) {
    Mbuf[0][i] = -1;
    pcnt[i] = i < n1 ? (int)S.pcount[(long long)f * cap + i] : 0;
  }
  if (tid < 32) hist[tid] = 0;
  if (tid == 32) misc[0] = 0;
  __syncthreads();
  if (tid < 64) {   // exclusive prefix of the candidate counts
    int run = 0;
    for (int b = 0; b < cap; b += 64) {
      const int v = b + tid < cap ? pcnt[b + tid] : 0;
      int s = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(s, o, 64);
        if (tid >= o) s += t;
      }
      if (b + tid < cap) off[b + tid] = run + s - v;
      run += __shfl(s, 63, 64);
    }
    if (tid == 0) off[cap] = run;
  }
  __syncthreads();
  const uint32_t* gpool = S.pool + (long long)f * cap * cap;
  const int total = off[cap];
  const bool inLds = total <= ldsPool;
  if (inLds) {
    for (int e = tid; e < total; e += kSfiThreads) {
      int lo = 0, hi = n1 - 1;   // last query whose offset is <= e
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid - 1;
      }
      lpool[e] = gpool[(long long)lo * cap + (e - off[lo])];
    }
  }
  __syncthreads();
  int cur = 0, rounds = 0;
  for (;;) {
    const int* M = Mbuf[cur];
    int* Mn = Mbuf[cur ^ 1];
    for (int i = tid; i < n2; i += kSfiThreads) head[i] = -1;
    __syncthreads();
    for (int j = tid; j < n1; j += kSfiThreads) {
      const int m = M[j];
      if (m >= 0) next[j] = atomicExch(&head[m & 0xffff], j);
    }
    __syncthreads();
    int changed = 0;
    for (int i1 = tid; i1 < n1; i1 += kSfiThreads) {
      const int cnt = pcnt[i1];
      int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
      for (int c = 0; c < cnt; c++) {
        const uint32_t entry = inLds ? lpool[off[i1] + c] : gpool[(long long)i1 * cap + c];
        const int i2 = (int)(entry & 0xffff), dist = (int)(entry >> 16);
        int vmd = 0x7fffffff;   // vMatchedDistance[i2] as the reference sees it when it reaches i1
        for (int j = head[i2]; j >= 0; j = next[j])
          if (j < i1) vmd = min(vmd, M[j] >> 16);
        if (vmd <= dist) continue;                                           // :439
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }   // :441-450
        else if (dist < bestDist2) bestDist2 = dist;
      }
      int m = -1;
      if (bestDist <= kSfiThLow && (float)bestDist < (float)bestDist2 * S.nnratio) m = bestIdx2 | (bestDist << 16);
      Mn[i1] = m;
      changed |= m != M[i1];
    }
    const int any = __syncthreads_or(changed);
    cur ^= 1;
    if (!any) break;
    if (++rounds >= maxRounds) {   // long steal chains: finish with one serial pass, then rebuild the taker lists from it
      dbgSerial = 1;
      int* M2 = Mbuf[cur];
      int* vmd = head;             // [cap] >= n2 entries, lists are rebuilt below
      for (int i = tid; i < n2; i += kSfiThreads) vmd[i] = 0x7fffffff;
      __syncthreads();
      if (tid < 64) {
        if (inLds) sfi_serial_outcomes(tid, n1, pcnt, [&](int q, int c) { return lpool[off[q] + c]; }, S.nnratio, vmd, M2);
        else sfi_serial_outcomes(tid, n1, pcnt, [&](int q, int c) { return gpool[(long long)q * cap + c]; }, S.nnratio, vmd, M2);
      }
      __syncthreads();
      for (int i = tid; i < n2; i += kSfiThreads) head[i] = -1;
      __syncthreads();
      for (int j = tid; j < n1; j += kSfiThreads) {
        const int m = M2[j];
        if (m >= 0) next[j] = atomicExch(&head[m & 0xffff], j);
      }
      __syncthreads();
      break;
    }
  }
  // the lists were built from the outcomes that just reproduced themselves
  const int* M = Mbuf[cur];
  int* vn12 = Mbuf[cur ^ 1];
  int* binOf = off;   // offsets are no longer needed (cap + 1 >= n1)
  __syncthreads();
  for (int j = tid; j < n1; j += kSfiThreads) {
    const int m = M[j];
    int v = -1, bin = -1;
    if (m >= 0) {
      const int i2 = m & 0xffff;
      bool last = true;
      for (int k = head[i2]; k >= 0; k = next[k]) last = last && k <= j;
      if (last) v = i2;
      if (S.checkOri) {
        bin = sfi_rot_bin(F1.angle[j], F2.angle[i2]);
        atomicAdd(&hist[bin], 1);
      }
    }
    vn12[j] = v;
    binOf[j] = bin;
  }
  __syncthreads();
  int ind1 = -1, ind2 = -1, ind3 = -1;
  if (S.checkOri) {
    // ComputeThreeMaxima, ORBmatcher.cc:1554-1595 (every thread evaluates it identically)
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < kSfiHisto; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
  }
  int nm = 0;
  for (int i = tid; i < cap; i += kSfiThreads) {
    int v = -1;
    if (i < n1) {
      v = vn12[i];
      const int b = binOf[i];
      if (S.checkOri && b >= 0 && b != ind1 && b != ind2 && b != ind3) v = -1;
    }
    out[i] = v;
    nm += v >= 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nm += __shfl_xor(nm, o, 64);
  if ((tid & 63) == 0 && nm) atomicAdd(&misc[0], nm);
  __syncthreads();
  if (tid == 0) S.nmatches[f] = misc[0];
  if (g_sfiDbg && tid == 0) {
    const int r = atomicAdd(&g_sfiDbg[0], 1);
    if (r < kSfiDbgRecords) {
      int* rec = g_sfiDbg + 8 + 8 * r;
      rec[0] = f; rec[1] = rounds; rec[2] = total; rec[3] = dbgSerial;
      rec[4] = (int)(__builtin_amdgcn_s_memtime() - tStart); rec[5] = n1; rec[6] = n2; rec[7] = inLds;
    }
  }
}

static int* s_sfiDbg = nullptr;
static void sfi_debug_setup() {
  static bool done = false;
  if (done) return;
  done = true;
  if (!ORBFE_EXP_ENV("ORBFE_SFI_DEBUG")) return;
  if (hipMalloc((void**)&s_sfiDbg, sizeof(int) * (8 + 8 * kSfiDbgRecords)) != hipSuccess) { s_sfiDbg = nullptr; return; }
  (void)hipMemset(s_sfiDbg, 0, sizeof(int) * (8 + 8 * kSfiDbgRecords));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sfiDbg), &s_sfiDbg, sizeof s_sfiDbg);
}
// -> records copied (8 ints each: frame, rounds, candidate entries, serial finish, cycles, n1, n2, pool in LDS); reset clears the buffer
int sfi_debug_read(int* out, int capRecords, int reset) {
  if (!s_sfiDbg) return 0;
  int n = 0;
  if (hipMemcpy(&n, s_sfiDbg, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  n = std::min(std::min(n, kSfiDbgRecords), capRecords);
  if (n > 0 && hipMemcpy(out, s_sfiDbg + 8, sizeof(int) * 8 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset) (void)hipMemset(s_sfiDbg, 0, sizeof(int));
  return n;
}

// Hand the level-0 data of the batch's last frame to the next batch (one small kernel instead of four D2D copies).
__global__ __launch_bounds__(256) void k_sfi_carry(SfiParams S, int lastFrame, SelKp* cSel, float* cAngle, uint8_t* cDesc,
                                                   uint32_t* cCount) {
  const long long base = (long long)lastFrame * S.selPerFrame;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < S.n0cap) {
    cSel[i] = S.sel[base + i];
    cAngle[i] = S.angle[base + i];
  }
  if (i < S.n0cap * 8)
    reinterpret_cast<uint32_t*>(cDesc)[i] = reinterpret_cast<const uint32_t*>(S.desc + base * 32)[i];
  if (i == 0) *cCount = S.selCount[(long long)lastFrame * kMaxLevels];
}

void launch_sfi_carry(const SfiParams& S, int lastFrame, SelKp* cSel, float* cAngle, uint8_t* cDesc, uint32_t* cCount,
                      hipStream_t st) {
  hipLaunchKernelGGL(k_sfi_carry, dim3((S.n0cap * 8 + 255) / 256), dim3(256), 0, st, S, lastFrame, cSel, cAngle, cDesc, cCount);
}

void launch_sfi(const SfiParams& S, int nframes, hipStream_t st) {
  sfi_debug_setup();
  hipLaunchKernelGGL(k_sfi_sort, dim3(nframes), dim3(256), sizeof(int) * 3 * S.n0cap, st, S);
  hipLaunchKernelGGL(k_sfi_candidates, dim3(S.n0cap, nframes), dim3(64), 0, st, S);
  const bool seq = getenv("ORBFE_SFI_SEQUENTIAL") != nullptr;   // the serial replay, kept for A/B runs and tests
  if (seq) {
    hipLaunchKernelGGL(k_sfi_resolve_seq, dim3(nframes), dim3(64), sizeof(int) * (5 * S.n0cap + 32), st, S);
    return;
  }
  const int fixedWords = 6 * S.n0cap + 1 + 32 + 2;
  int ldsPool = (60 * 1024 / 4) - fixedWords;   // candidate entries kept in LDS; longer pools are read from HBM
  if (ldsPool > 8192) ldsPool = 8192;
  if (const char* e = ORBFE_EXP_ENV("ORBFE_SFI_LDS_POOL")) ldsPool = std::min(ldsPool, std::max(0, atoi(e)));
  if (ldsPool < 0) ldsPool = 0;
  // rounds of the fixed point before the kernel finishes with one serial pass on the device (typical inputs settle in
  // 3-6 rounds; a round costs O(queries x candidates x takers), so a cap keeps adversarial steal chains bounded)
  int maxRounds = 32;
  if (const char* e = getenv("ORBFE_SFI_MAX_ROUNDS")) maxRounds = atoi(e) < 1 ? 1 : atoi(e);
  static const int threads = [] { const char* e = ORBFE_EXP_ENV("ORBFE_SFI_THREADS"); return e ? atoi(e) : 512; }();
  if (threads == 256)
    hipLaunchKernelGGL(k_sfi_resolve<256>, dim3(nframes), dim3(256), sizeof(int) * (fixedWords + ldsPool), st, S, ldsPool, maxRounds);
  else
    hipLaunchKernelGGL(k_sfi_resolve<512>, dim3(nframes), dim3(512), sizeof(int) * (fixedWords + ldsPool), st, S, ldsPool, maxRounds);
}

}  // namespace orbfe
