// orbfe_frame.hip -- a Frame's features resident on the GPU, and the searches that run on them without leaving it.
//
// The reference builds a frame's grid once (Frame::AssignFeaturesToGrid, src/Frame.cc:114-129, called from the
// constructor :111) and every search of that frame reuses it -- 2-3 searches per tracked frame
// (src/Tracking.cc:608, 614, 824).  An orbfe_frame is that object on the device: undistorted keypoints (x, y, octave,
// angle), descriptors, and the cell-sorted table GetFeaturesInArea (Frame.cc:209-262) walks, built ONCE -- from host
// arrays (one upload) or straight from the extractor's result arena, where the features were produced microseconds
// earlier (orbfe_frame_create_from_extract: no descriptor ever crosses PCIe again).
//
// A search on a resident frame is one stream submission of two kernels, without a copy command: k_window_match reads the
// caller's query arrays where they are (page-locked or device memory; ordinary memory is first memcpy'd into a page-locked
// arena) and leaves the candidate lists in reference order as packed records -> k_resolve (the reference's sequential
// bookkeeping) writes the result vector and, last of all, the call's number into page-locked memory, which the host polls.
// The bookkeeping of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:45-124), the Frame / KeyFrame projection
// searches (:1292-1552) and the projected best-match loops (:357-392, :872-936, :1014-1050, :1066-1290) is sequential
// only through ONE mechanism: an accepted match may make its keypoint unavailable to LATER queries (mvpMapPoints
// occupancy with Observations() > 0, :89-91 / :1364-1366 / :1493-1494; vpMatched, :376-377).  The outcome of query i is
// therefore a function of its own candidate list and of {outcomes of queries j < i}, a recurrence with exactly one
// solution; k_resolve settles it chunk by chunk in query order, each chunk by rounds of "outcomes against the claim
// table -> claims" until nothing changes, and finishes a chunk with a serial pass if a bound on the rounds is hit -- the
// same idea as k_sfi_resolve (orbfe_sfi.hip).
#include "orbfe_matcher_internal.h"
#include <atomic>
#include "orbfe_internal.h"

struct orbfe_extractor;
namespace orbfe {
int extractor_view(orbfe_extractor* h, int frame, ExtractView* out);
}

namespace {

constexpr int kCells = kGridCols * kGridRows;   // 3072

// ORBFE_FRAME_ZEROCOPY=0: copy commands (query arena marshalled on the host, frame arrays) instead of kernels reading
// page-locked host memory in place
bool frame_zero_copy() {
  const char* zc = getenv("ORBFE_FRAME_ZEROCOPY");
  return !(zc && atoi(zc) == 0);
}

struct FrameDev {   // device pointers of a resident frame
  float* x; float* y; float* angle; int* oct; uint8_t* desc;                       // keypoint order
  float* sx; float* sy; int* soct; int* sidx; uint8_t* tdesc; int* cellStart;      // grid order (Frame.cc:114-129)
  PairInfo* pair;
  int* tmpCell; int* tmpArr; int* tmpBucket;                                       // build scratch
  int n;
};

__device__ __forceinline__ void copy32(uint8_t* dst, const uint8_t* src) {
  const uint4 a = reinterpret_cast<const uint4*>(src)[0], b = reinterpret_cast<const uint4*>(src)[1];
  reinterpret_cast<uint4*>(dst)[0] = a;
  reinterpret_cast<uint4*>(dst)[1] = b;
}

// keypoints (cv::KeyPoint layout, mvKeysUn) + descriptor rows -> keypoint-order arrays.  desc != nullptr: the rows are
// fetched by this kernel too (page-locked staging memory read over PCIe: no copy command in front of the build);
// nullptr: they were uploaded straight to F.desc.
__global__ __launch_bounds__(256) void k_frame_from_host(const OrbfeKeyPoint* __restrict__ kps, const uint8_t* __restrict__ desc, FrameDev F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.n) return;
  const OrbfeKeyPoint k = kps[i];
  F.x[i] = k.x; F.y[i] = k.y; F.angle[i] = k.angle; F.oct[i] = k.octave;
  if (desc) copy32(F.desc + (size_t)i * 32, desc + (size_t)i * 32);
}

struct ExtractViewDev {
  const orbfe::SelKp* sel; const float* angle; const uint8_t* desc;
  int selOff[orbfe::kMaxLevels + 1], count[orbfe::kMaxLevels];
  float sf[orbfe::kMaxLevels];
  int nlevels;
};

// one frame of an extractor's result arena -> keypoint-order arrays, in the order orbfe_extract returns the keypoints
// (levels 0..n-1, list order inside a level; ORBextractor.cc:959-967: pt *= mvScaleFactor[level] for level > 0).
// xyUn != nullptr: the caller's undistorted coordinates (Frame::UndistortKeyPoints, Frame.cc:286-320) replace pt.
__global__ __launch_bounds__(256) void k_frame_from_extract(ExtractViewDev V, const float* __restrict__ xyUn, FrameDev F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F.n) return;
  int l = 0, first = 0;
  while (l + 1 < V.nlevels && i >= first + V.count[l]) { first += V.count[l]; l++; }
  const int slot = V.selOff[l] + (i - first);
  const orbfe::SelKp s = V.sel[slot];
  float x = (float)(s.xy & 0xffff), y = (float)(s.xy >> 16);
  if (l != 0) { x *= V.sf[l]; y *= V.sf[l]; }
  if (xyUn) { x = xyUn[2 * i]; y = xyUn[2 * i + 1]; }
  F.x[i] = x; F.y[i] = y; F.angle[i] = V.angle[slot]; F.oct[i] = l;
  copy32(F.desc + (size_t)i * 32, V.desc + (size_t)slot * 32);
}

// Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:114-129, 264-274) as a stable counting sort by cell: cell = ix * 48 +
// iy ascending, insertion (= keypoint index) order inside a cell -- the order GetFeaturesInArea's loops visit.  One block.
__global__ __launch_bounds__(1024) void k_frame_grid(FrameDev F, float minX, float minY, float invW, float invH) {
  constexpr int kPer = kCells / 1024;
  static_assert(kCells % 1024 == 0, "cells per thread");
  __shared__ int start[kCells + 1];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, n = F.n;
  for (int c = tid; c <= kCells; c += 1024) start[c] = 0;
  __syncthreads();
  for (int k = tid; k < n; k += 1024) {
    const int px = (int)roundf((F.x[k] - minX) * invW), py = (int)roundf((F.y[k] - minY) * invH);   // Frame.cc:266-267
    const int c = (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? -1 : px * kGridRows + py;
    F.tmpCell[k] = c;
    if (c >= 0) F.tmpArr[k] = atomicAdd(&start[c], 1);
  }
  __syncthreads();
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
  if (tid == 1023) start[kCells] = base;
  __syncthreads();
  for (int k = tid; k < n; k += 1024) {
    const int c = F.tmpCell[k];
    if (c >= 0) F.tmpBucket[start[c] + F.tmpArr[k]] = k;
  }
  __syncthreads();   // block-scope visibility of the bucket writes
  for (int k = tid; k < n; k += 1024) {
    const int c = F.tmpCell[k];
    if (c < 0) continue;
    const int b0 = start[c], b1 = start[c + 1];
    int rank = 0;
    for (int j = b0; j < b1; j++) rank += F.tmpBucket[j] < k;
    const int p = b0 + rank;
    F.sx[p] = F.x[k]; F.sy[p] = F.y[k]; F.soct[p] = F.oct[k]; F.sidx[p] = k;
    copy32(F.tdesc + (size_t)p * 32, F.desc + (size_t)k * 32);
  }
  for (int c = tid; c <= kCells; c += 1024) F.cellStart[c] = start[c];
  if (tid == 0) {
    PairInfo pi;
    pi.trainOff = 0; pi.cellOff = 0; pi.tdescOff = 0;
    pi.minX = minX; pi.minY = minY; pi.invW = invW; pi.invH = invH;
    *F.pair = pi;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The sequential bookkeeping on the device.
// ---------------------------------------------------------------------------------------------------------------------
enum { kModeMapPoints = 0, kModeUv = 1, kModeProjected = 2 };

// Candidate entries as k_window_match leaves them for a resident frame: keypoint index | Hamming distance << 16 |
// keypoint octave << 25, in GetFeaturesInArea order; candidates failing Fuse's chi-square gate are already gone.  Per
// query a 16-byte record {count, first three entries}; longer lists lie in the pool at qoff.
struct ResolveParams {
  const uint2* qword;       // [nq] packed query words as k_window_match leaves them (MatchParams::qword)
  const uint32_t* rec;      // [nq][4] full records (the distances of the accepted candidates are read from here)
  const uint32_t* qoff; const uint32_t* pool; uint32_t* total; uint32_t poolCap;
  const uint32_t* wpool; uint32_t* wtotal;   // wide records (four to six candidates) and the words of them in use
  int nq, n;
  const uint32_t* claimBits;   // bit i: an accepted match of query i makes its keypoint unavailable to later queries
  const uint32_t* occBits;     // bit k: keypoint k unavailable from the start (mvpMapPoints occupancy / kp_skip); nullptr: none
  const float* qangle; const float* kangle;
  float nnratio; int maxDist; int checkOri;
  int* scratch;             // generic kernel: [2 n] the table + kp_assigned
  int* outHost;             // page-locked host memory, written by the kernel itself (no copy command behind it):
                            // kp_assigned[n] (modes 0, 1) or best_idx[nq], best_dist[nq] (mode 2)
  int* hdrHost;             // [kHdr] result header, page-locked host memory
  int maxRounds;
  uint32_t tagMax;          // first round tag (kTagMax; the tests start lower to exercise the re-basing of used-up tags)
  int seq;                  // the call's number: written to hdrHost[5] last of all (the host polls it instead of waiting on the stream)
};

// Best / second best of one query over the candidates earlier queries have left it (`taken`: the keypoint's table word says
// an earlier query holds it, or it was unavailable from the start).
template <int MODE>
struct Best {
  int bestDist = MODE == kModeProjected ? INT_MAX : 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
  __device__ __forceinline__ void consider(uint32_t e, bool taken) {
    if (taken) return;                                     // ORBmatcher.cc:89-91, :1364-1366, :1493-1494, :376-377
    const int idx = (int)(e & 0xffff), dist = (int)((e >> 16) & 0x1ff), oct = (int)(e >> 25);
    if (MODE == kModeMapPoints) {                          // :95-109
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
        bestLevel = oct; bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = oct; bestDist2 = dist;
      }
    } else if (dist < bestDist) {
      bestDist = dist; bestIdx = idx;
    }
  }
  __device__ __forceinline__ int accept(const ResolveParams& R, int& distOut) const {
    distOut = -1;
    if (MODE == kModeMapPoints) {
      if (bestDist > TH_HIGH) return -1;
      if (bestLevel == bestLevel2 && (float)bestDist > R.nnratio * (float)bestDist2) return -1;   // :113-114
    } else if (bestDist > R.maxDist) {
      return -1;
    }
    distOut = bestDist;
    return bestIdx;
  }
};

__device__ __forceinline__ int rot_bin_dev(float a1, float a2) {   // ORBmatcher.cc:1385-1390 (factor = 1/HISTO_LENGTH quirk kept)
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0f) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

constexpr int kResolveThreads = 1024;
constexpr uint32_t kFree = 0xffffffffu;   // claim table: no claim
constexpr uint32_t kTagMax = 0xffeu;      // first round tag (12 bits above 20 bits of query index + 1)
constexpr int kResolveQpt = 2;   // queries per thread and chunk: 2 048 queries settle together (measured: 1 -> 66 k cycles for config 5's
                                 // 10 000 MapPoints, 2 -> 52 k, 4 -> 56 k, 6 -> 48 k, 10 -> 56 k; 2 keeps a tracking-sized search in one chunk)

// One block.  A query's outcome for every subset of its (at most three) candidates being available was tabulated by the
// window kernel (decision_code), so a round is a table lookup per query: fetch the candidates' table words, form the
// availability pattern, read the outcome.  LDS = true (the launcher checks that everything fits the budget): the packed
// query words (3 x 16-bit keypoint index + 16-bit code), the "first claiming query" table, kp_assigned and the claim /
// occupancy flags as bits live in LDS.  The queries are settled chunk by chunk in query order (kResolveThreads * QPT at a
// time, see below).  Lists longer than the record (wide windows) are walked from an LDS copy while there is room, else from
// the pool in global memory.  LDS = false keeps everything in global scratch.
template <int MODE, bool LDS, int QPT>
__global__ __launch_bounds__(kResolveThreads) void k_resolve(ResolveParams R, int ldsEntries) {
  extern __shared__ __align__(16) int dyn[];
  __shared__ int hist[36];   // 30 bins of the rotation histogram, then the three kept bins
  __shared__ int acc[2];
  __shared__ int orFlag[3];
  const int tid = threadIdx.x, nq = R.nq, n = R.n;
  const uint32_t tot = *R.total, wtot = *R.wtotal;
  auto stamp = [&](int k) { if (tid == 0) R.hdrHost[16 + k] = (int)__builtin_readcyclecounter(); };   // phase clock (debug)
  stamp(0);
  if (tid < 32) hist[tid] = 0;
  if (tid < 2) acc[tid] = 0;
  if (tid < 3) orFlag[tid] = 0;
  const int claimWords = (nq + 31) >> 5, occWords = (n + 31) >> 5;
  uint2* q8 = LDS ? reinterpret_cast<uint2*>(dyn) : nullptr;                         // [nq] packed query words, later the outcomes
  uint32_t* fc = reinterpret_cast<uint32_t*>(LDS ? dyn + 2 * nq : R.scratch);        // [n] the claim table (below)
  int* kpAssigned = reinterpret_cast<int*>(fc) + n;                                   // [n]
  uint32_t* claimL = LDS ? reinterpret_cast<uint32_t*>(kpAssigned + n) : nullptr;    // [claimWords]
  uint32_t* occL = LDS ? claimL + claimWords : nullptr;                               // [occWords]
  const bool withOri = MODE == kModeUv && R.checkOri;
  uint32_t* ovf = LDS ? occL + occWords : nullptr;                                    // [ldsEntries] the pools' copy
  const uint32_t* claimW = LDS ? claimL : R.claimBits;
  const uint32_t* occW = LDS ? occL : R.occBits;
  auto occupied = [&](int k) -> bool { return R.occBits && ((occW[k >> 5] >> (k & 31)) & 1u) != 0; };
  if (LDS) {
    // straight copies, every load independent of every other (and of the counter's): query words, flag masks
    for (int i0 = tid; i0 < nq; i0 += 8 * kResolveThreads) {   // eight loads in flight per thread before the first LDS store
      uint2 w[8];
#pragma unroll
      for (int u = 0; u < 8; u++) w[u] = R.qword[min(i0 + u * kResolveThreads, nq - 1)];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + u * kResolveThreads < nq) q8[i0 + u * kResolveThreads] = w[u];
    }
    for (int w = tid; w < claimWords; w += kResolveThreads) claimL[w] = R.claimBits[w];
    if (R.occBits)
      for (int w = tid; w < occWords; w += kResolveThreads) occL[w] = R.occBits[w];
  }
  __syncthreads();        // (every thread has read the counter)
  if (tid == 0) {
    R.hdrHost[4] = (int)tot;
    *R.total = 0u;        // ready for the next search
    *R.wtotal = 0u;
    R.hdrHost[1] = tot > R.poolCap ? 1 : 0;
    if (tot > R.poolCap) {
      __threadfence_system();
      __hip_atomic_store(&R.hdrHost[5], R.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (tot > R.poolCap) return;   // candidate pool too small: the host grows it and submits again
  // what lies beyond the records moves to LDS while there is room: the wide records first (every round looks them up), then
  // the pool (the lists that are walked)
  const bool wideInLds = LDS && wtot <= (uint32_t)ldsEntries;
  const bool poolInLds = wideInLds && wtot + tot <= (uint32_t)ldsEntries;
  const uint32_t* wides = wideInLds ? ovf : R.wpool;
  const uint32_t* lists = poolInLds ? ovf + wtot : R.pool;
  if (wideInLds)
    for (uint32_t e = (uint32_t)tid; e < wtot; e += kResolveThreads) ovf[e] = R.wpool[e];
  if (poolInLds)
    for (uint32_t e = (uint32_t)tid; e < tot; e += kResolveThreads) ovf[wtot + e] = R.pool[e];
  for (int k = tid; k < n; k += kResolveThreads) {
    fc[k] = occupied(k) ? 0u : kFree;
    kpAssigned[k] = -1;
  }
  __syncthreads();
  stamp(1);
  auto claims = [&](int i) -> bool { return ((claimW[i >> 5] >> (i & 31)) & 1u) != 0; };
  auto query_word = [&](int i) -> uint2 { return LDS ? q8[i] : R.qword[i]; };
  // The claim table.  Word of keypoint k = tag << 20 | (claiming query + 1), kept minimal by atomicMin: tag 0 = a settled claim
  // (0 itself: unavailable from the start), tag T > 0 = a claim made in the round with tag T; rounds count their tags DOWN, so
  // a newer claim overwrites any older tentative one, and a reader in the round with tag T holds
  //   keypoint taken for query i  <=>  word < (T << 20 | i + 1):
  // settled claims (all by earlier chunks' queries or, inside a serial pass, by earlier queries), this round's input claims by
  // queries j < i, and nothing else -- claims of older rounds that were not renewed carry a larger tag and drop out by
  // themselves, nobody has to take them back.  kFree = no claim.
  // outcome of query i (packed word r) given thr = T << 20 | i + 1: accepted keypoint or -1; `which` = 1 + position of the
  // accepted candidate in the record, 9 + position for a wide record, -(distance) - 1 for a walked list
  auto eval = [&](const uint2 r, const uint32_t thr, int& which) -> int {
    const uint32_t code = r.y >> 16;
    if (code < kCodeWide) {
      // branch-free (code 0 = no candidates = outcome 0 for every pattern; unused slots hold index 0: their bit does not
      // change the outcome)
      const uint32_t i0 = r.x & 0xffffu, i1 = r.x >> 16, i2 = r.y & 0xffffu;
      const uint32_t f0 = fc[i0], f1 = fc[i1], f2 = fc[i2];
      const unsigned p = (f0 >= thr ? 1u : 0u) + (f1 >= thr ? 2u : 0u) + (f2 >= thr ? 4u : 0u);
      const unsigned o = (code >> (2u * p)) & 3u;
      const unsigned long long pk = (unsigned long long)r.x | ((unsigned long long)i2 << 32);
      const int sel = (int)((uint32_t)(pk >> (16u * ((o + 3u) & 3u))) & 0xffffu);   // candidate o - 1 (o = 0: unused)
      which = (int)o;
      return o ? sel : -1;
    }
    which = 0;
    if (code == kCodeWide) {                             // four to six candidates: the same lookup over 2^N patterns
      const uint32_t* wr = wides + r.x;
      const unsigned N = r.y & 0xffffu;
      const uint32_t a = wr[0], b = wr[1], c = wr[2];
      const uint32_t f0 = fc[a & 0xffffu], f1 = fc[a >> 16], f2 = fc[b & 0xffffu], f3 = fc[b >> 16], f4 = fc[c & 0xffffu], f5 = fc[c >> 16];
      const unsigned p = ((f0 >= thr ? 1u : 0u) | (f1 >= thr ? 2u : 0u) | (f2 >= thr ? 4u : 0u) | (f3 >= thr ? 8u : 0u) |
                          (f4 >= thr ? 16u : 0u) | (f5 >= thr ? 32u : 0u)) & ((1u << N) - 1u);   // (slots beyond N hold index 0)
      const unsigned o = (wr[3 + (p >> 3)] >> (3u * (p & 7u))) & 7u;
      const uint32_t pair = o <= 2u ? a : (o <= 4u ? b : c);
      which = o ? 8 + (int)o : 0;
      return o ? (int)((pair >> (16u * ((o - 1u) & 1u))) & 0xffffu) : -1;
    }
    Best<MODE> B;                                        // a longer list: walk it
    const int cnt = (int)(r.y & 0xffffu);
    const uint32_t* l = lists + r.x;
    for (int c = 0; c < cnt; c += 4) {   // four entries, then their four table words, per pair of dependent trips
      uint32_t e[4], fv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) e[u] = l[min(c + u, cnt - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) fv[u] = fc[e[u] & 0xffff];
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (c + u < cnt) B.consider(e[u], fv[u] < thr);
    }
    int d;
    const int m = B.accept(R, d);
    which = -d - 1;
    return m;
  };
  auto dist_of = [&](int i, int which) -> int {
    if (which < 0) return -which - 1;
    if (which >= 8) return (int)((R.pool[R.qoff[i] + (uint32_t)(which - 9)] >> 16) & 0x1ffu);   // wide record: entries in the pool
    return (int)((R.rec[(size_t)i * 4 + which] >> 16) & 0x1ffu);
  };
  // ---- the recurrence, chunk by chunk in query order ----------------------------------------------------------------
  // Query i depends on queries j < i only, so once every earlier chunk is settled a chunk of kResolveThreads * QPT queries
  // settles in (longest chain inside the chunk) + 1 rounds -- two or three for most chunks at tracking radii: outcomes
  // against the table, claims, outcomes again = unchanged.  A round is: outcomes (one dependent LDS trip: the chunk's query
  // words and claim flags stay in registers) | barrier carrying the block-wide "any outcome changed" | claims | barrier.
  // block-wide OR with ONE barrier: three rotating flags (the one of this call is set, the next one cleared, the third may
  // still be read by a wave that has not left the previous call)
  // a round communicates through LDS only (LDS = true): its barriers wait for the LDS counter, not for vector memory -- the
  // angle prefetch below stays in flight across the rounds
  auto round_barrier = [&]() {
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  // Frame / KeyFrame searches with the rotation check: the angles of the first sources of every thread (they may lie in
  // page-locked host memory: a PCIe round trip) are requested now and used after the rounds
  float qaPre[4] = {0.f, 0.f, 0.f, 0.f};
  if (withOri) {
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (tid + u * kResolveThreads < nq) qaPre[u] = R.qangle[tid + u * kResolveThreads];
  }
  int orPhase = 0;
  auto block_or = [&](int v) -> int {
    if (__ballot(v != 0) != 0ull && (tid & 63) == 0) orFlag[orPhase] = 1;
    const int nextPhase = orPhase == 2 ? 0 : orPhase + 1;
    if (tid == 0) orFlag[nextPhase] = 0;
    round_barrier();
    const int r = orFlag[orPhase];
    orPhase = nextPhase;
    return r;
  };
  const int maxRounds = min(R.maxRounds, 2000);
  uint32_t tag = R.tagMax;      // tag of the claims the NEXT round reads (none carry it yet)
  int rounds = 0;
  bool serialUsed = false;
  for (int c0 = 0; c0 < nq; c0 += kResolveThreads * QPT) {
    if (tag < (uint32_t)maxRounds + 4u) {   // the tags are used up (thousands of rounds): forget every unsettled claim, start over
      for (int k = tid; k < n; k += kResolveThreads)
        if (fc[k] >> 20) fc[k] = kFree;
      tag = R.tagMax;
      __syncthreads();
    }
    int mPrev[QPT], mNew[QPT], wNew[QPT];
    uint2 rq[QPT];
    bool cl[QPT];
#pragma unroll
    for (int u = 0; u < QPT; u++) {
      const int i = c0 + u * kResolveThreads + tid;
      mPrev[u] = -1; mNew[u] = -1; wNew[u] = 0;
      rq[u] = make_uint2(0u, 0u);
      cl[u] = false;
      if (i < nq) { rq[u] = query_word(i); cl[u] = claims(i); }
    }
    int it = 0;
    bool serial = false;
    for (;;) {
      int changed = 0;
#pragma unroll
      for (int u = 0; u < QPT; u++) {
        const int i = c0 + u * kResolveThreads + tid;
        if (i < nq) {
          mNew[u] = eval(rq[u], (tag << 20) | (uint32_t)(i + 1), wNew[u]);
          changed |= mNew[u] != mPrev[u];
        }
      }
      const int any = block_or(changed);
      it++;
      if (!any) break;                   // outcomes are a function of the table and the table of the outcomes: the fixed point
      tag--;                             // this round's claims; everything tentative so far drops out of sight with it
      if (it >= maxRounds) {             // a long chain inside the chunk: one serial pass over it, settling claims as it goes
        serial = true;
        if (tid == 0) {
          const int cEnd = min(nq, c0 + kResolveThreads * QPT);
          for (int i = c0; i < cEnd; i++) {
            int w;
            const int m = eval(query_word(i), (tag << 20) | (uint32_t)(i + 1), w);
            if (m >= 0 && claims(i) && fc[m] >= (tag << 20)) fc[m] = (uint32_t)(i + 1);
            if (LDS) q8[i] = make_uint2((uint32_t)m, (uint32_t)w);
          }
        }
        __syncthreads();
        break;
      }
#pragma unroll
      for (int u = 0; u < QPT; u++) {
        const int i = c0 + u * kResolveThreads + tid;
        if (mNew[u] >= 0 && cl[u]) atomicMin(&fc[mNew[u]], (tag << 20) | (uint32_t)(i + 1));
        mPrev[u] = mNew[u];
      }
      round_barrier();
    }
    if (!serial) {
      // settled: the claims lose their tag (the table word of a keypoint two of the chunk's queries accept is the earlier one's
      // either way), and the outcome replaces the query word (nobody reads that again)
#pragma unroll
      for (int u = 0; u < QPT; u++) {
        const int i = c0 + u * kResolveThreads + tid;
        if (mNew[u] >= 0 && cl[u]) atomicMin(&fc[mNew[u]], (uint32_t)(i + 1));
        if (LDS && i < nq) q8[i] = make_uint2((uint32_t)mNew[u], (uint32_t)wNew[u]);
      }
      round_barrier();
    }
    serialUsed |= serial;
    rounds = max(rounds, it);
  }
  if (serialUsed) rounds = -rounds;
  __syncthreads();
  stamp(2);
  // the settled outcome of query i: kept in LDS, or evaluated once more against the final table (global-scratch route)
  auto outcome = [&](int i, int& which) -> int {
    if (LDS) { const uint2 r = q8[i]; which = (int)r.y; return (int)r.x; }
    return eval(query_word(i), (uint32_t)(i + 1), which);   // (tag 0: settled claims only)
  };
  // ---- outputs -----------------------------------------------------------------------------------------------------
  // (four queries per thread and step: the angles of the rotation histogram -- the sources' may lie in page-locked host
  // memory -- are requested together, one PCIe round trip per step instead of one per query; the bin is kept in the query's
  // LDS word for the second sweep)
  int nm = 0;
  for (int i0 = tid; i0 < nq; i0 += 4 * kResolveThreads) {
    int mm[4], ww[4];
    float qa[4] = {0.f, 0.f, 0.f, 0.f}, ka[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * kResolveThreads;
      mm[u] = -1; ww[u] = 0;
      if (i < nq) mm[u] = outcome(i, ww[u]);
    }
    if (withOri) {
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (mm[u] >= 0) { qa[u] = i0 == tid ? qaPre[u] : R.qangle[i0 + u * kResolveThreads]; ka[u] = R.kangle[mm[u]]; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * kResolveThreads, m = mm[u];
      if (i >= nq) continue;
      if (MODE == kModeProjected) { R.outHost[i] = m; R.outHost[nq + i] = m >= 0 ? dist_of(i, ww[u]) : -1; }
      if (m < 0) continue;
      nm++;
      if (MODE != kModeProjected) atomicMax(&kpAssigned[m], i);   // F.mvpMapPoints[bestIdx] = pMP: the last writer stays
      if (withOri) {
        const int bin = rot_bin_dev(qa[u], ka[u]);
        if (LDS) q8[i].y = (uint32_t)bin;
        // most matches of a frame-to-frame search share one or two bins: one LDS atomic per distinct bin of the wave
        // (64 lanes on one address are served one after the other)
        unsigned long long todo = __ballot(1);
        while (todo) {
          const int lead = __builtin_ctzll(todo);
          const int b0 = __builtin_amdgcn_readlane(bin, lead);
          const unsigned long long same = __ballot(bin == b0);
          if ((int)(threadIdx.x & 63) == lead) atomicAdd(&hist[b0], (int)__popcll(same));
          todo &= ~same;
          if (bin == b0) break;
        }
      }
    }
  }
  __syncthreads();
  if (withOri) {
    // ComputeThreeMaxima, ORBmatcher.cc:1554-1595: its sequential scan with strict comparisons keeps the three largest
    // non-empty bins ordered by (count descending, index ascending) -- three wave-wide maxima over count << 8 | 31 - bin by
    // the first wave, one lane per bin; every push of a losing bin resets its keypoint and takes one off the count (:1404-1416)
    if (tid < 64) {
      const int cnt = tid < HISTO_LENGTH ? hist[tid] : 0;
      int key = cnt > 0 ? (cnt << 8) | (31 - tid) : 0;
      int top[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        int mx = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        top[t] = mx;
        if (key == mx) key = 0;
      }
      if (tid == 0) {
        const int max1 = top[0] >> 8, max2 = top[1] >> 8, max3 = top[2] >> 8;
        int ind1 = top[0] ? 31 - (top[0] & 0xff) : -1, ind2 = top[1] ? 31 - (top[1] & 0xff) : -1, ind3 = top[2] ? 31 - (top[2] & 0xff) : -1;
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        hist[30] = ind1; hist[31] = ind2; hist[32] = ind3;
      }
    }
    __syncthreads();
    const int ind1 = hist[30], ind2 = hist[31], ind3 = hist[32];
    for (int i = tid; i < nq; i += kResolveThreads) {
      int w;
      const int m = outcome(i, w);
      if (m < 0) continue;
      const int b = LDS ? w : rot_bin_dev(R.qangle[i], R.kangle[m]);
      if (b != ind1 && b != ind2 && b != ind3) { kpAssigned[m] = -2; nm--; }
    }
    __syncthreads();
  }
  if (MODE != kModeProjected)
    for (int k = tid; k < n; k += kResolveThreads) R.outHost[k] = kpAssigned[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nm += __shfl_xor(nm, o, 64);
  if ((tid & 63) == 0 && nm) atomicAdd(&acc[0], nm);
  // this thread's result words (stores to uncached page-locked memory: no cache to write back -- a system-scope fence per
  // wave would write the whole L2 back sixteen times) are acknowledged before ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    R.hdrHost[0] = acc[0]; R.hdrHost[2] = rounds; R.hdrHost[3] = LDS ? (poolInLds ? 2 : 1) : 0;
    stamp(3);
    __threadfence_system();
    __hip_atomic_store(&R.hdrHost[5], R.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... the host sees the call's number
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
struct orbfe_frame {
  int device = 0, n = 0, maxOctave = 0;
  float bounds[4] = {0, 0, 0, 0};
  float invW = 0, invH = 0;
  DevBuf<uint8_t> buf;
  PinBuf<uint8_t> stage;
  FrameDev D{};
  hipEvent_t ready = nullptr;
  ~orbfe_frame() {
    (void)hipSetDevice(device);
    // `ready` may have been recorded on the stream of an extractor that no longer exists (its destructor waited for the
    // stream's work, so the build is complete): errors of these two calls are not errors of the frame
    if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); }
    buf.release(); stage.release();
    (void)hipGetLastError();
  }
  int carve(int count) {
    const size_t c = (size_t)std::max(count, 1);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
    const size_t ox = take(4 * c), oy = take(4 * c), oa = take(4 * c), oo = take(4 * c), od = take(32 * c), osx = take(4 * c),
                 osy = take(4 * c), oso = take(4 * c), osi = take(4 * c), otd = take(32 * c), ocs = take(4 * (kCells + 1)),
                 op = take(sizeof(PairInfo)), ot1 = take(4 * c), ot2 = take(4 * c), ot3 = take(4 * c);
    int rc;
    if ((rc = buf.ensure(o))) return rc;
    uint8_t* B = buf.p;
    D.x = (float*)(B + ox); D.y = (float*)(B + oy); D.angle = (float*)(B + oa); D.oct = (int*)(B + oo); D.desc = B + od;
    D.sx = (float*)(B + osx); D.sy = (float*)(B + osy); D.soct = (int*)(B + oso); D.sidx = (int*)(B + osi); D.tdesc = B + otd;
    D.cellStart = (int*)(B + ocs); D.pair = (PairInfo*)(B + op);
    D.tmpCell = (int*)(B + ot1); D.tmpArr = (int*)(B + ot2); D.tmpBucket = (int*)(B + ot3);
    D.n = count;
    n = count;
    return ORBFE_OK;
  }
  void setBounds(const float b[4]) {
    for (int i = 0; i < 4; i++) bounds[i] = b[i];
    invW = static_cast<float>(kGridCols) / static_cast<float>(b[1] - b[0]);   // Frame.cc:98
    invH = static_cast<float>(kGridRows) / static_cast<float>(b[3] - b[2]);   // Frame.cc:99
  }
  int grid(hipStream_t st) {
    hipLaunchKernelGGL(k_frame_grid, dim3(1), dim3(1024), 0, st, D, bounds[0], bounds[2], invW, invH);
    HIP_TRY(hipGetLastError());
    if (!ready) HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ready, st));
    return ORBFE_OK;
  }
  // (re)fill from host arrays on stream st
  int fromHost(const OrbfeKeyPoint* kps, const uint8_t* desc, int count, const float b[4], hipStream_t st) {
    int rc;
    if ((rc = carve(count))) return rc;
    setBounds(b);
    maxOctave = 0;
    for (int i = 0; i < count; i++) maxOctave = std::max(maxOctave, kps[i].octave < 0 ? INT_MAX : kps[i].octave);
    if (count > 0) {
      const size_t kb = al(sizeof(OrbfeKeyPoint) * (size_t)count), total = kb + 32 * (size_t)count;
      if ((rc = stage.ensure(total))) return rc;
      memcpy(stage.p, kps, sizeof(OrbfeKeyPoint) * (size_t)count);
      memcpy(stage.p + kb, desc, 32 * (size_t)count);
      if (frame_zero_copy()) {
        hipLaunchKernelGGL(k_frame_from_host, dim3((count + 255) / 256), dim3(256), 0, st, (const OrbfeKeyPoint*)stage.p, stage.p + kb, D);
      } else {
        // the keypoint records (28 B each) are staged in the grid-order descriptor array (32 B per keypoint), which the
        // grid kernel fills only afterwards; the descriptor rows go straight to their final place
        HIP_TRY(hipMemcpyAsync(D.tdesc, stage.p, sizeof(OrbfeKeyPoint) * (size_t)count, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(D.desc, stage.p + kb, 32 * (size_t)count, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_frame_from_host, dim3((count + 255) / 256), dim3(256), 0, st, (const OrbfeKeyPoint*)D.tdesc, (const uint8_t*)nullptr, D);
      }
    }
    return grid(st);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// One search on a resident frame: query arena (page-locked) -> device, k_window_match, k_resolve, result vector back.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct SearchPlan {
  int nq = 0, n = 0, mode = 0, nlevels = 0;
  float *qx = nullptr, *qy = nullptr, *qr = nullptr, *qangle = nullptr, *invSigma2 = nullptr;
  int *qa = nullptr, *qb = nullptr;
  uint8_t *qclaim = nullptr, *occ0 = nullptr, *qdesc = nullptr;
  size_t oQx = 0, oQy = 0, oQr = 0, oQang = 0, oQa = 0, oQb = 0, oQc = 0, oOcc = 0, oSig = 0, oQd = 0, head = 0, total = 0;
};

// Raw queries: the caller's own arrays, read by k_window_match in place when they are page-locked, else from a plain copy
// in the query arena -- no per-query loop on the host at all (that loop was 13 of the 85 us of a 10 000-MapPoint search).
struct RawQ {
  int kind = 0;                       // MatchParams::rawKind
  const float* xy = nullptr;
  const int* level = nullptr;
  const float* aux = nullptr;         // viewcos (kind 1) / radius (kind 3)
  const uint8_t* flags = nullptr;     // ORBFE_MP_* (kind 1) / valid (kinds 2, 3)
  const uint8_t* claimSrc = nullptr;  // claim flag = (byte & claimMask) != 0 ...
  uint8_t claimMask = 0xff;
  int claimConst = -1;                // ... or this constant
  const float* angle = nullptr;       // kind 2: source keypoint angles (rotation histogram)
  const uint8_t* occ = nullptr;       // [n] or nullptr
  const float* sf = nullptr;          // mvScaleFactors (kinds 1, 2)
  int nlevels = 0;
  float th = 1.f;
  int factor = 0;
  const int* descRow = nullptr;       // indexed descriptor rows (orbfe_search_by_projection_frame_rows): host array [nq] ...
  const uint8_t* descHost = nullptr;  // ... and the table's complete page-locked host mirror
  int descRowWhere = 0;
  // where each array lives (gpu_readable), filled by classify_host_arrays: xy, level, aux, flags, claimSrc, angle, occ
  int where[7] = {0, 0, 0, 0, 0, 0, 0};
};

// > 0: the kernels can read p in place (1: page-locked host memory, 2: memory of device `device`); 0: ordinary host memory
// (the caller copies it into the arena); -1: memory of another device
int gpu_readable(const void* p, int device) {
  hipPointerAttribute_t attr;
  if (p && hipPointerGetAttributes(&attr, p) == hipSuccess) {
    if (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged) return 1;
    if (attr.type == hipMemoryTypeDevice) return attr.device == device ? 2 : -1;
    return 0;
  }
  (void)hipGetLastError();
  return 0;
}
// The per-query arrays other than the descriptor rows, and the occupancy bytes, are also READ BY THE HOST (level range,
// largest radius, the marshalled routes): they must be host memory -- page-locked (then the kernels read them in place) or
// ordinary.  A device pointer is refused here, before anything dereferences it (include/orbfe.h states the contract).
int classify_host_arrays(const orbfe_matcher* m, RawQ* Q) {
  const void* p[7] = {Q->xy, Q->level, Q->aux, Q->flags, Q->claimSrc, Q->angle, Q->occ};
  static const char* const name[7] = {"coordinates", "levels", "viewing cosines / radii", "flags", "claim flags", "angles", "occupancy bytes"};
  for (int k = 0; k < 7; k++) {
    if (!p[k]) { Q->where[k] = 0; continue; }
    if (k == 4 && p[4] == p[3]) { Q->where[4] = Q->where[3]; continue; }
    const int g = gpu_readable(p[k], m->device);
    if (g == 2 || g < 0) {
      set_err("the %s live in device memory: of a search's input arrays only the descriptor rows may (the others are read by the host too)", name[k]);
      return ORBFE_ERR_INVALID;
    }
    Q->where[k] = g;
  }
  if (Q->descRow) {
    const int g = gpu_readable(Q->descRow, m->device);
    if (g == 2 || g < 0) { set_err("the descriptor row indices live in device memory: they must be host memory"); return ORBFE_ERR_INVALID; }
    Q->descRowWhere = g;
  }
  return ORBFE_OK;
}
// levels of the active queries -- (flags[i] & need) == want, or flags[i] != 0 when need == 0 -- all inside [0, nlevels)?  *maxSf = the largest scale
// factor among them.  One pass of min / max over ALL levels first (vectorised; enough when no level is out of range).
int check_levels(const int* level, const uint8_t* flags, uint8_t need, uint8_t want, int nq, const float* sf, int nlevels, float* maxSf,
                 const char* what) {
  int lo = INT_MAX, hi = INT_MIN;
  for (int i = 0; i < nq; i++) { lo = std::min(lo, level[i]); hi = std::max(hi, level[i]); }
  if (lo >= 0 && hi < nlevels) {
    float s = 0.f;
    for (int l = lo; l <= hi; l++) s = std::max(s, sf[l]);   // (an upper bound: the level of an inactive query may be among them)
    *maxSf = s;
    return ORBFE_OK;
  }
  float s = 0.f;
  for (int i = 0; i < nq; i++) {
    if (need ? (flags[i] & need) != want : flags[i] == 0) continue;
    const int lvl = level[i];
    if (lvl < 0 || lvl >= nlevels) { set_err("%s %d: level %d out of range", what, i, lvl); return ORBFE_ERR_INVALID; }
    s = std::max(s, sf[lvl]);
  }
  *maxSf = s;
  return ORBFE_OK;
}

// carve the query arena for nq queries against a frame of n keypoints; the caller fills the host views
int plan_search(orbfe_matcher* m, const orbfe_frame* f, int mode, int nq, bool withOcc, int nlevelsSigma, SearchPlan* P) {
  P->nq = nq; P->n = f->n; P->mode = mode; P->nlevels = nlevelsSigma;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t q = (size_t)std::max(nq, 1);
  P->oQx = take(4 * q); P->oQy = take(4 * q); P->oQr = take(4 * q); P->oQang = take(4 * q); P->oQa = take(4 * q); P->oQb = take(4 * q);
  P->oQc = take(q); P->oOcc = take((size_t)std::max(f->n, 1)); P->oSig = take(4 * (size_t)std::max(nlevelsSigma, 1));
  P->head = o;                 // everything up to here travels in one copy; the descriptors follow (or come from the caller's page-locked rows)
  P->oQd = take(32 * q);
  P->total = o;
  int rc;
  if ((rc = m->h_q.ensure(P->total))) return rc;
  if ((rc = m->d_q.ensure(P->total))) return rc;
  uint8_t* H = m->h_q.p;
  P->qx = (float*)(H + P->oQx); P->qy = (float*)(H + P->oQy); P->qr = (float*)(H + P->oQr); P->qangle = (float*)(H + P->oQang);
  P->qa = (int*)(H + P->oQa); P->qb = (int*)(H + P->oQb); P->qclaim = H + P->oQc;
  P->occ0 = withOcc ? H + P->oOcc : nullptr;
  P->invSigma2 = nlevelsSigma > 0 ? (float*)(H + P->oSig) : nullptr;
  P->qdesc = H + P->oQd;
  return ORBFE_OK;
}

// result header (ints): [0] nmatches, [1] candidate pool overflow, [2] rounds of the fixed point (< 0: finished serially),
// [4] candidate entries needed, [8] / [9] the window kernel's running counters of pool entries / wide-record words (zero
// between searches)
constexpr int kHdr = 64;

// out: pointer into the page-locked result area: kp_assigned[n] (modes 0, 1) or best_idx[nq], best_dist[nq] (mode 2)
int run_search(orbfe_matcher* m, orbfe_frame* f, const SearchPlan& P, const uint8_t* qdescHost, float rmax, float nnratio,
               int maxDist, double chi2, int checkOri, const int** out, int* nmatches, const RawQ* raw = nullptr) {
  HIP_TRY(hipSetDevice(m->device));
  (void)hipGetLastError();
  if (f->device != m->device) { set_err("frame and matcher live on different devices"); return ORBFE_ERR_INVALID; }
  if (P.nq >= (1 << 20) - 1) { set_err("at most 1 048 574 queries per search"); return ORBFE_ERR_INVALID; }
  const double tA = orbfe_matcher::nowMs();
  hipStream_t st = m->stream;
  const int nq = P.nq, n = P.n;
  int rc;
  const size_t outInts = P.mode == kModeProjected ? 2 * (size_t)nq : (size_t)n;
  // device result area (ints): header | qcount[nq] | qoff[nq] | records[4 nq] | query words[2 nq] | generic tables[2 n] |
  // claim mask, occupancy mask (64-bit words)
  const size_t oCnt = kHdr, oOff = oCnt + nq, oRec = (oOff + nq + 3) & ~(size_t)3, oQw = oRec + 4 * (size_t)nq,
               oFc = oQw + 2 * (size_t)nq, oFlags = (oFc + 2 * (size_t)n + 1) & ~(size_t)1,
               claimW64 = ((size_t)nq + 63) / 64, occW64 = ((size_t)n + 63) / 64, words = oFlags + 2 * (claimW64 + occW64) + 64;
  if (m->d_r.n < words) {
    if ((rc = m->d_r.ensure(words + words / 2))) return rc;
    HIP_TRY(hipMemsetAsync(m->d_r.p, 0, kHdr * sizeof(int), st));   // a fresh allocation: the running counter starts at zero
  }
  if ((rc = m->h_r.ensure(kHdr + outInts + 64))) return rc;
  if (f->ready && hipStreamWaitEvent(st, f->ready, 0) != hipSuccess) (void)hipGetLastError();   // (the recording stream is gone: the build is complete)
  // queries: one copy for the scalar arrays; descriptor rows straight from the caller's memory when it is page-locked
  // (the kernel fetches rows as two 16-byte words: rows it reads in place must be 16-byte aligned)
  const bool indexed = raw && raw->descRow;
  const int rowsWhere = gpu_readable(qdescHost, m->device);
  if (indexed && (rowsWhere != 2 || gpu_readable(raw->descHost, m->device) != 1 || ((uintptr_t)raw->descHost & 15u) != 0)) {
    set_err("indexed descriptor rows: the table must live in the frame's device memory and its mirror in page-locked host memory, both 16-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  if (rowsWhere < 0 || (rowsWhere == 2 && ((uintptr_t)qdescHost & 15u) != 0)) {
    set_err(rowsWhere < 0 ? "the descriptor rows live in the memory of another device" : "device-resident descriptor rows must be 16-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  const bool pinned = rowsWhere > 0 && ((uintptr_t)qdescHost & 15u) == 0;
  // No upload command at all by default: the kernels read the page-locked query arena (and the caller's page-locked
  // descriptor rows) over PCIe themselves -- every query word is read once, and a DMA in front of the first kernel costs its
  // own latency plus a copy-engine -> compute hand-over (16 + 8 us measured for the 490 KB of 10 000 MapPoints).
  // ORBFE_FRAME_ZEROCOPY=0 brings the upload back.
  const bool zeroCopy = frame_zero_copy();
  if (raw && !zeroCopy) { set_err("raw queries need the zero-copy route"); return ORBFE_ERR_INVALID; }
  const uint8_t* qdescDev = m->d_q.p + P.oQd;
  if (zeroCopy) {
    if (pinned) qdescDev = qdescHost;
    else { memcpy(P.qdesc, qdescHost, 32 * (size_t)nq); qdescDev = P.qdesc; }
  } else if (pinned) {
    HIP_TRY(hipMemcpyAsync(m->d_q.p, m->h_q.p, P.head, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->d_q.p + P.oQd, qdescHost, 32 * (size_t)nq, hipMemcpyDefault, st));
  } else {
    memcpy(P.qdesc, qdescHost, 32 * (size_t)nq);
    HIP_TRY(hipMemcpyAsync(m->d_q.p, m->h_q.p, P.oQd + 32 * (size_t)nq, hipMemcpyHostToDevice, st));
  }
  // raw queries: page-locked arrays are read where they are, the others from a plain copy in the arena
  RawQ W;
  if (raw) {
    W = *raw;
    uint8_t* H = m->h_q.p;
    // (classify_host_arrays has vetted them: page-locked = 1, ordinary = 0, nothing device-resident)
    auto view = [&](const void* src, int where, size_t bytes, size_t at) -> const void* {
      if (!src || where > 0) return src;
      memcpy(H + at, src, bytes);
      return H + at;
    };
    W.xy = (const float*)view(raw->xy, raw->where[0], 8 * (size_t)nq, P.oQx);          // (the qx and qy regions are adjacent)
    W.level = (const int*)view(raw->level, raw->where[1], 4 * (size_t)nq, P.oQa);
    W.aux = (const float*)view(raw->aux, raw->where[2], 4 * (size_t)nq, P.oQr);
    W.flags = (const uint8_t*)view(raw->flags, raw->where[3], (size_t)nq, P.oQc);
    W.claimSrc = raw->claimSrc == raw->flags ? W.flags : (const uint8_t*)view(raw->claimSrc, raw->where[4], (size_t)nq, P.oQb);
    W.angle = (const float*)view(raw->angle, raw->where[5], 4 * (size_t)nq, P.oQang);
    W.occ = (const uint8_t*)view(raw->occ, raw->where[6], (size_t)n, P.oOcc);
    if (indexed) W.descRow = (const int*)view(raw->descRow, raw->descRowWhere, 4 * (size_t)nq, P.oQd);   // (the descriptor region of the arena is free)
    if (raw->sf) {   // scale factors in device memory, refreshed when the caller's differ from the copy there
      if (raw->nlevels > 32) { set_err("more than 32 levels"); return ORBFE_ERR_INVALID; }
      if (!m->d_sf.p) { if ((rc = m->d_sf.ensure(32))) return rc; m->sfN = -1; }
      if (m->sfN != raw->nlevels || memcmp(m->sfHost, raw->sf, sizeof(float) * (size_t)raw->nlevels) != 0) {
        memset(m->sfHost, 0, sizeof m->sfHost);
        memcpy(m->sfHost, raw->sf, sizeof(float) * (size_t)raw->nlevels);
        m->sfN = raw->nlevels;
        HIP_TRY(hipMemcpyAsync(m->d_sf.p, m->sfHost, sizeof m->sfHost, hipMemcpyHostToDevice, st));
      }
    }
  }
  const double tB = orbfe_matcher::nowMs();
  m->stageMs[0] = tB - (m->tEntry > 0 ? m->tEntry : tA);   // query marshalling (the caller's loop) + arena set-up
  uint8_t* Dq = zeroCopy ? m->h_q.p : m->d_q.p;
  int* Dr = (int*)m->d_r.p;
  size_t poolCap = m->d_pool.n ? m->d_pool.n : std::max<size_t>(1 << 16, (size_t)nq * 32);
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = m->d_pool.ensure(poolCap))) return rc;
    MatchParams M;
    M.sx = f->D.sx; M.sy = f->D.sy; M.soct = f->D.soct; M.sidx = f->D.sidx; M.cellStart = f->D.cellStart; M.tdesc = f->D.tdesc;
    M.pairs = f->D.pair; M.qpair = nullptr;
    M.qx = (const float*)(Dq + P.oQx); M.qy = (const float*)(Dq + P.oQy); M.qr = (const float*)(Dq + P.oQr);
    M.qminL = (const int*)(Dq + P.oQa); M.qmaxL = (const int*)(Dq + P.oQb); M.qdesc = qdescDev;
    M.nq = nq;
    M.total = (uint32_t*)(Dr + 8); M.qcount = (uint32_t*)(Dr + oCnt); M.qoff = (uint32_t*)(Dr + oOff);
    M.pool = m->d_pool.p; M.poolCap = (uint32_t)m->d_pool.n;
    if ((rc = m->d_wpool.ensure(wide_words(kWideMax) * (size_t)std::max(nq, 1)))) return rc;   // every query a six-candidate list
    M.wpool = m->d_wpool.p; M.wtotal = (uint32_t*)(Dr + 9);
    M.rec = (uint32_t*)(Dr + oRec);
    M.qword = (uint2*)(Dr + oQw);
    unsigned long long* dClaim = (unsigned long long*)(Dr + oFlags);
    unsigned long long* dOcc = dClaim + claimW64;
    M.bitSrc[0] = Dq + P.oQc; M.bitDst[0] = dClaim; M.bitN[0] = nq;
    bool withOcc = P.occ0 != nullptr;
    if (withOcc) { M.bitSrc[1] = Dq + P.oOcc; M.bitDst[1] = dOcc; M.bitN[1] = n; }
    const float* qangleDev = (const float*)(Dq + P.oQang);
    if (raw) {
      M.rawKind = W.kind; M.rawXY = W.xy; M.rawLevel = W.level; M.rawAux = W.aux; M.rawFlags = W.flags;
      M.rawSf = m->d_sf.p; M.rawTh = W.th; M.rawFactor = W.factor;
      if (indexed) { M.qdescRow = W.descRow; M.qdescAlt = W.descHost; }
      M.bitSrc[0] = W.claimSrc; M.bitMask[0] = W.claimMask; M.bitConst[0] = W.claimConst;
      withOcc = W.occ != nullptr;
      if (withOcc) { M.bitSrc[1] = W.occ; M.bitDst[1] = dOcc; M.bitN[1] = n; }
      qangleDev = W.angle;
    }
    M.invSigma2 = P.invSigma2 ? (const float*)(Dq + P.oSig) : nullptr;
    M.chi2 = chi2;
    M.packOctave = 1;
    M.codeMode = P.mode; M.nnratio = nnratio; M.maxDist = maxDist;
    {
      const float cols = 2.f * rmax * f->invW + 3.f;   // widest window in grid columns (+3: floor / ceil slack of the cell range)
      const int maxCols = cols >= 64.f ? 64 : std::max(1, (int)std::ceil(cols));
      int lpq = 8;
      while (lpq < 64 && lpq < maxCols) lpq <<= 1;
      const unsigned nblk = (unsigned)((nq + (kWinThreads / lpq) - 1) / (kWinThreads / lpq));
      if (lpq == 8) hipLaunchKernelGGL(k_window_match<8>, dim3(nblk), dim3(kWinThreads), 0, st, M);
      else if (lpq == 16) hipLaunchKernelGGL(k_window_match<16>, dim3(nblk), dim3(kWinThreads), 0, st, M);
      else if (lpq == 32) hipLaunchKernelGGL(k_window_match<32>, dim3(nblk), dim3(kWinThreads), 0, st, M);
      else hipLaunchKernelGGL(k_window_match<64>, dim3(nblk), dim3(kWinThreads), 0, st, M);
    }
    ResolveParams R;
    R.qword = M.qword; R.rec = M.rec; R.qoff = M.qoff; R.pool = M.pool; R.total = M.total; R.poolCap = M.poolCap;
    R.wpool = M.wpool; R.wtotal = M.wtotal;
    R.nq = nq; R.n = n;
    R.occBits = withOcc ? (const uint32_t*)dOcc : nullptr;
    R.claimBits = (const uint32_t*)dClaim; R.qangle = qangleDev; R.kangle = f->D.angle;
    R.nnratio = nnratio; R.maxDist = maxDist; R.checkOri = checkOri;
    R.scratch = Dr + oFc;
    R.hdrHost = m->h_r.p; R.outHost = m->h_r.p + kHdr;
    const char* mr = getenv("ORBFE_RESOLVE_MAX_ROUNDS");   // rounds of the fixed point before the serial finish
    R.maxRounds = mr ? std::max(1, atoi(mr)) : 48;
    const char* tm = getenv("ORBFE_RESOLVE_TAG_MAX");     // (tests) first round tag: a small one makes the kernel run out of tags
    R.tagMax = tm ? (uint32_t)std::min(std::max(atoi(tm), R.maxRounds + 8), (int)kTagMax) : kTagMax;
    m->seq = m->seq == INT_MAX ? 1 : m->seq + 1;
    R.seq = m->seq;
    // LDS-resident tables when they fit (152 KB of the CU's 160): query words, claim table, kp_assigned, flag masks; what is
    // left holds the candidate entries (the kernel checks their number at run time)
    const size_t fixedBytes = 4 * (2 * (size_t)nq + 2 * (size_t)n + (((size_t)nq + 31) >> 5) + (((size_t)n + 31) >> 5) + 8);
    const size_t budget = 152 * 1024;
    const char* gen = getenv("ORBFE_RESOLVE_GENERIC");      // 1: tables in global scratch whatever the size (tests)
    const bool lds = fixedBytes <= budget && !(gen && atoi(gen) != 0);
    const int ldsEntries = lds ? (int)((budget - fixedBytes) / 4) : 0;   // room for the lists longer than a record
    const size_t dynBytes = lds ? budget : 0;
#define ORBFE_LAUNCH_RESOLVE(MODE)                                                                                          \
    do {                                                                                                                      \
      if (lds) {                                                                                                              \
        if (!m->resolveAttr[MODE]) {                                                                                          \
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resolve<MODE, true, kResolveQpt>),                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));                            \
          m->resolveAttr[MODE] = true;                                                                                        \
        }                                                                                                                     \
        hipLaunchKernelGGL((k_resolve<MODE, true, kResolveQpt>), dim3(1), dim3(kResolveThreads), dynBytes, st, R, ldsEntries); \
      } else {                                                                                                                \
        hipLaunchKernelGGL((k_resolve<MODE, false, kResolveQpt>), dim3(1), dim3(kResolveThreads), 0, st, R, 0);               \
      }                                                                                                                       \
    } while (0)
    if (P.mode == kModeMapPoints) ORBFE_LAUNCH_RESOLVE(kModeMapPoints);
    else if (P.mode == kModeUv) ORBFE_LAUNCH_RESOLVE(kModeUv);
    else ORBFE_LAUNCH_RESOLVE(kModeProjected);
#undef ORBFE_LAUNCH_RESOLVE
    HIP_TRY(hipGetLastError());
    // The kernel wrote header and result vector into page-locked host memory itself, the call's number last: the host polls
    // that word (the end-of-kernel signal + hipStreamSynchronize's wake-up arrive microseconds later) and falls back to the
    // stream when it does not show up within 2 ms (ORBFE_FRAME_POLL=0: always the stream).
    {
      static const bool poll = [] { const char* e = ORBFE_EXP_ENV("ORBFE_FRAME_POLL"); return !(e && atoi(e) == 0); }();
      bool seen = false;
      if (poll) {
        const volatile int* done = m->h_r.p + 5;
        const double tW = orbfe_matcher::nowMs();
        for (unsigned spin = 1;; spin++) {
          if (*done == R.seq) { seen = true; break; }
          if ((spin & 255u) == 0 && orbfe_matcher::nowMs() - tW > 2.0) break;
          __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      if (!seen) HIP_TRY(hipStreamSynchronize(st));
    }
    if (!m->h_r.p[1]) {
      m->tSynced = orbfe_matcher::nowMs();
      m->stageMs[1] = m->tSynced - tB;
      m->stageMs[2] = 0;
      m->tEntry = 0;
      m->lastRounds = m->h_r.p[2];
      m->lastResolveRoute = m->h_r.p[3];
      *nmatches = m->h_r.p[0];
      *out = m->h_r.p + kHdr;
      return ORBFE_OK;
    }
    poolCap = (size_t)(uint32_t)m->h_r.p[4] + 1024;   // pool too small: grow to the demand and submit again
  }
  set_err("candidate pool sizing failed");
  return ORBFE_ERR_HIP;
}

// the transient frame behind the host-array call forms
int scratch_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps, const uint8_t* desc, int n, const float bounds[4], orbfe_frame** out) {
  HIP_TRY(hipSetDevice(m->device));
  if (!m->scratch) {
    m->scratch = new orbfe_frame();
    m->scratch->device = m->device;
  }
  const int rc = m->scratch->fromHost(kps, desc, n, bounds, m->stream);
  *out = m->scratch;
  return rc;
}

}  // namespace

namespace orbfe {
bool match_host_resolve() {   // ORBFE_MATCH_HOST_RESOLVE=1: the host-array searches keep the round-2 route (candidate lists to the host, bookkeeping there)
  const char* e = getenv("ORBFE_MATCH_HOST_RESOLVE");   // read per call: the parity tests run both routes in one process
  return e && atoi(e) != 0;
}
}  // namespace orbfe

extern "C" {

int orbfe_frame_create(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                       orbfe_frame** out) {
  if (!m || !out || !bounds || n < 0 || n > 65535 || (n && (!kps_un || !desc))) {
    set_err("bad argument (note: at most 65535 keypoints per frame)");
    return ORBFE_ERR_INVALID;
  }
  *out = nullptr;
  HIP_TRY(hipSetDevice(m->device));
  (void)hipGetLastError();
  orbfe_frame* f = new orbfe_frame();
  f->device = m->device;
  const int rc = f->fromHost(kps_un, desc, n, bounds, m->stream);
  if (rc) { delete f; return rc; }
  *out = f;
  return ORBFE_OK;
}

int orbfe_frame_create_from_extract(orbfe_extractor* h, int frame_index, const float bounds[4], const float* xy_un,
                                    orbfe_frame** out) {
  if (!h || !out || !bounds) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  *out = nullptr;
  orbfe::ExtractView V;
  int rc = orbfe::extractor_view(h, frame_index, &V);
  if (rc) return rc;
  if (V.n > 65535) { set_err("at most 65535 keypoints per frame"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(V.device));
  (void)hipGetLastError();
  hipStream_t st = (hipStream_t)V.stream;
  orbfe_frame* f = new orbfe_frame();
  f->device = V.device;
  if ((rc = f->carve(V.n))) { delete f; return rc; }
  f->setBounds(bounds);
  f->maxOctave = V.nlevels - 1;
  if (V.n > 0) {
    const float* dxy = nullptr;
    if (xy_un) {   // undistorted coordinates: 8 bytes per keypoint, staged in the (not yet filled) grid-order x array
      if ((rc = f->stage.ensure(8 * (size_t)V.n))) { delete f; return rc; }
      memcpy(f->stage.p, xy_un, 8 * (size_t)V.n);
      hipError_t e = hipMemcpyAsync(f->D.tdesc, f->stage.p, 8 * (size_t)V.n, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) { delete f; set_err("hipMemcpyAsync failed: %s", hipGetErrorString(e)); return ORBFE_ERR_HIP; }
      dxy = (const float*)f->D.tdesc;
    }
    ExtractViewDev D;
    D.sel = V.sel; D.angle = V.angle; D.desc = V.desc; D.nlevels = V.nlevels;
    for (int l = 0; l <= orbfe::kMaxLevels; l++) D.selOff[l] = l <= V.nlevels ? V.selOff[l] : 0;
    for (int l = 0; l < orbfe::kMaxLevels; l++) { D.count[l] = l < V.nlevels ? V.count[l] : 0; D.sf[l] = l < V.nlevels ? V.sf[l] : 1.f; }
    hipLaunchKernelGGL(k_frame_from_extract, dim3((V.n + 255) / 256), dim3(256), 0, st, D, dxy, f->D);
  }
  if ((rc = f->grid(st))) { delete f; return rc; }
  *out = f;
  return ORBFE_OK;
}

void orbfe_frame_destroy(orbfe_frame* f) { delete f; }
int orbfe_frame_size(const orbfe_frame* f) { return f ? f->n : 0; }
int orbfe_frame_device(const orbfe_frame* f) { return f ? f->device : -1; }

// generation of the host-object <-> resident-frame association (include/orbfe.h)
static std::atomic<unsigned long long> g_residentEpoch{1};
unsigned long long orbfe_resident_epoch(void) { return g_residentEpoch.load(std::memory_order_acquire); }
unsigned long long orbfe_resident_invalidate(void) { return g_residentEpoch.fetch_add(1, std::memory_order_acq_rel) + 1; }

const uint8_t* orbfe_frame_descriptors_device(orbfe_frame* f) {
  if (!f || !f->n) return nullptr;
  (void)hipSetDevice(f->device);
  if (f->ready && hipEventSynchronize(f->ready) != hipSuccess) (void)hipGetLastError();   // the rows are complete when this returns
  return f->D.desc;
}

int orbfe_frame_download(orbfe_frame* f, OrbfeKeyPoint* kps_un, uint8_t* desc, int32_t* grid_order, int32_t* cell_start) {
  if (!f) { set_err("frame is NULL"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(f->device));
  if (f->ready) HIP_TRY(hipEventSynchronize(f->ready));
  const int n = f->n;
  std::vector<float> x(n), y(n), a(n);
  std::vector<int> o(n);
  if (n) {
    HIP_TRY(hipMemcpy(x.data(), f->D.x, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(y.data(), f->D.y, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(a.data(), f->D.angle, 4 * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(o.data(), f->D.oct, 4 * (size_t)n, hipMemcpyDeviceToHost));
    if (desc) HIP_TRY(hipMemcpy(desc, f->D.desc, 32 * (size_t)n, hipMemcpyDeviceToHost));
  }
  if (kps_un)
    for (int i = 0; i < n; i++) {
      kps_un[i].x = x[i]; kps_un[i].y = y[i]; kps_un[i].angle = a[i]; kps_un[i].octave = o[i];
      kps_un[i].size = 0.f; kps_un[i].response = 0.f; kps_un[i].class_id = -1;
    }
  int cs[kCells + 1];
  HIP_TRY(hipMemcpy(cs, f->D.cellStart, sizeof cs, hipMemcpyDeviceToHost));
  if (cell_start) memcpy(cell_start, cs, sizeof cs);
  if (grid_order && cs[kCells] > 0) HIP_TRY(hipMemcpy(grid_order, f->D.sidx, 4 * (size_t)cs[kCells], hipMemcpyDeviceToHost));
  return ORBFE_OK;
}

int orbfe_debug_resolve_rounds(const orbfe_matcher* m) { return m ? m->lastRounds : 0; }
int orbfe_debug_resolve_route(const orbfe_matcher* m) { return m ? m->lastResolveRoute : -1; }
int orbfe_debug_resolve_phases(const orbfe_matcher* m, int out[4]) {
  if (!m || !out || !m->h_r.p) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  for (int k = 0; k < 4; k++) out[k] = m->h_r.p[16 + k];
  return ORBFE_OK;
}

// int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)  (ORBmatcher.cc:45-132)
// mp_desc_row == nullptr: MapPoint i's descriptor is row i of mp_desc.  Else: row (mp_desc_row[i] & 0x7fffffff) of the device
// table mp_desc (bit 31 clear) or of its page-locked host mirror desc_host (bit 31 set) -- orbfe_search_by_projection_frame_rows.
static int sbp_frame_impl(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                          const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                          const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, const int32_t* mp_desc_row,
                          const uint8_t* desc_host, int n_mp, float th, float nnratio, int32_t* kp_assigned, int* nmatches) {
  const double tEntry = orbfe_matcher::nowMs();
  if (!m || !f || !nmatches || n_mp < 0 || !scale_factors || (f->n && (!kp_occupied || !kp_assigned)) ||
      (n_mp && (!mp_proj_xy || !mp_level || !mp_viewcos || !mp_flags || !mp_desc)) || (mp_desc_row && !desc_host)) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  if (mp_desc_row && !(frame_zero_copy() && nlevels <= 32)) {
    // the marshalled routes take plain rows: gathered from the mirror, which holds every row
    if (gpu_readable(mp_desc_row, m->device) == 2 || gpu_readable(desc_host, m->device) == 2) {
      set_err("descriptor row indices and the table's mirror must be host memory");
      return ORBFE_ERR_INVALID;
    }
    std::vector<uint8_t> rows(32 * (size_t)std::max(n_mp, 1), 0);
    for (int i = 0; i < n_mp; i++)
      if ((mp_flags[i] & ORBFE_MP_IN_VIEW) && !(mp_flags[i] & ORBFE_MP_BAD))
        memcpy(&rows[32 * (size_t)i], desc_host + 32 * (size_t)((unsigned)mp_desc_row[i] & 0x7fffffffu), 32);
    return sbp_frame_impl(m, f, scale_factors, nlevels, kp_occupied, mp_proj_xy, mp_level, mp_viewcos, mp_flags, rows.data(), nullptr,
                          nullptr, n_mp, th, nnratio, kp_assigned, nmatches);
  }
  const int n = f->n;
  *nmatches = 0;
  for (int i = 0; i < n; i++) kp_assigned[i] = -1;
  if (n_mp == 0 || n == 0) return ORBFE_OK;
  m->tEntry = tEntry;
  SearchPlan P;
  int rc = plan_search(m, f, kModeMapPoints, n_mp, true, 0, &P);
  if (rc) return rc;
  const bool bFactor = th != 1.0;
  const int* out = nullptr;
  RawQ Q;
  Q.kind = 1; Q.xy = mp_proj_xy; Q.level = mp_level; Q.aux = mp_viewcos; Q.flags = mp_flags;
  Q.claimSrc = mp_flags; Q.claimMask = ORBFE_MP_OBSERVED;
  Q.occ = kp_occupied; Q.sf = scale_factors; Q.nlevels = nlevels; Q.th = th; Q.factor = bFactor ? 1 : 0;
  Q.descRow = mp_desc_row; Q.descHost = desc_host;
  if ((rc = classify_host_arrays(m, &Q))) return rc;
  if (frame_zero_copy() && nlevels <= 32) {   // no loop over the MapPoints here: the window kernel reads the caller's arrays
    float maxSf = 0.f;
    if ((rc = check_levels(mp_level, mp_flags, ORBFE_MP_IN_VIEW | ORBFE_MP_BAD, ORBFE_MP_IN_VIEW, n_mp, scale_factors, nlevels, &maxSf, "MapPoint")))
      return rc;
    float rmaxRaw = 4.0f;
    if (bFactor) rmaxRaw *= th;
    if ((rc = run_search(m, f, P, mp_desc, rmaxRaw * maxSf, nnratio, TH_HIGH, 0.0, 0, &out, nmatches, &Q))) return rc;
    memcpy(kp_assigned, out, sizeof(int32_t) * (size_t)n);
    m->stageMs[2] = orbfe_matcher::nowMs() - m->tSynced;
    return ORBFE_OK;
  }
  float rmax = 0.f;
  for (int i = 0; i < n_mp; i++) {
    const uint8_t fl = mp_flags[i];
    P.qx[i] = mp_proj_xy[2 * i];
    P.qy[i] = mp_proj_xy[2 * i + 1];
    const int lvl = mp_level[i];
    P.qa[i] = lvl - 1;       // F.GetFeaturesInArea(x, y, r * scale, nPredictedLevel - 1, nPredictedLevel), :72-73
    P.qb[i] = lvl;
    P.qclaim[i] = (fl & ORBFE_MP_OBSERVED) ? 1 : 0;
    P.qangle[i] = 0.f;
    if (!(fl & ORBFE_MP_IN_VIEW) || (fl & ORBFE_MP_BAD)) { P.qr[i] = -1.f; continue; }
    if (lvl < 0 || lvl >= nlevels) { set_err("MapPoint %d: level %d out of range", i, lvl); return ORBFE_ERR_INVALID; }
    float r = (fl & ORBFE_MP_CANDIDATO) ? 4.0 : (mp_viewcos[i] > 0.998 ? 2.5 : 4.0);  // ORBmatcher.cc:63-65, 126-132
    if (bFactor) r *= th;
    P.qr[i] = r * scale_factors[lvl];
    rmax = std::max(rmax, P.qr[i]);
  }
  memcpy(P.occ0, kp_occupied, (size_t)n);
  if ((rc = run_search(m, f, P, mp_desc, rmax, nnratio, TH_HIGH, 0.0, 0, &out, nmatches))) return rc;
  memcpy(kp_assigned, out, sizeof(int32_t) * (size_t)n);
  m->stageMs[2] = orbfe_matcher::nowMs() - m->tSynced;
  return ORBFE_OK;
}

int orbfe_search_by_projection_frame(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                                     const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                                     const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp,
                                     float th, float nnratio, int32_t* kp_assigned, int* nmatches) {
  return sbp_frame_impl(m, f, scale_factors, nlevels, kp_occupied, mp_proj_xy, mp_level, mp_viewcos, mp_flags, mp_desc, nullptr, nullptr,
                        n_mp, th, nnratio, kp_assigned, nmatches);
}
int orbfe_search_by_projection_frame_rows(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                                          const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                                          const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* desc_table_device,
                                          const uint8_t* desc_table_host, const int32_t* mp_desc_row, int n_rows, int n_mp,
                                          float th, float nnratio, int32_t* kp_assigned, int* nmatches) {
  if (!mp_desc_row || !desc_table_host || !desc_table_device || n_rows < 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  if (n_mp > 0 && mp_flags) {
    // a row index is an address: validated like the levels (one branch-free sweep; only if that finds an index outside the table, a
    // second one that looks at the MapPoints that are actually searched -- absent ones may carry anything)
    if (gpu_readable(mp_desc_row, m ? m->device : 0) == 2) { set_err("descriptor row indices must be host memory"); return ORBFE_ERR_INVALID; }
    unsigned hi = 0;
    for (int i = 0; i < n_mp; i++) hi = std::max(hi, (unsigned)mp_desc_row[i] & 0x7fffffffu);
    if (hi >= (unsigned)n_rows)
      for (int i = 0; i < n_mp; i++)
        if ((mp_flags[i] & ORBFE_MP_IN_VIEW) && !(mp_flags[i] & ORBFE_MP_BAD) && ((unsigned)mp_desc_row[i] & 0x7fffffffu) >= (unsigned)n_rows) {
          set_err("MapPoint %d: descriptor row %u outside the table (%d rows)", i, (unsigned)mp_desc_row[i] & 0x7fffffffu, n_rows);
          return ORBFE_ERR_INVALID;
        }
  }
  return sbp_frame_impl(m, f, scale_factors, nlevels, kp_occupied, mp_proj_xy, mp_level, mp_viewcos, mp_flags, desc_table_device,
                        mp_desc_row, desc_table_host, n_mp, th, nnratio, kp_assigned, nmatches);
}

// SearchByProjection(Frame&, const Frame&, th) / (Frame&, KeyFrame*, set, th, ORBdist) from the projection onwards
// (ORBmatcher.cc:1292-1423, 1425-1552)
int orbfe_search_by_projection_uv_frame(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                                        const uint8_t* kp_occupied, const float* src_uv, const int32_t* src_level,
                                        const float* src_angle, const uint8_t* src_flags, const uint8_t* src_valid,
                                        const uint8_t* src_desc, int n_src, float th, int max_dist, int skip_any_occupied,
                                        int check_orientation, int32_t* kp_assigned, int* nmatches) {
  const double tEntry = orbfe_matcher::nowMs();
  if (!m || !f || !nmatches || n_src < 0 || !scale_factors || (f->n && (!kp_occupied || !kp_assigned)) ||
      (n_src && (!src_uv || !src_level || !src_angle || !src_flags || !src_valid || !src_desc))) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  const int n = f->n;
  *nmatches = 0;
  for (int i = 0; i < n; i++) kp_assigned[i] = -1;
  if (n_src == 0 || n == 0) return ORBFE_OK;
  m->tEntry = tEntry;
  SearchPlan P;
  int rc = plan_search(m, f, kModeUv, n_src, true, 0, &P);
  if (rc) return rc;
  const int* out = nullptr;
  RawQ Q;
  Q.kind = 2; Q.xy = src_uv; Q.level = src_level; Q.flags = src_valid; Q.angle = src_angle;
  Q.claimSrc = src_flags; Q.claimMask = ORBFE_MP_OBSERVED;
  Q.occ = kp_occupied; Q.sf = scale_factors; Q.nlevels = nlevels; Q.th = th;
  if ((rc = classify_host_arrays(m, &Q))) return rc;
  if (skip_any_occupied) { Q.claimSrc = nullptr; Q.claimMask = 0xff; Q.claimConst = 1; }
  if (frame_zero_copy() && nlevels <= 32) {
    float maxSf = 0.f;
    if ((rc = check_levels(src_level, src_valid, 0, 0, n_src, scale_factors, nlevels, &maxSf, "source"))) return rc;
    if ((rc = run_search(m, f, P, src_desc, th * maxSf, 0.f, max_dist, 0.0, check_orientation, &out, nmatches, &Q))) return rc;
    memcpy(kp_assigned, out, sizeof(int32_t) * (size_t)n);
    m->stageMs[2] = orbfe_matcher::nowMs() - m->tSynced;
    return ORBFE_OK;
  }
  float rmax = 0.f;
  for (int i = 0; i < n_src; i++) {
    P.qx[i] = src_uv[2 * i];
    P.qy[i] = src_uv[2 * i + 1];
    const int lvl = src_level[i];
    P.qa[i] = lvl - 1;       // GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1), :1356
    P.qb[i] = lvl + 1;
    P.qangle[i] = src_angle[i];
    P.qclaim[i] = skip_any_occupied ? 1 : ((src_flags[i] & ORBFE_MP_OBSERVED) ? 1 : 0);
    if (!src_valid[i]) { P.qr[i] = -1.f; continue; }
    if (lvl < 0 || lvl >= nlevels) { set_err("source %d: level %d out of range", i, lvl); return ORBFE_ERR_INVALID; }
    P.qr[i] = th * scale_factors[lvl];
    rmax = std::max(rmax, P.qr[i]);
  }
  memcpy(P.occ0, kp_occupied, (size_t)n);
  if ((rc = run_search(m, f, P, src_desc, rmax, 0.f, max_dist, 0.0, check_orientation, &out, nmatches))) return rc;
  memcpy(kp_assigned, out, sizeof(int32_t) * (size_t)n);
  m->stageMs[2] = orbfe_matcher::nowMs() - m->tSynced;
  return ORBFE_OK;
}

// the projected best-match loop of SearchByProjection(KeyFrame*, Scw, ...), Fuse x2 and SearchBySim3
// (ORBmatcher.cc:357-392, 872-936, 1014-1050, 1066-1290)
int orbfe_search_projected_frame(orbfe_matcher* m, orbfe_frame* f, int n_src, const float* src_uv, const float* src_radius,
                                 const int32_t* src_level, const uint8_t* src_valid, const uint8_t* src_desc,
                                 const uint8_t* kp_skip, int claim, const float* inv_level_sigma2, int nlevels, double chi2,
                                 int max_dist, int32_t* best_idx, int32_t* best_dist, int* nmatches) {
  const double tEntry = orbfe_matcher::nowMs();
  if (!m || !f || !nmatches || n_src < 0 ||
      (n_src && (!src_uv || !src_radius || !src_level || !src_valid || !src_desc || !best_idx)) ||
      (inv_level_sigma2 && (nlevels < 1 || nlevels > 64))) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  const int n = f->n;
  *nmatches = 0;
  for (int i = 0; i < n_src; i++) {
    best_idx[i] = -1;
    if (best_dist) best_dist[i] = -1;
  }
  if (n_src == 0 || n == 0) return ORBFE_OK;
  if (inv_level_sigma2 && f->maxOctave >= nlevels) { set_err("keypoint octave out of range"); return ORBFE_ERR_INVALID; }
  m->tEntry = tEntry;
  SearchPlan P;
  int rc = plan_search(m, f, kModeProjected, n_src, kp_skip != nullptr, inv_level_sigma2 ? nlevels : 0, &P);
  if (rc) return rc;
  const int* out = nullptr;
  RawQ Q;
  Q.kind = 3; Q.xy = src_uv; Q.level = src_level; Q.aux = src_radius; Q.flags = src_valid;
  Q.claimConst = claim ? 1 : 0;
  Q.occ = kp_skip;
  if ((rc = classify_host_arrays(m, &Q))) return rc;
  if (frame_zero_copy()) {
    float rmaxRaw = 0.f;
    for (int i = 0; i < n_src; i++) rmaxRaw = std::max(rmaxRaw, src_radius[i]);   // (of all sources: an upper bound)
    if (inv_level_sigma2) memcpy(P.invSigma2, inv_level_sigma2, sizeof(float) * (size_t)nlevels);
    if ((rc = run_search(m, f, P, src_desc, rmaxRaw, 0.f, max_dist, chi2, 0, &out, nmatches, &Q))) return rc;
    memcpy(best_idx, out, sizeof(int32_t) * (size_t)n_src);
    if (best_dist) memcpy(best_dist, out + n_src, sizeof(int32_t) * (size_t)n_src);
    return ORBFE_OK;
  }
  float rmax = 0.f;
  for (int i = 0; i < n_src; i++) {
    P.qx[i] = src_uv[2 * i];
    P.qy[i] = src_uv[2 * i + 1];
    P.qa[i] = src_level[i] - 1;                      // kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel => skip
    P.qb[i] = src_level[i];
    P.qangle[i] = 0.f;
    P.qclaim[i] = claim ? 1 : 0;
    P.qr[i] = (src_valid[i] && src_level[i] >= 0) ? src_radius[i] : -1.f;   // no octave lies in [level-1, level] for level < 0
    rmax = std::max(rmax, P.qr[i]);
  }
  if (kp_skip) memcpy(P.occ0, kp_skip, (size_t)n);
  if (inv_level_sigma2) memcpy(P.invSigma2, inv_level_sigma2, sizeof(float) * (size_t)nlevels);
  if ((rc = run_search(m, f, P, src_desc, rmax, 0.f, max_dist, chi2, 0, &out, nmatches))) return rc;
  memcpy(best_idx, out, sizeof(int32_t) * (size_t)n_src);
  if (best_dist) memcpy(best_dist, out + n_src, sizeof(int32_t) * (size_t)n_src);
  return ORBFE_OK;
}

}  // extern "C"

// host-array forms (orbfe_matcher.hip) route here: a transient frame owned by the matcher, then the search above
namespace orbfe {
int sbp_via_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                  const float* scale_factors, int nlevels, const uint8_t* kp_occupied, const float* mp_proj_xy,
                  const int32_t* mp_level, const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp,
                  float th, float nnratio, int32_t* kp_assigned, int* nmatches) {
  orbfe_frame* f = nullptr;
  const int rc = scratch_frame(m, kps_un, desc, n, bounds, &f);
  if (rc) return rc;
  return orbfe_search_by_projection_frame(m, f, scale_factors, nlevels, kp_occupied, mp_proj_xy, mp_level, mp_viewcos, mp_flags,
                                          mp_desc, n_mp, th, nnratio, kp_assigned, nmatches);
}
int sbp_uv_via_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                     const float* scale_factors, int nlevels, const uint8_t* kp_occupied, const float* src_uv,
                     const int32_t* src_level, const float* src_angle, const uint8_t* src_flags, const uint8_t* src_valid,
                     const uint8_t* src_desc, int n_src, float th, int max_dist, int skip_any_occupied, int check_orientation,
                     int32_t* kp_assigned, int* nmatches) {
  orbfe_frame* f = nullptr;
  const int rc = scratch_frame(m, kps_un, desc, n, bounds, &f);
  if (rc) return rc;
  return orbfe_search_by_projection_uv_frame(m, f, scale_factors, nlevels, kp_occupied, src_uv, src_level, src_angle, src_flags,
                                             src_valid, src_desc, n_src, th, max_dist, skip_any_occupied, check_orientation,
                                             kp_assigned, nmatches);
}
int projected_via_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4], int n_src,
                        const float* src_uv, const float* src_radius, const int32_t* src_level, const uint8_t* src_valid,
                        const uint8_t* src_desc, const uint8_t* kp_skip, int claim, const float* inv_level_sigma2, int nlevels,
                        double chi2, int max_dist, int32_t* best_idx, int32_t* best_dist, int* nmatches) {
  orbfe_frame* f = nullptr;
  const int rc = scratch_frame(m, kps_un, desc, n, bounds, &f);
  if (rc) return rc;
  return orbfe_search_projected_frame(m, f, n_src, src_uv, src_radius, src_level, src_valid, src_desc, kp_skip, claim,
                                      inv_level_sigma2, nlevels, chi2, max_dist, best_idx, best_dist, nmatches);
}
}  // namespace orbfe
