// orbfe_bow.hip -- bag-of-words side of the path:
//   * Frame::ComputeBoW (Frame.cc:277-284) = DBoW2 TemplatedVocabulary<FORB>::transform(features, BowVector&,
//     FeatureVector&, levelsup) -- Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1204, descent :1306-1347,
//     BowVector.cpp:30-82, FeatureVector.cpp:27-41, FORB::distance FORB.cpp:81-101;
//   * ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) ORBmatcher.cc:154-283 and (KeyFrame*, KeyFrame*, ...) :517-650.
// C ABI: include/orbfe.h (bag-of-words section).
//
// Split of work.  The data-parallel part runs on the GPU: the k-ary tree descent of every descriptor (16 lanes per
// descriptor, one lane per child, 256-bit Hamming per lane, DPP row minimum with "first minimum wins"), and the
// per-node best / second-best Hamming search of SearchByBoW (one wave per common vocabulary node; the sequential
// "already matched" dependency lives inside one node group, so groups are independent).  What is left is
// bookkeeping in the order the reference does it, on a few thousand scalars: summing word weights in feature order
// (double, so the order is part of the result), the L1/L2 normalisation in ascending word order, grouping feature
// indices by node, and the rotation-histogram pruning.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/orbfe.h"

namespace orbfe {
void set_err(const char* fmt, ...);
int matcher_device(const orbfe_matcher* m);
hipStream_t matcher_stream(const orbfe_matcher* m);
std::shared_ptr<void>& matcher_bow_slot(orbfe_matcher* m);
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

namespace {

constexpr int TH_LOW = 50, HISTO_LENGTH = 30;   // ORBmatcher.cc:37-39
constexpr int kRecord = 45;                     // bytes per node record of the binary vocabulary file
constexpr int kMaxGroup = 65535;                // SearchByBoW: features of one frame under one vocabulary node

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n) return ORBFE_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
    return ORBFE_OK;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
};
template <class T>
struct PinBuf {
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n) return ORBFE_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
    // Coherent (fine-grained, uncached on the GPU side) EXPLICITLY: kernels store results and completion words here and the
    // host polls them while the kernel runs; with hipHostMallocDefault that property would hang on HIP_HOST_COHERENT.
    HIP_TRY(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocCoherent));
    n = count;
    return ORBFE_OK;
  }
  ~PinBuf() { if (p) (void)hipHostFree(p); }
};

// row_ror:n -- rotate right by n lanes inside each row of 16 lanes
template <int N>
__device__ inline unsigned row_ror(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x120 + N, 0xf, 0xf, false);
}
// minimum over the 16 lanes of a DPP row, in every lane of the row
__device__ inline unsigned row_min16(unsigned v) {
  v = min(v, row_ror<8>(v));
  v = min(v, row_ror<4>(v));
  v = min(v, row_ror<2>(v));
  v = min(v, row_ror<1>(v));
  return v;
}
// minimum over the 64 lanes of the wave (wave-uniform result)
__device__ inline unsigned wave_min(unsigned v) {
  v = row_min16(v);
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

__device__ inline int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Tree descent (TemplatedVocabulary.h:1306-1347).  16 lanes per descriptor; lane c compares child c, c+16, ... of the
// current node; key = distance << 8 | child position, so the row minimum is the first child with the least distance
// (`d < best_d` strict).  Children of a node are contiguous "slots" (descriptor + node id) in the order the reference
// appends them.  out[f] = (leaf node id, node id at level nidLevel or 0).
__global__ void __launch_bounds__(256) k_bow_descend(const uint4* __restrict__ desc, int n, const int* __restrict__ childBegin,
                                                     const int* __restrict__ childCount, const int* __restrict__ slotNode,
                                                     const uint4* __restrict__ slotDesc, int nidLevel, int maxDepth,
                                                     uint2* __restrict__ out) {
  const int lane16 = threadIdx.x & 15;
  const int f = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  const bool live = f < n;
  uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0;
  if (live) { f0 = desc[2 * (size_t)f]; f1 = desc[2 * (size_t)f + 1]; }
  int cur = 0;
  unsigned nid = 0;
  bool done = !live;
  for (int level = 1; level <= maxDepth; level++) {   // uniform trip count: the DPP reduction needs every lane
    unsigned key = 0xffffffffu;
    int cb = 0;
    if (!done) {
      cb = childBegin[cur];
      const int cc = childCount[cur];
      for (int c = lane16; c < cc; c += 16) {
        const uint4 d0 = slotDesc[2 * (size_t)(cb + c)], d1 = slotDesc[2 * (size_t)(cb + c) + 1];
        key = min(key, ((unsigned)hamming256(f0, f1, d0, d1) << 8) | (unsigned)c);
      }
    }
    key = row_min16(key);
    if (!done) {
      cur = slotNode[cb + (int)(key & 255u)];
      if (level == nidLevel) nid = (unsigned)cur;
      done = childCount[cur] == 0;
    }
  }
  if (live && lane16 == 0) out[f] = make_uint2((unsigned)cur, nid);
}

// SearchByBoW inner search: one wave per vocabulary node common to both FeatureVectors.  Frame-1 features of the node
// are visited in order; for each, lanes scan the node's frame-2 features (skipping matched / invalid ones), the wave
// takes best (first minimum) and second-best distance, applies the reference's acceptance test and marks the winner
// matched (LDS bitmap) before the next frame-1 feature.
struct BowPair { int b1, e1, b2, e2, base1, tk; };   // [b,e) ranges into fv1_feat / fv2_feat; base1 = first descriptor row of side 1's
                                                   // frame in desc1 / valid1 / matches12 (several keyframes against one frame in one launch)

// LARGE NODES (round 4): more than kBowSide features of either frame under one vocabulary node -- a vocabulary with no more
// levels than `levelsup` puts EVERY feature under the root; low-texture frames pile features into few words.  The reference's
// loop over the node's frame-1 features is sequential only through "which frame-2 feature is already matched"
// (ORBmatcher.cc:196-222): feature i takes the least-distance frame-2 feature that is still free (first on ties) if it is
// <= TH_LOW and passes the ratio test against the next least free one.  So k_bow_topk computes, for every frame-1 feature of
// a large node and in parallel over the whole chip, its kBowTopK least keys (distance << 16 | position in the node's frame-2
// list) in ascending order; the walk (one wave, in k_bow_match) then decides feature after feature from those eight keys and
// the matched bitmap -- the first two free keys are best and second best; only a feature that finds its list exhausted
// (seven of its eight nearest already taken and the ratio test not decided by the eighth) rescans the node.  2 000 x 2 000
// features under one node: 6.8 ms -> 0.25 ms.
constexpr int kBowTopK = 8;
constexpr int kTopkRows = 8;          // frame-1 features per block of k_bow_topk (two per wave)
constexpr int kTopkRegs = 32;         // keys a lane keeps in registers (nodes of up to 2 048 frame-2 features)
constexpr int kTopkLdsFeatures = 4096; // frame-2 features of a node staged in LDS (36 bytes each); larger nodes read them from memory
struct TopkItem { int pair, row0; };
__global__ void __launch_bounds__(256) k_bow_topk(const uint4* __restrict__ desc1, const uint8_t* __restrict__ valid1,
                                                  const uint32_t* __restrict__ feat1, const uint4* __restrict__ desc2,
                                                  const uint8_t* __restrict__ valid2, const uint32_t* __restrict__ feat2,
                                                  const BowPair* __restrict__ pairs, const TopkItem* __restrict__ items,
                                                  uint32_t* __restrict__ topk) {
  extern __shared__ uint4 ldsTopk[];   // [2 * nStage] descriptors, then [nStage] row words (| 0x80000000: not valid)
  const TopkItem it = items[blockIdx.x];
  const BowPair P = pairs[it.pair];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n1g = P.e1 - P.b1, n2g = P.e2 - P.b2;
  const int nStage = min(n2g, kTopkLdsFeatures);
  uint32_t* rowS = reinterpret_cast<uint32_t*>(ldsTopk + 2 * nStage);
  for (int p = tid; p < nStage; p += 256) {
    const unsigned idx2 = feat2[P.b2 + p];
    rowS[p] = idx2 | ((valid2 && !valid2[idx2]) ? 0x80000000u : 0u);
    ldsTopk[2 * p] = desc2[2 * (size_t)idx2];
    ldsTopk[2 * p + 1] = desc2[2 * (size_t)idx2 + 1];
  }
  __syncthreads();
  for (int r = it.row0 + wave; r < min(n1g, it.row0 + kTopkRows); r += 4) {
    uint32_t* out = topk + ((size_t)P.tk + r) * kBowTopK;
    const unsigned idx1 = (unsigned)P.base1 + feat1[P.b1 + r];
    if (!valid1[idx1]) {   // wave-uniform: the walk skips the feature, its list is never read
      if (lane < kBowTopK) out[lane] = 0xffffffffu;
      continue;
    }
    const uint4 a0 = desc1[2 * (size_t)idx1], a1 = desc1[2 * (size_t)idx1 + 1];
    auto keyAt = [&](int p) -> unsigned {   // 0xffffffff: not a candidate
      if (p < nStage) {
        if (rowS[p] & 0x80000000u) return 0xffffffffu;
        return ((unsigned)hamming256(a0, a1, ldsTopk[2 * p], ldsTopk[2 * p + 1]) << 16) | (unsigned)p;
      }
      const unsigned idx2 = feat2[P.b2 + p];
      if (valid2 && !valid2[idx2]) return 0xffffffffu;
      return ((unsigned)hamming256(a0, a1, desc2[2 * (size_t)idx2], desc2[2 * (size_t)idx2 + 1]) << 16) | (unsigned)p;
    };
    unsigned lo = 0;           // keys below `lo` are already in the list
    if (n2g <= 64 * kTopkRegs) {
      // the lane's (up to 32) keys once, in registers; the eight selection passes then cost three instructions per key
      unsigned kreg[kTopkRegs];
#pragma unroll
      for (int j = 0; j < kTopkRegs; j++) kreg[j] = (lane + 64 * j < n2g) ? keyAt(lane + 64 * j) : 0xffffffffu;
      for (int k = 0; k < kBowTopK; k++) {
        unsigned cur = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < kTopkRegs; j++) cur = (kreg[j] >= lo && kreg[j] < cur) ? kreg[j] : cur;
        cur = wave_min(cur);
        if (lane == 0) out[k] = cur;
        if (cur == 0xffffffffu) {
          if (lane > k && lane < kBowTopK) out[lane] = 0xffffffffu;
          break;
        }
        lo = cur + 1u;
      }
      continue;
    }
    for (int k = 0; k < kBowTopK; k++) {
      unsigned cur = 0xffffffffu;
      for (int p = lane; p < n2g; p += 64) {
        const unsigned key = keyAt(p);
        if (key >= lo && key < cur) cur = key;
      }
      cur = wave_min(cur);
      if (lane == 0) out[k] = cur;
      if (cur == 0xffffffffu) {   // fewer than kBowTopK valid frame-2 features: the rest of the list is empty
        if (lane > k && lane < kBowTopK) out[lane] = 0xffffffffu;
        break;
      }
      lo = cur + 1u;
    }
  }
}

// Round 3: the distances of a node's pairs do not depend on the bookkeeping, so they are computed first, all pairs in
// parallel by the block's four waves, into an LDS matrix (nodes of up to kBowMatrix pairs; larger ones compute them inside
// the walk as before); the sequential walk over the node's frame-1 features -- which F feature is already matched decides
// what the next one may take, ORBmatcher.cc:196-222 -- then costs one LDS row and two wave-wide minima per feature instead
// of a chain of global loads (84 -> 12 us for two 2 000-feature frames over 100 nodes).
constexpr int kBowThreads = 256;
constexpr int kBowMatrix = 12288;
constexpr int kBowSide = 256;      // features of one frame under a node handled from LDS (lists, descriptors, distances)
__global__ void __launch_bounds__(kBowThreads) k_bow_match(const uint4* __restrict__ desc1, const uint8_t* __restrict__ valid1,
                                                           const uint32_t* __restrict__ feat1, const uint4* __restrict__ desc2,
                                                           const uint8_t* __restrict__ valid2, const uint32_t* __restrict__ feat2,
                                                           const BowPair* __restrict__ pairs, int maxDist, float nnratio,
                                                           int32_t* __restrict__ matches12, unsigned* doneCounter, int* doneHost,
                                                           int doneSeq, const uint32_t* __restrict__ topk) {
  __shared__ unsigned matched[(kMaxGroup + 1) / 32];
  __shared__ uint16_t dmat[kBowMatrix];
  __shared__ uint32_t row1[kBowSide], row2[kBowSide];   // descriptor rows of the node's features; | 0x80000000: not valid
  __shared__ uint4 d1s[2 * kBowSide], d2s[2 * kBowSide];   // their descriptors
  const BowPair P = pairs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n1g = P.e1 - P.b1, n2g = P.e2 - P.b2;
  for (int i = tid; i < (n2g + 31) / 32; i += kBowThreads) matched[i] = 0;
  const bool inLds = n1g <= kBowSide && n2g <= kBowSide;
  // completion word (the route without copy commands: inputs read from, matches written to page-locked host memory): the
  // block's stores are acknowledged, it counts itself off, the last one writes the call's number for the polling host.
  // Why an acknowledgement wait is enough for the blocks that are not last: matches12 / doneHost live in COHERENT page-locked
  // memory (PinBuf: hipHostMallocCoherent), which the GPU maps uncached -- a store that has been acknowledged (vmcnt 0) has
  // left every XCD's L2 and is visible to the host; the counter only orders "all blocks got that far" before the flag.
  auto done = [&]() {
    if (!doneCounter || tid != 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (atomicAdd(doneCounter, 1u) != gridDim.x - 1u) return;
    *doneCounter = 0u;
    __threadfence_system();
    __hip_atomic_store(doneHost, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (inLds) {
    // the node's feature lists and descriptors first -- the walk below must not chase global pointers: every dependent
    // load there is a microsecond per frame-1 feature --
    for (int i = tid; i < n1g; i += kBowThreads) {
      const unsigned idx1 = (unsigned)P.base1 + feat1[P.b1 + i];
      row1[i] = idx1 | (valid1[idx1] ? 0u : 0x80000000u);
      d1s[2 * i] = desc1[2 * (size_t)idx1]; d1s[2 * i + 1] = desc1[2 * (size_t)idx1 + 1];
    }
    for (int i = tid; i < n2g; i += kBowThreads) {
      const unsigned idx2 = feat2[P.b2 + i];
      row2[i] = idx2 | ((valid2 && !valid2[idx2]) ? 0x80000000u : 0u);
      d2s[2 * i] = desc2[2 * (size_t)idx2]; d2s[2 * i + 1] = desc2[2 * (size_t)idx2 + 1];
    }
    __syncthreads();
    // ... then, for as many frame-1 features at a time as the matrix holds: the distances of all their pairs by the whole
    // block, the walk over them by the first wave
    const int rowsPer = min(kBowThreads, max(1, kBowMatrix / n2g));
    for (int r0 = 0; r0 < n1g; r0 += rowsPer) {
      const int rows = min(rowsPer, n1g - r0);
      int r = tid / n2g, p = tid - r * n2g;
      for (int t = tid; t < rows * n2g; t += kBowThreads) {
        int d = 256;   // (a pair that the walk skips: it can neither win nor pull the second-best below its initial 256)
        if (!((row1[r0 + r] | row2[p]) & 0x80000000u)) d = hamming256(d1s[2 * (r0 + r)], d1s[2 * (r0 + r) + 1], d2s[2 * p], d2s[2 * p + 1]);
        dmat[t] = (uint16_t)d;
        p += kBowThreads;
        while (p >= n2g) { p -= n2g; r++; }
      }
      __syncthreads();
      if (tid < 64) {   // (one wave: its LDS operations execute in order)
        // a lane's (up to four) distances of the NEXT feature's row and its row word travel while the current feature is decided
        constexpr int kPer = kBowSide / 64;
        unsigned dn[kPer], idxN = row1[r0];
        // (the matched flags of a lane's own positions -- pp = lane + 64 j -- as bits of a register: the walk's only
        // dependent LDS trip per feature was reading them back; the LDS bitmap is kept for the next tile)
        unsigned mineTaken = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
          const int pp = lane + 64 * j;
          dn[j] = pp < n2g ? dmat[pp] : 256u;
          if (pp < n2g && ((matched[pp >> 5] >> (pp & 31)) & 1u)) mineTaken |= 1u << j;
        }
        for (int rr = 0; rr < rows; rr++) {
          unsigned dc[kPer];
          const unsigned idx1 = idxN;
#pragma unroll
          for (int j = 0; j < kPer; j++) dc[j] = dn[j];
          if (rr + 1 < rows) {
            idxN = row1[r0 + rr + 1];
            const uint16_t* nrow = dmat + (rr + 1) * n2g;
#pragma unroll
            for (int j = 0; j < kPer; j++) dn[j] = (lane + 64 * j < n2g) ? nrow[lane + 64 * j] : 256u;
          }
          if (idx1 & 0x80000000u) continue;   // wave-uniform
          unsigned k1 = (256u << 16) | 0xffffu, k2 = k1;   // lane-local least and second-least key (distance << 16 | position)
#pragma unroll
          for (int j = 0; j < kPer; j++) {
            if (dc[j] >= 256u || ((mineTaken >> j) & 1u)) continue;
            const unsigned key = (dc[j] << 16) | (unsigned)(lane + 64 * j);
            if (key < k1) { k2 = k1; k1 = key; }
            else if (key < k2) k2 = key;
          }
          const unsigned best = wave_min(k1);
          const int bestDist = (int)(best >> 16);
          if (bestDist > maxDist) continue;   // (wave-uniform; most features end here: the second-best is not needed)
          const unsigned second = wave_min(k1 == best ? k2 : k1) >> 16;   // least distance among everything but the winner
          if (static_cast<float>(bestDist) < nnratio * static_cast<float>((int)second)) {
            const int q = (int)(best & 0xffffu);
            if ((q & 63) == lane) mineTaken |= 1u << (q >> 6);
            if (lane == 0) {
              matches12[idx1] = (int32_t)row2[q];
              atomicOr(&matched[q >> 5], 1u << (q & 31));
            }
          }
        }
      }
      __syncthreads();
    }
    done();
    return;
  }
  __syncthreads();
  // larger nodes: the walk by ONE wave over the lists k_bow_topk prepared (see there).  Eight frame-1 features at a time: lane
  // (f, k) = (lane >> 3, lane & 7) holds key k of feature f.  ONE read of the matched bitmap per group gives every lane its
  // candidate's state before the group; inside the group a feature that takes candidate q marks the later lanes that hold q
  // (a compare, no LDS trip), so the eight decisions cost a ballot, two lane reads and a few scalar operations each.  The
  // winners' positions go to LDS; frame-2 indices are looked up and matches12 written by all lanes afterwards.  (A feature
  // without a MapPoint has an all-empty list -- k_bow_topk wrote it -- and decides "no match" by itself.)
  if (tid >= 64) { done(); return; }
  const uint32_t* tkRows = topk + (size_t)P.tk * kBowTopK;
  uint16_t* sel = dmat;   // [n1g] winner position + 1, 0 = none (n1g <= kMaxGroup < 2^16; dmat is free on this path: 12 288 entries ...
  int32_t* selBig = nullptr;   // ... larger nodes keep the winners in matches12 itself, as positions, and translate in place)
  const bool selInLds = n1g <= kBowMatrix;
  for (int i = lane; i < n1g && selInLds; i += 64) sel[i] = 0;
  unsigned keyN = (lane >> 3) < n1g ? tkRows[lane] : 0xffffffffu;   // keys of the first group
  for (int i0 = 0; i0 < n1g; i0 += 8) {
    const int f = lane >> 3;
    const unsigned key = keyN;
    if (i0 + 8 < n1g) {   // the next group's keys travel while this one is decided
      const int rowN = i0 + 8 + f;
      keyN = rowN < n1g ? tkRows[(size_t)rowN * kBowTopK + (lane & 7)] : 0xffffffffu;
    }
    const unsigned q = key & 0xffffu;
    bool free_ = key != 0xffffffffu && !((matched[(q >> 5) & 2047u] >> (q & 31)) & 1u);   // (index masked: an empty key reads word 2047, harmless)
    const int nf = min(8, n1g - i0);
    {
      // ALL EIGHT decisions at once, as if the group's features did not interfere -- then a check that they do not: a feature's
      // outcome depends on its first two free candidates only, so the parallel outcomes are the sequential ones unless some
      // feature of the group accepts a candidate that is another feature's first or second free one (or a feature's list is
      // exhausted).  Such groups -- rare outside near-duplicate descriptors -- take the one-by-one loop below.
      const unsigned long long fb = __ballot(free_);
      const unsigned m8 = (unsigned)(fb >> (8 * f)) & 0xffu;
      const unsigned m2 = m8 & (m8 - 1u);
      const int kb = m8 ? __builtin_ctz(m8) : 0, k2i = m2 ? __builtin_ctz(m2) : 0;
      const unsigned last = (unsigned)__shfl((int)key, 8 * f + 7);
      const unsigned kbest = (unsigned)__shfl((int)key, 8 * f + kb), ksec = (unsigned)__shfl((int)key, 8 * f + k2i);
      const bool none = m8 == 0 && last == 0xffffffffu;
      const int bestDist = (int)(kbest >> 16), bqp = (int)(kbest & 0xffffu);
      int second = 256;
      bool decided = none;
      if (m8) {
        if (bestDist > maxDist) decided = true;
        else if (m2) { second = (int)(ksec >> 16); decided = true; }
        else if (last == 0xffffffffu) decided = true;
        else if (static_cast<float>(bestDist) < nnratio * static_cast<float>((int)(last >> 16))) { second = (int)(last >> 16); decided = true; }
      }
      const bool accept = decided && m8 && bestDist <= maxDist && static_cast<float>(bestDist) < nnratio * static_cast<float>(second);
      const bool rel = free_ && ((lane & 7) == kb || (m2 && (lane & 7) == k2i));   // this lane's key is its feature's first / second free one
      bool conflict = !decided;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const int accG = __builtin_amdgcn_readlane(accept ? 1 : 0, 8 * g), bqG = __builtin_amdgcn_readlane(bqp, 8 * g);
        if (accG && g != f && rel && q == (unsigned)bqG) conflict = true;
      }
      if (__ballot(conflict) == 0ull) {
        if ((lane & 7) == 0 && accept) {
          atomicOr(&matched[bqp >> 5], 1u << (bqp & 31));
          if (selInLds) sel[i0 + f] = (uint16_t)(bqp + 1);
          else matches12[(unsigned)P.base1 + feat1[P.b1 + i0 + f]] = (int32_t)feat2[P.b2 + bqp];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        continue;
      }
    }
    for (int ff = 0; ff < nf; ff++) {
      const unsigned m8 = (unsigned)((__ballot(free_) >> (8 * ff)) & 0xffull);
      const unsigned last = (unsigned)__builtin_amdgcn_readlane((int)key, 8 * ff + 7);   // the list's largest key (0xffffffff: list not full)
      if (m8 == 0 && last == 0xffffffffu) continue;   // nothing free and nothing beyond the list (also: a feature without a MapPoint)
      int bestDist = 256, second = 256, bq = -1;
      bool decided = false;
      if (m8) {
        const int kb = __builtin_ctz(m8);
        const unsigned kbest = (unsigned)__builtin_amdgcn_readlane((int)key, 8 * ff + kb);
        bestDist = (int)(kbest >> 16);
        bq = (int)(kbest & 0xffffu);
        const unsigned m2 = m8 & (m8 - 1u);
        if (bestDist > maxDist) decided = true;   // the best free candidate of the list IS the node's best free one: nothing to accept
        else if (m2) {
          second = (int)((unsigned)__builtin_amdgcn_readlane((int)key, 8 * ff + __builtin_ctz(m2)) >> 16);
          decided = true;
        } else if (last == 0xffffffffu) {
          decided = true;                          // the list holds every valid candidate: no second one is free (bestDist2 stays 256)
        } else if (static_cast<float>(bestDist) < nnratio * static_cast<float>((int)(last >> 16))) {
          second = (int)(last >> 16);              // the true second best is at least the list's largest: the ratio test passes either way
          decided = true;
        }
      }
      if (!decided) {
        // the list is exhausted: rescan the node for this feature (rare: seven of its eight nearest taken already); the bitmap
        // in LDS is current up to the previous GROUP, this group's winners so far are in `sel`
        const int row = i0 + ff;
        const unsigned idx1 = (unsigned)P.base1 + feat1[P.b1 + row];
        const uint4 a0 = desc1[2 * (size_t)idx1], a1 = desc1[2 * (size_t)idx1 + 1];
        unsigned k1 = (256u << 16) | 0xffffu, k2 = k1;
        for (int p = lane; p < n2g; p += 64) {
          const unsigned idx2 = feat2[P.b2 + p];
          if (((matched[p >> 5] >> (p & 31)) & 1u) || (valid2 && !valid2[idx2])) continue;
          const unsigned kk = ((unsigned)hamming256(a0, a1, desc2[2 * (size_t)idx2], desc2[2 * (size_t)idx2 + 1]) << 16) | (unsigned)p;
          if (kk < k1) { k2 = k1; k1 = kk; }
          else if (kk < k2) k2 = kk;
        }
        const unsigned best = wave_min(k1);
        second = (int)(wave_min(k1 == best ? k2 : k1) >> 16);
        bestDist = (int)(best >> 16);
        bq = (int)(best & 0xffffu);
      }
      if (bestDist <= maxDist && static_cast<float>(bestDist) < nnratio * static_cast<float>(second)) {
        if (free_ && q == (unsigned)bq) free_ = false;   // later features of this group that list the same candidate
        if (lane == 0) {
          matched[bq >> 5] |= 1u << (bq & 31);          // (visible to the rescans of this group and to the next group's read)
          if (selInLds) sel[i0 + ff] = (uint16_t)(bq + 1);
          else matches12[(unsigned)P.base1 + feat1[P.b1 + i0 + ff]] = (int32_t)feat2[P.b2 + bq];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the group's bits are in LDS before the next group reads the bitmap
  }
  if (selInLds)
    for (int i = lane; i < n1g; i += 64) {
      const unsigned w = sel[i];
      if (w) matches12[(unsigned)P.base1 + feat1[P.b1 + i]] = (int32_t)feat2[P.b2 + (int)w - 1];
    }
  (void)selBig;
  done();
}

// SearchForTriangulation inner search (ORBmatcher.cc:695-747): the reference never sets vbMatched2, so every
// frame-1 feature is independent: among the node's frame-2 features without a MapPoint, with distance <= TH_LOW, far
// enough from the epipole and close enough to the epipolar line, take the least distance -- the LAST one on ties
// (`dist > bestDist` is the skip test).  key = distance << 16 | (0xffff - position).
struct TriParams {
  float F12[9];
  float ex, ey;
  float scale2[16], sigma2[16];
};

// Round 3: a block of four waves per node.  Both feature lists with their flags, descriptors and keypoint coordinates are
// fetched once into LDS (nodes of up to kBowSide features per side; from the page-locked arena directly on the route
// without copy commands); the frame-1 features are independent (see above), so the four waves take every fourth one and
// scan the node's frame-2 features from LDS.  Larger nodes: the same split, everything from global memory.
__global__ void __launch_bounds__(kBowThreads) k_bow_triangulate(const OrbfeKeyPoint* __restrict__ kps1, const uint4* __restrict__ desc1,
                                                                 const uint8_t* __restrict__ hasMP1, const uint32_t* __restrict__ feat1,
                                                                 const OrbfeKeyPoint* __restrict__ kps2, const uint4* __restrict__ desc2,
                                                                 const uint8_t* __restrict__ hasMP2, const uint32_t* __restrict__ feat2,
                                                                 const BowPair* __restrict__ pairs, TriParams T,
                                                                 int32_t* __restrict__ matches12, unsigned* doneCounter, int* doneHost,
                                                                 int doneSeq) {
  __shared__ uint4 d1s[2 * kBowSide], d2s[2 * kBowSide];
  __shared__ float x1s[kBowSide], y1s[kBowSide], x2s[kBowSide], y2s[kBowSide];
  __shared__ uint32_t row1[kBowSide], row2[kBowSide];   // descriptor row | 0x80000000: has a MapPoint (skipped)
  __shared__ int o2s[kBowSide];
  const BowPair P = pairs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n1g = P.e1 - P.b1, n2g = P.e2 - P.b2;
  auto done = [&]() {   // (every wave's stores are acknowledged before the block counts itself off)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!doneCounter || tid != 0) return;
    if (atomicAdd(doneCounter, 1u) != gridDim.x - 1u) return;
    *doneCounter = 0u;
    __threadfence_system();
    __hip_atomic_store(doneHost, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (n1g <= kBowSide && n2g <= kBowSide) {
    for (int i = tid; i < n1g; i += kBowThreads) {
      const unsigned idx1 = feat1[P.b1 + i];
      row1[i] = idx1 | (hasMP1[idx1] ? 0x80000000u : 0u);
      d1s[2 * i] = desc1[2 * (size_t)idx1]; d1s[2 * i + 1] = desc1[2 * (size_t)idx1 + 1];
      x1s[i] = kps1[idx1].x; y1s[i] = kps1[idx1].y;
    }
    for (int i = tid; i < n2g; i += kBowThreads) {
      const unsigned idx2 = feat2[P.b2 + i];
      row2[i] = idx2 | (hasMP2[idx2] ? 0x80000000u : 0u);
      d2s[2 * i] = desc2[2 * (size_t)idx2]; d2s[2 * i + 1] = desc2[2 * (size_t)idx2 + 1];
      x2s[i] = kps2[idx2].x; y2s[i] = kps2[idx2].y; o2s[i] = kps2[idx2].octave;
    }
    __syncthreads();
    for (int r = wave; r < n1g; r += kBowThreads / 64) {
      const unsigned idx1 = row1[r];
      if (idx1 & 0x80000000u) continue;   // wave-uniform
      const uint4 a0 = d1s[2 * r], a1 = d1s[2 * r + 1];
      const float x1 = x1s[r], y1 = y1s[r];
      // epipolar line in image 2, l = x1' F12 (ORBmatcher.cc:138-140)
      const float la = x1 * T.F12[0] + y1 * T.F12[3] + T.F12[6];
      const float lb = x1 * T.F12[1] + y1 * T.F12[4] + T.F12[7];
      const float lc = x1 * T.F12[2] + y1 * T.F12[5] + T.F12[8];
      const float den = la * la + lb * lb;
      unsigned key = 0xffffffffu;
      for (int p = lane; p < n2g; p += 64) {
        if (row2[p] & 0x80000000u) continue;
        const int dist = hamming256(a0, a1, d2s[2 * p], d2s[2 * p + 1]);
        if (dist > TH_LOW) continue;
        const float x2 = x2s[p], y2 = y2s[p];
        const int oct = o2s[p];
        const float distex = T.ex - x2, distey = T.ey - y2;
        if (distex * distex + distey * distey < 100 * T.scale2[oct]) continue;
        const float num = la * x2 + lb * y2 + lc;
        if (den == 0) continue;
        const float dsqr = num * num / den;
        if (!((double)dsqr < 3.84 * (double)T.sigma2[oct])) continue;
        key = min(key, ((unsigned)dist << 16) | (0xffffu - (unsigned)p));
      }
      key = wave_min(key);
      if (key != 0xffffffffu && lane == 0) matches12[idx1] = (int32_t)row2[(int)(0xffffu - (key & 0xffffu))];
    }
    done();
    return;
  }
  {   // (a node too large for LDS: the four waves still share its frame-1 features, everything comes from global memory)
    for (int i1 = P.b1 + wave; i1 < P.e1; i1 += kBowThreads / 64) {
      const unsigned idx1 = feat1[i1];
      if (hasMP1[idx1]) continue;   // wave-uniform
      const uint4 a0 = desc1[2 * (size_t)idx1], a1 = desc1[2 * (size_t)idx1 + 1];
      const float x1 = kps1[idx1].x, y1 = kps1[idx1].y;
      const float la = x1 * T.F12[0] + y1 * T.F12[3] + T.F12[6];
      const float lb = x1 * T.F12[1] + y1 * T.F12[4] + T.F12[7];
      const float lc = x1 * T.F12[2] + y1 * T.F12[5] + T.F12[8];
      const float den = la * la + lb * lb;
      unsigned key = 0xffffffffu;
      for (int p = lane; p < n2g; p += 64) {
        const unsigned idx2 = feat2[P.b2 + p];
        if (hasMP2[idx2]) continue;
        const int dist = hamming256(a0, a1, desc2[2 * (size_t)idx2], desc2[2 * (size_t)idx2 + 1]);
        if (dist > TH_LOW) continue;
        const float x2 = kps2[idx2].x, y2 = kps2[idx2].y;
        const int oct = kps2[idx2].octave;
        const float distex = T.ex - x2, distey = T.ey - y2;
        if (distex * distex + distey * distey < 100 * T.scale2[oct]) continue;
        const float num = la * x2 + lb * y2 + lc;
        if (den == 0) continue;
        const float dsqr = num * num / den;
        if (!((double)dsqr < 3.84 * (double)T.sigma2[oct])) continue;
        key = min(key, ((unsigned)dist << 16) | (0xffffu - (unsigned)p));
      }
      key = wave_min(key);
      if (key != 0xffffffffu && lane == 0) matches12[idx1] = (int32_t)feat2[P.b2 + (int)(0xffffu - (key & 0xffffu))];
    }
  }
  done();
}

}  // namespace

struct orbfe_vocabulary {
  int device = 0, k = 0, L = 0, scoring = 0, weighting = 0;
  int nNodes = 0, maxDepth = 0;
  std::vector<double> weight;     // per node
  std::vector<uint32_t> word;     // per node (leaves flagged in the file count up from 0)
  DevBuf<int> d_childBegin, d_childCount, d_slotNode;
  DevBuf<uint4> d_slotDesc;
  DevBuf<uint4> d_desc;
  DevBuf<uint2> d_out;
  PinBuf<uint2> h_out;
  hipStream_t stream = nullptr;
};

namespace orbfe {
// BowVector / FeatureVector of n descriptors from their (leaf node, node at the FeatureVector level) pairs, in the
// reference's order (TemplatedVocabulary.h:1155-1203).  Host bookkeeping shared by orbfe_bow_transform and the
// extractor's fused path.
int bow_assemble(orbfe_vocabulary* v, const uint2* ln, int n, uint32_t* bow_ids, double* bow_values, int* n_words,
                 uint32_t* fv_nodes, uint32_t* fv_offsets, uint32_t* fv_features, int* n_fv_nodes,
                 uint32_t* word_of_feature, uint32_t* node_of_feature) {
  *n_words = 0;
  *n_fv_nodes = 0;
  fv_offsets[0] = 0;
  // ---- BowVector / FeatureVector in the reference's order (TemplatedVocabulary.h:1155-1203) ----
  const bool tf = v->weighting == 0 || v->weighting == 1;   // TF_IDF, TF: weights add up; IDF, BINARY: first one stays
  std::vector<std::pair<uint32_t, uint32_t>> key;   // local: several extractor handles may share one vocabulary
  key.reserve(n);
  for (int i = 0; i < n; i++) {
    const uint32_t leaf = ln[i].x;
    if (word_of_feature) word_of_feature[i] = v->word[leaf];
    if (node_of_feature) node_of_feature[i] = ln[i].y;
    if (v->weight[leaf] > 0) key.emplace_back(v->word[leaf], (uint32_t)i);   // "not stopped"
  }
  std::sort(key.begin(), key.end());   // by word, then feature index: the order std::map accumulation sees
  int nw = 0;
  for (size_t a = 0; a < key.size();) {
    size_t b = a;
    double w = 0;
    for (; b < key.size() && key[b].first == key[a].first; b++) {
      const double wi = v->weight[ln[key[b].second].x];
      if (b == a) w = wi;
      else if (tf) w += wi;
    }
    bow_ids[nw] = key[a].first;
    bow_values[nw] = w;
    nw++;
    a = b;
  }
  const bool must = v->scoring != 5;   // every scoring but DOT_PRODUCT normalises (ScoringObject.h:73-90)
  if (tf && nw > 0 && !must) {
    const double nd = nw;
    for (int i = 0; i < nw; i++) bow_values[i] /= nd;
  }
  if (must) {   // BowVector::normalize, BowVector.cpp:60-82
    double norm = 0.0;
    if (v->scoring != 1) { for (int i = 0; i < nw; i++) norm += fabs(bow_values[i]); }
    else { for (int i = 0; i < nw; i++) norm += bow_values[i] * bow_values[i]; norm = sqrt(norm); }
    if (norm > 0.0) for (int i = 0; i < nw; i++) bow_values[i] /= norm;
  }
  *n_words = nw;
  for (auto& e : key) e.first = ln[e.second].y;
  std::sort(key.begin(), key.end());   // by node, then feature index (push_back order)
  int nn = 0;
  for (size_t a = 0; a < key.size(); a++) {
    if (a == 0 || key[a].first != key[a - 1].first) { fv_nodes[nn] = key[a].first; fv_offsets[nn] = (uint32_t)a; nn++; }
    fv_features[a] = key[a].second;
  }
  fv_offsets[nn] = (uint32_t)key.size();
  *n_fv_nodes = nn;
  return ORBFE_OK;
}

// Enqueue the tree descent of n descriptors that already live in device memory (32-byte rows) on `st`.
int bow_launch_descend(orbfe_vocabulary* v, const uint8_t* d_desc, int n, int levelsup, uint2* d_out, hipStream_t st) {
  if (n <= 0) return ORBFE_OK;
  const int blocks = (int)(((size_t)n * 16 + 255) / 256);
  hipLaunchKernelGGL(k_bow_descend, dim3(blocks), dim3(256), 0, st, (const uint4*)d_desc, n, v->d_childBegin.p,
                     v->d_childCount.p, v->d_slotNode.p, v->d_slotDesc.p, v->L - levelsup, v->maxDepth, d_out);
  HIP_TRY(hipGetLastError());
  return ORBFE_OK;
}
int bow_device(const orbfe_vocabulary* v) { return v->device; }
}  // namespace orbfe

extern "C" {

int orbfe_vocabulary_create(int device_id, int k, int L, int scoring, int weighting, const void* records, int n_records,
                            orbfe_vocabulary** out) {
  if (!out || !records || n_records < 1 || k < 1 || k > 255 || L < 1 || scoring < 0 || scoring > 5 || weighting < 0 ||
      weighting > 3) {
    set_err("bad vocabulary arguments");
    return ORBFE_ERR_INVALID;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
    set_err("no HIP device %d (the bag-of-words path has no CPU fallback)", device_id);
    return ORBFE_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device_id));
  const uint8_t* rec = (const uint8_t*)records;
  const int nNodes = n_records + 1;
  std::vector<int> parent(nNodes, 0), childCount(nNodes, 0), childBegin(nNodes + 1, 0), depth(nNodes, 0);
  std::unique_ptr<orbfe_vocabulary> v(new orbfe_vocabulary);
  v->weight.assign(nNodes, 0.0);
  v->word.assign(nNodes, 0);
  uint32_t nwords = 0;
  int maxDepth = 0;
  for (int nid = 1; nid < nNodes; nid++) {
    const uint8_t* b = rec + (size_t)kRecord * (nid - 1);
    int pid;
    memcpy(&pid, b, 4);
    if (pid < 0 || pid >= nid) { set_err("vocabulary record %d: parent %d is not an earlier node", nid - 1, pid); return ORBFE_ERR_INVALID; }
    parent[nid] = pid;
    childCount[pid]++;
    depth[nid] = depth[pid] + 1;
    maxDepth = std::max(maxDepth, depth[nid]);
    memcpy(&v->weight[nid], b + 37, 8);
    if (b[4] > 0) v->word[nid] = nwords++;
  }
  for (int i = 0; i < nNodes; i++) {
    if (childCount[i] > 255) { set_err("vocabulary node %d has %d children (max 255)", i, childCount[i]); return ORBFE_ERR_OVERFLOW; }
    childBegin[i + 1] = childBegin[i] + childCount[i];
  }
  std::vector<int> fill(childBegin.begin(), childBegin.end() - 1), slotNode(n_records);
  std::vector<uint8_t> slotDesc((size_t)n_records * 32);
  for (int nid = 1; nid < nNodes; nid++) {   // children in the order the reference appends them (record order)
    const int s = fill[parent[nid]]++;
    slotNode[s] = nid;
    memcpy(&slotDesc[(size_t)s * 32], rec + (size_t)kRecord * (nid - 1) + 5, 32);
  }
  v->device = device_id; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
  v->nNodes = nNodes; v->maxDepth = maxDepth;
  int rc;
  if ((rc = v->d_childBegin.ensure(nNodes)) || (rc = v->d_childCount.ensure(nNodes)) || (rc = v->d_slotNode.ensure(n_records)) ||
      (rc = v->d_slotDesc.ensure((size_t)n_records * 2)))
    return rc;
  HIP_TRY(hipMemcpy(v->d_childBegin.p, childBegin.data(), sizeof(int) * nNodes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(v->d_childCount.p, childCount.data(), sizeof(int) * nNodes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(v->d_slotNode.p, slotNode.data(), sizeof(int) * n_records, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(v->d_slotDesc.p, slotDesc.data(), slotDesc.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
  *out = v.release();
  return ORBFE_OK;
}

int orbfe_vocabulary_create_from_image(int device_id, const void* image, size_t bytes, orbfe_vocabulary** out) {
  if (!image || bytes < 4 + kRecord) { set_err("vocabulary image too short"); return ORBFE_ERR_INVALID; }
  const uint8_t* b = (const uint8_t*)image;
  // header checks of the reference's loader (TemplatedVocabulary.h:1574-1586)
  if (b[0] > 20 || b[1] < 1 || b[1] > 10 || b[2] > 5 || b[3] > 3) { set_err("not a binary ORB vocabulary"); return ORBFE_ERR_INVALID; }
  return orbfe_vocabulary_create(device_id, b[0], b[1], b[2], b[3], b + 4, (int)((bytes - 4) / kRecord), out);
}

int orbfe_bow_assemble(orbfe_vocabulary* v, const uint32_t* leaf_node, const uint32_t* level_node, int n, uint32_t* bow_ids,
                       double* bow_values, int* n_words, uint32_t* fv_nodes, uint32_t* fv_offsets, uint32_t* fv_features,
                       int* n_fv_nodes, uint32_t* word_of_feature) {
  if (!v || n < 0 || (n > 0 && (!leaf_node || !level_node)) || !bow_ids || !bow_values || !n_words || !fv_nodes || !fv_offsets ||
      !fv_features || !n_fv_nodes) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  std::vector<uint2> ln(n);
  for (int i = 0; i < n; i++) {
    if (leaf_node[i] >= (uint32_t)v->nNodes) { set_err("feature %d: node id out of range", i); return ORBFE_ERR_INVALID; }
    ln[i] = make_uint2(leaf_node[i], level_node[i]);
  }
  return orbfe::bow_assemble(v, ln.data(), n, bow_ids, bow_values, n_words, fv_nodes, fv_offsets, fv_features, n_fv_nodes,
                             word_of_feature, nullptr);
}

void orbfe_vocabulary_destroy(orbfe_vocabulary* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
  delete v;
}

int orbfe_vocabulary_info(const orbfe_vocabulary* v, int* k, int* L, int* scoring, int* weighting, int* n_nodes, int* n_words) {
  if (!v) { set_err("vocabulary is NULL"); return ORBFE_ERR_INVALID; }
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (scoring) *scoring = v->scoring;
  if (weighting) *weighting = v->weighting;
  if (n_nodes) *n_nodes = v->nNodes;
  if (n_words) {
    uint32_t m = 0;
    bool any = false;
    for (int i = 1; i < v->nNodes; i++) if (v->word[i] >= m) { m = v->word[i]; any = true; }
    *n_words = any ? (int)m + 1 : 0;
  }
  return ORBFE_OK;
}

int orbfe_bow_transform(orbfe_vocabulary* v, const uint8_t* desc, int n, int in_device_memory, int levelsup,
                        uint32_t* bow_ids, double* bow_values, int* n_words, uint32_t* fv_nodes, uint32_t* fv_offsets,
                        uint32_t* fv_features, int* n_fv_nodes, uint32_t* word_of_feature, uint32_t* node_of_feature) {
  if (!v || n < 0 || (n > 0 && !desc) || !bow_ids || !bow_values || !n_words || !fv_nodes || !fv_offsets || !fv_features ||
      !n_fv_nodes) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  *n_words = 0;
  *n_fv_nodes = 0;
  fv_offsets[0] = 0;
  if (n == 0) return ORBFE_OK;
  HIP_TRY(hipSetDevice(v->device));
  int rc;
  if ((rc = v->d_out.ensure(n)) || (rc = v->h_out.ensure(n))) return rc;
  const uint4* d = (const uint4*)desc;
  if (!in_device_memory) {
    if ((rc = v->d_desc.ensure((size_t)n * 2))) return rc;
    HIP_TRY(hipMemcpyAsync(v->d_desc.p, desc, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
    d = v->d_desc.p;
  }
  const int nidLevel = v->L - levelsup;   // <= 0: the node is the root (0), TemplatedVocabulary.h:1315-1316
  const int blocks = (int)(((size_t)n * 16 + 255) / 256);
  hipLaunchKernelGGL(k_bow_descend, dim3(blocks), dim3(256), 0, v->stream, d, n, v->d_childBegin.p, v->d_childCount.p,
                     v->d_slotNode.p, v->d_slotDesc.p, nidLevel, v->maxDepth, v->d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(v->h_out.p, v->d_out.p, sizeof(uint2) * n, hipMemcpyDeviceToHost, v->stream));
  HIP_TRY(hipStreamSynchronize(v->stream));

  return orbfe::bow_assemble(v, v->h_out.p, n, bow_ids, bow_values, n_words, fv_nodes, fv_offsets, fv_features, n_fv_nodes,
                             word_of_feature, node_of_feature);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// SearchByBoW
// ---------------------------------------------------------------------------------------------
namespace {

// ComputeThreeMaxima on bin counts (ORBmatcher.cc:1554-1595)
void three_maxima(const int* count, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = count[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

struct BowScratch {
  DevBuf<uint4> d_desc1, d_desc2;
  DevBuf<uint8_t> d_valid1, d_valid2;
  DevBuf<uint32_t> d_feat1, d_feat2;
  DevBuf<BowPair> d_pairs;
  DevBuf<int32_t> d_m12;
  PinBuf<int32_t> h_m12;
  DevBuf<OrbfeKeyPoint> d_kps1, d_kps2;
  DevBuf<uint8_t> d_arena;    // batched search: everything that goes up in one copy
  PinBuf<uint8_t> h_arena;
  DevBuf<uint32_t> d_topk;    // large nodes: kBowTopK least keys per frame-1 feature (k_bow_topk)
  DevBuf<TopkItem> d_items;
  DevBuf<unsigned> d_done;    // route without copy commands: block counter, the call's number (page-locked)
  PinBuf<int> h_done;
  int seq = 0;
};

// common vocabulary nodes of two FeatureVectors (the lower_bound zig-zag of ORBmatcher.cc:175-258 visits exactly the
// intersection, in ascending order) as feature ranges
int common_nodes(const uint32_t* fv1_nodes, const uint32_t* fv1_offsets, int n_fv1, const uint32_t* fv2_nodes,
                 const uint32_t* fv2_offsets, int n_fv2, std::vector<BowPair>& pairs) {
  pairs.clear();
  for (int a = 0, b = 0; a < n_fv1 && b < n_fv2;) {
    if (fv1_nodes[a] == fv2_nodes[b]) {
      BowPair p = {(int)fv1_offsets[a], (int)fv1_offsets[a + 1], (int)fv2_offsets[b], (int)fv2_offsets[b + 1], 0, -1};
      if (p.e2 - p.b2 > kMaxGroup) { set_err("a vocabulary node holds %d features (max %d)", p.e2 - p.b2, kMaxGroup); return ORBFE_ERR_OVERFLOW; }
      if (p.e1 > p.b1 && p.e2 > p.b2) pairs.push_back(p);
      a++; b++;
    } else if (fv1_nodes[a] < fv2_nodes[b]) a++;
    else b++;
  }
  return ORBFE_OK;
}

// rotation-histogram pruning shared by the BoW searches (ORBmatcher.cc:226-236,261-280)
int prune_by_orientation(const float* angle1, size_t stride1, const float* angle2, size_t stride2, int n1, int32_t* matches12) {
  int count[HISTO_LENGTH] = {};
  std::vector<int8_t> bin(n1, -1);
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < n1; i++) {
    const int j = matches12[i];
    if (j < 0) continue;
    float rot = *(const float*)((const char*)angle1 + stride1 * i) - *(const float*)((const char*)angle2 + stride2 * j);
    if (rot < 0.0) rot += 360.0f;
    int b = (int)round(rot * factor);
    if (b == HISTO_LENGTH) b = 0;
    if (b < 0 || b >= HISTO_LENGTH) continue;   // the reference asserts; cannot happen for angles in [0,360)
    bin[i] = (int8_t)b;
    count[b]++;
  }
  int ind1 = -1, ind2 = -1, ind3 = -1, removed = 0;
  three_maxima(count, HISTO_LENGTH, ind1, ind2, ind3);
  for (int i = 0; i < n1; i++) {
    const int b = bin[i];
    if (b < 0 || b == ind1 || b == ind2 || b == ind3) continue;
    matches12[i] = -1;
    removed++;
  }
  return removed;
}


}  // namespace

// int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, ...) for SEVERAL keyframes against one frame -- the loop of
// Tracking::Relocalization (src/Tracking.cc:1005-1030: one SearchByBoW per candidate keyframe, all against mCurrentFrame) --
// as ONE upload, ONE launch of k_bow_match over every (keyframe, common vocabulary node) pair and ONE download.  The
// searches are independent (each has its own vpMapPointMatches / matched set), so results equal the per-keyframe calls.
// dev1 / dev2 != nullptr (single keyframe only): that side's descriptor rows already lie in device memory (a resident
// frame's, orbfe_frame_descriptors_device) -- they are neither copied into the arena nor read from it
static int bow_batch_core(orbfe_matcher* m, int n_kf, const uint8_t* const* desc1, const float* const* angle1,
                          const uint8_t* const* valid1, const int* n1, const uint32_t* const* fv1_nodes,
                          const uint32_t* const* fv1_offsets, const uint32_t* const* fv1_features, const int* n_fv1,
                          const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                          const uint32_t* fv2_nodes, const uint32_t* fv2_offsets, const uint32_t* fv2_features,
                          int n_fv2, float nnratio, int check_orientation, int strict_threshold,
                          int32_t* const* matches12, int* nmatches, const uint4* dev1, const uint4* dev2) {
  if (!m || n_kf < 0 || n2 < 0 || n_fv2 < 0 || (n_kf && (!desc1 || !valid1 || !n1 || !fv1_nodes || !fv1_offsets || !fv1_features || !n_fv1 ||
      !matches12 || !nmatches)) || (n2 > 0 && !desc2) || (n_fv2 > 0 && (!fv2_nodes || !fv2_offsets || !fv2_features)) ||
      (check_orientation && (!angle2 || !angle1))) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  const int nf2 = n_fv2 ? (int)fv2_offsets[n_fv2] : 0;
  for (int i = 0; i < nf2; i++) if (fv2_features[i] >= (uint32_t)n2) { set_err("fv2 feature index out of range"); return ORBFE_ERR_INVALID; }
  std::vector<BowPair> pairs, all;
  std::vector<size_t> rowBase(n_kf + 1, 0), featBase(n_kf + 1, 0);
  int rc;
  for (int k = 0; k < n_kf; k++) {
    nmatches[k] = 0;
    if (n1[k] < 0 || n_fv1[k] < 0 || (n1[k] > 0 && (!desc1[k] || !valid1[k] || !matches12[k])) ||
        (n_fv1[k] > 0 && (!fv1_nodes[k] || !fv1_offsets[k] || !fv1_features[k])) || (check_orientation && n1[k] > 0 && !angle1[k])) {
      set_err("bad argument for keyframe %d", k);
      return ORBFE_ERR_INVALID;
    }
    for (int i = 0; i < n1[k]; i++) matches12[k][i] = -1;
    const int nf1 = n_fv1[k] ? (int)fv1_offsets[k][n_fv1[k]] : 0;
    for (int i = 0; i < nf1; i++) if (fv1_features[k][i] >= (uint32_t)n1[k]) { set_err("fv1 feature index out of range"); return ORBFE_ERR_INVALID; }
    if ((rc = common_nodes(fv1_nodes[k], fv1_offsets[k], n_fv1[k], fv2_nodes, fv2_offsets, n_fv2, pairs))) return rc;
    for (BowPair p : pairs) {
      p.b1 += (int)featBase[k]; p.e1 += (int)featBase[k]; p.base1 = (int)rowBase[k];
      all.push_back(p);
    }
    rowBase[k + 1] = rowBase[k] + (size_t)n1[k];
    featBase[k + 1] = featBase[k] + (size_t)nf1;
  }
  if (all.empty()) return ORBFE_OK;
  // large nodes (more than kBowSide features of either frame): their frame-1 features get a top-K list first (k_bow_topk)
  std::vector<TopkItem> items;
  size_t topkRows = 0;
  int maxStage = 0;
  for (size_t i = 0; i < all.size(); i++) {
    BowPair& p = all[i];
    const int n1g = p.e1 - p.b1, n2g = p.e2 - p.b2;
    if (n1g <= kBowSide && n2g <= kBowSide) continue;
    p.tk = (int)topkRows;
    topkRows += (size_t)n1g;
    maxStage = std::max(maxStage, std::min(n2g, kTopkLdsFeatures));
    for (int r = 0; r < n1g; r += kTopkRows) items.push_back(TopkItem{(int)i, r});
  }
  const size_t rows1 = rowBase[n_kf], feats1 = featBase[n_kf];
  HIP_TRY(hipSetDevice(orbfe::matcher_device(m)));
  std::shared_ptr<void>& slot = orbfe::matcher_bow_slot(m);
  if (!slot) slot = std::make_shared<BowScratch>();
  BowScratch* S = static_cast<BowScratch*>(slot.get());
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t oD1 = 0, oD2 = oD1 + al(32 * rows1), oV1 = oD2 + al(32 * (size_t)n2), oV2 = oV1 + al(rows1), oF1 = oV2 + al((size_t)std::max(n2, 1)),
               oF2 = oF1 + al(4 * feats1), oP = oF2 + al(4 * (size_t)nf2), total = oP + al(sizeof(BowPair) * all.size());
  if ((rc = S->h_arena.ensure(total)) || (rc = S->d_arena.ensure(total)) || (rc = S->d_m12.ensure(rows1)) || (rc = S->h_m12.ensure(rows1))) return rc;
  uint8_t* H = S->h_arena.p;
  for (int k = 0; k < n_kf; k++) {
    if (n1[k]) {
      if (!dev1) memcpy(H + oD1 + 32 * rowBase[k], desc1[k], 32 * (size_t)n1[k]);
      memcpy(H + oV1 + rowBase[k], valid1[k], (size_t)n1[k]);
    }
    const size_t nf1 = featBase[k + 1] - featBase[k];
    if (nf1) memcpy(H + oF1 + 4 * featBase[k], fv1_features[k], 4 * nf1);
  }
  if (n2 && !dev2) memcpy(H + oD2, desc2, 32 * (size_t)n2);
  if (valid2 && n2) memcpy(H + oV2, valid2, (size_t)n2);
  if (nf2) memcpy(H + oF2, fv2_features, 4 * (size_t)nf2);
  memcpy(H + oP, all.data(), sizeof(BowPair) * all.size());
  hipStream_t st = orbfe::matcher_stream(m);
  // Nodes whose two feature lists fit the kernel's LDS (up to kBowSide features per side: every node of a real vocabulary at
  // levelsup 4) are read ONCE per feature, so the kernel reads the page-locked arena itself and writes the matches into
  // page-locked memory: no copy command, no fill, and the host polls the kernel's completion word instead of waiting on the
  // stream (ORBFE_BOW_ZEROCOPY=0, or a larger node: upload, fill, launch, download as before).
  bool small = true;
  for (const BowPair& p : all) small = small && p.e1 - p.b1 <= kBowSide && p.e2 - p.b2 <= kBowSide;
  const char* zce = getenv("ORBFE_BOW_ZEROCOPY");   // (read per call: the parity tests run both routes in one process)
  const bool zc = !(zce && atoi(zce) == 0);
  if (small && zc) {
    if (!S->d_done.p) {
      if ((rc = S->d_done.ensure(16)) || (rc = S->h_done.ensure(16))) return rc;
      HIP_TRY(hipMemsetAsync(S->d_done.p, 0, 16 * sizeof(unsigned), st));
      S->h_done.p[0] = 0;
    }
    memset(S->h_m12.p, 0xff, sizeof(int32_t) * rows1);
    S->seq = S->seq == INT_MAX ? 1 : S->seq + 1;
    hipLaunchKernelGGL(k_bow_match, dim3((unsigned)all.size()), dim3(kBowThreads), 0, st, dev1 ? dev1 : (const uint4*)(H + oD1), (const uint8_t*)(H + oV1),
                       (const uint32_t*)(H + oF1), dev2 ? dev2 : (const uint4*)(H + oD2), valid2 ? (const uint8_t*)(H + oV2) : (const uint8_t*)nullptr,
                       (const uint32_t*)(H + oF2), (const BowPair*)(H + oP), strict_threshold ? TH_LOW - 1 : TH_LOW, nnratio, S->h_m12.p,
                       S->d_done.p, S->h_done.p, S->seq, (const uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    const volatile int* flag = S->h_done.p;
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1;; spin++) {
      if (*flag == S->seq) { seen = true; break; }
      if ((spin & 255u) == 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIP_TRY(hipStreamSynchronize(st));
  } else {
    uint8_t* D = S->d_arena.p;
    HIP_TRY(hipMemcpyAsync(D, H, total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(S->d_m12.p, 0xff, sizeof(int32_t) * rows1, st));
    if (!items.empty()) {
      if ((rc = S->d_topk.ensure(topkRows * kBowTopK)) || (rc = S->d_items.ensure(items.size()))) return rc;
      HIP_TRY(hipMemcpyAsync(S->d_items.p, items.data(), sizeof(TopkItem) * items.size(), hipMemcpyHostToDevice, st));   // (waited for below)
      const size_t ldsBytes = (size_t)maxStage * 36 + 16;
      static std::mutex mu;
      static size_t attr[64] = {};
      {
        std::lock_guard<std::mutex> lk(mu);
        const int dv = orbfe::matcher_device(m);
        if (dv >= 0 && dv < 64 && ldsBytes > attr[dv]) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_topk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
          attr[dv] = ldsBytes;
        }
      }
      hipLaunchKernelGGL(k_bow_topk, dim3((unsigned)items.size()), dim3(256), ldsBytes, st, dev1 ? dev1 : (const uint4*)(D + oD1), (const uint8_t*)(D + oV1),
                         (const uint32_t*)(D + oF1), dev2 ? dev2 : (const uint4*)(D + oD2), valid2 ? (const uint8_t*)(D + oV2) : (const uint8_t*)nullptr,
                         (const uint32_t*)(D + oF2), (const BowPair*)(D + oP), (const TopkItem*)S->d_items.p, S->d_topk.p);
      HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_bow_match, dim3((unsigned)all.size()), dim3(kBowThreads), 0, st, dev1 ? dev1 : (const uint4*)(D + oD1), (const uint8_t*)(D + oV1),
                       (const uint32_t*)(D + oF1), dev2 ? dev2 : (const uint4*)(D + oD2), valid2 ? (const uint8_t*)(D + oV2) : (const uint8_t*)nullptr,
                       (const uint32_t*)(D + oF2), (const BowPair*)(D + oP), strict_threshold ? TH_LOW - 1 : TH_LOW, nnratio, S->d_m12.p,
                       (unsigned*)nullptr, (int*)nullptr, 0, (const uint32_t*)S->d_topk.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(S->h_m12.p, S->d_m12.p, sizeof(int32_t) * rows1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  for (int k = 0; k < n_kf; k++) {
    int nm = 0;
    const int32_t* src = S->h_m12.p + rowBase[k];
    for (int i = 0; i < n1[k]; i++) {
      matches12[k][i] = src[i];
      if (src[i] >= 0) nm++;
    }
    if (check_orientation && n1[k]) nm -= prune_by_orientation(angle1[k], sizeof(float), angle2, sizeof(float), n1[k], matches12[k]);
    nmatches[k] = nm;
  }
  return ORBFE_OK;
}

extern "C" int orbfe_search_by_bow_batch(orbfe_matcher* m, int n_kf, const uint8_t* const* desc1, const float* const* angle1,
                                         const uint8_t* const* valid1, const int* n1, const uint32_t* const* fv1_nodes,
                                         const uint32_t* const* fv1_offsets, const uint32_t* const* fv1_features, const int* n_fv1,
                                         const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                                         const uint32_t* fv2_nodes, const uint32_t* fv2_offsets, const uint32_t* fv2_features,
                                         int n_fv2, float nnratio, int check_orientation, int strict_threshold,
                                         int32_t* const* matches12, int* nmatches) {
  return bow_batch_core(m, n_kf, desc1, angle1, valid1, n1, fv1_nodes, fv1_offsets, fv1_features, n_fv1, desc2, angle2, valid2, n2,
                        fv2_nodes, fv2_offsets, fv2_features, n_fv2, nnratio, check_orientation, strict_threshold, matches12, nmatches,
                        nullptr, nullptr);
}

// descriptor rows: 1 = in the memory of device `device`, 16-byte aligned (read in place); 0 = host memory (copied into the
// arena); -1 = device memory the kernels cannot take as it is (another device's, or misaligned) -- refused, never memcpy'd
static int bow_rows_where(const uint8_t* p, int device) {
  hipPointerAttribute_t attr;
  if (p && hipPointerGetAttributes(&attr, p) == hipSuccess) {
    if (attr.type != hipMemoryTypeDevice) return 0;
    return attr.device == device && ((uintptr_t)p & 15u) == 0 ? 1 : -1;
  }
  (void)hipGetLastError();
  return 0;
}

extern "C" int orbfe_search_by_bow(orbfe_matcher* m, const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                                   const uint32_t* fv1_nodes, const uint32_t* fv1_offsets, const uint32_t* fv1_features,
                                   int n_fv1, const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                                   const uint32_t* fv2_nodes, const uint32_t* fv2_offsets, const uint32_t* fv2_features,
                                   int n_fv2, float nnratio, int check_orientation, int strict_threshold,
                                   int32_t* matches12, int* nmatches) {
  if (!m || n1 < 0 || n2 < 0 || n_fv1 < 0 || n_fv2 < 0 || !matches12 || !nmatches || (n1 > 0 && (!desc1 || !valid1)) ||
      (n2 > 0 && !desc2) || (n_fv1 > 0 && (!fv1_nodes || !fv1_offsets || !fv1_features)) ||
      (n_fv2 > 0 && (!fv2_nodes || !fv2_offsets || !fv2_features)) || (check_orientation && (!angle1 || !angle2))) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  *nmatches = 0;
  // a side whose rows are a resident frame's (orbfe_frame_descriptors_device) stays where it is
  const int dev = orbfe::matcher_device(m);
  const int w1 = bow_rows_where(desc1, dev), w2 = bow_rows_where(desc2, dev);
  if (w1 < 0 || w2 < 0) {
    set_err("descriptor rows in device memory must belong to the matcher's device and be 16-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  const uint4* dev1 = w1 ? (const uint4*)desc1 : nullptr;
  const uint4* dev2 = w2 ? (const uint4*)desc2 : nullptr;
  return bow_batch_core(m, 1, &desc1, &angle1, &valid1, &n1, &fv1_nodes, &fv1_offsets, &fv1_features, &n_fv1, desc2, angle2,
                        valid2, n2, fv2_nodes, fv2_offsets, fv2_features, n_fv2, nnratio, check_orientation, strict_threshold,
                        &matches12, nmatches, dev1, dev2);
}

extern "C" int orbfe_search_for_triangulation(orbfe_matcher* m, const OrbfeKeyPoint* kps1_un, const uint8_t* desc1,
                                              const uint8_t* has_mp1, int n1, const uint32_t* fv1_nodes,
                                              const uint32_t* fv1_offsets, const uint32_t* fv1_features, int n_fv1,
                                              const OrbfeKeyPoint* kps2_un, const uint8_t* desc2, const uint8_t* has_mp2,
                                              int n2, const uint32_t* fv2_nodes, const uint32_t* fv2_offsets,
                                              const uint32_t* fv2_features, int n_fv2, const float F12[9], float ex, float ey,
                                              const float* scale_factors2, const float* level_sigma2_2, int nlevels2,
                                              int check_orientation, int32_t* pairs_out, int* nmatches) {
  if (!m || n1 < 0 || n2 < 0 || n_fv1 < 0 || n_fv2 < 0 || !pairs_out || !nmatches || !F12 || !scale_factors2 || !level_sigma2_2 ||
      nlevels2 < 1 || nlevels2 > 16 || (n1 > 0 && (!kps1_un || !desc1 || !has_mp1)) || (n2 > 0 && (!kps2_un || !desc2 || !has_mp2)) ||
      (n_fv1 > 0 && (!fv1_nodes || !fv1_offsets || !fv1_features)) || (n_fv2 > 0 && (!fv2_nodes || !fv2_offsets || !fv2_features))) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  *nmatches = 0;
  std::vector<BowPair> pairs;
  int rc = common_nodes(fv1_nodes, fv1_offsets, n_fv1, fv2_nodes, fv2_offsets, n_fv2, pairs);
  if (rc) return rc;
  const int nf1 = n_fv1 ? (int)fv1_offsets[n_fv1] : 0, nf2 = n_fv2 ? (int)fv2_offsets[n_fv2] : 0;
  for (int i = 0; i < nf1; i++) if (fv1_features[i] >= (uint32_t)n1) { set_err("fv1 feature index out of range"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < nf2; i++) if (fv2_features[i] >= (uint32_t)n2) { set_err("fv2 feature index out of range"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < n2; i++) if (kps2_un[i].octave < 0 || kps2_un[i].octave >= nlevels2) { set_err("keypoint octave out of range"); return ORBFE_ERR_INVALID; }
  if (pairs.empty()) return ORBFE_OK;
  HIP_TRY(hipSetDevice(orbfe::matcher_device(m)));
  std::shared_ptr<void>& slot = orbfe::matcher_bow_slot(m);
  if (!slot) slot = std::make_shared<BowScratch>();
  BowScratch* S = static_cast<BowScratch*>(slot.get());
  // everything the kernel reads in ONE page-locked arena (nine copy commands from ordinary memory, about 12 us each, were
  // half of the call): read in place by the kernel when every node fits its LDS -- each item is then fetched once --, else
  // uploaded with one copy
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t oK1 = 0, oK2 = oK1 + al(sizeof(OrbfeKeyPoint) * (size_t)n1), oD1 = oK2 + al(sizeof(OrbfeKeyPoint) * (size_t)n2),
               oD2 = oD1 + al(32 * (size_t)n1), oV1 = oD2 + al(32 * (size_t)n2), oV2 = oV1 + al((size_t)std::max(n1, 1)),
               oF1 = oV2 + al((size_t)std::max(n2, 1)), oF2 = oF1 + al(4 * (size_t)nf1), oP = oF2 + al(4 * (size_t)nf2),
               total = oP + al(sizeof(BowPair) * pairs.size());
  if ((rc = S->h_arena.ensure(total)) || (rc = S->d_arena.ensure(total)) || (rc = S->d_m12.ensure(n1)) || (rc = S->h_m12.ensure(n1))) return rc;
  uint8_t* H = S->h_arena.p;
  memcpy(H + oK1, kps1_un, sizeof(OrbfeKeyPoint) * (size_t)n1);
  memcpy(H + oK2, kps2_un, sizeof(OrbfeKeyPoint) * (size_t)n2);
  memcpy(H + oD1, desc1, 32 * (size_t)n1);
  memcpy(H + oD2, desc2, 32 * (size_t)n2);
  memcpy(H + oV1, has_mp1, (size_t)n1);
  memcpy(H + oV2, has_mp2, (size_t)n2);
  memcpy(H + oF1, fv1_features, 4 * (size_t)nf1);
  memcpy(H + oF2, fv2_features, 4 * (size_t)nf2);
  memcpy(H + oP, pairs.data(), sizeof(BowPair) * pairs.size());
  hipStream_t st = orbfe::matcher_stream(m);
  TriParams T;
  memcpy(T.F12, F12, sizeof T.F12);
  T.ex = ex; T.ey = ey;
  for (int i = 0; i < 16; i++) { T.scale2[i] = scale_factors2[std::min(i, nlevels2 - 1)]; T.sigma2[i] = level_sigma2_2[std::min(i, nlevels2 - 1)]; }
  bool small = true;
  for (const BowPair& p : pairs) small = small && p.e1 - p.b1 <= kBowSide && p.e2 - p.b2 <= kBowSide;
  const char* zce = getenv("ORBFE_BOW_ZEROCOPY");
  const bool zc = !(zce && atoi(zce) == 0);
  if (small && zc) {
    if (!S->d_done.p) {
      if ((rc = S->d_done.ensure(16)) || (rc = S->h_done.ensure(16))) return rc;
      HIP_TRY(hipMemsetAsync(S->d_done.p, 0, 16 * sizeof(unsigned), st));
      S->h_done.p[0] = 0;
    }
    memset(S->h_m12.p, 0xff, sizeof(int32_t) * (size_t)n1);
    S->seq = S->seq == INT_MAX ? 1 : S->seq + 1;
    hipLaunchKernelGGL(k_bow_triangulate, dim3((unsigned)pairs.size()), dim3(kBowThreads), 0, st, (const OrbfeKeyPoint*)(H + oK1),
                       (const uint4*)(H + oD1), (const uint8_t*)(H + oV1), (const uint32_t*)(H + oF1), (const OrbfeKeyPoint*)(H + oK2),
                       (const uint4*)(H + oD2), (const uint8_t*)(H + oV2), (const uint32_t*)(H + oF2), (const BowPair*)(H + oP), T,
                       S->h_m12.p, S->d_done.p, S->h_done.p, S->seq);
    HIP_TRY(hipGetLastError());
    const volatile int* flag = S->h_done.p;
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1;; spin++) {
      if (*flag == S->seq) { seen = true; break; }
      if ((spin & 255u) == 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIP_TRY(hipStreamSynchronize(st));
  } else {
    uint8_t* D = S->d_arena.p;
    HIP_TRY(hipMemcpyAsync(D, H, total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(S->d_m12.p, 0xff, sizeof(int32_t) * (size_t)n1, st));
    hipLaunchKernelGGL(k_bow_triangulate, dim3((unsigned)pairs.size()), dim3(kBowThreads), 0, st, (const OrbfeKeyPoint*)(D + oK1),
                       (const uint4*)(D + oD1), (const uint8_t*)(D + oV1), (const uint32_t*)(D + oF1), (const OrbfeKeyPoint*)(D + oK2),
                       (const uint4*)(D + oD2), (const uint8_t*)(D + oV2), (const uint32_t*)(D + oF2), (const BowPair*)(D + oP), T,
                       S->d_m12.p, (unsigned*)nullptr, (int*)nullptr, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(S->h_m12.p, S->d_m12.p, sizeof(int32_t) * (size_t)n1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  int nm = 0;
  for (int i = 0; i < n1; i++) if (S->h_m12.p[i] >= 0) nm++;
  if (check_orientation)
    nm -= prune_by_orientation(&kps1_un[0].angle, sizeof(OrbfeKeyPoint), &kps2_un[0].angle, sizeof(OrbfeKeyPoint), n1, S->h_m12.p);
  int np = 0;
  for (int i = 0; i < n1; i++) {   // vMatchedPairs in ascending idx1 (ORBmatcher.cc:792-800)
    if (S->h_m12.p[i] < 0) continue;
    pairs_out[2 * np] = i;
    pairs_out[2 * np + 1] = S->h_m12.p[i];
    np++;
  }
  *nmatches = nm;
  return ORBFE_OK;
}
