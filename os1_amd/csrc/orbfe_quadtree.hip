// orbfe_quadtree.hip -- GPU-resident keypoint thinning (DistributeOctTree) for gfx950.
//
// Behaviour contract: ORBextractor::DistributeOctTree + ExtractorNode::DivideNode of the reference
// (src/ORBextractor.cc:513-569, 571-795), same tie-break convention as quadtree.h / the oracle
// (equal-sized nodes: later-created first).  One workgroup per
// (frame, level) problem (k_quadtree2, 512 threads); problems of a batch run concurrently on different CUs.
//
// The reference algorithm is a sequential walk over a std::list, but its result is a pure
// function of three orderings, all of which can be produced with scans and one sort:
//   * inside a node, candidates keep their input order         -> stable 4-way partition by
//     segmented prefix sums over the candidate array (positions of a node are contiguous);
//   * list order: children are pushed to the FRONT in n1..n4 order and parents erased, so after
//     processing nodes D[0..k) (in processing order) the new list is
//       [children(D[k-1]) reversed, ..., children(D[0]) reversed] ++ [old list without D];
//     node ids are re-assigned to list positions after every pass, so "the list" is an array;
//   * final phase: nodes created in the previous round with >1 candidate are processed by
//     (size desc, creation desc) and the walk stops right after the division that makes
//     size >= N: children counts of ALL of them are computed first, a prefix sum over the sorted
//     order finds the cut, and only the prefix is applied.
// Creation order (seq) = processing order, n1..n4 inside a parent.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "orbfe_internal.h"

namespace orbfe {

// Implementation notes:
//   * 512 threads / block: a block leaves most of its CU to the other in-flight batch's kernels;
//   * candidates carry their packed (x, y, score) word, so there is no index indirection;
//   * element passes are tile loops (4 elements per thread and step, coalesced, independent loads in
//     flight together); quadrant ranks come from wave ballots + one block barrier per step;
//   * per-node lookups (split point, child positions, range deltas, scan bases, the node tables
//     themselves when CAP = 1024) live in LDS;
//   * node passes: processing order (scan / bitonic sort), cut (prefix sum), list positions (scans).
constexpr int kQt2Threads = 512;

template <int CAP>
struct Qt2Shared {
  QtNode nodes[CAP <= 1024 ? 2 : 1][CAP <= 1024 ? CAP : 1];   // node tables in LDS when they fit (CAP = 1024)
  unsigned short proc[CAP];   // node ids in processing order
  uint32_t ninfo[CAP];        // midX | midY << 12 | divide << 24
  uint16_t cpos[CAP][4];      // new list position of child q; kept nodes: [0] = new position
  alignas(16) uint32_t baseS[CAP][4];     // exclusive quadrant scan at the node's first element -> later: child.begin - baseS
  union alignas(16) {         // lifetimes do not overlap: endS lives from element pass 1 to the node pass of an iteration,
    uint32_t endS[CAP][4];    //   inclusive quadrant scan at the node's last element
    unsigned long long sortKeys[CAP];   // sortKeys from the record step of one iteration to the ordering step of the next
  };
  short tproc[CAP];           // index in processing order, -1 = not divided
  alignas(16) uint32_t wcnt[2][4][kQt2Threads / 64][4];   // [parity][sub-tile][wave][class]
  int wsumI[kQt2Threads / 64];
  int s_int[4];
};

template <int CAP>
__device__ int blockScanInt2(Qt2Shared<CAP>& sh, int v, int& total) {  // exclusive, 512 threads
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int iv = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(iv, o, 64);
    if (lane >= o) iv += t;
  }
  __syncthreads();
  if (lane == 63) sh.wsumI[wv] = iv;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kQt2Threads / 64; w++) {
    if (w < wv) base += sh.wsumI[w];
    tot += sh.wsumI[w];
  }
  total = tot;
  return base + iv - v;
}

template <int CAP>
__device__ void blockSortDesc2(Qt2Shared<CAP>& sh, int n) {
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = n + threadIdx.x; i < m; i += kQt2Threads) sh.sortKeys[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += kQt2Threads) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = sh.sortKeys[i], b = sh.sortKeys[p];
          const bool desc = ((i & k) == 0);
          if (desc ? (a < b) : (a > b)) { sh.sortKeys[i] = b; sh.sortKeys[p] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// One tile pass: class c(p) in 0..3 (4 = none) for every element; computes the exclusive count of equal-class
// elements before p (stored with the class in rankq) and per-segment (= per node) base / end scan values.
// Each step covers kEpt * 512 consecutive elements: a thread owns kEpt elements 512 apart (coalesced,
// independent loads in flight together), ranks come from wave ballots, one block barrier per step.
constexpr int kEpt = 4;

template <int CAP, class ClassFn>
__device__ void tileScan(Qt2Shared<CAP>& sh, int n, const uint16_t* own, const uint32_t* val, uint32_t* rankq,
                         bool recordSegments, ClassFn cls, uint32_t total[4]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = kQt2Threads / 64;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t carry[4] = {0, 0, 0, 0};
  int parity = 0;
  // the next tile's loads are issued before the current tile is processed (the steps are serialised by the running
  // class counts, so an unhidden L2 round trip per step is most of a step's time)
  int oN[kEpt];
  uint32_t vN[kEpt];
  auto loadTile = [&](int b) {
#pragma unroll
    for (int j = 0; j < kEpt; j++) {
      const int p = b + j * kQt2Threads + tid;
      oN[j] = -1;
      vN[j] = 0;
      if (p < n) {
        oN[j] = own ? (int)own[p] : 0;
        vN[j] = val[p];
      }
    }
  };
  loadTile(0);
  for (int b = 0; b < n; b += kEpt * kQt2Threads, parity ^= 1) {
    int o[kEpt], q[kEpt];
    uint32_t v[kEpt];
#pragma unroll
    for (int j = 0; j < kEpt; j++) { o[j] = oN[j]; v[j] = vN[j]; }
    if (b + kEpt * kQt2Threads < n) loadTile(b + kEpt * kQt2Threads);
    uint32_t lanePre[kEpt][4];
#pragma unroll
    for (int j = 0; j < kEpt; j++) {
      const int p = b + j * kQt2Threads + tid;
      q[j] = p < n ? cls(p, o[j], v[j]) : 4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned long long m = __ballot(q[j] == k);
        lanePre[j][k] = __popcll(m & below);
        if (lane == 0) sh.wcnt[parity][j][wv][k] = __popcll(m);
      }
    }
    __syncthreads();
    // exclusive prefix over the (sub-tile j, wave w) pairs in element order, derived by every wave for itself: lane
    // p = j * NW + w reads the pair's four class counts (one 16-byte LDS load), a DPP wave scan sums them, and the
    // prefix of this thread's pair (j, wv) comes back through v_readlane -- 4 LDS loads and 24 DPP adds per wave and
    // step instead of every thread adding up all 128 counts.
    static_assert(kEpt * NW <= 64, "one lane per (sub-tile, wave) pair");
    uint32_t inc[4];
    {
      const uint4 c4 = lane < kEpt * NW ? *reinterpret_cast<const uint4*>(&sh.wcnt[parity][0][0][0] + 4 * lane) : make_uint4(0, 0, 0, 0);
      inc[0] = c4.x; inc[1] = c4.y; inc[2] = c4.z; inc[3] = c4.w;
    }
    uint32_t exc[4], tot[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int v = (int)inc[k];
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
      v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
      v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
      v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
      v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
      exc[k] = (uint32_t)v - inc[k];
      tot[k] = (uint32_t)__builtin_amdgcn_readlane(v, 63);
    }
    const int wvU = __builtin_amdgcn_readfirstlane(wv);
#pragma unroll
    for (int j = 0; j < kEpt; j++) {
      uint32_t pre[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
        pre[k] = carry[k] + (uint32_t)__builtin_amdgcn_readlane((int)exc[k], j * NW + wvU) + lanePre[j][k];   // exclusive scan value of class k at p
      const int p = b + j * kQt2Threads + tid;
      // segment boundaries = neighbours with a different owner (DPP wave shifts; lanes 0 / 63 are patched below)
      int oPrev = __builtin_amdgcn_update_dpp(-1, o[j], 0x138, 0xf, 0xf, false);   // wave_shr:1: lane i <- lane i-1
      int oNext = __builtin_amdgcn_update_dpp(-1, o[j], 0x130, 0xf, 0xf, false);   // wave_shl:1: lane i <- lane i+1
      if (p < n && q[j] < 4) {
        rankq[p] = pre[q[j]] | ((uint32_t)q[j] << 30);
        if (recordSegments) {
          if (lane == 0) oPrev = p > 0 ? (int)own[p - 1] : -1;
          if (lane == 63 || p == n - 1) oNext = p + 1 < n ? (int)own[p + 1] : -1;
          if (oPrev != o[j]) *reinterpret_cast<uint4*>(sh.baseS[o[j]]) = make_uint4(pre[0], pre[1], pre[2], pre[3]);
          if (oNext != o[j])
            *reinterpret_cast<uint4*>(sh.endS[o[j]]) = make_uint4(pre[0] + (q[j] == 0), pre[1] + (q[j] == 1), pre[2] + (q[j] == 2), pre[3] + (q[j] == 3));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) carry[k] += tot[k];
  }
#pragma unroll
  for (int k = 0; k < 4; k++) total[k] = carry[k];
  __syncthreads();
}

template <int CAP>
__global__ __launch_bounds__(kQt2Threads) void k_quadtree2(QtParams Q) {
  __shared__ Qt2Shared<CAP> sh;
  constexpr int IPT = CAP / kQt2Threads;   // node items per thread
  const int level = blockIdx.x, f = Q.frameBase + blockIdx.y, tid = threadIdx.x;
  const uint32_t* ls = Q.levelStart + (long long)f * (kMaxLevels + 1);
  const uint32_t first = ls[level];
  const int n = (int)(ls[level + 1] - first);
  const int N = Q.nfeat[level];
  uint32_t* selCount = Q.selCount + (long long)f * kMaxLevels + level;
  SelKp* selOut = Q.sel + (long long)f * Q.selPerFrame + Q.selOff[level];
  if (n <= 0) {
    if (tid == 0) *selCount = 0;
    return;
  }
  const long long eo = (long long)f * Q.candCap + first;
  const uint32_t* cand = Q.cand + eo;
  uint32_t* valCur = Q.idxA + eo;
  uint32_t* valNxt = Q.idxB + eo;
  uint16_t* ownCur = Q.ownA + eo;
  uint16_t* ownNxt = Q.ownB + eo;
  uint32_t* rankq = Q.rank + eo;
  const long long no = ((long long)f * Q.nlevels + level) * kQtNodeCap;
  constexpr bool kNodesInLds = CAP <= 1024;
  QtNode* cur = kNodesInLds ? sh.nodes[0] : Q.nodesA + no;
  QtNode* nxt = kNodesInLds ? sh.nodes[kNodesInLds ? 1 : 0] : Q.nodesB + no;
  unsigned short* proc = sh.proc;
  const int maxX = Q.levW[level] - kBorder, maxY = Q.levH[level] - kBorder;

  // ---- roots (ORBextractor.cc:574-617) ---------------------------------------------------------------
  const int nIni = (int)roundf(static_cast<float>(maxX - kBorder) / (maxY - kBorder));
  const float hX = static_cast<float>(maxX - kBorder) / nIni;
  int m = 0;
  uint32_t seq = 0;
  {
    uint32_t cnt[4];
    tileScan<CAP>(sh, n, nullptr, cand, rankq, false,
                  [&](int, int, uint32_t v) {
                    const int x = (int)(v & 0xfff) - kBorder;
                    return min((int)((float)x / hX), nIni - 1);
                  },
                  cnt);
    uint32_t start[4];
    start[0] = 0; start[1] = cnt[0]; start[2] = cnt[0] + cnt[1]; start[3] = cnt[0] + cnt[1] + cnt[2];
    int rootId[4], nr = 0;
    for (int r = 0; r < 4; r++) rootId[r] = (r < nIni && cnt[r] > 0) ? nr++ : -1;
    for (int p = tid; p < n; p += kQt2Threads) {
      const uint32_t rq = rankq[p];
      const int r = rq >> 30;
      const uint32_t np = start[r] + (rq & 0x3fffffffu);
      valCur[np] = cand[p];
      ownCur[np] = (uint16_t)rootId[r];
    }
    if (tid < 4 && tid < nIni && cnt[tid] > 0) {
      QtNode nd;
      nd.x0 = (short)(int)(hX * static_cast<float>(tid));
      nd.x1 = (short)(int)(hX * static_cast<float>(tid + 1));
      nd.y0 = 0;
      nd.y1 = (short)(maxY - kBorder);
      nd.begin = start[tid];
      nd.end = start[tid] + cnt[tid];
      nd.seq = (uint32_t)tid;
      cur[rootId[tid]] = nd;
    }
    m = nr;
    seq = (uint32_t)nIni;
    __syncthreads();
  }

  bool finalPhase = false;
  int nRec = 0;
  for (int iter = 0; iter < 64; iter++) {
    const int prevSize = m;
    int nproc = 0;
    // ---- processing order + per-node split info ----------------------------------------------------
    if (!finalPhase) {
      int flag[IPT], loc = 0;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        flag[k] = (i < m && cur[i].end - cur[i].begin > 1) ? 1 : 0;
        loc += flag[k];
      }
      int tot;
      int ex = blockScanInt2<CAP>(sh, loc, tot);
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        if (i < m) {
          sh.tproc[i] = (short)(flag[k] ? ex : -1);
          if (flag[k]) proc[ex++] = (unsigned short)i;
        }
      }
      nproc = tot;
    } else {
      blockSortDesc2<CAP>(sh, nRec);
      for (int i = tid; i < m; i += kQt2Threads) sh.tproc[i] = -1;
      __syncthreads();
      for (int t = tid; t < nRec; t += kQt2Threads) {
        const int id = (int)(sh.sortKeys[t] & 0xffff);
        proc[t] = (unsigned short)id;
        sh.tproc[id] = (short)t;
      }
      nproc = nRec;
    }
    __syncthreads();
    if (nproc == 0) break;
    for (int i = tid; i < m; i += kQt2Threads) {
      const QtNode nd = cur[i];
      const int midX = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midY = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
      sh.ninfo[i] = (uint32_t)midX | ((uint32_t)midY << 12) | (sh.tproc[i] >= 0 ? (1u << 24) : 0u);
    }
    __syncthreads();

    // ---- element pass 1: quadrant + segmented scan -------------------------------------------------
    {
      uint32_t dummy[4];
      tileScan<CAP>(sh, n, ownCur, valCur, rankq, true,
                    [&](int, int o, uint32_t v) {
                      const uint32_t info = sh.ninfo[o];
                      if (!(info >> 24)) return 4;
                      const int x = (int)(v & 0xfff) - kBorder, y = (int)((v >> 12) & 0xfff) - kBorder;
                      return (x < (int)(info & 0xfff) ? 0 : 1) + (y < (int)((info >> 12) & 0xfff) ? 0 : 2);
                    },
                    dummy);
    }

    // ---- node pass: children counts, cut, new list positions -----------------------------------------
    int C[IPT], locC = 0, locGrow = 0;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int t = IPT * tid + k;
      C[k] = 0;
      if (t < nproc) {
        const int id = proc[t];
        for (int q = 0; q < 4; q++) C[k] += (sh.endS[id][q] - sh.baseS[id][q]) > 0;
        locC += C[k];
        locGrow += C[k] - 1;
      }
    }
    int totC, totGrow;
    int exC = blockScanInt2<CAP>(sh, locC, totC);
    int exGrow = blockScanInt2<CAP>(sh, locGrow, totGrow);
    int cutT = nproc - 1;
    if (finalPhase) {
      if (tid == 0) sh.s_int[0] = nproc - 1;
      __syncthreads();
      int g = exGrow;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int t = IPT * tid + k;
        if (t < nproc) {
          g += C[k] - 1;
          if (prevSize + g >= N) atomicMin(&sh.s_int[0], t);
        }
      }
      __syncthreads();
      cutT = sh.s_int[0];
    }
    if (tid == 0) sh.s_int[1] = 0;
    __syncthreads();
    {
      int pc = exC;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int t = IPT * tid + k;
        if (t < nproc) {
          if (t == cutT) sh.s_int[1] = pc + C[k];
          pc += C[k];
        }
      }
    }
    __syncthreads();
    const int T = sh.s_int[1];
    int keptFlag[IPT], locK = 0;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int i = IPT * tid + k;
      keptFlag[k] = 0;
      if (i < m) {
        const int t = sh.tproc[i];
        keptFlag[k] = (t < 0 || t > cutT) ? 1 : 0;
      }
      locK += keptFlag[k];
    }
    int totK;
    int exK = blockScanInt2<CAP>(sh, locK, totK);
    const int mNew = T + totK;
    {
      int pc = exC;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int t = IPT * tid + k;
        if (t < nproc) {
          const int id = proc[t];
          if (t <= cutT) {
            const QtNode nd = cur[id];
            const int midX = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midY = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
            uint32_t c[4];
            int nonEmptyAfter = 0;
            for (int q = 0; q < 4; q++) { c[q] = sh.endS[id][q] - sh.baseS[id][q]; nonEmptyAfter += c[q] > 0; }
            const int groupBase = T - pc - C[k];
            uint32_t b = nd.begin;
            int before = 0;
            for (int q = 0; q < 4; q++) {
              if (c[q] == 0) { sh.cpos[id][q] = 0xffff; continue; }
              nonEmptyAfter--;
              const int pos = groupBase + nonEmptyAfter;
              QtNode ch;
              ch.x0 = (q & 1) ? (short)midX : nd.x0;
              ch.x1 = (q & 1) ? nd.x1 : (short)midX;
              ch.y0 = (q & 2) ? (short)midY : nd.y0;
              ch.y1 = (q & 2) ? nd.y1 : (short)midY;
              ch.begin = b;
              ch.end = b + c[q];
              ch.seq = seq + (uint32_t)(pc + before);
              nxt[pos] = ch;
              sh.cpos[id][q] = (uint16_t)pos;
              sh.baseS[id][q] = b - sh.baseS[id][q];   // delta: new position = rank + delta
              b += c[q];
              before++;
            }
          } else {
            sh.tproc[id] = -1;  // beyond the cut: not divided after all (keptFlag was computed before)
          }
          pc += C[k];
        }
      }
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        if (i < m && keptFlag[k]) {
          const int pos = T + exK++;
          nxt[pos] = cur[i];
          sh.cpos[i][0] = (uint16_t)pos;
        }
      }
    }
    __syncthreads();

    // ---- element pass 2: move candidates into their children, re-own (4 independent elements in flight)
    {
      int oN[kEpt];
      uint32_t vN[kEpt], rN[kEpt];
      auto loadTile2 = [&](int b) {
#pragma unroll
        for (int j = 0; j < kEpt; j++) {
          const int p = b + j * kQt2Threads + tid;
          oN[j] = 0; vN[j] = 0; rN[j] = 0;
          if (p < n) { oN[j] = ownCur[p]; vN[j] = valCur[p]; rN[j] = rankq[p]; }
        }
      };
      loadTile2(0);
      for (int b = 0; b < n; b += kEpt * kQt2Threads) {
        int o[kEpt];
        uint32_t v[kEpt], rq[kEpt];
#pragma unroll
        for (int j = 0; j < kEpt; j++) { o[j] = oN[j]; v[j] = vN[j]; rq[j] = rN[j]; }
        if (b + kEpt * kQt2Threads < n) loadTile2(b + kEpt * kQt2Threads);
#pragma unroll
        for (int j = 0; j < kEpt; j++) {
          const int p = b + j * kQt2Threads + tid;
          if (p < n) {
            if (sh.tproc[o[j]] >= 0) {
              const int q = rq[j] >> 30;
              const uint32_t np = (rq[j] & 0x3fffffffu) + sh.baseS[o[j]][q];
              valNxt[np] = v[j];
              ownNxt[np] = sh.cpos[o[j]][q];
            } else {
              valNxt[p] = v[j];
              ownNxt[p] = sh.cpos[o[j]][0];
            }
          }
        }
      }
    }
    __syncthreads();
    { uint32_t* t0 = valCur; valCur = valNxt; valNxt = t0; }
    { uint16_t* t1 = ownCur; ownCur = ownNxt; ownNxt = t1; }
    { QtNode* t2 = cur; cur = nxt; nxt = t2; }
    m = mNew;
    seq += (uint32_t)T;

    // ---- record of nodes created in this pass with more than one candidate (positions [0,T)) --------
    int recFlag[IPT], locR = 0;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int i = IPT * tid + k;
      recFlag[k] = (i < T && cur[i].end - cur[i].begin > 1) ? 1 : 0;
      locR += recFlag[k];
    }
    int totR;
    int exR = blockScanInt2<CAP>(sh, locR, totR);
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int i = IPT * tid + k;
      if (recFlag[k]) {
        const QtNode nd = cur[i];
        sh.sortKeys[exR++] = ((unsigned long long)(nd.end - nd.begin) << 40) | ((unsigned long long)nd.seq << 16) |
                             (unsigned long long)i;
      }
    }
    nRec = totR;
    __syncthreads();

    if (m >= N || m == prevSize) break;
    if (!finalPhase && m + 3 * totR > N) finalPhase = true;
    if (finalPhase && nRec == 0) break;
  }

  // ---- one keypoint per node: highest response, first wins (ORBextractor.cc:774-792) ------------------
  // 16 lanes per node (a node holds about 20 candidates when the quota is reached), four nodes per wave at a time;
  // key = score << 24 | (0xffffff - offset inside the node): the maximum is the highest score at the lowest position
  const int lane = tid & 63, wv = tid >> 6, sub = lane >> 4, sl = lane & 15;
  for (int i0 = wv * 4; i0 < m; i0 += (kQt2Threads / 64) * 4) {
    const int i = i0 + sub;
    uint32_t bestKey = 0, bestVal = 0;
    if (i < m) {
      const QtNode nd = cur[i];
      for (uint32_t p = nd.begin + sl; p < nd.end; p += 16) {
        const uint32_t c = valCur[p];
        const uint32_t key = ((c >> 24) << 24) | (0xffffffu - (p - nd.begin));
        if (key > bestKey) { bestKey = key; bestVal = c; }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const uint32_t ok = __shfl_xor(bestKey, o, 64), ov = __shfl_xor(bestVal, o, 64);
      if (ok > bestKey) { bestKey = ok; bestVal = ov; }
    }
    if (i < m && sl == 0) {
      const uint32_t c = bestVal;
      SelKp s;
      s.xy = (c & 0xfff) | (((c >> 12) & 0xfff) << 16);
      s.lf = (uint32_t)level | ((uint32_t)f << 8) | ((c >> 24) << 24);
      selOut[i] = s;
    }
  }
  if (tid == 0) *selCount = (uint32_t)m;
}

void launch_quadtree(const QtParams& Q, int nframes, hipStream_t st) {
  int maxN = 0;
  for (int l = 0; l < Q.nlevels; l++) maxN = Q.nfeat[l] > maxN ? Q.nfeat[l] : maxN;
  // the node-table capacity sets the block's LDS footprint (50 / 99 / 116 KB), and with it how many waves of the
  // kernels that run beside the quadtree still fit on the CU
  if (maxN + 4 <= 512)
    hipLaunchKernelGGL(k_quadtree2<512>, dim3(Q.nlevels, nframes), dim3(kQt2Threads), 0, st, Q);
  else if (maxN + 4 <= 1024)
    hipLaunchKernelGGL(k_quadtree2<1024>, dim3(Q.nlevels, nframes), dim3(kQt2Threads), 0, st, Q);
  else
    hipLaunchKernelGGL(k_quadtree2<2048>, dim3(Q.nlevels, nframes), dim3(kQt2Threads), 0, st, Q);
}

}  // namespace orbfe
