// orbfe_quadtree.hip -- GPU-resident keypoint thinning (DistributeOctTree) for gfx950.
//
// Behaviour contract: ORBextractor::DistributeOctTree + ExtractorNode::DivideNode of the reference
// (src/ORBextractor.cc:513-569, 571-795), same tie-break convention as quadtree.h / the oracle
// (equal-sized nodes: later-created first).  One workgroup per
// (frame, level) problem (k_quadtree3, 512 threads); problems of a batch run concurrently on different CUs.
//
// The reference algorithm is a sequential walk over a std::list that moves keypoints from node to node, but its
// result is a pure function of
//   * which node every candidate belongs to after each pass -- geometry only (DivideNode halves boxes with ceil);
//   * how many candidates each child of a divided node receives -- counts, no order;
//   * the list order of the nodes: children are pushed to the FRONT in n1..n4 order and parents erased, so after
//     processing nodes D[0..k) (in processing order) the new list is
//       [children(D[k-1]) reversed, ..., children(D[0]) reversed] ++ [old list without D];
//     node ids are re-assigned to list positions after every pass, so "the list" is an array;
//   * final phase: nodes created in the previous round with >1 candidate are processed by (size desc, creation desc)
//     and the walk stops right after the division that makes size >= N: children counts of ALL of them are computed
//     first, a prefix sum over the sorted order finds the cut, and only the prefix is applied;
//   * the keypoint kept per node: highest response, FIRST in the node's vKeys order on ties (:774-792).  DivideNode
//     partitions vKeys stably, so "first in vKeys order" = lowest index in the level's candidate list.
// Candidates therefore never move (round 2; the first version re-partitioned the candidate array in every pass, which
// serialised each pass over 2048-candidate tiles): every candidate carries the id of its node (u16 in HBM), a pass is
// one fully parallel sweep -- new node id through the previous pass's child table, quadrant in the new node, one LDS
// atomic on the node's child counter -- and the final pick is an LDS atomic max on (score, lowest index) keys.
// Creation order (seq) = processing order, n1..n4 inside a parent.
#include <mutex>
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "orbfe_internal.h"

namespace orbfe {

constexpr int kQt3Threads = 512;

// Experiments build only: s_memrealtime stamps (100 MHz) of one block's phases, one row of 32 per (frame 0) level: tools/qt_phases.py
#ifdef ORBFE_EXPERIMENTS
__device__ unsigned long long g_qtStamps[kMaxLevels][32];
#define QT_STAMP(i)                                                                                         \
  do {                                                                                                      \
    if (threadIdx.x == 0 && blockIdx.y == 0) g_qtStamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define QT_STAMP(i) ((void)0)
#endif
constexpr int kQtEpt = 8;   // candidates per thread and sweep step (independent loads in flight)

struct Qt3Node {
  short x0, x1, y0, y1;    // UL.x, UR.x, UL.y, BL.y
  uint32_t size;           // candidates in the node
  uint32_t seq;            // creation order
};

template <int CAP>
struct Qt3Shared {
  Qt3Node nodes[2][CAP];
  unsigned short proc[CAP];   // node ids in processing order
  uint32_t ninfo[CAP];        // (midX + border) | (midY + border) << 12 | divide << 24
  uint16_t cpos[CAP][4];      // new list position of a candidate of node id that sits in quadrant q (kept nodes: all four equal)
  // LDS atomics of the sweeps.  64 lanes adding to one address are served one after the other (about 130 cycles per
  // wave instruction measured), and neighbouring candidates do share their node while the list is short, so every
  // counter is replicated R = 2^rlog times in consecutive words (different banks) and lane L uses replica L % R:
  //   child counts   cnt[((id * 4 + q) << rlog) + rep], R = largest power of two <= 32 with nodes * R <= CAP; the
  //                  node pass reads replica 0 after a fold;
  //   final pick     max of score << 24 | (0xffffff - candidate index) in cnt[id * 4 + rep].
  alignas(16) uint32_t cnt[CAP * 4];
  alignas(16) unsigned long long sortKeys[CAP];
  short tproc[CAP];           // index in processing order, -1 = not divided
  int wsumI[kQt3Threads / 64];
  int s_int[4];
};

__device__ __forceinline__ int wave_incl_scan_i(int x) {   // DPP: six dependent adds instead of six LDS-crossbar shuffles
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return x;
}

template <int CAP>
__device__ int blockScanInt3(Qt3Shared<CAP>& sh, int v, int& total) {  // exclusive, 512 threads
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int iv = wave_incl_scan_i(v);
  __syncthreads();
  if (lane == 63) sh.wsumI[wv] = iv;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kQt3Threads / 64; w++) {
    if (w < wv) base += sh.wsumI[w];
    tot += sh.wsumI[w];
  }
  total = tot;
  return base + iv - v;
}

template <int CAP>
__device__ void blockSortDesc3(Qt3Shared<CAP>& sh, int n) {
  int m = 1;
  while (m < n) m <<= 1;
  if (n <= kQt3Threads) {
    // RANK sort (round 6): the keys are distinct (they carry the creation number), so key i belongs at position #{j : key j > key i}.
    // 512 / m threads share a key (m = n rounded up to a power of two) and count over their slice of the others -- LDS reads of
    // one address for all lanes of a wave (broadcasts), two keys per 16-byte read, eight reads in flight -- then add their counts
    // with one LDS atomic each: a few hundred independent instructions per thread instead of the bitonic network's 28 .. 45 DEPENDENT
    // exchange steps through the LDS crossbar (3.8 us for the 128 keys of a 1080p level, profiles/r06_qt_phases.txt).
    const int t = threadIdx.x, i = t & (m - 1), part = t / m, parts = kQt3Threads / m;   // m <= 512: parts >= 1
    const unsigned long long key = i < n ? sh.sortKeys[i] : 0ull;
    uint32_t* rankAcc = sh.cnt;   // (not live between passes)
    if (t < n) rankAcc[t] = 0u;
    __syncthreads();
    if (i < n) {
      const int per = ((n + parts - 1) / parts + 15) & ~15;   // keys per slice, whole trips
      const int j1 = min(n, (part + 1) * per);
      int rank = 0;
      for (int j0 = part * per; j0 < j1; j0 += 16) {
        ulonglong2 kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) kk[u] = *reinterpret_cast<const ulonglong2*>(&sh.sortKeys[min(j0 + 2 * u, CAP - 2)]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          rank += (j0 + 2 * u < j1) && (kk[u].x > key);
          rank += (j0 + 2 * u + 1 < j1) && (kk[u].y > key);
        }
      }
      if (rank) atomicAdd(&rankAcc[i], (uint32_t)rank);
    }
    __syncthreads();
    const int pos = (t < n) ? (int)rankAcc[t] : 0;
    const unsigned long long mine = (t < n) ? sh.sortKeys[t] : 0ull;
    __syncthreads();
    if (t < n) sh.sortKeys[pos] = mine;
    __syncthreads();
    return;
  }
  for (int i = n + threadIdx.x; i < m; i += kQt3Threads) sh.sortKeys[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += kQt3Threads) {
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

// Candidate words and node ids either stay in HBM (LC = false: batches, where LDS is better spent on the occupancy of
// the kernels running beside the quadtree) or are copied into LDS by the root pass (LC = true: latency-bound one- or
// two-frame calls; a sweep then costs LDS round trips instead of L2 ones).
template <bool LC>
struct Qt3Cands {
  const uint32_t* cg;
  uint16_t* og;
  uint32_t* cl;
  uint16_t* ol;
  __device__ __forceinline__ uint32_t cand(int p) const { return LC ? cl[p] : cg[p]; }
  __device__ __forceinline__ unsigned own(int p) const { return LC ? ol[p] : og[p]; }
  __device__ __forceinline__ void setOwn(int p, unsigned v) const {
    if (LC) ol[p] = (uint16_t)v;
    else og[p] = (uint16_t)v;
  }
};

template <int CAP, bool LC>
__device__ __forceinline__ void qt3_problem(const QtParams& Q, Qt3Shared<CAP>& sh, uint32_t* candL, uint16_t* ownL, const int level, const int f,
                                            const long long first, const int n) {
  constexpr int IPT = CAP >= kQt3Threads ? CAP / kQt3Threads : 1;   // node items per thread
  const int tid = threadIdx.x;
  const int N = Q.nfeat[level];
  uint32_t* selCount = Q.selCount + (long long)f * kMaxLevels + level;
  uint32_t* selCountHost = Q.selCountHost ? Q.selCountHost + (long long)f * kMaxLevels + level : nullptr;
  SelKp* selHost = Q.selHost ? Q.selHost + (long long)f * Q.selPerFrame + Q.selOff[level] : nullptr;
  SelKp* selOut = Q.sel + (long long)f * Q.selPerFrame + Q.selOff[level];
  QT_STAMP(0);
  if (n <= 0) {
    if (tid == 0) {
      *selCount = 0;
      if (selCountHost) *selCountHost = 0;
    }
    return;
  }
  const long long eo = (long long)f * Q.candCap + first;
  const uint32_t* cand = Q.cand + eo;
  // node id * 4 + quadrant inside it, per candidate (= index into the child table)
  const Qt3Cands<LC> ca{cand, Q.own + eo, candL, ownL};
  Qt3Node* cur = sh.nodes[0];
  Qt3Node* nxt = sh.nodes[1];
  unsigned short* proc = sh.proc;
  const int maxX = Q.levW[level] - kBorder, maxY = Q.levH[level] - kBorder;
  // ---- roots (ORBextractor.cc:574-617): raw root index per candidate, counts by LDS atomics ------------------
  const int nIni = (int)roundf(static_cast<float>(maxX - kBorder) / (maxY - kBorder));
  const float hX = static_cast<float>(maxX - kBorder) / nIni;
  // root of a candidate: min((int)((float)x / hX), nIni - 1) (ORBextractor.cc:591-594), monotone in x, so it is the number of
  // thresholds b_k = smallest x whose quotient reaches k that lie at or below x: three compares instead of an IEEE division per candidate
  int rootB[3];
#pragma unroll
  for (int k = 1; k <= 3; k++) {
    int b = 0x7fffffff;
    if (k < nIni) {
      b = (int)(hX * static_cast<float>(k));   // near the threshold; settle it by the reference's own expression
      while (b > 0 && (int)((float)(b - 1) / hX) >= k) b--;
      while ((int)((float)b / hX) < k) b++;
    }
    rootB[k - 1] = b;
  }
  auto rootOf = [&](int x) { return (x >= rootB[0]) + (x >= rootB[1]) + (x >= rootB[2]); };
  int m = 0;
  uint32_t seq = 0;
  bool finalPhase = false;
  int nRec = 0;
  // ---- the first passes of a dense level in one sweep -------------------------------------------------------------
  // While every node holds more than one candidate and all four children of every node are non-empty, the walk's result
  // after pass t is known in closed form: the list has nIni * 4^t nodes, the child q of the node at list position i is
  // created (4 i + q)-th in its pass and -- children go to the FRONT -- sits at position (m_t - 1) - (4 i + q); boxes follow
  // from DivideNode's ceil-halving along the path.  So instead of the root pass and T full sweeps (new node through the
  // child table, quadrant, one count each) ONE sweep walks every candidate T levels down by arithmetic and counts it at its
  // leaf.  T = the passes the reference runs before its final phase when every node divides (pass t + 1 is an ordinary one
  // iff t == 0 or 4 m_t <= N, ORBextractor.cc:642-731); the jump is taken only if every leaf received a candidate,
  // otherwise the ordinary passes start from the roots as always.
  bool jumped = false;
  int T = 0, mT = nIni;
  if (Q.jump) {
    int mt = nIni;
    while (T < 3 && (T == 0 || 4 * mt <= N) && mt * 4 <= CAP) { mt *= 4; T++; }
    mT = mt;
  }
  QT_STAMP(1);
  if (T >= 2 && n >= 8 * mT) {
    int rl = 5;
    while (rl > 0 && (mT << rl) > CAP * 4) rl--;
    for (int i = tid; i < (mT << rl); i += kQt3Threads) sh.cnt[i] = 0u;
    const int yTop = maxY - kBorder;
    // DivideNode halves the two axes independently, so a candidate's leaf follows from a per-COLUMN and a per-ROW table: root and
    // the T left / right decisions for every x of the level, the T up / down decisions for every y, and one table from (root,
    // decisions) to the leaf's creation index.  The walk itself -- T rounds of midpoints, compares and selects, 55 vector
    // instructions -- then runs once per column and row instead of once per candidate (round 6: this sweep is bound by ONE CU's
    // vector issue; 8.2 -> 3.x us for the 10 400 candidates of a 1080p level 0, profiles/r06_qt_phases.txt).  The tables live in
    // the second node array, which nothing uses before the first node pass.
    uint8_t* lutX = reinterpret_cast<uint8_t*>(sh.nodes[1]);                 // [levW]  root << 3 | x decisions (first = MSB), by RAW x (border included)
    uint8_t* lutY = lutX + ((Q.levW[level] + 15) & ~15);                      // [levH]  y decisions
    static_assert(sizeof(sh.nodes[1]) >= 8192 || CAP < 512, "axis tables of a 4095-pixel level");
    uint16_t* lutC = reinterpret_cast<uint16_t*>(sh.proc);                    // [4][8][8] creation index of the leaf
    const bool lutFits = ((Q.levW[level] + 15) & ~15) + Q.levH[level] <= (int)sizeof(sh.nodes[1]);
    // the candidates of the NEXT step travel while the current one is processed; the first request goes out before the tables are built
    uint32_t vNext[kQtEpt];
    auto loadJump = [&](int b) {
#pragma unroll
      for (int j = 0; j < kQtEpt; j++) vNext[j] = cand[min(b + j * kQt3Threads + tid, n - 1)];
    };
    loadJump(0);
    if (lutFits) {
      for (int xr = kBorder + tid; xr < maxX; xr += kQt3Threads) {
        const int x = xr - kBorder, r = rootOf(x);
        int x0 = (int)(hX * static_cast<float>(r)), x1 = (int)(hX * static_cast<float>(r + 1)), bits = 0;
        for (int t = 1; t <= T; t++) {
          const int midX = x0 + ((x1 - x0 + 1) >> 1), qx = x < midX ? 0 : 1;
          x0 = qx ? midX : x0; x1 = qx ? x1 : midX;
          bits = bits * 2 + qx;
        }
        lutX[xr] = (uint8_t)((r << 3) | bits);
      }
      for (int yr = kBorder + tid; yr < maxY; yr += kQt3Threads) {
        const int y = yr - kBorder;
        int y0 = 0, y1 = yTop, bits = 0;
        for (int t = 1; t <= T; t++) {
          const int midY = y0 + ((y1 - y0 + 1) >> 1), qy = y < midY ? 0 : 1;
          y0 = qy ? midY : y0; y1 = qy ? y1 : midY;
          bits = bits * 2 + qy;
        }
        lutY[yr] = (uint8_t)bits;
      }
      if (tid < 256) {   // (root, x decisions, y decisions) -> creation index at pass T
        const int r = tid >> 6, bx = (tid >> 3) & 7, by = tid & 7;
        int pos = r, mt = nIni, created = 0;
        for (int t = 1; t <= T; t++) {
          const int q = ((bx >> (T - t)) & 1) + 2 * ((by >> (T - t)) & 1);
          created = 4 * pos + q;
          mt *= 4;
          if (t < T) pos = (mt - 1) - created;
        }
        lutC[tid] = (uint16_t)created;
      }
    }
    __syncthreads();
    for (int b = 0; b < n; b += kQtEpt * kQt3Threads) {
      uint32_t v[kQtEpt];
#pragma unroll
      for (int j = 0; j < kQtEpt; j++) v[j] = vNext[j];
      if (b + kQtEpt * kQt3Threads < n) loadJump(b + kQtEpt * kQt3Threads);
      if (lutFits) {
        unsigned ex[kQtEpt], ey[kQtEpt], cr[kQtEpt];
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) { ex[j] = lutX[v[j] & 0xfff]; ey[j] = lutY[(v[j] >> 12) & 0xfff]; }
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) cr[j] = lutC[((ex[j] << 3) | ey[j]) & 255u];
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) {
          const int p = b + j * kQt3Threads + tid;
          if (p < n) {
            if (LC) candL[p] = v[j];
            ca.setOwn(p, cr[j]);          // parent's position * 4 + quadrant: what an ordinary sweep leaves
            atomicAdd(&sh.cnt[(((unsigned)(mT - 1) - cr[j]) << rl) + (tid & ((1 << rl) - 1))], 1u);
          }
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < kQtEpt; j++) {
        const int p = b + j * kQt3Threads + tid;
        if (p < n) {
          const int x = (int)(v[j] & 0xfff) - kBorder, y = (int)((v[j] >> 12) & 0xfff) - kBorder;
          const int r = rootOf(x);
          int x0 = (int)(hX * static_cast<float>(r)), x1 = (int)(hX * static_cast<float>(r + 1)), y0 = 0, y1 = yTop;
          int pos = r, mt = nIni, created = 0, q = 0;
          for (int t = 1; t <= T; t++) {
            const int midX = x0 + ((x1 - x0 + 1) >> 1), midY = y0 + ((y1 - y0 + 1) >> 1);
            const int qx = x < midX ? 0 : 1, qy = y < midY ? 0 : 1;
            q = qx + 2 * qy;
            x0 = qx ? midX : x0; x1 = qx ? x1 : midX; y0 = qy ? midY : y0; y1 = qy ? y1 : midY;
            created = 4 * pos + q;
            mt *= 4;
            if (t < T) pos = (mt - 1) - created;
          }
          if (LC) candL[p] = v[j];
          ca.setOwn(p, (unsigned)created);          // parent's position * 4 + quadrant: what an ordinary sweep leaves
          atomicAdd(&sh.cnt[(((mT - 1) - created) << rl) + (tid & ((1 << rl) - 1))], 1u);
        }
      }
    }
    __syncthreads();
    QT_STAMP(2);
    int empty = 0;
    for (int g = tid; g < mT; g += kQt3Threads) {
      uint32_t t = 0;
      for (int k = 0; k < (1 << rl); k++) t += sh.cnt[(g << rl) + k];
      sh.cnt[g << rl] = t;
      empty |= t == 0u;
    }
    jumped = __syncthreads_or(empty) == 0;
    QT_STAMP(3);
    if (jumped) {
      // the nodes of the list after pass T, their child table, and the record of those with more than one candidate
      const int mPrev = mT >> 2;
      uint32_t seqBase = (uint32_t)nIni;
      for (int t = 1, mt = nIni; t < T; t++) { mt *= 4; seqBase += (uint32_t)mt; }
      for (int p = tid; p < mT; p += kQt3Threads) {
        int digits[3], c = (mT - 1) - p, mt = mT;
        for (int t = T; t >= 1; t--) {      // created index at pass t -> quadrant and the parent's position
          digits[t - 1] = c & 3;
          const int parentPos = c >> 2;
          mt >>= 2;
          c = t > 1 ? (mt - 1) - parentPos : parentPos;   // (pass 0: the position is the root index)
        }
        const int r = c;
        int x0 = (int)(hX * static_cast<float>(r)), x1 = (int)(hX * static_cast<float>(r + 1)), y0 = 0, y1 = yTop;
        for (int t = 0; t < T; t++) {
          const int midX = x0 + ((x1 - x0 + 1) >> 1), midY = y0 + ((y1 - y0 + 1) >> 1);
          const int qd = digits[t];
          x0 = (qd & 1) ? midX : x0; x1 = (qd & 1) ? x1 : midX; y0 = (qd & 2) ? midY : y0; y1 = (qd & 2) ? y1 : midY;
        }
        Qt3Node nd;
        nd.x0 = (short)x0; nd.x1 = (short)x1; nd.y0 = (short)y0; nd.y1 = (short)y1;
        nd.size = sh.cnt[p << rl];
        nd.seq = seqBase + (uint32_t)((mT - 1) - p);
        cur[p] = nd;
      }
      for (int i = tid; i < mPrev * 4; i += kQt3Threads) (&sh.cpos[0][0])[i] = (uint16_t)((mT - 1) - i);
      __syncthreads();
      m = mT;
      seq = seqBase + (uint32_t)mT;
      int recFlag[IPT], locR = 0;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        recFlag[k] = (i < mT && cur[i].size > 1) ? 1 : 0;
        locR += recFlag[k];
      }
      int totR;
      int exR = blockScanInt3<CAP>(sh, locR, totR);
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        if (recFlag[k]) {
          const Qt3Node nd = cur[i];
          sh.sortKeys[exR++] = ((unsigned long long)nd.size << 40) | ((unsigned long long)nd.seq << 16) | (unsigned long long)i;
        }
      }
      nRec = totR;
      __syncthreads();
      finalPhase = m + 3 * totR > N;   // (what the end of pass T decides; m < N or m == N are handled by the loop's entry test below)
      QT_STAMP(4);
    }
  }
  if (!jumped) {
    if (tid < 128) sh.cnt[tid] = 0u;
    __syncthreads();
    for (int b = 0; b < n; b += kQtEpt * kQt3Threads) {
      uint32_t v[kQtEpt];
#pragma unroll
      for (int j = 0; j < kQtEpt; j++) {
        const int p = b + j * kQt3Threads + tid;
        v[j] = p < n ? cand[p] : 0u;
      }
#pragma unroll
      for (int j = 0; j < kQtEpt; j++) {
        const int p = b + j * kQt3Threads + tid;
        if (p < n) {
          const int x = (int)(v[j] & 0xfff) - kBorder;
          const int r = rootOf(x);
          if (LC) candL[p] = v[j];
          ca.setOwn(p, (unsigned)r << 2);
          atomicAdd(&sh.cnt[r * 32 + (tid & 31)], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 4) {
      uint32_t t = 0;
      for (int k = 0; k < 32; k++) t += sh.cnt[tid * 32 + k];
      sh.s_int[tid] = (int)t;
    }
    __syncthreads();
    uint32_t cnt[4];
    for (int r = 0; r < 4; r++) cnt[r] = (uint32_t)sh.s_int[r];
    int rootId[4], nr = 0;
    for (int r = 0; r < 4; r++) rootId[r] = (r < nIni && cnt[r] > 0) ? nr++ : -1;
    __syncthreads();
    if (tid < 4) {
      // the raw roots act as "kept nodes" of a pass zero: a candidate of raw root r continues in list position rootId[r]
      for (int q = 0; q < 4; q++) sh.cpos[tid][q] = (uint16_t)(rootId[tid] < 0 ? 0 : rootId[tid]);
      if (tid < nIni && cnt[tid] > 0) {
        Qt3Node nd;
        nd.x0 = (short)(int)(hX * static_cast<float>(tid));
        nd.x1 = (short)(int)(hX * static_cast<float>(tid + 1));
        nd.y0 = 0;
        nd.y1 = (short)(maxY - kBorder);
        nd.size = cnt[tid];
        nd.seq = (uint32_t)tid;
        cur[rootId[tid]] = nd;
      }
    }
    m = nr;
    seq = (uint32_t)nIni;
    __syncthreads();
  }
  // (after a jump the walk continues where pass T's end-of-pass tests leave it: done when the list reached N -- ORBextractor.cc:
  // 642-650 -- or nothing is left to divide)
  for (int iter = 0; iter < 64; iter++) {
    if (jumped && iter == 0 && (m >= N || (finalPhase && nRec == 0))) break;
    const int prevSize = m;
    int nproc = 0;
    // ---- processing order + per-node split info ----------------------------------------------------
    if (!finalPhase) {
      int flag[IPT], loc = 0;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        flag[k] = (i < m && cur[i].size > 1) ? 1 : 0;
        loc += flag[k];
      }
      int tot;
      int ex = blockScanInt3<CAP>(sh, loc, tot);
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
      blockSortDesc3<CAP>(sh, nRec);
      for (int i = tid; i < m; i += kQt3Threads) sh.tproc[i] = -1;
      __syncthreads();
      for (int t = tid; t < nRec; t += kQt3Threads) {
        const int id = (int)(sh.sortKeys[t] & 0xffff);
        proc[t] = (unsigned short)id;
        sh.tproc[id] = (short)t;
      }
      nproc = nRec;
    }
    __syncthreads();
    if (nproc == 0) break;
    if (iter < 2) QT_STAMP(5 + 5 * iter);
    for (int i = tid; i < m; i += kQt3Threads) {
      const Qt3Node nd = cur[i];
      const int midX = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midY = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
      // image coordinates (candidate words hold x, y with the border included), so the sweep compares fields directly
      sh.ninfo[i] = (uint32_t)(midX + kBorder) | ((uint32_t)(midY + kBorder) << 12) | (sh.tproc[i] >= 0 ? (1u << 24) : 0u);
    }
    int rlog = 5;   // counter replicas of this pass
    while (rlog > 0 && (m << rlog) > CAP) rlog--;
    for (int i = tid; i < ((m * 4) << rlog); i += kQt3Threads) sh.cnt[i] = 0u;
    __syncthreads();
    if (iter < 2) QT_STAMP(6 + 5 * iter);
    // ---- candidate sweep: node id in the current list (through the previous pass's child table), quadrant inside
    //      a node that is divided in this pass, one count per candidate ----------------------------------------
    {
      uint32_t vN[kQtEpt];
      unsigned owN[kQtEpt];
      auto loadStep = [&](int b) {   // the next step's candidates travel while the current step is processed
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) {
          // not predicated: all loads of a step issue back to back; the index is clamped to the last candidate in both
          // modes (LDS-resident or HBM), so no read leaves the problem's arrays
          const int p = min(b + j * kQt3Threads + tid, n - 1);
          owN[j] = ca.own(p);
          vN[j] = ca.cand(p);
        }
      };
      loadStep(0);
      for (int b = 0; b < n; b += kQtEpt * kQt3Threads) {
        uint32_t v[kQtEpt];
        unsigned ow[kQtEpt];
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) { v[j] = vN[j]; ow[j] = owN[j]; }
        if (b + kQtEpt * kQt3Threads < n) loadStep(b + kQtEpt * kQt3Threads);
        // three dependent LDS round trips per step, not per candidate: all table reads of a stage are issued before
        // the first atomic / store of the step (which the compiler must assume to alias them)
        unsigned id[kQtEpt];
        uint32_t info[kQtEpt];
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) id[j] = (&sh.cpos[0][0])[ow[j] & (CAP * 4 - 1)];
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) info[j] = sh.ninfo[id[j] & (CAP - 1)];
#pragma unroll
        for (int j = 0; j < kQtEpt; j++) {
          const int p = b + j * kQt3Threads + tid;
          if (p < n) {
            unsigned q = 0;
            if (info[j] >> 24) {   // the node is divided in this pass: one count for the child the candidate falls into
              q = ((v[j] & 0xfff) < (info[j] & 0xfff) ? 0u : 1u) + ((v[j] & 0xfff000) < (info[j] & 0xfff000) ? 0u : 2u);
              atomicAdd(&sh.cnt[((id[j] * 4 + q) << rlog) + (tid & ((1 << rlog) - 1))], 1u);
            }
            ca.setOwn(p, id[j] * 4 + q);
          }
        }
      }
    }
    __syncthreads();
    if (iter < 2) QT_STAMP(7 + 5 * iter);
    if (rlog > 0) {   // fold the replicas into replica 0 (a group is touched by one thread only)
      for (int g = tid; g < m * 4; g += kQt3Threads) {
        uint32_t t = 0;
        for (int k = 0; k < (1 << rlog); k++) t += sh.cnt[(g << rlog) + k];
        sh.cnt[g << rlog] = t;
      }
      __syncthreads();
    }

    if (iter < 2) QT_STAMP(8 + 5 * iter);
    // ---- node pass: children counts, cut, new list positions -----------------------------------------
    int C[IPT], locC = 0, locGrow = 0;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int t = IPT * tid + k;
      C[k] = 0;
      if (t < nproc) {
        const int id = proc[t];
        for (int q = 0; q < 4; q++) C[k] += sh.cnt[(id * 4 + q) << rlog] > 0;
        locC += C[k];
        locGrow += C[k] - 1;
      }
    }
    int totC, totGrow;
    int exC = blockScanInt3<CAP>(sh, locC, totC);
    int exGrow = blockScanInt3<CAP>(sh, locGrow, totGrow);
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
    int exK = blockScanInt3<CAP>(sh, locK, totK);
    const int mNew = T + totK;
    {
      int pc = exC;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int t = IPT * tid + k;
        if (t < nproc) {
          const int id = proc[t];
          if (t <= cutT) {
            const Qt3Node nd = cur[id];
            const int midX = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midY = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
            uint32_t c[4];
            int nonEmptyAfter = 0;
            for (int q = 0; q < 4; q++) { c[q] = sh.cnt[(id * 4 + q) << rlog]; nonEmptyAfter += c[q] > 0; }
            const int groupBase = T - pc - C[k];
            int before = 0;
            for (int q = 0; q < 4; q++) {
              if (c[q] == 0) { sh.cpos[id][q] = 0xffff; continue; }
              nonEmptyAfter--;
              const int pos = groupBase + nonEmptyAfter;
              Qt3Node ch;
              ch.x0 = (q & 1) ? (short)midX : nd.x0;
              ch.x1 = (q & 1) ? nd.x1 : (short)midX;
              ch.y0 = (q & 2) ? (short)midY : nd.y0;
              ch.y1 = (q & 2) ? nd.y1 : (short)midY;
              ch.size = c[q];
              ch.seq = seq + (uint32_t)(pc + before);
              nxt[pos] = ch;
              sh.cpos[id][q] = (uint16_t)pos;
              before++;
            }
          }
          pc += C[k];
        }
      }
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const int i = IPT * tid + k;
        if (i < m && keptFlag[k]) {   // not divided (or beyond the final phase's cut): the node moves behind the new children
          const int pos = T + exK++;
          nxt[pos] = cur[i];
          for (int q = 0; q < 4; q++) sh.cpos[i][q] = (uint16_t)pos;
        }
      }
    }
    __syncthreads();
    { Qt3Node* t2 = cur; cur = nxt; nxt = t2; }
    m = mNew;
    seq += (uint32_t)T;
    // ---- record of nodes created in this pass with more than one candidate (positions [0,T)) --------
    int recFlag[IPT], locR = 0;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int i = IPT * tid + k;
      recFlag[k] = (i < T && cur[i].size > 1) ? 1 : 0;
      locR += recFlag[k];
    }
    int totR;
    int exR = blockScanInt3<CAP>(sh, locR, totR);
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const int i = IPT * tid + k;
      if (recFlag[k]) {
        const Qt3Node nd = cur[i];
        sh.sortKeys[exR++] = ((unsigned long long)nd.size << 40) | ((unsigned long long)nd.seq << 16) | (unsigned long long)i;
      }
    }
    nRec = totR;
    __syncthreads();
    if (iter < 2) QT_STAMP(9 + 5 * iter);
    if (m >= N || m == prevSize) break;
    if (!finalPhase && m + 3 * totR > N) finalPhase = true;
    if (finalPhase && nRec == 0) break;
  }

  QT_STAMP(15);
  // ---- one keypoint per node: highest response, first in candidate order wins (ORBextractor.cc:774-792) --------
  __syncthreads();
  for (int i = tid; i < m * 4; i += kQt3Threads) sh.cnt[i] = 0u;
  __syncthreads();
  for (int b = 0; b < n; b += kQtEpt * kQt3Threads) {
    uint32_t v[kQtEpt];
    unsigned ow[kQtEpt];
#pragma unroll
    for (int j = 0; j < kQtEpt; j++) {
      const int p = b + j * kQt3Threads + tid;
      v[j] = 0u;
      ow[j] = 0u;
      if (p < n) { ow[j] = ca.own(p); v[j] = ca.cand(p); }
    }
    unsigned id[kQtEpt];
#pragma unroll
    for (int j = 0; j < kQtEpt; j++) id[j] = (&sh.cpos[0][0])[ow[j] & (CAP * 4 - 1)];
#pragma unroll
    for (int j = 0; j < kQtEpt; j++) {
      const int p = b + j * kQt3Threads + tid;
      if (p < n) atomicMax(&sh.cnt[id[j] * 4 + (tid & 3)], ((v[j] >> 24) << 24) | (0xffffffu - (uint32_t)p));
    }
  }
  __syncthreads();
  QT_STAMP(16);
  for (int i = tid; i < m; i += kQt3Threads) {
    const uint4 b4 = *reinterpret_cast<const uint4*>(&sh.cnt[i * 4]);
    const uint32_t best = max(max(b4.x, b4.y), max(b4.z, b4.w));
    const uint32_t c = ca.cand((int)(0xffffffu - (best & 0xffffffu)));
    SelKp s;
    s.xy = (c & 0xfff) | (((c >> 12) & 0xfff) << 16);
    s.lf = (uint32_t)level | ((uint32_t)f << 8) | ((c >> 24) << 24);
    selOut[i] = s;
    if (selHost) selHost[i] = s;
  }
  QT_STAMP(17);
  if (tid == 0) {
    *selCount = (uint32_t)m;
    if (selCountHost) *selCountHost = (uint32_t)m;
  }
}

template <int CAP>
__global__ __launch_bounds__(kQt3Threads) void k_quadtree3(QtParams Q, int ldsCand) {
  __shared__ Qt3Shared<CAP> sh;
  extern __shared__ __align__(16) uint32_t qtDyn[];   // [ldsCand] candidate words, then [ldsCand] u16 node ids
  const int level = blockIdx.x, f = Q.frameBase + blockIdx.y;
  const uint32_t* ls = Q.levelStart + (long long)f * (kMaxLevels + 1);
  // packed list (k_compact: prefix offsets) or level-local lists (k_compact_local: fixed bases, lengths)
  const long long first = Q.levelLocal ? Q.candBase[level] : (long long)ls[level];
  const int n = Q.levelLocal ? (int)ls[level] : (int)(ls[level + 1] - ls[level]);
  if (n <= ldsCand) qt3_problem<CAP, true>(Q, sh, qtDyn, reinterpret_cast<uint16_t*>(qtDyn + ldsCand), level, f, first, n);
  else qt3_problem<CAP, false>(Q, sh, nullptr, nullptr, level, f, first, n);
}

// ldsBudget: bytes of LDS one problem may take for its candidates on top of the node tables (0 = keep them in HBM)
template <int CAP>
static int launch_qt3(const QtParams& Q, int nframes, hipStream_t st, int ldsBudget) {
  int ldsCand = 0;
  const int room = 156 * 1024 - (int)sizeof(Qt3Shared<CAP>);
  if (ldsBudget > room) ldsBudget = room;
  if (ldsBudget >= 6 * 1024) ldsCand = (ldsBudget / 6) & ~3;
  const int dynBytes = ldsCand * 6;
  if (dynBytes) {
    // handles on different host threads launch this kernel: the (check, set, record) sequence is one critical section,
    // so nobody launches on the strength of a record whose hipFuncSetAttribute has not returned yet
    static std::mutex mu;
    static int attrBytes[64] = {};   // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    std::lock_guard<std::mutex> lk(mu);
    if (dynBytes > attrBytes[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree3<CAP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              dynBytes) != hipSuccess)
        return 1;
      attrBytes[dev] = dynBytes;
    }
  }
  hipLaunchKernelGGL(k_quadtree3<CAP>, dim3(Q.nlevels, nframes), dim3(kQt3Threads), dynBytes, st, Q, ldsCand);
  return 0;
}

int launch_quadtree(const QtParams& Q, int nframes, hipStream_t st, int ldsBudget) {
  int maxN = 0;
  for (int l = 0; l < Q.nlevels; l++) maxN = Q.nfeat[l] > maxN ? Q.nfeat[l] : maxN;
  // the node-table capacity sets the block's LDS footprint (36 / 72 / 144 KB), and with it how many waves of the
  // kernels that run beside the quadtree still fit on the CU
  if (maxN + 4 <= 512) return launch_qt3<512>(Q, nframes, st, ldsBudget);
  if (maxN + 4 <= 1024) return launch_qt3<1024>(Q, nframes, st, ldsBudget);
  return launch_qt3<2048>(Q, nframes, st, ldsBudget);
}

}  // namespace orbfe

#ifdef ORBFE_EXPERIMENTS
extern "C" int orbfe_exp_qt_stamps(unsigned long long* out /* [kMaxLevels][32] */, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(orbfe::g_qtStamps), sizeof(unsigned long long) * orbfe::kMaxLevels * 32) != hipSuccess) return 1;
  if (reset) {
    static unsigned long long zeros[orbfe::kMaxLevels][32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(orbfe::g_qtStamps), zeros, sizeof zeros) != hipSuccess) return 1;
  }
  return 0;
}
#endif
