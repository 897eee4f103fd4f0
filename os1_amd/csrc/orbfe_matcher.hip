// orbfe_matcher.hip -- windowed 256-bit Hamming search on the GPU + the reference's sequential
// match bookkeeping on the host.  C ABI: include/orbfe.h (matcher section).
//
// Behaviour contract (reference, paths relative to its src/):
//   Frame grid + window query      Frame.cc:98-99,114-129,209-274  (64x48 grid, FRAME_GRID_* Frame.h:36-37)
//   DescriptorDistance             ORBmatcher.cc:1605-1621
//   SearchForInitialization        ORBmatcher.cc:400-515
//   SearchByProjection (MapPoints) ORBmatcher.cc:45-132
//   SearchByProjection (Frame/KF)  ORBmatcher.cc:1292-1552 (from the projection onwards)
//   ComputeThreeMaxima             ORBmatcher.cc:1554-1595
//
// Split of work: every search is "for each query, scan the grid window, compare
// descriptors, keep best/second-best" wrapped in bookkeeping that later iterations read
// (vMatchedDistance / vnMatches21, F.mvpMapPoints occupancy).  The data-parallel part -- window
// test, level filter and Hamming distance for every (query, candidate) pair, emitted in exactly
// the order Frame::GetFeaturesInArea would return them -- is one kernel (one wave per query,
// ballot-ordered compaction).  The order-dependent bookkeeping is replayed on the host over those
// short candidate lists, statement for statement.
#include <hip/hip_runtime.h>

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

namespace {

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
  const int* qpair;       // pair of each query
  const float* qx;
  const float* qy;
  const float* qr;        // < 0 : inactive query
  const int* qminL;
  const int* qmaxL;
  const uint8_t* qdesc;   // [nq][32]
  int nq;
  // outputs
  uint32_t* qcount;       // [nq]
  uint32_t* qoff;         // [nq] offset into pool
  uint32_t* pool;         // entries: idx | dist << 16, reference candidate order per query
  uint32_t poolCap;
  uint32_t* total;        // [1] pool entries claimed (may exceed poolCap: host retries with a larger pool)
};

__device__ __forceinline__ int hamming256(const uint32_t* __restrict__ a, const uint32_t q[8]) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a[i] ^ q[i]);
  return d;
}

// LPQ lanes per query (64/LPQ queries per wave), one LANE per grid column of the window.  Window
// semantics: Frame::GetFeaturesInArea, Frame.cc:209-262: columns ix ascending, rows iy ascending inside a
// column, insertion order inside a cell -- i.e. for column ix the contiguous run
// [cellStart[ix*48+cy0], cellStart[ix*48+cy1+1]) of the cell-sorted keypoint table.  Lane l of a query's
// group walks the run of column cx0+l (a handful of entries), a prefix sum over the group's hit counts
// gives every lane its output offset, so the candidate list comes out in exactly the reference order in
// a single pass with all columns in flight at once.  The host picks LPQ >= the widest window in columns.
template <int LPQ>
__global__ __launch_bounds__(64) void k_window_match(MatchParams M) {
  const int lane = threadIdx.x;
  const int sub = lane & (LPQ - 1);
  const int q = blockIdx.x * (64 / LPQ) + lane / LPQ;
  const bool live = q < M.nq;
  float r = -1.f, x = 0.f, y = 0.f;
  int minL = 0, maxL = -1;
  PairInfo pi = M.pairs[0];
  if (live) {
    r = M.qr[q]; x = M.qx[q]; y = M.qy[q]; minL = M.qminL[q]; maxL = M.qmaxL[q];
    pi = M.pairs[M.qpair[q]];
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
    return fabsf(dx) < r && fabsf(dy) < r;
  };
  // pass 1: hits per column, prefix sum inside the query's lane group
  int hits = 0;
  for (int e = b; e < e1; e++) hits += inWindow(e) ? 1 : 0;
  int incl = hits;
#pragma unroll
  for (int o = 1; o < LPQ; o <<= 1) {
    const int t = __shfl_up(incl, o, LPQ);
    if (sub >= o) incl += t;
  }
  const uint32_t count = (uint32_t)__shfl(incl, LPQ - 1, LPQ);
  uint32_t off = 0;
  if (live && sub == 0) {
    if (count) off = atomicAdd(M.total, count);
    M.qcount[q] = count;
    M.qoff[q] = off;
  }
  off = __shfl(off, 0, LPQ);
  if (hits == 0) return;
  uint32_t qd[8];
  const uint32_t* qp = reinterpret_cast<const uint32_t*>(M.qdesc + (size_t)q * 32);
#pragma unroll
  for (int i = 0; i < 8; i++) qd[i] = qp[i];
  // pass 2: distances, written at the lane's offset in column order
  uint32_t pos = off + (uint32_t)(incl - hits);
  for (int e = b; e < e1; e++) {
    if (!inWindow(e)) continue;
    const int idx = sidx[e];
    const int d = hamming256(reinterpret_cast<const uint32_t*>(tdesc + (size_t)e * 32), qd);  // descriptors are grid-sorted
    if (pos < M.poolCap) M.pool[pos] = (uint32_t)idx | ((uint32_t)d << 16);
    pos++;
  }
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:258-286): one wave per MapPoint.  Lane i owns row i of
// the N x N distance matrix (rows beyond 64 in further passes); the row median (element of rank (N-1)/2) is the
// smallest v with #{d_ij <= v} >= rank+1, found by bisection over the 257 possible distances with the row's
// distances recomputed from LDS each step; the best (median, index) pair is a wave min-reduction.
__global__ __launch_bounds__(64) void k_distinctive(const int* __restrict__ offsets, const uint8_t* __restrict__ descs,
                                                    int* __restrict__ best, int maxN) {
  extern __shared__ __align__(16) uint32_t dd[];   // [N][8]
  const int p = blockIdx.x, lane = threadIdx.x;
  const int o0 = offsets[p], N = offsets[p + 1] - o0;
  if (N <= 0) {
    if (lane == 0) best[p] = -1;
    return;
  }
  const uint32_t* src = reinterpret_cast<const uint32_t*>(descs + (size_t)o0 * 32);
  for (int i = lane; i < N * 8; i += 64) dd[i] = src[i];
  __syncthreads();
  const int rank = (N - 1) >> 1;   // (size_t)(0.5 * (N - 1))
  int bestMed = 0x7fffffff, bestIdx = 0x7fffffff;
  for (int i = lane; i < N; i += 64) {
    uint32_t qi[8];
#pragma unroll
    for (int w = 0; w < 8; w++) qi[w] = dd[i * 8 + w];
    int lo = 0, hi = 256;   // median in [lo, hi]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
      for (int j = 0; j < N; j++) {
        int d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d += __popc(qi[w] ^ dd[j * 8 + w]);
        cnt += d <= mid;
      }
      if (cnt >= rank + 1) hi = mid; else lo = mid + 1;
    }
    if (lo < bestMed) { bestMed = lo; bestIdx = i; }   // ascending i per lane: first minimum kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int om = __shfl_xor(bestMed, o, 64), oi = __shfl_xor(bestIdx, o, 64);
    if (om < bestMed || (om == bestMed && oi < bestIdx)) { bestMed = om; bestIdx = oi; }
  }
  if (lane == 0) best[p] = bestIdx;
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
    HIP_TRY(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
    n = count;
    return ORBFE_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

void computeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

inline int rotBin(float a1, float a2) {  // ORBmatcher.cc:470-475 (factor = 1/HISTO_LENGTH quirk kept)
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

}  // namespace

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
  ~orbfe_matcher() {
    (void)hipSetDevice(device);
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
    // ORBFE_MATCH_ZEROCOPY=1: the kernel reads the pinned host arena directly over PCIe instead of a DMA upload
    static const bool zeroCopy = getenv("ORBFE_MATCH_ZEROCOPY") && atoi(getenv("ORBFE_MATCH_ZEROCOPY")) != 0;
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
        const unsigned nblk = (unsigned)((nq + (64 / lpq) - 1) / (64 / lpq));
        if (lpq == 8) hipLaunchKernelGGL(k_window_match<8>, dim3(nblk), dim3(64), 0, stream, M);
        else if (lpq == 16) hipLaunchKernelGGL(k_window_match<16>, dim3(nblk), dim3(64), 0, stream, M);
        else if (lpq == 32) hipLaunchKernelGGL(k_window_match<32>, dim3(nblk), dim3(64), 0, stream, M);
        else hipLaunchKernelGGL(k_window_match<64>, dim3(nblk), dim3(64), 0, stream, M);
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

// sequential bookkeeping of SearchForInitialization, ORBmatcher.cc:402-512, over one job's candidate lists
static int resolveSearchForInitialization(const orbfe_matcher* m, int q0, const OrbfeKeyPoint* kps1, int n1,
                                          const OrbfeKeyPoint* kps2, int n2, float* prev_xy, int32_t* matches12,
                                          float nnratio, int check_orientation) {
  const uint32_t* pool = m->h_pool.p;
  int nm = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
  for (int i1 = 0; i1 < n1; i1++) {
    if (kps1[i1].octave > 0) continue;
    const uint32_t cnt = m->qcount[q0 + i1];
    if (cnt == 0) continue;
    const uint32_t* cl = pool + m->qoff[q0 + i1];
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (uint32_t c = 0; c < cnt; c++) {
      const int i2 = (int)(cl[c] & 0xffff), dist = (int)(cl[c] >> 16);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nm--; }
        matches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nm++;
        if (check_orientation) rotHist[rotBin(kps1[i1].angle, kps2[bestIdx2].angle)].push_back(i1);
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    computeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nm--; }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) {
      prev_xy[2 * i1] = kps2[matches12[i1]].x;
      prev_xy[2 * i1 + 1] = kps2[matches12[i1]].y;
    }
  return nm;
}

extern "C" {

int orbfe_hamming(const uint8_t a[32], const uint8_t b[32]) {
  int d = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t x, y;
    memcpy(&x, a + 8 * i, 8);
    memcpy(&y, b + 8 * i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

int orbfe_window_candidates(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n,
                            const float bounds[4], int nq, const float* qx, const float* qy, const float* qr,
                            const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* qdesc,
                            uint32_t* counts, uint32_t* offsets, uint32_t* pool, size_t pool_cap, size_t* pool_used) {
  if (!m || !bounds || n < 0 || n > 65535 || nq < 0 || !pool_used || (n && (!kps_un || !desc)) ||
      (nq && (!qx || !qy || !qr || !qmin_level || !qmax_level || !qdesc || !counts || !offsets))) {
    set_err("bad argument (note: at most 65535 keypoints per frame)");
    return ORBFE_ERR_INVALID;
  }
  *pool_used = 0;
  if (nq == 0) return ORBFE_OK;
  int rc = m->candidates(kps_un, desc, n, bounds, qx, qy, qr, qmin_level, qmax_level, qdesc, nq);
  if (rc) return rc;
  // repack per query so the caller's pool is dense and in query order
  size_t used = 0;
  for (int q = 0; q < nq; q++) {
    counts[q] = m->qcount[q];
    offsets[q] = (uint32_t)used;
    used += m->qcount[q];
  }
  *pool_used = used;
  if (used > pool_cap || (used && !pool)) {
    set_err("candidate pool too small: %zu entries needed, %zu given", used, pool_cap);
    return ORBFE_ERR_OVERFLOW;
  }
  for (int q = 0; q < nq; q++)
    if (counts[q]) memcpy(pool + offsets[q], m->h_pool.p + m->qoff[q], sizeof(uint32_t) * counts[q]);
  return ORBFE_OK;
}

int orbfe_distinctive_descriptors(orbfe_matcher* m, int n_mp, const int32_t* offsets, const uint8_t* descs,
                                  int32_t* best_idx) {
  if (!m || n_mp < 0 || (n_mp && (!offsets || !best_idx))) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  if (n_mp == 0) return ORBFE_OK;
  int maxN = 0;
  for (int p = 0; p < n_mp; p++) {
    const int N = offsets[p + 1] - offsets[p];
    if (N < 0) { set_err("offsets must be non-decreasing"); return ORBFE_ERR_INVALID; }
    maxN = std::max(maxN, N);
  }
  const size_t total = (size_t)offsets[n_mp] - (size_t)offsets[0];
  if (total && !descs) { set_err("descs is NULL"); return ORBFE_ERR_INVALID; }
  if ((size_t)maxN * 32 > 150 * 1024) { set_err("a MapPoint with %d observations exceeds the LDS budget", maxN); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(m->device));
  int rc;
  const size_t oOff = 0, oDesc = al(4 * (size_t)(n_mp + 1)), bytes = oDesc + al(32 * total + 32);
  if ((rc = m->h_in.ensure(bytes))) return rc;
  if ((rc = m->d_in.ensure(bytes))) return rc;
  if ((rc = m->d_out.ensure((size_t)n_mp + 64))) return rc;
  if ((rc = m->h_out.ensure((size_t)n_mp + 64))) return rc;
  int* ho = (int*)(m->h_in.p + oOff);
  for (int p = 0; p <= n_mp; p++) ho[p] = offsets[p] - offsets[0];
  if (total) memcpy(m->h_in.p + oDesc, descs + (size_t)offsets[0] * 32, 32 * total);
  HIP_TRY(hipMemcpyAsync(m->d_in.p, m->h_in.p, bytes, hipMemcpyHostToDevice, m->stream));
  const size_t lds = std::max<size_t>((size_t)maxN * 32, 32);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_distinctive), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_distinctive, dim3(n_mp), dim3(64), lds, m->stream, (const int*)(m->d_in.p + oOff),
                     (const uint8_t*)(m->d_in.p + oDesc), (int*)m->d_out.p, maxN);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(m->h_out.p, m->d_out.p, sizeof(int) * n_mp, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  memcpy(best_idx, m->h_out.p, sizeof(int) * n_mp);
  return ORBFE_OK;
}

int orbfe_undistort_equidistant(float* xy, int n, float fx, float fy, float cx, float cy) {
  if (n < 0 || (n && !xy)) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  // Frame.cc:357-383: f, c as double copies of the float intrinsics; K (Matx33f) * Vec3d in double
  const double f0 = fx, f1 = fy, c0 = cx, c1 = cy;
  for (int i = 0; i < n; i++) {
    const double pi0 = xy[2 * i], pi1 = xy[2 * i + 1];
    const double pw0 = (pi0 - c0) / f0, pw1 = (pi1 - c1) / f1;
    const double theta_d = std::sqrt(pw0 * pw0 + pw1 * pw1);
    const double scale = theta_d > 1e-8 ? std::tan(theta_d) / theta_d : 1.0;
    const double pu0 = pw0 * scale, pu1 = pw1 * scale;
    const double pr0 = (double)fx * pu0 + 0.0 * pu1 + (double)cx * 1.0;
    const double pr1 = 0.0 * pu0 + (double)fy * pu1 + (double)cy * 1.0;
    const double pr2 = 0.0 * pu0 + 0.0 * pu1 + 1.0 * 1.0;
    xy[2 * i] = (float)(pr0 / pr2);
    xy[2 * i + 1] = (float)(pr1 / pr2);
  }
  return ORBFE_OK;
}

// cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) as Frame::UndistortKeyPoints / ComputeImageBounds call it
// (Frame.cc:307, 339): normalise with K, five fixed-point iterations of the inverse Brown / rational distortion (OpenCV's
// default TermCriteria(COUNT, 5, 0.01)), project back with K -- all in double, one final rounding to float.  Host math:
// a few thousand points per frame.
int orbfe_undistort_pinhole(float* xy, int n, float fx, float fy, float cx, float cy, const float* dist, int ndist) {
  if (n < 0 || (n && !xy) || ndist < 0 || ndist > 8 || (ndist && !dist)) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // k1 k2 p1 p2 k3 k4 k5 k6
  for (int i = 0; i < ndist; i++) k[i] = dist[i];
  const double dfx = fx, dfy = fy, dcx = cx, dcy = cy, ifx = 1. / dfx, ify = 1. / dfy;
  for (int i = 0; i < n; i++) {
    const double u = xy[2 * i], v = xy[2 * i + 1];
    double x = (u - dcx) * ifx, y = (v - dcy) * ify;
    const double x0 = x, y0 = y;
    for (int it = 0; it < 5; it++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {   // the model folds back on itself here: OpenCV returns the undistorted guess
        x = (u - dcx) * ifx;
        y = (v - dcy) * ify;
        break;
      }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = dfx * x + 0 * y + dcx, yy = 0 * x + dfy * y + dcy, ww = 1. / (0 * x + 0 * y + 1);
    xy[2 * i] = (float)(xx * ww);
    xy[2 * i + 1] = (float)(yy * ww);
  }
  return ORBFE_OK;
}

// Frame::ComputeImageBounds (Frame.cc:322-353): the four image corners through the camera model's undistortion.
int orbfe_compute_image_bounds(int cols, int rows, int camera_mode, float fx, float fy, float cx, float cy, const float* dist,
                               int ndist, float bounds[4]) {
  if (!bounds || cols <= 0 || rows <= 0 || ndist < 0 || ndist > 8 || (ndist && !dist)) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  if (camera_mode != 0 || (ndist > 0 && dist[0] != 0.0)) {
    float mat[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
    const int rc = camera_mode ? orbfe_undistort_equidistant(mat, 4, fx, fy, cx, cy)
                               : orbfe_undistort_pinhole(mat, 4, fx, fy, cx, cy, dist, ndist);
    if (rc) return rc;
    bounds[0] = std::min(mat[0], mat[4]);
    bounds[1] = std::max(mat[2], mat[6]);
    bounds[2] = std::min(mat[1], mat[3]);
    bounds[3] = std::max(mat[5], mat[7]);
  } else {
    bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
  }
  return ORBFE_OK;
}

int orbfe_matcher_create(int device_id, orbfe_matcher** out) {
  if (!out) { set_err("out is NULL"); return ORBFE_ERR_INVALID; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    set_err("no usable HIP device (count=%d, requested %d): this library has no CPU fallback", ndev, device_id);
    return ORBFE_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device_id));
  orbfe_matcher* m = new orbfe_matcher();
  m->device = device_id;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete m;
    return ORBFE_ERR_HIP;
  }
  int nthreads = (int)std::thread::hardware_concurrency();
  if (nthreads > 8) nthreads = 8;
  if (const char* ev = getenv("ORBFE_HOST_THREADS")) nthreads = atoi(ev);
  if (nthreads < 1) nthreads = 1;
  m->pool.reset(new orbfe::HostPool(nthreads));
  *out = m;
  return ORBFE_OK;
}

void orbfe_matcher_destroy(orbfe_matcher* m) { delete m; }

extern "C++" {
namespace orbfe {
// for orbfe_bow.hip, which keeps its device scratch on the matcher handle
int matcher_device(const orbfe_matcher* m) { return m->device; }
hipStream_t matcher_stream(const orbfe_matcher* m) { return m->stream; }
std::shared_ptr<void>& matcher_bow_slot(orbfe_matcher* m) { return m->bow; }
}  // namespace orbfe
}  // extern "C++"

int orbfe_debug_matcher_ms(const orbfe_matcher* m, double out[3]) {
  if (!m || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < 3; i++) out[i] = m->stageMs[i];
  return ORBFE_OK;
}

int orbfe_debug_features_in_area(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, int n, const float bounds[4], float x,
                                 float y, float r, int min_level, int max_level, int32_t* out, int cap, int* n_out) {
  if (!m || !kps_un || !bounds || !n_out || n < 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  std::vector<uint8_t> zdesc((size_t)std::max(n, 1) * 32, 0), qd(32, 0);
  int rc = m->candidates(kps_un, zdesc.data(), n, bounds, &x, &y, &r, &min_level, &max_level, qd.data(), 1);
  if (rc) return rc;
  const int c = (int)m->qcount[0];
  *n_out = c;
  for (int i = 0; i < c && i < cap; i++) out[i] = (int)(m->h_pool.p[m->qoff[0] + i] & 0xffff);
  return ORBFE_OK;
}

int orbfe_search_for_initialization_batch(orbfe_matcher* m, int npairs, const OrbfeKeyPoint* const* kps1,
                                          const uint8_t* const* desc1, const int* n1,
                                          const OrbfeKeyPoint* const* kps2, const uint8_t* const* desc2,
                                          const int* n2, const float bounds[4], float* const* prev_xy,
                                          int32_t* const* matches12, int window_size, float nnratio,
                                          int check_orientation, int* nmatches) {
  if (!m || !nmatches || npairs < 0 || !bounds || (npairs && (!kps1 || !desc1 || !n1 || !kps2 || !desc2 || !n2 ||
                                                             !prev_xy || !matches12))) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  size_t nqTot = 0;
  for (int p = 0; p < npairs; p++) {
    nmatches[p] = 0;
    if (n1[p] < 0 || n2[p] < 0 || n2[p] > 65535 || (n1[p] && (!kps1[p] || !desc1[p] || !prev_xy[p] || !matches12[p])) ||
        (n2[p] && (!kps2[p] || !desc2[p]))) {
      set_err("bad argument in pair %d (note: at most 65535 keypoints per frame)", p);
      return ORBFE_ERR_INVALID;
    }
    for (int i = 0; i < n1[p]; i++) matches12[p][i] = -1;
    nqTot += n1[p];
  }
  if (nqTot == 0) return ORBFE_OK;
  // queries: level-0 keypoints of F1, window centred on vbPrevMatched (ORBmatcher.cc:413-420)
  std::vector<float> qx(nqTot), qy(nqTot), qr(nqTot);
  std::vector<int> qa(nqTot), qb(nqTot);
  std::vector<orbfe_matcher::Job> jobs(npairs);
  size_t q0 = 0;
  for (int p = 0; p < npairs; p++) {
    for (int i = 0; i < n1[p]; i++) {
      const int level1 = kps1[p][i].octave;
      qx[q0 + i] = prev_xy[p][2 * i];
      qy[q0 + i] = prev_xy[p][2 * i + 1];
      qr[q0 + i] = level1 > 0 ? -1.f : (float)window_size;
      qa[q0 + i] = level1;
      qb[q0 + i] = level1;
    }
    jobs[p] = orbfe_matcher::Job{kps2[p], desc2[p], n2[p], bounds, qx.data() + q0, qy.data() + q0, qr.data() + q0,
                                 qa.data() + q0, qb.data() + q0, desc1[p], n1[p]};
    q0 += n1[p];
  }
  int rc = m->candidates(jobs.data(), npairs);
  if (rc) return rc;
  const double tR = orbfe_matcher::nowMs();
  m->pool->parallelFor(npairs, [&](int p, int) {
    nmatches[p] = resolveSearchForInitialization(m, m->jobQ0[p], kps1[p], n1[p], kps2[p], n2[p], prev_xy[p],
                                                 matches12[p], nnratio, check_orientation);
  });
  m->stageMs[2] = orbfe_matcher::nowMs() - tR;
  return ORBFE_OK;
}

int orbfe_search_for_initialization(orbfe_matcher* m, const OrbfeKeyPoint* kps1, const uint8_t* desc1, int n1,
                                    const OrbfeKeyPoint* kps2, const uint8_t* desc2, int n2, const float bounds[4],
                                    float* prev_xy, int32_t* matches12, int window_size, float nnratio,
                                    int check_orientation, int* nmatches) {
  if (!nmatches) { set_err("nmatches is NULL"); return ORBFE_ERR_INVALID; }
  return orbfe_search_for_initialization_batch(m, 1, &kps1, &desc1, &n1, &kps2, &desc2, &n2, bounds, &prev_xy,
                                               &matches12, window_size, nnratio, check_orientation, nmatches);
}

int orbfe_search_by_projection(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n,
                               const float bounds[4], const float* scale_factors, int nlevels,
                               const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                               const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp,
                               float th, float nnratio, int32_t* kp_assigned, int* nmatches) {
  if (!m || !nmatches || n < 0 || n > 65535 || n_mp < 0 || !bounds || !scale_factors ||
      (n && (!kps_un || !desc || !kp_occupied || !kp_assigned)) ||
      (n_mp && (!mp_proj_xy || !mp_level || !mp_viewcos || !mp_flags || !mp_desc))) {
    set_err("bad argument (note: at most 65535 keypoints per frame)");
    return ORBFE_ERR_INVALID;
  }
  *nmatches = 0;
  for (int i = 0; i < n; i++) kp_assigned[i] = -1;
  if (n_mp == 0 || n == 0) return ORBFE_OK;
  const bool bFactor = th != 1.0;
  std::vector<float> qr(n_mp);
  std::vector<float> qx(n_mp), qy(n_mp);
  std::vector<int> qa(n_mp), qb(n_mp);
  for (int i = 0; i < n_mp; i++) {
    const uint8_t fl = mp_flags[i];
    qx[i] = mp_proj_xy[2 * i];
    qy[i] = mp_proj_xy[2 * i + 1];
    const int lvl = mp_level[i];
    qa[i] = lvl - 1;
    qb[i] = lvl;
    if (!(fl & ORBFE_MP_IN_VIEW) || (fl & ORBFE_MP_BAD)) { qr[i] = -1.f; continue; }
    if (lvl < 0 || lvl >= nlevels) { set_err("MapPoint %d: level %d out of range", i, lvl); return ORBFE_ERR_INVALID; }
    float r = (fl & ORBFE_MP_CANDIDATO) ? 4.0 : (mp_viewcos[i] > 0.998 ? 2.5 : 4.0);  // ORBmatcher.cc:63-65,126-132
    if (bFactor) r *= th;
    qr[i] = r * scale_factors[lvl];
  }
  int rc = m->candidates(kps_un, desc, n, bounds, qx.data(), qy.data(), qr.data(), qa.data(), qb.data(), mp_desc, n_mp);
  if (rc) return rc;
  const uint32_t* pool = m->h_pool.p;
  std::vector<uint8_t> occ(kp_occupied, kp_occupied + n);
  int nm = 0;
  for (int iMP = 0; iMP < n_mp; iMP++) {
    if (qr[iMP] < 0.f) continue;
    const uint32_t cnt = m->qcount[iMP];
    if (cnt == 0) continue;
    const uint32_t* cl = pool + m->qoff[iMP];
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (uint32_t c = 0; c < cnt; c++) {
      const int idx = (int)(cl[c] & 0xffff), dist = (int)(cl[c] >> 16);
      if (occ[idx]) continue;
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
        bestLevel = kps_un[idx].octave; bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = kps_un[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      kp_assigned[bestIdx] = iMP;
      occ[bestIdx] = (mp_flags[iMP] & ORBFE_MP_OBSERVED) ? 1 : 0;
      nm++;
    }
  }
  *nmatches = nm;
  return ORBFE_OK;
}

int orbfe_search_by_projection_uv(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n,
                                  const float bounds[4], const float* scale_factors, int nlevels,
                                  const uint8_t* kp_occupied, const float* src_uv, const int32_t* src_level,
                                  const float* src_angle, const uint8_t* src_flags, const uint8_t* src_valid,
                                  const uint8_t* src_desc, int n_src, float th, int max_dist, int skip_any_occupied,
                                  int check_orientation, int32_t* kp_assigned, int* nmatches) {
  if (!m || !nmatches || n < 0 || n > 65535 || n_src < 0 || !bounds || !scale_factors ||
      (n && (!kps_un || !desc || !kp_occupied || !kp_assigned)) ||
      (n_src && (!src_uv || !src_level || !src_angle || !src_flags || !src_valid || !src_desc))) {
    set_err("bad argument (note: at most 65535 keypoints per frame)");
    return ORBFE_ERR_INVALID;
  }
  *nmatches = 0;
  for (int i = 0; i < n; i++) kp_assigned[i] = -1;
  if (n_src == 0 || n == 0) return ORBFE_OK;
  std::vector<float> qx(n_src), qy(n_src), qr(n_src);
  std::vector<int> qa(n_src), qb(n_src);
  for (int i = 0; i < n_src; i++) {
    qx[i] = src_uv[2 * i];
    qy[i] = src_uv[2 * i + 1];
    const int lvl = src_level[i];
    qa[i] = lvl - 1;
    qb[i] = lvl + 1;
    if (!src_valid[i]) { qr[i] = -1.f; continue; }
    if (lvl < 0 || lvl >= nlevels) { set_err("source %d: level %d out of range", i, lvl); return ORBFE_ERR_INVALID; }
    qr[i] = th * scale_factors[lvl];
  }
  int rc = m->candidates(kps_un, desc, n, bounds, qx.data(), qy.data(), qr.data(), qa.data(), qb.data(), src_desc, n_src);
  if (rc) return rc;
  const uint32_t* pool = m->h_pool.p;
  std::vector<uint8_t> occ(kp_occupied, kp_occupied + n);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nm = 0;
  for (int i = 0; i < n_src; i++) {
    if (qr[i] < 0.f) continue;
    const uint32_t cnt = m->qcount[i];
    if (cnt == 0) continue;
    const uint32_t* cl = pool + m->qoff[i];
    int bestDist = 256, bestIdx2 = -1;
    for (uint32_t c = 0; c < cnt; c++) {
      const int i2 = (int)(cl[c] & 0xffff), dist = (int)(cl[c] >> 16);
      if (occ[i2]) continue;
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= max_dist) {
      kp_assigned[bestIdx2] = i;
      occ[bestIdx2] = skip_any_occupied ? 1 : ((src_flags[i] & ORBFE_MP_OBSERVED) ? 1 : 0);
      nm++;
      if (check_orientation) rotHist[rotBin(src_angle[i], kps_un[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    computeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (int idx : rotHist[i]) {
          kp_assigned[idx] = -2;
          nm--;
        }
      }
    }
  }
  *nmatches = nm;
  return ORBFE_OK;
}

// The projected best-match loop of SearchByProjection(KeyFrame*, Scw, ...) (ORBmatcher.cc:357-392), Fuse x2
// (:872-936, :1014-1050) and SearchBySim3 (:1066-1290): see include/orbfe.h.
int orbfe_search_projected(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                           int n_src, const float* src_uv, const float* src_radius, const int32_t* src_level,
                           const uint8_t* src_valid, const uint8_t* src_desc, const uint8_t* kp_skip, int claim,
                           const float* inv_level_sigma2, int nlevels, double chi2, int max_dist, int32_t* best_idx,
                           int32_t* best_dist, int* nmatches) {
  if (!m || !bounds || !nmatches || n < 0 || n > 65535 || n_src < 0 || (n && (!kps_un || !desc)) ||
      (n_src && (!src_uv || !src_radius || !src_level || !src_valid || !src_desc || !best_idx))) {
    set_err("bad argument (note: at most 65535 keypoints per frame)");
    return ORBFE_ERR_INVALID;
  }
  *nmatches = 0;
  for (int i = 0; i < n_src; i++) {
    best_idx[i] = -1;
    if (best_dist) best_dist[i] = -1;
  }
  if (n_src == 0 || n == 0) return ORBFE_OK;
  if (inv_level_sigma2)
    for (int i = 0; i < n; i++)
      if (kps_un[i].octave < 0 || kps_un[i].octave >= nlevels) { set_err("keypoint octave out of range"); return ORBFE_ERR_INVALID; }
  std::vector<float> qx(n_src), qy(n_src), qr(n_src);
  std::vector<int> qa(n_src), qb(n_src);
  for (int i = 0; i < n_src; i++) {
    qx[i] = src_uv[2 * i];
    qy[i] = src_uv[2 * i + 1];
    qa[i] = src_level[i] - 1;                      // kpLevel < nPredictedLevel-1 || kpLevel > nPredictedLevel => skip
    qb[i] = src_level[i];
    qr[i] = (src_valid[i] && src_level[i] >= 0) ? src_radius[i] : -1.f;   // no octave lies in [level-1, level] for level < 0
  }
  int rc = m->candidates(kps_un, desc, n, bounds, qx.data(), qy.data(), qr.data(), qa.data(), qb.data(), src_desc, n_src);
  if (rc) return rc;
  const uint32_t* pool = m->h_pool.p;
  std::vector<uint8_t> occ(n, 0);
  if (kp_skip) occ.assign(kp_skip, kp_skip + n);
  const bool useOcc = kp_skip || claim;
  int nm = 0;
  for (int i = 0; i < n_src; i++) {
    if (qr[i] < 0.f) continue;
    const uint32_t cnt = m->qcount[i];
    const uint32_t* cl = pool + m->qoff[i];
    const float u = qx[i], v = qy[i];
    int bestDist = INT_MAX, bestIdx = -1;
    for (uint32_t c = 0; c < cnt; c++) {
      const int idx = (int)(cl[c] & 0xffff), dist = (int)(cl[c] >> 16);
      if (useOcc && occ[idx]) continue;
      if (inv_level_sigma2) {   // ORBmatcher.cc:896-903
        const float ex = u - kps_un[idx].x;
        const float ey = v - kps_un[idx].y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_level_sigma2[kps_un[idx].octave] > chi2) continue;
      }
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= max_dist) {
      best_idx[i] = bestIdx;
      if (best_dist) best_dist[i] = bestDist;
      if (claim) occ[bestIdx] = 1;
      nm++;
    }
  }
  *nmatches = nm;
  return ORBFE_OK;
}

}  // extern "C"
