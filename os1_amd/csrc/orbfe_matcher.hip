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
#include "orbfe_matcher_internal.h"

namespace orbfe {
// orbfe_frame.hip: the host-array searches run on a transient device-resident frame + GPU-side bookkeeping
bool match_host_resolve();
int sbp_via_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                  const float* scale_factors, int nlevels, const uint8_t* kp_occupied, const float* mp_proj_xy,
                  const int32_t* mp_level, const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp,
                  float th, float nnratio, int32_t* kp_assigned, int* nmatches);
int sbp_uv_via_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                     const float* scale_factors, int nlevels, const uint8_t* kp_occupied, const float* src_uv,
                     const int32_t* src_level, const float* src_angle, const uint8_t* src_flags, const uint8_t* src_valid,
                     const uint8_t* src_desc, int n_src, float th, int max_dist, int skip_any_occupied, int check_orientation,
                     int32_t* kp_assigned, int* nmatches);
int projected_via_frame(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4], int n_src,
                        const float* src_uv, const float* src_radius, const int32_t* src_level, const uint8_t* src_valid,
                        const uint8_t* src_desc, const uint8_t* kp_skip, int claim, const float* inv_level_sigma2, int nlevels,
                        double chi2, int max_dist, int32_t* best_idx, int32_t* best_dist, int* nmatches);
}  // namespace orbfe

namespace {

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
int orbfe_matcher_device(const orbfe_matcher* m) { return m ? m->device : -1; }
int orbfe_matcher_upload_async(orbfe_matcher* m, void* dst_device, const void* src_host, size_t bytes) {
  if (!m || !dst_device || !src_host) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  if (bytes == 0) return ORBFE_OK;
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, m->stream));
  return ORBFE_OK;
}
int orbfe_matcher_synchronize(orbfe_matcher* m) {
  if (!m) { set_err("matcher is NULL"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return ORBFE_OK;
}

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
  if (!orbfe::match_host_resolve())
    return orbfe::sbp_via_frame(m, kps_un, desc, n, bounds, scale_factors, nlevels, kp_occupied, mp_proj_xy, mp_level, mp_viewcos,
                                mp_flags, mp_desc, n_mp, th, nnratio, kp_assigned, nmatches);
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
  if (!orbfe::match_host_resolve())
    return orbfe::sbp_uv_via_frame(m, kps_un, desc, n, bounds, scale_factors, nlevels, kp_occupied, src_uv, src_level, src_angle,
                                   src_flags, src_valid, src_desc, n_src, th, max_dist, skip_any_occupied, check_orientation,
                                   kp_assigned, nmatches);
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
  if (!orbfe::match_host_resolve())
    return orbfe::projected_via_frame(m, kps_un, desc, n, bounds, n_src, src_uv, src_radius, src_level, src_valid, src_desc, kp_skip,
                                      claim, inv_level_sigma2, nlevels, chi2, max_dist, best_idx, best_dist, nmatches);
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
