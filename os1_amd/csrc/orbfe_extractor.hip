// orbfe_extractor.hip -- host engine + C ABI of the extractor (include/orbfe.h).
//
// Behaviour contract: ORB_SLAM2::ORBextractor of the reference (src/ORBextractor.cc:442-502 ctor,
// :907-969 operator(), :971-996 ComputePyramid, :797-895 ComputeKeyPointsOctTree).  The pixel work
// runs in the kernels of orbfe_kernels.hip, the order-defining quadtree in orbfe_quadtree.hip (one
// stream submission per batch, submitGpuQt); a second, host-side quadtree (quadtree.h, run()) serves
// geometries the kernel does not hold (more than 4 roots or 2044 features per level) and
// ORBFE_HOST_QUADTREE=1.  There is no CPU fallback for the kernels.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"
#include "glibc_sincosf.h"
#include "host_pool.h"
#include "orbfe_internal.h"
#include "quadtree.h"

namespace orbfe {
int launch_pyramid(const PyramidParams& P, int nframes, hipStream_t st, const ConeParams* cone);
void launch_ingest(const uint8_t* src, long long sstride, uint8_t* dst, long long dpitch, int rowBytes, int rows, hipStream_t st);
void launch_to_gray(const uint8_t* const* raw, long long rawStride, const uint8_t* const* gray, long long grayPitch, int rows,
                    int cols, int channels, const int coef[3], int shift, bool aligned, int nframes, hipStream_t st);
void launch_fast(const PyramidParams& P, int nframes, hipStream_t st);
uint32_t fast_task_geo(int fastW, int hCell);
void launch_compact(const PyramidParams& P, int nframes, hipStream_t st);
void launch_compact_local(const PyramidParams& P, int nframes, hipStream_t st, int level0, int level1);
void launch_describe(const PyramidParams& P, const SelKp* sel, int nsel, float* angle, uint8_t* desc,
                     hipStream_t st);
void launch_sincos(const float* deg, int n, float* c, float* s, hipStream_t st);
void launch_spin(unsigned long long ticks, hipStream_t st);
void launch_describe_slots(const PyramidParams& P, const SelKp* sel, int nslots, float* angle, uint8_t* desc,
                           const uint32_t* selCount, int selPerFrame, const int* selOff, hipStream_t st, bool fourWaves, float* angle2 = nullptr,
                           uint8_t* desc2 = nullptr);
int launch_quadtree(const QtParams& Q, int nframes, hipStream_t st, int ldsBudget);
// orbfe_bow.hip
int bow_launch_descend(orbfe_vocabulary* v, const uint8_t* d_desc, int n, int levelsup, uint2* d_out, hipStream_t st);
int bow_assemble(orbfe_vocabulary* v, const uint2* ln, int n, uint32_t* bow_ids, double* bow_values, int* n_words,
                 uint32_t* fv_nodes, uint32_t* fv_offsets, uint32_t* fv_features, int* n_fv_nodes,
                 uint32_t* word_of_feature, uint32_t* node_of_feature);
int bow_device(const orbfe_vocabulary* v);
void launch_sfi(const SfiParams& S, int nframes, hipStream_t st);
void launch_sfi_carry(const SfiParams& S, int lastFrame, SelKp* cSel, float* cAngle, uint8_t* cDesc, uint32_t* cCount,
                      hipStream_t st);

thread_local std::string g_err;
void set_err(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
}  // namespace orbfe

using namespace orbfe;

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return ORBFE_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)

namespace {
inline int cv_round_f(float v) { return (int)lrintf(v); }
inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

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
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
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
    HIP_TRY(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
    n = count;
    return ORBFE_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};
}  // namespace

// Predecessor hand-over between consecutive batches of one stream (the batches alternate between extractor
// handles): level-0 data of a batch's last frame, double-buffered, with an event per buffer.
struct orbfe_sfi_chain {
  int device = 0, n0cap = 0;
  DevBuf<SelKp> sel[2];
  DevBuf<float> angle[2];
  DevBuf<uint8_t> desc[2];
  DevBuf<uint32_t> count;      // [0],[1]: level-0 count of the buffers; [2]: constant 0xffffffff ("none")
  hipEvent_t ready[2] = {};
  long long seq = 0;           // batches submitted so far
  bool isolated = false;       // orbfe_sfi_chain_set_isolated: no batch has a predecessor (frame 0 of every batch reports no match)
  // host-quadtree route (geometries outside the GPU quadtree's limits): the predecessor frame lives on the host and the
  // searches go through an ordinary matcher handle
  orbfe_matcher* hostMatcher = nullptr;
  std::vector<OrbfeKeyPoint> hostPrevKps;
  std::vector<uint8_t> hostPrevDesc;
  bool hostPrevValid = false;
  ~orbfe_sfi_chain() {
    (void)hipSetDevice(device);
    if (hostMatcher) orbfe_matcher_destroy(hostMatcher);
    for (int i = 0; i < 2; i++) { sel[i].release(); angle[i].release(); desc[i].release(); if (ready[i]) (void)hipEventDestroy(ready[i]); }
    count.release();
  }
};

// One upload lane (HIP stream) per device, shared by every extractor handle: host frames of successive
// batches cross PCIe one after the other at full link rate while the previous batch computes.  With each handle
// copying on its own stream, two handles settle into lock-step (both copy at half rate, then both compute).
struct UploadLane {
  std::mutex mu;
  hipStream_t stream = nullptr;
};
static UploadLane* upload_lane(int device) {
  static std::mutex mu;
  static UploadLane* lanes[64] = {};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!lanes[device]) {
    UploadLane* l = new UploadLane;
    if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) { delete l; return nullptr; }
    lanes[device] = l;   // lives for the process
  }
  return lanes[device];
}

struct orbfe_extractor {
  int nfeatures, nlevels, iniTh, minTh, device;
  double scaleFactor;  // the reference keeps the float ctor argument in a double member (ORBextractor.h:313)
  std::vector<float> sf, isf, sigma2, isigma2;
  std::vector<int> nfeat;
  static constexpr int kMaxSub = 4;
  hipStream_t stream = nullptr;          // == streams[0]
  hipStream_t streams[kMaxSub] = {};
  static constexpr int subBatches = 4;   // host-quadtree route: sub-batches in flight (one HIP stream each)
  hipEvent_t evUpload = nullptr;
  bool lastLocalCand = false, submitLocalCand = false;   // candidate lists of the last collected / submitted batch are level-local (k_compact_local)
  bool localLists = true;         // (experiments build, ORBFE_LOCAL_LISTS=0: the packed list of k_compact in latency-bound calls too)
  hipEvent_t evFrame0 = nullptr, evS1[kMaxSub] = {};
  int subSel[kMaxSub] = {};
  size_t candHostCap = 0;
  // GPU quadtree path (default): everything from the frame to descriptors in one stream submission
  bool gpuQuadtree = true;
  // Frame::ComputeBoW fused behind k_describe (orbfe_extractor_set_vocabulary): the descent runs on the descriptors
  // while they are still in HBM; (leaf, node) pairs of the last collected batch in keypoint order, per frame
  orbfe_vocabulary* voc = nullptr;
  int vocLevelsup = 4;
  bool pendingBow = false;
  DevBuf<uint2> d_bow;
  PinBuf<uint2> h_bow;
  std::vector<std::vector<uint2>> bowKp;
  int kpPerFrameCap = 0;     // internal per-frame keypoint capacity of the result arena (set by setGeometry)
  bool geomGpuQtOk = true;   // every level of the current image size has 1..4 quadtree roots (aspect ratio < 4.5)
  QtParams QP{};
  int selOff[kMaxLevels + 1] = {};
  int selPerFrame = 0;
  DevBuf<uint16_t> d_own;   // node id per candidate (quadtree)
  hipEvent_t evQt[2] = {};
  std::vector<int> frameKpBase, frameKpCount;

  int rows = 0, cols = 0, batchCap = 0;
  PyramidParams P{};
  DevBuf<uint8_t> d_tables, d_slab, d_in;
  DevBuf<ConeRange> d_coneTab, d_coneTailTab;
  ConeParams coneTail{};      // (experiments build, ORBFE_TAIL_CONE_BASE / _TILE: levels base + 1 .. top of a BATCH in one cone launch)
  bool tailOk = false;
  int tailBase = 0, tailTile = 48;
  ConeParams cone{};
  bool coneOk = false;
  bool qtJump = true;             // ORBFE_QT_JUMP=0: every pass of the quadtree is an ordinary sweep
  int qtLdsBudget = 120 * 1024;   // LDS bytes a quadtree problem may use for its candidates in small batches
  int pollWaitUs = 0;             // > 0: collect polls the stream and sleeps this long between polls
  bool zeroCopyOut = true;        // one- / two-frame calls: results written to host memory by the kernels (ORBFE_ZERO_COPY=0: one copy command behind the call)
  bool zeroCopyBatches = false;   // ORBFE_ZERO_COPY=2: batches too
  bool ingestKernel = true;       // small batches from page-locked host memory: fetched by a kernel on the compute stream (ORBFE_INGEST_KERNEL=0: copy command)
  int coneTile = 0;        // tile edge on the top level (0 = chosen from the level's size)
  int coneMaxFrames = 2;   // batches up to this size build the pyramid with k_pyramid_cone
  DevBuf<CellInfo> d_cells;
  DevBuf<FastTask> d_tasks;
  DevBuf<uint8_t> d_zeros;
  bool pairCells = false;  // ORBFE_FAST_PAIRS=1: two adjacent cells per wave (7 % fewer vector instructions, but 10 % slower: DESIGN.md s5)
  DevBuf<uint32_t> d_cellCount, d_slots, d_cand;
  DevBuf<const uint8_t*> d_frame0;
  // results leave the GPU in ONE copy: [levelStart][selCount][sel][angle][desc] carved from one arena
  template <class T> struct View { T* p = nullptr; };
  DevBuf<uint8_t> d_outArena;
  PinBuf<uint8_t> h_outArena;
  size_t outArenaBytes = 0;
  View<uint32_t> d_levelStart, d_selCount, h_levelStart, h_selCount;
  View<SelKp> d_sel, h_sel;
  View<float> d_angle, h_angle;
  View<uint8_t> d_desc, h_desc;
  View<int32_t> d_m12, h_m12, d_nm, h_nm;   // GPU SearchForInitialization outputs ([B][n0cap], [B])
  DevBuf<uint16_t> d_sfiOrder;
  DevBuf<int> d_sfiOrderCount;
  DevBuf<uint32_t> d_sfiPool, d_sfiPcount;
  bool pendingMatched = false;
  DevBuf<float> d_f32tmp;
  bool profileKernels = false;  // HIP events around every kernel group (adds ~5-10 us of gaps per event)
  PinBuf<const uint8_t*> h_frame0;
  PinBuf<uint32_t> h_cand;
  long long inPitch = 0;
  int inFormat = ORBFE_INPUT_GRAY8, grayVariant = ORBFE_GRAY_Q15;   // orbfe_extractor_set_input_format
  int gaussVariant = ORBFE_GAUSS_ED;                                // orbfe_extractor_set_blur_variant (ORBFE_GAUSS_VARIANT presets it)
  long long grayPitch = 0;
  DevBuf<uint8_t> d_gray;   // level 0 of colour input
  bool inLinear = false;   // host frames are uploaded with linear copies (inPitch == host stride)
  int lastFrames = 0;
  // deferred route of _submit / _submit_matched when the GPU quadtree cannot take the geometry: the batch is extracted by the
  // blocking host-quadtree path at submit time (and matched through a matcher handle), _collect hands the results out
  int deferredFrames = 0, deferredCap = 0;
  bool deferredMatched = false;
  std::vector<OrbfeKeyPoint> defKps;
  std::vector<uint8_t> defDesc;
  std::vector<int> defN, defNm;
  std::vector<int32_t> defM12;
  bool lastGpuQt = false, lastZeroCopy = false, submitZeroCopy = false;   // route of the last collected / submitted batch
  bool describe4 = true;         // ORBFE_DESCRIBE_WAVES=1: one wave per keypoint in one- / two-frame calls too
  float stageMs[5] = {0, 0, 0, 0, 0};
  hipEvent_t ev[kMaxSub][6] = {};
  double kernMs[5] = {0, 0, 0, 0, 0};
  long long kernBatches = 0, kernFrames = 0;

  struct Meta { int16_t x, y; uint8_t score, level; };
  struct Worker {  // per-thread quadtree scratch
    QuadTree qt;
    std::vector<int16_t> qx, qy;
    std::vector<uint8_t> qs;
    std::vector<int> qsel;
  };
  std::unique_ptr<HostPool> pool;
  int hostThreads = 1;
  std::vector<Worker> workers;
  std::vector<std::vector<Meta>> taskOut;  // [frame*nlevels + level]
  std::vector<Meta> meta;

  ~orbfe_extractor() {
    (void)hipSetDevice(device);
    for (auto& st : streams) if (st) (void)hipStreamSynchronize(st);   // a batch may still be in flight
    d_bow.release(); h_bow.release();
    d_tables.release(); d_coneTab.release(); d_coneTailTab.release(); d_slab.release(); d_in.release(); d_cellCount.release();
    d_sfiOrder.release(); d_sfiOrderCount.release(); d_sfiPool.release(); d_sfiPcount.release();
    d_cells.release(); d_tasks.release(); d_zeros.release(); d_slots.release(); d_cand.release(); d_frame0.release(); d_gray.release(); d_outArena.release(); h_outArena.release();
    d_f32tmp.release();
    h_frame0.release(); h_cand.release();
    for (auto& es : ev) for (auto& e : es) if (e) (void)hipEventDestroy(e);
    for (auto& e : evS1) if (e) (void)hipEventDestroy(e);
    for (auto& e : evQt) if (e) (void)hipEventDestroy(e);
    d_own.release();
    if (evFrame0) (void)hipEventDestroy(evFrame0);
    if (evUpload) (void)hipEventDestroy(evUpload);

    for (auto& st : streams) if (st) (void)hipStreamDestroy(st);
  }

  // Level sizes, FAST cell grids, bilinear tables.  ORBextractor.cc:975-976 (sizes), :807-823 (cells).
  int setGeometry(int r, int c) {
    if (r == rows && c == cols) return ORBFE_OK;
    if (c > 4095 || r > 4095) {
      set_err("image %dx%d exceeds the 4095-pixel coordinate packing limit", c, r);
      return ORBFE_ERR_INVALID;
    }
    bool qtOk = true;
    int kpSum = 8;
    PyramidParams Q{};
    Q.nlevels = nlevels;
    Q.iniTh = iniTh;
    Q.minTh = minTh;
    long long off = 0, slot = 0;
    int cellBase = 0;
    size_t tableBytes = 0;
    for (int l = 0; l < nlevels; l++) {
      LevelGeom& L = Q.lv[l];
      L.w = cv_round_f((float)c * isf[l]);
      L.h = cv_round_f((float)r * isf[l]);
      const float width = (float)((L.w - kEdge + 3) - (kEdge - 3));
      const float height = (float)((L.h - kEdge + 3) - (kEdge - 3));
      L.nCols = (int)(width / 30.f);
      L.nRows = (int)(height / 30.f);
      if (L.nCols < 1 || L.nRows < 1) {
        set_err("image %dx%d too small: level %d (%dx%d) has no FAST cell", c, r, l, L.w, L.h);
        return ORBFE_ERR_TOO_SMALL;
      }
      L.wCell = (int)std::ceil(width / L.nCols);
      L.hCell = (int)std::ceil(height / L.nRows);
      {   // root nodes of DistributeOctTree (ORBextractor.cc:574-578); k_quadtree2 holds up to 4 of them
        const int nIni = (int)roundf(static_cast<float>(L.w - 2 * kBorder) / (L.h - 2 * kBorder));
        if (nIni < 1 || nIni > 4) qtOk = false;
        kpSum += std::max(nfeat[l] + 4, 4 * std::max(nIni, 4));
      }
      L.cellBase = cellBase;
      cellBase += L.nCols * L.nRows;
      L.slotCap = (((L.wCell + 1) / 2) * ((L.hCell + 1) / 2) + 3) & ~3;   // the strict-local-max bound, rounded up so that every cell's slots start 16-byte aligned
      L.slotBase = slot;
      slot += (long long)L.nCols * L.nRows * L.slotCap;
      if (l >= 1) {
        L.pitch = (int)align_up(L.w, 64);
        L.off = off;
        off += align_up((long long)L.pitch * L.h, 256);
        tableBytes += align_up(L.w * 4, 16) + align_up(L.w * 4, 16) + align_up(L.h * 4, 16) + align_up(L.h * 4, 16) + align_up(L.h * 4, 16);
      }
    }
    Q.ncells = cellBase;
    Q.slabBytes = off;
    Q.slotsPerFrame = slot;
    Q.candCap = slot;
    if (slot >= (1ll << 24)) qtOk = false;   // k_quadtree3 packs a candidate index into 24 bits (not reachable below 4096 x 4096)

    // bilinear tables, SURVEY.md Appendix B.2 (float/double arithmetic exactly as cv::resize)
    std::vector<uint8_t> tab(tableBytes);
    int rc = d_tables.ensure(tableBytes);
    if (rc) return rc;
    size_t cur = 0;
    const int* xofsH[kMaxLevels] = {};
    const int* yofsH[kMaxLevels] = {};
    for (int l = 1; l < nlevels; l++) {
      LevelGeom& L = Q.lv[l];
      const int sw = Q.lv[l - 1].w, sh = Q.lv[l - 1].h, dw = L.w, dh = L.h;
      const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
      int* xofs = (int*)(tab.data() + cur);
      xofsH[l] = xofs;
      L.xofs = (const int*)(d_tables.p + cur);
      cur += align_up(dw * 4, 16);
      short* xa = (short*)(tab.data() + cur);
      L.xalpha = (const short*)(d_tables.p + cur);
      cur += align_up(dw * 4, 16);
      int* yofs = (int*)(tab.data() + cur);
      yofsH[l] = yofs;
      L.yofs = (const int*)(d_tables.p + cur);
      cur += align_up(dh * 4, 16);
      short* yb = (short*)(tab.data() + cur);
      L.ybeta = (const short*)(d_tables.p + cur);
      cur += align_up(dh * 4, 16);
      unsigned* yc = (unsigned*)(tab.data() + cur);
      L.yofc = (const unsigned*)(d_tables.p + cur);
      cur += align_up(dh * 4, 16);
      for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[2 * dx] = sat_short(cv_round_f((1.f - fx) * 2048));
        xa[2 * dx + 1] = sat_short(cv_round_f(fx * 2048));
      }
      for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        yc[dy] = (unsigned)std::min(std::max(sy, 0), sh - 1) | ((unsigned)std::min(std::max(sy + 1, 0), sh - 1) << 16);
        yb[2 * dy] = sat_short(cv_round_f((1.f - fy) * 2048));
        yb[2 * dy + 1] = sat_short(cv_round_f(fy * 2048));
      }
      // k_resize_fixed fetches the four rows of a thread (4k .. 4k + 3) as ONE 16-byte piece of each table: the padding up to a
      // multiple of four rows repeats the last row
      for (int dy = dh; dy < (int)align_up(dh, 4); dy++) {
        yc[dy] = yc[dh - 1];
        yb[2 * dy] = yb[2 * (dh - 1)];
        yb[2 * dy + 1] = yb[2 * (dh - 1) + 1];
      }
      // largest source footprint of a 64x64 output tile (k_resize stages it in LDS)
      int maxW = 1, maxH = 1;
      for (int x0 = 0; x0 < dw; x0 += 64) {
        const int x1 = std::min(x0 + 64, dw) - 1;
        maxW = std::max(maxW, std::min(xofs[x1] + 1, sw - 1) - xofs[x0] + 1);
      }
      for (int y0 = 0; y0 < dh; y0 += 64) {
        const int y1 = std::min(y0 + 64, dh) - 1;
        const int r0 = std::min(std::max(yofs[y0], 0), sh - 1), r1 = std::min(std::max(yofs[y1] + 1, 0), sh - 1);
        maxH = std::max(maxH, r1 - r0 + 1);
      }
      L.rzPitch = (maxW + 3 + 3) & ~3;
      L.rzRows = maxH;
    }
    HIP_TRY(hipMemcpyAsync(d_tables.p, tab.data(), tableBytes, hipMemcpyHostToDevice, stream));
    // one-launch pyramid for small batches (k_pyramid_cone): per tile column / row of the top level, the range of every
    // lower level it depends on; the cone starts at the lowest level whose regions still fit the LDS budget
    // (the same cone from a higher base level serves an experiment of round 6: the TAIL of a batch's pyramid in one launch)
    auto buildCone = [&](int tile, int baseMin, ConeParams& out, DevBuf<ConeRange>& tabBuf, bool& ok) -> int {
      ok = false;
      if (!(nlevels >= 3 && scaleFactor <= 1.5f)) return ORBFE_OK;
      std::vector<ConeRange> coneTab;
      const int top = nlevels - 1;
      // a block is bound by its own CU's issue rate (1 366 vector + 1 141 scalar instructions per wave, 16 waves: tools/pmc_cone.sh): the
      // smallest tile that still gives every block a CU of its own -- 26 at 1080p: 21 x 12 = 252 blocks on 256 CUs (28: 220; round 6)
      if (tile <= 0) {
        tile = 32;
        for (int t = 16; t < 32; t += 2)
          if (((Q.lv[top].w + t - 1) / t) * ((Q.lv[top].h + t - 1) / t) <= 256) { tile = t; break; }
      }
      const int tx = (Q.lv[top].w + tile - 1) / tile, ty = (Q.lv[top].h + tile - 1) / tile;
      coneTab.assign((size_t)(tx + ty) * kMaxLevels, ConeRange{0, 0});
      int maxW[kMaxLevels] = {}, maxH[kMaxLevels] = {};
      for (int b = 0; b < tx + ty; b++) {
        const bool isX = b < tx;
        int lo = (isX ? b : b - tx) * tile, hi = std::min(lo + tile, isX ? Q.lv[top].w : Q.lv[top].h) - 1;
        ConeRange* R = coneTab.data() + (size_t)b * kMaxLevels;
        for (int l = top;; l--) {
          R[l] = ConeRange{lo, hi};
          int& m = isX ? maxW[l] : maxH[l];
          m = std::max(m, hi - lo + 1);
          if (l == 0) break;
          const int dn = isX ? Q.lv[l].w : Q.lv[l].h, sn = isX ? Q.lv[l - 1].w : Q.lv[l - 1].h;
          const int* ofs = isX ? xofsH[l] : yofsH[l];
          const int nlo = lo == 0 ? 0 : std::min(std::max(ofs[lo], 0), sn - 1);
          const int nhi = hi == dn - 1 ? sn - 1 : std::min(std::max(ofs[hi] + 1, 0), sn - 1);
          lo = nlo; hi = nhi;
        }
      }
      for (int base = std::max(0, baseMin); base < top - 1; base++) {
        // regions are padded to dword columns on both sides (k_pyramid_cone)
        auto pitchOf = [&](int l) { return (maxW[l] + 3 + 3) & ~3; };
        auto area = [&](int l) { return (size_t)align_up(pitchOf(l) * maxH[l], 16); };
        size_t hBytes = 0;
        int coefLen = 0;
        for (int l = base + 1; l <= top; l++) {
          hBytes = std::max(hBytes, (size_t)align_up(maxH[l - 1] * pitchOf(l) * 2, 16));
          coefLen = std::max(coefLen, std::max(pitchOf(l), maxH[l]));
        }
        const size_t coefBytes = (size_t)kMaxLevels * 2 * coefLen * sizeof(int2);
        const size_t total = area(base) + area(base + 1) + hBytes + coefBytes;
        if (total > 128 * 1024 || coefLen > 512) continue;
        out.base = base; out.top = top; out.tile = tile; out.tilesX = tx; out.tilesY = ty;
        out.offB = (int)area(base); out.offH = out.offB + (int)area(base + 1); out.offC = out.offH + (int)hBytes;
        out.coefLen = coefLen; out.ldsBytes = (int)total;
        ok = true;
        break;
      }
      if (ok) {
        int rc2;
        if ((rc2 = tabBuf.ensure(coneTab.size()))) return rc2;
        HIP_TRY(hipMemcpyAsync(tabBuf.p, coneTab.data(), sizeof(ConeRange) * coneTab.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // (coneTab is a local)
        out.regX = tabBuf.p;
        out.regY = tabBuf.p + (size_t)tx * kMaxLevels;
      }
      return ORBFE_OK;
    };
    if ((rc = buildCone(coneTile, 0, cone, d_coneTab, coneOk))) return rc;
    tailOk = false;
    if (tailBase > 0 && tailBase < nlevels - 2)
      if ((rc = buildCone(tailTile, tailBase, coneTail, d_coneTailTab, tailOk))) return rc;
    if (tailOk && coneTail.base != tailBase) tailOk = false;
    // per-cell geometry (emit regions of ComputeKeyPointsOctTree's cell grid, ORBextractor.cc:826-844)
    std::vector<CellInfo> cells(Q.ncells);
    for (int l = 0; l < nlevels; l++) {
      const LevelGeom& L = Q.lv[l];
      for (int i = 0; i < L.nRows; i++)
        for (int j = 0; j < L.nCols; j++) {
          CellInfo ci{};
          const int ex0 = kEdge + j * L.wCell, ey0 = kEdge + i * L.hCell;
          const int ew = std::min(L.wCell, L.w - kEdge - ex0), eh = std::min(L.hCell, L.h - kEdge - ey0);
          ci.ex0 = (uint16_t)ex0;
          ci.ey0 = (uint16_t)ey0;
          ci.ew = (int8_t)std::max(-1, std::min(ew, 127));
          ci.eh = (int8_t)std::max(-1, std::min(eh, 127));
          ci.level = (uint8_t)l;
          ci.local = (uint32_t)(i * L.nCols + j);
          ci.slotOff = (uint32_t)(L.slotBase + (long long)ci.local * L.slotCap);
          cells[L.cellBase + ci.local] = ci;
        }
    }
    if ((rc = d_cells.ensure(cells.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(d_cells.p, cells.data(), sizeof(CellInfo) * cells.size(), hipMemcpyHostToDevice, stream));
    // FAST tasks (k_fast_tasks): two horizontally adjacent cells of a cell row per wave where the pair is at most 64
    // pixels wide (queue entries hold x in 8 bits, the pre-test handles 4 groups of 16 pixels per row)
    std::vector<FastTask> tasks;
    tasks.reserve(Q.ncells);
    Q.fastLean = 1;
    for (int l = 0; l < nlevels; l++) {
      LevelGeom& L = Q.lv[l];
      Q.taskStart[l] = (int)tasks.size();
      const bool pairOk = pairCells && 2 * L.wCell <= 64 && L.nCols >= 2;
      L.fastW = pairOk ? 2 * L.wCell : L.wCell;
      const uint32_t geo = fast_task_geo(L.fastW, L.hCell);
      if (geo == 0u && L.nRows * L.nCols > 0) Q.fastLean = 0;
      for (int i = 0; i < L.nRows; i++)
        for (int j = 0; j < L.nCols;) {
          const CellInfo& c0 = cells[L.cellBase + i * L.nCols + j];
          FastTask t{};
          t.ex0 = c0.ex0; t.ey0 = c0.ey0; t.level = (uint8_t)l;
          t.cell0 = (uint32_t)(L.cellBase + i * L.nCols + j);
          t.slotOff0 = c0.slotOff;
          t.roiOff = l == 0 ? 0u : (uint32_t)(L.off + (long long)((int)c0.ey0 - 3) * L.pitch + ((int)c0.ex0 - 3));
          t.pitch = l == 0 ? 0u : (uint32_t)L.pitch;
          t.fastW = (uint8_t)L.fastW; t.hCell = (uint8_t)L.hCell; t.slotCap = (uint16_t)L.slotCap;
          t.geo = geo;
          const bool valid0 = c0.ew > 0 && c0.eh > 0;
          t.ew0 = valid0 ? (uint8_t)c0.ew : 0;
          t.eh = valid0 ? (uint8_t)c0.eh : 0;
          int used = 1;
          if (pairOk && valid0 && j + 1 < L.nCols && c0.ew == L.wCell) {
            const CellInfo& c1 = cells[L.cellBase + i * L.nCols + j + 1];
            if (c1.ew > 0 && c1.eh == c0.eh) { t.ew1 = (uint8_t)c1.ew; used = 2; }
          }
          tasks.push_back(t);
          j += used;
        }
    }
    for (int l = nlevels; l <= kMaxLevels; l++) Q.taskStart[l] = (int)tasks.size();
    if ((rc = d_tasks.ensure(tasks.size()))) return rc;
    if (!d_zeros.p) {
      if ((rc = d_zeros.ensure(256))) return rc;
      HIP_TRY(hipMemsetAsync(d_zeros.p, 0, 256, stream));
    }
    Q.zeros = d_zeros.p;
    HIP_TRY(hipMemcpyAsync(d_tasks.p, tasks.data(), sizeof(FastTask) * tasks.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    Q.cells = d_cells.p;
    Q.tasks = d_tasks.p;
    Q.ntasks = (int)tasks.size();
    P = Q;
    rows = r;
    geomGpuQtOk = qtOk;
    // keypoints one frame can yield: a level returns at most max(N_l + 2, 4 * roots) (ORBextractor.cc:620-773)
    const int kpCap = std::max(kpSum, selPerFrame);
    if (kpCap > kpPerFrameCap) {
      kpPerFrameCap = kpCap;
      batchCap = 0;   // the result arena is carved again
    }
    cols = c;
    batchCap = 0;
    return ORBFE_OK;
  }

  int ensureSubStreams(int nsub) {
    for (int s = 1; s < nsub; s++)
      if (!streams[s]) HIP_TRY(hipStreamCreateWithFlags(&streams[s], hipStreamNonBlocking));
    return ORBFE_OK;
  }

  int inChannels() const { return inFormat == ORBFE_INPUT_GRAY8 ? 1 : (inFormat == ORBFE_INPUT_RGB8 || inFormat == ORBFE_INPUT_BGR8) ? 3 : 4; }

  int setBatch(int nframes, bool hostInput, size_t hostStride = 0) {
    int rc;
    if (nframes > batchCap) {
      // + 64 bytes: k_describe's LDS-DMA fetches 48 bytes per patch row from the dword-aligned-down start, i.e. up to 5 bytes past the
      // last pixel it needs; on the last row of the last level of the last frame that is past the slab
      if ((rc = d_slab.ensure((size_t)P.slabBytes * nframes + 64))) return rc;
      if ((rc = d_cellCount.ensure((size_t)P.ncells * nframes))) return rc;
      if ((rc = d_slots.ensure((size_t)P.slotsPerFrame * nframes))) return rc;
      if ((rc = d_cand.ensure((size_t)P.candCap * nframes))) return rc;
      if ((rc = d_frame0.ensure(2 * (size_t)nframes))) return rc;   // [gray level-0 pointers][raw colour pointers]
      if ((rc = h_frame0.ensure(2 * (size_t)nframes))) return rc;
      const size_t maxKp = (size_t)kpPerFrameCap * nframes;  // >= selPerFrame * nframes
      {
        auto al256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t oLs = 0, oSc = oLs + al256(sizeof(uint32_t) * (kMaxLevels + 1) * nframes),
                     oSel = oSc + al256(sizeof(uint32_t) * kMaxLevels * nframes), oAng = oSel + al256(sizeof(SelKp) * maxKp),
                     oDesc = oAng + al256(sizeof(float) * maxKp), oM12 = oDesc + al256(32 * maxKp),
                     oNm = oM12 + al256(sizeof(int32_t) * (size_t)(selOff[1] - selOff[0]) * nframes), total = oNm + al256(sizeof(int32_t) * nframes);
        if ((rc = d_outArena.ensure(total))) return rc;
        if ((rc = h_outArena.ensure(total))) return rc;
        outArenaBytes = total;
        uint8_t *D = d_outArena.p, *Hh = h_outArena.p;
        d_levelStart.p = (uint32_t*)(D + oLs); h_levelStart.p = (uint32_t*)(Hh + oLs);
        d_selCount.p = (uint32_t*)(D + oSc); h_selCount.p = (uint32_t*)(Hh + oSc);
        d_sel.p = (SelKp*)(D + oSel); h_sel.p = (SelKp*)(Hh + oSel);
        d_angle.p = (float*)(D + oAng); h_angle.p = (float*)(Hh + oAng);
        d_desc.p = D + oDesc; h_desc.p = Hh + oDesc;
        d_m12.p = (int32_t*)(D + oM12); h_m12.p = (int32_t*)(Hh + oM12);
        d_nm.p = (int32_t*)(D + oNm); h_nm.p = (int32_t*)(Hh + oNm);
      }
      if (gpuQuadtree) {
        if ((rc = d_own.ensure((size_t)P.candCap * nframes))) return rc;
      }
      batchCap = nframes;
      if (candHostCap == 0) candHostCap = 96 * 1024;
      if ((rc = h_cand.ensure(candHostCap * (size_t)batchCap))) return rc;
    }
    if (inChannels() > 1) {
      grayPitch = align_up(cols, 256);
      if ((rc = d_gray.ensure((size_t)grayPitch * rows * nframes))) return rc;
    }
    if (hostInput) {
      // rows that are (nearly) contiguous on the host travel as ONE linear DMA per run of frames; a narrow view
      // of a much wider image is copied row by row into a compact device frame
      const size_t rowBytes = (size_t)cols * inChannels();
      inLinear = hostStride >= rowBytes && hostStride - rowBytes <= rowBytes / 8;
      inPitch = inLinear ? (long long)hostStride : align_up((long long)rowBytes, 256);
      if ((rc = d_in.ensure((size_t)inPitch * rows * nframes))) return rc;
    }
    P.slab = d_slab.p;
    P.cellCount = d_cellCount.p;
    P.slots = d_slots.p;
    P.cand = d_cand.p;
    P.levelStart = d_levelStart.p;
    P.frame0 = d_frame0.p;
    return ORBFE_OK;
  }

  // GPU-quadtree path: frame -> pyramid -> FAST -> compaction -> quadtree -> orientation/blur/rBRIEF in ONE
  // stream submission, one synchronisation, fixed-size D2H of the selection slots.
  // H2D of host frames [f0, f0+nf) into d_in on `st`
  // H2D of host frames [f0, f0+nf) on `st`.  devAt[f] receives the device address frame f lands at.  Frames that tile
  // one contiguous host block -- in ANY order: a ring buffer read forwards, backwards or rotated -- travel as ONE DMA
  // (a 66 MB copy runs at 56.8 GB/s on this link, 32 separate 2 MB copies at 28 GB/s); otherwise ascending runs are
  // coalesced and the rest goes frame by frame.
  int uploadFrames(int f0, int nf, const uint8_t* const* gray, size_t stride, int r, int c, hipStream_t st,
                   const uint8_t** devAt) {
    if (!inLinear) {
      for (int f = f0; f < f0 + nf; f++) {
        HIP_TRY(hipMemcpy2DAsync(d_in.p + (size_t)inPitch * rows * f, inPitch, gray[f], stride, c, r, hipMemcpyHostToDevice, st));
        devAt[f] = d_in.p + (size_t)inPitch * rows * f;
      }
      return ORBFE_OK;
    }
    const size_t frameBytes = stride * (size_t)r;
    const size_t lastBytes = stride * (size_t)(r - 1) + c;   // the last frame of a block may end with its last pixel
    if (nf > 1 && nf <= 64) {
      const uint8_t* lo = gray[f0];
      for (int f = f0 + 1; f < f0 + nf; f++) lo = std::min(lo, gray[f]);
      uint64_t seen = 0;
      bool tile = true;
      for (int f = f0; f < f0 + nf && tile; f++) {
        const size_t d = (size_t)(gray[f] - lo);
        const size_t k = d / frameBytes;
        tile = d % frameBytes == 0 && k < (size_t)nf && !(seen >> k & 1);
        seen |= (uint64_t)1 << k;
      }
      if (tile) {
        uint8_t* base = d_in.p + frameBytes * f0;
        HIP_TRY(hipMemcpyAsync(base, lo, frameBytes * (size_t)(nf - 1) + lastBytes, hipMemcpyHostToDevice, st));
        for (int f = f0; f < f0 + nf; f++) devAt[f] = base + (size_t)(gray[f] - lo);
        return ORBFE_OK;
      }
    }
    for (int f = f0; f < f0 + nf;) {
      int g = f + 1;
      while (g < f0 + nf && gray[g] == gray[g - 1] + frameBytes) g++;   // frames contiguous in host memory
      HIP_TRY(hipMemcpyAsync(d_in.p + frameBytes * f, gray[f], frameBytes * (size_t)(g - f - 1) + lastBytes, hipMemcpyHostToDevice, st));
      for (int k = f; k < g; k++) devAt[k] = d_in.p + frameBytes * k;
      f = g;
    }
    return ORBFE_OK;
  }

  // host-quadtree path: frame f always lands in slot f of d_in (the pointer table is uploaded before the frames)
  int uploadSlots(int f0, int nf, const uint8_t* const* gray, size_t stride, int r, int c, hipStream_t st) {
    for (int f = f0; f < f0 + nf; f++) {
      if (inLinear)
        HIP_TRY(hipMemcpyAsync(d_in.p + (size_t)inPitch * rows * f, gray[f], stride * (size_t)(r - 1) + c, hipMemcpyHostToDevice, st));
      else
        HIP_TRY(hipMemcpy2DAsync(d_in.p + (size_t)inPitch * rows * f, inPitch, gray[f], stride, c, r, hipMemcpyHostToDevice, st));
    }
    return ORBFE_OK;
  }

  int runGpuQt(int nframes, const uint8_t* const* gray, bool onDevice, int r, int c, size_t stride, OrbfeKeyPoint* kps,
               uint8_t* desc, int cap, int* n_out) {
    int rc = submitGpuQt(nframes, gray, onDevice, r, c, stride);
    if (rc) return rc;
    return waitGpuQt(kps, desc, cap, n_out);
  }

  bool submitProfiled = false, fastTimed = false;
  std::vector<const uint8_t*> devAt;   // device address of every uploaded host frame (uploadFrames)
  int pendingFrames = 0;   // frames of the submitted, not yet collected batch (0 = none)
  double tSubmit0 = 0, tSubmit1 = 0;

  struct MatchSpec { orbfe_sfi_chain* chain; float bounds[4]; int window; float nnratio; int checkOri; };

  int submitGpuQt(int nframes, const uint8_t* const* gray, bool onDevice, int r, int c, size_t stride,
                  const MatchSpec* ms = nullptr) {
    if (pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    int rc;
    if ((rc = setGeometry(r, c))) return rc;
    if (!geomGpuQtOk) {
      set_err("image %dx%d: a pyramid level has more than 4 (or no) quadtree root nodes; only the blocking "
              "orbfe_extract / orbfe_extract_batch calls (host quadtree path) handle such aspect ratios", c, r);
      return ORBFE_ERR_INVALID;
    }
    if ((rc = setBatch(nframes, !onDevice, stride))) return rc;
    const double t0 = now_ms();
    hipStream_t st = streams[0];
    const int ch = inChannels();
    const int rowBytes = c * ch;
    for (int f = 0; f < nframes; f++)
      if (!gray[f]) { set_err("frame %d is NULL", f); return ORBFE_ERR_INVALID; }
    devAt.resize(nframes);
    // Where the host frames live decides how they travel.  EVERY frame is looked at (a one- or two-frame call may mix a
    // page-locked frame with a pageable one, or hand over a view that runs past a registered range):
    //   mapped   = page-locked (orbfe_host_alloc) or registered (orbfe_host_register) over its whole extent: the GPU can read it in place;
    //   pinned   = all frames mapped -> asynchronous copy commands on the upload lane are truly asynchronous;
    //   otherwise the runtime stages the copy synchronously on this handle's stream.
    bool pinned = !onDevice;
    std::vector<const uint8_t*>& mappedAt = devAt;   // (re-used below: device-visible address of every mapped host frame)
    if (!onDevice) {
      const size_t extent = stride * (size_t)(r - 1) + (size_t)rowBytes;
      // (a batch for the copy engine needs no mapping: frame 0 decides the route, a pageable straggler is merely staged)
      const int ncheck = nframes <= coneMaxFrames ? nframes : 1;
      for (int f = 0; f < ncheck && pinned; f++) {
        void* dp = nullptr;
        void* dpEnd = nullptr;
        // (hipHostGetDevicePointer succeeds for page-locked and registered host memory only: no separate attribute query)
        const bool ok = hipHostGetDevicePointer(&dp, const_cast<uint8_t*>(gray[f]), 0) == hipSuccess && dp &&
                        hipHostGetDevicePointer(&dpEnd, const_cast<uint8_t*>(gray[f]) + extent - 1, 0) == hipSuccess &&
                        (const uint8_t*)dpEnd - (const uint8_t*)dp == (ptrdiff_t)(extent - 1);
        if (!ok) { (void)hipGetLastError(); pinned = false; break; }
        mappedAt[f] = static_cast<const uint8_t*>(dp);
      }
    }
    if (!onDevice && !pinned) {
      // pageable (or partly pageable) memory: the runtime stages the copy synchronously; keep it on this handle's stream
      if ((rc = uploadFrames(0, nframes, gray, stride, r, rowBytes, st, devAt.data()))) return rc;
    } else if (!onDevice && ingestKernel && nframes <= coneMaxFrames &&
               ((((uintptr_t)gray[0] | (uintptr_t)gray[nframes - 1] | (uintptr_t)stride | (uintptr_t)rowBytes) & 3) == 0)) {   // (byte-aligned views: the copy engine)
      // a one- or two-frame call from page-locked memory: the compute stream fetches the frame itself (k_ingest) -- no copy command
      // on another queue, no event between the queues.  The kernel reads the frame at the address the GPU sees it at: the host
      // address itself for orbfe_host_alloc memory, the mapping of a buffer the caller registered (orbfe_host_register) otherwise.
      for (int f = 0; f < nframes; f++) {
        uint8_t* d = d_in.p + (size_t)inPitch * rows * f;
        launch_ingest(mappedAt[f], (long long)stride, d, inPitch, rowBytes, r, st);
        devAt[f] = d;
      }
      HIP_TRY(hipGetLastError());
    } else if (!onDevice) {
      UploadLane* lane = upload_lane(device);
      if (!lane) { set_err("cannot create the upload stream"); return ORBFE_ERR_HIP; }
      std::lock_guard<std::mutex> lk(lane->mu);
      if ((rc = uploadFrames(0, nframes, gray, stride, r, rowBytes, lane->stream, devAt.data()))) return rc;
      HIP_TRY(hipEventRecord(evUpload, lane->stream));
      HIP_TRY(hipStreamWaitEvent(st, evUpload, 0));
    }
    bool rawAligned = ((onDevice ? (long long)stride : inPitch) & 3) == 0;
    for (int f = 0; f < nframes; f++) {
      const uint8_t* raw = onDevice ? gray[f] : devAt[f];
      rawAligned = rawAligned && ((uintptr_t)raw & (ch == 4 ? 15 : 3)) == 0;
      h_frame0.p[f] = ch == 1 ? raw : d_gray.p + (size_t)grayPitch * rows * f;
      h_frame0.p[nframes + f] = raw;
    }
    const long long rawStride = onDevice ? (long long)stride : inPitch;
    P.stride0 = ch == 1 ? rawStride : grayPitch;
    P.frameBase = 0;
    P.gaussVariant = gaussVariant;
    if (ch == 1 && nframes <= 2) {   // the level-0 pointers travel in the kernel arguments
      P.frame0 = nullptr;
      P.frameInline[0] = h_frame0.p[0];
      P.frameInline[1] = h_frame0.p[nframes - 1];
    } else {
      P.frame0 = d_frame0.p;
      HIP_TRY(hipMemcpyAsync(d_frame0.p, h_frame0.p, sizeof(void*) * nframes * (ch == 1 ? 1 : 2), hipMemcpyHostToDevice, st));
    }
    if (ch > 1) {
      // OpenCV RGB2Gray<uchar>: 15-bit coefficients {R 9798, G 19235, B 3735} (>= 4.1.1) or 14-bit {4899, 9617, 1868}
      const bool q15 = grayVariant == ORBFE_GRAY_Q15;
      const int cr = q15 ? 9798 : 4899, cg = q15 ? 19235 : 9617, cb = q15 ? 3735 : 1868;
      const bool rgb = inFormat == ORBFE_INPUT_RGB8 || inFormat == ORBFE_INPUT_RGBA8;
      const int coef[3] = {rgb ? cr : cb, cg, rgb ? cb : cr};
      launch_to_gray(d_frame0.p + nframes, rawStride, d_frame0.p, grayPitch, r, c, ch, coef, q15 ? 15 : 14, rawAligned, nframes, st);
    }
    const bool prof = profileKernels;
    QP.cand = d_cand.p; QP.levelStart = d_levelStart.p; QP.candCap = P.candCap; QP.nlevels = nlevels; QP.frameBase = 0;
    QP.own = d_own.p;
    QP.sel = d_sel.p; QP.selCount = d_selCount.p; QP.selPerFrame = selPerFrame;
    // Latency-bound plain extraction: the kernels write the results straight into the page-locked host arena
    // (the quadtree a second copy of its selection, the descriptor kernel its only copy), so no copy command follows
    // the last kernel.  Matching and bag-of-words read angles / descriptors on the device and keep the copy.
    const bool small = nframes <= coneMaxFrames;
    const bool zeroCopy = zeroCopyOut && small && !voc && !(ms && ms->chain);
    submitZeroCopy = zeroCopy;
    QP.jump = qtJump ? 1 : 0;
    // Batches, ORBFE_ZERO_COPY=2 (round 6): the kernels that produce a result store a second copy of it into the page-locked arena -- the
    // quadtree its selection, the descriptor kernel angles and descriptors, SearchForInitialization its match vectors (their only copy)
    // -- so no copy command follows the batch (16 % of the GPU's kernel time in round 5's profile); the first copies stay in HBM for the
    // matching chain, the bag-of-words descent and resident frames.  Measured equal to the copy command (92.0 against 92.1 k frames/s
    // with four batches in flight, 88.6 against 89.0 k with three: gpurun r06l) -- the stores cost what the copy costs -- so the copy
    // command stays the default for batches.
    const bool kernelOut = zeroCopyBatches && !small;
    QP.selHost = (zeroCopy || kernelOut) ? h_sel.p : nullptr;
    QP.selCountHost = (zeroCopy || kernelOut) ? h_selCount.p : nullptr;
    for (int l = 0; l < nlevels; l++) {
      QP.levW[l] = P.lv[l].w; QP.levH[l] = P.lv[l].h; QP.nfeat[l] = nfeat[l]; QP.selOff[l] = selOff[l];
      QP.candBase[l] = P.lv[l].slotBase;
    }
    float* angOut = zeroCopy ? h_angle.p : d_angle.p;
    uint8_t* descOut = zeroCopy ? h_desc.p : d_desc.p;
    // a one- or two-frame call is latency-bound and alone on the chip: the quadtree keeps its candidates in LDS, four waves share a keypoint
    const int qtLds = small ? qtLdsBudget : 0;
    const bool four = describe4 && small;
    // a latency-bound call compacts per level (k_compact_local: level-local lists, every wave's prefix in one memory round trip)
    const bool localCand = small && localLists;
    submitLocalCand = localCand;
    QP.levelLocal = localCand ? 1 : 0;
    {
      if (prof) HIP_TRY(hipEventRecord(ev[0][0], st));
      if (launch_pyramid(P, nframes, st, coneOk && small ? &cone : (tailOk && !small ? &coneTail : nullptr))) { set_err("cannot configure the pyramid kernel"); return ORBFE_ERR_HIP; }
      // the dominant kernel is timed in every batch (bench.py roofline); a latency-bound one- or two-frame call does
      // without the two markers (each costs a few microseconds of dependent-launch gap) unless profiling is on
      const bool timeFast = prof || !small;
      fastTimed = timeFast;
      if (timeFast) HIP_TRY(hipEventRecord(ev[0][1], st));
      launch_fast(P, nframes, st);
      if (timeFast) HIP_TRY(hipEventRecord(ev[0][2], st));
      if (localCand) launch_compact_local(P, nframes, st, 0, nlevels);
      else launch_compact(P, nframes, st);
      if (prof) HIP_TRY(hipEventRecord(ev[0][3], st));
      if (prof) HIP_TRY(hipEventRecord(evQt[0], st));
      if (launch_quadtree(QP, nframes, st, qtLds)) { set_err("cannot configure the quadtree kernel"); return ORBFE_ERR_HIP; }
      if (prof) HIP_TRY(hipEventRecord(evQt[1], st));
      if (prof) HIP_TRY(hipEventRecord(ev[0][4], st));
      launch_describe_slots(P, d_sel.p, nframes * selPerFrame, angOut, descOut, d_selCount.p, selPerFrame, selOff, st, four,
                            kernelOut ? h_angle.p : nullptr, kernelOut ? h_desc.p : nullptr);
      if (prof) HIP_TRY(hipEventRecord(ev[0][5], st));
      HIP_TRY(hipGetLastError());
    }
    const int nslots = nframes * selPerFrame;
    pendingBow = false;
    if (voc) {
      if ((rc = d_bow.ensure(nslots)) || (rc = h_bow.ensure(nslots))) return rc;
      if ((rc = bow_launch_descend(voc, d_desc.p, nslots, vocLevelsup, d_bow.p, st))) return rc;   // dead slots descend too
      HIP_TRY(hipMemcpyAsync(h_bow.p, d_bow.p, sizeof(uint2) * nslots, hipMemcpyDeviceToHost, st));
      pendingBow = true;
    }
    pendingMatched = false;
    if (ms && ms->chain) {
      // SearchForInitialization of every frame against its predecessor, on the data that is already in HBM
      orbfe_sfi_chain& ch = *ms->chain;
      const int n0cap = selOff[1] - selOff[0];
      if (ch.n0cap != n0cap || ch.device != device) { set_err("match chain belongs to a different extractor configuration"); return ORBFE_ERR_INVALID; }
      if ((rc = d_sfiOrder.ensure((size_t)batchCap * n0cap))) return rc;
      if ((rc = d_sfiOrderCount.ensure(batchCap))) return rc;
      if ((rc = d_sfiPool.ensure((size_t)batchCap * n0cap * n0cap))) return rc;
      if ((rc = d_sfiPcount.ensure((size_t)batchCap * n0cap))) return rc;
      SfiParams SP{};
      SP.sel = d_sel.p; SP.angle = d_angle.p; SP.desc = d_desc.p; SP.selCount = d_selCount.p;
      SP.selPerFrame = selPerFrame; SP.n0cap = n0cap; SP.frameBase = 0;
      const int prev = (int)((ch.seq + 1) & 1), cur = (int)(ch.seq & 1);   // buffer written by the previous / this batch
      SP.carrySel = ch.sel[prev].p; SP.carryAngle = ch.angle[prev].p; SP.carryDesc = ch.desc[prev].p;
      const bool iso = ch.isolated;   // every batch stands alone: no carry in, no carry out
      SP.carryCount = ch.count.p + ((ch.seq == 0 || iso) ? 2 : prev);
      SP.minX = ms->bounds[0]; SP.minY = ms->bounds[2];
      SP.invW = static_cast<float>(64) / static_cast<float>(ms->bounds[1] - ms->bounds[0]);   // Frame.cc:98
      SP.invH = static_cast<float>(48) / static_cast<float>(ms->bounds[3] - ms->bounds[2]);   // Frame.cc:99
      SP.window = (float)ms->window; SP.nnratio = ms->nnratio; SP.checkOri = ms->checkOri;
      SP.order = d_sfiOrder.p; SP.orderCount = d_sfiOrderCount.p; SP.pool = d_sfiPool.p; SP.pcount = d_sfiPcount.p;
      SP.matches12 = kernelOut ? h_m12.p : d_m12.p; SP.nmatches = kernelOut ? h_nm.p : d_nm.p;
      if (ch.seq > 0 && !iso) HIP_TRY(hipStreamWaitEvent(st, ch.ready[prev], 0));
      launch_sfi(SP, nframes, st);
      HIP_TRY(hipGetLastError());
      if (!iso) {
        // hand the last frame's level-0 data to the next batch
        launch_sfi_carry(SP, nframes - 1, ch.sel[cur].p, ch.angle[cur].p, ch.desc[cur].p, ch.count.p + cur, st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ch.ready[cur], st));
      }
      ch.seq++;
      pendingMatched = true;
    }
    if (!zeroCopy && !kernelOut) HIP_TRY(hipMemcpyAsync(h_outArena.p, d_outArena.p, outArenaBytes, hipMemcpyDeviceToHost, st));   // all results, one copy
    submitProfiled = prof;
    tSubmit0 = t0;
    tSubmit1 = now_ms();
    pendingFrames = nframes;
    return ORBFE_OK;
  }

  // the GPU side of the submitted batch only (orbfe_extract_batch_wait): the caller collects later
  int waitOnly() {
    if (!pendingFrames) return ORBFE_OK;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = streams[0];
    if (pollWaitUs > 0) {
      hipError_t q;
      while ((q = hipStreamQuery(st)) == hipErrorNotReady) usleep((useconds_t)pollWaitUs);
      if (q != hipSuccess) HIP_TRY(q);
    }
    HIP_TRY(hipStreamSynchronize(st));
    return ORBFE_OK;
  }

  int waitGpuQt(OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out, int32_t* matches12 = nullptr,
                int* nmatches = nullptr) {
    if (!pendingFrames) { set_err("no submitted batch to collect"); return ORBFE_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    const int nframes = pendingFrames;
    pendingFrames = 0;
    hipStream_t st = streams[0];
    const double t0 = tSubmit0, t1 = tSubmit1;
    const double t1b = now_ms();
    if (pollWaitUs > 0) {   // sleep-poll instead of the runtime's spinning wait (pipelined callers: the wake-up delay is hidden)
      hipError_t q;
      while ((q = hipStreamQuery(st)) == hipErrorNotReady) usleep((useconds_t)pollWaitUs);
      if (q != hipSuccess) HIP_TRY(q);
    }
    HIP_TRY(hipStreamSynchronize(st));
    const double t2 = now_ms();
    {
      float ms = 0;
      if (fastTimed && hipEventElapsedTime(&ms, ev[0][1], ev[0][2]) == hipSuccess) kernMs[1] += ms;
      if (submitProfiled) {
        if (hipEventElapsedTime(&ms, ev[0][0], ev[0][1]) == hipSuccess) kernMs[0] += ms;
        if (hipEventElapsedTime(&ms, ev[0][2], ev[0][3]) == hipSuccess) kernMs[2] += ms;
        if (hipEventElapsedTime(&ms, ev[0][4], ev[0][5]) == hipSuccess) kernMs[3] += ms;
        if (hipEventElapsedTime(&ms, evQt[0], evQt[1]) == hipSuccess) kernMs[4] += ms;
      }
      if (fastTimed) {
        kernBatches++;
        kernFrames += nframes;
      }
    }
    int status = ORBFE_OK;
    if (pendingBow) bowKp.resize(nframes);
    else bowKp.clear();
    for (int f = 0; f < nframes; f++) {
      OrbfeKeyPoint* ko = kps + (size_t)f * cap;
      uint8_t* dout = desc + (size_t)f * cap * 32;
      if (pendingBow) bowKp[f].clear();
      int n = 0;
      for (int l = 0; l < nlevels; l++) {
        const int cnt = (int)h_selCount.p[(size_t)f * kMaxLevels + l];
        const size_t base = (size_t)f * selPerFrame + selOff[l];
        const float scale = sf[l];
        const float size = (float)(int)(31 * sf[l]);
        for (int i = 0; i < cnt; i++) {
          if (n >= cap) { n++; continue; }
          const SelKp s = h_sel.p[base + i];
          OrbfeKeyPoint kp;
          kp.x = (float)(s.xy & 0xffff);
          kp.y = (float)(s.xy >> 16);
          if (l != 0) { kp.x *= scale; kp.y *= scale; }
          kp.size = size;
          kp.angle = h_angle.p[base + i];
          kp.response = (float)(s.lf >> 24);
          kp.octave = l;
          kp.class_id = -1;
          ko[n] = kp;
          memcpy(dout + (size_t)n * 32, h_desc.p + (base + i) * 32, 32);
          if (pendingBow) bowKp[f].push_back(h_bow.p[base + i]);
          n++;
        }
      }
      n_out[f] = n;
      if (n > cap) {
        set_err("frame %d produced %d keypoints, cap is %d", f, n, cap);
        status = ORBFE_ERR_OVERFLOW;
      }
    }
    if (matches12 && nmatches) {
      const int n0cap = selOff[1] - selOff[0];
      for (int f = 0; f < nframes; f++) {
        int32_t* row = matches12 + (size_t)f * cap;
        const int m = std::min(cap, n0cap);
        if (pendingMatched) {
          memcpy(row, h_m12.p + (size_t)f * n0cap, sizeof(int32_t) * m);
          nmatches[f] = h_nm.p[f];
        } else {
          nmatches[f] = 0;
        }
        for (int i = pendingMatched ? m : 0; i < cap; i++) row[i] = -1;
      }
    }
    const double t3 = now_ms();
    stageMs[0] = (float)(t1 - t0);   // enqueue
    stageMs[1] = (float)(t2 - t1b);  // GPU wait inside the collect call
    stageMs[2] = 0;
    stageMs[3] = (float)(t3 - t2);   // output assembly
    stageMs[4] = (float)(t3 - t0);
    lastFrames = nframes;
    lastGpuQt = true;
    lastZeroCopy = submitZeroCopy;
    lastLocalCand = submitLocalCand;
    return status;
  }

  // One batch = up to kMaxSub sub-batches, each on its own HIP stream, software-pipelined so that the
  // host-side work of sub-batch s (candidate D2H, quadtrees) overlaps the GPU work of the others:
  //   all s : [upload] pyramid -> FAST -> compaction -> D2H level offsets        (enqueued up front)
  //   per s : wait -> D2H candidates -> host quadtrees -> H2D selection -> describe -> D2H (async)
  //   all s : wait, assemble cv::KeyPoint-compatible outputs.
  int run(int nframes, const uint8_t* const* gray, bool onDevice, int r, int c, size_t stride, OrbfeKeyPoint* kps,
          uint8_t* desc, int cap, int* n_out) {
    if (pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    int rc;
    if ((rc = setGeometry(r, c))) return rc;
    if ((rc = setBatch(nframes, !onDevice, stride))) return rc;
    const double t0 = now_ms();
    if (!pool) {
      pool.reset(new HostPool(hostThreads));
      workers.resize(hostThreads);
    }
    if (inChannels() != 1) { set_err("colour input needs the GPU quadtree path (unset ORBFE_HOST_QUADTREE)"); return ORBFE_ERR_INVALID; }
    const int nsub = std::min(nframes, std::min(kMaxSub, std::max(1, subBatches)));
    const int maxKp = kpPerFrameCap;
    if ((rc = ensureSubStreams(nsub))) return rc;
    int subF0[kMaxSub + 1];
    for (int s = 0; s <= nsub; s++) subF0[s] = (int)((long long)nframes * s / nsub);

    devAt.resize(nframes);
    for (int f = 0; f < nframes; f++) {
      if (!gray[f]) { set_err("frame %d is NULL", f); return ORBFE_ERR_INVALID; }
      // per-frame copies on this path (sub-batches upload on their own streams): frame f lands in slot f
      h_frame0.p[f] = onDevice ? gray[f] : d_in.p + (size_t)inPitch * rows * f;
    }
    P.stride0 = onDevice ? (long long)stride : inPitch;
    P.gaussVariant = gaussVariant;   // (the host-quadtree route describes with launch_describe: same kernel, same variant)
    P.frame0 = d_frame0.p;
    HIP_TRY(hipMemcpyAsync(d_frame0.p, h_frame0.p, sizeof(void*) * nframes, hipMemcpyHostToDevice, streams[0]));
    HIP_TRY(hipEventRecord(evFrame0, streams[0]));
    // ---- stage 1 for every sub-batch -----------------------------------------------------
    for (int s = 0; s < nsub; s++) {
      hipStream_t st = streams[s];
      const int f0 = subF0[s], nf = subF0[s + 1] - f0;
      if (s) HIP_TRY(hipStreamWaitEvent(st, evFrame0, 0));
      if (!onDevice && (rc = uploadSlots(f0, nf, gray, stride, r, c, st))) return rc;
      PyramidParams Q = P;
      Q.frameBase = f0;
      HIP_TRY(hipEventRecord(ev[s][0], st));
      if (launch_pyramid(Q, nf, st, coneOk && nf <= coneMaxFrames ? &cone : nullptr)) { set_err("cannot configure the pyramid kernel"); return ORBFE_ERR_HIP; }
      HIP_TRY(hipEventRecord(ev[s][1], st));
      launch_fast(Q, nf, st);
      HIP_TRY(hipEventRecord(ev[s][2], st));
      launch_compact(Q, nf, st);
      HIP_TRY(hipEventRecord(ev[s][3], st));
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(h_levelStart.p + (size_t)f0 * (kMaxLevels + 1), d_levelStart.p + (size_t)f0 * (kMaxLevels + 1),
                             sizeof(uint32_t) * (kMaxLevels + 1) * nf, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(evS1[s], st));
    }
    // ---- per sub-batch: candidates to the host, quadtrees, stage 2 ----------------------------
    double tHostQt = 0, tWait = 0;
    meta.assign((size_t)nframes * maxKp, Meta{});
    frameKpCount.assign(nframes, 0);
    frameKpBase.assign(nframes, 0);
    if ((int)taskOut.size() < nframes * nlevels) taskOut.resize((size_t)nframes * nlevels);
    for (int s = 0; s < nsub; s++) {
      hipStream_t st = streams[s];
      const int f0 = subF0[s], nf = subF0[s + 1] - f0;
      double ta = now_ms();
      HIP_TRY(hipEventSynchronize(evS1[s]));
      // candidates: per-frame host regions of candHostCap entries (grown on demand)
      uint32_t need = 0;
      for (int f = f0; f < f0 + nf; f++) need = std::max(need, h_levelStart.p[(size_t)f * (kMaxLevels + 1) + nlevels]);
      if (need > candHostCap) {
        for (int q = 0; q < nsub; q++) HIP_TRY(hipStreamSynchronize(streams[q]));  // nobody may be writing h_cand
        // sub-batches < s are already consumed, later ones have not copied yet: nothing to preserve
        candHostCap = (size_t)need + need / 4 + 1024;
        if ((rc = h_cand.ensure(candHostCap * (size_t)batchCap))) return rc;
      }
      for (int f = f0; f < f0 + nf; f++) {
        const size_t n = h_levelStart.p[(size_t)f * (kMaxLevels + 1) + nlevels];
        if (n)
          HIP_TRY(hipMemcpyAsync(h_cand.p + candHostCap * f, d_cand.p + (size_t)P.candCap * f, n * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, st));
      }
      HIP_TRY(hipStreamSynchronize(st));
      double tb = now_ms();
      tWait += tb - ta;
      // host quadtree per (frame, level): DistributeOctTree, ORBextractor.cc:876-877
      pool->parallelFor(nf * nlevels, [&](int task, int wid) {
        const int f = f0 + task / nlevels, l = task % nlevels;
        std::vector<Meta>& out = taskOut[(size_t)f * nlevels + l];
        out.clear();
        const uint32_t* ls = h_levelStart.p + (size_t)f * (kMaxLevels + 1);
        const int n = (int)(ls[l + 1] - ls[l]);
        if (n <= 0) return;
        Worker& w = workers[wid];
        const uint32_t* cd = h_cand.p + candHostCap * f + ls[l];
        w.qx.resize(n); w.qy.resize(n); w.qs.resize(n);
        for (int i = 0; i < n; i++) {
          const uint32_t v = cd[i];
          w.qx[i] = (int16_t)((int)(v & 0xfff) - kBorder);
          w.qy[i] = (int16_t)((int)((v >> 12) & 0xfff) - kBorder);
          w.qs[i] = (uint8_t)(v >> 24);
        }
        const LevelGeom& L = P.lv[l];
        w.qt.distribute(w.qx.data(), w.qy.data(), w.qs.data(), n, kBorder, L.w - kBorder, kBorder, L.h - kBorder,
                        nfeat[l], w.qsel);
        out.reserve(w.qsel.size());
        for (int k : w.qsel) {
          Meta m;
          m.x = (int16_t)(w.qx[k] + kBorder);
          m.y = (int16_t)(w.qy[k] + kBorder);
          m.score = w.qs[k];
          m.level = (uint8_t)l;
          out.push_back(m);
        }
      });
      // selection of this sub-batch, packed from entry f0*maxKp
      const int selBase = f0 * maxKp;
      int nsel = 0;
      for (int f = f0; f < f0 + nf; f++) {
        frameKpBase[f] = selBase + nsel;
        for (int l = 0; l < nlevels; l++) {
          for (const Meta& m : taskOut[(size_t)f * nlevels + l]) {
            meta[selBase + nsel] = m;
            SelKp sk;
            sk.xy = (uint32_t)m.x | ((uint32_t)m.y << 16);
            sk.lf = (uint32_t)l | ((uint32_t)f << 8);
            h_sel.p[selBase + nsel++] = sk;
          }
        }
        frameKpCount[f] = selBase + nsel - frameKpBase[f];
      }
      subSel[s] = nsel;
      tHostQt += now_ms() - tb;
      // stage 2: orientation + blur + rBRIEF for this sub-batch (asynchronous)
      if (nsel > 0) {
        HIP_TRY(hipMemcpyAsync(d_sel.p + selBase, h_sel.p + selBase, sizeof(SelKp) * nsel, hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(ev[s][4], st));
        launch_describe(P, d_sel.p + selBase, nsel, d_angle.p + selBase, d_desc.p + (size_t)selBase * 32, st);
        HIP_TRY(hipEventRecord(ev[s][5], st));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_angle.p + selBase, d_angle.p + selBase, sizeof(float) * nsel, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_desc.p + (size_t)selBase * 32, d_desc.p + (size_t)selBase * 32, (size_t)32 * nsel,
                               hipMemcpyDeviceToHost, st));
      }
    }
    const double t3 = now_ms();
    for (int s = 0; s < nsub; s++) HIP_TRY(hipStreamSynchronize(streams[s]));
    const double t4 = now_ms();
    for (int s = 0; s < nsub; s++) {  // per-kernel GPU time (HIP events on the launch streams)
      float ms = 0;
      for (int i = 0; i < 3; i++)
        if (hipEventElapsedTime(&ms, ev[s][i], ev[s][i + 1]) == hipSuccess) kernMs[i] += ms;
      if (subSel[s] > 0 && hipEventElapsedTime(&ms, ev[s][4], ev[s][5]) == hipSuccess) kernMs[3] += ms;
      kernBatches++;
    }
    kernFrames += nframes;
    // ---- assemble cv::KeyPoint-compatible outputs (ORBextractor.cc:879-889, 959-967) --------
    int status = ORBFE_OK;
    for (int f = 0; f < nframes; f++) {
      const int b = frameKpBase[f], n = frameKpCount[f];
      n_out[f] = n;
      if (n > cap) {
        set_err("frame %d produced %d keypoints, cap is %d", f, n, cap);
        status = ORBFE_ERR_OVERFLOW;
      }
      const int m = n < cap ? n : cap;
      OrbfeKeyPoint* ko = kps + (size_t)f * cap;
      for (int i = 0; i < m; i++) {
        const Meta& mt = meta[b + i];
        OrbfeKeyPoint kp;
        kp.x = (float)mt.x;
        kp.y = (float)mt.y;
        if (mt.level != 0) {
          const float scale = sf[mt.level];
          kp.x *= scale;
          kp.y *= scale;
        }
        kp.size = (float)(int)(31 * sf[mt.level]);
        kp.angle = h_angle.p[b + i];
        kp.response = (float)mt.score;
        kp.octave = mt.level;
        kp.class_id = -1;
        ko[i] = kp;
      }
      if (m > 0) memcpy(desc + (size_t)f * cap * 32, h_desc.p + (size_t)b * 32, (size_t)m * 32);
    }
    const double t5 = now_ms();
    stageMs[0] = (float)tWait;          // waiting for stage 1 + candidate D2H
    stageMs[1] = (float)tHostQt;        // host quadtrees + selection packing
    stageMs[2] = (float)(t4 - t3);      // tail wait for stage 2
    stageMs[3] = (float)(t5 - t4);      // output assembly
    stageMs[4] = (float)(t5 - t0);
    lastFrames = nframes;
    lastGpuQt = false;
    lastLocalCand = false;
    return status;
  }
};

extern "C" {

const char* orbfe_last_error(void) { return g_err.c_str(); }

}  // extern "C" (reopened below)

namespace orbfe {
// orbfe_frame_create_from_extract (orbfe_frame.hip): where frame `frame` of the last collected batch lies.
int extractor_view(orbfe_extractor* h, int frame, ExtractView* out) {
  if (!h || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  if (h->pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  if (frame < 0 || frame >= h->lastFrames) { set_err("frame %d is not part of the last collected batch (%d frames)", frame, h->lastFrames); return ORBFE_ERR_INVALID; }
  if (!h->lastGpuQt) {
    set_err("the last batch took the host-quadtree route, whose selections are not kept per slot: build the frame with orbfe_frame_create");
    return ORBFE_ERR_INVALID;
  }
  const size_t base = (size_t)frame * h->selPerFrame;
  const bool z = h->lastZeroCopy;
  out->sel = (z ? h->h_sel.p : h->d_sel.p) + base;
  out->angle = (z ? h->h_angle.p : h->d_angle.p) + base;
  out->desc = (z ? h->h_desc.p : h->d_desc.p) + base * 32;
  out->nlevels = h->nlevels;
  out->n = 0;
  for (int l = 0; l <= h->nlevels; l++) out->selOff[l] = h->selOff[l];
  for (int l = 0; l < h->nlevels; l++) {
    out->count[l] = (int)h->h_selCount.p[(size_t)frame * kMaxLevels + l];
    out->sf[l] = h->sf[l];
    out->n += out->count[l];
  }
  out->device = h->device;
  out->stream = (void*)h->streams[0];
  return ORBFE_OK;
}
}  // namespace orbfe

extern "C" {
int orbfe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int orbfe_device_malloc(int device_id, size_t bytes, void** out) {
  if (!out || bytes == 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipMalloc(out, bytes));
  return ORBFE_OK;
}
int orbfe_device_free(int device_id, void* ptr) {
  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipFree(ptr));
  return ORBFE_OK;
}
int orbfe_device_upload(int device_id, void* dst_device, const void* src_host, size_t bytes) {
  if (!dst_device || !src_host) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
  return ORBFE_OK;
}
int orbfe_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
  return ORBFE_OK;
}
int orbfe_host_free(void* ptr) {
  HIP_TRY(hipHostFree(ptr));
  return ORBFE_OK;
}
int orbfe_host_register(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  return ORBFE_OK;
}
int orbfe_host_unregister(void* ptr) {
  if (!ptr) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipHostUnregister(ptr));
  return ORBFE_OK;
}
int orbfe_device_synchronize(int device_id) {
  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipDeviceSynchronize());
  return ORBFE_OK;
}

// NUMA placement (with one stream runner per GPU the expected limiter is the host side -- worker
// threads and page-locked buffers on the wrong socket halve the PCIe rate).  The node comes from sysfs via the
// device's PCI address; -1 when the platform does not say (single-socket boxes, containers without sysfs).
int orbfe_device_numa_node(int device_id) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, device_id) != hipSuccess) { (void)hipGetLastError(); return -1; }
  for (char* c = bdf; *c; c++) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

int orbfe_bind_thread_to_device(int device_id) {
  const int node = orbfe_device_numa_node(device_id);
  if (node < 0) return 0;
  char path[96];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  char list[4096] = {0};
  const bool ok = fgets(list, sizeof list, f) != nullptr;
  fclose(f);
  if (!ok) return 0;
  cpu_set_t cur, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof cur, &cur) != 0) return 0;
  int ncpu = 0;
  for (char* p = list; *p;) {   // "0-31,64-95"
    char* e;
    long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++)
      if (CPU_ISSET(c, &cur)) { CPU_SET(c, &want); ncpu++; }   // never widen the mask the process was given
    p = (*e == ',') ? e + 1 : e;
    if (*e != ',' ) break;
  }
  if (ncpu == 0) return 0;
  if (sched_setaffinity(0, sizeof want, &want) != 0) return 0;
  return ncpu;
}

// Link-rate probe for the bench's PCIe-inclusive leg: `reps` back-to-back host-to-device copies of `bytes` from the
// caller's (page-locked) buffer on a private stream, timed with HIP events.
int orbfe_debug_h2d_rate(int device_id, const void* host, size_t bytes, int reps, double* gb_per_s) {
  if (!host || !bytes || reps < 1 || !gb_per_s) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(device_id));
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, bytes));
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = ORBFE_OK;
  float ms = 0;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess) { set_err("stream/event creation failed"); rc = ORBFE_ERR_HIP; }
  if (rc == ORBFE_OK) {
    bool ok = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    ok = ok && hipEventRecord(e0, st) == hipSuccess;
    for (int i = 0; i < reps && ok; i++) ok = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (!ok) { set_err("h2d probe failed: %s", hipGetErrorString(hipGetLastError())); rc = ORBFE_ERR_HIP; }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  (void)hipFree(d);
  if (rc == ORBFE_OK) *gb_per_s = ms > 0 ? (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
  return rc;
}

int orbfe_extractor_set_wait_mode(orbfe_extractor* h, int poll_us) {
  if (!h || poll_us < 0) { set_err("bad wait mode"); return ORBFE_ERR_INVALID; }
  if (h->pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  h->pollWaitUs = poll_us;
  return ORBFE_OK;
}

int orbfe_extractor_set_blur_variant(orbfe_extractor* h, int variant) {
  if (!h || (variant != ORBFE_GAUSS_ED && variant != ORBFE_GAUSS_ROUNDED)) { set_err("bad blur variant"); return ORBFE_ERR_INVALID; }
  if (h->pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  h->gaussVariant = variant;
  return ORBFE_OK;
}

int orbfe_extractor_set_input_format(orbfe_extractor* h, int format, int gray_variant) {
  if (!h || format < ORBFE_INPUT_GRAY8 || format > ORBFE_INPUT_BGRA8 || (gray_variant != ORBFE_GRAY_Q15 && gray_variant != ORBFE_GRAY_Q14)) {
    set_err("bad input format");
    return ORBFE_ERR_INVALID;
  }
  if (h->pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  h->inFormat = format;
  h->grayVariant = gray_variant;
  h->batchCap = 0;   // buffers are re-sized on the next call
  return ORBFE_OK;
}

int orbfe_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device_id,
                           orbfe_extractor** out) {
  if (!out) { set_err("out is NULL"); return ORBFE_ERR_INVALID; }
  *out = nullptr;
  if (nfeatures <= 0 || nlevels < 1 || nlevels > kMaxLevels || !(scaleFactor > 1.0f) || iniThFAST < 1 ||
      minThFAST < 1 || iniThFAST > 255 || minThFAST > 255) {
    set_err("invalid extractor parameters (nfeatures=%d scale=%g nlevels=%d ini=%d min=%d; nlevels<=%d)", nfeatures,
            scaleFactor, nlevels, iniThFAST, minThFAST, kMaxLevels);
    return ORBFE_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    set_err("no usable HIP device (count=%d, requested %d): this library has no CPU fallback", ndev, device_id);
    return ORBFE_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device_id));
  orbfe_extractor* h = new orbfe_extractor();
  h->nfeatures = nfeatures;
  h->scaleFactor = scaleFactor;
  h->nlevels = nlevels;
  h->iniTh = iniThFAST;
  h->minTh = minThFAST;
  h->device = device_id;
  // scale tables and per-level quotas, ORBextractor.cc:447-478
  h->sf.resize(nlevels);
  h->sigma2.resize(nlevels);
  h->sf[0] = 1.0f;
  h->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    h->sf[i] = (float)(h->sf[i - 1] * h->scaleFactor);
    h->sigma2[i] = h->sf[i] * h->sf[i];
  }
  h->isf.resize(nlevels);
  h->isigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    h->isf[i] = 1.0f / h->sf[i];
    h->isigma2[i] = 1.0f / h->sigma2[i];
  }
  h->nfeat.resize(nlevels);
  const float factor = (float)(1.0f / h->scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    h->nfeat[l] = cv_round_f(nDesired);
    sum += h->nfeat[l];
    nDesired *= factor;
  }
  h->nfeat[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  // One HIP stream per handle.  The runtime multiplexes streams onto a few hardware queues (4 by default) and two
  // streams on one queue serialise, so nothing is created that the chosen path does not use: the extra streams of the
  // host-quadtree path appear on its first call (ensureSubStreams), the quadtree stream only with ORBFE_QT_STREAM=1.
  {
    hipError_t e = hipStreamCreateWithFlags(&h->streams[0], hipStreamNonBlocking);
    if (e != hipSuccess) {
      set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
      delete h;
      return ORBFE_ERR_HIP;
    }
  }
  h->stream = h->streams[0];
  bool evOk = hipEventCreate(&h->evFrame0) == hipSuccess &&
              hipEventCreateWithFlags(&h->evUpload, hipEventDisableTiming) == hipSuccess;
  for (auto& e : h->evS1) evOk = evOk && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  for (auto& es : h->ev) for (auto& e : es) evOk = evOk && hipEventCreate(&e) == hipSuccess;
  if (!evOk) { set_err("hipEventCreate failed"); delete h; return ORBFE_ERR_HIP; }
  for (auto& e : h->evQt) evOk = evOk && hipEventCreate(&e) == hipSuccess;
  if (!evOk) { set_err("hipEventCreate failed"); delete h; return ORBFE_ERR_HIP; }
  if (const char* hv = getenv("ORBFE_HOST_QUADTREE")) h->gpuQuadtree = atoi(hv) == 0;
  if (const char* pv = getenv("ORBFE_FAST_PAIRS")) h->pairCells = atoi(pv) != 0;
  if (const char* cv = getenv("ORBFE_CONE_MAX_FRAMES")) h->coneMaxFrames = atoi(cv);
  if (const char* pv = getenv("ORBFE_POLL_WAIT_US")) h->pollWaitUs = atoi(pv);
  if (const char* dw = ORBFE_EXP_ENV("ORBFE_DESCRIBE_WAVES")) h->describe4 = atoi(dw) != 1;
  if (const char* zv = getenv("ORBFE_ZERO_COPY")) { h->zeroCopyOut = atoi(zv) != 0; h->zeroCopyBatches = atoi(zv) == 2; }
  if (const char* iv = getenv("ORBFE_INGEST_KERNEL")) h->ingestKernel = atoi(iv) != 0;
  if (const char* gv = ORBFE_EXP_ENV("ORBFE_LOCAL_LISTS")) h->localLists = atoi(gv) != 0;
  if (const char* qv = getenv("ORBFE_QT_LDS_BYTES")) h->qtLdsBudget = atoi(qv);
  if (const char* qj = getenv("ORBFE_QT_JUMP")) h->qtJump = atoi(qj) != 0;
  if (const char* tb = ORBFE_EXP_ENV("ORBFE_TAIL_CONE_BASE")) h->tailBase = atoi(tb);
  if (const char* tt = ORBFE_EXP_ENV("ORBFE_TAIL_CONE_TILE")) h->tailTile = std::max(8, atoi(tt)) & ~3;
  if (const char* cv = ORBFE_EXP_ENV("ORBFE_CONE_TILE")) h->coneTile = atoi(cv) <= 0 ? 0 : std::max(8, atoi(cv));
  if (const char* pv = ORBFE_EXP_ENV("ORBFE_PROFILE_KERNELS")) h->profileKernels = atoi(pv) != 0;
  if (const char* gv = getenv("ORBFE_GAUSS_VARIANT")) h->gaussVariant = atoi(gv) == ORBFE_GAUSS_ROUNDED ? ORBFE_GAUSS_ROUNDED : ORBFE_GAUSS_ED;
  h->selPerFrame = 0;
  for (int l = 0; l < nlevels; l++) {
    h->selOff[l] = h->selPerFrame;
    // a level returns at most max(N_l + 2, 4 * roots) keypoints: the first sweep of DistributeOctTree divides every root
    // before any size check (ORBextractor.cc:620-700), later ones overshoot N_l by at most 2; roots <= 4 on this path
    h->selPerFrame += std::max(h->nfeat[l] + 4, 16);
    // k_quadtree3 holds at most kQtNodeCap - 4 features per level; beyond that the handle uses the host quadtree
    // (blocking orbfe_extract / orbfe_extract_batch only, like ORBFE_HOST_QUADTREE=1)
    if (h->nfeat[l] + 4 > kQtNodeCap) h->gpuQuadtree = false;
  }
  h->selOff[nlevels] = h->selPerFrame;
  // host workers for the per-(frame, level) quadtrees: ORBFE_HOST_THREADS, default min(cores, 16)
  int nthreads = (int)std::thread::hardware_concurrency();
  if (nthreads > 16) nthreads = 16;
  if (const char* ev = getenv("ORBFE_HOST_THREADS")) nthreads = atoi(ev);
  if (nthreads < 1) nthreads = 1;
  h->hostThreads = nthreads;   // the pool itself is created on first use (host-quadtree path only)
  *out = h;
  return ORBFE_OK;
}

void orbfe_extractor_destroy(orbfe_extractor* h) { delete h; }
int orbfe_extractor_levels(const orbfe_extractor* h) { return h ? h->nlevels : 0; }
int orbfe_extractor_device(const orbfe_extractor* h) { return h ? h->device : -1; }

// How many of the first n handles' streams run SIDE BY SIDE?  The HIP runtime folds a process's streams onto GPU_MAX_HW_QUEUES hardware
// queues (4 unless the variable said otherwise before the runtime started) and two streams on one queue serialise -- a property of the
// process no API reports, so it is measured: a 200 us do-nothing kernel on each of the first k streams finishes in about 200 us when the k
// streams have queues of their own and in 400 us when two of them share one.  Returns the largest k <= n for which they all overlap
// (at least 1), or a negative error code.  About a millisecond, once per runner (orbfe_stream_create).
extern "C++" int orbfe_concurrent_streams(orbfe_extractor* const* hs, int n) {
  if (!hs || n < 1) return ORBFE_ERR_INVALID;
  HIP_TRY(hipSetDevice(hs[0]->device));
  constexpr unsigned long long kTicks = 20000;   // 200 us of the 100 MHz reference clock
  for (int i = 0; i < n; i++) launch_spin(100, hs[i]->streams[0]);   // (code object loaded, queues created)
  for (int i = 0; i < n; i++) HIP_TRY(hipStreamSynchronize(hs[i]->streams[0]));
  // do the first k streams run side by side?
  auto overlap = [&](int k, bool& yes) -> int {
    double best = 1e30;
    for (int rep = 0; rep < 3 && best >= 0.2 * 1.6; rep++) {   // (the best of up to three: a delayed wake-up of this thread must not read as a shared queue)
      const double t0 = now_ms();
      for (int i = 0; i < k; i++) launch_spin(kTicks, hs[i]->streams[0]);
      HIP_TRY(hipGetLastError());
      for (int i = 0; i < k; i++) HIP_TRY(hipStreamSynchronize(hs[i]->streams[0]));
      best = std::min(best, now_ms() - t0);
    }
    yes = best < 0.2 * 1.6;   // (two on one queue: >= 0.4 ms)
    return ORBFE_OK;
  };
  // Grow the overlapping prefix one stream at a time.  Which hardware queue a stream lands on depends on everything the process
  // created before it, so a stream that shares its queue with an earlier one of ours is simply replaced by a fresh one -- a few times --
  // before the prefix is declared final: with enough hardware queues every runner ends up with `n` streams of their own whatever the
  // process did before.
  int k = 1;
  while (k < n) {
    bool yes = false;
    int rc = overlap(k + 1, yes);
    if (rc) return rc;
    for (int attempt = 0; !yes && attempt < 6; attempt++) {
      orbfe_extractor* h = hs[k];
      hipStream_t fresh = nullptr;
      HIP_TRY(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
      HIP_TRY(hipStreamSynchronize(h->streams[0]));
      (void)hipStreamDestroy(h->streams[0]);
      h->streams[0] = fresh;
      h->stream = fresh;
      launch_spin(100, fresh);
      HIP_TRY(hipStreamSynchronize(fresh));
      if ((rc = overlap(k + 1, yes))) return rc;
    }
    if (!yes) break;
    k++;
  }
  return k;
}
extern "C++" { namespace orbfe { int fast_stamps(unsigned long long out[8], int reset); } }
int orbfe_debug_fast_stamps(orbfe_extractor* h, unsigned long long out[8], int reset) {
  if (!h || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
#ifndef ORBFE_EXPERIMENTS
  set_err("orbfe_debug_fast_stamps needs a library built with make EXPERIMENTS=1");
  return ORBFE_ERR_INVALID;
#endif
  if (orbfe::fast_stamps(out, reset)) { set_err("reading the stamp counters failed"); return ORBFE_ERR_HIP; }
  return ORBFE_OK;
}
extern "C++" { namespace orbfe { int sfi_debug_read(int* out, int capRecords, int reset); } }
int orbfe_debug_sfi_records(orbfe_extractor* h, int32_t* out, int cap_records, int* n_out, int reset) {
  if (!h || !out || !n_out || cap_records < 0) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const int n = orbfe::sfi_debug_read(out, cap_records, reset);
  if (n < 0) { set_err("reading the SearchForInitialization records failed"); return ORBFE_ERR_HIP; }
  *n_out = n;
  return ORBFE_OK;
}
float orbfe_extractor_scale_factor(const orbfe_extractor* h) { return h ? (float)h->scaleFactor : 0.f; }
int orbfe_extractor_scale_tables(const orbfe_extractor* h, float* a, float* b, float* c, float* d) {
  if (!h) { set_err("handle is NULL"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < h->nlevels; i++) {
    if (a) a[i] = h->sf[i];
    if (b) b[i] = h->isf[i];
    if (c) c[i] = h->sigma2[i];
    if (d) d[i] = h->isigma2[i];
  }
  return ORBFE_OK;
}
int orbfe_extractor_features_per_level(const orbfe_extractor* h, int32_t* out) {
  if (!h || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < h->nlevels; i++) out[i] = h->nfeat[i];
  return ORBFE_OK;
}
int orbfe_extractor_max_keypoints(const orbfe_extractor* h) {
  if (!h) return 0;
  int n = 0;   // per level max(N_l + 2, 4 * roots) with up to 4 roots (aspect ratio < 4.5); == nfeatures + 2 * nlevels when N_l >= 14
  for (int l = 0; l < h->nlevels; l++) n += std::max(h->nfeat[l] + 2, 16);
  return n;
}

int orbfe_extractor_set_vocabulary(orbfe_extractor* h, orbfe_vocabulary* v, int levelsup) {
  if (!h) { set_err("extractor is NULL"); return ORBFE_ERR_INVALID; }
  if (h->pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  if (v && bow_device(v) != h->device) { set_err("vocabulary lives on device %d, extractor on %d", bow_device(v), h->device); return ORBFE_ERR_INVALID; }
  if (v && !h->gpuQuadtree) { set_err("the fused bag-of-words path needs the GPU quadtree path"); return ORBFE_ERR_INVALID; }
  h->voc = v;
  h->vocLevelsup = levelsup;
  return ORBFE_OK;
}

int orbfe_extract_bow(orbfe_extractor* h, int frame, uint32_t* bow_ids, double* bow_values, int* n_words, uint32_t* fv_nodes,
                      uint32_t* fv_offsets, uint32_t* fv_features, int* n_fv_nodes, uint32_t* word_of_feature,
                      uint32_t* node_of_feature) {
  if (!h || !bow_ids || !bow_values || !n_words || !fv_nodes || !fv_offsets || !fv_features || !n_fv_nodes) {
    set_err("bad argument");
    return ORBFE_ERR_INVALID;
  }
  if (!h->voc || frame < 0 || frame >= (int)h->bowKp.size()) {
    set_err("no bag-of-words result for frame %d (set a vocabulary before extracting)", frame);
    return ORBFE_ERR_INVALID;
  }
  const std::vector<uint2>& ln = h->bowKp[frame];
  return bow_assemble(h->voc, ln.data(), (int)ln.size(), bow_ids, bow_values, n_words, fv_nodes, fv_offsets, fv_features,
                      n_fv_nodes, word_of_feature, node_of_feature);
}

int orbfe_extract_bow_raw(orbfe_extractor* h, int frame, uint32_t* leaf_node, uint32_t* level_node, int cap, int* n_out) {
  if (!h || !leaf_node || !level_node || !n_out || cap < 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  if (!h->voc || frame < 0 || frame >= (int)h->bowKp.size()) {
    set_err("no bag-of-words result for frame %d (set a vocabulary before extracting)", frame);
    return ORBFE_ERR_INVALID;
  }
  const std::vector<uint2>& ln = h->bowKp[frame];
  *n_out = (int)ln.size();
  const int m = std::min(cap, (int)ln.size());
  for (int i = 0; i < m; i++) { leaf_node[i] = ln[i].x; level_node[i] = ln[i].y; }
  return ORBFE_OK;
}

int orbfe_extractor_max_keypoints_for_size(const orbfe_extractor* h, int rows, int cols) {
  if (!h || rows <= 0 || cols <= 0) return 0;
  int n = 0;
  for (int l = 0; l < h->nlevels; l++) {
    const int w = cv_round_f((float)cols * h->isf[l]), hh = cv_round_f((float)rows * h->isf[l]);
    int roots = hh > 2 * kBorder ? (int)roundf(static_cast<float>(w - 2 * kBorder) / (hh - 2 * kBorder)) : 0;
    if (roots < 0) roots = 0;
    n += std::max(h->nfeat[l] + 2, 4 * roots);
  }
  return n;
}

int orbfe_extract_batch(orbfe_extractor* h, int nframes, const uint8_t* const* gray, int in_device_memory, int rows,
                        int cols, size_t stride_bytes, OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out) {
  if (!h || !n_out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  if (nframes <= 0) return ORBFE_OK;
  for (int f = 0; f < nframes; f++) n_out[f] = 0;
  if (rows == 0 || cols == 0 || !gray) return ORBFE_OK;  // empty image: silent return (ORBextractor.cc:910-911)
  if (rows < 0 || cols < 0 || stride_bytes < (size_t)cols * h->inChannels() || !kps || !desc || cap <= 0) {
    set_err("invalid image / output arguments");
    return ORBFE_ERR_INVALID;
  }
  // a blocking call between _submit and _collect would re-carve the arenas the pending batch still writes
  if (h->pendingFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  {
    HIP_TRY(hipSetDevice(h->device));
    const int rc = h->setGeometry(rows, cols);
    if (rc) return rc;
  }
  if (h->gpuQuadtree && h->geomGpuQtOk)
    return h->runGpuQt(nframes, gray, in_device_memory != 0, rows, cols, stride_bytes, kps, desc, cap, n_out);
  return h->run(nframes, gray, in_device_memory != 0, rows, cols, stride_bytes, kps, desc, cap, n_out);
}

// The GPU quadtree takes 1..4 root nodes per level and at most 2 044 features per level; outside that (very wide strips,
// nfeatures beyond about 9 000) the asynchronous entry points run the batch through the blocking host-quadtree path at
// submit time and keep the results for _collect: same results, no overlap.
static bool gpu_route_ok(orbfe_extractor* h, int rows, int cols, int* rc) {
  *rc = ORBFE_OK;
  if (!h->gpuQuadtree) return false;
  if (hipSetDevice(h->device) != hipSuccess) { set_err("hipSetDevice failed"); *rc = ORBFE_ERR_HIP; return false; }
  if (h->pendingFrames || h->deferredFrames) return true;   // reported by the submit itself
  *rc = h->setGeometry(rows, cols);
  return *rc == ORBFE_OK && h->geomGpuQtOk;
}

static int deferred_extract(orbfe_extractor* h, int nframes, const uint8_t* const* gray, int in_device_memory, int rows, int cols,
                            size_t stride_bytes) {
  if (h->pendingFrames || h->deferredFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  const int cap = orbfe_extractor_max_keypoints_for_size(h, rows, cols);
  h->deferredCap = cap;
  h->defKps.resize((size_t)nframes * cap);
  h->defDesc.resize((size_t)nframes * cap * 32);
  h->defN.assign(nframes, 0);
  const int rc = orbfe_extract_batch(h, nframes, gray, in_device_memory, rows, cols, stride_bytes, h->defKps.data(), h->defDesc.data(),
                                     cap, h->defN.data());
  if (rc) return rc;
  h->deferredFrames = nframes;
  h->deferredMatched = false;
  return ORBFE_OK;
}

static int deferred_collect(orbfe_extractor* h, OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out, int32_t* matches12,
                            int* nmatches) {
  const int nframes = h->deferredFrames, dcap = h->deferredCap;
  h->deferredFrames = 0;
  int status = ORBFE_OK;
  for (int f = 0; f < nframes; f++) {
    const int n = h->defN[f];
    n_out[f] = n;
    if (n > cap) { set_err("frame %d produced %d keypoints, cap is %d", f, n, cap); status = ORBFE_ERR_OVERFLOW; }
    const int m = std::min(n, cap);
    memcpy(kps + (size_t)f * cap, h->defKps.data() + (size_t)f * dcap, sizeof(OrbfeKeyPoint) * (size_t)m);
    memcpy(desc + (size_t)f * cap * 32, h->defDesc.data() + (size_t)f * dcap * 32, 32 * (size_t)m);
    if (matches12 && nmatches) {
      int32_t* row = matches12 + (size_t)f * cap;
      for (int i = 0; i < cap; i++) row[i] = -1;
      nmatches[f] = 0;
      if (h->deferredMatched) {
        memcpy(row, h->defM12.data() + (size_t)f * dcap, sizeof(int32_t) * (size_t)std::min(cap, dcap));
        nmatches[f] = h->defNm[f];
      }
    }
  }
  return status;
}

int orbfe_extract_batch_submit(orbfe_extractor* h, int nframes, const uint8_t* const* gray, int in_device_memory,
                               int rows, int cols, size_t stride_bytes) {
  if (!h || !gray || nframes <= 0 || rows <= 0 || cols <= 0 || stride_bytes < (size_t)cols * h->inChannels()) {
    set_err("invalid arguments");
    return ORBFE_ERR_INVALID;
  }
  int rc;
  if (!gpu_route_ok(h, rows, cols, &rc)) {
    if (rc) return rc;
    return deferred_extract(h, nframes, gray, in_device_memory, rows, cols, stride_bytes);
  }
  if (h->deferredFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  return h->submitGpuQt(nframes, gray, in_device_memory != 0, rows, cols, stride_bytes);
}

int orbfe_extract_batch_wait(orbfe_extractor* h) {
  if (!h) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  if (h->deferredFrames) return ORBFE_OK;   // (a batch that took the blocking host-quadtree route at submit time is done already)
  return h->waitOnly();
}

int orbfe_extract_batch_collect(orbfe_extractor* h, OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out) {
  if (!h || !kps || !desc || !n_out || cap <= 0) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  if (h->deferredFrames) return deferred_collect(h, kps, desc, cap, n_out, nullptr, nullptr);
  return h->waitGpuQt(kps, desc, cap, n_out);
}

int orbfe_sfi_chain_create(const orbfe_extractor* h, orbfe_sfi_chain** out) {
  if (!h || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  *out = nullptr;
  HIP_TRY(hipSetDevice(h->device));
  orbfe_sfi_chain* c = new orbfe_sfi_chain();
  c->device = h->device;
  c->n0cap = h->selOff[1] - h->selOff[0];
  int rc = ORBFE_OK;
  for (int i = 0; i < 2 && rc == ORBFE_OK; i++) {
    if ((rc = c->sel[i].ensure(c->n0cap))) break;
    if ((rc = c->angle[i].ensure(c->n0cap))) break;
    if ((rc = c->desc[i].ensure((size_t)c->n0cap * 32))) break;
    if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); rc = ORBFE_ERR_HIP; }
  }
  if (rc == ORBFE_OK) rc = c->count.ensure(4);
  if (rc == ORBFE_OK) {
    const uint32_t init[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (hipMemcpy(c->count.p, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) { set_err("hipMemcpy failed"); rc = ORBFE_ERR_HIP; }
  }
  if (rc != ORBFE_OK) { delete c; return rc; }
  *out = c;
  return ORBFE_OK;
}

void orbfe_sfi_chain_destroy(orbfe_sfi_chain* c) { delete c; }

int orbfe_sfi_chain_set_isolated(orbfe_sfi_chain* c, int isolated) {
  if (!c) { set_err("chain is NULL"); return ORBFE_ERR_INVALID; }
  c->isolated = isolated != 0;
  if (c->isolated) c->hostPrevValid = false;
  return ORBFE_OK;
}

int orbfe_extract_batch_submit_matched(orbfe_extractor* h, orbfe_sfi_chain* chain, int nframes, const uint8_t* const* gray,
                                       int in_device_memory, int rows, int cols, size_t stride_bytes, const float bounds[4],
                                       int window_size, float nnratio, int check_orientation) {
  if (!h || !chain || !gray || !bounds || nframes <= 0 || rows <= 0 || cols <= 0 || stride_bytes < (size_t)cols * h->inChannels() || window_size <= 0) {
    set_err("invalid arguments");
    return ORBFE_ERR_INVALID;
  }
  int rc;
  if (!gpu_route_ok(h, rows, cols, &rc)) {
    if (rc) return rc;
    // host-quadtree route: extract now, then SearchForInitialization of every frame against its predecessor (the chain
    // carries the previous batch's last frame on the host) as one batched matcher call
    if ((rc = deferred_extract(h, nframes, gray, in_device_memory, rows, cols, stride_bytes))) return rc;
    if (!chain->hostMatcher && (rc = orbfe_matcher_create(h->device, &chain->hostMatcher))) { h->deferredFrames = 0; return rc; }
    const int cap = h->deferredCap;
    h->defM12.assign((size_t)nframes * cap, -1);
    h->defNm.assign(nframes, 0);
    std::vector<const OrbfeKeyPoint*> k1, k2;
    std::vector<const uint8_t*> d1, d2;
    std::vector<int> n1, n2, frameOf, nm;
    std::vector<std::vector<float>> pxy;
    std::vector<float*> prev;
    std::vector<int32_t*> m12;
    for (int f = 0; f < nframes; f++) {
      const OrbfeKeyPoint* pk;
      const uint8_t* pd;
      int pn;
      if (f == 0) {
        if (!chain->hostPrevValid || chain->isolated) continue;   // very first frame of the stream (or isolated batches): no predecessor
        pk = chain->hostPrevKps.data(); pd = chain->hostPrevDesc.data(); pn = (int)chain->hostPrevKps.size();
      } else {
        pk = h->defKps.data() + (size_t)(f - 1) * cap; pd = h->defDesc.data() + (size_t)(f - 1) * cap * 32; pn = h->defN[f - 1];
      }
      pxy.emplace_back((size_t)std::max(pn, 1) * 2);
      for (int j = 0; j < pn; j++) { pxy.back()[2 * j] = pk[j].x; pxy.back()[2 * j + 1] = pk[j].y; }   // vbPrevMatched := F1's keypoints (Tracking.cc:355-357)
      k1.push_back(pk); d1.push_back(pd); n1.push_back(pn);
      k2.push_back(h->defKps.data() + (size_t)f * cap); d2.push_back(h->defDesc.data() + (size_t)f * cap * 32); n2.push_back(h->defN[f]);
      m12.push_back(h->defM12.data() + (size_t)f * cap);
      frameOf.push_back(f);
    }
    for (auto& v : pxy) prev.push_back(v.data());
    nm.assign(k1.size(), 0);
    if (!k1.empty()) {
      rc = orbfe_search_for_initialization_batch(chain->hostMatcher, (int)k1.size(), k1.data(), d1.data(), n1.data(), k2.data(), d2.data(),
                                                 n2.data(), bounds, prev.data(), m12.data(), window_size, nnratio, check_orientation,
                                                 nm.data());
      if (rc) { h->deferredFrames = 0; return rc; }
      for (size_t p = 0; p < frameOf.size(); p++) h->defNm[frameOf[p]] = nm[p];
    }
    const int last = nframes - 1, ln = h->defN[last];
    chain->hostPrevKps.assign(h->defKps.begin() + (size_t)last * cap, h->defKps.begin() + (size_t)last * cap + ln);
    chain->hostPrevDesc.assign(h->defDesc.begin() + (size_t)last * cap * 32, h->defDesc.begin() + ((size_t)last * cap + ln) * 32);
    chain->hostPrevValid = !chain->isolated;
    h->deferredMatched = true;
    return ORBFE_OK;
  }
  if (h->deferredFrames) { set_err("a submitted batch has not been collected yet"); return ORBFE_ERR_INVALID; }
  orbfe_extractor::MatchSpec ms;
  ms.chain = chain;
  memcpy(ms.bounds, bounds, sizeof ms.bounds);
  ms.window = window_size;
  ms.nnratio = nnratio;
  ms.checkOri = check_orientation;
  return h->submitGpuQt(nframes, gray, in_device_memory != 0, rows, cols, stride_bytes, &ms);
}

int orbfe_extract_batch_collect_matched(orbfe_extractor* h, OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out,
                                        int32_t* matches12, int* nmatches) {
  if (!h || !kps || !desc || !n_out || !matches12 || !nmatches || cap <= 0) { set_err("invalid arguments"); return ORBFE_ERR_INVALID; }
  if (h->deferredFrames) return deferred_collect(h, kps, desc, cap, n_out, matches12, nmatches);
  return h->waitGpuQt(kps, desc, cap, n_out, matches12, nmatches);
}

int orbfe_extract(orbfe_extractor* h, const uint8_t* gray, int rows, int cols, size_t stride_bytes, OrbfeKeyPoint* kps,
                  uint8_t* desc, int cap, int* n_out) {
  if (!n_out) { set_err("n_out is NULL"); return ORBFE_ERR_INVALID; }
  *n_out = 0;
  if (!gray || rows == 0 || cols == 0) return ORBFE_OK;
  const uint8_t* frames[1] = {gray};
  return orbfe_extract_batch(h, 1, frames, 0, rows, cols, stride_bytes, kps, desc, cap, n_out);
}

int orbfe_debug_level_size(const orbfe_extractor* h, int level, int* w, int* hgt) {
  if (!h || level < 0 || level >= h->nlevels || h->rows == 0) { set_err("bad level / no frame yet"); return ORBFE_ERR_INVALID; }
  *w = h->P.lv[level].w;
  *hgt = h->P.lv[level].h;
  return ORBFE_OK;
}

int orbfe_debug_level_copy(orbfe_extractor* h, int frame, int level, uint8_t* out) {
  if (!h || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastFrames) {
    set_err("bad frame/level");
    return ORBFE_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(h->device));
  const LevelGeom& L = h->P.lv[level];
  const uint8_t* src;
  size_t pitch;
  if (level == 0) {
    src = h->h_frame0.p[frame];
    pitch = (size_t)h->P.stride0;
  } else {
    src = h->d_slab.p + (size_t)h->P.slabBytes * frame + L.off;
    pitch = L.pitch;
  }
  HIP_TRY(hipMemcpy2D(out, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return ORBFE_OK;
}

int orbfe_debug_candidates(orbfe_extractor* h, int frame, int level, int32_t* xys, int cap, int* n_out) {
  if (!h || !n_out || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastFrames) {
    set_err("bad frame/level");
    return ORBFE_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(h->device));
  // read from the device: small batches hand their results over without copying the whole arena
  uint32_t ls[kMaxLevels + 1];
  HIP_TRY(hipMemcpy(ls, h->d_levelStart.p + (size_t)frame * (kMaxLevels + 1), sizeof(ls), hipMemcpyDeviceToHost));
  // packed list with prefix offsets (k_compact), or level-local lists with their lengths (k_compact_local: the latency route)
  const size_t start = h->lastLocalCand ? (size_t)h->P.lv[level].slotBase : (size_t)ls[level];
  const int n = h->lastLocalCand ? (int)ls[level] : (int)(ls[level + 1] - ls[level]);
  *n_out = n;
  if (n <= 0 || !xys) return ORBFE_OK;
  std::vector<uint32_t> tmp(n);
  HIP_TRY(hipMemcpy(tmp.data(), h->d_cand.p + (size_t)h->P.candCap * frame + start, sizeof(uint32_t) * n,
                    hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap; i++) {
    xys[3 * i] = (int)(tmp[i] & 0xfff) - kBorder;
    xys[3 * i + 1] = (int)((tmp[i] >> 12) & 0xfff) - kBorder;
    xys[3 * i + 2] = (int)(tmp[i] >> 24);
  }
  return ORBFE_OK;
}

int orbfe_debug_stage_ms(const orbfe_extractor* h, float out[5]) {
  if (!h || !out) { set_err("NULL argument"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < 5; i++) out[i] = h->stageMs[i];
  return ORBFE_OK;
}

int orbfe_debug_kernel_ms(orbfe_extractor* h, double out_ms[5], long long* batches, long long* frames, int reset) {
  if (!h) { set_err("handle is NULL"); return ORBFE_ERR_INVALID; }
  for (int i = 0; i < 5; i++) if (out_ms) out_ms[i] = h->kernMs[i];
  if (batches) *batches = h->kernBatches;
  if (frames) *frames = h->kernFrames;
  if (reset) {
    for (double& v : h->kernMs) v = 0;
    h->kernBatches = h->kernFrames = 0;
  }
  return ORBFE_OK;
}

int orbfe_debug_set_profiling(orbfe_extractor* h, int enable) {
  if (!h) { set_err("handle is NULL"); return ORBFE_ERR_INVALID; }
  h->profileKernels = enable != 0;
  return ORBFE_OK;
}

int orbfe_debug_sincos(orbfe_extractor* h, const float* angle_deg, int n, float* cos_out, float* sin_out) {
  if (!h || !angle_deg || !cos_out || !sin_out || n < 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  if (n == 0) return ORBFE_OK;
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  if ((rc = h->d_f32tmp.ensure((size_t)3 * n))) return rc;
  float* d = h->d_f32tmp.p;
  HIP_TRY(hipMemcpyAsync(d, angle_deg, sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
  launch_sincos(d, n, d + n, d + 2 * (size_t)n, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(cos_out, d + n, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(sin_out, d + 2 * (size_t)n, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return ORBFE_OK;
}

int orbfe_debug_quadtree(const int16_t* x, const int16_t* y, const uint8_t* score, int n, int min_x, int max_x,
                         int min_y, int max_y, int n_target, int32_t* out_idx, int cap, int* n_out) {
  if (!n_out || n < 0 || (n && (!x || !y || !score))) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  QuadTree qt;
  std::vector<int> sel;
  qt.distribute(x, y, score, n, min_x, max_x, min_y, max_y, n_target, sel);
  *n_out = (int)sel.size();
  for (int i = 0; i < (int)sel.size() && i < cap; i++) out_idx[i] = sel[i];
  return ORBFE_OK;
}

int orbfe_debug_sincos_host_check(uint32_t lo_bits, uint32_t hi_bits, uint32_t step, long long* mismatches) {
  if (!mismatches || step == 0) { set_err("bad argument"); return ORBFE_ERR_INVALID; }
  long long bad = 0;
  for (uint64_t u = lo_bits; u <= hi_bits; u += step) {
    const uint32_t b = (uint32_t)u;
    float f;
    memcpy(&f, &b, 4);
    volatile float vf = f;
    float s, c;
    sincosf_glibc(f, &s, &c);
    const float s0 = sinf(vf), c0 = cosf(vf);
    if (memcmp(&s, &s0, 4) || memcmp(&c, &c0, 4)) bad++;
  }
  *mismatches = bad;
  return ORBFE_OK;
}

}  // extern "C"
