// orb_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// This file is a plain, scalar, single-threaded CPU restatement of the per-frame ORB feature
// front end of AlejandroSilvestri/os1 (ORB-SLAM2 fork).  It exists only so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP product
// (os1_amd/).  NOTHING under os1_amd/ may include, link, load or call it.
//
// PARITY STATUS: ** parity unpinned **.
//   The reference ships no tests / golden vectors and its pixel arithmetic is
//   split with an un-vendored, un-pinned OpenCV 4 (reference .cproject:40,58-64) that is not
//   present in this image, so the reference cannot be built here and there is nothing of the
//   reference's own to pin this oracle against.  What IS pinned:
//     * everything the reference source spells out itself (cell loop, quadtree, IC-angle,
//       rBRIEF sampling, grid, the searches, Hamming) follows the cited lines one-to-one;
//     * the OpenCV-side primitives (cv::FAST, cv::resize INTER_LINEAR 8u, cv::GaussianBlur 8u
//       7x7 s=2, cv::fastAtan2, cvRound/cvFloor/cvCeil) restate OpenCV 4.x's *published
//       generic (non-IPP, non-FMA) algorithms*; the variant chosen is:
//         - GaussianBlur: 8.8 fixed-point kernel, error-diffused rounding (OpenCV >= 4.1.1):
//           [18,34,48,56,48,34,18]/256, reflect-101, (S + 32768) >> 16 -- the default; variant 1
//           (orc_extractor_set_gauss_variant) is the plainly rounded kernel of OpenCV 4.0.0 - 4.1.0
//           (and 3.4.2 - 3.4.6), [18,34,49,55,49,34,18] (sum 257), with the final saturation its
//           ufixedpoint arithmetic needs: min(255, (S + 32768) >> 16);
//         - resize: INTER_RESIZE_COEF_BITS = 11 fixed-point bilinear;
//         - fastAtan2: 7th-order odd polynomial, float32, no FMA contraction;
//         - cosf/sinf: this image's glibc (2.35) float routines;
//       they are pinned by first-principles known-answer tests in tests/test_oracle_kat.py.
//   Hazard H1: the reference breaks ties between equal-sized quadtree nodes by
//   heap address (ORBextractor.cc:716); this oracle uses node creation order instead
//   (later-created node == "larger pointer").  orc_extractor_set_tie_rule selects the opposite
//   rule, the real addresses of this process's list nodes, or a seeded random order, so that
//   tools/h1_sensitivity.py can MEASURE how much of the output hangs on that tie-break.
//
// Build: g++ -O3 -std=gnu++17 -ffp-contract=off (mirrors the reference's -O3, no arch flags;
// contraction off so x*b+y*a and the atan polynomial round like separate mul/add).
//
// Citations "ORBextractor.cc:NNN" etc. are relative to /root/reference/src.

#include <dlfcn.h>
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <utility>
#include <vector>

extern "C" {
// Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id).
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} OrcKeyPoint;
}

using std::ptrdiff_t;

namespace {

// ---------------------------------------------------------------------------------------------
// OpenCV rounding helpers.  cvRound = round-half-to-even.
// ---------------------------------------------------------------------------------------------
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(float v) { int i = (int)v; return i + (i < v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

struct View {  // borrowed u8 image view
  const uint8_t* p;
  int w, h;
  ptrdiff_t stride;
  const uint8_t* row(int y) const { return p + (ptrdiff_t)y * stride; }
};

struct Image {  // owned, tightly packed
  std::vector<uint8_t> d;
  int w = 0, h = 0;
  void alloc(int W, int H) { w = W; h = H; d.assign((size_t)W * H, 0); }
  View view() const { return View{d.data(), w, h, (ptrdiff_t)w}; }
  uint8_t* row(int y) { return d.data() + (size_t)y * w; }
};

// ---------------------------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1 -- SURVEY.md Appendix B.2.
// Call site: ORBextractor.cc:984.
// ---------------------------------------------------------------------------------------------
void resize_linear_u8(const View& s, uint8_t* dst, int dw, int dh, ptrdiff_t dstride) {
  const int sw = s.w, sh = s.h;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * (size_t)dw), beta(2 * (size_t)dh);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    alpha[2 * dx] = sat_short(cv_round((1.f - fx) * 2048));
    alpha[2 * dx + 1] = sat_short(cv_round(fx * 2048));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = sat_short(cv_round((1.f - fy) * 2048));
    beta[2 * dy + 1] = sat_short(cv_round(fy * 2048));
  }
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  std::vector<int> h0(dw), h1(dw);
  auto hline = [&](const uint8_t* S, std::vector<int>& D) {
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      D[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
    }
    for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
  };
  for (int dy = 0; dy < dh; dy++) {
    const int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hline(s.row(sy0), h0);
    hline(s.row(sy1), h1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst + (ptrdiff_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (h0[x] >> 4)) >> 16) + ((b1 * (h1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(u8, Size(7,7), 2, 2, BORDER_REFLECT_101) -- SURVEY.md Appendix B.3.
// Call site: ORBextractor.cc:950 (on a clone of the level, so the border is the level's own).
// ---------------------------------------------------------------------------------------------
// 8.8 fixed-point kernel by error diffusion from the ends to the centre; centre takes the rest.
void gauss_kernel_fixed_ed(int n, double sigma, int out[]) {
  std::vector<double> k(n);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    k[i] = std::exp(-0.5 * x * x / (sigma * sigma));
    sum += k[i];
  }
  for (int i = 0; i < n; i++) k[i] /= sum;
  double err = 0;
  int isum = 0;
  for (int i = 0; i < n / 2; i++) {
    double adj = k[i] * 256.0 + err;
    int v = cv_round(adj);
    err = adj - v;
    out[i] = out[n - 1 - i] = v;
    isum += v;
  }
  out[n / 2] = 256 - 2 * isum;
}

// Variant 1: OpenCV 4.0.0 - 4.1.0 (3.4.2 - 3.4.6).  getFixedpointGaussianKernel there converts every normalised tap on its own,
// ufixedpoint16(double) = cvRound(v * 256): [18,34,49,55,49,34,18], sum 257.  (Recalled from upstream, like variant 0: unpinned.)
void gauss_kernel_fixed_rounded(int n, double sigma, int out[]) {
  std::vector<double> k(n);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    k[i] = std::exp(-0.5 * x * x / (sigma * sigma));
    sum += k[i];
  }
  for (int i = 0; i < n; i++) out[i] = cv_round(k[i] / sum * 256.0);
}

inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// variant 0: error-diffused taps (sum 256: no intermediate or final saturation can occur).  variant 1: rounded taps (sum 257): the
// horizontal sums reach at most 257 * 255 = 65 535 (still a ufixedpoint16 without saturation), the vertical ones 257 * 65 535 < 2^32,
// and the rounded result can be 256 or 257 over an (almost) white neighbourhood: saturate_cast<uchar> clips it to 255.
void gauss7_u8(const View& s, uint8_t* dst, ptrdiff_t dstride, int variant = 0) {
  int k[7];
  if (variant == 1) gauss_kernel_fixed_rounded(7, 2.0, k);
  else gauss_kernel_fixed_ed(7, 2.0, k);
  const int w = s.w, h = s.h;
  std::vector<uint16_t> hb((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = s.row(y);
    for (int x = 0; x < w; x++) {
      unsigned acc = 0;
      for (int t = 0; t < 7; t++) acc += (unsigned)k[t] * S[reflect101(x + t - 3, w)];
      hb[(size_t)y * w + x] = (uint16_t)acc;  // <= 255*256, no saturation
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst + (ptrdiff_t)y * dstride;
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int t = 0; t < 7; t++) acc += (uint32_t)k[t] * hb[(size_t)reflect101(y + t - 3, h) * w + x];
      const uint32_t v = (acc + 32768u) >> 16;
      D[x] = (uint8_t)(v > 255u ? 255u : v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST(roi, kps, threshold, nonmaxSuppression=true), FAST-9/16 -- SURVEY.md Appendix B.1.
// Call sites: ORBextractor.cc:848,854.
// ---------------------------------------------------------------------------------------------
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// Is the centre a FAST-9 corner at threshold t (strict compares)?
bool fast_is_corner(const int d[16], int t) {  // d[k] = centre - ring[k]
  for (int pol = 0; pol < 2; pol++) {
    int run = 0;
    for (int k = 0; k < 16 + 8; k++) {  // wrap: a run of 9 starting anywhere
      int v = pol ? -d[k & 15] : d[k & 15];
      run = (v > t) ? run + 1 : 0;
      if (run >= 9) return true;
    }
  }
  return false;
}

// cornerScore<16>: the largest threshold for which the pixel is still a corner, given it is
// one at `threshold`: max(threshold, max_arc min(d), max_arc min(-d)) - 1.
int fast_corner_score(const int d[16], int threshold) {
  int a0 = threshold;
  for (int k = 0; k < 16; k++) {
    int mn = INT_MAX, mx = INT_MIN;
    for (int j = 0; j < 9; j++) {
      int v = d[(k + j) & 15];
      mn = std::min(mn, v);
      mx = std::max(mx, v);
    }
    a0 = std::max(a0, mn);   // all ring darker than centre by more than mn-1
    a0 = std::max(a0, -mx);  // all ring brighter
  }
  return a0 - 1;
}

struct RoiKp { int x, y, score; };

void fast9_roi(const View& roi, int threshold, bool nms, std::vector<RoiKp>& out) {
  out.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  const int w = roi.w, h = roi.h;
  if (w < 7 || h < 7) return;
  std::vector<uint8_t> score((size_t)w * h, 0);  // 0 outside [3,w-4]x[3,h-4] and at non-corners
  std::vector<uint8_t> corner((size_t)w * h, 0);
  for (int y = 3; y < h - 3; y++) {
    for (int x = 3; x < w - 3; x++) {
      const int v = roi.row(y)[x];
      int d[16];
      for (int k = 0; k < 16; k++) d[k] = v - roi.row(y + kRingDy[k])[x + kRingDx[k]];
      if (!fast_is_corner(d, threshold)) continue;
      corner[(size_t)y * w + x] = 1;
      score[(size_t)y * w + x] = (uint8_t)fast_corner_score(d, threshold);
    }
  }
  for (int y = 3; y < h - 3; y++) {
    for (int x = 3; x < w - 3; x++) {
      if (!corner[(size_t)y * w + x]) continue;
      const int s = score[(size_t)y * w + x];
      if (nms) {
        bool keep = true;
        for (int dy = -1; dy <= 1 && keep; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            if (!(s > score[(size_t)(y + dy) * w + x + dx])) { keep = false; break; }
          }
        if (!keep) continue;
      }
      out.push_back(RoiKp{x, y, s});
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cv::fastAtan2(y, x) in degrees -- SURVEY.md Appendix B.4.  Call site: ORBextractor.cc:112.
// ---------------------------------------------------------------------------------------------
const float kAtanP1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
const float kAtanP3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
const float kAtanP5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
const float kAtanP7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);

float fast_atan2(float y, float x) {
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((kAtanP7 * c2 + kAtanP5) * c2 + kAtanP3) * c2 + kAtanP1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((kAtanP7 * c2 + kAtanP5) * c2 + kAtanP3) * c2 + kAtanP1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---------------------------------------------------------------------------------------------
// Reference constants.  ORBextractor.cc:73-79.
// ---------------------------------------------------------------------------------------------
const int PATCH_SIZE = 31;
const int HALF_PATCH_SIZE = 15;
const int EDGE_THRESHOLD = 19;

const int8_t kBriefPattern[256 * 4] = {
#include "brief_pattern.inc"
};

// IC_Angle -- ORBextractor.cc:86-113.  `img` is the UNBLURRED level.
float ic_angle(const View& img, float ptx, float pty, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = img.row(cv_round(pty)) + cv_round(ptx);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  const ptrdiff_t step = img.stride;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor -- ORBextractor.cc:120,132-171.  `img` is the BLURRED level.
const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);

void orb_descriptor(const OrcKeyPoint& kpt, const View& img, uint8_t* desc) {
  float angle = (float)kpt.angle * factorPI;
  float a = (float)cosf(angle), b = (float)sinf(angle);  // std::cos(float)/std::sin(float)
  const uint8_t* center = img.row(cv_round(kpt.y)) + cv_round(kpt.x);
  const ptrdiff_t step = img.stride;
  const int8_t* pat = kBriefPattern;
  auto get = [&](int idx) -> int {
    const float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
    return center[cv_round(px * b + py * a) * step + cv_round(px * a - py * b)];
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int k = 0; k < 8; k++) {
      int t0 = get(2 * k), t1 = get(2 * k + 1);
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

// ---------------------------------------------------------------------------------------------
// ExtractorNode / DivideNode -- ORBextractor.h:44-92, ORBextractor.cc:513-569.
// ---------------------------------------------------------------------------------------------
struct Node {
  std::vector<OrcKeyPoint> keys;
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::list<Node>::iterator lit;
  bool noMore = false;
  int32_t seq = 0;  // creation order: stands in for the heap address in the (size, ptr) sort (H1).  In the padding behind the flag:
                    // sizeof(Node) == 72 == sizeof(ExtractorNode) (vector + 4 Point2i + iterator + bool), so tie rule 2 compares
                    // addresses of heap blocks of the reference's size class

  void divide(Node& n1, Node& n2, Node& n3, Node& n4) const {
    const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
    n1.ULx = ULx; n1.ULy = ULy;
    n1.URx = ULx + halfX; n1.URy = ULy;
    n1.BLx = ULx; n1.BLy = ULy + halfY;
    n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = URx; n2.URy = URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = URx; n2.BRy = ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = BLx; n3.BLy = BLy;
    n3.BRx = n1.BRx; n3.BRy = BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = BRx; n4.BRy = BRy;
    // ORBextractor.cc:524,530,536,542: no effect on the result; kept so that the heap sees the reference's allocation pattern (tie rule 2)
    n1.keys.reserve(keys.size()); n2.keys.reserve(keys.size()); n3.keys.reserve(keys.size()); n4.keys.reserve(keys.size());
    for (size_t i = 0; i < keys.size(); i++) {
      const OrcKeyPoint& kp = keys[i];
      if (kp.x < n1.URx) {
        if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp);
      } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
      else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
  }
};

static_assert(sizeof(Node) == 72, "Node must have the size of the reference's ExtractorNode (tie rule 2)");

// How `sort(vPrevSizeAndPointerToNode)` (ORBextractor.cc:716) orders nodes of EQUAL size -- the reference compares heap addresses (H1):
//   0  creation order, a later node counts as the larger pointer (the pinned rule of this oracle and of the product)
//   1  the opposite: an earlier node counts as the larger pointer
//   2  the real addresses of the std::list nodes of THIS process (same node size and allocation pattern as the reference; what the
//      reference does on this allocator -- a function of the heap's history, not of the image)
//   3+ a seeded pseudo-random order (seed = rule)
struct TieStats { long sorts = 0, sorted_nodes = 0, nodes_in_ties = 0, breaks = 0, breaks_inside_a_tie = 0; };
thread_local TieStats g_tieStats;   // per thread: the cpu_baseline legs run one oracle extractor per thread; orc_tie_stats reports the calling thread's

// DistributeOctTree -- ORBextractor.cc:571-795.
std::vector<OrcKeyPoint> distribute_octtree(const std::vector<OrcKeyPoint>& vToDistributeKeys, int minX,
                                            int maxX, int minY, int maxY, int N, int tieRule = 0) {
  int32_t seq = 0;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> lNodes;
  std::vector<Node*> vpIniNodes(nIni > 0 ? nIni : 0);
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
    ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
    ni.BLx = ni.ULx; ni.BLy = maxY - minY;
    ni.BRx = ni.URx; ni.BRy = maxY - minY;
    ni.seq = seq++;
    ni.keys.reserve(vToDistributeKeys.size());   // ORBextractor.cc:590
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
    const OrcKeyPoint& kp = vToDistributeKeys[i];
    vpIniNodes[(size_t)(kp.x / hX)]->keys.push_back(kp);
  }
  for (auto lit = lNodes.begin(); lit != lNodes.end();) {
    if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
    else if (lit->keys.empty()) lit = lNodes.erase(lit);
    else ++lit;
  }

  bool bFinish = false;
  typedef std::pair<int, Node*> SizeNode;
  auto scramble = [tieRule](int32_t q) {
    uint64_t z = ((uint64_t)(uint32_t)q + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)tieRule * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto by_size_then_age = [&](const SizeNode& a, const SizeNode& b) {
    if (a.first != b.first) return a.first < b.first;
    switch (tieRule) {
      case 0: return a.second->seq < b.second->seq;
      case 1: return a.second->seq > b.second->seq;
      case 2: return std::less<const Node*>()(a.second, b.second);
      default: return scramble(a.second->seq) < scramble(b.second->seq);
    }
  };
  std::vector<SizeNode> vSizeAndPointerToNode;
  vSizeAndPointerToNode.reserve(lNodes.size() * 4);   // ORBextractor.cc:625

  // push children n1..n4 to the front (those with keys), recording the ones with >1 key
  auto push_children = [&](Node* ch[4], int& nToExpand) {
    for (int c = 0; c < 4; c++) {
      Node& n = *ch[c];
      if (n.keys.size() > 0) {
        n.seq = seq++;
        lNodes.push_front(n);
        if (n.keys.size() > 1) {
          nToExpand++;
          vSizeAndPointerToNode.push_back(std::make_pair((int)n.keys.size(), &lNodes.front()));
          lNodes.front().lit = lNodes.begin();
        }
      }
    }
  };

  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    auto lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->noMore) { ++lit; continue; }
      Node n1, n2, n3, n4;
      lit->divide(n1, n2, n3, n4);
      Node* ch[4] = {&n1, &n2, &n3, &n4};
      push_children(ch, nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<SizeNode> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), by_size_then_age);
        g_tieStats.sorts++;
        g_tieStats.sorted_nodes += (long)vPrev.size();
        for (size_t q = 0; q < vPrev.size(); q++)
          if ((q > 0 && vPrev[q - 1].first == vPrev[q].first) || (q + 1 < vPrev.size() && vPrev[q + 1].first == vPrev[q].first))
            g_tieStats.nodes_in_ties++;
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          vPrev[j].second->divide(n1, n2, n3, n4);
          Node* ch[4] = {&n1, &n2, &n3, &n4};
          int dummy = 0;
          push_children(ch, dummy);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) {
            g_tieStats.breaks++;
            if (j > 0 && vPrev[j - 1].first == vPrev[j].first) g_tieStats.breaks_inside_a_tie++;   // an equal-sized node stays undivided
            break;
          }
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }

  std::vector<OrcKeyPoint> vResultKeys;
  vResultKeys.reserve(lNodes.size());
  for (auto lit = lNodes.begin(); lit != lNodes.end(); ++lit) {
    const std::vector<OrcKeyPoint>& vNodeKeys = lit->keys;
    const OrcKeyPoint* pKP = &vNodeKeys[0];
    float maxResponse = pKP->response;
    for (size_t k = 1; k < vNodeKeys.size(); k++)
      if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
    vResultKeys.push_back(*pKP);
  }
  return vResultKeys;
}

// ---------------------------------------------------------------------------------------------
// ORBextractor -- ORBextractor.cc:442-502 (ctor), 797-895, 907-996.
// ---------------------------------------------------------------------------------------------
struct Extractor {
  int nfeatures;
  double scaleFactor;
  int nlevels, iniThFAST, minThFAST;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<Image> pyramid;                         // mvImagePyramid (without the dead 19-px border)
  std::vector<std::vector<OrcKeyPoint>> lastCandidates;  // per level, pre-quadtree (for stage tests)
  std::vector<Image> lastBlurred;
  int gaussVariant = 0;   // 0: error-diffused taps (OpenCV >= 4.1.1), 1: rounded taps + saturation (4.0.0 - 4.1.0)
  int tieRule = 0;        // H1, see distribute_octtree

  Extractor(int nf, float sf, int nl, int ini, int mn)
      : nfeatures(nf), scaleFactor(sf), nlevels(nl), iniThFAST(ini), minThFAST(mn) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    pyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale =
        nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cv_round(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // ComputePyramid -- ORBextractor.cc:971-996
  void computePyramid(const View& image) {
    for (int level = 0; level < nlevels; ++level) {
      float scale = mvInvScaleFactor[level];
      int sw = cv_round((float)image.w * scale), sh = cv_round((float)image.h * scale);
      pyramid[level].alloc(sw, sh);
      if (level != 0) {
        resize_linear_u8(pyramid[level - 1].view(), pyramid[level].d.data(), sw, sh, sw);
      } else {
        for (int y = 0; y < sh; y++) memcpy(pyramid[0].row(y), image.row(y), sw);
      }
    }
  }

  // ComputeKeyPointsOctTree -- ORBextractor.cc:797-895
  void computeKeyPointsOctTree(std::vector<std::vector<OrcKeyPoint>>& allKeypoints) {
    allKeypoints.resize(nlevels);
    lastCandidates.assign(nlevels, {});
    const float W = 30;
    std::vector<RoiKp> vKeysCell;
    for (int level = 0; level < nlevels; ++level) {
      const View img = pyramid[level].view();
      const int minBorderX = EDGE_THRESHOLD - 3;
      const int minBorderY = minBorderX;
      const int maxBorderX = img.w - EDGE_THRESHOLD + 3;
      const int maxBorderY = img.h - EDGE_THRESHOLD + 3;
      std::vector<OrcKeyPoint> vToDistributeKeys;
      vToDistributeKeys.reserve(nfeatures * 10);
      const float width = (maxBorderX - minBorderX);
      const float height = (maxBorderY - minBorderY);
      const int nCols = width / W;
      const int nRows = height / W;
      const int wCell = std::ceil(width / nCols);
      const int hCell = std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = minBorderY + i * hCell;
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = minBorderX + j * wCell;
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = maxBorderX;
          const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
          View roi{img.row(y0) + x0, x1 - x0, y1 - y0, img.stride};
          fast9_roi(roi, iniThFAST, true, vKeysCell);
          if (vKeysCell.empty()) fast9_roi(roi, minThFAST, true, vKeysCell);
          for (const RoiKp& k : vKeysCell) {
            OrcKeyPoint kp;  // cv::KeyPoint(x, y, 7.f, -1, score)
            kp.x = (float)k.x; kp.y = (float)k.y; kp.size = 7.f; kp.angle = -1.f;
            kp.response = (float)k.score; kp.octave = 0; kp.class_id = -1;
            kp.x += j * wCell;
            kp.y += i * hCell;
            vToDistributeKeys.push_back(kp);
          }
        }
      }
      lastCandidates[level] = vToDistributeKeys;
      std::vector<OrcKeyPoint>& keypoints = allKeypoints[level];
      keypoints = distribute_octtree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                     mnFeaturesPerLevel[level], tieRule);
      const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level];
      for (OrcKeyPoint& kp : keypoints) {
        kp.x += minBorderX;
        kp.y += minBorderY;
        kp.octave = level;
        kp.size = scaledPatchSize;
      }
    }
    for (int level = 0; level < nlevels; ++level) {
      const View img = pyramid[level].view();
      for (OrcKeyPoint& kp : allKeypoints[level]) kp.angle = ic_angle(img, kp.x, kp.y, umax);
    }
  }

  // operator() -- ORBextractor.cc:907-969.  Returns keypoints; desc gets 32 B per keypoint.
  void extract(const View& image, std::vector<OrcKeyPoint>& kps, std::vector<uint8_t>& desc) {
    kps.clear();
    desc.clear();
    if (image.w <= 0 || image.h <= 0 || !image.p) return;
    computePyramid(image);
    std::vector<std::vector<OrcKeyPoint>> allKeypoints;
    computeKeyPointsOctTree(allKeypoints);
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    desc.assign((size_t)nkeypoints * 32, 0);
    kps.reserve(nkeypoints);
    lastBlurred.assign(nlevels, Image());
    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<OrcKeyPoint>& keypoints = allKeypoints[level];
      int nkeypointsLevel = (int)keypoints.size();
      if (nkeypointsLevel == 0) continue;
      Image& working = lastBlurred[level];
      working.alloc(pyramid[level].w, pyramid[level].h);
      gauss7_u8(pyramid[level].view(), working.d.data(), working.w, gaussVariant);
      const View wv = working.view();
      for (int i = 0; i < nkeypointsLevel; i++) orb_descriptor(keypoints[i], wv, &desc[(size_t)(offset + i) * 32]);
      offset += nkeypointsLevel;
      if (level != 0) {
        float scale = mvScaleFactor[level];
        for (OrcKeyPoint& kp : keypoints) { kp.x *= scale; kp.y *= scale; }
      }
      kps.insert(kps.end(), keypoints.begin(), keypoints.end());
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Hamming -- ORBmatcher.cc:1605-1621.
// ---------------------------------------------------------------------------------------------
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t wa, wb;
    memcpy(&wa, a + 4 * i, 4);
    memcpy(&wb, b + 4 * i, 4);
    unsigned int v = wa ^ wb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// ---------------------------------------------------------------------------------------------
// Frame grid -- Frame.h:36-37, Frame.cc:98-99,114-129,209-274.
// ---------------------------------------------------------------------------------------------
const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;

struct FrameGrid {
  const OrcKeyPoint* kpsUn;
  int N;
  float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
  std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];

  FrameGrid(const OrcKeyPoint* k, int n, const float bounds[4]) : kpsUn(k), N(n) {
    mnMinX = bounds[0]; mnMaxX = bounds[1]; mnMinY = bounds[2]; mnMaxY = bounds[3];
    mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(mnMaxX - mnMinX);
    mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(mnMaxY - mnMinY);
    for (int i = 0; i < N; i++) {  // AssignFeaturesToGrid
      int gx, gy;
      if (posInGrid(kpsUn[i], gx, gy)) mGrid[gx][gy].push_back(i);
    }
  }
  bool posInGrid(const OrcKeyPoint& kp, int& posX, int& posY) const {
    posX = (int)roundf((kp.x - mnMinX) * mfGridElementWidthInv);
    posY = (int)roundf((kp.y - mnMinY) * mfGridElementHeightInv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
    return true;
  }
  std::vector<size_t> getFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel,
                                        const int maxLevel) const {
    std::vector<size_t> vIndices;
    const int nMinCellX = std::max(0, (int)floorf((x - mnMinX - r) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceilf((x - mnMinX + r) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floorf((y - mnMinY - r) * mfGridElementHeightInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceilf((y - mnMinY + r) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const std::vector<size_t>& vCell = mGrid[ix][iy];
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const OrcKeyPoint& kpUn = kpsUn[vCell[j]];
          if (bCheckLevels) {
            if (kpUn.octave < minLevel) continue;
            if (maxLevel >= 0)
              if (kpUn.octave > maxLevel) continue;
          }
          const float distx = kpUn.x - x;
          const float disty = kpUn.y - y;
          if (fabsf(distx) < r && fabsf(disty) < r) vIndices.push_back(vCell[j]);
        }
      }
    }
    return vIndices;
  }
};

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // ORBmatcher.cc:37-39

// ComputeThreeMaxima -- ORBmatcher.cc:1554-1595
void compute_three_maxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
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

int rot_bin(float a1, float a2) {  // ORBmatcher.cc:470-475
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

}  // namespace

// =============================================================================================
// C ABI used by the tests / bench (ctypes).
// =============================================================================================
extern "C" {

int orc_cv_round_f(float v) { return cv_round(v); }
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void orc_sincos_host(float angle_deg, float* a, float* b) {
  float ang = angle_deg * factorPI;
  *a = cosf(ang);
  *b = sinf(ang);
}
void orc_gauss_kernel(int n, double sigma, int* out) { gauss_kernel_fixed_ed(n, sigma, out); }
void orc_gauss_kernel_variant(int n, double sigma, int variant, int* out) {
  if (variant == 1) gauss_kernel_fixed_rounded(n, sigma, out);
  else gauss_kernel_fixed_ed(n, sigma, out);
}
void orc_gauss7_u8_variant(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int variant) {
  gauss7_u8(View{src, w, h, sstride}, dst, dstride, variant);
}
void orc_extractor_set_gauss_variant(void* h, int variant) { ((Extractor*)h)->gaussVariant = variant; }
void orc_extractor_set_tie_rule(void* h, int rule) { ((Extractor*)h)->tieRule = rule; }
// {sorts, nodes sorted, nodes that had an equal-sized neighbour, early breaks, breaks that left an equal-sized node undivided}
void orc_tie_stats(long* out5, int reset) {
  out5[0] = g_tieStats.sorts; out5[1] = g_tieStats.sorted_nodes; out5[2] = g_tieStats.nodes_in_ties;
  out5[3] = g_tieStats.breaks; out5[4] = g_tieStats.breaks_inside_a_tie;
  if (reset) g_tieStats = TieStats();
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride) {
  resize_linear_u8(View{src, sw, sh, sstride}, dst, dw, dh, dstride);
}
void orc_gauss7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  gauss7_u8(View{src, w, h, sstride}, dst, dstride);
}
// cv::FAST on a ROI.  out_xys: triples (x, y, score).  Returns count (<= cap written).
int orc_fast9(const uint8_t* img, int w, int h, int stride, int threshold, int nms, int* out_xys, int cap) {
  std::vector<RoiKp> k;
  fast9_roi(View{img, w, h, stride}, threshold, nms != 0, k);
  for (int i = 0; i < (int)k.size() && i < cap; i++) {
    out_xys[3 * i] = k[i].x; out_xys[3 * i + 1] = k[i].y; out_xys[3 * i + 2] = k[i].score;
  }
  return (int)k.size();
}
// brute-force definition of the score: largest t for which the pixel is still a corner (-1: never)
int orc_fast_score_bruteforce(const uint8_t* img, int stride, int x, int y) {
  int d[16];
  const int v = img[(ptrdiff_t)y * stride + x];
  for (int k = 0; k < 16; k++) d[k] = v - img[(ptrdiff_t)(y + kRingDy[k]) * stride + x + kRingDx[k]];
  int best = -1;
  for (int t = 0; t < 256; t++)
    if (fast_is_corner(d, t)) best = t;
  return best;
}

int orc_distribute_octtree(const OrcKeyPoint* in, int n, int minX, int maxX, int minY, int maxY, int N,
                           OrcKeyPoint* out, int cap) {
  std::vector<OrcKeyPoint> v(in, in + n);
  std::vector<OrcKeyPoint> r = distribute_octtree(v, minX, maxX, minY, maxY, N);
  for (int i = 0; i < (int)r.size() && i < cap; i++) out[i] = r[i];
  return (int)r.size();
}

void* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  return new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void orc_extractor_destroy(void* h) { delete (Extractor*)h; }
void orc_extractor_tables(void* h, float* sf, float* isf, float* s2, float* is2, int* nfeat, int* umax16) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; i++) {
    if (sf) sf[i] = e->mvScaleFactor[i];
    if (isf) isf[i] = e->mvInvScaleFactor[i];
    if (s2) s2[i] = e->mvLevelSigma2[i];
    if (is2) is2[i] = e->mvInvLevelSigma2[i];
    if (nfeat) nfeat[i] = e->mnFeaturesPerLevel[i];
  }
  if (umax16) for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}
// Returns number of keypoints (writes at most cap).
int orc_extract(void* h, const uint8_t* gray, int rows, int cols, int stride, OrcKeyPoint* kps, uint8_t* desc,
                int cap) {
  Extractor* e = (Extractor*)h;
  std::vector<OrcKeyPoint> k;
  std::vector<uint8_t> d;
  e->extract(View{gray, cols, rows, stride}, k, d);
  int n = (int)k.size();
  int m = std::min(n, cap);
  if (m > 0) {
    memcpy(kps, k.data(), (size_t)m * sizeof(OrcKeyPoint));
    memcpy(desc, d.data(), (size_t)m * 32);
  }
  return n;
}
// Stage accessors (valid after orc_extract).
void orc_level_size(void* h, int level, int* w, int* hgt) {
  Extractor* e = (Extractor*)h;
  *w = e->pyramid[level].w; *hgt = e->pyramid[level].h;
}
void orc_level_copy(void* h, int level, int blurred, uint8_t* out) {
  Extractor* e = (Extractor*)h;
  const Image& im = blurred ? e->lastBlurred[level] : e->pyramid[level];
  if (!im.d.empty()) memcpy(out, im.d.data(), im.d.size());
}
int orc_level_candidates(void* h, int level, OrcKeyPoint* out, int cap) {
  Extractor* e = (Extractor*)h;
  const auto& c = e->lastCandidates[level];
  for (int i = 0; i < (int)c.size() && i < cap; i++) out[i] = c[i];
  return (int)c.size();
}

int orc_hamming(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

int orc_get_features_in_area(const OrcKeyPoint* kpsUn, int n, const float bounds[4], float x, float y, float r,
                             int minLevel, int maxLevel, int* out, int cap) {
  FrameGrid g(kpsUn, n, bounds);
  std::vector<size_t> v = g.getFeaturesInArea(x, y, r, minLevel, maxLevel);
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = (int)v[i];
  return (int)v.size();
}

// ORBmatcher::SearchForInitialization -- ORBmatcher.cc:400-515.  prev_xy: in/out (vbPrevMatched).
int orc_search_for_initialization(const OrcKeyPoint* kps1, const uint8_t* desc1, int n1, const OrcKeyPoint* kps2,
                                  const uint8_t* desc2, int n2, const float bounds[4], float* prev_xy,
                                  int* vnMatches12, int windowSize, float mfNNratio, int mbCheckOrientation) {
  FrameGrid F2(kps2, n2, bounds);
  int nmatches = 0;
  for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance(n2, INT_MAX);
  std::vector<int> vnMatches21(n2, -1);
  for (size_t i1 = 0, iend1 = n1; i1 < iend1; i1++) {
    const OrcKeyPoint& kp1 = kps1[i1];
    int level1 = kp1.octave;
    if (level1 > 0) continue;
    std::vector<size_t> vIndices2 =
        F2.getFeaturesInArea(prev_xy[2 * i1], prev_xy[2 * i1 + 1], windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + 32 * i1;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (size_t i2 : vIndices2) {
      int dist = descriptor_distance(d1, desc2 + 32 * i2);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = (int)i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) rotHist[rot_bin(kps1[i1].angle, kps2[bestIdx2].angle)].push_back((int)i1);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (size_t i1 = 0, iend1 = n1; i1 < iend1; i1++)
    if (vnMatches12[i1] >= 0) {
      prev_xy[2 * i1] = kps2[vnMatches12[i1]].x;
      prev_xy[2 * i1 + 1] = kps2[vnMatches12[i1]].y;
    }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) -- ORBmatcher.cc:45-132.
// MapPoint fields are passed as flat arrays (the reference reads them through pMP->...):
//   mp_flags bit0 = mbTrackInView, bit1 = isBad(), bit2 = plCandidato, bit3 = Observations()>0.
// kp_occupied[idx] in: 1 iff F.mvpMapPoints[idx] != NULL && ->Observations()>0 at call time.
// kp_assigned[idx] out: index of the MapPoint written to F.mvpMapPoints[idx] (last writer), -1 if untouched.
int orc_search_by_projection(const OrcKeyPoint* kpsUn, const uint8_t* desc, int n, const float bounds[4],
                             const float* mvScaleFactors, const uint8_t* kp_occupied, const float* mp_proj_xy,
                             const int* mp_level, const float* mp_viewcos, const uint8_t* mp_flags,
                             const uint8_t* mp_desc, int n_mp, float th, float mfNNratio, int* kp_assigned) {
  FrameGrid F(kpsUn, n, bounds);
  std::vector<uint8_t> occ(kp_occupied, kp_occupied + n);
  for (int i = 0; i < n; i++) kp_assigned[i] = -1;
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < n_mp; iMP++) {
    const uint8_t fl = mp_flags[iMP];
    if (!(fl & 1)) continue;
    if (fl & 2) continue;
    const int nPredictedLevel = mp_level[iMP];
    float r = (fl & 4) ? 4.0 : (mp_viewcos[iMP] > 0.998 ? 2.5 : 4.0);
    if (bFactor) r *= th;
    const std::vector<size_t> vIndices =
        F.getFeaturesInArea(mp_proj_xy[2 * iMP], mp_proj_xy[2 * iMP + 1], r * mvScaleFactors[nPredictedLevel],
                            nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    const uint8_t* MPdescriptor = mp_desc + 32 * (size_t)iMP;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (size_t idx : vIndices) {
      if (occ[idx]) continue;
      const int dist = descriptor_distance(MPdescriptor, desc + 32 * idx);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
        bestLevel = kpsUn[idx].octave; bestIdx = (int)idx;
      } else if (dist < bestDist2) {
        bestLevel2 = kpsUn[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      kp_assigned[bestIdx] = iMP;
      occ[bestIdx] = (fl & 8) ? 1 : 0;
      nmatches++;
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th) -- ORBmatcher.cc:1292-1423 and
// (Frame&, KeyFrame*, set, th, ORBdist) -- :1425-1552, from the projection onwards: the caller has
// already projected each source point to (u,v) (float cv::Mat arithmetic, :1326-1336 / :1451-1459),
// applied the reference's rejections, and passes src_valid=0 for rejected / absent points.
//   src_level: nLastOctave (:1345) resp. nPredictedLevel (:1477); window levels [l-1, l+1].
//   max_dist: TH_HIGH (:1381) resp. ORBdist (:1506).
//   skip_any_occupied: 0 => skip kps whose MapPoint has Observations()>0 (:1364-1366);
//                      1 => skip kps with any MapPoint (:1493-1494).
//   src_flags bit3: Observations()>0 of the source MapPoint (needed for occupancy replay, mode 0).
// kp_assigned out as above; pruned slots (rotation check) are reported as -2 ("set to NULL").
int orc_search_by_projection_uv(const OrcKeyPoint* kpsUn, const uint8_t* desc, int n, const float bounds[4],
                                const float* mvScaleFactors, const uint8_t* kp_occupied, const float* src_uv,
                                const int* src_level, const float* src_angle, const uint8_t* src_flags,
                                const uint8_t* src_valid, const uint8_t* src_desc, int n_src, float th,
                                int max_dist, int skip_any_occupied, int mbCheckOrientation, int* kp_assigned) {
  FrameGrid F(kpsUn, n, bounds);
  std::vector<uint8_t> occ(kp_occupied, kp_occupied + n);
  for (int i = 0; i < n; i++) kp_assigned[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < n_src; i++) {
    if (!src_valid[i]) continue;
    const int lvl = src_level[i];
    const float radius = th * mvScaleFactors[lvl];
    std::vector<size_t> vIndices2 = F.getFeaturesInArea(src_uv[2 * i], src_uv[2 * i + 1], radius, lvl - 1, lvl + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = src_desc + 32 * (size_t)i;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t i2 : vIndices2) {
      if (occ[i2]) continue;
      const int dist = descriptor_distance(dMP, desc + 32 * i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= max_dist) {
      kp_assigned[bestIdx2] = i;
      occ[bestIdx2] = skip_any_occupied ? 1 : ((src_flags[i] & 8) ? 1 : 0);
      nmatches++;
      if (mbCheckOrientation) rotHist[rot_bin(src_angle[i], kpsUn[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          kp_assigned[rotHist[i][j]] = -2;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

// MapPoint::ComputeDistinctiveDescriptors -- MapPoint.cc:227-292, from the descriptor list onwards:
// index of the descriptor with the least median Hamming distance to the rest (first minimum wins).
int orc_distinctive_descriptor(const uint8_t* descs, int N) {
  if (N <= 0) return -1;
  std::vector<float> Distances((size_t)N * N);
  for (int i = 0; i < N; i++) {
    Distances[(size_t)i * N + i] = 0;
    for (int j = i + 1; j < N; j++) {
      int distij = descriptor_distance(descs + 32 * (size_t)i, descs + 32 * (size_t)j);
      Distances[(size_t)i * N + j] = distij;
      Distances[(size_t)j * N + i] = distij;
    }
  }
  int BestMedian = INT_MAX;
  int BestIdx = 0;
  for (int i = 0; i < N; i++) {
    std::vector<int> vDists(Distances.begin() + (size_t)i * N, Distances.begin() + (size_t)(i + 1) * N);
    std::sort(vDists.begin(), vDists.end());
    int median = vDists[0.5 * (N - 1)];
    if (median < BestMedian) { BestMedian = median; BestIdx = i; }
  }
  return BestIdx;
}

// Frame::antidistorsionarProyeccionEquidistante -- Frame.cc:355-384 (os1's equidistant fisheye, modo 1).
// K = [fx 0 cx; 0 fy cy; 0 0 1] as floats; points in/out as float pairs.
void orc_undistort_equidistant(float* xy, int n, float fx, float fy, float cx, float cy) {
  const double f0 = fx, f1 = fy, c0 = cx, c1 = cy;
  for (int i = 0; i < n; i++) {
    double pi0 = xy[2 * i], pi1 = xy[2 * i + 1];
    double pw0 = (pi0 - c0) / f0, pw1 = (pi1 - c1) / f1;
    double theta_d = sqrt(pw0 * pw0 + pw1 * pw1);
    double scale = theta_d > 1e-8 ? std::tan(theta_d) / theta_d : 1.0;
    double pu0 = pw0 * scale, pu1 = pw1 * scale;
    // pr = K(float 3x3) * Vec3d(pu, 1): Matx33f * Vec3d promotes to double
    double pr0 = (double)fx * pu0 + 0.0 * pu1 + (double)cx * 1.0;
    double pr1 = 0.0 * pu0 + (double)fy * pu1 + (double)cy * 1.0;
    double pr2 = 0.0 * pu0 + 0.0 * pu1 + 1.0 * 1.0;
    xy[2 * i] = (float)(pr0 / pr2);
    xy[2 * i + 1] = (float)(pr1 / pr2);
  }
}

// ---------------------------------------------------------------------------------------------
// Bag of words: Frame::ComputeBoW (Frame.cc:277-284) = DBoW2 TemplatedVocabulary<FORB>::transform
// (features, BowVector&, FeatureVector&, levelsup) -- Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1204,
// single-feature descent :1306-1347, BowVector::addWeight/addIfNotExist/normalize (BowVector.cpp:30-82),
// FeatureVector::addFeature (FeatureVector.cpp:27-41), FORB::distance (FORB.cpp:81-101).
// The vocabulary is the fork's binary file format (TemplatedVocabulary.h:1563-1640): header bytes k, L, scoring,
// weighting, then 45-byte records {int32 parent, u8 isLeaf, u8 descriptor[32], double weight}; node ids count
// from 1 in record order, children are appended in record order, word ids count leaves in record order.
// (The reference's `while(!eof)` loop appends one stale copy of the last record; a copy can never win the strict
// `d < best_d` descent, so it is not reproduced.)
// ---------------------------------------------------------------------------------------------
struct BowNode {
  int parent = 0;
  std::vector<int> children;
  uint8_t descriptor[32] = {};
  double weight = 0;
  int word_id = 0;
  bool isLeaf() const { return children.empty(); }
};
struct BowVocabulary {
  int k, L, scoring, weighting;
  std::vector<BowNode> nodes;
};

void* orc_vocab_create(int k, int L, int scoring, int weighting, const uint8_t* records, int nrecords) {
  BowVocabulary* v = new BowVocabulary{k, L, scoring, weighting, {}};
  v->nodes.resize(1);
  int nwords = 0;
  for (int r = 0; r < nrecords; r++) {
    const uint8_t* buffer = records + 45 * (size_t)r;
    int nid = (int)v->nodes.size();
    v->nodes.resize(v->nodes.size() + 1);
    int pid;
    memcpy(&pid, buffer, 4);
    v->nodes[nid].parent = pid;
    v->nodes[pid].children.push_back(nid);
    memcpy(v->nodes[nid].descriptor, buffer + 5, 32);
    memcpy(&v->nodes[nid].weight, buffer + 37, 8);
    if (buffer[4] > 0) v->nodes[nid].word_id = nwords++;
  }
  return v;
}
void orc_vocab_destroy(void* h) { delete (BowVocabulary*)h; }

// Accumulation half of transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1136-1204): per-feature
// (word id, weight, node id) -> BowVector / FeatureVector.  Two interchangeable implementations:
//  * the restatement below (BowVector.cpp:34-90 addWeight / addIfNotExist / normalize, FeatureVector.cpp:31-46);
//  * the REFERENCE'S OWN BowVector.cpp + FeatureVector.cpp, compiled from /root/reference into
//    oracle/_ref/libdbow2_vec.so (oracle/Makefile, wrapper oracle/dbow2_ref_wrap.cpp) and switched in with
//    orc_use_dbow2_ref().  tests/test_bow_oracle.py checks both against each other; tests/golden/vga_seed1_bow.npz
//    is generated through the reference code (tools/gen_golden.py).
typedef int (*BowAccumFn)(const uint32_t* word, const double* weight, const uint32_t* node, int n, int weighting,
                          int must, int norm_l2, uint32_t* bow_ids, double* bow_vals, int* n_words, uint32_t* fv_nodes,
                          uint32_t* fv_off, uint32_t* fv_feat, int* n_fv);
static int bow_accumulate_restated(const uint32_t* word, const double* weight, const uint32_t* node, int n,
                                   int weighting, int must, int norm_l2, uint32_t* bow_ids, double* bow_vals,
                                   int* n_words, uint32_t* fv_nodes, uint32_t* fv_off, uint32_t* fv_feat, int* n_fv) {
  std::map<uint32_t, double> v;
  std::map<uint32_t, std::vector<unsigned>> fv;
  const bool tf = weighting == 0 || weighting == 1;
  for (int i = 0; i < n; i++) {
    if (!(weight[i] > 0)) continue;                            // stopped word
    if (tf) v[word[i]] += weight[i];                           // addWeight
    else v.insert({word[i], weight[i]});                       // addIfNotExist
    fv[node[i]].push_back(i);                                  // addFeature
  }
  if (tf && !v.empty() && !must) {
    const double nd = v.size();
    for (auto& e : v) e.second /= nd;
  }
  if (must) {
    double norm = 0.0;
    if (!norm_l2) { for (auto& e : v) norm += fabs(e.second); }
    else { for (auto& e : v) norm += e.second * e.second; norm = sqrt(norm); }
    if (norm > 0.0) for (auto& e : v) e.second /= norm;
  }
  int nw = 0;
  for (auto& e : v) { bow_ids[nw] = e.first; bow_vals[nw] = e.second; nw++; }
  *n_words = nw;
  int nn = 0, pos = 0;
  for (auto& e : fv) {
    fv_nodes[nn] = e.first;
    fv_off[nn] = pos;
    for (unsigned f : e.second) fv_feat[pos++] = f;
    nn++;
  }
  fv_off[nn] = pos;
  *n_fv = nn;
  return 0;
}
static BowAccumFn g_bow_accum = bow_accumulate_restated;
static void* g_dbow2_ref = nullptr;
// Switch the accumulation to the reference's own code (path of oracle/_ref/libdbow2_vec.so), or back with NULL.
// Returns 0 on success, -1 if the library or its entry point is missing.
int orc_use_dbow2_ref(const char* so_path) {
  if (!so_path) { g_bow_accum = bow_accumulate_restated; return 0; }
  if (!g_dbow2_ref) g_dbow2_ref = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!g_dbow2_ref) return -1;
  BowAccumFn f = (BowAccumFn)dlsym(g_dbow2_ref, "dbow2ref_accumulate");
  if (!f) return -1;
  g_bow_accum = f;
  return 0;
}
int orc_bow_accumulator_is_reference() { return g_bow_accum != bow_accumulate_restated; }

// transform(features, v, fv, levelsup).  Outputs: BowVector as (ascending word id, value) pairs; FeatureVector as
// ascending node ids with CSR offsets into the feature-index list; per-feature word / node for debugging.
int orc_bow_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* bow_ids, double* bow_vals,
                      int* n_words, uint32_t* fv_nodes, uint32_t* fv_off, uint32_t* fv_feat, int* n_fv,
                      uint32_t* word_of_feature, uint32_t* node_of_feature) {
  const BowVocabulary& V = *(BowVocabulary*)h;
  const bool must = V.scoring != 5;                 // DotProductScoring does not normalise (ScoringObject.h:74-89)
  const bool normL2 = V.scoring == 1;
  *n_words = 0; *n_fv = 0; fv_off[0] = 0;
  if (V.nodes.size() <= 1) return 0;                // empty(): v and fv stay clear (TemplatedVocabulary.h:1143-1146)
  std::vector<uint32_t> word(n), node(n);
  std::vector<double> weight(n);
  for (int i = 0; i < n; i++) {
    const uint8_t* feature = desc + 32 * (size_t)i;
    // TemplatedVocabulary.h:1306-1347
    const int nid_level = V.L - levelsup;
    uint32_t nid = 0;
    int final_id = 0, current_level = 0;
    do {
      ++current_level;
      const std::vector<int>& nodes = V.nodes[final_id].children;
      final_id = nodes[0];
      double best_d = descriptor_distance(feature, V.nodes[final_id].descriptor);
      for (size_t c = 1; c < nodes.size(); c++) {
        double d = descriptor_distance(feature, V.nodes[nodes[c]].descriptor);
        if (d < best_d) { best_d = d; final_id = nodes[c]; }
      }
      if (current_level == nid_level) nid = final_id;
    } while (!V.nodes[final_id].isLeaf());
    word[i] = V.nodes[final_id].word_id;
    weight[i] = V.nodes[final_id].weight;
    node[i] = nid;
    if (word_of_feature) word_of_feature[i] = word[i];
    if (node_of_feature) node_of_feature[i] = nid;
  }
  return g_bow_accum(word.data(), weight.data(), node.data(), n, V.weighting, must ? 1 : 0, normL2 ? 1 : 0, bow_ids,
                     bow_vals, n_words, fv_nodes, fv_off, fv_feat, n_fv);
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) -- ORBmatcher.cc:154-283.
// FeatureVectors as ascending node ids + CSR offsets + feature indices; valid1[i] != 0 <=> the keyframe keypoint has a
// MapPoint that is not bad; angles from pKF->mvKeysUn / F.mvKeys.  matches21[i2] = keyframe index whose MapPoint
// is assigned to frame keypoint i2, or -1.  Returns nmatches.
int orc_search_by_bow(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                      const uint32_t* fv1_nodes, const uint32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
                      const uint8_t* desc2, const float* angle2, int n2, const uint32_t* fv2_nodes,
                      const uint32_t* fv2_off, const uint32_t* fv2_feat, int nfv2, float nnratio, int checkOri,
                      int32_t* matches21) {
  (void)n1;
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  for (int i = 0; i < n2; i++) matches21[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < nfv1 && b < nfv2) {
    if (fv1_nodes[a] == fv2_nodes[b]) {
      for (uint32_t iKF = fv1_off[a]; iKF < fv1_off[a + 1]; iKF++) {
        const unsigned realIdxKF = fv1_feat[iKF];
        if (!valid1[realIdxKF]) continue;
        const uint8_t* dKF = desc1 + 32 * (size_t)realIdxKF;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (uint32_t iF = fv2_off[b]; iF < fv2_off[b + 1]; iF++) {
          const unsigned realIdxF = fv2_feat[iF];
          if (matches21[realIdxF] >= 0) continue;
          const int dist = descriptor_distance(dKF, desc2 + 32 * (size_t)realIdxF);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches21[bestIdxF] = realIdxKF;
            if (checkOri) {
              float rot = angle1[realIdxKF] - angle2[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back(bestIdxF);
            }
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (fv1_nodes[a] < fv2_nodes[b]) {
      a = (int)(std::lower_bound(fv1_nodes, fv1_nodes + nfv1, fv2_nodes[b]) - fv1_nodes);
    } else {
      b = (int)(std::lower_bound(fv2_nodes, fv2_nodes + nfv2, fv1_nodes[a]) - fv2_nodes);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) { matches21[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}


// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) -- ORBmatcher.cc:517-650.  valid1/valid2: the keypoint
// has a MapPoint that is not bad.  matches12[i1] = index in keyframe 2 or -1.  Returns nmatches.
int orc_search_by_bow_kf(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                         const uint32_t* fv1_nodes, const uint32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
                         const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                         const uint32_t* fv2_nodes, const uint32_t* fv2_off, const uint32_t* fv2_feat, int nfv2,
                         float nnratio, int checkOri, int32_t* matches12) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<bool> vbMatched2(n2, false);
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  int a = 0, b = 0;
  while (a < nfv1 && b < nfv2) {
    if (fv1_nodes[a] == fv2_nodes[b]) {
      for (uint32_t i1 = fv1_off[a]; i1 < fv1_off[a + 1]; i1++) {
        const size_t idx1 = fv1_feat[i1];
        if (!valid1[idx1]) continue;
        const uint8_t* d1 = desc1 + 32 * idx1;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (uint32_t i2 = fv2_off[b]; i2 < fv2_off[b + 1]; i2++) {
          const size_t idx2 = fv2_feat[i2];
          if (vbMatched2[idx2] || !valid2[idx2]) continue;
          int dist = descriptor_distance(d1, desc2 + 32 * idx2);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < TH_LOW) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = true;
            if (checkOri) {
              float rot = angle1[idx1] - angle2[bestIdx2];
              if (rot < 0.0) rot += 360.0f;
              int bin = round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back((int)idx1);
            }
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (fv1_nodes[a] < fv2_nodes[b]) {
      a = (int)(std::lower_bound(fv1_nodes, fv1_nodes + nfv1, fv2_nodes[b]) - fv1_nodes);
    } else {
      b = (int)(std::lower_bound(fv2_nodes, fv2_nodes + nfv2, fv1_nodes[a]) - fv2_nodes);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) for 8-bit 3- or 4-channel images, as called from
// Tracking::GrabImageMonocular (Tracking.cc:96-109).  OpenCV's RGB2Gray<uchar> (imgproc color_rgb): fixed-point
// gray = (R*cr + G*cg + B*cb + (1 << (shift-1))) >> shift; variant 0: shift 15, {9798, 19235, 3735}
// (OpenCV >= 4.1.1); variant 1: shift 14, {4899, 9617, 1868} (earlier releases).  Alpha is ignored.  UNPINNED like the
// other OpenCV-side primitives (recalled from upstream; pinned by known-answer tests only).
void orc_cvt_gray(const uint8_t* src, int rows, int cols, int stride, int channels, int rgb_order, int variant,
                  uint8_t* dst, int dstride) {
  const int shift = variant == 0 ? 15 : 14;
  const int cr = variant == 0 ? 9798 : 4899, cg = variant == 0 ? 19235 : 9617, cb = variant == 0 ? 3735 : 1868;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      const uint8_t* p = src + (size_t)y * stride + (size_t)x * channels;
      const int r = rgb_order ? p[0] : p[2], g = p[1], b = rgb_order ? p[2] : p[0];
      dst[(size_t)y * dstride + x] = (uint8_t)((r * cr + g * cg + b * cb + (1 << (shift - 1))) >> shift);
    }
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:652-804) from the epipole onwards, with
// ORBmatcher::CheckDistEpipolarLine (:135-152).  hasMP1/hasMP2: the keypoint already has a MapPoint (skipped);
// F12 row-major float[9]; (ex, ey) the epipole in image 2 (:660-666); scale2 / sigma2: pKF2->mvScaleFactors /
// mvLevelSigma2.  NOTE vbMatched2 is never set by the reference (:676, :718), so it is not modelled.
// pairs = (idx1, idx2) in ascending idx1 (:792-800).  Returns nmatches.
int orc_search_for_triangulation(const OrcKeyPoint* kps1, const uint8_t* desc1, const uint8_t* hasMP1, int n1,
                                 const uint32_t* fv1_nodes, const uint32_t* fv1_off, const uint32_t* fv1_feat, int nfv1,
                                 const OrcKeyPoint* kps2, const uint8_t* desc2, const uint8_t* hasMP2, int n2,
                                 const uint32_t* fv2_nodes, const uint32_t* fv2_off, const uint32_t* fv2_feat, int nfv2,
                                 const float* F12, float ex, float ey, const float* scale2, const float* sigma2,
                                 int checkOri, int32_t* pairs) {
  (void)n2;
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  int nmatches = 0;
  std::vector<int> vMatches12(n1, -1);
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < nfv1 && b < nfv2) {
    if (fv1_nodes[a] == fv2_nodes[b]) {
      for (uint32_t i1 = fv1_off[a]; i1 < fv1_off[a + 1]; i1++) {
        const size_t idx1 = fv1_feat[i1];
        if (hasMP1[idx1]) continue;
        const OrcKeyPoint& kp1 = kps1[idx1];
        const uint8_t* d1 = desc1 + 32 * idx1;
        int bestDist = TH_LOW;
        int bestIdx2 = -1;
        for (uint32_t i2 = fv2_off[b]; i2 < fv2_off[b + 1]; i2++) {
          size_t idx2 = fv2_feat[i2];
          if (hasMP2[idx2]) continue;
          const int dist = descriptor_distance(d1, desc2 + 32 * idx2);
          if (dist > TH_LOW || dist > bestDist) continue;
          const OrcKeyPoint& kp2 = kps2[idx2];
          const float distex = ex - kp2.x;
          const float distey = ey - kp2.y;
          if (distex * distex + distey * distey < 100 * scale2[kp2.octave]) continue;
          // CheckDistEpipolarLine
          const float la = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
          const float lb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
          const float lc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
          const float num = la * kp2.x + lb * kp2.y + lc;
          const float den = la * la + lb * lb;
          if (den == 0) continue;
          const float dsqr = num * num / den;
          if (dsqr < 3.84 * sigma2[kp2.octave]) { bestIdx2 = (int)idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (checkOri) {
            float rot = kp1.angle - kps2[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back((int)idx1);
          }
        }
      }
      a++; b++;
    } else if (fv1_nodes[a] < fv2_nodes[b]) {
      a = (int)(std::lower_bound(fv1_nodes, fv1_nodes + nfv1, fv2_nodes[b]) - fv1_nodes);
    } else {
      b = (int)(std::lower_bound(fv2_nodes, fv2_nodes + nfv2, fv1_nodes[a]) - fv2_nodes);
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  int np = 0;
  for (int i = 0; i < n1; i++) {
    if (vMatches12[i] < 0) continue;
    pairs[2 * np] = i;
    pairs[2 * np + 1] = vMatches12[i];
    np++;
  }
  return nmatches;
}

// The projected best-match loop shared by ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)
// (ORBmatcher.cc:285-398, loop :357-392), ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (:806-939, loop :872-936),
// ORBmatcher::Fuse(KeyFrame*, Scw, ...) (:941-1064, loop :1014-1050) and both directions of SearchBySim3 (:1066-1290),
// from KeyFrame::GetFeaturesInArea (KeyFrame.cc:637-676) onwards.  Per source (a projected MapPoint): window
// (u, v, radius), predicted level p => keypoint levels [p-1, p], descriptor.  Variants:
//   kp_skip / claim : vpMatched[idx] skips a keypoint and an accepted one is claimed (SearchByProjection, :366-367,:388);
//   inv_sigma2 + chi2: Fuse's reprojection gate, `e2*mvInvLevelSigma2[kpLevel] > 5.99` => skip (:896-903);
//   max_dist         : TH_LOW (50) or TH_HIGH (100).
// best_idx[i] = accepted keypoint or -1, best_dist[i] = its distance.  Returns the number accepted.
int orc_search_projected(const OrcKeyPoint* kpsUn, const uint8_t* desc, int n, const float bounds[4], int n_src,
                         const float* src_uv, const float* src_radius, const int32_t* src_level,
                         const uint8_t* src_valid, const uint8_t* src_desc, const uint8_t* kp_skip, int claim,
                         const float* inv_sigma2, double chi2, int max_dist, int32_t* best_idx, int32_t* best_dist) {
  std::vector<uint8_t> vpMatched(n, 0);
  if (kp_skip) vpMatched.assign(kp_skip, kp_skip + n);
  FrameGrid grid(kpsUn, n, bounds);
  int nmatches = 0;
  for (int i = 0; i < n_src; i++) {
    best_idx[i] = -1;
    best_dist[i] = -1;
    if (!src_valid[i]) continue;
    const float u = src_uv[2 * i], v = src_uv[2 * i + 1];
    const int nPredictedLevel = src_level[i];
    const std::vector<size_t> vIndices = grid.getFeaturesInArea(u, v, src_radius[i], -1, -1);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = src_desc + 32 * (size_t)i;
    int bestDist = INT_MAX;
    int bestIdx = -1;
    for (size_t idx : vIndices) {
      if ((kp_skip || claim) && vpMatched[idx]) continue;
      const OrcKeyPoint& kp = kpsUn[idx];
      const int kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (inv_sigma2) {
        const float ex = u - kp.x;
        const float ey = v - kp.y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_sigma2[kpLevel] > chi2) continue;
      }
      const int dist = descriptor_distance(dMP, desc + 32 * idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    if (bestDist <= max_dist) {
      best_idx[i] = bestIdx;
      best_dist[i] = bestDist;
      if (claim) vpMatched[bestIdx] = 1;
      nmatches++;
    }
  }
  return nmatches;
}

}  // extern "C"

// =============================================================================================
// Whole-function restatements of the pose-driven searches:
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th)            ORBmatcher.cc:1292-1423
// ORBmatcher::SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist) ORBmatcher.cc:1425-1552
// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched) ORBmatcher.cc:285-398
// ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th)                        ORBmatcher.cc:806-939
// ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)      ORBmatcher.cc:941-1064
// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12)    ORBmatcher.cc:1066-1290
// Frame::UndistortKeyPoints (pinhole branch) / ComputeImageBounds     Frame.cc:286-353
//
// The reference does the projections with cv::Mat expressions; what those expressions compute is OpenCV arithmetic
// (RECALLED from OpenCV 4.x core/src/matmul.simd.hpp, matrix_expressions.cpp, norm.cpp -- parity unpinned like the rest):
//   A*b + c, A 3x3, b 3x1 (gemm, flags 0, len 3: small-matrix path): t = a0*b0 + a1*b1 + a2*b2 in FLOAT, left to
//                          right; d = (float)((double)t*alpha + (double)c*beta)
//   -A.t()*b              (gemm with GEMM_1_T: generic GEMMSingleMul<float,double>): double products and sums,
//                          d = (float)(s*alpha)
//   cv::norm(v)           sqrt of a double sum of double squares (returned as double)
//   a.dot(b)              double sum of double products
//   M / s, s * M          convertTo with a float scale: m * (float)(1.0/s) resp. m * (float)s
// MapPoint / KeyFrame bookkeeping (Replace, AddObservation, AddMapPoint) follows the simplified model of
// orb_oracle_pose.h; the search part (which keypoint each point selects) is exact reference logic.
// =============================================================================================
#define ORB_ORACLE_IMPLEMENTATION
#include "orb_oracle_pose.h"

namespace {
static_assert(sizeof(OrcKp) == sizeof(OrcKeyPoint), "layout");

void cvGemm3(const float A[9], const float b[3], double alpha, const float* c, double beta, float d[3]) {
  for (int i = 0; i < 3; i++) {
    const float t = A[3 * i] * b[0] + A[3 * i + 1] * b[1] + A[3 * i + 2] * b[2];
    d[i] = (float)((double)t * alpha + (double)(c ? c[i] : 0.f) * beta);
  }
}
void cvGemmT3(const float A[9], const float b[3], double alpha, float d[3]) {   // alpha * A^T * b
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)A[3 * k + i] * (double)b[k];
    d[i] = (float)(s * alpha);
  }
}
double cvNorm3(const float v[3]) {
  double s = 0;
  for (int k = 0; k < 3; k++) s += (double)v[k] * (double)v[k];
  return std::sqrt(s);
}
double cvDot3(const float a[3], const float b[3]) {
  double r = 0;
  for (int k = 0; k < 3; k++) r += (double)a[k] * (double)b[k];
  return r;
}
void poseRt(const float T[16], float R[9], float t[3]) {   // rowRange(0,3).colRange(0,3) / .col(3)
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
    t[r] = T[4 * r + 3];
  }
}
// Scw -> Rcw, tcw, Ow  (ORBmatcher.cc:293-298 / 949-954)
void decomposeScw(const float S[16], float Rcw[9], float tcw[3], float Ow[3]) {
  float sR[9], st[3];
  poseRt(S, sR, st);
  const float scw = (float)std::sqrt(cvDot3(sR, sR));          // sqrt(sRcw.row(0).dot(sRcw.row(0)))
  const float inv = (float)(1.0 / (double)scw);
  for (int i = 0; i < 9; i++) Rcw[i] = sR[i] * inv;           // sRcw/scw
  for (int i = 0; i < 3; i++) tcw[i] = st[i] * inv;           // Scw...col(3)/scw
  cvGemmT3(Rcw, tcw, -1.0, Ow);                                // -Rcw.t()*tcw
}
// MapPoint::GetMinDistanceInvariance / GetMaxDistanceInvariance (MapPoint.cc:358-368): 0.8f / 1.2f times the RAW fields;
// MapPoint::PredictScale (MapPoint.cc:370-379) divides the raw mfMaxDistance -- two different values, kept apart here.
inline float minDistanceInvariance(const OrcPoints* P, int i) { return 0.8f * P->mfMinDistance[i]; }
inline float maxDistanceInvariance(const OrcPoints* P, int i) { return 1.2f * P->mfMaxDistance[i]; }
int predictScale(float mfMaxDistance, float currentDist, float logScaleFactor) {   // MapPoint.cc:370-379
  const float ratio = mfMaxDistance / currentDist;
  return (int)ceilf(logf(ratio) / logScaleFactor);
}
bool isInImage(const OrcView* v, float x, float y) {   // KeyFrame.cc:678-681
  return x >= v->bounds[0] && x < v->bounds[1] && y >= v->bounds[2] && y < v->bounds[3];
}
// the simplified bookkeeping model (orb_oracle_pose.h)
void addObservation(OrcPoints* P, int id, int idx) {
  if (P->idxInKF[id] >= 0) return;
  P->idxInKF[id] = idx;
  P->nObs[id]++;
}
void replacePoint(OrcPoints* P, int32_t* slot, int a, int b) {   // a->Replace(b)
  if (a == b) return;
  P->bad[a] = 1;
  const int ia = P->idxInKF[a];
  P->nObs[b] += P->nObs[a] - (ia >= 0 ? 1 : 0);   // observations in other keyframes move to b
  P->nObs[a] = 0;
  P->idxInKF[a] = -1;
  if (ia >= 0) {
    if (P->idxInKF[b] < 0) { slot[ia] = b; addObservation(P, b, ia); }
    else slot[ia] = -1;
  }
}
// best keypoint of a KeyFrame window (the loops at :357-386, 888-916, 1026-1045, 1164-1185): levels [p-1, p]
int bestInWindow(const FrameGrid& g, const OrcView* v, float u, float vv, float radius, int nPredictedLevel, const uint8_t* dMP,
                 const int32_t* skipIfSet, bool chi2, int initBest, int& bestDist) {
  const std::vector<size_t> vIndices = g.getFeaturesInArea(u, vv, radius, -1, -1);   // KeyFrame::GetFeaturesInArea: no level filter
  bestDist = initBest;
  int bestIdx = -1;
  for (size_t idx : vIndices) {
    if (skipIfSet && skipIfSet[idx] >= 0) continue;
    const OrcKeyPoint& kp = reinterpret_cast<const OrcKeyPoint*>(v->kpsUn)[idx];
    const int kpLevel = kp.octave;
    if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
    if (chi2) {
      const float ex = u - kp.x, ey = vv - kp.y;
      const float e2 = ex * ex + ey * ey;
      if (e2 * v->invLevelSigma2[kpLevel] > 5.99) continue;
    }
    const int dist = descriptor_distance(dMP, v->desc + 32 * idx);
    if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
  }
  return bestIdx;
}
}  // namespace

extern "C" {

int orc_sbp_frame(const OrcView* cur, const float Tcw[16], const OrcKp* lastKeys, const OrcKp* lastKeysUn, int nLast,
                  const int32_t* last_mp, const uint8_t* last_outlier, OrcPoints* P, int32_t* cur_mp, float th,
                  int check_orientation) {
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  float Rcw[9], tcw[3];
  poseRt(Tcw, Rcw, tcw);
  const OrcKeyPoint* kpsUn = reinterpret_cast<const OrcKeyPoint*>(cur->kpsUn);
  FrameGrid F(kpsUn, cur->n, cur->bounds);
  for (int i = 0; i < nLast; i++) {
    const int pMP = last_mp[i];
    if (pMP < 0) continue;
    if (last_outlier[i]) continue;
    float x3Dc[3];
    cvGemm3(Rcw, P->pos + 3 * pMP, 1.0, tcw, 1.0, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    float u = cur->fx * xc * invzc + cur->cx;
    float v = cur->fy * yc * invzc + cur->cy;
    if (u < cur->bounds[0] || u > cur->bounds[1]) continue;
    if (v < cur->bounds[2] || v > cur->bounds[3]) continue;
    const int nLastOctave = lastKeys[i].octave;
    const float radius = th * cur->scaleFactors[nLastOctave];
    std::vector<size_t> vIndices2 = F.getFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = P->desc + 32 * (size_t)pMP;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t i2 : vIndices2) {
      if (cur_mp[i2] >= 0)
        if (P->nObs[cur_mp[i2]] > 0) continue;
      const int dist = descriptor_distance(dMP, cur->desc + 32 * i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= TH_HIGH) {
      cur_mp[bestIdx2] = pMP;
      nmatches++;
      if (check_orientation) rotHist[rot_bin(lastKeysUn[i].angle, kpsUn[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
  }
  return nmatches;
}

int orc_sbp_keyframe(const OrcView* cur, const float Tcw[16], const OrcKp* kfKeysUn, int nKF, const int32_t* kf_mp,
                     const uint8_t* already, OrcPoints* P, int32_t* cur_mp, float th, int ORBdist, int check_orientation) {
  int nmatches = 0;
  float Rcw[9], tcw[3], Ow[3];
  poseRt(Tcw, Rcw, tcw);
  cvGemmT3(Rcw, tcw, -1.0, Ow);
  std::vector<int> rotHist[HISTO_LENGTH];
  const OrcKeyPoint* kpsUn = reinterpret_cast<const OrcKeyPoint*>(cur->kpsUn);
  FrameGrid F(kpsUn, cur->n, cur->bounds);
  for (int i = 0; i < nKF; i++) {
    const int pMP = kf_mp[i];
    if (pMP < 0) continue;
    if (P->bad[pMP] || already[pMP]) continue;
    const float* x3Dw = P->pos + 3 * pMP;
    float x3Dc[3];
    cvGemm3(Rcw, x3Dw, 1.0, tcw, 1.0, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / x3Dc[2]);
    const float u = cur->fx * xc * invzc + cur->cx;
    const float v = cur->fy * yc * invzc + cur->cy;
    if (u < cur->bounds[0] || u > cur->bounds[1]) continue;
    if (v < cur->bounds[2] || v > cur->bounds[3]) continue;
    const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
    float dist3D = (float)cvNorm3(PO);
    const float maxDistance = maxDistanceInvariance(P, pMP), minDistance = minDistanceInvariance(P, pMP);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    int nPredictedLevel = predictScale(P->mfMaxDistance[pMP], dist3D, cur->logScaleFactor);
    const float radius = th * cur->scaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices2 = F.getFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = P->desc + 32 * (size_t)pMP;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t i2 : vIndices2) {
      if (cur_mp[i2] >= 0) continue;
      const int dist = descriptor_distance(dMP, cur->desc + 32 * i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= ORBdist) {
      cur_mp[bestIdx2] = pMP;
      nmatches++;
      if (check_orientation) rotHist[rot_bin(kfKeysUn[i].angle, kpsUn[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
  }
  return nmatches;
}

int orc_sbp_scw(const OrcView* kf, const float Scw[16], const int32_t* points, int npoints, OrcPoints* P,
                int32_t* vpMatched, int th) {
  float Rcw[9], tcw[3], Ow[3];
  decomposeScw(Scw, Rcw, tcw, Ow);
  std::vector<uint8_t> spAlreadyFound(P->M, 0);
  for (int i = 0; i < kf->n; i++)
    if (vpMatched[i] >= 0) spAlreadyFound[vpMatched[i]] = 1;
  FrameGrid G(reinterpret_cast<const OrcKeyPoint*>(kf->kpsUn), kf->n, kf->bounds);
  int nmatches = 0;
  for (int iMP = 0; iMP < npoints; iMP++) {
    const int pMP = points[iMP];
    if (P->bad[pMP] || spAlreadyFound[pMP]) continue;
    const float* p3Dw = P->pos + 3 * pMP;
    float p3Dc[3];
    cvGemm3(Rcw, p3Dw, 1.0, tcw, 1.0, p3Dc);
    if (p3Dc[2] < 0.0) continue;
    const float invz = 1 / p3Dc[2];
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    const float u = kf->fx * x + kf->cx, v = kf->fy * y + kf->cy;
    if (!isInImage(kf, u, v)) continue;
    const float maxDistance = maxDistanceInvariance(P, pMP), minDistance = minDistanceInvariance(P, pMP);
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist = (float)cvNorm3(PO);
    if (dist < minDistance || dist > maxDistance) continue;
    if (cvDot3(PO, P->normal + 3 * pMP) < 0.5 * dist) continue;
    int nPredictedLevel = predictScale(P->mfMaxDistance[pMP], dist, kf->logScaleFactor);
    const float radius = th * kf->scaleFactors[nPredictedLevel];
    int bestDist;
    const int bestIdx = bestInWindow(G, kf, u, v, radius, nPredictedLevel, P->desc + 32 * (size_t)pMP, vpMatched, false, 256, bestDist);
    if (bestDist <= TH_LOW) { vpMatched[bestIdx] = pMP; nmatches++; }
  }
  return nmatches;
}

int orc_fuse(const OrcView* kf, const float Tcw[16], const int32_t* cand, int ncand, OrcPoints* P, int32_t* slot, float th) {
  float Rcw[9], tcw[3], Ow[3];
  poseRt(Tcw, Rcw, tcw);
  {   // pKF->GetCameraCenter(): KeyFrame::SetPose computes Ow = -Rwc*tcw with Rwc = Rcw.t() already evaluated (KeyFrame.cc:93-97)
    float Rwc[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rwc[3 * r + c] = Rcw[3 * c + r];
    cvGemm3(Rwc, tcw, -1.0, nullptr, 0.0, Ow);
  }
  FrameGrid G(reinterpret_cast<const OrcKeyPoint*>(kf->kpsUn), kf->n, kf->bounds);
  int nFused = 0;
  for (int i = 0; i < ncand; i++) {
    const int pMP = cand[i];
    if (pMP < 0) continue;
    if (P->bad[pMP] || P->idxInKF[pMP] >= 0) continue;
    const float* p3Dw = P->pos + 3 * pMP;
    float p3Dc[3];
    cvGemm3(Rcw, p3Dw, 1.0, tcw, 1.0, p3Dc);
    if (p3Dc[2] < 0.0f) continue;
    const float invz = 1 / p3Dc[2];
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    const float u = kf->fx * x + kf->cx, v = kf->fy * y + kf->cy;
    if (!isInImage(kf, u, v)) continue;
    const float maxDistance = maxDistanceInvariance(P, pMP), minDistance = minDistanceInvariance(P, pMP);
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist3D = (float)cvNorm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    if (cvDot3(PO, P->normal + 3 * pMP) < 0.5 * dist3D) continue;
    int nPredictedLevel = predictScale(P->mfMaxDistance[pMP], dist3D, kf->logScaleFactor);
    const float radius = th * kf->scaleFactors[nPredictedLevel];
    int bestDist;
    const int bestIdx = bestInWindow(G, kf, u, v, radius, nPredictedLevel, P->desc + 32 * (size_t)pMP, nullptr, true, 256, bestDist);
    if (bestDist <= TH_LOW) {
      const int pMPinKF = slot[bestIdx];
      if (pMPinKF >= 0) {
        if (!P->bad[pMPinKF]) {
          if (P->nObs[pMPinKF] > P->nObs[pMP]) replacePoint(P, slot, pMP, pMPinKF);
          else replacePoint(P, slot, pMPinKF, pMP);
        }
      } else {
        addObservation(P, pMP, bestIdx);
        slot[bestIdx] = pMP;
      }
      nFused++;
    }
  }
  return nFused;
}

int orc_fuse_scw(const OrcView* kf, const float Scw[16], const int32_t* points, int npoints, OrcPoints* P, int32_t* slot,
                 float th, int32_t* replace_out) {
  float Rcw[9], tcw[3], Ow[3];
  decomposeScw(Scw, Rcw, tcw, Ow);
  std::vector<uint8_t> spAlreadyFound(P->M, 0);   // pKF->GetMapPoints(): non-NULL, not bad
  for (int i = 0; i < kf->n; i++)
    if (slot[i] >= 0 && !P->bad[slot[i]]) spAlreadyFound[slot[i]] = 1;
  FrameGrid G(reinterpret_cast<const OrcKeyPoint*>(kf->kpsUn), kf->n, kf->bounds);
  int nFused = 0;
  for (int iMP = 0; iMP < npoints; iMP++) {
    const int pMP = points[iMP];
    if (P->bad[pMP] || spAlreadyFound[pMP]) continue;
    const float* p3Dw = P->pos + 3 * pMP;
    float p3Dc[3];
    cvGemm3(Rcw, p3Dw, 1.0, tcw, 1.0, p3Dc);
    if (p3Dc[2] < 0.0f) continue;
    const float invz = (float)(1.0 / p3Dc[2]);
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    const float u = kf->fx * x + kf->cx, v = kf->fy * y + kf->cy;
    if (!isInImage(kf, u, v)) continue;
    const float maxDistance = maxDistanceInvariance(P, pMP), minDistance = minDistanceInvariance(P, pMP);
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist3D = (float)cvNorm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    if (cvDot3(PO, P->normal + 3 * pMP) < 0.5 * dist3D) continue;
    const int nPredictedLevel = predictScale(P->mfMaxDistance[pMP], dist3D, kf->logScaleFactor);
    const float radius = th * kf->scaleFactors[nPredictedLevel];
    int bestDist;
    const int bestIdx = bestInWindow(G, kf, u, v, radius, nPredictedLevel, P->desc + 32 * (size_t)pMP, nullptr, false, INT_MAX, bestDist);
    if (bestDist <= TH_LOW) {
      const int pMPinKF = slot[bestIdx];
      if (pMPinKF >= 0) {
        if (!P->bad[pMPinKF]) replace_out[iMP] = pMPinKF;
      } else {
        addObservation(P, pMP, bestIdx);
        slot[bestIdx] = pMP;
      }
      nFused++;
    }
  }
  return nFused;
}

int orc_search_by_sim3(const OrcView* kf1, const float T1w[16], const int32_t* mp1, const OrcView* kf2, const float T2w[16],
                       const int32_t* mp2, OrcPoints* P, int32_t* matches12, float s12, const float R12[9],
                       const float t12[3], float th) {
  float R1w[9], t1w[3], R2w[9], t2w[3];
  poseRt(T1w, R1w, t1w);
  poseRt(T2w, R2w, t2w);
  float sR12[9], sR21[9], t21[3];
  for (int i = 0; i < 9; i++) sR12[i] = R12[i] * s12;                       // s12*R12
  const float is = (float)(1.0 / (double)s12);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) sR21[3 * r + c] = R12[3 * c + r] * is;   // (1.0/s12)*R12.t()
  cvGemm3(sR21, t12, -1.0, nullptr, 0.0, t21);                               // -sR21*t12
  const int N1 = kf1->n, N2 = kf2->n;
  std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
  for (int i = 0; i < N1; i++) {
    const int pMP = matches12[i];
    if (pMP >= 0) {
      vbAlreadyMatched1[i] = true;
      const int idx2 = P->idxInKF[pMP];
      if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
    }
  }
  std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
  FrameGrid G1(reinterpret_cast<const OrcKeyPoint*>(kf1->kpsUn), N1, kf1->bounds), G2(reinterpret_cast<const OrcKeyPoint*>(kf2->kpsUn), N2, kf2->bounds);
  for (int i1 = 0; i1 < N1; i1++) {
    const int pMP = mp1[i1];
    if (pMP < 0 || vbAlreadyMatched1[i1]) continue;
    if (P->bad[pMP]) continue;
    float p3Dc1[3], p3Dc2[3];
    cvGemm3(R1w, P->pos + 3 * pMP, 1.0, t1w, 1.0, p3Dc1);
    cvGemm3(sR21, p3Dc1, 1.0, t21, 1.0, p3Dc2);
    if (p3Dc2[2] < 0.0) continue;
    const float invz = (float)(1.0 / p3Dc2[2]);
    const float x = p3Dc2[0] * invz, y = p3Dc2[1] * invz;
    const float u = kf1->fx * x + kf1->cx, v = kf1->fy * y + kf1->cy;   // fx.. of pKF1 (:1069-1072)
    if (!isInImage(kf2, u, v)) continue;
    const float maxDistance = maxDistanceInvariance(P, pMP), minDistance = minDistanceInvariance(P, pMP);
    const float dist3D = (float)cvNorm3(p3Dc2);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = predictScale(P->mfMaxDistance[pMP], dist3D, kf2->logScaleFactor);
    const float radius = th * kf2->scaleFactors[nPredictedLevel];
    int bestDist;
    const int bestIdx = bestInWindow(G2, kf2, u, v, radius, nPredictedLevel, P->desc + 32 * (size_t)pMP, nullptr, false, INT_MAX, bestDist);
    if (bestDist <= TH_HIGH) vnMatch1[i1] = bestIdx;
  }
  for (int i2 = 0; i2 < N2; i2++) {
    const int pMP = mp2[i2];
    if (pMP < 0 || vbAlreadyMatched2[i2]) continue;
    if (P->bad[pMP]) continue;
    float p3Dc2[3], p3Dc1[3];
    cvGemm3(R2w, P->pos + 3 * pMP, 1.0, t2w, 1.0, p3Dc2);
    cvGemm3(sR12, p3Dc2, 1.0, t12, 1.0, p3Dc1);
    if (p3Dc1[2] < 0.0) continue;
    const float invz = (float)(1.0 / p3Dc1[2]);
    const float x = p3Dc1[0] * invz, y = p3Dc1[1] * invz;
    const float u = kf1->fx * x + kf1->cx, v = kf1->fy * y + kf1->cy;
    if (!isInImage(kf1, u, v)) continue;
    const float maxDistance = maxDistanceInvariance(P, pMP), minDistance = minDistanceInvariance(P, pMP);
    const float dist3D = (float)cvNorm3(p3Dc1);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = predictScale(P->mfMaxDistance[pMP], dist3D, kf1->logScaleFactor);
    const float radius = th * kf1->scaleFactors[nPredictedLevel];
    int bestDist;
    const int bestIdx = bestInWindow(G1, kf1, u, v, radius, nPredictedLevel, P->desc + 32 * (size_t)pMP, nullptr, false, INT_MAX, bestDist);
    if (bestDist <= TH_HIGH) vnMatch2[i2] = bestIdx;
  }
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0) {
      const int idx1 = vnMatch2[idx2];
      if (idx1 == i1) { matches12[i1] = mp2[idx2]; nFound++; }
    }
  }
  return nFound;
}

// cv::undistortPoints with R = empty, P = K (OpenCV 4.x calib3d/imgproc undistort.dispatch.cpp, cvUndistortPointsInternal;
// RECALLED): x = (u-cx)/fx, y = (v-cy)/fy in double, 5 fixed-point iterations of the inverse distortion (default
// TermCriteria(COUNT, 5, 0.01)), then u' = fx*x + cx.  k = {k1,k2,p1,p2,k3,k4,k5,k6}, missing ones 0.
void orc_undistort_pinhole(float* xy, int n, float fxf, float fyf, float cxf, float cyf, const float* dist, int ndist) {
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < ndist && i < 8; i++) k[i] = dist[i];
  const double fx = fxf, fy = fyf, cx = cxf, cy = cyf, ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; i++) {
    double x = xy[2 * i], y = xy[2 * i + 1];
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {   // test: undistortPoints.regression_14583
        x = (u - cx) * ifx;
        y = (v - cy) * ify;
        break;
      }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    // RR = P (= K): xx = fx*x + 0*y + cx, yy = fy*y + cy, ww = 1./1
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    xy[2 * i] = (float)(xx * ww);
    xy[2 * i + 1] = (float)(yy * ww);
  }
}

void orc_image_bounds(int cols, int rows, int mode, float fx, float fy, float cx, float cy, const float* dist, int ndist,
                      float bounds[4]) {
  if (mode != 0 || (ndist > 0 && dist[0] != 0.0)) {
    float mat[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
    if (mode) orc_undistort_equidistant(mat, 4, fx, fy, cx, cy);
    else orc_undistort_pinhole(mat, 4, fx, fy, cx, cy, dist, ndist);
    bounds[0] = std::min(mat[0], mat[4]);
    bounds[1] = std::max(mat[2], mat[6]);
    bounds[2] = std::min(mat[1], mat[3]);
    bounds[3] = std::max(mat[5], mat[7]);
  } else {
    bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
  }
}

// The small-matrix arithmetic of the pose-driven searches, exported for the live-OpenCV cross-check (tests/test_opencv_live.py):
// op 0 gemm3 (alpha * A * b + beta * c; c may be NULL), 1 gemmT3 (alpha * A^T * b), 2 norm3 (out[0], as double), 3 dot3.
void orc_cv_small(int op, const float* A, const float* b, double alpha, const float* c, double beta, float* out3, double* out1) {
  if (op == 0) cvGemm3(A, b, alpha, c, beta, out3);
  else if (op == 1) cvGemmT3(A, b, alpha, out3);
  else if (op == 2) *out1 = cvNorm3(b);
  else *out1 = cvDot3(A, b);
}

}  // extern "C"

