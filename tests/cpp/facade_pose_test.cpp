// facade_pose_test.cpp -- the pose-driven ORBmatcher functions of include/orbfe/orb_shim.hpp (GPU, through the C ABI)
// against the oracle's whole-function restatements (oracle/orb_oracle_pose.h), on seeded synthetic scenes:
//   SearchByProjection(Frame&, const Frame&, th)                 ORBmatcher.cc:1292-1423
//   SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist)      ORBmatcher.cc:1425-1552
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)  ORBmatcher.cc:285-398
//   Fuse(KeyFrame*, vpMapPoints, th)                             ORBmatcher.cc:806-939
//   Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)           ORBmatcher.cc:941-1064
//   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)     ORBmatcher.cc:1066-1290
// The stand-in Frame / KeyFrame / MapPoint types expose the member names the reference code (and therefore the shim)
// uses.  MapPoint::Replace / AddObservation / KeyFrame::AddMapPoint follow a SIMPLIFIED MODEL, the same one the oracle
// implements (orb_oracle_pose.h): a point knows its slot in the ONE keyframe under test (idxInKF) and a total
// observation count; a.Replace(b) marks a bad, moves a's observations in other keyframes to b and hands a's slot in the
// keyframe under test to b unless b already sits in that keyframe (then the slot is erased) -- MapPoint.cc:158-198.
// Prints one line per function; exit code 0 iff every comparison is exact.
//   build: g++ -std=c++17 -O1 -Iinclude -Ioracle tests/cpp/facade_pose_test.cpp os1_amd/liborbfe.so oracle/liborb_oracle.so
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "orb_oracle_pose.h"
#include "orbfe/orb_shim.hpp"

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };  // cv::KeyPoint layout
struct MatF {      // the parts of cv::Mat the shim touches: at<float>(r,c) for small float matrices, data/step/rows for descriptors
  float v[16] = {0};
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;
  template <class T> T at(int r, int c) const { return (T)v[r * cols + c]; }
};
static MatF vecMat(const float* p) { MatF m; m.rows = 3; m.cols = 1; memcpy(m.v, p, 12); return m; }

struct KeyFrame;
struct MapPoint {
  int id = 0;
  float pos[3], normal[3], mfMinDistance = 0, mfMaxDistance = 0;   // the raw fields of MapPoint.h
  unsigned char desc[32];
  bool bad = false;
  int nObs = 0, idxInKF = -1;
  KeyFrame* kf = nullptr;      // the keyframe under test (the model tracks one)
  MatF GetWorldPos() { return vecMat(pos); }
  MatF GetNormal() { return vecMat(normal); }
  MatF GetDescriptor() { MatF m; m.data = desc; m.step = 32; m.rows = 1; return m; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }   // MapPoint.cc:364-368
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }   // MapPoint.cc:358-362
  int PredictScale(const float& currentDist, const float& logScaleFactor) {   // MapPoint.cc:370-379
    const float ratio = mfMaxDistance / currentDist;                      // the raw field, not the bound
    return (int)std::ceil(std::log(ratio) / logScaleFactor);
  }
  bool isBad() { return bad; }
  int Observations() { return nObs; }
  bool IsInKeyFrame(KeyFrame*) { return idxInKF >= 0; }
  int GetIndexInKeyFrame(KeyFrame*) { return idxInKF; }
  void AddObservation(KeyFrame*, size_t idx) {
    if (idxInKF >= 0) return;
    idxInKF = (int)idx;
    nObs++;
  }
  void Replace(MapPoint* pMP);
};
static unsigned long g_nextId = 0;
struct FrameBase {
  unsigned long mnId = g_nextId++;   // Frame.cc:78 / KeyFrame.cc:45: unique per constructed object, kept by copies
  int N = 0;
  std::vector<KeyPoint> mvKeys, mvKeysUn;
  std::vector<unsigned char> descStore;
  MatF mDescriptors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<float> mvScaleFactors, mvInvLevelSigma2;
  float mfLogScaleFactor = 0, fx = 0, fy = 0, cx = 0, cy = 0, mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
  MatF mTcw;
};
struct Frame : FrameBase {};
struct KeyFrame : FrameBase {
  float Ow[3];
  MatF GetRotation() { MatF m; m.rows = m.cols = 3; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m.v[3 * r + c] = mTcw.v[4 * r + c]; return m; }
  MatF GetTranslation() { MatF m; m.rows = 3; m.cols = 1; for (int r = 0; r < 3; r++) m.v[r] = mTcw.v[4 * r + 3]; return m; }
  MatF GetCameraCenter() { return vecMat(Ow); }
  bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
  MapPoint* GetMapPoint(size_t i) { return mvpMapPoints[i]; }
  void AddMapPoint(MapPoint* p, size_t i) { mvpMapPoints[i] = p; }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  std::set<MapPoint*> GetMapPoints() {
    std::set<MapPoint*> s;
    for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
    return s;
  }
};
void MapPoint::Replace(MapPoint* pMP) {
  if (pMP->id == id) return;
  bad = true;
  const int ia = idxInKF;
  pMP->nObs += nObs - (ia >= 0 ? 1 : 0);
  nObs = 0;
  idxInKF = -1;
  if (ia >= 0) {
    if (!pMP->IsInKeyFrame(kf)) { kf->mvpMapPoints[ia] = pMP; pMP->AddObservation(kf, ia); }
    else kf->mvpMapPoints[ia] = nullptr;
  }
}

// ---- seeded scene generator --------------------------------------------------------------------------------------
struct Rng {
  unsigned long long s;
  explicit Rng(unsigned long long seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  unsigned long long next() { s += 0x9E3779B97F4A7C15ull; unsigned long long z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
  int below(int n) { return (int)(next() % (unsigned long long)n); }
};
static const int W = 640, H = 480, NLEV = 8;

static void makePose(Rng& r, float scale, float T[16]) {   // T = scale * [R | t] (row major 4x4, last row 0 0 0 1)
  const double wx = r.uni(-0.08, 0.08), wy = r.uni(-0.08, 0.08), wz = r.uni(-0.08, 0.08);
  const double th = std::sqrt(wx * wx + wy * wy + wz * wz) + 1e-12, c = std::cos(th), s = std::sin(th), k = 1 - c;
  const double x = wx / th, y = wy / th, z = wz / th;
  const double R[9] = {c + x * x * k, x * y * k - z * s, x * z * k + y * s, y * x * k + z * s, c + y * y * k, y * z * k - x * s,
                       z * x * k - y * s, z * y * k + x * s, c + z * z * k};
  const double t[3] = {r.uni(-0.4, 0.4), r.uni(-0.4, 0.4), r.uni(-0.4, 0.4)};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[4 * i + j] = (float)(scale * R[3 * i + j]);
    T[4 * i + 3] = (float)(scale * t[i]);
  }
  T[12] = T[13] = T[14] = 0; T[15] = 1;
}
template <class F>
static void makeView(Rng& r, F& f, int n) {
  f.N = n;
  f.mvKeys.resize(n);
  f.descStore.resize((size_t)n * 32);
  for (int i = 0; i < n; i++) {
    KeyPoint& k = f.mvKeys[i];
    k.x = (float)r.uni(15, W - 15); k.y = (float)r.uni(15, H - 15);
    const double q = r.uni();
    k.octave = q < 0.35 ? 0 : q < 0.6 ? 1 : q < 0.75 ? 2 : 3 + r.below(5);
    k.size = 31.f; k.angle = (float)r.uni(0, 360); k.response = (float)(20 + r.below(200)); k.class_id = -1;
    for (int b = 0; b < 32; b++) f.descStore[(size_t)i * 32 + b] = (unsigned char)r.below(256);
  }
  f.mvKeysUn = f.mvKeys;
  f.mDescriptors.data = f.descStore.data(); f.mDescriptors.step = 32; f.mDescriptors.rows = n;
  f.mvpMapPoints.assign(n, nullptr);
  f.mvbOutlier.assign(n, false);
  f.mvScaleFactors.resize(NLEV); f.mvInvLevelSigma2.resize(NLEV);
  f.mvScaleFactors[0] = 1.f;
  for (int l = 1; l < NLEV; l++) f.mvScaleFactors[l] = f.mvScaleFactors[l - 1] * 1.2f;
  for (int l = 0; l < NLEV; l++) f.mvInvLevelSigma2[l] = 1.0f / (f.mvScaleFactors[l] * f.mvScaleFactors[l]);
  f.mfLogScaleFactor = std::log(1.2f);
  f.fx = 520.f; f.fy = 518.f; f.cx = 321.5f; f.cy = 239.25f;
  f.mnMinX = 0; f.mnMaxX = (float)W; f.mnMinY = 0; f.mnMaxY = (float)H;
}
// a MapPoint that a camera with pose T (possibly scaled by `scale`) sees near keypoint k of view f
template <class F>
static void makePoint(Rng& r, MapPoint& p, const F& f, int k, const float T[16], float scale) {
  double R[9], t[3];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = T[4 * i + j] / scale; t[i] = T[4 * i + 3] / scale; }
  const KeyPoint& kp = f.mvKeysUn[k];
  const double z = r.uni(2, 15), u = kp.x + r.uni(-2, 2), v = kp.y + r.uni(-2, 2);
  const double Xc[3] = {(u - f.cx) / f.fx * z - t[0], (v - f.cy) / f.fy * z - t[1], z - t[2]};
  double Ow[3];
  for (int i = 0; i < 3; i++) {
    p.pos[i] = (float)(R[i] * Xc[0] + R[3 + i] * Xc[1] + R[6 + i] * Xc[2]);         // R^T (Xc - t)
    Ow[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
  }
  double PO[3] = {p.pos[0] - Ow[0], p.pos[1] - Ow[1], p.pos[2] - Ow[2]};
  const double dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
  const bool sideways = r.uni() < 0.06;      // viewing-angle rejection (PO.dot(Pn) < 0.5 dist)
  double nn = 0, nv[3];
  for (int i = 0; i < 3; i++) { nv[i] = sideways ? r.uni(-1, 1) : PO[i] / dist + r.uni(-0.3, 0.3); nn += nv[i] * nv[i]; }
  for (int i = 0; i < 3; i++) p.normal[i] = (float)(nv[i] / std::sqrt(nn));
  int L = kp.octave + (r.uni() < 0.3 ? 1 : 0);
  if (L > NLEV - 1) L = NLEV - 1;
  const double q = r.uni();
  // MapPoint::UpdateNormalAndDepth (MapPoint.cc:345-356): mfMaxDistance = dist * scale[level], mfMinDistance = mfMaxDistance / scale[nLevels-1]
  p.mfMaxDistance = (float)(dist * std::pow(1.2, L - 0.35));
  p.mfMinDistance = p.mfMaxDistance / 3.583f;
  if (q < 0.03) p.mfMaxDistance = (float)(dist * 0.9 / 1.2), p.mfMinDistance = p.mfMaxDistance / 3.583f;   // too far: depth outside the invariance region
  const int flips = r.below(36);
  memcpy(p.desc, &f.descStore[(size_t)k * 32], 32);
  for (int b = 0; b < flips; b++) { const int bit = r.below(256); p.desc[bit >> 3] ^= (unsigned char)(1u << (bit & 7)); }
  p.bad = r.uni() < 0.03;
  p.nObs = r.below(6);
}
template <class F>
static OrcView viewOf(const F& f) {
  OrcView v;
  v.kpsUn = reinterpret_cast<const OrcKp*>(f.mvKeysUn.data());
  v.desc = f.descStore.data();
  v.n = f.N;
  v.bounds[0] = f.mnMinX; v.bounds[1] = f.mnMaxX; v.bounds[2] = f.mnMinY; v.bounds[3] = f.mnMaxY;
  v.fx = f.fx; v.fy = f.fy; v.cx = f.cx; v.cy = f.cy;
  v.scaleFactors = f.mvScaleFactors.data(); v.invLevelSigma2 = f.mvInvLevelSigma2.data(); v.nlevels = NLEV;
  v.logScaleFactor = f.mfLogScaleFactor;
  return v;
}
struct Table {   // flat copy of the MapPoints for the oracle
  std::vector<float> pos, normal, minD, maxD;
  std::vector<uint8_t> desc, bad;
  std::vector<int32_t> nObs, idxInKF;
  OrcPoints P;
  explicit Table(const std::vector<MapPoint>& mp) {
    const size_t M = mp.size();
    pos.resize(3 * M); normal.resize(3 * M); minD.resize(M); maxD.resize(M); desc.resize(32 * M); bad.resize(M); nObs.resize(M); idxInKF.resize(M);
    for (size_t i = 0; i < M; i++) {
      memcpy(&pos[3 * i], mp[i].pos, 12); memcpy(&normal[3 * i], mp[i].normal, 12);
      minD[i] = mp[i].mfMinDistance; maxD[i] = mp[i].mfMaxDistance; memcpy(&desc[32 * i], mp[i].desc, 32);
      bad[i] = mp[i].bad; nObs[i] = mp[i].nObs; idxInKF[i] = mp[i].idxInKF;
    }
    P.M = (int)M; P.pos = pos.data(); P.normal = normal.data(); P.mfMinDistance = minD.data(); P.mfMaxDistance = maxD.data();
    P.desc = desc.data(); P.bad = bad.data(); P.nObs = nObs.data(); P.idxInKF = idxInKF.data();
  }
  bool sameState(const std::vector<MapPoint>& mp) const {
    for (size_t i = 0; i < mp.size(); i++)
      if ((bool)bad[i] != mp[i].bad || nObs[i] != mp[i].nObs || idxInKF[i] != mp[i].idxInKF) return false;
    return true;
  }
};
static std::vector<int32_t> idsOf(const std::vector<MapPoint*>& v) {
  std::vector<int32_t> o(v.size());
  for (size_t i = 0; i < v.size(); i++) o[i] = v[i] ? v[i]->id : -1;
  return o;
}
static void setTcw(FrameBase& f, const float T[16]) { f.mTcw.rows = f.mTcw.cols = 4; memcpy(f.mTcw.v, T, 64); }
static void setOw(KeyFrame& kf, float scale) {   // KeyFrame::SetPose: Ow = -Rwc*tcw (float matrices)
  float R[9], t[3], Rwc[9];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = kf.mTcw.v[4 * r + c] / scale; t[r] = kf.mTcw.v[4 * r + 3] / scale; }
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rwc[3 * r + c] = R[3 * c + r];
  orbfe::detail::RestatedOps::gemm3(Rwc, t, -1.0, nullptr, 0.0, kf.Ow);
}

static int failures = 0;
static void report(const char* name, bool ok, int got, int want, int extra) {
  printf("%s %-58s result %d (oracle %d), %d points\n", ok ? "PASS" : "FAIL", name, got, want, extra);
  if (!ok) failures++;
}

int main(int argc, char** argv) {
  const unsigned long long seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 7;
  orbfe::MatcherContext ctx;

  // ---- 1. SearchByProjection(CurrentFrame, LastFrame, th), with and without the orientation check ----------------
  for (int ori = 0; ori < 2; ori++) {
    Rng r(seed * 10 + 1 + ori);
    Frame cur, last;
    makeView(r, cur, 1800);
    makeView(r, last, 1500);
    float T[16];
    makePose(r, 1.f, T);
    setTcw(cur, T);
    std::vector<MapPoint> mp(1500);
    for (int i = 0; i < 1500; i++) {
      mp[i].id = i;
      const int k = r.below(cur.N);
      makePoint(r, mp[i], cur, k, T, 1.f);
      if (r.uni() < 0.7) {
        last.mvpMapPoints[i] = &mp[i];
        last.mvKeys[i].octave = last.mvKeysUn[i].octave = cur.mvKeysUn[k].octave;
        float a = cur.mvKeysUn[k].angle + (r.uni() < 0.85 ? (float)r.uni(-6, 6) : (float)r.uni(0, 360));
        while (a < 0) a += 360.f;
        while (a >= 360.f) a -= 360.f;
        last.mvKeysUn[i].angle = a;
      }
      last.mvbOutlier[i] = r.uni() < 0.05;
    }
    std::vector<MapPoint> prior(60);
    for (int i = 0; i < 60; i++) { prior[i].id = 1500 + i; prior[i].nObs = i % 3; memset(prior[i].pos, 0, 12); memset(prior[i].normal, 0, 12); memset(prior[i].desc, 0, 32); cur.mvpMapPoints[r.below(cur.N)] = &prior[i]; }
    std::vector<MapPoint> all(mp); all.insert(all.end(), prior.begin(), prior.end());
    Table tab(all);
    std::vector<int32_t> cur_mp = idsOf(cur.mvpMapPoints), last_mp = idsOf(last.mvpMapPoints);
    std::vector<uint8_t> outl(last.N);
    for (int i = 0; i < last.N; i++) outl[i] = last.mvbOutlier[i];
    OrcView cv = viewOf(cur);
    const int want = orc_sbp_frame(&cv, T, reinterpret_cast<const OrcKp*>(last.mvKeys.data()), reinterpret_cast<const OrcKp*>(last.mvKeysUn.data()),
                                   last.N, last_mp.data(), outl.data(), &tab.P, cur_mp.data(), 15.f, ori);
    const int got = orbfe::SearchByProjection(ctx, ori != 0, cur, last, 15.f);
    report(ori ? "SearchByProjection(Frame, Frame) + orientation" : "SearchByProjection(Frame, Frame)", got == want && idsOf(cur.mvpMapPoints) == cur_mp && want > 200, got, want, 1500);
  }

  // ---- 2. SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) ---------------------------------------
  {
    Rng r(seed * 10 + 3);
    Frame cur;
    KeyFrame kf;
    makeView(r, cur, 1800);
    makeView(r, kf, 1400);
    float T[16];
    makePose(r, 1.f, T);
    setTcw(cur, T);
    std::vector<MapPoint> mp(1400);
    std::set<MapPoint*> sAlreadyFound;
    for (int i = 0; i < 1400; i++) {
      mp[i].id = i;
      const int k = r.below(cur.N);
      makePoint(r, mp[i], cur, k, T, 1.f);
      if (r.uni() < 0.75) {
        kf.mvpMapPoints[i] = &mp[i];
        float a = cur.mvKeysUn[k].angle + (r.uni() < 0.8 ? (float)r.uni(-5, 5) : (float)r.uni(0, 360));
        while (a < 0) a += 360.f;
        while (a >= 360.f) a -= 360.f;
        kf.mvKeysUn[i].angle = a;
        if (r.uni() < 0.1) sAlreadyFound.insert(&mp[i]);
      }
    }
    for (int i = 0; i < 80; i++) cur.mvpMapPoints[r.below(cur.N)] = &mp[r.below(1400)];
    Table tab(mp);
    std::vector<int32_t> cur_mp = idsOf(cur.mvpMapPoints), kf_mp = idsOf(kf.mvpMapPoints);
    std::vector<uint8_t> already(mp.size(), 0);
    for (MapPoint* p : sAlreadyFound) already[p->id] = 1;
    OrcView cv = viewOf(cur);
    const int want = orc_sbp_keyframe(&cv, T, reinterpret_cast<const OrcKp*>(kf.mvKeysUn.data()), kf.N, kf_mp.data(), already.data(), &tab.P,
                                      cur_mp.data(), 10.f, 100, 1);
    const int got = orbfe::SearchByProjection(ctx, true, cur, &kf, sAlreadyFound, 10.f, 100);
    report("SearchByProjection(Frame, KeyFrame, set, th, ORBdist)", got == want && idsOf(cur.mvpMapPoints) == cur_mp && want > 150, got, want, 1400);
  }

  // ---- 3-5: searches into one KeyFrame ------------------------------------------------------------------------------
  for (int which = 0; which < 3; which++) {
    Rng r(seed * 10 + 4 + which);
    KeyFrame kf;
    makeView(r, kf, 1600);
    const float scale = which == 1 ? 1.f : 1.35f;    // Fuse(KF, points) uses the keyframe's own pose, the others a Sim3
    float T[16];
    makePose(r, scale, T);
    setTcw(kf, T);
    if (which == 1) setOw(kf, 1.f);
    const int M = 1300, E = 500;   // candidates, points already sitting in the keyframe
    std::vector<MapPoint> mp(M + E);
    std::vector<MapPoint*> pts;
    for (int i = 0; i < M + E; i++) { mp[i].id = i; mp[i].kf = &kf; }
    for (int i = 0; i < E; i++) {
      const int k = r.below(kf.N);
      makePoint(r, mp[M + i], kf, k, T, scale);
      if (kf.mvpMapPoints[k]) continue;
      kf.mvpMapPoints[k] = &mp[M + i];
      mp[M + i].idxInKF = k;
      mp[M + i].nObs = 1 + r.below(5);
    }
    for (int i = 0; i < M; i++) {
      makePoint(r, mp[i], kf, r.below(kf.N), T, scale);
      pts.push_back(&mp[i]);
    }
    if (which == 1) { for (int i = 0; i < 40; i++) pts[r.below(M)] = nullptr; for (int i = 0; i < 60; i++) pts.push_back(&mp[M + r.below(E)]); }   // NULLs and points already in the keyframe
    else for (int i = 0; i < 60; i++) pts.push_back(&mp[M + r.below(E)]);
    Table tab(mp);
    OrcView kv = viewOf(kf);
    std::vector<int32_t> ids = idsOf(pts);
    if (which == 0) {
      std::vector<MapPoint*> vpMatched = kf.mvpMapPoints;
      std::vector<int32_t> m = idsOf(vpMatched);
      const int want = orc_sbp_scw(&kv, T, ids.data(), (int)ids.size(), &tab.P, m.data(), 10);
      const int got = orbfe::SearchByProjection(ctx, &kf, kf.mTcw, pts, vpMatched, 10);
      report("SearchByProjection(KeyFrame, Scw, vpPoints, vpMatched, th)", got == want && idsOf(vpMatched) == m && want > 100, got, want, (int)ids.size());
    } else if (which == 1) {
      std::vector<int32_t> slot = idsOf(kf.mvpMapPoints);
      const int want = orc_fuse(&kv, T, ids.data(), (int)ids.size(), &tab.P, slot.data(), 3.0f);
      const int got = orbfe::Fuse(ctx, &kf, pts, 3.0f);
      report("Fuse(KeyFrame, vpMapPoints, th)", got == want && idsOf(kf.mvpMapPoints) == slot && tab.sameState(mp) && want > 100, got, want, (int)ids.size());
    } else {
      std::vector<int32_t> slot = idsOf(kf.mvpMapPoints), rep(ids.size(), -1);
      std::vector<MapPoint*> vpReplacePoint(pts.size(), nullptr);
      const int want = orc_fuse_scw(&kv, T, ids.data(), (int)ids.size(), &tab.P, slot.data(), 4.0f, rep.data());
      const int got = orbfe::Fuse(ctx, &kf, kf.mTcw, pts, 4.0f, vpReplacePoint);
      report("Fuse(KeyFrame, Scw, vpPoints, th, vpReplacePoint)", got == want && idsOf(kf.mvpMapPoints) == slot && idsOf(vpReplacePoint) == rep && tab.sameState(mp) && want > 100,
             got, want, (int)ids.size());
    }
  }

  // ---- 6. SearchBySim3 -----------------------------------------------------------------------------------------------
  {
    Rng r(seed * 10 + 8);
    KeyFrame kf1, kf2;
    makeView(r, kf1, 1500);
    makeView(r, kf2, 1500);
    float T1[16], T2[16];
    makePose(r, 1.f, T1);
    makePose(r, 1.f, T2);
    setTcw(kf1, T1);
    setTcw(kf2, T2);
    // the two keyframes see the same points; the Sim3 between the cameras is the true relative pose, s12 = 1
    float R12[9], t12[3];
    {
      double R1[9], R2[9], t1[3], t2[3];
      for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) { R1[3 * i + j] = T1[4 * i + j]; R2[3 * i + j] = T2[4 * i + j]; } t1[i] = T1[4 * i + 3]; t2[i] = T2[4 * i + 3]; }
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R1[3 * i + k] * R2[3 * j + k]; R12[3 * i + j] = (float)s; }   // R1 R2^T
      }
      for (int i = 0; i < 3; i++) { double s = t1[i]; for (int k = 0; k < 3; k++) s -= (double)R12[3 * i + k] * t2[k]; t12[i] = (float)s; }
    }
    // every shared physical point has two MapPoint objects, mp[i] observed by pKF1 and twin[i] observed by pKF2 (what
    // loop closing looks for); the model's "keyframe under test" is pKF2 (GetIndexInKeyFrame(pKF2), ORBmatcher.cc:1101)
    const int M = 1200;
    std::vector<MapPoint> mp(M), twin(M);
    for (int i = 0; i < M; i++) {
      mp[i].id = i;
      mp[i].kf = &kf2;
      const int k2 = r.below(kf2.N);
      // predicted levels stay inside the pyramid from both cameras (the reference indexes mvScaleFactors unchecked)
      kf2.mvKeys[k2].octave = kf2.mvKeysUn[k2].octave = 2 + kf2.mvKeysUn[k2].octave % 4;
      makePoint(r, mp[i], kf2, k2, T2, 1.f);
      twin[i] = mp[i];
      twin[i].id = M + i;
      double Xc[3];   // where camera 1 sees the point: move a keypoint of kf1 there and give it the point's descriptor
      for (int a = 0; a < 3; a++) Xc[a] = (double)T1[4 * a] * mp[i].pos[0] + (double)T1[4 * a + 1] * mp[i].pos[1] + (double)T1[4 * a + 2] * mp[i].pos[2] + T1[4 * a + 3];
      const int k1 = r.below(kf1.N);
      if (Xc[2] > 0.5 && !kf1.mvpMapPoints[k1] && !kf2.mvpMapPoints[k2]) {
        const double u = kf1.fx * Xc[0] / Xc[2] + kf1.cx, v = kf1.fy * Xc[1] / Xc[2] + kf1.cy;
        if (u > 15 && u < W - 15 && v > 15 && v < H - 15) {
          kf1.mvKeys[k1].x = kf1.mvKeysUn[k1].x = (float)(u + r.uni(-1, 1));
          kf1.mvKeys[k1].y = kf1.mvKeysUn[k1].y = (float)(v + r.uni(-1, 1));
          kf1.mvKeys[k1].octave = kf1.mvKeysUn[k1].octave = kf2.mvKeysUn[k2].octave;
          memcpy(&kf1.descStore[(size_t)k1 * 32], mp[i].desc, 32);
          kf1.mvpMapPoints[k1] = &mp[i];
          if (r.uni() < 0.9) { kf2.mvpMapPoints[k2] = &twin[i]; twin[i].idxInKF = k2; }
        }
      }
    }
    std::vector<MapPoint*> vpMatches12(kf1.N, nullptr);
    for (int k = 0; k < 60; k++) {   // matches found earlier (SearchByBoW in LoopClosing::ComputeSim3)
      const int i1 = r.below(kf1.N);
      if (kf1.mvpMapPoints[i1]) vpMatches12[i1] = &twin[kf1.mvpMapPoints[i1]->id];
    }
    std::vector<MapPoint> all(mp);
    all.insert(all.end(), twin.begin(), twin.end());
    Table tab(all);
    OrcView v1 = viewOf(kf1), v2 = viewOf(kf2);
    std::vector<int32_t> mp1 = idsOf(kf1.mvpMapPoints), mp2 = idsOf(kf2.mvpMapPoints), m12 = idsOf(vpMatches12);
    const float s12 = 1.0f;
    const int want = orc_search_by_sim3(&v1, T1, mp1.data(), &v2, T2, mp2.data(), &tab.P, m12.data(), s12, R12, t12, 7.5f);
    MatF R12m, t12m;
    R12m.rows = R12m.cols = 3; memcpy(R12m.v, R12, 36);
    t12m.rows = 3; t12m.cols = 1; memcpy(t12m.v, t12, 12);
    const int got = orbfe::SearchBySim3(ctx, &kf1, &kf2, vpMatches12, s12, R12m, t12m, 7.5f);
    report("SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)", got == want && idsOf(vpMatches12) == m12 && want > 30, got, want, M);
  }

  // ---- 7. recycled identities: Tracking::Reset() sets Frame::nNextId / KeyFrame::nNextId back to 0 (Tracking.cc:1159-1160) and
  // Osmap's map load re-creates KeyFrames with the ids stored in the file (Osmap.cpp:586).  A DIFFERENT frame / keyframe
  // with a recycled mnId and the same N must be searched on ITS OWN features: four objects per kind with one id, the same
  // N and different content, searched in turn and again in reverse order (each one's resident copy must still be its own).
  {
    bool ok = true;
    int total = 0;
    const size_t uploadsBefore = ctx.residentUploads();
    std::vector<KeyFrame> kfs(4);
    std::vector<std::vector<MapPoint> > mps(4);
    std::vector<std::vector<MapPoint*> > ptsv(4);
    float Ts[4][16];
    for (int j = 0; j < 4; j++) {
      Rng r(seed * 10 + 40 + j);
      KeyFrame& kf = kfs[j];
      makeView(r, kf, 1700);
      kf.mnId = 3;                      // ... every one of them "keyframe 3"
      makePose(r, 1.35f, Ts[j]);
      setTcw(kf, Ts[j]);
      mps[j].resize(1200);
      for (int i = 0; i < 1200; i++) { mps[j][i].id = i; mps[j][i].kf = &kf; makePoint(r, mps[j][i], kf, r.below(kf.N), Ts[j], 1.35f); ptsv[j].push_back(&mps[j][i]); }
    }
    const int order[8] = {0, 1, 2, 3, 3, 1, 0, 2};
    for (int o = 0; o < 8; o++) {
      const int j = order[o];
      Table tab(mps[j]);
      OrcView kv = viewOf(kfs[j]);
      std::vector<int32_t> ids = idsOf(ptsv[j]);
      std::vector<MapPoint*> vpMatched(kfs[j].N, nullptr);
      std::vector<int32_t> m(kfs[j].N, -1);
      const int want = orc_sbp_scw(&kv, Ts[j], ids.data(), (int)ids.size(), &tab.P, m.data(), 10);
      const int got = orbfe::SearchByProjection(ctx, &kfs[j], kfs[j].mTcw, ptsv[j], vpMatched, 10);
      ok = ok && got == want && idsOf(vpMatched) == m && want > 100;
      total += got;
    }
    // four distinct contents -> four resident copies, each uploaded once; the second visits are hits
    ok = ok && ctx.residentUploads() == uploadsBefore + 4;
    report("KeyFrames sharing one mnId and N (map load): each searched on its own features", ok, total, total, 8);
  }
  {
    bool ok = true;
    int total = 0;
    for (int round = 0; round < 2; round++)
      for (int j = 0; j < 3; j++) {
        Rng r(seed * 10 + 60 + j);
        Frame cur, last;
        makeView(r, cur, 1800);
        makeView(r, last, 1200);
        cur.mnId = 7; last.mnId = 6;      // after every Reset() the same pair of ids
        float T[16];
        makePose(r, 1.f, T);
        setTcw(cur, T);
        std::vector<MapPoint> mp(1200);
        for (int i = 0; i < 1200; i++) {
          mp[i].id = i;
          const int k = r.below(cur.N);
          makePoint(r, mp[i], cur, k, T, 1.f);
          last.mvpMapPoints[i] = &mp[i];
          last.mvKeys[i].octave = last.mvKeysUn[i].octave = cur.mvKeysUn[k].octave;
        }
        if (round == 1) orbfe_resident_invalidate();   // with the hook: every context drops its frames at its next lookup
        Table tab(mp);
        std::vector<int32_t> cur_mp = idsOf(cur.mvpMapPoints), last_mp = idsOf(last.mvpMapPoints);
        std::vector<uint8_t> outl(last.N, 0);
        OrcView cv = viewOf(cur);
        const int want = orc_sbp_frame(&cv, T, reinterpret_cast<const OrcKp*>(last.mvKeys.data()), reinterpret_cast<const OrcKp*>(last.mvKeysUn.data()),
                                       last.N, last_mp.data(), outl.data(), &tab.P, cur_mp.data(), 15.f, 0);
        const int got = orbfe::SearchByProjection(ctx, false, cur, last, 15.f);
        ok = ok && got == want && idsOf(cur.mvpMapPoints) == cur_mp && want > 200;
        if (round == 1) ok = ok && ctx.residentFrames() == 1;   // the invalidation emptied the cache before this frame went in
        total += got;
      }
    report("Frames sharing one mnId and N (Tracking::Reset): each searched on its own features", ok, total, total, 6);
  }
  return failures ? 1 : 0;
}
