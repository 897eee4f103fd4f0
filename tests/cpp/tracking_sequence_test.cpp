// tracking_sequence_test.cpp -- the front end driven the way Tracking.cc drives it, end to end through
// include/orbfe/orb_shim.hpp (GPU, C ABI), every call compared with the CPU oracle (oracle/orb_oracle_pose.h):
//
//   per frame      Frame::Frame: ExtractORB -> UndistortKeyPoints -> (grid)               Frame.cc:100-111, 131-134, 286-320
//   frames 0, 1    Tracking::MonocularInitialization: SearchForInitialization(F0, F1, ..., 100)        Tracking.cc:383-384
//   frames >= 2    Tracking::TrackWithMotionModel: SearchByProjection(Cur, Last, th), again with 2*th if < 20 matches
//                                                                                         Tracking.cc:596-614
//                  Tracking::SearchLocalPoints: SearchByProjection(Cur, mvpLocalMapPoints, th)          Tracking.cc:776-824
//                  mLastFrame = Frame(mCurrentFrame)  (the copy keeps mnId, Frame.cc:39-62)
//   every 5th      "keyframe": new MapPoints join the local map, a few descriptors are recomputed
//                  (MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:227-292), a few points go bad
//   after frame R  Tracking::Reset(): Frame::nNextId = 0 (Tracking.cc:1159-1160), map cleared -- the ids 0, 1, 2 ... come
//                  round again on DIFFERENT frames (with nFeatures = 1000 the counts repeat as well).  The second half runs
//                  once with and once without orbfe_resident_invalidate(): the shim's frame cache is keyed by content, the
//                  hook only frees memory.
//
// Scene: a textured plane at depth Z in front of a camera that translates parallel to it -- frame k is frame 0 shifted by
// k * (dx, dy) px (tests/test_facade.py writes the frames), so Tcw_k = [I | t_k] with t_k = (k dx Z / fx, k dy Z / fy, 0) and a
// MapPoint back-projected from a keypoint of frame j reprojects onto the same texture in every frame.  The pose handed to
// the searches carries a small error (a motion-model prediction), one frame a large one (first search < 20 matches -> retry).
// Distortion variant (argv[3] = 1): pinhole k1 k2 p1 p2 through orbfe::UndistortKeyPoints, so mvKeysUn != mvKeys and the
// resident frame takes 8 bytes per keypoint of undistorted coordinates besides the extractor's arena.
//
//   usage: tracking_sequence_test <dir> <invalidate 0|1> <distort 0|1>     reads <dir>/meta.txt, <dir>/f%03d.gray
//   build: g++ -std=c++17 -O1 -ffp-contract=off -Iinclude -Ioracle tests/cpp/tracking_sequence_test.cpp os1_amd/liborbfe.so oracle/liborb_oracle.so
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orb_oracle_pose.h"
#include "orbfe/orb_shim.hpp"

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };  // cv::KeyPoint layout
struct Point2f { float x, y; };
struct MatF {   // the parts of cv::Mat the shim touches
  float v[16] = {0};
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;
  template <class T> T at(int r, int c) const { return (T)v[r * cols + c]; }
};
struct MapPoint {
  int id = 0;
  float pos[3] = {0, 0, 0};
  unsigned char desc[32];
  bool bad = false, mbTrackInView = false, plCandidato = false;
  int nObs = 0, mnTrackScaleLevel = 0;
  float mTrackProjX = 0, mTrackProjY = 0, mTrackViewCos = 1;
  MatF GetWorldPos() { MatF m; m.rows = 3; m.cols = 1; memcpy(m.v, pos, 12); return m; }
  MatF GetDescriptor() { MatF m; m.data = desc; m.step = 32; m.rows = 1; return m; }
  bool isBad() { return bad; }
  int Observations() { return nObs; }
};
static unsigned long g_nextFrameId = 0;   // Frame::nNextId
struct Frame {
  unsigned long mnId = 0;
  int N = 0;
  std::vector<KeyPoint> mvKeys, mvKeysUn;
  std::vector<unsigned char> descStore;
  MatF mDescriptors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<float> mvScaleFactors;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  MatF mTcw;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
  void bind() { mDescriptors.data = descStore.data(); mDescriptors.step = 32; mDescriptors.rows = N; }
  Frame() {}
  Frame(const Frame& o) { *this = o; }   // Frame.cc:39-62: a copy keeps mnId and the features
  Frame& operator=(const Frame& o) {
    mnId = o.mnId; N = o.N; mvKeys = o.mvKeys; mvKeysUn = o.mvKeysUn; descStore = o.descStore; mvpMapPoints = o.mvpMapPoints;
    mvbOutlier = o.mvbOutlier; mvScaleFactors = o.mvScaleFactors; fx = o.fx; fy = o.fy; cx = o.cx; cy = o.cy; mTcw = o.mTcw;
    bind();
    return *this;
  }
};
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;

struct Rng {
  unsigned long long s;
  explicit Rng(unsigned long long seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  unsigned long long next() { s += 0x9E3779B97F4A7C15ull; unsigned long long z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
  int below(int n) { return (int)(next() % (unsigned long long)n); }
};

static std::vector<unsigned char> readFile(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> v(n);
  if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}

static int failures = 0, checks = 0;
static void expect(bool ok, const char* what, int frame, int got, int want) {
  checks++;
  if (!ok) {
    failures++;
    printf("FAIL frame %d: %s (got %d, oracle %d)\n", frame, what, got, want);
  }
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string dir = argv[1];
  const bool invalidate = atoi(argv[2]) != 0, distort = atoi(argv[3]) != 0;
  int W = 0, H = 0, NF = 0, nfeat = 0, dxs = 0, dys = 0, R = 0;
  {
    FILE* f = fopen((dir + "/meta.txt").c_str(), "r");
    if (!f || fscanf(f, "%d %d %d %d %d %d %d", &W, &H, &NF, &nfeat, &dxs, &dys, &R) != 7) return 2;
    fclose(f);
  }
  const float fx = 500.f, fy = 498.f, cx = W * 0.5f + 0.5f, cy = H * 0.5f - 0.25f, Z = 5.f;
  const float dist[4] = {-0.12f, 0.05f, 0.0008f, -0.0006f};
  const int device = orbfe::detail::defaultDevice();
  orbfe::Extractor extractor(nfeat, 1.2f, 8, 20, 7, device);          // Tracking.cc:65 (ORBFE_DEVICE, like the facades)
  orbfe::MatcherContext ctx(device);
  void* ox = orc_extractor_create(nfeat, 1.2f, 8, 20, 7);
  const std::vector<float> sf = extractor.GetScaleFactors();
  const float logSf = std::log(1.2f);
  // Frame::ComputeImageBounds (Frame.cc:322-353), once (mbInitialComputations)
  orbfe::ComputeImageBounds(W, H, 0, fx, fy, cx, cy, distort ? dist : nullptr, distort ? 4 : 0, Frame::mnMinX, Frame::mnMaxX,
                            Frame::mnMinY, Frame::mnMaxY);
  {
    float b[4];
    orc_image_bounds(W, H, 0, fx, fy, cx, cy, distort ? dist : nullptr, distort ? 4 : 0, b);
    expect(b[0] == Frame::mnMinX && b[1] == Frame::mnMaxX && b[2] == Frame::mnMinY && b[3] == Frame::mnMaxY, "image bounds", -1, 0, 0);
  }
  const float bounds[4] = {Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY};

  Rng rng(4711);
  std::vector<MapPoint*> map;            // all MapPoints ever created in this half (owned; ids = index)
  std::vector<MapPoint*> local;          // mvpLocalMapPoints
  Frame last, ini;
  int state = 0;                         // 0 = no initial frame, 1 = initial frame stored, 2 = tracking
  std::vector<Point2f> vbPrevMatched;
  int framesSearched = 0, retries = 0, sbpCalls = 0, localCalls = 0, totalLast = 0, totalLocal = 0, failedSearches = 0, duplicates = 0;
  std::vector<KeyPoint> okps(nfeat + 256);
  std::vector<unsigned char> odesc((size_t)(nfeat + 256) * 32);

  auto poseOf = [&](int k, float ex, float ey, Frame& F) {   // Tcw = [I | t_k + error]
    F.mTcw = MatF();
    F.mTcw.rows = F.mTcw.cols = 4;
    F.mTcw.v[0] = F.mTcw.v[5] = F.mTcw.v[10] = F.mTcw.v[15] = 1.f;
    F.mTcw.v[3] = (float)(k * dxs) * Z / fx + ex;
    F.mTcw.v[7] = (float)(k * dys) * Z / fy + ey;
  };
  auto project = [&](const Frame& F, const float* pw, float& u, float& v, float& zc) {   // the test's isInFrustum arithmetic
    const float xc = pw[0] + F.mTcw.v[3], yc = pw[1] + F.mTcw.v[7];
    zc = pw[2] + F.mTcw.v[11];
    u = fx * xc / zc + cx;
    v = fy * yc / zc + cy;
  };
  auto newPoint = [&](const Frame& F, int idx) {   // a MapPoint triangulated at keypoint idx of F (on the plane)
    MapPoint* p = new MapPoint();
    p->id = (int)map.size();
    const KeyPoint& k = F.mvKeysUn[idx];
    p->pos[0] = (k.x - cx) / fx * Z - F.mTcw.v[3];
    p->pos[1] = (k.y - cy) / fy * Z - F.mTcw.v[7];
    p->pos[2] = Z;
    memcpy(p->desc, &F.descStore[(size_t)idx * 32], 32);
    p->nObs = rng.below(4);
    map.push_back(p);
    return p;
  };

  for (int k = 0; k < NF; k++) {
    if (k == R) {   // Tracking::Reset()
      g_nextFrameId = 0;                                   // Frame::nNextId = 0   (Tracking.cc:1160)
      for (MapPoint* p : map) delete p;
      map.clear(); local.clear();
      state = 0;
      if (invalidate) orbfe_resident_invalidate();         // the hook INTEGRATION.md asks Reset() to call
    }
    // ---- Frame::Frame ------------------------------------------------------------------------------------------------
    char name[64];
    snprintf(name, sizeof name, "/f%03d.gray", k);
    std::vector<unsigned char> img = readFile(dir + name);
    Frame cur;
    cur.mnId = g_nextFrameId++;
    cur.fx = fx; cur.fy = fy; cur.cx = cx; cur.cy = cy;
    cur.mvScaleFactors = sf;
    extractor.extract(img.data(), H, W, (size_t)W, cur.mvKeys, cur.descStore);           // ExtractORB
    cur.N = (int)cur.mvKeys.size();
    cur.bind();
    {
      const int on = orc_extract(ox, img.data(), H, W, W, reinterpret_cast<OrcKp*>(okps.data()), odesc.data(), (int)okps.size());
      expect(on == cur.N && memcmp(okps.data(), cur.mvKeys.data(), (size_t)on * sizeof(KeyPoint)) == 0 &&
                 memcmp(odesc.data(), cur.descStore.data(), (size_t)on * 32) == 0,
             "extraction differs from the oracle", k, cur.N, on);
    }
    orbfe::UndistortKeyPoints(cur.mvKeys, cur.mvKeysUn, 0, fx, fy, cx, cy, distort ? dist : nullptr, distort ? 4 : 0);
    if (distort) {
      std::vector<float> xy((size_t)cur.N * 2);
      for (int i = 0; i < cur.N; i++) { xy[2 * i] = cur.mvKeys[i].x; xy[2 * i + 1] = cur.mvKeys[i].y; }
      orc_undistort_pinhole(xy.data(), cur.N, fx, fy, cx, cy, dist, 4);
      bool same = true;
      for (int i = 0; i < cur.N; i++) same = same && xy[2 * i] == cur.mvKeysUn[i].x && xy[2 * i + 1] == cur.mvKeysUn[i].y;
      expect(same, "undistortion differs from the oracle", k, 0, 0);
    }
    cur.mvpMapPoints.assign(cur.N, nullptr);
    cur.mvbOutlier.assign(cur.N, false);
    poseOf(k, 0.f, 0.f, cur);

    if (state == 0) {   // Tracking::MonocularInitialization, first frame (Tracking.cc:340-362)
      ini = cur;
      vbPrevMatched.resize(ini.N);
      for (int i = 0; i < ini.N; i++) vbPrevMatched[i] = Point2f{ini.mvKeysUn[i].x, ini.mvKeysUn[i].y};
      state = 1;
      continue;
    }
    if (state == 1) {   // second frame: SearchForInitialization (Tracking.cc:383-384), then the initial map
      std::vector<int> vnMatches12, want12(ini.N);
      std::vector<Point2f> wantPrev = vbPrevMatched;
      const int want = orc_search_for_initialization(reinterpret_cast<const OrcKp*>(ini.mvKeysUn.data()), ini.descStore.data(), ini.N,
                                                     reinterpret_cast<const OrcKp*>(cur.mvKeysUn.data()), cur.descStore.data(), cur.N, bounds,
                                                     reinterpret_cast<float*>(wantPrev.data()), want12.data(), 100, 0.9f, 1);
      const int got = orbfe::SearchForInitialization(ctx, 0.9f, true, ini, cur, vbPrevMatched, vnMatches12, 100);
      expect(got == want && vnMatches12 == want12 && memcmp(wantPrev.data(), vbPrevMatched.data(), sizeof(Point2f) * wantPrev.size()) == 0 && want >= 100,
             "SearchForInitialization", k, got, want);
      // CreateInitialMapMonocular (Tracking.cc:428-520): one MapPoint per match, observed by both frames
      for (int i = 0; i < ini.N; i++)
        if (vnMatches12[i] >= 0) {
          MapPoint* p = newPoint(cur, vnMatches12[i]);
          p->nObs = 2;
          cur.mvpMapPoints[vnMatches12[i]] = p;
          local.push_back(p);
        }
      last = Frame(cur);
      state = 2;
      continue;
    }
    // ---- Tracking::TrackWithMotionModel (Tracking.cc:573-650) ---------------------------------------------------------
    const bool badPrediction = (k % 9) == 4;   // a prediction far enough off that th = 15 finds < 20 matches
    poseOf(k, (float)rng.uni(-0.012, 0.012) + (badPrediction ? 0.6f : 0.f), (float)rng.uni(-0.012, 0.012), cur);
    std::vector<int32_t> lastIds(last.N), curIds(cur.N, -1);
    std::vector<uint8_t> lastOut(last.N);
    std::vector<float> tpos(3 * map.size());
    std::vector<uint8_t> tdesc(32 * map.size()), tbad(map.size());
    std::vector<int32_t> tobs(map.size()), tidx(map.size(), -1);
    std::vector<float> tzero(3 * map.size(), 0.f), tone(map.size(), 1.f);
    auto table = [&](OrcPoints& P) {
      for (size_t i = 0; i < map.size(); i++) {
        memcpy(&tpos[3 * i], map[i]->pos, 12); memcpy(&tdesc[32 * i], map[i]->desc, 32);
        tbad[i] = map[i]->bad; tobs[i] = map[i]->nObs;
      }
      P.M = (int)map.size(); P.pos = tpos.data(); P.normal = tzero.data(); P.mfMinDistance = tone.data(); P.mfMaxDistance = tone.data();
      P.desc = tdesc.data(); P.bad = tbad.data(); P.nObs = tobs.data(); P.idxInKF = tidx.data();
    };
    for (int i = 0; i < last.N; i++) { lastIds[i] = last.mvpMapPoints[i] ? last.mvpMapPoints[i]->id : -1; lastOut[i] = last.mvbOutlier[i]; }
    OrcView cv;
    cv.kpsUn = reinterpret_cast<const OrcKp*>(cur.mvKeysUn.data()); cv.desc = cur.descStore.data(); cv.n = cur.N;
    memcpy(cv.bounds, bounds, sizeof bounds);
    cv.fx = fx; cv.fy = fy; cv.cx = cx; cv.cy = cy; cv.scaleFactors = sf.data(); cv.invLevelSigma2 = sf.data(); cv.nlevels = 8; cv.logScaleFactor = logSf;
    float th = 15.f;                                                              // monocular (Tracking.cc:596-600)
    for (int attempt = 0; attempt < 2; attempt++) {
      std::fill(cur.mvpMapPoints.begin(), cur.mvpMapPoints.end(), static_cast<MapPoint*>(nullptr));   // :594 / :611
      std::fill(curIds.begin(), curIds.end(), -1);
      OrcPoints P;
      table(P);
      const int want = orc_sbp_frame(&cv, cur.mTcw.v, reinterpret_cast<const OrcKp*>(last.mvKeys.data()),
                                     reinterpret_cast<const OrcKp*>(last.mvKeysUn.data()), last.N, lastIds.data(), lastOut.data(), &P,
                                     curIds.data(), th, 1);
      const int got = orbfe::SearchByProjection(ctx, true, cur, last, th);
      bool same = got == want;
      for (int i = 0; i < cur.N && same; i++) same = (cur.mvpMapPoints[i] ? cur.mvpMapPoints[i]->id : -1) == curIds[i];
      expect(same, attempt ? "SearchByProjection(Cur, Last, 2*th)" : "SearchByProjection(Cur, Last, th)", k, got, want);
      sbpCalls++;
      totalLast += got;
      if (got >= 20) break;
      if (attempt == 0) { th = 2 * th; retries++; }                               // :608-614
    }
    // ---- Tracking::SearchLocalPoints (Tracking.cc:776-824) -----------------------------------------------------------
    for (MapPoint* p : local) p->mbTrackInView = false;
    std::vector<MapPoint*> already(cur.mvpMapPoints.begin(), cur.mvpMapPoints.end());
    for (MapPoint* p : local) {
      bool inFrame = false;
      for (MapPoint* q : already) if (q == p) { inFrame = true; break; }
      if (inFrame || p->bad) continue;                                            // mnLastFrameSeen == mnId / isBad
      float u, v, zc;
      project(cur, p->pos, u, v, zc);
      if (zc < 0.f || u < Frame::mnMinX || u > Frame::mnMaxX || v < Frame::mnMinY || v > Frame::mnMaxY) continue;
      p->mbTrackInView = true;
      p->mTrackProjX = u; p->mTrackProjY = v;
      p->mnTrackScaleLevel = p->id % 3;                                           // PredictScale stand-in, spread over levels 0..2
      p->mTrackViewCos = (p->id % 5) ? 0.9995f : 0.99f;                           // both radius classes (:63-65)
    }
    {
      const float thLocal = (k - (k >= R ? R : 0)) < 4 ? 3.f : 1.f;              // th = 3 / 5 shortly after a relocalisation (:818-822)
      // Two things the descriptor table must survive.  (a) k % 4 == 1: descriptors change, then a search FAILS after
      // its snapshot rewrote the table's mirror rows (an out-of-range level makes the C call return ORBFE_ERR_INVALID, the shim
      // throws, tableCommit never runs): the next search must not read those rows from the device.  (b) k % 4 == 3: the same MapPoint
      // twice in vpMapPoints (the reference tolerates it) right after its descriptor changed: the second occurrence must not read the
      // device row either.
      std::vector<MapPoint*> searched = local;
      if (k % 4 == 1 || k % 4 == 3) {
        int changedRows = 0;
        for (MapPoint* p : local)
          if (p->mbTrackInView && !p->bad && rng.uni() < 0.2) { for (int b = 0; b < 32; b += 3) p->desc[b] ^= (unsigned char)(1 + rng.below(255)); changedRows++; }
        expect(changedRows > 0, "descriptors changed in front of the table stress", k, changedRows, 1);
      }
      if (k % 4 == 1) {
        MapPoint* victim = nullptr;
        for (MapPoint* p : local) if (p->mbTrackInView && !p->bad) victim = p;
        if (victim) {
          const int keep = victim->mnTrackScaleLevel;
          victim->mnTrackScaleLevel = 99;
          std::vector<MapPoint*> before = cur.mvpMapPoints;
          bool threw = false;
          try { orbfe::SearchByProjection(ctx, 0.8f, cur, local, thLocal); } catch (const std::exception&) { threw = true; }
          victim->mnTrackScaleLevel = keep;
          expect(threw, "a MapPoint level outside the pyramid fails the search", k, threw, 1);
          expect(before == cur.mvpMapPoints, "a failed search leaves F.mvpMapPoints untouched", k, 0, 0);
          failedSearches++;
        }
      }
      if (k % 4 == 3) {
        std::vector<MapPoint*> dup;
        for (MapPoint* p : local) if (p->mbTrackInView && !p->bad && rng.uni() < 0.1) dup.push_back(p);
        for (MapPoint* p : dup) searched.insert(searched.begin() + rng.below((int)searched.size() + 1), p);
        duplicates += (int)dup.size();
      }
      const int nmp = (int)searched.size();
      std::vector<float> xy(2 * (size_t)nmp, 0.f), vcos(nmp, 0.f);
      std::vector<int> lvl(nmp, 0), assigned(cur.N, -1);
      std::vector<uint8_t> flags(nmp, 0), mdesc(32 * (size_t)nmp, 0), occ(cur.N, 0);
      for (int i = 0; i < cur.N; i++) occ[i] = cur.mvpMapPoints[i] && cur.mvpMapPoints[i]->nObs > 0;
      for (int i = 0; i < nmp; i++) {
        MapPoint* p = searched[i];
        flags[i] = (p->mbTrackInView ? 1 : 0) | (p->bad ? 2 : 0) | (p->plCandidato ? 4 : 0) | (p->nObs > 0 ? 8 : 0);
        if (!p->mbTrackInView) continue;
        xy[2 * i] = p->mTrackProjX; xy[2 * i + 1] = p->mTrackProjY; lvl[i] = p->mnTrackScaleLevel; vcos[i] = p->mTrackViewCos;
        memcpy(&mdesc[32 * (size_t)i], p->desc, 32);
      }
      const int want = orc_search_by_projection(reinterpret_cast<const OrcKp*>(cur.mvKeysUn.data()), cur.descStore.data(), cur.N, bounds,
                                                sf.data(), occ.data(), xy.data(), lvl.data(), vcos.data(), flags.data(), mdesc.data(), nmp,
                                                thLocal, 0.8f, assigned.data());
      std::vector<MapPoint*> before = cur.mvpMapPoints;
      const int got = orbfe::SearchByProjection(ctx, 0.8f, cur, searched, thLocal);
      bool same = got == want;
      for (int i = 0; i < cur.N && same; i++) same = cur.mvpMapPoints[i] == (assigned[i] >= 0 ? searched[assigned[i]] : before[i]);
      expect(same, "SearchByProjection(Cur, LocalMapPoints, th)", k, got, want);
      localCalls++;
      totalLocal += got;
    }
    framesSearched++;
    // ---- map maintenance between frames ---------------------------------------------------------------------------------
    for (int i = 0; i < cur.N; i++)
      if (cur.mvpMapPoints[i] && rng.uni() < 0.05) cur.mvbOutlier[i] = true;        // pose optimisation marks outliers
    if (k % 5 == 0) {   // a keyframe: new points from unmatched keypoints, recomputed descriptors, culled points
      int added = 0;
      for (int i = 0; i < cur.N && added < 150; i++)
        if (!cur.mvpMapPoints[i] && cur.mvKeysUn[i].octave <= 2 && rng.uni() < 0.5) {
          poseOf(k, 0.f, 0.f, cur);                                                 // triangulated with the optimised pose
          MapPoint* p = newPoint(cur, i);
          cur.mvpMapPoints[i] = p;
          // the local map is rebuilt from the keyframes' points (Tracking.cc:870-900): new ones land in the middle too
          local.insert(local.begin() + rng.below((int)local.size() + 1), p);
          added++;
        }
      for (int i = 0; i < cur.N; i++)
        if (cur.mvpMapPoints[i] && rng.uni() < 0.15) memcpy(cur.mvpMapPoints[i]->desc, &cur.descStore[(size_t)i * 32], 32);   // ComputeDistinctiveDescriptors
      for (MapPoint* p : local) if (rng.uni() < 0.02) p->bad = true;              // MapPointCulling
    } else if (k % 5 == 2 && !local.empty()) {
      // ONE descriptor changes between two searches of an otherwise identical local map: the device table must notice
      MapPoint* p = local[rng.below((int)local.size())];
      for (int b = 0; b < 32; b++) p->desc[b] ^= (unsigned char)rng.below(256);
    }
    for (int i = 0; i < cur.N; i++)
      if (cur.mvpMapPoints[i] && !cur.mvbOutlier[i] && rng.uni() < 0.3) cur.mvpMapPoints[i]->nObs++;
    last = Frame(cur);                                                              // mLastFrame = Frame(mCurrentFrame)
  }
  for (MapPoint* p : map) delete p;
  orc_extractor_destroy(ox);
  printf("frames %d searched %d sbp_calls %d retries %d local_calls %d matches_last %d matches_local %d\n", NF, framesSearched, sbpCalls,
         retries, localCalls, totalLast, totalLocal);
  printf("resident uploads %zu from_extract %zu hits %zu frames_cached %zu table_rows %zu rows_changed %zu rows_from_device %zu clean_searches %zu\n",
         ctx.residentUploads(), ctx.residentFromExtract(), ctx.residentHits(), ctx.residentFrames(), ctx.tableRows(), ctx.tableRowsChanged(),
         ctx.tableRowsFromDevice(), ctx.tableCleanSearches());
  // every searched Frame came straight from the extractor's arena: no keypoint or descriptor was uploaded a second time
  expect(ctx.residentUploads() == 0, "a frame's features were uploaded although the extractor held them", -1, (int)ctx.residentUploads(), 0);
  expect((int)ctx.residentFromExtract() == framesSearched, "frames built from the extractor's arena", -1, (int)ctx.residentFromExtract(), framesSearched);
  expect(retries >= 2, "the 2*th retry was exercised", -1, retries, 2);
  expect(failedSearches >= 2 && duplicates >= 10, "the descriptor table was stressed: failed searches / duplicated MapPoints", -1, failedSearches, duplicates);
  // the local map's descriptors: most reads are served by the device table (a row crosses PCIe when it is new or its bytes changed)
  expect(ctx.tableRowsFromDevice() > 2 * ctx.tableRowsChanged(), "descriptor reads served by the device table vs rows sent", -1,
         (int)ctx.tableRowsFromDevice(), (int)ctx.tableRowsChanged());
  expect(totalLast > 0 && totalLocal > 0, "the searches found matches", -1, totalLast, totalLocal);
  printf("%s %d checks, %d failures\n", failures ? "FAIL" : "PASS", checks, failures);
  return failures ? 1 : 0;
}
