// facade_test.cpp -- drives include/orbfe/orb_shim.hpp the way the reference's Tracking/Frame code
// would (Frame.cc:131-134, Tracking.cc:383-384, 818-824), with stand-in structs that expose the
// member names the shim reads.  Inputs/outputs are raw binary files so tests/test_gpu_parity.py can
// compare every result with the CPU oracle.
//   usage: facade_test <dir>      reads <dir>/A.gray, <dir>/B.gray (W,H from <dir>/meta.txt)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "orbfe/orb_shim.hpp"

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };  // cv::KeyPoint layout
struct Point2f { float x, y; };
struct Mat {  // the three public cv::Mat members the shim touches
  unsigned char* data = nullptr;
  size_t step = 0;
  int rows = 0;
};
struct MapPoint {
  bool mbTrackInView = true, plCandidato = false, bad = false;
  float mTrackProjX = 0, mTrackProjY = 0, mTrackViewCos = 1;
  int mnTrackScaleLevel = 0, nObs = 1;
  unsigned char desc[32];
  bool isBad() { return bad; }
  int Observations() { return nObs; }
  Mat GetDescriptor() { Mat m; m.data = desc; m.step = 32; m.rows = 1; return m; }
};
typedef std::map<unsigned, double> BowVector;                         // DBoW2::BowVector
typedef std::map<unsigned, std::vector<unsigned> > FeatureVector;     // DBoW2::FeatureVector
static unsigned long g_nextFrameId = 0;
struct Frame {
  unsigned long mnId = g_nextFrameId++;   // Frame.cc:78: unique per constructed object (the shim's resident-frame cache key)
  int N = 0;
  BowVector mBowVec;
  FeatureVector mFeatVec;
  std::vector<KeyPoint> mvKeys, mvKeysUn;
  std::vector<unsigned char> descStore;
  Mat mDescriptors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<float> mvScaleFactors;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
struct KeyFrame : Frame {   // the members SearchByBoW / SearchForTriangulation read (KeyFrame.h)
  std::vector<float> mvLevelSigma2;
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  MapPoint* GetMapPoint(size_t i) { return mvpMapPoints[i]; }
};
struct Mat33f {   // cv::Mat F12 as far as the shim reads it
  float v[9];
  template <class T> T at(int r, int c) const { return v[3 * r + c]; }
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
static void writeFile(const std::string& p, const void* d, size_t n) {
  FILE* f = fopen(p.c_str(), "wb");
  fwrite(d, 1, n, f);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  int W = 0, H = 0, N = 0, nmp = 0;
  {
    FILE* f = fopen((dir + "/meta.txt").c_str(), "r");
    if (!f || fscanf(f, "%d %d %d %d", &W, &H, &N, &nmp) != 4) return 2;
    fclose(f);
  }
  orbfe::Extractor extractor(N, 1.2f, 8, 20, 7);           // Tracking.cc:65
  orbfe::MatcherContext ctx;
  Frame::mnMinX = 0; Frame::mnMaxX = (float)W; Frame::mnMinY = 0; Frame::mnMaxY = (float)H;  // Frame.cc:347-352
  KeyFrame F[2];
  const char* names[2] = {"A", "B"};
  for (int i = 0; i < 2; i++) {
    std::vector<unsigned char> img = readFile(dir + "/" + names[i] + ".gray");
    extractor.extract(img.data(), H, W, (size_t)W, F[i].mvKeys, F[i].descStore);  // Frame::ExtractORB
    F[i].mvKeysUn = F[i].mvKeys;                                                 // no distortion
    F[i].mDescriptors.data = F[i].descStore.data();
    F[i].mDescriptors.step = 32;
    F[i].mDescriptors.rows = (int)F[i].mvKeys.size();
    F[i].N = (int)F[i].mvKeys.size();
    F[i].mvpMapPoints.assign(F[i].mvKeys.size(), nullptr);
    F[i].mvScaleFactors = extractor.GetScaleFactors();
    F[i].mvLevelSigma2 = extractor.GetScaleSigmaSquares();
    writeFile(dir + "/" + names[i] + ".kps", F[i].mvKeys.data(), F[i].mvKeys.size() * sizeof(KeyPoint));
    writeFile(dir + "/" + names[i] + ".desc", F[i].descStore.data(), F[i].descStore.size());
  }
  // Tracking::MonocularInitialization, Tracking.cc:355-357,383-384
  std::vector<Point2f> vbPrevMatched(F[0].mvKeysUn.size());
  for (size_t i = 0; i < F[0].mvKeysUn.size(); i++) vbPrevMatched[i] = Point2f{F[0].mvKeysUn[i].x, F[0].mvKeysUn[i].y};
  std::vector<int> vnMatches12;
  int nm = orbfe::SearchForInitialization(ctx, 0.9f, true, F[0], F[1], vbPrevMatched, vnMatches12, 100);
  writeFile(dir + "/sfi.matches", vnMatches12.data(), vnMatches12.size() * sizeof(int));
  writeFile(dir + "/sfi.prev", vbPrevMatched.data(), vbPrevMatched.size() * sizeof(Point2f));
  // Tracking::SearchLocalPoints, Tracking.cc:818-824: MapPoints come from <dir>/mp.bin
  //   per MapPoint: float x, y, viewcos; int level; uchar flags(inview|bad<<1|cand<<2|obs<<3); uchar desc[32]
  std::vector<unsigned char> mpraw = readFile(dir + "/mp.bin");
  const size_t rec = 12 + 4 + 1 + 32;
  std::vector<MapPoint> mps(nmp);
  std::vector<MapPoint*> vp(nmp);
  for (int i = 0; i < nmp; i++) {
    const unsigned char* r = &mpraw[i * rec];
    MapPoint& p = mps[i];
    memcpy(&p.mTrackProjX, r, 4); memcpy(&p.mTrackProjY, r + 4, 4); memcpy(&p.mTrackViewCos, r + 8, 4);
    memcpy(&p.mnTrackScaleLevel, r + 12, 4);
    const unsigned char fl = r[16];
    p.mbTrackInView = fl & 1; p.bad = fl & 2; p.plCandidato = fl & 4; p.nObs = (fl & 8) ? 3 : 0;
    memcpy(p.desc, r + 17, 32);
    vp[i] = &p;
  }
  int nsbp = orbfe::SearchByProjection(ctx, 0.8f, F[1], vp, 1.0f);
  std::vector<int> assigned(F[1].mvpMapPoints.size(), -1);
  for (size_t i = 0; i < assigned.size(); i++)
    if (F[1].mvpMapPoints[i]) assigned[i] = (int)(F[1].mvpMapPoints[i] - mps.data());
  writeFile(dir + "/sbp.assigned", assigned.data(), assigned.size() * sizeof(int));
  // Relocalisation / TrackReferenceKeyFrame (Tracking.cc:543-548): ComputeBoW of both, SearchByBoW(KF, F) and the
  // keyframe-keyframe overload (LoopClosing.cc:242).  A and B carry "MapPoints" flagged in <dir>/bow.valid.
  int nbow1 = -1, nbow2 = -1, ntri = -1;
  std::vector<MapPoint> own[2];   // the keyframes' MapPoints of the bag-of-words part (they outlive it: the timing mode reuses them)
  FILE* vf = fopen((dir + "/voc.bin").c_str(), "rb");
  if (vf) {
    fclose(vf);
    orbfe::Vocabulary voc;
    if (!voc.loadFromBinaryFile(dir + "/voc.bin")) return 3;
    std::vector<unsigned char> valid = readFile(dir + "/bow.valid");   // n1 + n2 bytes: 0 none, 1 MapPoint, 2 bad MapPoint
    for (int i = 0; i < 2; i++) {
      orbfe::ComputeBoW(voc, F[i]);
      std::vector<double> bv;
      for (auto& e : F[i].mBowVec) { bv.push_back((double)e.first); bv.push_back(e.second); }
      writeFile(dir + "/" + names[i] + ".bow", bv.data(), bv.size() * sizeof(double));
      std::vector<unsigned> fvv;
      for (auto& e : F[i].mFeatVec) { fvv.push_back(e.first); fvv.push_back((unsigned)e.second.size()); fvv.insert(fvv.end(), e.second.begin(), e.second.end()); }
      writeFile(dir + "/" + names[i] + ".fv", fvv.data(), fvv.size() * sizeof(unsigned));
      const size_t n = F[i].mvKeys.size(), base = i ? F[0].mvKeys.size() : 0;
      own[i].resize(n);
      F[i].mvpMapPoints.assign(n, nullptr);
      for (size_t j = 0; j < n; j++)
        if (valid[base + j]) { own[i][j].bad = valid[base + j] == 2; F[i].mvpMapPoints[j] = &own[i][j]; }
    }
    std::vector<MapPoint*> vpMapPointMatches, vpMatches12;
    nbow1 = orbfe::SearchByBoW(ctx, 0.7f, true, &F[0], static_cast<Frame&>(F[1]), vpMapPointMatches);
    std::vector<int> r1(vpMapPointMatches.size(), -1), r2;
    for (size_t i = 0; i < r1.size(); i++)
      if (vpMapPointMatches[i]) r1[i] = (int)(vpMapPointMatches[i] - own[0].data());
    writeFile(dir + "/bow1.matches", r1.data(), r1.size() * sizeof(int));
    {   // the relocalisation loop in one submission (Tracking.cc:1005-1030): the same keyframe twice + one skipped
      std::vector<KeyFrame*> kfs = {&F[0], &F[0], &F[0]};
      std::vector<bool> skip = {false, true, false};
      std::vector<std::vector<MapPoint*> > vv;
      const std::vector<int> cnt = orbfe::SearchByBoW(ctx, 0.7f, true, kfs, skip, static_cast<Frame&>(F[1]), vv);
      if (cnt.size() != 3 || cnt[0] != nbow1 || cnt[2] != nbow1 || cnt[1] != 0 || vv[0] != vpMapPointMatches || vv[2] != vpMapPointMatches) return 4;
    }
    nbow2 = orbfe::SearchByBoW(ctx, 0.75f, true, &F[0], &F[1], vpMatches12);
    r2.assign(vpMatches12.size(), -1);
    for (size_t i = 0; i < r2.size(); i++)
      if (vpMatches12[i]) r2[i] = (int)(vpMatches12[i] - own[1].data());
    writeFile(dir + "/bow2.matches", r2.data(), r2.size() * sizeof(int));
    {   // a frame cache of ONE entry: both sides' resident rows must survive the call (the two most recent frames are never evicted)
      ctx.setFrameCacheCapacity(1);
      std::vector<MapPoint*> again;
      const int nb = orbfe::SearchByBoW(ctx, 0.75f, true, &F[0], &F[1], again);
      ctx.setFrameCacheCapacity(48);
      if (nb != nbow2 || again != vpMatches12) return 7;
    }
    // LocalMapping::CreateNewMapPoints (LocalMapping.cc:396): F12 = [t]x for the image translation A -> B
    Mat33f F12 = {{0.f, 0.f, 4e-3f, 0.f, 0.f, 10e-3f, -4e-3f, -10e-3f, 0.f}};
    std::vector<std::pair<size_t, size_t> > vMatchedPairs;
    ntri = orbfe::SearchForTriangulation(ctx, true, &F[0], &F[1], F12, 480.f, 270.f, vMatchedPairs);
    std::vector<int> tp;
    for (auto& pr : vMatchedPairs) { tp.push_back((int)pr.first); tp.push_back((int)pr.second); }
    writeFile(dir + "/tri.pairs", tp.data(), tp.size() * sizeof(int));
  }
  printf("%zu %zu %d %d %d %d %d\n", F[0].mvKeys.size(), F[1].mvKeys.size(), nm, nsbp, nbow1, nbow2, ntri);
  if (argc > 2 && std::string(argv[2]) == "time") {
    // what an integrated build pays per call THROUGH the facade (marshalling of the mock Frame / MapPoint objects included;
    // their accessors take no mutex, unlike the reference's): median of 200 blocking calls each
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<unsigned char> img = readFile(dir + "/B.gray");
    std::vector<KeyPoint> kk;
    std::vector<unsigned char> dd;
    std::vector<double> tEx, tSbp, tBow;
    for (int r = 0; r < 200; r++) {
      double t0 = now();
      extractor.extract(img.data(), H, W, (size_t)W, kk, dd);
      tEx.push_back(now() - t0);
      std::vector<MapPoint*> keep = F[1].mvpMapPoints;
      F[1].mvpMapPoints.assign(F[1].mvKeys.size(), nullptr);
      t0 = now();
      const int n2 = orbfe::SearchByProjection(ctx, 0.8f, F[1], vp, 1.0f);
      tSbp.push_back(now() - t0);
      if (n2 != nsbp) return 5;
      F[1].mvpMapPoints = keep;
      if (nbow1 >= 0) {
        std::vector<MapPoint*> vv;
        t0 = now();
        const int nb = orbfe::SearchByBoW(ctx, 0.7f, true, &F[0], static_cast<Frame&>(F[1]), vv);
        tBow.push_back(now() - t0);
        if (nb != nbow1) return 6;
      }
    }
    printf("TIMING extract_host_frame_ms %.4f search_by_projection_%d_mappoints_ms %.4f search_by_bow_ms %.4f\n", med(tEx), nmp, med(tSbp),
           tBow.empty() ? -1.0 : med(tBow));
  }
  return 0;
}
