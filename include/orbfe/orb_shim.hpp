// orb_shim.hpp -- header-only C++ host side above the C ABI (include/orbfe.h).
//
// Generic (duck-typed) so it compiles with or without OpenCV: the reference's own types
// (cv::KeyPoint, cv::Mat, ORB_SLAM2::Frame, ORB_SLAM2::MapPoint) are template parameters and are
// only touched through the member names the reference uses.  include/orbfe/ORBextractor.h binds the
// extractor part to the exact ORB_SLAM2::ORBextractor signature when OpenCV headers are present;
// INTEGRATION.md shows the three-line bodies that replace the hot ORBmatcher functions.
//
// Error behaviour mirrors the reference: no exceptions for data conditions (empty image => outputs
// untouched, matchers return the match count); a failing GPU call throws std::runtime_error with
// orbfe_last_error() because the reference has no channel to report it and continuing would
// silently corrupt tracking.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../orbfe.h"

namespace orbfe {

inline void check(int rc) {
  if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error());
}

// ORBextractor (reference include/ORBextractor.h:155-373) minus the cv:: types.
class Extractor {
 public:
  Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0) {
    check(orbfe_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &h_));
    const int n = orbfe_extractor_levels(h_);
    sf_.resize(n); isf_.resize(n); s2_.resize(n); is2_.resize(n);
    check(orbfe_extractor_scale_tables(h_, sf_.data(), isf_.data(), s2_.data(), is2_.data()));
    cap_ = orbfe_extractor_max_keypoints(h_);
  }
  ~Extractor() { orbfe_extractor_destroy(h_); }
  Extractor(const Extractor&) = delete;
  Extractor& operator=(const Extractor&) = delete;

  // operator() core: KeyPointT must have cv::KeyPoint's 28-byte layout.
  template <class KeyPointT>
  void extract(const uint8_t* gray, int rows, int cols, size_t step, std::vector<KeyPointT>& keypoints,
               std::vector<uint8_t>& descriptors) {
    static_assert(sizeof(KeyPointT) == sizeof(OrbfeKeyPoint), "KeyPointT must match cv::KeyPoint's layout");
    if (!gray || rows == 0 || cols == 0) return;  // reference: silent return, outputs untouched
    cap_ = std::max(cap_, orbfe_extractor_max_keypoints_for_size(h_, rows, cols));   // strips wider than 4.5 : 1
    kp_.resize(cap_);
    desc_.resize((size_t)cap_ * 32);
    int n = 0;
    check(orbfe_extract(h_, gray, rows, cols, step, kp_.data(), desc_.data(), cap_, &n));
    keypoints.clear();
    keypoints.resize(n);
    if (n) std::memcpy(static_cast<void*>(keypoints.data()), kp_.data(), (size_t)n * sizeof(OrbfeKeyPoint));
    descriptors.assign(desc_.begin(), desc_.begin() + (size_t)n * 32);
  }

  int GetLevels() const { return (int)sf_.size(); }
  float GetScaleFactor() const { return orbfe_extractor_scale_factor(h_); }
  std::vector<float> GetScaleFactors() const { return sf_; }
  std::vector<float> GetInverseScaleFactors() const { return isf_; }
  std::vector<float> GetScaleSigmaSquares() const { return s2_; }
  std::vector<float> GetInverseScaleSigmaSquares() const { return is2_; }
  orbfe_extractor* handle() const { return h_; }
  int capacity() const { return cap_; }

 private:
  orbfe_extractor* h_ = nullptr;
  int cap_ = 0;
  std::vector<float> sf_, isf_, s2_, is2_;
  std::vector<OrbfeKeyPoint> kp_;
  std::vector<uint8_t> desc_;
};

// One GPU matcher context per thread that runs searches (Tracking constructs ORBmatcher objects on
// the stack per use, Tracking.cc:383,596,818; the context is the long-lived part).
class MatcherContext {
 public:
  explicit MatcherContext(int device = 0) { check(orbfe_matcher_create(device, &m_)); }
  ~MatcherContext() { orbfe_matcher_destroy(m_); }
  MatcherContext(const MatcherContext&) = delete;
  MatcherContext& operator=(const MatcherContext&) = delete;
  orbfe_matcher* get() const { return m_; }

 private:
  orbfe_matcher* m_ = nullptr;
};

namespace detail {
// rows of a CV_8U N x 32 descriptor matrix -> contiguous bytes (cv::Mat has public data/step/rows)
template <class MatT>
inline const uint8_t* packedDescriptors(const MatT& m, int n, std::vector<uint8_t>& tmp) {
  if (n == 0) return nullptr;
  if ((size_t)m.step == 32) return m.data;
  tmp.resize((size_t)n * 32);
  for (int i = 0; i < n; i++) std::memcpy(&tmp[(size_t)i * 32], m.data + (size_t)i * m.step, 32);
  return tmp.data();
}
template <class FrameT>
inline void frameBounds(const FrameT& F, float b[4]) {
  b[0] = F.mnMinX; b[1] = F.mnMaxX; b[2] = F.mnMinY; b[3] = F.mnMaxY;
}
}  // namespace detail

// int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)   (ORBmatcher.cc:1605-1621)
template <class MatT>
inline int DescriptorDistance(const MatT& a, const MatT& b) { return orbfe_hamming(a.data, b.data); }

// int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched,
//                                         vector<int>& vnMatches12, int windowSize)   (ORBmatcher.cc:400-515)
template <class FrameT, class Point2fT>
inline int SearchForInitialization(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, FrameT& F1,
                                   FrameT& F2, std::vector<Point2fT>& vbPrevMatched, std::vector<int>& vnMatches12,
                                   int windowSize) {
  static_assert(sizeof(Point2fT) == 8, "Point2fT must be two packed floats (cv::Point2f)");
  const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
  vnMatches12.assign(n1, -1);
  std::vector<uint8_t> t1, t2;
  float b[4];
  detail::frameBounds(F2, b);
  int nmatches = 0;
  check(orbfe_search_for_initialization(
      ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(F1.mvKeysUn.data()),
      detail::packedDescriptors(F1.mDescriptors, n1, t1), n1,
      reinterpret_cast<const OrbfeKeyPoint*>(F2.mvKeysUn.data()), detail::packedDescriptors(F2.mDescriptors, n2, t2),
      n2, b, reinterpret_cast<float*>(vbPrevMatched.data()), vnMatches12.data(), windowSize, mfNNratio,
      mbCheckOrientation ? 1 : 0, &nmatches));
  return nmatches;
}

// int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)
// (ORBmatcher.cc:45-124).  MapPoint fields are snapshotted through the reference's own accessors
// (isBad(), GetDescriptor(), Observations() take the MapPoint mutexes, MapPoint.cc:126-129,294-298)
// BEFORE the GPU call; assignments are written back to F.mvpMapPoints afterwards.
template <class FrameT, class MapPointT>
inline int SearchByProjection(MatcherContext& ctx, float mfNNratio, FrameT& F,
                              const std::vector<MapPointT*>& vpMapPoints, const float th) {
  const int n = (int)F.mvKeysUn.size(), nmp = (int)vpMapPoints.size();
  std::vector<uint8_t> occ(n, 0), flags(nmp, 0), mdesc((size_t)nmp * 32, 0), tmp;
  std::vector<float> xy((size_t)nmp * 2, 0.f), vcos(nmp, 0.f);
  std::vector<int32_t> lvl(nmp, 0), assigned(n, -1);
  for (int i = 0; i < n; i++)
    if (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) occ[i] = 1;
  for (int i = 0; i < nmp; i++) {
    MapPointT* p = vpMapPoints[i];
    if (!p->mbTrackInView) continue;
    if (p->isBad()) { flags[i] = ORBFE_MP_IN_VIEW | ORBFE_MP_BAD; continue; }
    flags[i] = ORBFE_MP_IN_VIEW | (p->plCandidato ? ORBFE_MP_CANDIDATO : 0) |
               (p->Observations() > 0 ? ORBFE_MP_OBSERVED : 0);
    xy[2 * i] = p->mTrackProjX;
    xy[2 * i + 1] = p->mTrackProjY;
    lvl[i] = p->mnTrackScaleLevel;
    vcos[i] = p->mTrackViewCos;
    const auto d = p->GetDescriptor();
    std::memcpy(&mdesc[(size_t)i * 32], d.data, 32);
  }
  float b[4];
  detail::frameBounds(F, b);
  int nmatches = 0;
  check(orbfe_search_by_projection(ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(F.mvKeysUn.data()),
                                   detail::packedDescriptors(F.mDescriptors, n, tmp), n, b, F.mvScaleFactors.data(),
                                   (int)F.mvScaleFactors.size(), occ.data(), xy.data(), lvl.data(), vcos.data(),
                                   flags.data(), mdesc.data(), nmp, th, mfNNratio, assigned.data(), &nmatches));
  for (int i = 0; i < n; i++)
    if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];
  return nmatches;
}

// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) as far as the path uses it: loaded from the
// fork's binary vocabulary file (TemplatedVocabulary.h:1563-1640), resident in HBM.
class Vocabulary {
 public:
  explicit Vocabulary(int device = 0) : device_(device) {}
  ~Vocabulary() { orbfe_vocabulary_destroy(v_); }
  Vocabulary(const Vocabulary&) = delete;
  Vocabulary& operator=(const Vocabulary&) = delete;
  // bool loadFromBinaryFile(const std::string& filename)
  bool loadFromBinaryFile(const std::string& filename) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> image;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) image.insert(image.end(), buf, buf + got);
    std::fclose(f);
    orbfe_vocabulary* nv = nullptr;
    if (orbfe_vocabulary_create_from_image(device_, image.data(), image.size(), &nv) != ORBFE_OK) return false;
    orbfe_vocabulary_destroy(v_);
    v_ = nv;
    return true;
  }
  bool empty() const { return v_ == nullptr; }
  orbfe_vocabulary* get() const { return v_; }

  // void transform(const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup) const
  // (TemplatedVocabulary.h:1136-1204) on the rows of a CV_8U N x 32 matrix.  BowVectorT / FeatureVectorT are the
  // DBoW2 map types (std::map<WordId, WordValue>, std::map<NodeId, std::vector<unsigned int>>).
  template <class MatT, class BowVectorT, class FeatureVectorT>
  void transform(const MatT& descriptors, int n, BowVectorT& v, FeatureVectorT& fv, int levelsup) {
    v.clear();
    fv.clear();
    if (!v_ || n == 0) return;
    std::vector<uint8_t> tmp;
    ids_.resize(n); vals_.resize(n); nodes_.resize(n); offs_.resize(n + 1); feats_.resize(n);
    int nw = 0, nn = 0;
    check(orbfe_bow_transform(v_, detail::packedDescriptors(descriptors, n, tmp), n, 0, levelsup, ids_.data(),
                              vals_.data(), &nw, nodes_.data(), offs_.data(), feats_.data(), &nn, nullptr, nullptr));
    for (int i = 0; i < nw; i++) v.insert(v.end(), typename BowVectorT::value_type(ids_[i], vals_[i]));
    for (int i = 0; i < nn; i++) {
      auto it = fv.insert(fv.end(), typename FeatureVectorT::value_type(nodes_[i], typename FeatureVectorT::mapped_type()));
      it->second.assign(feats_.begin() + offs_[i], feats_.begin() + offs_[i + 1]);
    }
  }

 private:
  int device_;
  orbfe_vocabulary* v_ = nullptr;
  std::vector<uint32_t> ids_, nodes_, offs_, feats_;
  std::vector<double> vals_;
};

// void Frame::ComputeBoW()   (Frame.cc:277-284); KeyFrame::ComputeBoW (KeyFrame.cc) is the same call.
template <class FrameT>
inline void ComputeBoW(Vocabulary& voc, FrameT& F) {
  if (F.mBowVec.empty()) voc.transform(F.mDescriptors, (int)F.mDescriptors.rows, F.mBowVec, F.mFeatVec, 4);
}

namespace detail {
template <class FeatureVectorT>
inline void flattenFeatureVector(const FeatureVectorT& fv, std::vector<uint32_t>& nodes, std::vector<uint32_t>& offs,
                                 std::vector<uint32_t>& feats) {
  nodes.clear(); offs.clear(); feats.clear();
  for (const auto& e : fv) {
    nodes.push_back((uint32_t)e.first);
    offs.push_back((uint32_t)feats.size());
    feats.insert(feats.end(), e.second.begin(), e.second.end());
  }
  offs.push_back((uint32_t)feats.size());
}
template <class KeyPointT>
inline void angles(const std::vector<KeyPointT>& k, std::vector<float>& a) {
  a.resize(k.size());
  for (size_t i = 0; i < k.size(); i++) a[i] = k[i].angle;
}
template <class MapPointT>
inline void validFlags(const std::vector<MapPointT*>& mps, std::vector<uint8_t>& valid) {
  valid.assign(mps.size(), 0);
  for (size_t i = 0; i < mps.size(); i++)
    if (mps[i] && !mps[i]->isBad()) valid[i] = 1;   // ORBmatcher.cc:191-196 / 553-557
}
}  // namespace detail

// int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  (ORBmatcher.cc:154-283)
template <class KeyFrameT, class FrameT, class MapPointT>
inline int SearchByBoW(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, KeyFrameT* pKF, FrameT& F,
                       std::vector<MapPointT*>& vpMapPointMatches) {
  const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
  const int n1 = (int)vpMapPointsKF.size(), n2 = (int)F.N;
  vpMapPointMatches.assign(n2, static_cast<MapPointT*>(nullptr));
  std::vector<uint8_t> valid1, t1, t2;
  std::vector<float> a1, a2;
  std::vector<uint32_t> n1v, o1v, f1v, n2v, o2v, f2v;
  detail::validFlags(vpMapPointsKF, valid1);
  detail::angles(pKF->mvKeysUn, a1);
  detail::angles(F.mvKeys, a2);
  detail::flattenFeatureVector(pKF->mFeatVec, n1v, o1v, f1v);
  detail::flattenFeatureVector(F.mFeatVec, n2v, o2v, f2v);
  std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
  int nmatches = 0;
  check(orbfe_search_by_bow(ctx.get(), detail::packedDescriptors(pKF->mDescriptors, n1, t1), a1.data(), valid1.data(), n1,
                            n1v.data(), o1v.data(), f1v.data(), (int)n1v.size(),
                            detail::packedDescriptors(F.mDescriptors, n2, t2), a2.data(), nullptr, n2, n2v.data(),
                            o2v.data(), f2v.data(), (int)n2v.size(), mfNNratio, mbCheckOrientation ? 1 : 0, 0, m12.data(),
                            &nmatches));
  for (int i = 0; i < n1; i++)
    if (m12[i] >= 0) vpMapPointMatches[m12[i]] = vpMapPointsKF[i];
  return nmatches;
}

// int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)  (ORBmatcher.cc:517-650)
template <class KeyFrameT, class MapPointT>
inline int SearchByBoW(MatcherContext& ctx, float mfNNratio, bool mbCheckOrientation, KeyFrameT* pKF1, KeyFrameT* pKF2,
                       std::vector<MapPointT*>& vpMatches12) {
  const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
  const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
  vpMatches12.assign(n1, static_cast<MapPointT*>(nullptr));
  std::vector<uint8_t> valid1, valid2, t1, t2;
  std::vector<float> a1, a2;
  std::vector<uint32_t> n1v, o1v, f1v, n2v, o2v, f2v;
  detail::validFlags(vpMapPoints1, valid1);
  detail::validFlags(vpMapPoints2, valid2);
  detail::angles(pKF1->mvKeysUn, a1);
  detail::angles(pKF2->mvKeysUn, a2);
  detail::flattenFeatureVector(pKF1->mFeatVec, n1v, o1v, f1v);
  detail::flattenFeatureVector(pKF2->mFeatVec, n2v, o2v, f2v);
  std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
  if (valid2.empty()) valid2.push_back(0);
  int nmatches = 0;
  check(orbfe_search_by_bow(ctx.get(), detail::packedDescriptors(pKF1->mDescriptors, n1, t1), a1.data(), valid1.data(), n1,
                            n1v.data(), o1v.data(), f1v.data(), (int)n1v.size(),
                            detail::packedDescriptors(pKF2->mDescriptors, n2, t2), a2.data(), valid2.data(), n2,
                            n2v.data(), o2v.data(), f2v.data(), (int)n2v.size(), mfNNratio, mbCheckOrientation ? 1 : 0, 1,
                            m12.data(), &nmatches));
  for (int i = 0; i < n1; i++)
    if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
  return nmatches;
}

// int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
//                                        vector<pair<size_t,size_t>>& vMatchedPairs)   (ORBmatcher.cc:652-804).
// The body keeps the reference's epipole lines (:657-666) and passes (ex, ey); F12 is read through .at<float>(r,c).
template <class KeyFrameT, class MatT>
inline int SearchForTriangulation(MatcherContext& ctx, bool mbCheckOrientation, KeyFrameT* pKF1, KeyFrameT* pKF2,
                                  const MatT& F12, float ex, float ey,
                                  std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
  const int n1 = (int)pKF1->N, n2 = (int)pKF2->N;
  std::vector<uint8_t> has1(n1 > 0 ? n1 : 1, 0), has2(n2 > 0 ? n2 : 1, 0), t1, t2;
  for (int i = 0; i < n1; i++) has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;   // ORBmatcher.cc:704-709
  for (int i = 0; i < n2; i++) has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;   // :723-727
  std::vector<uint32_t> n1v, o1v, f1v, n2v, o2v, f2v;
  detail::flattenFeatureVector(pKF1->mFeatVec, n1v, o1v, f1v);
  detail::flattenFeatureVector(pKF2->mFeatVec, n2v, o2v, f2v);
  float F[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) F[3 * r + c] = F12.template at<float>(r, c);
  std::vector<int32_t> pairs((size_t)(n1 > 0 ? n1 : 1) * 2, -1);
  int nmatches = 0;
  check(orbfe_search_for_triangulation(
      ctx.get(), reinterpret_cast<const OrbfeKeyPoint*>(pKF1->mvKeysUn.data()), detail::packedDescriptors(pKF1->mDescriptors, n1, t1),
      has1.data(), n1, n1v.data(), o1v.data(), f1v.data(), (int)n1v.size(),
      reinterpret_cast<const OrbfeKeyPoint*>(pKF2->mvKeysUn.data()), detail::packedDescriptors(pKF2->mDescriptors, n2, t2),
      has2.data(), n2, n2v.data(), o2v.data(), f2v.data(), (int)n2v.size(), F, ex, ey, pKF2->mvScaleFactors.data(),
      pKF2->mvLevelSigma2.data(), (int)pKF2->mvScaleFactors.size(), mbCheckOrientation ? 1 : 0, pairs.data(), &nmatches));
  vMatchedPairs.clear();
  vMatchedPairs.reserve(nmatches);
  for (int i = 0; i < nmatches; i++) vMatchedPairs.push_back(std::make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]));
  return nmatches;
}

}  // namespace orbfe
