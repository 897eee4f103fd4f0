// ORBmatcher.h -- drop-in ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:61-270) backed by liborbfe.so.
// Same constructor, same member functions, same signatures: put this file in place of the reference's
// include/ORBmatcher.h and drop src/ORBmatcher.cc from the build; Tracking.cc, LocalMapping.cc, LoopClosing.cc and
// MapPoint.cc compile unchanged.  Every search runs on the GPU through the C ABI (include/orbfe.h); the MapPoint /
// KeyFrame bookkeeping around it is the reference's, statement for statement (include/orbfe/orb_shim.hpp).
//
// Needs OpenCV headers and the reference's own Frame.h / KeyFrame.h / MapPoint.h (the signatures use their types).
// The projections use CvOps below, whose members ARE the reference's cv::Mat expressions, so the arithmetic is that
// of the OpenCV the application links -- not a restatement.
#pragma once
#if !__has_include(<opencv2/core/core.hpp>)
#error "include/orbfe/ORBmatcher.h needs OpenCV headers; use include/orbfe/orb_shim.hpp (cv-free) instead"
#else
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <cstdlib>
#include <set>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "orb_shim.hpp"

namespace ORB_SLAM2 {

namespace orbfe_detail {
// The matrix arithmetic of the pose-driven searches as genuine cv::Mat expressions (the ones ORBmatcher.cc writes).
struct CvOps {
  static cv::Mat m33(const float* p) { return cv::Mat(3, 3, CV_32F, const_cast<float*>(p)); }
  static cv::Mat v3(const float* p) { return cv::Mat(3, 1, CV_32F, const_cast<float*>(p)); }
  static void out(const cv::Mat& r, float* d, int n) { for (int i = 0; i < n; i++) d[i] = r.at<float>(i); }
  static void gemm3(const float A[9], const float b[3], double alpha, const float* c, double, float d[3]) {
    cv::Mat r;
    if (c) r = m33(A) * v3(b) + v3(c);             // Rcw*x3Dw+tcw
    else if (alpha < 0) r = -m33(A) * v3(b);       // -sR21*t12, -Rwc*tcw
    else r = m33(A) * v3(b);
    out(r, d, 3);
  }
  static void gemmT3(const float A[9], const float b[3], double, float d[3]) {
    cv::Mat r = -m33(A).t() * v3(b);               // -Rcw.t()*tcw
    out(r, d, 3);
  }
  static double norm3(const float v[3]) { return cv::norm(v3(v)); }
  static double dot3(const float a[3], const float b[3]) { return v3(a).dot(v3(b)); }
  static void scale(const float* M, int n, double s, float* o) {
    cv::Mat r = s * cv::Mat(n, 1, CV_32F, const_cast<float*>(M));          // s12*R12, (1.0/s12)*R12.t()
    out(r, o, n);
  }
  static void divide(const float* M, int n, double s, float* o) {
    cv::Mat r = cv::Mat(n, 1, CV_32F, const_cast<float*>(M)) / s;          // sRcw/scw
    out(r, o, n);
  }
};
inline orbfe::MatcherContext& context() {   // ORBmatcher objects live on the stack per use; the GPU context is per thread
  thread_local orbfe::MatcherContext ctx(orbfe::detail::defaultDevice());   // ORBFE_DEVICE, as the extractor
  return ctx;
}
}  // namespace orbfe_detail

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbfe::DescriptorDistance(a, b); }

  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) {
    return orbfe::SearchByProjection(ctx(), mfNNratio, F, vpMapPoints, th);
  }
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th) {
    return orbfe::SearchByProjection<Ops>(ctx(), mbCheckOrientation, CurrentFrame, LastFrame, th);
  }
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                         const int ORBdist) {
    return orbfe::SearchByProjection<Ops>(ctx(), mbCheckOrientation, CurrentFrame, pKF, sAlreadyFound, th, ORBdist);
  }
  int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                         int th) {
    return orbfe::SearchByProjection<Ops>(ctx(), pKF, Scw, vpPoints, vpMatched, th);
  }
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    return orbfe::SearchByBoW(ctx(), mfNNratio, mbCheckOrientation, pKF, F, vpMapPointMatches);
  }
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    return orbfe::SearchByBoW(ctx(), mfNNratio, mbCheckOrientation, pKF1, pKF2, vpMatches12);
  }
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    return orbfe::SearchForInitialization(ctx(), mfNNratio, mbCheckOrientation, F1, F2, vbPrevMatched, vnMatches12, windowSize);
  }
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
    // epipole of camera 1 in image 2 (ORBmatcher.cc:658-666)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    return orbfe::SearchForTriangulation(ctx(), mbCheckOrientation, pKF1, pKF2, F12, ex, ey, vMatchedPairs);
  }
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                   const cv::Mat& t12, const float th) {
    return orbfe::SearchBySim3<Ops>(ctx(), pKF1, pKF2, vpMatches12, s12, R12, t12, th);
  }
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0) {
    return orbfe::Fuse<Ops>(ctx(), pKF, vpMapPoints, th);
  }
  int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {
    return orbfe::Fuse<Ops>(ctx(), pKF, Scw, vpPoints, th, vpReplacePoint);
  }

 public:
  static const int TH_LOW = 50;         // ORBmatcher.cc:37-39
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;

 protected:
  typedef orbfe_detail::CvOps Ops;
  static orbfe::MatcherContext& ctx() { return orbfe_detail::context(); }
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2
#endif
