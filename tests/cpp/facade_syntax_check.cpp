// facade_syntax_check.cpp -- instantiates every member of the cv-typed facades (include/orbfe/ORBextractor.h,
// include/orbfe/ORBmatcher.h) against the declaration-level stand-ins in tests/cpp/opencv_stub and tests/cpp/os1_stub.
// Compiled with -fsyntax-only (tests/test_facade.py): it proves that the facades' signatures are the reference's
// (the calls below are the reference's call sites) and that every template they pull from orb_shim.hpp type-checks with
// cv::Mat / cv::KeyPoint / ORB_SLAM2::Frame / KeyFrame / MapPoint.  Never linked, never run.
#include "orbfe/ORBextractor.h"
#include "orbfe/ORBmatcher.h"

using namespace ORB_SLAM2;

// With -DORBFE_FACADE_GAUSS_BY_CV_VERSION the facade reproduces the GaussianBlur of the OpenCV it is compiled against
// (ORBextractor.cc:950), otherwise the error-diffused taps: the test compiles this file once per release named on the command line
// (-DCV_VERSION_MAJOR=.. -DEXPECT_GAUSS=..), with and without the opt-in.
#ifdef EXPECT_GAUSS
static_assert(ORBFE_FACADE_GAUSS_VARIANT == EXPECT_GAUSS, "blur variant: error-diffused unless chosen from CV_VERSION_* by opt-in");
#else
static_assert(ORBFE_FACADE_GAUSS_VARIANT == ORBFE_GAUSS_ED, "no version macros: the default");
#endif

int useEverything(Frame& F, Frame& F2, KeyFrame* pKF, KeyFrame* pKF2, std::vector<MapPoint*>& vp, std::set<MapPoint*>& sp,
                  cv::Mat& im, cv::Mat& Scw, std::vector<cv::Point2f>& prev, std::vector<int>& m12,
                  std::vector<std::pair<size_t, size_t> >& pairs) {
  ORBextractor* ex = new ORBextractor(2000, 1.2f, 8, 20, 7);                 // Tracking.cc:65
  (*ex)(im, cv::Mat(), F.mvKeys, F.mDescriptors);                            // Frame.cc:133
  int n = ex->GetLevels();
  n += (int)ex->GetScaleFactor();
  F.mvScaleFactors = ex->GetScaleFactors();                                  // Frame.cc:69-75
  F.mvScaleFactors = ex->GetInverseScaleFactors();
  F.mvScaleFactors = ex->GetScaleSigmaSquares();
  F.mvScaleFactors = ex->GetInverseScaleSigmaSquares();
  ORBmatcher matcher(0.9, true);                                             // Tracking.cc:383
  n += matcher.SearchForInitialization(F, F2, prev, m12, 100);               // Tracking.cc:384
  n += matcher.SearchByProjection(F, vp, 3);                                 // Tracking.cc:818-824
  n += matcher.SearchByProjection(F, F2, 15.f);                              // Tracking.cc:608
  n += matcher.SearchByProjection(F, pKF, sp, 10.f, 100);                    // Tracking.cc:1066
  n += matcher.SearchByProjection(pKF, Scw, vp, vp, 10);                     // LoopClosing.cc:359
  n += matcher.SearchByBoW(pKF, F, vp);                                      // Tracking.cc:548
  n += matcher.SearchByBoW(pKF, pKF2, vp);                                   // LoopClosing.cc:262
  n += matcher.SearchForTriangulation(pKF, pKF2, Scw, pairs);                // LocalMapping.cc:226
  n += matcher.SearchBySim3(pKF, pKF2, vp, 1.f, Scw, Scw, 7.5f);             // LoopClosing.cc:326
  n += matcher.Fuse(pKF, vp);                                                // LocalMapping.cc:397
  n += matcher.Fuse(pKF, Scw, vp, 4.f, vp);                                  // LoopClosing.cc:606
  n += ORBmatcher::DescriptorDistance(im, im);                               // MapPoint.cc:266
  n += ORBmatcher::TH_LOW + ORBmatcher::TH_HIGH + ORBmatcher::HISTO_LENGTH;
  return n;
}
