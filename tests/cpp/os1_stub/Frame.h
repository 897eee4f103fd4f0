// TEST INFRASTRUCTURE ONLY -- member NAMES of the reference's Frame / KeyFrame / MapPoint as far as the ORBmatcher
// facade touches them (reference include/Frame.h, KeyFrame.h, MapPoint.h), declarations only, for the syntax check
// of include/orbfe/ORBmatcher.h in tests/test_facade.py.  In a real build the reference's own headers are used.
#pragma once
#include <map>
#include <opencv2/core/core.hpp>
#include <set>
#include <vector>
namespace DBoW2 {
typedef std::map<unsigned int, double> BowVector;
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;
}
namespace ORB_SLAM2 {
class KeyFrame;
class MapPoint {
 public:
  bool mbTrackInView, plCandidato;
  float mTrackProjX, mTrackProjY, mTrackViewCos;
  int mnTrackScaleLevel;
  cv::Mat GetWorldPos();
  cv::Mat GetNormal();
  cv::Mat GetDescriptor();
  float GetMinDistanceInvariance();
  float GetMaxDistanceInvariance();
  int PredictScale(const float& currentDist, const float& logScaleFactor);
  bool isBad();
  int Observations();
  bool IsInKeyFrame(KeyFrame* pKF);
  int GetIndexInKeyFrame(KeyFrame* pKF);
  void AddObservation(KeyFrame* pKF, size_t idx);
  void Replace(MapPoint* pMP);
};
class Frame {
 public:
  long unsigned int mnId;   // Frame.h:399
  int N;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  cv::Mat mDescriptors, mTcw;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<float> mvScaleFactors;
  float mfLogScaleFactor;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY;
};
class KeyFrame {
 public:
  long unsigned int mnId;   // KeyFrame.h:574
  const int N;
  const std::vector<cv::KeyPoint> mvKeys, mvKeysUn;   // KeyFrame.h:653, 662
  const cv::Mat mDescriptors;
  const float fx, fy, cx, cy, mfLogScaleFactor;
  const int mnMinX, mnMinY, mnMaxX, mnMaxY;
  const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  cv::Mat GetRotation();
  cv::Mat GetTranslation();
  cv::Mat GetCameraCenter();
  bool IsInImage(const float& x, const float& y) const;
  MapPoint* GetMapPoint(const size_t& idx);
  void AddMapPoint(MapPoint* pMP, const size_t& idx);
  std::vector<MapPoint*> GetMapPointMatches();
  std::set<MapPoint*> GetMapPoints();
};
}  // namespace ORB_SLAM2
