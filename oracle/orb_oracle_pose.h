/* orb_oracle_pose.h -- C interface of the oracle's whole-function restatements of the pose-driven ORBmatcher searches
 * (oracle/orb_oracle.cpp, last section).  TEST INFRASTRUCTURE ONLY: included by the C++ tests under tests/cpp, never by os1_amd/. */
#ifndef ORB_ORACLE_POSE_H_
#define ORB_ORACLE_POSE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } OrcKp;   /* = cv::KeyPoint */

/* A Frame / KeyFrame as the searches read it. */
typedef struct {
  const OrcKp* kpsUn;          /* mvKeysUn */
  const uint8_t* desc;         /* mDescriptors, 32-byte rows */
  int n;
  float bounds[4];             /* mnMinX, mnMaxX, mnMinY, mnMaxY */
  float fx, fy, cx, cy;
  const float* scaleFactors;   /* mvScaleFactors */
  const float* invLevelSigma2; /* mvInvLevelSigma2 */
  int nlevels;
  float logScaleFactor;        /* mfLogScaleFactor */
} OrcView;

/* MapPoint table: the fields the searches read plus the bookkeeping state of the SIMPLIFIED MODEL the facade tests
 * use for MapPoint::Replace / AddObservation (documented in tests/cpp/facade_pose_test.cpp): a point observes the
 * keyframe under test at slot idxInKF (-1 = not), nObs counts its observations in all keyframes. */
typedef struct {
  int M;
  const float* pos;            /* 3M  GetWorldPos()                 */
  const float* normal;         /* 3M  GetNormal()                   */
  const float* mfMinDistance;  /* M   raw field: GetMinDistanceInvariance() = 0.8f * it (MapPoint.cc:358-362) */
  const float* mfMaxDistance;  /* M   raw field: GetMaxDistanceInvariance() = 1.2f * it (:364-368); PredictScale
                                      divides THIS, not the invariance bound (:370-379) */
  const uint8_t* desc;         /* 32M GetDescriptor()               */
  uint8_t* bad;                /* M   isBad()            (in/out)   */
  int32_t* nObs;               /* M   Observations()     (in/out)   */
  int32_t* idxInKF;            /* M   GetIndexInKeyFrame (in/out)   */
} OrcPoints;

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th)  (ORBmatcher.cc:1292-1423).
 * last_mp[i] = MapPoint id of LastFrame.mvpMapPoints[i] or -1, last_outlier[i] = mvbOutlier; cur_mp (in/out) =
 * CurrentFrame.mvpMapPoints as ids. */
int orc_sbp_frame(const OrcView* cur, const float Tcw[16], const OrcKp* lastKeys, const OrcKp* lastKeysUn, int nLast,
                  const int32_t* last_mp, const uint8_t* last_outlier, OrcPoints* P, int32_t* cur_mp, float th,
                  int check_orientation);
/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>& sAlreadyFound, th, ORBdist)  (:1425-1552).
 * kf_mp[i] = pKF->GetMapPointMatches()[i] as id or -1; already[id] != 0 iff the point is in sAlreadyFound. */
int orc_sbp_keyframe(const OrcView* cur, const float Tcw[16], const OrcKp* kfKeysUn, int nKF, const int32_t* kf_mp,
                     const uint8_t* already, OrcPoints* P, int32_t* cur_mp, float th, int ORBdist, int check_orientation);
/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th)  (:285-398). */
int orc_sbp_scw(const OrcView* kf, const float Scw[16], const int32_t* points, int npoints, OrcPoints* P,
                int32_t* vpMatched, int th);
/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th)  (:806-939).  slot (in/out) = the
 * keyframe's mvpMapPoints as ids; cand[i] = id or -1 (NULL). */
int orc_fuse(const OrcView* kf, const float Tcw[16], const int32_t* cand, int ncand, OrcPoints* P, int32_t* slot, float th);
/* ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint)  (:941-1064); replace_out[i] = id or -1. */
int orc_fuse_scw(const OrcView* kf, const float Scw[16], const int32_t* points, int npoints, OrcPoints* P, int32_t* slot,
                 float th, int32_t* replace_out);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)  (:1066-1290).  mp1 / mp2 = GetMapPointMatches()
 * as ids; matches12 (in/out) ids; idxInKF of the table refers to pKF2 (GetIndexInKeyFrame(pKF2), :1101). */
int orc_search_by_sim3(const OrcView* kf1, const float T1w[16], const int32_t* mp1, const OrcView* kf2, const float T2w[16],
                       const int32_t* mp2, OrcPoints* P, int32_t* matches12, float s12, const float R12[9],
                       const float t12[3], float th);
/* cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) as Frame::UndistortKeyPoints / ComputeImageBounds call it
 * (Frame.cc:286-353): n (x, y) float pairs in place; dist = k1 k2 p1 p2 [k3 [k4 k5 k6]]. */
void orc_undistort_pinhole(float* xy, int n, float fx, float fy, float cx, float cy, const float* dist, int ndist);
/* Frame::ComputeImageBounds (Frame.cc:322-353): mode 0 pinhole, 1 equidistant fisheye. */
void orc_image_bounds(int cols, int rows, int mode, float fx, float fy, float cx, float cy, const float* dist, int ndist,
                      float bounds[4]);

/* The extractor and the two array-form searches a Tracking-shaped C++ test needs beside the whole-function restatements
 * above (definitions and line citations in orb_oracle.cpp): ORBextractor ctor + operator() (ORBextractor.cc:442-502, 907-969),
 * ORBmatcher::SearchForInitialization (ORBmatcher.cc:400-515; prev_xy = vbPrevMatched in/out) and
 * ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (:45-132; mp_flags bit0 mbTrackInView, bit1 isBad(),
 * bit2 plCandidato, bit3 Observations() > 0; kp_assigned[idx] = index of the last MapPoint written to mvpMapPoints[idx]). */
#ifndef ORB_ORACLE_IMPLEMENTATION   /* (orb_oracle.cpp defines them on its own keypoint type of the same layout) */
void* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void orc_extractor_destroy(void* h);
int orc_extract(void* h, const uint8_t* gray, int rows, int cols, int stride, OrcKp* kps, uint8_t* desc, int cap);
int orc_search_for_initialization(const OrcKp* kps1, const uint8_t* desc1, int n1, const OrcKp* kps2, const uint8_t* desc2, int n2,
                                  const float bounds[4], float* prev_xy, int* vnMatches12, int windowSize, float mfNNratio,
                                  int mbCheckOrientation);
int orc_search_by_projection(const OrcKp* kpsUn, const uint8_t* desc, int n, const float bounds[4], const float* mvScaleFactors,
                             const uint8_t* kp_occupied, const float* mp_proj_xy, const int* mp_level, const float* mp_viewcos,
                             const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp, float th, float mfNNratio, int* kp_assigned);
#endif

#ifdef __cplusplus
}
#endif
#endif
