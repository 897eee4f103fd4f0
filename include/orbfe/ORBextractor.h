// ORBextractor.h -- drop-in ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:155-373) backed by
// liborbfe.so.  Same constructor, operator() and getters, so Frame.cc:69-75,133 and Tracking.cc:65-67
// compile unchanged.  Needs OpenCV headers (the signature uses cv:: types); orb_shim.hpp is the
// OpenCV-free core.
#pragma once
#if !__has_include(<opencv2/core/core.hpp>)
#error "include/orbfe/ORBextractor.h needs OpenCV headers; use include/orbfe/orb_shim.hpp (cv-free) instead"
#else
#include <opencv2/core/core.hpp>
#if __has_include(<opencv2/core/version.hpp>)
#include <opencv2/core/version.hpp>
#endif

#include <cstdlib>
#include <vector>

#include "orb_shim.hpp"

// cv::GaussianBlur's 8-bit taps changed between OpenCV releases (orbfe.h, orbfe_extractor_set_blur_variant).  The facade reproduces
// the error-diffused taps (sum 256, ORBFE_GAUSS_ED: every release since 4.1.1 / 3.4.7) unless the integrator opts in to the
// version-based choice with -DORBFE_FACADE_GAUSS_BY_CV_VERSION: then a translation unit compiled against 4.0.0 - 4.1.0 or
// 3.4.2 - 3.4.6 gets the taps rounded one by one (sum 257, ORBFE_GAUSS_ROUNDED) -- the blur the reference's own
// ORBextractor.cc:950 would have called there.  Opt-in, because those taps, the version boundaries and the saturation at 255 are
// restated from the upstream sources and have not met a real OpenCV 4.0.x build on this pool; pin them with one
// GaussianBlur known-answer vector from such a build before relying on the switch.
#if defined(ORBFE_FACADE_GAUSS_BY_CV_VERSION) && defined(CV_VERSION_MAJOR) &&                                                     \
    ((CV_VERSION_MAJOR == 4 && (CV_VERSION_MINOR == 0 || (CV_VERSION_MINOR == 1 && CV_VERSION_REVISION == 0))) ||               \
     (CV_VERSION_MAJOR == 3 && CV_VERSION_MINOR == 4 && CV_VERSION_REVISION >= 2 && CV_VERSION_REVISION <= 6))
#define ORBFE_FACADE_GAUSS_VARIANT ORBFE_GAUSS_ROUNDED
#else
#define ORBFE_FACADE_GAUSS_VARIANT ORBFE_GAUSS_ED
#endif

namespace ORB_SLAM2 {

class ORBextractor {
 public:
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : impl_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, orbfe::detail::defaultDevice()) {   // ORBFE_DEVICE, as the matcher
    mvImagePyramid.resize(nlevels);  // kept for source compatibility; the pyramid lives in HBM
    if (!std::getenv("ORBFE_GAUSS_VARIANT")) impl_.SetBlurVariant(ORBFE_FACADE_GAUSS_VARIANT);   // the environment wins (orbfe.h)
  }
  ~ORBextractor() {}

  // ORBextractor.cc:907-969.  `mask` is ignored, as in the reference.
  void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                  cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    impl_.extract(image.data, image.rows, image.cols, image.step, _keypoints, desc_);
    const int n = (int)_keypoints.size();
    if (n == 0) {
      _descriptors.release();
    } else {
      _descriptors.create(n, 32, CV_8U);
      cv::Mat d = _descriptors.getMat();
      for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc_[(size_t)i * 32], 32);
    }
  }

  int inline GetLevels() { return impl_.GetLevels(); }
  float inline GetScaleFactor() { return impl_.GetScaleFactor(); }
  std::vector<float> inline GetScaleFactors() { return impl_.GetScaleFactors(); }
  std::vector<float> inline GetInverseScaleFactors() { return impl_.GetInverseScaleFactors(); }
  std::vector<float> inline GetScaleSigmaSquares() { return impl_.GetScaleSigmaSquares(); }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return impl_.GetInverseScaleSigmaSquares(); }

  std::vector<cv::Mat> mvImagePyramid;  // ORBextractor.h:245; no external reader in the reference

 private:
  orbfe::Extractor impl_;
  std::vector<uint8_t> desc_;
};

}  // namespace ORB_SLAM2
#endif
