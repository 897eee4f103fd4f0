// TEST INFRASTRUCTURE ONLY -- a declaration-level stand-in for the handful of OpenCV core types that the cv-typed
// facades (include/orbfe/ORBextractor.h, include/orbfe/ORBmatcher.h) mention, so that their SYNTAX and name lookup can
// be checked on a machine without OpenCV (tests/test_facade.py).  Nothing here computes anything; nothing links it.
#pragma once
#include <cstddef>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) ((void)(expr))
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
class MatExpr;
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type);
  Mat(int rows, int cols, int type, void* data, size_t step = 0);
  Mat(const MatExpr&);
  unsigned char* data;
  int rows, cols;
  struct Step { operator size_t() const; } step;
  int type() const;
  bool empty() const;
  unsigned char* ptr(int r = 0);
  template <class T> T& at(int r);
  template <class T> T& at(int r, int c);
  template <class T> const T& at(int r) const;
  template <class T> const T& at(int r, int c) const;
  Mat rowRange(int a, int b) const;
  Mat colRange(int a, int b) const;
  Mat row(int r) const;
  Mat col(int c) const;
  Mat clone() const;
  MatExpr t() const;
  double dot(const Mat& m) const;
};
class MatExpr {
 public:
  operator Mat() const;
};
MatExpr operator*(const Mat&, const Mat&);
MatExpr operator*(const MatExpr&, const Mat&);
MatExpr operator*(double, const Mat&);
MatExpr operator*(double, const MatExpr&);
MatExpr operator+(const MatExpr&, const Mat&);
MatExpr operator-(const Mat&, const Mat&);
MatExpr operator-(const Mat&);
MatExpr operator-(const MatExpr&);
MatExpr operator/(const Mat&, double);
double norm(const Mat&);
class _InputArray {
 public:
  _InputArray(const Mat&);
  bool empty() const;
  Mat getMat() const;
};
class _OutputArray {
 public:
  _OutputArray(Mat&);
  void create(int rows, int cols, int type) const;
  void release() const;
  Mat getMat() const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
