// restated_ops_export.cpp -- TEST INFRASTRUCTURE: orbfe::detail::RestatedOps (the cv-free small-matrix arithmetic the shim
// uses when it is built without OpenCV, include/orbfe/orb_shim.hpp) behind a C entry point, so that Python tests can
// compare it with the oracle's restatement (tests/test_host_logic.py) and with a live OpenCV (tests/test_opencv_live.py).
//   build: g++ -std=c++17 -O2 -ffp-contract=off -shared -fPIC -Iinclude tests/cpp/restated_ops_export.cpp -o librestated_ops.so
#include "orbfe/orb_shim.hpp"

extern "C" void shim_cv_small(int op, const float* A, const float* b, double alpha, const float* c, double beta, float* out,
                              double* out1, int n) {
  typedef orbfe::detail::RestatedOps Ops;
  switch (op) {
    case 0: Ops::gemm3(A, b, alpha, c, beta, out); break;      // alpha * A * b + beta * c
    case 1: Ops::gemmT3(A, b, alpha, out); break;              // alpha * A.t() * b
    case 2: *out1 = Ops::norm3(b); break;                      // cv::norm(b)
    case 3: *out1 = Ops::dot3(A, b); break;                    // a.dot(b)
    case 4: Ops::scale(A, n, alpha, out); break;               // alpha * M   (n elements)
    default: Ops::divide(A, n, alpha, out); break;             // M / alpha
  }
}
