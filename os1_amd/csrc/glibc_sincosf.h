// glibc_sincosf.h -- (cosf, sinf) bit-compatible with glibc >= 2.28's float routines, usable in
// device code.
//
// Why: computeOrbDescriptor evaluates `(float)cos(angle), (float)sin(angle)` on a float argument
// (reference src/ORBextractor.cc:136-137; `using namespace std` makes these std::cos(float) ==
// libm cosf).  glibc's cosf/sinf are NOT correctly rounded (<= 0.56 ULP), so a device libm would
// disagree in the last bit for some angles and flip rBRIEF sample coordinates.  glibc's algorithm
// (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h -- the ARM "optimized routines" sincosf) is
// a short double-precision polynomial; this header restates its published algorithm for the
// range the rBRIEF kernel needs (|x| < 120; the argument is angle_deg*pi/180 in [0, 2*pi)).
// tests/test_host_logic.py (test_sincos_restatement_matches_libm_exhaustively) checks it exhaustively (every float in [0, 6.3]) against the host libm on
// CPU, and tests/test_gpu_parity.py checks the device build against host libm on the GPU box.
// Plain mul/add and fused variants both agree with glibc bit-for-bit over that range, so the
// result does not depend on -ffp-contract.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define ORBFE_HD __host__ __device__
#else
#define ORBFE_HD
#endif

namespace orbfe {

ORBFE_HD inline uint32_t sc_abstop12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ff; }

// n even: sine polynomial, n odd: cosine polynomial; neg selects the negated cosine coefficients.
ORBFE_HD inline float sc_poly(double x, double x2, bool neg, int n) {
  const double sg = neg ? -1.0 : 1.0;
  if ((n & 1) == 0) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    double x3 = x * x2;
    double s1 = S2 + x2 * S3;
    double x7 = x3 * x2;
    double s = x + x3 * S1;
    return (float)(s + x7 * s1);
  } else {
    const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5,
                 C3 = sg * -0x1.6c087e89a359dp-10, C4 = sg * 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double c2 = C3 + x2 * C4;
    double c1 = C0 + x2 * C1;
    double x6 = x4 * x2;
    double c = c1 + x4 * C2;
    return (float)(c + x6 * c2);
  }
}

// Valid for |y| < 120 (asserted by the callers' domain: y in [0, 2*pi)).
ORBFE_HD inline void sincosf_glibc(float y, float* sinp, float* cosp) {
  double x = y;
  if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
    double x2 = x * x;
    if (sc_abstop12(y) < sc_abstop12(0x1p-12f)) {
      *sinp = y;
      *cosp = 1.0f;
      return;
    }
    *sinp = sc_poly(x, x2, false, 0);
    *cosp = sc_poly(x, x2, false, 1);
    return;
  }
  // reduce_fast: quotient prescaled by 2^24 so it sits in the low bits after truncation
  const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
  double r = x * HPI_INV;
  int n = ((int32_t)r + 0x800000) >> 24;
  x = x - n * HPI;
  const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // {1,-1,-1,1}[n&3]
  const bool neg = (n & 2) != 0;
  const double xs = x * sign, x2 = x * x;
  *sinp = sc_poly(xs, x2, neg, n);
  *cosp = sc_poly(xs, x2, neg, n ^ 1);
}

}  // namespace orbfe
