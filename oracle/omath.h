// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// omath.h: transcendental functions of the hot path.  The reference calls the host libm
// (std::sin/cos in src/math/sampling.hpp:28-33 and src/bsdf/microfacet.hpp:363-364, std::pow /
// std::exp in src/bsdf/sheen.hpp:32,48,63, std::log in src/bsdf/params.hpp:92), so its bits
// depend on the glibc it is linked with.  The oracle pins ONE definition: each function is
// evaluated in IEEE binary64 with only + - * fma and integer bit moves (fdlibm-style kernels),
// then rounded to fp32 once.  That is within 1 ulp of any faithful libm (test_oracle_math checks
// it against this box's glibc) and — because it uses no library call — can be reproduced bit for
// bit by the HIP kernels, which carry their own copy of the same recipe.
// ORC_USE_LIBM=1 switches back to the host libm (used to measure how much that choice matters).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {
namespace m {

inline double bits_to_double(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }
inline uint64_t double_to_bits(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }

// round-to-nearest-even of a double that is known to be |x| < 2^51
inline double rint_small(double x) {
  const double big = 6755399441055744.0;  // 1.5 * 2^52
  return (x + big) - big;
}

// sin and cos of x (any finite |x| < ~1e5; the path only uses [0, 2*pi]) in binary64
inline void sincos_d(double x, double* s, double* c) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  double k = rint_small(x * two_over_pi);
  double r = std::fma(-k, pio2_hi, x);
  r = std::fma(-k, pio2_lo, r);
  double z = r * r;
  // fdlibm __kernel_sin / __kernel_cos minimax coefficients on [-pi/4, pi/4]
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double ps = std::fma(z, std::fma(z, std::fma(z, std::fma(z, S6, S5), S4), S3), S2);
  double sr = std::fma(z * r, std::fma(z, ps, S1), r);
  double pc = std::fma(z, std::fma(z, std::fma(z, std::fma(z, std::fma(z, C6, C5), C4), C3), C2), C1);
  double cr = std::fma(z * z, pc, std::fma(-0.5, z, 1.0));
  int q = (int)((int64_t)k & 3);
  double ss, cc;
  switch (q) {
    case 0: ss = sr; cc = cr; break;
    case 1: ss = cr; cc = -sr; break;
    case 2: ss = -sr; cc = -cr; break;
    default: ss = -cr; cc = sr; break;
  }
  *s = ss; *c = cc;
}

// exp(x) in binary64 for |x| < 700
inline double exp_d(double x) {
  const double inv_ln2 = 1.44269504088896338700e+00;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  double k = rint_small(x * inv_ln2);
  double r = std::fma(-k, ln2_hi, x);
  r = std::fma(-k, ln2_lo, r);
  // Taylor to r^13/13!  (|r| <= 0.3466: truncation < 1e-17)
  double p = 1.0 / 6227020800.0;
  p = std::fma(p, r, 1.0 / 479001600.0);
  p = std::fma(p, r, 1.0 / 39916800.0);
  p = std::fma(p, r, 1.0 / 3628800.0);
  p = std::fma(p, r, 1.0 / 362880.0);
  p = std::fma(p, r, 1.0 / 40320.0);
  p = std::fma(p, r, 1.0 / 5040.0);
  p = std::fma(p, r, 1.0 / 720.0);
  p = std::fma(p, r, 1.0 / 120.0);
  p = std::fma(p, r, 1.0 / 24.0);
  p = std::fma(p, r, 1.0 / 6.0);
  p = std::fma(p, r, 0.5);
  p = std::fma(p, r, 1.0);
  p = std::fma(p, r, 1.0);
  int64_t ki = (int64_t)k;
  if (ki < -1000) return 0.0;
  if (ki > 1000) return bits_to_double(0x7ff0000000000000ull);
  // scale by 2^k in two steps so that subnormal results stay representable
  int64_t k1 = ki / 2, k2 = ki - k1;
  double s1 = bits_to_double((uint64_t)(1023 + k1) << 52);
  double s2 = bits_to_double((uint64_t)(1023 + k2) << 52);
  return p * s1 * s2;
}

// natural log of a positive, finite, normal binary64
inline double log_d(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  uint64_t u = double_to_bits(x);
  int64_t e = (int64_t)((u >> 52) & 0x7ff) - 1023;
  uint64_t mant = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double mval = bits_to_double(mant);          // [1,2)
  if (mval > 1.41421356237309514547) {          // bring to [sqrt(1/2), sqrt(2))
    mval = mval * 0.5;
    e += 1;
  }
  double f = mval - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  // 2*atanh(s) = 2s (1 + z/3 + z^2/5 + ... ), |s| <= 0.1716
  double p = 1.0 / 27.0;
  p = std::fma(p, z, 1.0 / 25.0);
  p = std::fma(p, z, 1.0 / 23.0);
  p = std::fma(p, z, 1.0 / 21.0);
  p = std::fma(p, z, 1.0 / 19.0);
  p = std::fma(p, z, 1.0 / 17.0);
  p = std::fma(p, z, 1.0 / 15.0);
  p = std::fma(p, z, 1.0 / 13.0);
  p = std::fma(p, z, 1.0 / 11.0);
  p = std::fma(p, z, 1.0 / 9.0);
  p = std::fma(p, z, 1.0 / 7.0);
  p = std::fma(p, z, 1.0 / 5.0);
  p = std::fma(p, z, 1.0 / 3.0);
  p = std::fma(p, z, 1.0);
  double lm = 2.0 * s * p;
  double ed = (double)e;
  return std::fma(ed, ln2_hi, std::fma(ed, ln2_lo, lm));
}

#ifndef ORC_USE_LIBM
#define ORC_USE_LIBM 0
#endif

// fp32 entry points used by the restatement
inline float sinf_(float x) {
#if ORC_USE_LIBM
  return std::sin(x);
#else
  double s, c; sincos_d((double)x, &s, &c); return (float)s;
#endif
}
inline float cosf_(float x) {
#if ORC_USE_LIBM
  return std::cos(x);
#else
  double s, c; sincos_d((double)x, &s, &c); return (float)c;
#endif
}
inline float expf_(float x) {
#if ORC_USE_LIBM
  return std::exp(x);
#else
  if (x != x) return x;
  if (x > 89.0f) return INFINITY;
  if (x < -104.0f) return 0.0f;
  return (float)exp_d((double)x);
#endif
}
inline float logf_(float x) {
#if ORC_USE_LIBM
  return std::log(x);
#else
  if (x != x || x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (std::isinf(x)) return x;
  return (float)log_d((double)x);  // fp32 subnormals are binary64 normals
#endif
}
// std::pow(float,float) for the sheen lobe: x >= 0 there (sin_theta, cos_theta in [0,1])
inline float powf_(float x, float y) {
#if ORC_USE_LIBM
  return std::pow(x, y);
#else
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return NAN;
  if (x == 1.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : INFINITY;
  if (x < 0.0f) return NAN;  // not reached by the path
  if (std::isinf(x)) return y > 0.0f ? INFINITY : 0.0f;
  if (std::isinf(y)) return ((x < 1.0f) == (y > 0.0f)) ? 0.0f : INFINITY;
  double t = (double)y * log_d((double)x);
  if (t > 89.0) return INFINITY;
  if (t < -104.0) return 0.0f;
  return (float)exp_d(t);
#endif
}

}  // namespace m
}  // namespace orc
