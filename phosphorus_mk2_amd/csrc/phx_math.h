// phx_math.h — fp32 vector arithmetic, deterministic transcendental functions and the counter-based
// sampler of the gfx950 device.  Everything here is bit-reproducible: only IEEE + - * / sqrt fma and
// integer operations, compiled with -ffp-contract=off, so the device's results can be compared exactly
// with a CPU evaluation of the same formulas.
//
// Semantics follow the reference: plain (unfused) Imath-style vector ops where the reference uses
// Imath (src/bsdf*.?pp, src/mesh.cpp, src/kernels/cpu/spt.hpp scalar parts) and explicit fused
// chains where it uses its AVX2 wrappers (src/math/simd/vector.hpp:98-109).
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PHX_HD __host__ __device__ __forceinline__
// sincosf_ / expf_ / logf_ / powf_ are real (leaf) functions on the device, not inlined into every lobe model that uses them:
// inlined, the 40 coefficients of their binary64 kernels (register pairs: a 64-bit literal cannot be an instruction operand) were
// hoisted in front of the lobe loops of bsdf_f / bsdf_sample and stayed alive across them - 46 VGPRs of constants in a kernel
// capped at 128.  The binary64 kernels themselves are inlined INTO those four, so no call is nested and no stack is needed; the
// arguments and results are fp32 (two or three registers per call).
#define PHX_HD_CALL inline __host__ __device__ __attribute__((noinline))
#else
#define PHX_HD inline
#define PHX_HD_CALL inline
#endif

namespace phx {

struct v3 {
  float x, y, z;
  PHX_HD v3() : x(0.f), y(0.f), z(0.f) {}
  PHX_HD explicit v3(float a) : x(a), y(a), z(a) {}
  PHX_HD v3(float a, float b, float c) : x(a), y(b), z(c) {}
};
PHX_HD v3 operator+(const v3& a, const v3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PHX_HD v3 operator-(const v3& a, const v3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PHX_HD v3 operator-(const v3& a) { return v3(-a.x, -a.y, -a.z); }
PHX_HD v3 operator*(const v3& a, const v3& b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PHX_HD v3 operator*(const v3& a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PHX_HD v3 operator*(float s, const v3& a) { return v3(s * a.x, s * a.y, s * a.z); }
// Imath Vec3 semantics
PHX_HD float dot(const v3& a, const v3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PHX_HD v3 cross(const v3& a, const v3& b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PHX_HD float length(const v3& v) {
  float l2 = dot(v, v);
  if (l2 < 2.0f * FLT_MIN) {  // Imath lengthTiny
    float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
    float m = ax; if (m < ay) m = ay; if (m < az) m = az;
    if (m == 0.0f) return 0.0f;
    ax /= m; ay /= m; az /= m;
    return m * sqrtf(ax * ax + ay * ay + az * az);
  }
  return sqrtf(l2);
}
PHX_HD v3 normalized(const v3& v) {
  float l = length(v);
  if (l == 0.0f) return v3(0.0f);
  return v3(v.x / l, v.y / l, v.z / l);
}
PHX_HD v3 normalize_inplace(const v3& v) {  // Vec3::normalize(): zero vectors stay as they are
  float l = length(v);
  if (l != 0.0f) return v3(v.x / l, v.y / l, v.z / l);
  return v;
}
// reference simd:: (fused) variants
PHX_HD float sdot(const v3& a, const v3& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
PHX_HD v3 scross(const v3& a, const v3& b) {
  return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// ---- deterministic transcendental functions (binary64 kernels, one rounding to fp32) -----------
PHX_HD double rint_small(double x) { const double big = 6755399441055744.0; return (x + big) - big; }
PHX_HD double u64_as_double(uint64_t u) { union { uint64_t u; double d; } c; c.u = u; return c.d; }
PHX_HD uint64_t double_as_u64(double d) { union { uint64_t u; double d; } c; c.d = d; return c.u; }

struct sincos_t { double s, c; };
PHX_HD sincos_t sincos_d(double x) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  double k = rint_small(x * two_over_pi);
  double r = fma(-k, pio2_hi, x);
  r = fma(-k, pio2_lo, r);
  double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double ps = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
  double sr = fma(z * r, fma(z, ps, S1), r);
  double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  int q = (int)((long long)k & 3);
  double ss, cc;
  if (q == 0) { ss = sr; cc = cr; }
  else if (q == 1) { ss = cr; cc = -sr; }
  else if (q == 2) { ss = -sr; cc = -cr; }
  else { ss = -cr; cc = sr; }
  return sincos_t{ss, cc};
}
PHX_HD double exp_d(double x) {
  const double inv_ln2 = 1.44269504088896338700e+00;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  double k = rint_small(x * inv_ln2);
  double r = fma(-k, ln2_hi, x);
  r = fma(-k, ln2_lo, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  long long ki = (long long)k;
  if (ki < -1000) return 0.0;
  if (ki > 1000) return u64_as_double(0x7ff0000000000000ull);
  long long k1 = ki / 2, k2 = ki - k1;
  double s1 = u64_as_double((uint64_t)(1023 + k1) << 52);
  double s2 = u64_as_double((uint64_t)(1023 + k2) << 52);
  return p * s1 * s2;
}
PHX_HD double log_d(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  uint64_t u = double_as_u64(x);
  long long e = (long long)((u >> 52) & 0x7ff) - 1023;
  uint64_t mant = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double mval = u64_as_double(mant);
  if (mval > 1.41421356237309514547) { mval = mval * 0.5; e += 1; }
  double f = mval - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double p = 1.0 / 27.0;
  p = fma(p, z, 1.0 / 25.0);
  p = fma(p, z, 1.0 / 23.0);
  p = fma(p, z, 1.0 / 21.0);
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p = fma(p, z, 1.0);
  double lm = 2.0 * s * p;
  double ed = (double)e;
  return fma(ed, ln2_hi, fma(ed, ln2_lo, lm));
}
struct sincosf_t { float s, c; };
PHX_HD_CALL sincosf_t sincosf_(float x) { const sincos_t r = sincos_d((double)x); return sincosf_t{(float)r.s, (float)r.c}; }  // by value: out-pointers would be stack slots
PHX_HD_CALL float expf_(float x) {
  if (x != x) return x;
  if (x > 89.0f) return INFINITY;
  if (x < -104.0f) return 0.0f;
  return (float)exp_d((double)x);
}
PHX_HD_CALL float logf_(float x) {
  if (x != x || x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (isinf(x)) return x;
  return (float)log_d((double)x);
}
PHX_HD_CALL float powf_(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return NAN;
  if (x == 1.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : INFINITY;
  if (x < 0.0f) return NAN;
  if (isinf(x)) return y > 0.0f ? INFINITY : 0.0f;
  if (isinf(y)) return ((x < 1.0f) == (y > 0.0f)) ? 0.0f : INFINITY;
  double t = (double)y * log_d((double)x);
  if (t > 89.0) return INFINITY;
  if (t < -104.0) return 0.0f;
  return (float)exp_d(t);
}

// ---- counter-based sampler: replaces sampler_t's sequential mt19937 (src/sampling.cpp:43-76) ----
PHX_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
PHX_HD uint32_t path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t k = mix32((uint32_t)seed ^ 0x85ebca6bu);
  k = mix32(k + pixel);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k + sample * 0x9e3779b1u);
  return k;
}
PHX_HD float draw_f32(uint32_t key, uint32_t dim) {
  uint32_t x = mix32(key + (dim + 1u) * 0x9e3779b9u);
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
enum { DIM_LIGHT_PICK = 0, DIM_LIGHT_U = 1, DIM_LIGHT_V = 2, DIM_RR = 3, DIM_BSDF_U = 4, DIM_BSDF_V = 5, DIMS_PER_STEP = 8,
       DIM_LENS_U = 6, DIM_LENS_V = 7 /* the two spare dimensions of step 0: the thin lens (camera_ray<true>) */ };
static const uint32_t FILM_JITTER_STREAM = 0xffffffffu;

}  // namespace phx
