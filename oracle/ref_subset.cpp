// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// ref_subset.cpp -> oracle/_ref/libphx_ref_subset.so
//
// The ONLY pieces of the reference that compile here without third-party code are the headers
// that include nothing but the C++ standard library / x86 intrinsics.  They are compiled from
// where they lie under /root/reference/src (never copied) and exported so that the restatement
// can be checked against the reference's own object code for them:
//   fresnel::dielectric          src/math/fresnel.hpp:6-28
//   trig::radians                src/math/trigonometry.hpp:5-9
//   simd::float_t<8> ops         src/math/simd/float8.hpp (select / compares / rcp / min / max)
//   simd::int32_t<8> ops         src/math/simd/int8.hpp (includes only float8.hpp): ==, <=, >=, - and the bit operations are
//                                FLOAT instructions on the reinterpreted integer bits (SURVEY A-20), e.g. the flag test
//                                (flags & HIT) == HIT of spt.hpp:138 compares denormals
//   __bscf                       src/utils/compiler.hpp:6-14
//   parsed_options_t defaults    src/options.hpp:6-43 (spp 16, paths per sample 16, depth 9, output "out.exr"; host_only is left
//                                uninitialised by the constructor and is not exported)
//   config::STREAM_SIZE          src/math/config.hpp:6 (the 1024-slot stream every stage of the hot path is sized by)
// Everything else of the hot path includes Imath / OpenImageIO / OSL headers, which this image
// does not have: per the build rules that part is "unbuildable here" (no stand-in headers).
#include <cstddef>
#include <cstdint>
#include <immintrin.h>

#include "math/fresnel.hpp"
#include "math/trigonometry.hpp"
#include "math/simd/float8.hpp"
#include "math/simd/int8.hpp"
#include "utils/compiler.hpp"
#include "math/config.hpp"
#include "options.hpp"

extern "C" {

void ref_fresnel_dielectric(uint32_t n, const float* cosi, const float* eta, float* out) {
  for (uint32_t i = 0; i < n; ++i) out[i] = fresnel::dielectric(cosi[i], eta[i]);
}
void ref_radians(uint32_t n, const float* a, float* out) {
  for (uint32_t i = 0; i < n; ++i) out[i] = trig::radians(a[i]);
}
// 8-wide helpers: all arrays have 8 floats
void ref_select8(const float* mask_bits, const float* l, const float* r, float* out) {
  simd::float_t<8> m(_mm256_loadu_ps(mask_bits)), a(_mm256_loadu_ps(l)), b(_mm256_loadu_ps(r));
  _mm256_storeu_ps(out, simd::select(m, a, b).v);
}
void ref_cmp8(int op, const float* l, const float* r, float* out_bits) {
  simd::float_t<8> a(_mm256_loadu_ps(l)), b(_mm256_loadu_ps(r));
  simd::float_t<8> m = op == 0 ? (a < b) : op == 1 ? (a <= b) : op == 2 ? (a > b) : (a >= b);
  _mm256_storeu_ps(out_bits, m.v);
}
void ref_minmax8(int is_max, const float* l, const float* r, float* out) {
  simd::float_t<8> a(_mm256_loadu_ps(l)), b(_mm256_loadu_ps(r));
  _mm256_storeu_ps(out, (is_max ? simd::max(a, b) : simd::min(a, b)).v);
}
void ref_rcp8(const float* x, float* out) { _mm256_storeu_ps(out, simd::rcp(simd::float_t<8>(_mm256_loadu_ps(x))).v); }
// simd::int32_t<8>: all arrays have 8 int32
void ref_int8_op(int op, const int32_t* l, const int32_t* r, int32_t* out) {
  const simd::int32_t<8> a = simd::int32_t<8>::loadu(l), b = simd::int32_t<8>::loadu(r);
  simd::int32_t<8> c;
  switch (op) {
    case 0: c = a + b; break;
    case 1: c = a; break;       // operator- is NOT exported: simd::sub(__m256i, __m256i) (int8.hpp:38-40) calls itself — unbounded
                                // recursion, undefined behaviour (this compiler happens to emit an xor); nothing on the hot path uses it
    case 2: c = a == b; break;  // _mm256_cmp_ps(EQ) on the bit patterns
    case 3: c = a <= b; break;
    case 4: c = a >= b; break;
    case 5: c = a & b; break;
    case 6: c = a | b; break;
    default: c = a ^ b; break;
  }
  _mm256_storeu_si256((__m256i*)out, c.v);
}
// the flag test of the hot path, e.g. (hits.flags & HIT) == HIT (kernels/cpu/spt.hpp:138): all-ones lanes where the bit is set
void ref_int8_flag_test(const int32_t* flags, int32_t bit, int32_t* out) {
  const simd::int32_t<8> f = simd::int32_t<8>::loadu(flags), m(bit);
  _mm256_storeu_si256((__m256i*)out, ((f & m) == m).v);
}
void ref_int8_from_float(const float* x, int32_t* out) {  // int32_t<8>(float_t<8>): _mm256_cvtps_epi32, round to nearest even
  _mm256_storeu_si256((__m256i*)out, simd::int32_t<8>(simd::float_t<8>(_mm256_loadu_ps(x))).v);
}
void ref_int8_select(const float* mask_bits, const int32_t* l, const int32_t* r, int32_t* out) {
  const simd::float_t<8> m(_mm256_loadu_ps(mask_bits));
  _mm256_storeu_si256((__m256i*)out, simd::select(m, simd::int32_t<8>::loadu(l), simd::int32_t<8>::loadu(r)).v);
}
// out[0..6] = samples_per_pixel, paths_per_sample, path_depth, single_threaded, progressive, render_normals, verbose of a
// default-constructed parsed_options_t; returns the length of its default output name, copied to `name`
uint32_t ref_options_defaults(uint32_t* out, char* name, uint32_t cap) {
  const parsed_options_t o;
  out[0] = o.samples_per_pixel; out[1] = o.paths_per_sample; out[2] = o.path_depth;
  out[3] = o.single_threaded; out[4] = o.progressive; out[5] = o.render_normals; out[6] = o.verbose;
  uint32_t n = 0;
  for (; n < o.output.size() && n + 1 < cap; ++n) name[n] = o.output[n];
  name[n] = 0;
  return n;
}
uint32_t ref_stream_size() { return config::STREAM_SIZE; }
uint64_t ref_bscf(uint64_t v, uint64_t* rest) { size_t x = v; size_t i = __bscf(x); *rest = x; return i; }

}  // extern "C"
