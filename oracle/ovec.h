// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// ovec.h: the fp32 vector semantics the reference gets from Imath (third-party, Imath/OpenEXR,
// version per reference README.md:12 — not vendored, absent from this image) and from its own
// AVX2 wrappers (reference src/math/simd/vector.hpp).  Restated from the published Imath
// definitions: Vec3::dot = x*x'+y*y'+z*z' (left to right, no fusion), cross, length =
// sqrt(dot(self)), normalize() divides each component by length() and leaves zero vectors alone,
// operator*(Vec3) is component-wise, Box3 empty = (+max, lowest), center = (max+min)/2.
// The simd:: variants use explicit FMA chains exactly as src/math/simd/vector.hpp:98-109 does.
// Build with -ffp-contract=off so nothing else is fused.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace orc {

struct V3 {
  float x, y, z;
  V3() : x(0), y(0), z(0) {}
  explicit V3(float a) : x(a), y(a), z(a) {}
  V3(float a, float b, float c) : x(a), y(b), z(c) {}
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
  V3 operator+(const V3& o) const { return V3(x + o.x, y + o.y, z + o.z); }
  V3 operator-(const V3& o) const { return V3(x - o.x, y - o.y, z - o.z); }
  V3 operator-() const { return V3(-x, -y, -z); }
  V3 operator*(const V3& o) const { return V3(x * o.x, y * o.y, z * o.z); }
  V3 operator*(float s) const { return V3(x * s, y * s, z * s); }
  V3& operator+=(const V3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  V3& operator*=(const V3& o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
  float dot(const V3& o) const { return x * o.x + y * o.y + z * o.z; }
  V3 cross(const V3& o) const { return V3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  float length2() const { return dot(*this); }
  // Imath Vec3<T>::length(): sqrt(length2) unless length2 < 2*FLT_MIN (lengthTiny rescales)
  float length() const {
    float l2 = length2();
    if (l2 < 2.0f * FLT_MIN) {
      float ax = std::fabs(x), ay = std::fabs(y), az = std::fabs(z);
      float m = ax; if (m < ay) m = ay; if (m < az) m = az;
      if (m == 0.0f) return 0.0f;
      ax /= m; ay /= m; az /= m;
      return m * std::sqrt(ax * ax + ay * ay + az * az);
    }
    return std::sqrt(l2);
  }
  V3& normalize() { float l = length(); if (l != 0.0f) { x /= l; y /= l; z /= l; } return *this; }
  V3 normalized() const { float l = length(); if (l == 0.0f) return V3(0.0f); return V3(x / l, y / l, z / l); }
};
inline V3 operator*(float s, const V3& v) { return V3(s * v.x, s * v.y, s * v.z); }

struct V2 { float x, y; V2() : x(0), y(0) {} V2(float a, float b) : x(a), y(b) {} };

struct Box3 {
  V3 min, max;
  Box3() : min(FLT_MAX), max(-FLT_MAX) {}
  Box3(const V3& a, const V3& b) : min(a), max(b) {}
  void extendBy(const V3& p) {
    if (p.x < min.x) min.x = p.x; if (p.x > max.x) max.x = p.x;
    if (p.y < min.y) min.y = p.y; if (p.y > max.y) max.y = p.y;
    if (p.z < min.z) min.z = p.z; if (p.z > max.z) max.z = p.z;
  }
  void extendBy(const Box3& b) {
    if (b.min.x < min.x) min.x = b.min.x; if (b.max.x > max.x) max.x = b.max.x;
    if (b.min.y < min.y) min.y = b.min.y; if (b.max.y > max.y) max.y = b.max.y;
    if (b.min.z < min.z) min.z = b.min.z; if (b.max.z > max.z) max.z = b.max.z;
  }
  V3 center() const { V3 s = max + min; return V3(s.x / 2, s.y / 2, s.z / 2); }
};

// ---- reference simd:: semantics (src/math/simd/vector.hpp, float8.hpp) on one lane ----------
namespace sv {
// madd(a,b,c) = a*b+c fused (_mm256_fmadd_ps), msub(a,b,c) = a*b-c fused (_mm256_fmsub_ps)
inline float madd(float a, float b, float c) { return std::fmaf(a, b, c); }
inline float msub(float a, float b, float c) { return std::fmaf(a, b, -c); }
// vector3_t::dot, vector.hpp:98-100
inline float dot(const V3& a, const V3& b) { return madd(a.x, b.x, madd(a.y, b.y, a.z * b.z)); }
// vector3_t::cross, vector.hpp:102-109
inline V3 cross(const V3& a, const V3& b) {
  return V3(msub(a.y, b.z, a.z * b.y), msub(a.z, b.x, a.x * b.z), msub(a.x, b.y, a.y * b.x));
}
}  // namespace sv

}  // namespace orc
