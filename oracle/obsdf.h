// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// obsdf.h: closure container and lobe models, restated from
//   bsdf_t                       src/bsdf.hpp:8-95, src/bsdf.cpp:19-248
//   params + precompute          src/bsdf/params.hpp:12-110
//   lambert / oren_nayar         src/bsdf/lambert.hpp:8-36, src/bsdf/oren_nayar.hpp:9-69
//   reflection / refraction      src/bsdf/reflection.hpp:8-21, src/bsdf/refraction.hpp:10-46
//   cook_torrance (+refract)     src/bsdf/microfacet.hpp:13-278, ggx_t :306-435
//   sheen                        src/bsdf/sheen.hpp:16-88
//   orthogonal/invertible base   src/math/orthogonal_base.hpp:5-71
//   ts::*, offset                src/math/vector.hpp:10-72
//   hemisphere sampling          src/math/sampling.hpp:11-36
//   fresnel::dielectric          src/math/fresnel.hpp:6-28
// Quirks kept on purpose are listed in SURVEY Appendix D; where the reference reads
// uninitialised memory the oracle DEFINES the result (marked "defined:").
#pragma once
#include "../include/phx_xpu.h"
#include "omath.h"
#include "ovec.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace orc {

static const double kPi = 3.14159265358979323846;   // M_PI
static const double kInvPi = 0.318309886183790671538;  // M_1_PI

struct onb_t {  // orthogonal_base_t(n), orthogonal_base.hpp:11-19
  V3 a, b, c;
  explicit onb_t(const V3& n) {
    a = ((n.x != n.y || n.x != n.z) ? V3(n.z - n.y, n.x - n.z, n.y - n.x) : V3(n.z - n.y, n.x + n.z, -n.y - n.x)).normalized();
    b = n;
    c = a.cross(n).normalized();
  }
  V3 to_world(const V3& v) const { return v.x * a + v.y * b + v.z * c; }
  // invertible_base_t::to_local, orthogonal_base.hpp:45-70: v.x*ia + v.y*ib + v.z*ic with the
  // transposed basis ia=(a.x,b.x,c.x) ...
  V3 to_local(const V3& v) const {
    const V3 ia(a.x, b.x, c.x), ib(a.y, b.y, c.y), ic(a.z, b.z, c.z);
    return v.x * ia + v.y * ib + v.z * ic;
  }
};

namespace ts {  // math/vector.hpp:24-72 — tangent space, y is "up"
inline bool in_same_hemisphere(const V3& a, const V3& b) { return (a.y * b.y) > 0.0f; }
inline float cos2_theta(const V3& v) { return v.y * v.y; }
inline float cos_theta(const V3& v) { return v.y; }
inline float sin2_theta(const V3& v) { return std::max(0.0f, 1.0f - cos2_theta(v)); }
inline float sin_theta(const V3& v) { return std::sqrt(sin2_theta(v)); }
inline float tan_theta(const V3& v) { return sin_theta(v) / cos_theta(v); }
inline float tan2_theta(const V3& v) { return sin2_theta(v) / cos2_theta(v); }
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float cos_phi(const V3& v) { float s = sin_theta(v); return (s == 0) ? 1.0f : clampf(v.x / s, -1.f, 1.f); }
inline float sin_phi(const V3& v) { float s = sin_theta(v); return (s == 0) ? 0.0f : clampf(v.z / s, -1.f, 1.f); }
inline float cos2_phi(const V3& v) { float x = cos_phi(v); return x * x; }
inline float sin2_phi(const V3& v) { float x = sin_phi(v); return x * x; }
}  // namespace ts

inline bool in_same_hemisphere_ws(const V3& a, const V3& b) { return (double)a.dot(b) > 0.0; }  // vector.hpp:10-12

// sample::hemisphere::cosine_weighted, math/sampling.hpp:23-36
inline void cosine_weighted(const V2& s, V3& out, float& pdf) {
  const float r = std::sqrt(s.x);
  const float theta = (float)(2 * kPi * (double)s.y);
  const float x = r * m::cosf_(theta);
  const float y = r * m::sinf_(theta);
  out = V3(x, std::sqrt(std::max(0.0f, 1.0f - s.x)), y);
  pdf = out.y * (float)(1.0 / kPi);  // UNIFORM_DISC_PDF = 1.0f / M_PI stored as float
}

// fresnel::dielectric, math/fresnel.hpp:6-28
inline float fresnel_dielectric(float cosi, float eta) {
  if (eta == 0) return 1;
  if (cosi < 0.0f) eta = 1.0f / eta;
  float c = std::fabs(cosi);
  float g = eta * eta - 1.0f + c * c;
  if (g > 0.0f) {
    g = std::sqrt(g);
    float A = (g - c) / (g + c);
    float B = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
    return 0.5f * A * A * (1 + B * B);
  }
  return 1.0f;
}

// ---- one lobe, parameters after precompute() ----------------------------------------------------
struct lobe_t {
  uint32_t type;    // PHX_LOBE_*
  uint32_t flags;   // PHX_BSDF_*
  V3 weight;
  V3 n;             // lobe_t::n — the shading normal of the hit
  float alpha, a, b;       // oren_nayar_t
  float eta;               // reflect/refract/microfacet
  float xalpha, yalpha;    // microfacet (after roughness_to_alpha + clamp)
  int refract;
  float r;                 // sheen
};

// microfacet_t::roughness_to_alpha + precompute, params.hpp:86-99
inline float roughness_to_alpha(float roughness) {
  roughness = std::max(roughness, (float)1e-5);
  float x = m::logf_(roughness);
  return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

struct bsdf_t {
  lobe_t lobe[PHX_MAX_LOBES];
  uint32_t lobes = 0;
  float sheen_L5 = 0.0f;  // see sheen_lambda
  bool has_sheen_L5 = false;

  // What OSL computes at a hit for the one hit-dependent input of the reference's node shaders — the mix factor of Blender's
  // glass node (plugins/blender/blender/shader.hpp:306-335): fresnel_dielectric_node.osl:16-20 on the shader globals of
  // material_t::evaluate (material.cpp:425-436: I = hits.wi, N = n, backfacing = N.I < 0) with the OSL helper
  // src/shaders/fresnel.h:1-19 (fp32, like every OSL float).  OSL itself is third-party and absent: "parity unpinned".
  static float osl_fresnel_dielectric(float cosi, float eta) {
    const float c = std::fabs(cosi);
    float g = eta * eta - 1.0f + c * c;
    if (g > 0.0f) {
      g = std::sqrt(g);
      const float A = (g - c) / (g + c);
      const float B = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
      return 0.5f * A * A * (1.0f + B * B);
    }
    return 1.0f;
  }
  static float fresnel_mix_factor(float ior, const V3& n, const V3& view) {
    const float f = std::max(1.0e-5f, ior);
    const bool backfacing = n.dot(view) < 0.0f;
    const float eta = backfacing ? 1.0f / f : f;
    return osl_fresnel_dielectric(view.dot(n), eta);
  }

  // add_lobe (bsdf.hpp:54-82) for every closure that material_t::eval_closure (material.cpp:218-305) meets at this hit:
  // `view` = hits.wi.  A closure under a Fresnel-driven mix gets the weight (pre * term) * weight, the order eval_closure
  // multiplies down the tree, and is absent when that is all zero (OSL: closure * 0 is the null closure).
  void from_material(const phx_material& mat, const V3& n, const V3& view) {
    lobes = 0;
    for (uint32_t i = 0; i < mat.num_lobes && i < PHX_MAX_LOBES; ++i) {
      const phx_lobe& s = mat.lobes[i];
      lobe_t& l = lobe[lobes];
      l = lobe_t();
      l.type = s.type; l.weight = V3(s.weight[0], s.weight[1], s.weight[2]); l.n = n;
      if (s.fac_mode != PHX_FAC_NONE) {
        const float fac = fresnel_mix_factor(s.fac_ior, n, view);
        const float term = s.fac_mode == PHX_FAC_MIX_B ? fac : 1.0f - fac;
        l.weight = V3((s.pre_weight[0] * term) * s.weight[0], (s.pre_weight[1] * term) * s.weight[1], (s.pre_weight[2] * term) * s.weight[2]);
        if (l.weight.x == 0.0f && l.weight.y == 0.0f && l.weight.z == 0.0f) continue;
      }
      switch (s.type) {
        case PHX_LOBE_DIFFUSE: l.flags = PHX_BSDF_REFLECT | PHX_BSDF_DIFFUSE; break;
        case PHX_LOBE_OREN_NAYAR: {
          l.flags = PHX_BSDF_REFLECT | PHX_BSDF_DIFFUSE;
          l.alpha = s.alpha;
          const float sg = (float)((double)s.alpha * (kPi / (double)180.0f));  // trig::radians: a * (M_PI/180.0f) in double
          const float s2 = sg * sg;
          l.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
          l.b = 0.45f * s2 / (s2 + 0.09f);
          break;
        }
        case PHX_LOBE_REFLECTION: l.flags = PHX_BSDF_REFLECT | PHX_BSDF_SPECULAR; l.eta = s.eta; break;
        case PHX_LOBE_REFRACTION: l.flags = PHX_BSDF_TRANSMIT | PHX_BSDF_SPECULAR; l.eta = s.eta; break;
        case PHX_LOBE_MICROFACET:
          l.flags = s.refract ? PHX_BSDF_TRANSMIT : PHX_BSDF_REFLECT;  // bsdf.hpp:70-72
          l.eta = s.eta; l.refract = (int)s.refract;
          l.xalpha = std::min(1.0f, std::max(0.0001f, roughness_to_alpha(s.xalpha)));
          l.yalpha = std::min(1.0f, std::max(0.0001f, roughness_to_alpha(s.yalpha)));
          break;
        case PHX_LOBE_SHEEN: l.flags = PHX_BSDF_REFLECT | PHX_BSDF_GLOSSY; l.r = s.r; break;
        case PHX_LOBE_TRANSPARENT: l.flags = PHX_BSDF_TRANSMIT; break;  // material.cpp:98-103
        default: continue;  // emission/background are not lobes (material.cpp:240-245)
      }
      ++lobes;
    }
  }
  bool is_reflective(uint32_t i) const { return (lobe[i].flags & PHX_BSDF_REFLECT) == PHX_BSDF_REFLECT; }
  bool is_transmissive(uint32_t i) const { return (lobe[i].flags & PHX_BSDF_TRANSMIT) == PHX_BSDF_TRANSMIT; }

  V3 f(const V3& wi, const V3& wo) const;
  V3 sample(const V2& s, const V3& wi, V3& wo, float& pdf, uint32_t& sample_flags) const;
  V3 eval(const lobe_t& l, const V3& wi, const V3& wo, float& pdf) const;
};

// ---- GGX (microfacet.hpp:306-435) ---------------------------------------------------------------
struct ggx_t {
  static float D(const lobe_t& p, const V3& v) {
    const float tan2 = ts::tan2_theta(v);
    if (std::isinf(tan2)) return 0.0f;
    const float ax = p.xalpha, ay = p.yalpha;
    const float cos2 = ts::cos2_theta(v);
    const float cos4 = cos2 * cos2;
    const float e = (ts::cos2_phi(v) / (ax * ax) + ts::sin2_phi(v) / (ay * ay)) * tan2;
    return (float)(1.0f / (kPi * (double)ax * (double)ay * (double)cos4 * (double)(1 + e) * (double)(1 + e)));
  }
  static float Lambda(const lobe_t& p, const V3& v) {
    const float att = std::fabs(ts::tan_theta(v));
    if (std::isinf(att)) return 0.0f;
    const float ax = p.xalpha, ay = p.yalpha;
    const float alpha = std::sqrt(ts::cos2_phi(v) * ax * ay + ts::sin2_phi(v) * ax * ay);
    const float a2t2 = (alpha * att) * (alpha * att);
    return (-1.0f + std::sqrt(1.0f + a2t2)) * 0.5f;
  }
  static void sample_slope(float cos_theta, float& slope_x, float& slope_y, const V2& uv) {
    float u = uv.x, v = uv.y;
    if ((double)cos_theta > .9999) {
      float r = std::sqrt(u / (1 - u));
      float phi = (float)(6.28318530718 * (double)v);
      slope_x = r * m::cosf_(phi);
      slope_y = r * m::sinf_(phi);
      return;
    }
    const float sin_theta = std::sqrt(std::max(0.0f, 1.0f - (cos_theta * cos_theta)));
    const float tan_theta = sin_theta / cos_theta;
    const float a = 1.0f / tan_theta;
    const float g1 = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (a * a)));
    const float A = 2.0f * u / g1 - 1.0f;
    float tmp = 1.0f / (A * A - 1.0f);
    if ((double)tmp > 1e10) tmp = (float)1e10;
    const float B = tan_theta;
    const float Dv = std::sqrt(std::max((float)(B * B * tmp * tmp - (A * A - B * B) * tmp), 0.0f));
    const float slope_x1 = B * tmp - Dv;
    const float slope_x2 = B * tmp + Dv;
    slope_x = (A < 0.0f || slope_x2 > 1.0f / tan_theta) ? slope_x1 : slope_x2;
    float S;
    if (v > 0.5f) { S = 1.0f; v = 2.0f * (v - 0.5f); } else { S = -1.0f; v = 2.0f * (0.5f - v); }
    const float z = (v * (v * (v * 0.27385f - 0.73369f) + 0.46341f)) / (v * (v * (v * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
    slope_y = S * z * std::sqrt(1.0f + slope_x * slope_x);
  }
  static float G1(const lobe_t& p, const V3& v) { return 1.0f / (1.0f + Lambda(p, v)); }
  static V3 sample(const lobe_t& p, const V3& wi, float& pdf, const V2& uv) {
    const float ax = p.xalpha, ay = p.yalpha;
    V3 stretched(ax * wi.x, wi.y, ay * wi.z);
    stretched.normalize();
    float slope_x, slope_y;
    sample_slope(ts::cos_theta(stretched), slope_x, slope_y, uv);
    const float tmp = ts::cos_phi(stretched) * slope_x - ts::sin_phi(stretched) * slope_y;
    slope_y = ts::sin_phi(stretched) * slope_x + ts::cos_phi(stretched) * slope_y;
    slope_x = tmp;
    slope_x = slope_x * ax;
    slope_y = slope_y * ay;
    V3 wh(-slope_x, 1.0f, -slope_y);
    wh.normalize();
    pdf = (D(p, wh) * G1(p, wi) * std::fabs(wi.dot(wh)) / std::fabs(ts::cos_theta(wi)));
    return wh;
  }
};

// ---- sheen distribution (sheen.hpp:16-64) -------------------------------------------------------
inline float sheen_L(float x, float r) {
  static const float p0[] = {25.3245f, 3.32435f, 0.16801f, -1.27393f, -4.85967f};
  static const float p1[] = {21.5473f, 3.82987f, 0.19823f, -1.97760f, -4.32054f};
  auto interp = [](float a, float b, float t) -> float { return t * a + (1.0f - t) * b; };
  const float t = (1.0f - r) * (1.0f - r);
  const float a = interp(p0[0], p1[0], t), b = interp(p0[1], p1[1], t), c = interp(p0[2], p1[2], t),
              d = interp(p0[3], p1[3], t), e = interp(p0[4], p1[4], t);
  const float xc = m::powf_(x, c);
  return a / (1 + b * xc) + d * x + e;
}
struct sheen_dist_t {
  float L5;  // defined: `static const auto L5 = L(0.5f, params.r)` (sheen.hpp:57) is initialised by the
             // first sheen lobe ever evaluated in the process; the oracle fixes it to the first sheen
             // lobe of the material table.
  float D(const lobe_t& p, const V3& v) const {
    const float st = ts::sin_theta(v);
    const float oor = 1.0f / p.r;
    return (float)((double)((2.0f + oor) * m::powf_(st, oor)) / (2.0f * kPi));
  }
  float Lambda(const lobe_t& p, const V3& v) const {
    const float ct = ts::cos_theta(v);
    const float l = (ct < 0.5f) ? sheen_L(ct, p.r) : 2.0f * L5 - sheen_L(1.0f - ct, p.r);
    return m::expf_(l);
  }
};

// ---- Cook-Torrance reflect (microfacet.hpp:175-277), templated on the distribution ---------------
template <typename Dist>
inline V3 ct_f(const lobe_t& p, const V3& wi, const V3& wo, const Dist& dist) {
  onb_t base(p.n);
  const V3 li = base.to_local(wi), lo = base.to_local(wo);
  if (!ts::in_same_hemisphere(li, lo)) return V3(0.0f);
  V3 wh = li + lo;
  const float cos_ti = std::fabs(ts::cos_theta(li)), cos_to = std::fabs(ts::cos_theta(lo));
  if (cos_ti == 0 || cos_to == 0) return V3(0.0f);
  if (wh.x == 0 || wh.y == 0 || wh.z == 0) return V3(0.0f);
  wh.normalize();
  const float d = dist.D(p, wh);
  const float g = 1.0f / (1.0f + dist.Lambda(p, li) + dist.Lambda(p, lo));
  // wh.dot({0,1.0,0}) < 0 ? -wh : wh ; Fresnel eta hard-coded 0.5 (microfacet.hpp:209)
  const float whdoty = wh.x * 0.0f + wh.y * 1.0f + wh.z * 0.0f;
  const V3 whf = whdoty < 0.0f ? -wh : wh;
  const float f = fresnel_dielectric(lo.dot(whf), 0.5f);
  const float c = d * g * f * (1.0f / (4.0f * cos_ti * cos_to));
  return V3(c);
}
struct ggx_adapter_t {
  float D(const lobe_t& p, const V3& v) const { return ggx_t::D(p, v); }
  float Lambda(const lobe_t& p, const V3& v) const { return ggx_t::Lambda(p, v); }
};
inline float ct_pdf(const lobe_t& p, const V3& wi, const V3& wo) {  // microfacet.hpp:216-235
  onb_t base(p.n);
  const V3 li = base.to_local(wi), lo = base.to_local(wo);
  if (!ts::in_same_hemisphere(li, lo)) return 0;
  V3 wh = (li + lo).normalize();
  // G1 is evaluated on the WORLD-space wi (quirk, :234)
  return (ggx_t::D(p, wh) * ggx_t::G1(p, wi) * std::fabs(li.dot(wh)) / std::fabs(ts::cos_theta(li))) / (4.0f * li.dot(wh));
}
inline V3 ct_sample(const lobe_t& p, const V3& wi, V3& wo, const V2& s, float& opdf, bool& pdf_set) {  // :237-277
  onb_t base(p.n);
  const V3 li = base.to_local(wi);
  if (li.y == 0.0f) return V3(0.0f);
  float dpdf;
  const V3 wh = ggx_t::sample(p, li, dpdf, s);
  if (li.dot(wh) < 0.0f) return V3(0.0f);
  const V3 lo = -li + (2.0f * li.dot(wh)) * wh;
  if (!ts::in_same_hemisphere(li, lo)) return V3(0.0f);
  opdf = dpdf / (4.0f * li.dot(wh));
  pdf_set = true;
  wo = base.to_world(lo);
  return ct_f(p, wi, wo, ggx_adapter_t());
}

// ---- Cook-Torrance refract (microfacet.hpp:36-172) ----------------------------------------------
inline V3 ctr_f(const lobe_t& p, const V3& wi, const V3& wo) {
  onb_t base(p.n);
  const V3 li = base.to_local(wi), lo = base.to_local(wo);
  if (ts::in_same_hemisphere(li, lo)) return V3(0.0f);
  const float eta = li.y > 0.0f ? p.eta : 1.0f / p.eta;
  const float cos_ti = ts::cos_theta(li), cos_to = ts::cos_theta(lo);
  if (cos_ti == 0.0f || cos_to == 0.0f) return V3(0.0f);
  V3 wh = (li + lo * eta).normalize();
  if (wh.y < 0) wh = -wh;
  if (lo.dot(wh) * li.dot(wh) > 0) return V3(0.0f);
  const float f = fresnel_dielectric(lo.dot(wh), eta);
  const float sqrt_denom = li.dot(wh) + eta * lo.dot(wh);
  const float factor = 1.0f / eta;
  const float d = ggx_t::D(p, wh);
  const float g = 1.0f / (1.0f + ggx_t::Lambda(p, li) + ggx_t::Lambda(p, lo));
  const float c = (1.0f - f) * std::fabs(d * g * eta * eta * std::fabs(lo.dot(wh)) * std::fabs(li.dot(wh)) * factor * factor /
                                         (cos_ti * cos_to * sqrt_denom * sqrt_denom));
  return V3(c);
}
inline float ctr_pdf(const lobe_t& p, const V3& wi, const V3& wo) {  // :94-116
  onb_t base(p.n);
  const V3 li = base.to_local(wi), lo = base.to_local(wo);
  const float eta = li.y > 0.0f ? p.eta : 1.0f / p.eta;
  if (in_same_hemisphere_ws(wo, wi)) return 0;
  V3 wh = (li + lo * eta).normalize();
  const float sqrt_denom = li.dot(wh) + eta * lo.dot(wh);
  const float dwh_dwi = std::fabs(eta * eta * lo.dot(wh)) / sqrt_denom * sqrt_denom;  // precedence quirk :113
  return (ggx_t::D(p, wh) * ts::cos_theta(wh)) * dwh_dwi;
}
inline V3 ctr_sample(const lobe_t& p, const V3& wi, V3& wo, const V2& s, float& pdf, bool& pdf_set) {  // :118-171
  if (p.eta == 1.0f) { wo = -wi; pdf = 1.0f; pdf_set = true; return V3(1.0f); }
  onb_t base(p.n);
  const V3 li = base.to_local(wi);
  if (li.y == 0.0f) return V3(0.0f);
  float dpdf;
  const V3 wh = ggx_t::sample(p, li, dpdf, s);
  if (wh.dot(li) < 0.0f) return V3(0.0f);
  const float eta = li.y > 0.0f ? 1.0f / p.eta : p.eta;
  const float cos_ti = wh.dot(li);
  const float sin2_ti = std::max(0.0f, 1.0f - cos_ti * cos_ti);
  const float sin2_tt = eta * eta * sin2_ti;
  if (sin2_tt >= 1.0f) return V3(0.0f);
  const float cos_tt = std::sqrt(1.0f - sin2_tt);
  const V3 lo = eta * -li + (eta * cos_ti - cos_tt) * wh;
  const float sqrt_denom = li.dot(wh) + eta * lo.dot(wh);
  const float dwh_dwi = std::fabs((eta * eta * lo.dot(wh)) / (sqrt_denom * sqrt_denom));
  pdf = dpdf * dwh_dwi;
  pdf_set = true;
  wo = base.to_world(lo);
  return ctr_f(p, wi, wo);
}

// ---- oren-nayar f (oren_nayar.hpp:9-47) ---------------------------------------------------------
inline V3 oren_nayar_f(const lobe_t& p, const V3& wi, const V3& wo) {
  onb_t base(p.n);
  const V3 li = base.to_local(wi), lo = base.to_local(wo);
  const float cos_theta_i = std::fabs(ts::cos_theta(li)), cos_theta_o = std::fabs(ts::cos_theta(lo));
  const float sin_theta_i = ts::sin_theta(li), sin_theta_o = ts::sin_theta(lo);
  float max_cos = 0.0f;
  if (sin_theta_i > 0.0001f && sin_theta_o > 0.0001f) {
    const float sin_phi_i = ts::sin_phi(li), cos_phi_i = ts::cos_phi(li);
    const float sin_phi_o = ts::sin_phi(lo), cos_phi_o = ts::cos_phi(lo);
    const float dcos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
    max_cos = std::max(0.0f, dcos);
  }
  float sin_alpha, tan_beta;
  if (cos_theta_i > cos_theta_o) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / cos_theta_i; }
  else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / cos_theta_o; }
  const float result = (p.a + p.b * max_cos * sin_alpha * tan_beta);
  return V3((float)((double)result * kInvPi));
}

// eval(), bsdf.cpp:29-107
inline V3 bsdf_t::eval(const lobe_t& l, const V3& wi, const V3& wo, float& pdf) const {
  V3 result(0.0f);
  switch (l.type) {
    case PHX_LOBE_DIFFUSE:
      pdf = (float)((double)l.n.dot(wi) * kInvPi);
      result = V3((float)kInvPi);
      break;
    case PHX_LOBE_OREN_NAYAR:
      pdf = (float)((double)l.n.dot(wi) * kInvPi);
      result = oren_nayar_f(l, wi, wo);
      break;
    case PHX_LOBE_MICROFACET:
      if (l.refract) { pdf = ctr_pdf(l, wi, wo); result = ctr_f(l, wi, wo); }
      else { pdf = ct_pdf(l, wi, wo); result = ct_f(l, wi, wo, ggx_adapter_t()); }
      break;
    case PHX_LOBE_SHEEN:
      pdf = (float)((double)l.n.dot(wi) * kInvPi);
      result = ct_f(l, wi, wo, sheen_dist_t{sheen_L5});
      break;
    case PHX_LOBE_REFLECTION: case PHX_LOBE_REFRACTION: case PHX_LOBE_TRANSPARENT: pdf = 0.0f; break;
    default: break;
  }
  return result;
}

// bsdf_t::f, bsdf.cpp:113-131
inline V3 bsdf_t::f(const V3& wi, const V3& wo) const {
  V3 out(0.0f);
  float ignored = 0.0f;
  for (uint32_t i = 0; i < lobes; ++i) {
    const V3 e = eval(lobe[i], wi, wo, ignored);
    const float atl = lobe[i].n.dot(wi);
    const bool reflect = atl * lobe[i].n.dot(wo) > 0.0f;
    if ((reflect && is_reflective(i)) || (!reflect && is_transmissive(i))) out += e * lobe[i].weight * atl;
  }
  return out;
}

// bsdf_t::sample, bsdf.cpp:133-248.  defined: a bsdf with 0 lobes, and every early return of a lobe
// sampler that leaves `pdf` unset, terminate the path (pdf = 0, black) — SURVEY A-10, Appendix D.
inline V3 bsdf_t::sample(const V2& s, const V3& wi, V3& wo, float& pdf, uint32_t& sample_flags) const {
  pdf = 0.0f; sample_flags = 0;
  if (lobes == 0) return V3(0.0f);
  const uint32_t index = std::min((uint32_t)std::floor(s.x * lobes), (lobes - 1));
  const float one_minus_epsilon = 1.0f - std::numeric_limits<float>::epsilon();
  const float u = std::min(s.x * lobes - index, one_minus_epsilon);
  const V2 remapped(u, s.y);
  V3 result(0.0f);
  const lobe_t& p = lobe[index];
  bool pdf_set = false;
  switch (p.type) {
    case PHX_LOBE_DIFFUSE: {
      onb_t base(p.n);
      cosine_weighted(remapped, wo, pdf); pdf_set = true;
      wo = base.to_world(wo);
      result = V3((float)kInvPi);
      break;
    }
    case PHX_LOBE_OREN_NAYAR: {
      onb_t base(p.n);
      cosine_weighted(remapped, wo, pdf); pdf_set = true;
      wo = base.to_world(wo);
      result = oren_nayar_f(p, wi, wo);
      break;
    }
    case PHX_LOBE_MICROFACET:
      result = p.refract ? ctr_sample(p, wi, wo, remapped, pdf, pdf_set) : ct_sample(p, wi, wo, remapped, pdf, pdf_set);
      break;
    case PHX_LOBE_SHEEN: {
      onb_t base(p.n);
      cosine_weighted(remapped, wo, pdf); pdf_set = true;
      wo = base.to_world(wo);
      result = ct_f(p, wi, wo, sheen_dist_t{sheen_L5});
      break;
    }
    case PHX_LOBE_REFLECTION: {  // reflection.hpp:8-21
      const float cos_theta = p.n.dot(wi);
      pdf = 1.0f; pdf_set = true;
      wo = -wi + (2.0f * cos_theta) * p.n;
      result = V3(1.0f);
      break;
    }
    case PHX_LOBE_REFRACTION: {  // refraction.hpp:10-46
      pdf = 1.0f; pdf_set = true;
      float cos_theta = p.n.dot(wi);
      const float sin_theta = std::max(0.0f, 1.0f - cos_theta * cos_theta);
      V3 n; float eta = p.eta;
      if (cos_theta > 0) { n = p.n; eta = 1.0f / eta; } else { n = -p.n; cos_theta = -cos_theta; }
      const float arg = 1.0f - (eta * eta * sin_theta);
      if (arg >= 0.0f) {
        const float dnp = std::sqrt(arg);
        const float nk = eta * cos_theta - dnp;
        wo = -wi * eta + n * nk;
        result = V3(1.0f);
      } else {
        result = V3(0.0f);  // defined: TIR returns a default-constructed Color3f (:45) -> black
      }
      break;
    }
    case PHX_LOBE_TRANSPARENT:  // bsdf.cpp:209-214
      wo = -wi; pdf = 1.0f; pdf_set = true; result = V3(1.0f);
      break;
    default: break;
  }
  if (!pdf_set) pdf = 0.0f;
  if (pdf == 0.0f) return V3(0.0f);
  result *= p.weight;
  int matched_lobes = 1;
  for (uint32_t i = 0; i < lobes; ++i) {
    if (i != index && ((lobe[index].flags & lobe[i].flags) == lobe[i].flags)) {
      const bool reflect = lobe[i].n.dot(wi) * lobe[i].n.dot(wo) > 0.0f;
      if ((reflect && is_reflective(i)) || (!reflect && is_transmissive(i))) {
        float lobe_pdf = 0.0f;
        result += eval(lobe[i], wi, wo, lobe_pdf) * lobe[i].weight;
        pdf += lobe_pdf;
        ++matched_lobes;
      }
    }
  }
  pdf /= matched_lobes;
  sample_flags = lobe[index].flags;
  return result;
}

}  // namespace orc
