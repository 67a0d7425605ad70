// bsdf.h — closure evaluation on the device: the per-hit work of bsdf_t::f (reference
// src/bsdf.cpp:113-131) and bsdf_t::sample (:133-248) over a baked closure recipe (the output
// contract of material_t::evaluate, src/material.cpp:218-305, with constant inputs).
//
// Lobe models: lambert (src/bsdf/lambert.hpp:8-36), oren-nayar (oren_nayar.hpp:9-69), reflection
// (reflection.hpp:8-21), refraction (refraction.hpp:10-46), Cook-Torrance GGX reflect/refract
// (microfacet.hpp:36-278, ggx_t :306-435), sheen (sheen.hpp:16-88), transparent (bsdf.cpp:209-214).
// The reference's quirks are kept: pdf averaged over matched lobes, f summed;
// Fresnel eta hard-wired to 0.5 in the reflective microfacet f; G1 evaluated on the world-space wi
// in its pdf; Lambda's alpha = sqrt(cos^2 phi ax ay + sin^2 phi ax ay); the `/ d * d` precedence
// slip in the refractive pdf.  Reads of uninitialised memory in the reference (0-lobe bsdf, lobe
// samplers that return before setting pdf, TIR colour) are defined as "terminate the path".
//
// All lobes of a recipe share the hit's shading normal, so the tangent frame
// (orthogonal_base_t, src/math/orthogonal_base.hpp:11-19) is built once per hit.
#pragma once
#include "phx_math.h"

namespace phx {

enum { L_DIFFUSE = 1, L_OREN_NAYAR = 2, L_REFLECTION = 4, L_REFRACTION = 8, L_MICROFACET = 16, L_SHEEN = 32, L_TRANSPARENT = 128 };
enum { B_DIFFUSE = 1, B_GLOSSY = 2, B_SPECULAR = 4, B_REFLECT = 8, B_TRANSMIT = 16 };

struct DevLobe {       // a lobe after add_lobe()/precompute() (src/bsdf.hpp:54-82, params.hpp)
  uint32_t type, flags;
  float wx, wy, wz;    // weight
  float a, b;          // oren-nayar A, B
  float eta;
  float xalpha, yalpha;
  uint32_t refract;
  float r;             // sheen roughness
  uint32_t fac_mode;   // PHX_FAC_*: per-hit Fresnel mix factor on the weight (material_at_hit)
  float fac_ior;
  float px, py, pz;    // constant weights above the factor in the closure tree
};
struct DevMaterial {
  uint32_t num_lobes;
  uint32_t is_emitter;
  float ex, ey, ez;    // hits.e
  float sheen_L5;      // L(0.5, r) of the first sheen lobe of the material table (sheen.hpp:57 static)
  uint32_t per_hit;    // some lobe's weight depends on the hit (fac_mode != 0)
  uint32_t pad;
  DevLobe lobes[8];
};

#if defined(__HIPCC__)
typedef const __attribute__((address_space(4))) DevMaterial ConstMat;  // the material table seen through the constant address space
#endif

// The lobe loops evaluate one of seven models per iteration, chosen by the lobe's type.  Left alone, the compiler hoists every
// loop-invariant subexpression of EVERY model (the local-frame directions, their sines, cosines, tangents, ...) in front of the loop
// and keeps them all alive across it: bsdf_f needed 129 VGPRs although its most expensive model needs 36, and the hoisted arithmetic ran
// for models the material does not have.  Passing the directions through an empty asm inside the loop makes them "new" values per
// iteration: nothing is hoisted, no instruction is emitted.
#if defined(__HIP_DEVICE_COMPILE__)
#define PHX_PIN_V3(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z))
#else
#define PHX_PIN_V3(v) ((void)0)
#endif

static const double kPiD = 3.14159265358979323846;
static const double kInvPiD = 0.318309886183790671538;

struct Frame {  // orthogonal_base_t(n) + invertible_base_t::to_local
  v3 a, b, c;
  PHX_HD explicit Frame(const v3& n) {
    a = normalized((n.x != n.y || n.x != n.z) ? v3(n.z - n.y, n.x - n.z, n.y - n.x) : v3(n.z - n.y, n.x + n.z, -n.y - n.x));
    b = n;
    c = normalized(cross(a, n));
  }
  PHX_HD v3 to_world(const v3& v) const { return v.x * a + v.y * b + v.z * c; }
  PHX_HD v3 to_local(const v3& v) const { return v.x * v3(a.x, b.x, c.x) + v.y * v3(a.y, b.y, c.y) + v.z * v3(a.z, b.z, c.z); }
};

namespace ts {  // src/math/vector.hpp:24-72, y is up
PHX_HD bool same_hemi(const v3& a, const v3& b) { return (a.y * b.y) > 0.0f; }
PHX_HD float cos2_theta(const v3& v) { return v.y * v.y; }
PHX_HD float sin2_theta(const v3& v) { return fmaxf(0.0f, 1.0f - v.y * v.y); }
PHX_HD float sin_theta(const v3& v) { return sqrtf(sin2_theta(v)); }
PHX_HD float tan_theta(const v3& v) { return sin_theta(v) / v.y; }
PHX_HD float tan2_theta(const v3& v) { return sin2_theta(v) / cos2_theta(v); }
PHX_HD float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
PHX_HD float cos_phi(const v3& v) { float s = sin_theta(v); return (s == 0.0f) ? 1.0f : clampf(v.x / s, -1.f, 1.f); }
PHX_HD float sin_phi(const v3& v) { float s = sin_theta(v); return (s == 0.0f) ? 0.0f : clampf(v.z / s, -1.f, 1.f); }
}  // namespace ts

PHX_HD void cosine_weighted(float u1, float u2, v3& out, float& pdf) {  // math/sampling.hpp:23-36
  const float r = sqrtf(u1);
  const float theta = (float)(2 * kPiD * (double)u2);
  const sincosf_t sc = sincosf_(theta); const float s = sc.s, c = sc.c;
  out = v3(r * c, sqrtf(fmaxf(0.0f, 1.0f - u1)), r * s);
  pdf = out.y * (float)(1.0 / kPiD);
}

PHX_HD float fresnel_dielectric(float cosi, float eta) {  // math/fresnel.hpp:6-28
  if (eta == 0.0f) return 1.0f;
  if (cosi < 0.0f) eta = 1.0f / eta;
  const float c = fabsf(cosi);
  float g = eta * eta - 1.0f + c * c;
  if (g > 0.0f) {
    g = sqrtf(g);
    const float A = (g - c) / (g + c);
    const float B = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
    return 0.5f * A * A * (1 + B * B);
  }
  return 1.0f;
}

// ---- per-hit closure weights -----------------------------------------------------------------------
// What OSL would compute at this hit for the one hit-dependent input of the reference's node shaders: the mix factor of
// Blender's glass node (plugins/blender/blender/shader.hpp:306-335).  fresnel_dielectric_node.osl:16-20 with the shader
// globals of material_t::evaluate (src/material.cpp:425-436: I = hits.wi, N = n, backfacing = N.I < 0) and the OSL helper
// src/shaders/fresnel.h:1-19 (NOT math/fresnel.hpp: no eta == 0 case, no inversion for cosi < 0); OSL floats are fp32.
PHX_HD float osl_fresnel_dielectric(float cosi, float eta) {
  const float c = fabsf(cosi);
  float g = eta * eta - 1.0f + c * c;
  if (g > 0.0f) {
    g = sqrtf(g);
    const float A = (g - c) / (g + c);
    const float B = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
    return 0.5f * A * A * (1.0f + B * B);
  }
  return 1.0f;
}
PHX_HD float fresnel_mix_factor(float ior, const v3& n, const v3& view /* hits.wi */) {
  const float f = fmaxf(1.0e-5f, ior);
  const bool backfacing = dot(n, view) < 0.0f;
  const float eta = backfacing ? 1.0f / f : f;
  return osl_fresnel_dielectric(dot(view, n), eta);
}
// The closure list of a material AT THIS HIT is its baked table with the hit-dependent weights resolved — (pre * term) * weight, the
// order eval_closure multiplies down the tree — and closures whose weight is all zero dropped (OSL's null closure), so the number
// of kept lobes is what bsdf_t::lobes would be.  Nothing is copied: bsdf_f / bsdf_sample resolve a lobe's weight where they use it
// (a per-hit copy of the 576-byte material lived in scratch memory: 700 B per lane, and every material read became a flat load).
// PERHIT = false compiles the resolution out (scenes without such materials).  `view` = hits.wi.
template <bool PERHIT, typename LobeT>
PHX_HD bool lobe_weight_at_hit(const LobeT& l, const v3& n, const v3& view, v3& w) {
  w = v3(l.wx, l.wy, l.wz);
  if (PERHIT && l.fac_mode != 0u) {
    const float fac = fresnel_mix_factor(l.fac_ior, n, view);
    const float term = l.fac_mode == 1u ? fac : 1.0f - fac;
    w = v3((l.px * term) * l.wx, (l.py * term) * l.wy, (l.pz * term) * l.wz);
    if (w.x == 0.0f && w.y == 0.0f && w.z == 0.0f) return false;
  }
  return true;
}

// ---- GGX ---------------------------------------------------------------------------------------
template <typename LobeT>
PHX_HD float ggx_D(const LobeT& p, const v3& v) {
  const float tan2 = ts::tan2_theta(v);
  if (isinf(tan2)) return 0.0f;
  const float ax = p.xalpha, ay = p.yalpha;
  const float cos2 = ts::cos2_theta(v);
  const float cos4 = cos2 * cos2;
  const float cp = ts::cos_phi(v), sp = ts::sin_phi(v);
  const float e = ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay)) * tan2;
  return (float)(1.0f / (kPiD * (double)ax * (double)ay * (double)cos4 * (double)(1 + e) * (double)(1 + e)));
}
template <typename LobeT>
PHX_HD float ggx_Lambda(const LobeT& p, const v3& v) {
  const float att = fabsf(ts::tan_theta(v));
  if (isinf(att)) return 0.0f;
  const float ax = p.xalpha, ay = p.yalpha;
  const float cp = ts::cos_phi(v), sp = ts::sin_phi(v);
  const float alpha = sqrtf((cp * cp) * ax * ay + (sp * sp) * ax * ay);
  const float a2t2 = (alpha * att) * (alpha * att);
  return (-1.0f + sqrtf(1.0f + a2t2)) * 0.5f;
}
template <typename LobeT>
PHX_HD float ggx_G1(const LobeT& p, const v3& v) { return 1.0f / (1.0f + ggx_Lambda(p, v)); }
PHX_HD void ggx_sample_slope(float cos_theta, float& slope_x, float& slope_y, float u, float v) {
  if ((double)cos_theta > .9999) {
    const float r = sqrtf(u / (1 - u));
    const float phi = (float)(6.28318530718 * (double)v);
    const sincosf_t sc = sincosf_(phi); const float s = sc.s, c = sc.c;
    slope_x = r * c; slope_y = r * s;
    return;
  }
  const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - (cos_theta * cos_theta)));
  const float tan_theta = sin_theta / cos_theta;
  const float a = 1.0f / tan_theta;
  const float g1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
  const float A = 2.0f * u / g1 - 1.0f;
  float tmp = 1.0f / (A * A - 1.0f);
  if ((double)tmp > 1e10) tmp = (float)1e10;
  const float B = tan_theta;
  const float Dv = sqrtf(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
  const float slope_x1 = B * tmp - Dv;
  const float slope_x2 = B * tmp + Dv;
  slope_x = (A < 0.0f || slope_x2 > 1.0f / tan_theta) ? slope_x1 : slope_x2;
  float S;
  if (v > 0.5f) { S = 1.0f; v = 2.0f * (v - 0.5f); } else { S = -1.0f; v = 2.0f * (0.5f - v); }
  const float z = (v * (v * (v * 0.27385f - 0.73369f) + 0.46341f)) / (v * (v * (v * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
  slope_y = S * z * sqrtf(1.0f + slope_x * slope_x);
}
template <typename LobeT>
PHX_HD v3 ggx_sample(const LobeT& p, const v3& wi, float& pdf, float u, float v) {
  const float ax = p.xalpha, ay = p.yalpha;
  const v3 stretched = normalize_inplace(v3(ax * wi.x, wi.y, ay * wi.z));
  float slope_x, slope_y;
  ggx_sample_slope(stretched.y, slope_x, slope_y, u, v);
  const float cp = ts::cos_phi(stretched), sp = ts::sin_phi(stretched);
  const float tmp = cp * slope_x - sp * slope_y;
  slope_y = sp * slope_x + cp * slope_y;
  slope_x = tmp;
  slope_x = slope_x * ax;
  slope_y = slope_y * ay;
  const v3 wh = normalize_inplace(v3(-slope_x, 1.0f, -slope_y));
  pdf = (ggx_D(p, wh) * ggx_G1(p, wi) * fabsf(dot(wi, wh)) / fabsf(wi.y));
  return wh;
}

// ---- sheen distribution ----------------------------------------------------------------------------
PHX_HD float sheen_L(float x, float r) {
  const float t = (1.0f - r) * (1.0f - r);
  const float a = t * 25.3245f + (1.0f - t) * 21.5473f;
  const float b = t * 3.32435f + (1.0f - t) * 3.82987f;
  const float c = t * 0.16801f + (1.0f - t) * 0.19823f;
  const float d = t * -1.27393f + (1.0f - t) * -1.97760f;
  const float e = t * -4.85967f + (1.0f - t) * -4.32054f;
  const float xc = powf_(x, c);
  return a / (1 + b * xc) + d * x + e;
}
template <typename LobeT>
PHX_HD float sheen_D(const LobeT& p, const v3& v) {
  const float st = ts::sin_theta(v);
  const float oor = 1.0f / p.r;
  return (float)((double)((2.0f + oor) * powf_(st, oor)) / (2.0f * kPiD));
}
template <typename LobeT>
PHX_HD float sheen_Lambda(const LobeT& p, const v3& v, float L5) {
  const float ct = v.y;
  const float l = (ct < 0.5f) ? sheen_L(ct, p.r) : 2.0f * L5 - sheen_L(1.0f - ct, p.r);
  return expf_(l);
}

// ---- Cook-Torrance (local-space inputs; SHEEN selects the distribution) -----------------------------
template <bool SHEEN, typename LobeT>
PHX_HD float ct_f_local(const LobeT& p, const v3& li, const v3& lo, float L5) {
  if (!ts::same_hemi(li, lo)) return 0.0f;
  v3 wh = li + lo;
  const float cos_ti = fabsf(li.y), cos_to = fabsf(lo.y);
  if (cos_ti == 0.0f || cos_to == 0.0f) return 0.0f;
  if (wh.x == 0.0f || wh.y == 0.0f || wh.z == 0.0f) return 0.0f;
  wh = normalize_inplace(wh);
  const float d = SHEEN ? sheen_D(p, wh) : ggx_D(p, wh);
  const float lam_i = SHEEN ? sheen_Lambda(p, li, L5) : ggx_Lambda(p, li);
  const float lam_o = SHEEN ? sheen_Lambda(p, lo, L5) : ggx_Lambda(p, lo);
  const float g = 1.0f / (1.0f + lam_i + lam_o);
  const float whdoty = wh.x * 0.0f + wh.y * 1.0f + wh.z * 0.0f;
  const v3 whf = whdoty < 0.0f ? -wh : wh;
  const float f = fresnel_dielectric(dot(lo, whf), 0.5f);
  return d * g * f * (1.0f / (4.0f * cos_ti * cos_to));
}
template <typename LobeT>
PHX_HD float ct_pdf(const LobeT& p, const v3& wi_world, const v3& li, const v3& lo) {
  if (!ts::same_hemi(li, lo)) return 0.0f;
  const v3 wh = normalize_inplace(li + lo);
  return (ggx_D(p, wh) * ggx_G1(p, wi_world) * fabsf(dot(li, wh)) / fabsf(li.y)) / (4.0f * dot(li, wh));
}
template <typename LobeT>
PHX_HD float ctr_f_local(const LobeT& p, const v3& li, const v3& lo) {
  if (ts::same_hemi(li, lo)) return 0.0f;
  const float eta = li.y > 0.0f ? p.eta : 1.0f / p.eta;
  const float cos_ti = li.y, cos_to = lo.y;
  if (cos_ti == 0.0f || cos_to == 0.0f) return 0.0f;
  v3 wh = normalize_inplace(li + lo * eta);
  if (wh.y < 0.0f) wh = -wh;
  if (dot(lo, wh) * dot(li, wh) > 0.0f) return 0.0f;
  const float f = fresnel_dielectric(dot(lo, wh), eta);
  const float sqrt_denom = dot(li, wh) + eta * dot(lo, wh);
  const float factor = 1.0f / eta;
  const float d = ggx_D(p, wh);
  const float g = 1.0f / (1.0f + ggx_Lambda(p, li) + ggx_Lambda(p, lo));
  return (1.0f - f) * fabsf(d * g * eta * eta * fabsf(dot(lo, wh)) * fabsf(dot(li, wh)) * factor * factor /
                            (cos_ti * cos_to * sqrt_denom * sqrt_denom));
}
template <typename LobeT>
PHX_HD float ctr_pdf(const LobeT& p, const v3& wi_world, const v3& wo_world, const v3& li, const v3& lo) {
  const float eta = li.y > 0.0f ? p.eta : 1.0f / p.eta;
  if ((double)dot(wo_world, wi_world) > 0.0) return 0.0f;
  const v3 wh = normalize_inplace(li + lo * eta);
  const float sqrt_denom = dot(li, wh) + eta * dot(lo, wh);
  const float dwh_dwi = fabsf(eta * eta * dot(lo, wh)) / sqrt_denom * sqrt_denom;
  return (ggx_D(p, wh) * wh.y) * dwh_dwi;
}
template <typename LobeT>
PHX_HD float oren_nayar_f_local(const LobeT& p, const v3& li, const v3& lo) {
  const float cos_theta_i = fabsf(li.y), cos_theta_o = fabsf(lo.y);
  const float sin_theta_i = ts::sin_theta(li), sin_theta_o = ts::sin_theta(lo);
  float max_cos = 0.0f;
  if (sin_theta_i > 0.0001f && sin_theta_o > 0.0001f) {
    const float sin_phi_i = ts::sin_phi(li), cos_phi_i = ts::cos_phi(li);
    const float sin_phi_o = ts::sin_phi(lo), cos_phi_o = ts::cos_phi(lo);
    const float dcos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
    max_cos = fmaxf(0.0f, dcos);
  }
  float sin_alpha, tan_beta;
  if (cos_theta_i > cos_theta_o) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / cos_theta_i; }
  else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / cos_theta_o; }
  const float result = (p.a + p.b * max_cos * sin_alpha * tan_beta);
  return (float)((double)result * kInvPiD);
}

// eval() of src/bsdf.cpp:29-107: value (grey) and pdf of lobe p for the world-space pair (wi, wo)
// DIFFUSE_ONLY: every lobe of every material is Lambert (decided once per scene on the host); the other
// lobe models are compiled out, the arithmetic of the diffuse case is the same code.
template <bool DIFFUSE_ONLY, typename LobeT>
PHX_HD float lobe_eval(const LobeT& p, const v3& n, const Frame& fr, const v3& wi, const v3& wo, float L5, float& pdf) {
  switch (DIFFUSE_ONLY ? (uint32_t)L_DIFFUSE : p.type) {
    case L_DIFFUSE:
      pdf = (float)((double)dot(n, wi) * kInvPiD);
      return (float)kInvPiD;
    case L_OREN_NAYAR:
      pdf = (float)((double)dot(n, wi) * kInvPiD);
      return oren_nayar_f_local(p, fr.to_local(wi), fr.to_local(wo));
    case L_MICROFACET: {
      const v3 li = fr.to_local(wi), lo = fr.to_local(wo);
      if (p.refract) { pdf = ctr_pdf(p, wi, wo, li, lo); return ctr_f_local(p, li, lo); }
      pdf = ct_pdf(p, wi, li, lo);
      return ct_f_local<false>(p, li, lo, L5);
    }
    case L_SHEEN:
      pdf = (float)((double)dot(n, wi) * kInvPiD);
      return ct_f_local<true>(p, fr.to_local(wi), fr.to_local(wo), L5);
    default:  // reflection, refraction, transparent: delta lobes
      pdf = 0.0f;
      return 0.0f;
  }
}

// bsdf_t::f, src/bsdf.cpp:113-131
// MAXL = 1: the caller guarantees num_lobes <= 1; every lobe index is then the constant 0, so a material assembled in registers
// (k_shade<2>: DevMatLite) never has to be addressed dynamically.  Same statements, same order, same results.
// The tangent frame of the hit is the caller's (one per hit, shared with bsdf_sample).  wo = hits.wi (the view direction).
// MatT: DevMaterial, or the same struct in the constant address space (PHX_CONST_MAT: a wave-uniform address is then read through the
// scalar cache into SGPRs — k_shade_g's material-uniform waves).
template <bool DIFFUSE_ONLY = false, int MAXL = 8, bool PERHIT = false, typename MatT>
PHX_HD v3 bsdf_f(const MatT& m, const v3& n, const Frame& fr, const v3& wi, const v3& wo) {
  v3 out(0.0f);
  if (m.num_lobes == 0) return out;
  const float atl = dot(n, wi);
  const bool reflect = atl * dot(n, wo) > 0.0f;
  const uint32_t nl = MAXL == 1 ? 1u : m.num_lobes;
#pragma nounroll
  for (uint32_t i = 0; i < nl; ++i) {
    const auto& p = m.lobes[MAXL == 1 ? 0u : i];
    v3 w;
    if (!lobe_weight_at_hit<PERHIT>(p, n, wo, w)) continue;
    if ((reflect && (p.flags & B_REFLECT)) || (!reflect && (p.flags & B_TRANSMIT))) {
      float ignored;
      v3 wi_ = wi, wo_ = wo;
      if (!DIFFUSE_ONLY) { PHX_PIN_V3(wi_); PHX_PIN_V3(wo_); }
      const float e = lobe_eval<DIFFUSE_ONLY>(p, n, fr, wi_, wo_, m.sheen_L5, ignored);  // evaluated only where it is used: no side effects
      const v3 ew = v3(e) * w;
      out = out + ew * atl;
    }
  }
  return out;
}
template <bool DIFFUSE_ONLY = false, int MAXL = 8, bool PERHIT = false, typename MatT>
PHX_HD v3 bsdf_f(const MatT& m, const v3& n, const v3& wi, const v3& wo) { return bsdf_f<DIFFUSE_ONLY, MAXL, PERHIT>(m, n, Frame(n), wi, wo); }

// bsdf_t::sample, src/bsdf.cpp:133-248.  Returns f (already weighted); pdf == 0 terminates.  wi = hits.wi (the view direction).
// PERHIT: the lobes of the hit are the baked lobes whose resolved weight is not all zero, in table order (lobe_weight_at_hit):
// `keep` has a bit per baked lobe, `lobes` counts them, and the sampled index picks the index-th KEPT lobe.
template <bool DIFFUSE_ONLY = false, int MAXL = 8, bool PERHIT = false, typename MatT>
PHX_HD v3 bsdf_sample(const MatT& m, const v3& n, const Frame& fr, float u1, float u2, const v3& wi, v3& wo, float& pdf, uint32_t& sample_flags) {
  pdf = 0.0f; sample_flags = 0; wo = v3(0.0f);
  uint32_t keep = 0xffu, lobes = MAXL == 1 ? (m.num_lobes ? 1u : 0u) : m.num_lobes;
  if (PERHIT && m.per_hit) {
    keep = 0u; lobes = 0u;
    for (uint32_t i = 0; i < m.num_lobes; ++i) {
      v3 w;
      if (lobe_weight_at_hit<true>(m.lobes[i], n, wi, w)) { keep |= 1u << i; ++lobes; }
    }
  }
  if (lobes == 0) return v3(0.0f);
  const float fl = (float)lobes;
  uint32_t index = (uint32_t)floorf(u1 * fl);
  if (index > lobes - 1) index = lobes - 1;
  if (MAXL == 1) index = 0;  // what the two lines above compute for lobes == 1 and u1 in [0, 1)
  const float u = fminf(u1 * fl - (float)index, 1.0f - FLT_EPSILON);
  uint32_t chosen = index;  // position of the index-th kept lobe in the baked table
  if (PERHIT && m.per_hit) {
    uint32_t seen = 0;
    for (uint32_t i = 0; i < m.num_lobes; ++i)
      if (keep & (1u << i)) { if (seen == index) chosen = i; ++seen; }
  }
  const auto& p = m.lobes[MAXL == 1 ? 0u : chosen];
  float res = 0.0f;
  bool pdf_set = false;
  switch (DIFFUSE_ONLY ? (uint32_t)L_DIFFUSE : p.type) {
    case L_DIFFUSE: {
      v3 l; cosine_weighted(u, u2, l, pdf); pdf_set = true;
      wo = fr.to_world(l);
      res = (float)kInvPiD;
      break;
    }
    case L_OREN_NAYAR: {
      v3 l; cosine_weighted(u, u2, l, pdf); pdf_set = true;
      wo = fr.to_world(l);
      res = oren_nayar_f_local(p, fr.to_local(wi), fr.to_local(wo));
      break;
    }
    case L_SHEEN: {
      v3 l; cosine_weighted(u, u2, l, pdf); pdf_set = true;
      wo = fr.to_world(l);
      res = ct_f_local<true>(p, fr.to_local(wi), fr.to_local(wo), m.sheen_L5);
      break;
    }
    case L_MICROFACET: {
      if (p.refract) {  // microfacet.hpp:118-171
        if (p.eta == 1.0f) { wo = -wi; pdf = 1.0f; pdf_set = true; res = 1.0f; break; }
        const v3 li = fr.to_local(wi);
        if (li.y == 0.0f) break;
        float dpdf;
        const v3 wh = ggx_sample(p, li, dpdf, u, u2);
        if (dot(wh, li) < 0.0f) break;
        const float eta = li.y > 0.0f ? 1.0f / p.eta : p.eta;
        const float cos_ti = dot(wh, li);
        const float sin2_ti = fmaxf(0.0f, 1.0f - cos_ti * cos_ti);
        const float sin2_tt = eta * eta * sin2_ti;
        if (sin2_tt >= 1.0f) break;
        const float cos_tt = sqrtf(1.0f - sin2_tt);
        const v3 lo = eta * -li + (eta * cos_ti - cos_tt) * wh;
        const float sqrt_denom = dot(li, wh) + eta * dot(lo, wh);
        const float dwh_dwi = fabsf((eta * eta * dot(lo, wh)) / (sqrt_denom * sqrt_denom));
        pdf = dpdf * dwh_dwi; pdf_set = true;
        wo = fr.to_world(lo);
        res = ctr_f_local(p, fr.to_local(wi), fr.to_local(wo));
      } else {  // microfacet.hpp:237-277
        const v3 li = fr.to_local(wi);
        if (li.y == 0.0f) break;
        float dpdf;
        const v3 wh = ggx_sample(p, li, dpdf, u, u2);
        if (dot(li, wh) < 0.0f) break;
        const v3 lo = -li + (2.0f * dot(li, wh)) * wh;
        if (!ts::same_hemi(li, lo)) break;
        pdf = dpdf / (4.0f * dot(li, wh)); pdf_set = true;
        wo = fr.to_world(lo);
        res = ct_f_local<false>(p, fr.to_local(wi), fr.to_local(wo), m.sheen_L5);
      }
      break;
    }
    case L_REFLECTION: {
      const float cos_theta = dot(n, wi);
      pdf = 1.0f; pdf_set = true;
      wo = -wi + (2.0f * cos_theta) * n;
      res = 1.0f;
      break;
    }
    case L_REFRACTION: {
      pdf = 1.0f; pdf_set = true;
      float cos_theta = dot(n, wi);
      const float sin_theta = fmaxf(0.0f, 1.0f - cos_theta * cos_theta);
      v3 nn; float eta = p.eta;
      if (cos_theta > 0.0f) { nn = n; eta = 1.0f / eta; } else { nn = -n; cos_theta = -cos_theta; }
      const float arg = 1.0f - (eta * eta * sin_theta);
      if (arg >= 0.0f) {
        const float dnp = sqrtf(arg);
        const float nk = eta * cos_theta - dnp;
        wo = -wi * eta + nn * nk;
        res = 1.0f;
      } else {
        res = 0.0f;  // TIR: defined as black
      }
      break;
    }
    case L_TRANSPARENT:
      wo = -wi; pdf = 1.0f; pdf_set = true; res = 1.0f;
      break;
    default: break;
  }
  if (!pdf_set) pdf = 0.0f;
  if (pdf == 0.0f) return v3(0.0f);
  v3 pw;
  (void)lobe_weight_at_hit<PERHIT>(p, n, wi, pw);
  v3 result = v3(res) * pw;
  int matched = 1;
#pragma nounroll
  for (uint32_t i = 0; MAXL > 1 && i < m.num_lobes; ++i) {
    const auto& q = m.lobes[i];
    if (i != chosen && (keep & (1u << i)) && ((p.flags & q.flags) == q.flags)) {
      const bool reflect = dot(n, wi) * dot(n, wo) > 0.0f;
      if ((reflect && (q.flags & B_REFLECT)) || (!reflect && (q.flags & B_TRANSMIT))) {
        float lobe_pdf = 0.0f;
        v3 wi_ = wi, wo_ = wo;
        if (!DIFFUSE_ONLY) { PHX_PIN_V3(wi_); PHX_PIN_V3(wo_); }
        const float e = lobe_eval<DIFFUSE_ONLY>(q, n, fr, wi_, wo_, m.sheen_L5, lobe_pdf);
        v3 qw;
        (void)lobe_weight_at_hit<PERHIT>(q, n, wi, qw);
        result = result + v3(e) * qw;
        pdf += lobe_pdf;
        ++matched;
      }
    }
  }
  pdf /= (float)matched;
  sample_flags = p.flags;
  return result;
}
template <bool DIFFUSE_ONLY = false, int MAXL = 8, bool PERHIT = false, typename MatT>
PHX_HD v3 bsdf_sample(const MatT& m, const v3& n, float u1, float u2, const v3& wi, v3& wo, float& pdf, uint32_t& sample_flags) {
  return bsdf_sample<DIFFUSE_ONLY, MAXL, PERHIT>(m, n, Frame(n), u1, u2, wi, wo, pdf, sample_flags);
}

}  // namespace phx
