// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// obvh.h: the reference's 8-wide BVH, restated.
//   * node / packet layout: mbvh::node_t<8> (src/accel/bvh/node.hpp:12-68, 288 B) and
//     accel::triangle::moeller_trumbore_t<8> (src/accel/triangle.hpp:25-68, 384 B)
//   * builder: bvh::from / find / split / largest_node (src/accel/bvh/binned_sah_builder.hpp:143-281)
//     and the adapter accel::builder_t (src/accel/bvh.cpp:22-79)
//   * traversal: MBVH-RS stream traversal `intersect` (src/kernels/cpu/stream_bvh_kernel.cpp:18-148,
//     detail/stream.hpp:16-111), slab test simd::intersect<8> (src/math/simd/aabb.hpp:26-62),
//     Moeller-Trumbore iterate_rays / iterate_triangles (src/accel/triangle.hpp:126-287)
//   * brute force: linear_mbvh_kernel_t (src/kernels/cpu/linear_bvh_kernel.cpp:14-19)
// Counters (node visits, packet visits) define V_n / V_l of SURVEY §8(d).
// The slab test and the triangle test exist twice: spelled out for one child box / one triangle (the readable
// restatement), and on 8 AVX2 lanes — the shape the reference itself uses — which is the default because it is what
// bench.py times as the CPU baseline.  Both perform the same IEEE operations in the same order (bit-identical, tested).
#pragma once
#include "oscene.h"

#include <algorithm>
#include <cstring>
#include <limits>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#include <xmmintrin.h>
#endif

namespace orc {

static const uint32_t F_HIT = 1, F_MASKED = 2, F_SHADOW = 4, F_SPECULAR = 8;  // state.hpp:33-36

// numeric modes (SURVEY §7 hard part 2)
inline int& scalar_default() { static int v = 0; return v; }
inline int& debug_nonfinite() { static int v = 0; return v; }
inline int* debug_pixel() { static int v[2] = {-1, -1}; return v; }  // orc_set_debug_pixel: orender.cpp prints the rays of this film pixel (diagnostic)  // orc_set_debug_nonfinite: orender.cpp reports where a non-finite value enters a path
inline int& tie_default() { static int v = 0; return v; }     // process-wide default of modes_t::tie_lowest_prim (orc_set_tie_rule)  // process-wide default of modes_t::scalar (orc_set_scalar)
struct modes_t {
  int rcp_approx = 0;   // 1: use the x86 RCPPS approximation where the reference does (this CPU only)
  int slab_literal = 0; // 1: reference slab test verbatim; 0: conservative (padded) slab test — never
                        //    loses a Moeller-Trumbore hit, so results do not depend on BVH topology
  int tie_lowest_prim = tie_default();  // 0 (the reference): of two triangles hit at bitwise the same distance the one met first
                        //    wins (strict d < ray.d, triangle.hpp:158-164) — an accident of the tree's layout; 1: the one with the
                        //    lower primitive index wins, which is what the device does (bvh8.h) and makes whole-frame
                        //    comparisons exact.  Measured difference: 1 of 8.8 M rays of the 100 k soup at 4 spp.
  int scalar = scalar_default();  // 1: one child box / one triangle at a time (the spelled-out restatement); 0: the same arithmetic
                        //    on 8 lanes with AVX2, as the reference's own simd::intersect<8> / moeller_trumbore_t<8> do.
                        //    Every lane performs the same IEEE operations in the same order, so the two are bit-identical
                        //    (tests/test_oracle_render.py::test_simd_and_scalar_restatements_agree).
};

struct node8_t {  // node.hpp:12-35
  float bounds[48];
  uint32_t offset[8];
  uint8_t num[8];
  uint32_t flags[8];
  uint8_t pad[24];
  node8_t() {
    for (int i = 0; i < 24; ++i) { bounds[i] = FLT_MAX; bounds[i + 24] = -FLT_MAX; }
    std::memset(offset, 0, sizeof(offset)); std::memset(num, 0, sizeof(num));
    std::memset(flags, 0, sizeof(flags)); std::memset(pad, 0, sizeof(pad));
  }
  void set_bounds(uint32_t i, const Box3& b) {
    bounds[i] = b.min.x; bounds[i + 8] = b.min.y; bounds[i + 16] = b.min.z;
    bounds[i + 24] = b.max.x; bounds[i + 32] = b.max.y; bounds[i + 40] = b.max.z;
  }
};
static_assert(sizeof(node8_t) == 288, "reference node_t<8> is 288 B (SURVEY App. C)");

struct packet8_t {  // triangle.hpp:33-68
  float e0x[8], e0y[8], e0z[8], e1x[8], e1y[8], e1z[8], v0x[8], v0y[8], v0z[8];
  uint32_t num;
  uint32_t meshid[8];
  uint32_t faceid[8];
  uint32_t prim[8];  // oracle extra: index into scene.triangles() order (not part of the 384 B)
};

// ---- builder ---------------------------------------------------------------------------------
struct primitive_t { uint32_t index; Box3 bounds; V3 centroid; };

struct geometry_t {  // binned_sah_builder.hpp:23-75
  std::vector<primitive_t>* prims;
  uint32_t start, end;
  Box3 bounds, centroid_bounds;
  geometry_t() : prims(nullptr), start(0), end(0) {}
  explicit geometry_t(std::vector<primitive_t>* p) : prims(p), start(0), end(0) {}
  geometry_t(std::vector<primitive_t>* p, uint32_t s, uint32_t e) : prims(p), start(s), end(e) {
    for (uint32_t i = s; i < e; ++i) { bounds.extendBy((*p)[i].bounds); centroid_bounds.extendBy((*p)[i].centroid); }
  }
  uint32_t count() const { return end - start; }
};

inline float box_area(const Box3& b) {  // aabb::area, math/aabb.hpp:11-14 (2.0 is a double)
  V3 d = b.max - b.min;
  return (float)(2.0 * (double)(d.x * d.y + d.x * d.z + d.y * d.z));
}
inline uint32_t bin_of(const Box3& l, const primitive_t& p, int axis) {  // bins_t::offset/find :113-124
  V3 o = p.centroid - l.min;
  if (l.max.x > l.min.x) o.x /= (l.max.x - l.min.x);
  if (l.max.y > l.min.y) o.y /= (l.max.y - l.min.y);
  if (l.max.z > l.min.z) o.z /= (l.max.z - l.min.z);
  return (uint32_t)std::min((int)(12 * o[axis]), 11);
}
struct split_t { uint32_t axis, bin; float cost; };

inline split_t find_split(const geometry_t& g) {  // bvh::find :143-189
  uint32_t best_axis = 0, best_bin = 0;
  float best_cost = FLT_MAX;
  for (int axis = 0; axis < 3; ++axis) {
    struct { uint32_t count = 0; Box3 bounds; } bins[12];
    if (g.centroid_bounds.max[axis] < g.centroid_bounds.min[axis]) continue;
    for (uint32_t i = 0; i < g.count(); ++i) {
      const primitive_t& p = (*g.prims)[g.start + i];
      auto& b = bins[bin_of(g.centroid_bounds, p, axis)];
      b.bounds.extendBy(p.bounds); b.count++;
    }
    float split_cost = FLT_MAX; uint32_t split_bin = 0;
    for (int i = 0; i < 11; ++i) {
      Box3 a, b; int left = 0, right = 0;
      for (int j = 0; j <= i; ++j) { a.extendBy(bins[j].bounds); left += bins[j].count; }
      for (int j = i + 1; j < 12; ++j) { b.extendBy(bins[j].bounds); right += bins[j].count; }
      float cost = (left * box_area(a) + right * box_area(b)) / box_area(g.bounds);
      if (cost < split_cost) { split_cost = cost; split_bin = i; }
    }
    if (split_cost < best_cost) { best_axis = axis; best_cost = split_cost; best_bin = split_bin; }
  }
  return split_t{best_axis, best_bin, best_cost};
}

inline void do_split(const split_t& s, geometry_t& parent, geometry_t& l, geometry_t& r) {  // bvh::split :191-196
  std::vector<primitive_t>& P = *parent.prims;
  const Box3 cb = parent.centroid_bounds;
  auto* first = &P[0] + parent.start;
  auto* last = &P[0] + parent.end;  // &primitives[end-1]+1
  auto* mid = std::partition(first, last, [&](const primitive_t& p) { return bin_of(cb, p, s.axis) <= s.bin; });
  uint32_t m = (uint32_t)(mid - &P[0]);
  l = geometry_t(parent.prims, parent.start, m);
  r = geometry_t(parent.prims, m, parent.end);
}

struct bvh8_t {
  std::vector<node8_t> nodes;
  std::vector<packet8_t> packets;
  bool has_root = false;

  uint32_t add(uint32_t begin, uint32_t end, const std::vector<primitive_t>& prims, const std::vector<tri_ref_t>& tris) {
    uint32_t off = (uint32_t)packets.size();  // accel::builder_t::add, bvh.cpp:58-78
    for (uint32_t i = begin; i < end; i += 8) {
      uint32_t num = std::min(8u, end - i);
      packet8_t pk; std::memset(&pk, 0, sizeof(pk));
      pk.num = num;
      for (uint32_t j = 0; j < num; ++j) {
        const tri_ref_t& t = tris[prims[i + j].index];
        const V3 a = t.a(), b = t.b(), c = t.c();
        const V3 e0 = b - a, e1 = c - a;
        pk.e0x[j] = e0.x; pk.e0y[j] = e0.y; pk.e0z[j] = e0.z;
        pk.e1x[j] = e1.x; pk.e1y[j] = e1.y; pk.e1z[j] = e1.z;
        pk.v0x[j] = a.x; pk.v0y[j] = a.y; pk.v0z[j] = a.z;
        pk.meshid[j] = t.meshid() | (t.matid() << 16);
        pk.faceid[j] = t.face;
        pk.prim[j] = prims[i + j].index;
      }
      packets.push_back(pk);
    }
    return off;
  }

  // bvh::from(geometry, things, bvh), binned_sah_builder.hpp:215-270
  uint32_t build_rec(geometry_t& g, const std::vector<tri_ref_t>& tris) {
    split_t s = find_split(g);
    if (g.count() < 8 || (float)g.count() <= 1.0f + s.cost) return 0;
    int num_children = 2;
    geometry_t children[8];
    for (auto& c : children) c = geometry_t(g.prims);
    do_split(s, g, children[0], children[1]);
    while (num_children < 8) {
      int split_child = -1;  // largest_node :198-213 — picks the SMALLEST area child with >= 8 prims
      float a = FLT_MAX;
      for (int i = 0; i < num_children; ++i) {
        if (children[i].count() < 8) continue;
        float na = box_area(children[i].bounds);
        if (na < a) { split_child = i; a = na; }
      }
      if (split_child == -1) break;
      split_t s2 = find_split(children[split_child]);
      geometry_t tmp(g.prims);
      do_split(s2, children[split_child], tmp, children[num_children]);
      children[split_child] = tmp;
      ++num_children;
    }
    nodes.emplace_back();
    uint32_t node_index = (uint32_t)nodes.size() - 1;
    uint32_t child_indices[8];
    for (int i = 0; i < num_children; ++i) child_indices[i] = build_rec(children[i], tris);
    for (int i = 0; i < num_children; ++i) {
      node8_t& node = nodes[node_index];
      node.set_bounds(i, children[i].bounds);
      if (child_indices[i]) {
        node.offset[i] = child_indices[i];
      } else {
        uint32_t index = add(children[i].start, children[i].end, *g.prims, tris);
        node8_t& nd = nodes[node_index];
        nd.flags[i] = 1; nd.offset[i] = index; nd.num[i] = (uint8_t)children[i].count();
      }
    }
    return node_index;
  }

  void build(const std::vector<tri_ref_t>& tris) {  // bvh::from(builder, things) :272-281
    nodes.clear(); packets.clear();
    std::vector<primitive_t> prims(tris.size());
    for (uint32_t i = 0; i < tris.size(); ++i) { prims[i].index = i; prims[i].bounds = tris[i].bounds(); prims[i].centroid = prims[i].bounds.center(); }
    geometry_t g(&prims, 0, (uint32_t)prims.size());
    build_rec(g, tris);
    has_root = !nodes.empty();  // SURVEY A-13: < 8 triangles (or cheap leaf) leaves no root node
  }
};

// ---- ray stream (ray_t<N>, state.hpp:40-169) as a plain SoA of arbitrary length ----------------
struct rays_t {
  std::vector<float> px, py, pz, wx, wy, wz, d, u, v;
  std::vector<uint32_t> mesh, face, flags, prim;
  void resize(size_t n) {
    px.resize(n); py.resize(n); pz.resize(n); wx.resize(n); wy.resize(n); wz.resize(n); d.resize(n); u.resize(n); v.resize(n);
    mesh.resize(n); face.resize(n); flags.resize(n); prim.resize(n, 0xffffffffu);
  }
  size_t size() const { return d.size(); }
  V3 p(uint32_t i) const { return V3(px[i], py[i], pz[i]); }
  V3 wi(uint32_t i) const { return V3(wx[i], wy[i], wz[i]); }
  bool is_hit(uint32_t i) const { return (flags[i] & F_HIT) == F_HIT; }
  bool is_masked(uint32_t i) const { return (flags[i] & F_MASKED) == F_MASKED; }
  bool is_shadow(uint32_t i) const { return (flags[i] & F_SHADOW) == F_SHADOW; }
  bool is_specular(uint32_t i) const { return (flags[i] & F_SPECULAR) == F_SPECULAR; }
  bool is_occluded(uint32_t i) const { return is_hit(i) || is_masked(i); }
  uint32_t meshid(uint32_t i) const { return mesh[i] & 0xffffu; }
  uint32_t matid(uint32_t i) const { return (mesh[i] & 0xffff0000u) >> 16; }
};

struct trace_counters_t { uint64_t rays = 0, node_visits = 0, packet_visits = 0; };

#if defined(__x86_64__)
inline float rcp_approx(float x) { return _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x))); }
#else
inline float rcp_approx(float x) { return 1.0f / x; }
#endif

// one ray against one triangle of a packet: the body of iterate_rays / iterate_triangles
// (triangle.hpp:149-164 and :232-247 are the same arithmetic)
inline bool mt_test(const packet8_t& pk, int j, const V3& o, const V3& wi, float dmax, float& us, float& vs, float& ds) {
  const V3 e0(pk.e0x[j], pk.e0y[j], pk.e0z[j]), e1(pk.e1x[j], pk.e1y[j], pk.e1z[j]), v0(pk.v0x[j], pk.v0y[j], pk.v0z[j]);
  const V3 t = o - v0;
  const V3 p = sv::cross(wi, e1);
  const float det = sv::dot(e0, p);
  const float ood = 1.0f / det;
  const V3 q = sv::cross(t, e0);
  us = sv::dot(t, p) * ood;
  vs = sv::dot(wi, q) * ood;
  ds = sv::dot(e1, q) * ood;
  const bool xmask = (det > 0.00000001f) || (det < -0.00000001f);
  const bool umask = us >= 0.0f;
  const bool vmask = (vs >= 0.0f) && ((us + vs) <= 1.0f);
  const bool dmask = (ds >= 0.0f) && (ds < dmax);
  return vmask && umask && dmask && xmask;
}

// packet vs. one ray, closest-of-packet by strict '<' in lane order (triangle.hpp:166-198)
inline void packet_vs_ray(const packet8_t& pk, rays_t& R, uint32_t index, bool tie = false) {
  const V3 o = R.p(index), wi = R.wi(index);
  float closest = R.d[index];
  int idx = -1; float bu = 0, bv = 0;
  // tie rule (modes_t::tie_lowest_prim): a hit at exactly the current best distance replaces it if its primitive index is lower
  bool have = tie && R.is_hit(index) && !R.is_shadow(index);
  uint32_t best_prim = have ? R.prim[index] : 0u;
  for (int j = 0; j < (int)pk.num; ++j) {
    float us, vs, ds;
    if (!tie) {
      if (mt_test(pk, j, o, wi, R.d[index], us, vs, ds) && ds < closest) { closest = ds; idx = j; bu = us; bv = vs; }
    } else if (mt_test(pk, j, o, wi, std::numeric_limits<float>::infinity(), us, vs, ds) &&
               (ds < closest || (have && ds == closest && pk.prim[j] < best_prim))) {
      closest = ds; idx = j; bu = us; bv = vs; best_prim = pk.prim[j]; have = !R.is_shadow(index);
    }
  }
  if (idx != -1) {
    if (!R.is_shadow(index)) { R.mesh[index] = pk.meshid[idx]; R.face[index] = pk.faceid[idx]; R.u[index] = bu; R.v[index] = bv; R.prim[index] = pk.prim[idx]; }
    R.flags[index] |= F_HIT; R.d[index] = closest;  // ray_t::hit, state.hpp:118-123
  }
}

// The reference's AVX2 wrapper semantics on one lane (src/math/simd/float8.hpp):
//   select(m,l,r) = _mm256_blendv_ps(l,r,m): r where the mask's sign bit is set, else l   (:103-105)
//   max(a,b) = _mm256_max_ps: a > b ? a : b  (b when either is NaN); min likewise with <   (:59-65)
//   compares are ordered (false on NaN), all-ones / all-zeros lanes                         (:67-85)
// Checked against the reference's own object code in tests/golden/ref_subset_vectors.npz.
inline float simd_select(bool mask, float l, float r) { return mask ? r : l; }
inline float simd_max(float a, float b) { return a > b ? a : b; }
inline float simd_min(float a, float b) { return a < b ? a : b; }
// __bscf (src/utils/compiler.hpp:6-14): index of the lowest set bit, which is then cleared
inline uint64_t bscf(uint64_t& v) { uint64_t i = 0; while (!((v >> i) & 1ull)) ++i; v &= v - 1; return i; }

// simd::intersect<8> for one child box (aabb.hpp:26-62); ood = 1/dir (or RCPPS in approx mode)
inline bool slab_literal(const node8_t& n, int c, const V3& o, const V3& ood, float d, float& dist) {
  const float bminx = n.bounds[c], bminy = n.bounds[c + 8], bminz = n.bounds[c + 16];
  const float bmaxx = n.bounds[c + 24], bmaxy = n.bounds[c + 32], bmaxz = n.bounds[c + 40];
  const bool gx = ood.x >= 0.0f, gy = ood.y >= 0.0f, gz = ood.z >= 0.0f;
  float nx = simd_select(gx, bmaxx, bminx), fx = simd_select(gx, bminx, bmaxx);  // aabb.hpp:38-43
  float ny = simd_select(gy, bmaxy, bminy), fy = simd_select(gy, bminy, bmaxy);
  float nz = simd_select(gz, bmaxz, bminz), fz = simd_select(gz, bminz, bmaxz);
  nx = (nx - o.x) * ood.x; ny = (ny - o.y) * ood.y; nz = (nz - o.z) * ood.z;
  fx = (fx - o.x) * ood.x; fy = (fy - o.y) * ood.y; fz = (fz - o.z) * ood.z;
  const float nn = simd_max(simd_max(nx, ny), simd_max(nz, 0.0f));
  const float ff = simd_min(simd_min(fx, fy), simd_min(fz, d));
  dist = nn;
  return nn <= ff;
}
// conservative variant: IEEE maxNum/minNum (NaN-ignoring), both ends padded by 4 ulp — which covers the rounding of THIS arithmetic — and the
// box itself inflated by 2^-18 x (|o|_1 + the box's largest coordinate), which covers the triangle test's: Moeller-Trumbore accepts u, v within a few
// epsilon x |o - v0| / |edge| of the triangle's border, i.e. hit points up to a few 1e-7 x distance OUTSIDE the triangle (and its box).  Round 6 found
// the 4-ulp form rejecting the leaf of a mirror-sphere facet whose edge a ray ran along: the restatement's traversal returned the neighbouring facet
// (t 1.4738613) where testing ALL triangles — and the device — find t 1.4738580 (profiles/r06_l_oracle_slab_miss.json); 1 ray in 115 M.
inline float slab_inflation(const node8_t& n, int c, const V3& o) {
  float m = std::fabs(n.bounds[c]);
  for (int k = 1; k < 6; ++k) m = std::fmax(m, std::fabs(n.bounds[c + 8 * k]));
  return 3.814697265625e-6f * (std::fabs(o.x) + std::fabs(o.y) + std::fabs(o.z) + m);
}
inline bool slab_conservative(const node8_t& n, int c, const V3& o, const V3& ood, float d, float& dist) {
  const float pad = slab_inflation(n, c, o);
  const float bminx = n.bounds[c] - pad, bminy = n.bounds[c + 8] - pad, bminz = n.bounds[c + 16] - pad;
  const float bmaxx = n.bounds[c + 24] + pad, bmaxy = n.bounds[c + 32] + pad, bmaxz = n.bounds[c + 40] + pad;
  float nx = (ood.x >= 0.0f) ? bminx : bmaxx, fx = (ood.x >= 0.0f) ? bmaxx : bminx;
  float ny = (ood.y >= 0.0f) ? bminy : bmaxy, fy = (ood.y >= 0.0f) ? bmaxy : bminy;
  float nz = (ood.z >= 0.0f) ? bminz : bmaxz, fz = (ood.z >= 0.0f) ? bmaxz : bminz;
  nx = (nx - o.x) * ood.x; ny = (ny - o.y) * ood.y; nz = (nz - o.z) * ood.z;
  fx = (fx - o.x) * ood.x; fy = (fy - o.y) * ood.y; fz = (fz - o.z) * ood.z;
  float nn = std::fmax(std::fmax(nx, ny), std::fmax(nz, 0.0f));
  float ff = std::fmin(std::fmin(fx, fy), std::fmin(fz, d));
  dist = nn;
  nn = nn - std::fabs(nn) * 4.76837158203125e-7f;  // 4 ulp
  ff = ff + std::fabs(ff) * 4.76837158203125e-7f;
  return nn <= ff;
}

// ---- the same two tests on 8 lanes (AVX2) ------------------------------------------------------------------
// One ray x the 8 child boxes of a node: bit i of the result = child i is hit, dist8[i] = its (unpadded) entry distance.
inline unsigned slab8(const node8_t& n, const V3& o, const V3& ood, float d, bool literal, float* dist8) {
  __m256 minx = _mm256_loadu_ps(n.bounds), miny = _mm256_loadu_ps(n.bounds + 8), minz = _mm256_loadu_ps(n.bounds + 16);
  __m256 maxx = _mm256_loadu_ps(n.bounds + 24), maxy = _mm256_loadu_ps(n.bounds + 32), maxz = _mm256_loadu_ps(n.bounds + 40);
  if (!literal) {  // the conservative form inflates every child box (slab_inflation above, the same operations per lane)
    const __m256 am = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff));
    __m256 m = _mm256_and_ps(minx, am);
    m = _mm256_max_ps(m, _mm256_and_ps(miny, am)); m = _mm256_max_ps(m, _mm256_and_ps(minz, am));
    m = _mm256_max_ps(m, _mm256_and_ps(maxx, am)); m = _mm256_max_ps(m, _mm256_and_ps(maxy, am)); m = _mm256_max_ps(m, _mm256_and_ps(maxz, am));
    const float ro = std::fabs(o.x) + std::fabs(o.y) + std::fabs(o.z);
    const __m256 pad = _mm256_mul_ps(_mm256_set1_ps(3.814697265625e-6f), _mm256_add_ps(_mm256_set1_ps(ro), m));
    minx = _mm256_sub_ps(minx, pad); miny = _mm256_sub_ps(miny, pad); minz = _mm256_sub_ps(minz, pad);
    maxx = _mm256_add_ps(maxx, pad); maxy = _mm256_add_ps(maxy, pad); maxz = _mm256_add_ps(maxz, pad);
  }
  const bool gx = ood.x >= 0.0f, gy = ood.y >= 0.0f, gz = ood.z >= 0.0f;
  const __m256 ox = _mm256_set1_ps(o.x), oy = _mm256_set1_ps(o.y), oz = _mm256_set1_ps(o.z);
  const __m256 rx = _mm256_set1_ps(ood.x), ry = _mm256_set1_ps(ood.y), rz = _mm256_set1_ps(ood.z);
  const __m256 nx = _mm256_mul_ps(_mm256_sub_ps(gx ? minx : maxx, ox), rx), fx = _mm256_mul_ps(_mm256_sub_ps(gx ? maxx : minx, ox), rx);
  const __m256 ny = _mm256_mul_ps(_mm256_sub_ps(gy ? miny : maxy, oy), ry), fy = _mm256_mul_ps(_mm256_sub_ps(gy ? maxy : miny, oy), ry);
  const __m256 nz = _mm256_mul_ps(_mm256_sub_ps(gz ? minz : maxz, oz), rz), fz = _mm256_mul_ps(_mm256_sub_ps(gz ? maxz : minz, oz), rz);
  const __m256 zero = _mm256_setzero_ps(), dd = _mm256_set1_ps(d);
  __m256 nn, ff;
  if (literal) {  // simd_max / simd_min are _mm256_max_ps / _mm256_min_ps
    nn = _mm256_max_ps(_mm256_max_ps(nx, ny), _mm256_max_ps(nz, zero));
    ff = _mm256_min_ps(_mm256_min_ps(fx, fy), _mm256_min_ps(fz, dd));
    _mm256_storeu_ps(dist8, nn);
  } else {        // IEEE maxNum / minNum: the non-NaN operand wins (max_ps returns its second operand when either is NaN)
    auto maxnum = [](__m256 a, __m256 b) { return _mm256_blendv_ps(_mm256_max_ps(a, b), a, _mm256_cmp_ps(b, b, _CMP_UNORD_Q)); };
    auto minnum = [](__m256 a, __m256 b) { return _mm256_blendv_ps(_mm256_min_ps(a, b), a, _mm256_cmp_ps(b, b, _CMP_UNORD_Q)); };
    nn = maxnum(maxnum(nx, ny), maxnum(nz, zero));
    ff = minnum(minnum(fx, fy), minnum(fz, dd));
    _mm256_storeu_ps(dist8, nn);
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff)), eps = _mm256_set1_ps(4.76837158203125e-7f);
    nn = _mm256_sub_ps(nn, _mm256_mul_ps(_mm256_and_ps(nn, absmask), eps));
    ff = _mm256_add_ps(ff, _mm256_mul_ps(_mm256_and_ps(ff, absmask), eps));
  }
  return (unsigned)_mm256_movemask_ps(_mm256_cmp_ps(nn, ff, _CMP_LE_OQ));
}

// One ray x the (up to) 8 triangles of a packet: mt_test on 8 lanes, then the closest-of-packet selection in lane order.
inline void packet_vs_ray_simd(const packet8_t& pk, rays_t& R, uint32_t index, bool tie = false) {
  const V3 o = R.p(index), wi = R.wi(index);
  const __m256 e0x = _mm256_loadu_ps(pk.e0x), e0y = _mm256_loadu_ps(pk.e0y), e0z = _mm256_loadu_ps(pk.e0z);
  const __m256 e1x = _mm256_loadu_ps(pk.e1x), e1y = _mm256_loadu_ps(pk.e1y), e1z = _mm256_loadu_ps(pk.e1z);
  const __m256 wx = _mm256_set1_ps(wi.x), wy = _mm256_set1_ps(wi.y), wz = _mm256_set1_ps(wi.z);
  const __m256 tx = _mm256_sub_ps(_mm256_set1_ps(o.x), _mm256_loadu_ps(pk.v0x));
  const __m256 ty = _mm256_sub_ps(_mm256_set1_ps(o.y), _mm256_loadu_ps(pk.v0y));
  const __m256 tz = _mm256_sub_ps(_mm256_set1_ps(o.z), _mm256_loadu_ps(pk.v0z));
  // sv::cross(a, b) = (msub(a.y,b.z, a.z*b.y), msub(a.z,b.x, a.x*b.z), msub(a.x,b.y, a.y*b.x)); sv::dot = madd(ax,bx, madd(ay,by, az*bz))
  const __m256 px = _mm256_fmsub_ps(wy, e1z, _mm256_mul_ps(wz, e1y)), py = _mm256_fmsub_ps(wz, e1x, _mm256_mul_ps(wx, e1z)),
               pz = _mm256_fmsub_ps(wx, e1y, _mm256_mul_ps(wy, e1x));
  const __m256 det = _mm256_fmadd_ps(e0x, px, _mm256_fmadd_ps(e0y, py, _mm256_mul_ps(e0z, pz)));
  const __m256 ood = _mm256_div_ps(_mm256_set1_ps(1.0f), det);
  const __m256 qx = _mm256_fmsub_ps(ty, e0z, _mm256_mul_ps(tz, e0y)), qy = _mm256_fmsub_ps(tz, e0x, _mm256_mul_ps(tx, e0z)),
               qz = _mm256_fmsub_ps(tx, e0y, _mm256_mul_ps(ty, e0x));
  const __m256 us = _mm256_mul_ps(_mm256_fmadd_ps(tx, px, _mm256_fmadd_ps(ty, py, _mm256_mul_ps(tz, pz))), ood);
  const __m256 vs = _mm256_mul_ps(_mm256_fmadd_ps(wx, qx, _mm256_fmadd_ps(wy, qy, _mm256_mul_ps(wz, qz))), ood);
  const __m256 ds = _mm256_mul_ps(_mm256_fmadd_ps(e1x, qx, _mm256_fmadd_ps(e1y, qy, _mm256_mul_ps(e1z, qz))), ood);
  const __m256 zero = _mm256_setzero_ps(), one = _mm256_set1_ps(1.0f), eps = _mm256_set1_ps(0.00000001f);
  const __m256 xmask = _mm256_or_ps(_mm256_cmp_ps(det, eps, _CMP_GT_OQ), _mm256_cmp_ps(det, _mm256_sub_ps(zero, eps), _CMP_LT_OQ));
  const __m256 umask = _mm256_cmp_ps(us, zero, _CMP_GE_OQ);
  const __m256 vmask = _mm256_and_ps(_mm256_cmp_ps(vs, zero, _CMP_GE_OQ), _mm256_cmp_ps(_mm256_add_ps(us, vs), one, _CMP_LE_OQ));
  const __m256 dlim = _mm256_set1_ps(R.d[index]);
  const __m256 dmask = _mm256_and_ps(_mm256_cmp_ps(ds, zero, _CMP_GE_OQ), tie ? _mm256_cmp_ps(ds, dlim, _CMP_LE_OQ) : _mm256_cmp_ps(ds, dlim, _CMP_LT_OQ));
  unsigned m = (unsigned)_mm256_movemask_ps(_mm256_and_ps(_mm256_and_ps(vmask, umask), _mm256_and_ps(dmask, xmask)));
  m &= pk.num >= 8 ? 0xffu : ((1u << pk.num) - 1u);
  if (!m) return;
  float u8[8], v8[8], d8[8];
  _mm256_storeu_ps(u8, us); _mm256_storeu_ps(v8, vs); _mm256_storeu_ps(d8, ds);
  float closest = R.d[index]; int idx = -1;
  bool have = tie && R.is_hit(index) && !R.is_shadow(index);
  uint32_t best_prim = have ? R.prim[index] : 0u;
  for (int j = 0; j < 8; ++j)
    if (((m >> j) & 1u) && (d8[j] < closest || (have && d8[j] == closest && pk.prim[j] < best_prim))) {
      closest = d8[j]; idx = j; best_prim = pk.prim[j]; have = tie && !R.is_shadow(index);
    }
  if (idx != -1) {
    if (!R.is_shadow(index)) { R.mesh[index] = pk.meshid[idx]; R.face[index] = pk.faceid[idx]; R.u[index] = u8[idx]; R.v[index] = v8[idx]; R.prim[index] = pk.prim[idx]; }
    R.flags[index] |= F_HIT; R.d[index] = closest;
  }
}

// MBVH-RS stream traversal over R[0..num) (stream_bvh_kernel.cpp:18-148).  Slots are traced
// unless MASKED (lanes_t::init, stream.hpp:25-32).
struct stream_tracer_t {
  const bvh8_t* bvh;
  modes_t modes;
  std::vector<uint32_t> lanes[8];
  std::vector<uint32_t> lane_num;  // unused; lanes[i].size() is the fill
  struct task_t { uint32_t offset, num_rays, lane, flags, prims; };
  std::vector<task_t> tasks;
  std::vector<V3> inv_dir;
  trace_counters_t ctr;

  explicit stream_tracer_t(const bvh8_t* b, modes_t m = modes_t()) : bvh(b), modes(m) {}

  void trace(rays_t& R, uint32_t num) {
    for (auto& l : lanes) l.clear();
    tasks.clear();
    // 1/dir is a function of the ray alone: computed once here instead of at every node visit (same values)
    inv_dir.resize(num);
    uint64_t not_finite = 0;
    for (uint32_t i = 0; i < num; ++i) if (!R.is_masked(i)) {
      const V3 w = R.wi(i), o = R.p(i);
      // A NaN ray (the thin lens with a zero lens sample, camera.hpp:140-147) hits nothing HERE.  In the reference it passes every box test
      // (simd max / min return their second operand on NaN: n = 0, f = d, aabb.hpp:56-59) — the empty child slots too, whose offset 0 is
      // the root: the stream re-enters the root for ever.  A stated deviation: the ray is counted and left alone.
      if (!std::isfinite((o.x + o.y + o.z) + (w.x + w.y + w.z))) { ++not_finite; continue; }
      lanes[0].push_back(i);
      inv_dir[i] = modes.rcp_approx ? V3(rcp_approx(w.x), rcp_approx(w.y), rcp_approx(w.z)) : V3(1.0f / w.x, 1.0f / w.y, 1.0f / w.z);
    }
    ctr.rays += lanes[0].size() + not_finite;
    if (lanes[0].empty() || !bvh->has_root) return;
    tasks.push_back(task_t{0, (uint32_t)lanes[0].size(), 0, 0, 0});
    std::vector<uint32_t> todo;
    while (!tasks.empty()) {
      task_t cur = tasks.back(); tasks.pop_back();
      // pop(lanes, cur.lane, cur.num_rays), stream.hpp:105-110
      std::vector<uint32_t>& L = lanes[cur.lane];
      todo.assign(L.end() - cur.num_rays, L.end());
      L.resize(L.size() - cur.num_rays);
      if (cur.flags != 1) {
        const node8_t& node = bvh->nodes[cur.offset];
        int num_active[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float length[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t ray : todo) {
          if (R.is_shadow(ray) && R.is_hit(ray)) continue;  // any-hit early out, :61-64
          const V3 o = R.p(ray);
          const V3 ood = inv_dir[ray];
          ++ctr.node_visits;
          if (modes.scalar) {
            for (int c = 0; c < 8; ++c) {
              float dist;
              bool h = modes.slab_literal ? slab_literal(node, c, o, ood, R.d[ray], dist) : slab_conservative(node, c, o, ood, R.d[ray], dist);
              if (h) { num_active[c] += 1; length[c] += dist; lanes[c].push_back(ray); }
            }
          } else {
            float dist8[8];
            unsigned hm = slab8(node, o, ood, R.d[ray], modes.slab_literal != 0, dist8);
            while (hm) {
              const int c = __builtin_ctz(hm); hm &= hm - 1;
              num_active[c] += 1; length[c] += dist8[c]; lanes[c].push_back(ray);
            }
          }
        }
        // insertion sort of the hit children by summed entry distance, :99-115 (verbatim: the
        // inner loop compares d with dists[ids[j]] and never stops early)
        uint32_t ids[8]; int n = 0;
        for (int i = 0; i < 8; ++i) {
          if (num_active[i] > 0) {
            float d = length[i];
            ids[n] = i;
            for (int j = n; j > 0; --j) {
              if (d < length[ids[j]]) { uint32_t t = ids[j]; ids[j] = ids[j - 1]; ids[j - 1] = t; }
            }
            ++n;
          }
        }
        for (int i = 0; i < n; ++i)
          tasks.push_back(task_t{node.offset[ids[i]], (uint32_t)num_active[ids[i]], ids[i], node.flags[ids[i]], node.num[ids[i]]});
      } else {
        // leaf: groups of <= 8 rays x the leaf's packets, :122-145
        for (size_t begin = 0; begin < todo.size(); begin += 8) {
          size_t nr = std::min(todo.size() - begin, (size_t)8);
          uint32_t index = cur.offset; uint32_t prims = 0;
          do {
            if (index < bvh->packets.size()) {  // guard for the empty-leaf quirk (count 0 leaves)
              for (size_t r = 0; r < nr; ++r) {
                ++ctr.packet_visits;
                if (modes.scalar) packet_vs_ray(bvh->packets[index], R, todo[begin + r], modes.tie_lowest_prim != 0);
                else packet_vs_ray_simd(bvh->packets[index], R, todo[begin + r], modes.tie_lowest_prim != 0);
              }
            }
            prims += 8; ++index;
          } while (prims < cur.prims);
        }
      }
    }
  }
};

// linear_mbvh_kernel_t: every non-masked ray against every packet (linear_bvh_kernel.cpp:14-19)
inline void trace_brute(const bvh8_t& bvh, rays_t& R, uint32_t num, bool tie = false) {
  for (uint32_t i = 0; i < num; ++i) {
    if (R.is_masked(i)) continue;
    for (const packet8_t& pk : bvh.packets) packet_vs_ray(pk, R, i, tie);
  }
}

}  // namespace orc
