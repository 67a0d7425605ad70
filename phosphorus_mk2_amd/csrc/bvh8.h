// bvh8.h — the device's acceleration structure: an 8-wide BVH with 80-byte quantised nodes
// ("nodelets": five 16-B words, child boxes as 8-bit grid coordinates relative to the node's own
// box) and 48-byte Moeller-Trumbore triangle records (v0, e0, e1, primitive id).  It takes the
// place of accel::mbvh_t (reference src/accel/bvh.hpp:17-49: 288-B nodes, 384-B packets) and of
// the stream traversal in src/kernels/cpu/stream_bvh_kernel.cpp:18-148, redesigned for one ray
// per lane on a 64-lane wavefront:
//   * children are stored in "octant order" slots so the visiting order comes from the ray's sign
//     bits alone — no per-lane distance sort, no per-child ray lists (the MBVH-RS lanes_t),
//   * one stack entry is a (base, hit-bitmask) group, so the per-lane stack is <= tree depth,
//   * the box test is conservative (boxes are quantised outwards, both slab ends padded), hence
//     the set of triangles a ray is tested against always contains every triangle it can hit and
//     the closest hit is independent of tree topology; the triangle test itself is the reference's
//     arithmetic (src/accel/triangle.hpp:149-164: FMA cross/dot, true division, eps 1e-8,
//     u>=0, v>=0, u+v<=1, 0<=t<tmax) evaluated per lane.
#pragma once
#include "phx_math.h"

namespace phx {

struct alignas(16) Node8 {
  float px, py, pz;           // origin of the quantisation grid (node box min)
  uint8_t ex, ey, ez, imask;  // grid scale exponents (biased like fp32), bit i of imask: child slot i is an inner node
  uint32_t child_base;        // index of this node's first inner child (children are contiguous, in slot order)
  uint32_t tri_base;          // index of this node's first triangle record
  uint32_t tmask;             // bit (s + 8*j), j < 3: leaf slot s holds a j-th triangle; records are stored in bit order
  uint32_t pad;
  uint8_t qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be five 16-byte words");

struct TriRec {  // 48 B: three 16-byte words
  float v0x, v0y, v0z, e0x;
  float e0y, e0z, e1x, e1y;
  float e1z;
  uint32_t prim;      // index in scene_t::triangles() order
  uint32_t material;  // material | smooth << 31 (filled by the device after the build: saves k_shade a dependent load)
  uint32_t pad1;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be three 16-byte words");

// One depth limit for every consumer of the tree: the builders refuse deeper trees, k_trace / k_trace_rays size their per-lane
// LDS stacks from the depth (<= 64 levels x 256 lanes x 8 B = 128 KB of the CU's 160 KB).
#define PHX_MAX_BVH_DEPTH 64
#define PHX_NODE_LDS_BYTES 80u  /* bytes a staged nodelet occupies in LDS */

struct Hit { float t, u, v; uint32_t tri; };  // tri = index of the TriRec, 0xffffffff = miss

PHX_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)x);
#else
  return x ? __builtin_clz(x) : 32;
#endif
}
PHX_HD int popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(x);
#else
  return __builtin_popcount(x);
#endif
}
PHX_HD float u32_as_f32(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }

// Reference Moeller-Trumbore (src/accel/triangle.hpp:149-164) for one ray and one triangle.
// `best_prim`: primitive of the hit that set tmax (0 while the ray has none).  Two triangles can be hit at bitwise the same
// distance (they intersect each other); the reference then keeps whichever its traversal order meets first (strict d < tmax),
// which makes its result depend on the tree.  Here the tie goes to the lowest primitive index, so the closest hit is a function
// of the ray and the triangle set alone — whichever builder made the tree.
PHX_HD bool mt_intersect(const TriRec& T, const v3& o, const v3& wi, float tmax, uint32_t best_prim, float& us, float& vs, float& ds) {
  const v3 e0(T.e0x, T.e0y, T.e0z), e1(T.e1x, T.e1y, T.e1z), v0(T.v0x, T.v0y, T.v0z);
  const v3 t = o - v0;
  const v3 p = scross(wi, e1);
  const float det = sdot(e0, p);
  const float ood = 1.0f / det;
  const v3 q = scross(t, e0);
  us = sdot(t, p) * ood;
  vs = sdot(wi, q) * ood;
  ds = sdot(e1, q) * ood;
  const bool xmask = (det > 0.00000001f) || (det < -0.00000001f);
  const bool umask = us >= 0.0f;
  const bool vmask = (vs >= 0.0f) && ((us + vs) <= 1.0f);
  const bool dmask = (ds >= 0.0f) && ((ds < tmax) || (ds == tmax && T.prim < best_prim));
  return vmask && umask && dmask && xmask;
}

// The 8 box tests of one node for one ray.  Returns the CWBVH-style hit mask: inner children set
// bit 24 + (slot ^ oct_inv), the triangles of a hit leaf child set their bits in [0,24).
struct RayCtx {
  v3 o, d;
  float idx, idy, idz;   // clamped reciprocal direction
  uint32_t oct_inv;      // (dx>=0?4:0)|(dy>=0?2:0)|(dz>=0?1:0)
};
PHX_HD RayCtx make_ray_ctx(const v3& o, const v3& d) {
  RayCtx r; r.o = o; r.d = d;
  const float big = 1e20f, tiny = 1e-20f;
  r.idx = fabsf(d.x) > tiny ? 1.0f / d.x : (d.x < 0.0f ? -big : big);
  r.idy = fabsf(d.y) > tiny ? 1.0f / d.y : (d.y < 0.0f ? -big : big);
  r.idz = fabsf(d.z) > tiny ? 1.0f / d.z : (d.z < 0.0f ? -big : big);
  r.oct_inv = (d.x < 0.0f ? 0u : 4u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 1u);
  return r;
}

// XOR-permutation of the low 8 bits of x: bit i moves to bit (i ^ oct), oct in [0,8)
PHX_HD uint32_t perm_xor8(uint32_t x, uint32_t oct) {
  x = (oct & 4u) ? (((x << 4) | (x >> 4)) & 0xffu) : x;
  x = (oct & 2u) ? (((x & 0x33u) << 2) | ((x & 0xccu) >> 2)) : x;
  x = (oct & 1u) ? (((x & 0x55u) << 1) | ((x & 0xaau) >> 1)) : x;
  return x;
}

// The 8 box tests of one node for one ray: bit i of the result = child slot i may be hit.
// Conservative: entry distance scaled by (1 - 2^-21), exit distance by (1 + 2^-21) (4 ulp each; tn >= 0, and a
// negative exit distance is a miss either way), IEEE maxNum/minNum.  On the device the near/far planes of an
// axis go through one packed FMA (v_pk_fma_f32) and the two pads through one packed multiply.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float phx_f2 __attribute__((ext_vector_type(2)));
#endif
PHX_HD uint32_t node_hit8(const uint32_t* w /* 20 words of the node */, const RayCtx& r, float tmax) {
  const float px = u32_as_f32(w[0]), py = u32_as_f32(w[1]), pz = u32_as_f32(w[2]);
  const uint32_t e = w[3];
  const float sx = u32_as_f32((e & 0xffu) << 23), sy = u32_as_f32(((e >> 8) & 0xffu) << 23), sz = u32_as_f32(((e >> 16) & 0xffu) << 23);
  const float ax = sx * r.idx, ay = sy * r.idy, az = sz * r.idz;
  const float bx = (px - r.o.x) * r.idx, by = (py - r.o.y) * r.idy, bz = (pz - r.o.z) * r.idz;
  const bool nx = r.idx < 0.0f, ny = r.idy < 0.0f, nz = r.idz < 0.0f;
  const float pad_near = 0.999999523162841796875f, pad_far = 1.000000476837158203125f;  // 1 -/+ 2^-21
  // words: 8,9 qlox | 10,11 qloy | 12,13 qloz | 14,15 qhix | 16,17 qhiy | 18,19 qhiz
  uint32_t hit8 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  const phx_f2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz}, pad2 = {pad_near, pad_far};
#endif
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const uint32_t nearx = nx ? w[14 + half] : w[8 + half], farx = nx ? w[8 + half] : w[14 + half];
    const uint32_t neary = ny ? w[16 + half] : w[10 + half], fary = ny ? w[10 + half] : w[16 + half];
    const uint32_t nearz = nz ? w[18 + half] : w[12 + half], farz = nz ? w[12 + half] : w[18 + half];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sh = 8 * j;
#if defined(__HIP_DEVICE_COMPILE__)
      const phx_f2 qx = {(float)((nearx >> sh) & 0xffu), (float)((farx >> sh) & 0xffu)};
      const phx_f2 qy = {(float)((neary >> sh) & 0xffu), (float)((fary >> sh) & 0xffu)};
      const phx_f2 qz = {(float)((nearz >> sh) & 0xffu), (float)((farz >> sh) & 0xffu)};
      const phx_f2 tx = __builtin_elementwise_fma(qx, ax2, bx2);
      const phx_f2 ty = __builtin_elementwise_fma(qy, ay2, by2);
      const phx_f2 tz = __builtin_elementwise_fma(qz, az2, bz2);
      phx_f2 t = {fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f)), fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax))};
      t = t * pad2;
      if (t.x <= t.y) hit8 |= 1u << (4 * half + j);  // empty slots have inverted boxes (qlo 255 > qhi 0)
#else
      const float tnx = fmaf((float)((nearx >> sh) & 0xffu), ax, bx);
      const float tny = fmaf((float)((neary >> sh) & 0xffu), ay, by);
      const float tnz = fmaf((float)((nearz >> sh) & 0xffu), az, bz);
      const float tfx = fmaf((float)((farx >> sh) & 0xffu), ax, bx);
      const float tfy = fmaf((float)((fary >> sh) & 0xffu), ay, by);
      const float tfz = fmaf((float)((farz >> sh) & 0xffu), az, bz);
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f)) * pad_near;
      const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax)) * pad_far;
      if (tn <= tf) hit8 |= 1u << (4 * half + j);
#endif
    }
  }
  return hit8;
}

// CWBVH-style hit mask of a node: inner children set bit 24 + (slot ^ oct_inv) (so that "highest bit first"
// visits them in the ray's octant order), the triangles of hit leaf slots set their bits in [0,24).
PHX_HD uint32_t node_hitmask(const uint32_t* w, const RayCtx& r, float tmax) {
  const uint32_t hit8 = node_hit8(w, r, tmax);
  const uint32_t imask = w[3] >> 24;
  const uint32_t leaf = hit8 & ~imask;
  return (perm_xor8(hit8 & imask, r.oct_inv) << 24) | ((leaf | (leaf << 8) | (leaf << 16)) & w[6]);
}

// Closest-hit (ANY=false) or any-hit (ANY=true) traversal of one ray.  Stack: push(uint32,uint32),
// pop(uint32&,uint32&), empty().  Counters are optional (host-side validation only).
template <bool ANY, typename Stack>
PHX_HD bool traverse8(const uint32_t* __restrict__ nodes /* 20 words per node */, const TriRec* __restrict__ tris,
                      const v3& o, const v3& d, float tmax, Hit& hit, Stack& stack,
                      uint32_t* node_visits = nullptr, uint32_t* tri_tests = nullptr) {
  const RayCtx r = make_ray_ctx(o, d);
  hit.t = tmax; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xffffffffu;
  uint32_t best_prim = 0;
  uint32_t ng_base = 0, ng_hits = 0x80000000u;  // the root as a one-child group
  for (;;) {
    // visit the nearest not-yet-visited inner child of the current group
    const uint32_t bit = 31u - (uint32_t)clz32(ng_hits);
    const uint32_t rest = ng_hits & ~(1u << bit);
    if (rest > 0x00ffffffu) stack.push(ng_base, rest);
    const uint32_t slot = (bit - 24u) ^ r.oct_inv;
    const uint32_t rel = (uint32_t)popc32(ng_hits & 0xffu & ~(0xffffffffu << slot));
    const uint32_t ni = ng_base + rel;
    uint32_t w[20];
    {
      const uint32_t* src = nodes + (size_t)ni * 20u;
#if defined(__HIP_DEVICE_COMPILE__)
      const uint4* s4 = reinterpret_cast<const uint4*>(src);
#pragma unroll
      for (int k = 0; k < 5; ++k) { uint4 q = s4[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
#else
      for (int k = 0; k < 20; ++k) w[k] = src[k];
#endif
    }
    if (node_visits) ++*node_visits;
    const uint32_t hm = node_hitmask(w, r, hit.t);
    ng_base = w[4];
    ng_hits = (hm & 0xff000000u) | (w[3] >> 24);
    uint32_t th = hm & 0x00ffffffu;
    const uint32_t tb = w[5], tm = w[6];
    while (th) {
      const uint32_t k = 31u - (uint32_t)clz32(th);
      th &= ~(1u << k);
      const uint32_t ti = tb + (uint32_t)popc32(tm & ~(0xffffffffu << k));
      const TriRec T = tris[ti];
      float us, vs, ds;
      if (tri_tests) ++*tri_tests;
      if (mt_intersect(T, o, d, hit.t, best_prim, us, vs, ds)) {
        hit.t = ds; hit.u = us; hit.v = vs; hit.tri = ti; best_prim = T.prim;
        if (ANY) return true;
      }
    }
    if (ng_hits <= 0x00ffffffu) {
      if (stack.empty()) break;
      stack.pop(ng_base, ng_hits);
    }
  }
  return hit.tri != 0xffffffffu;
}

}  // namespace phx
