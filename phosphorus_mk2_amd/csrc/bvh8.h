// bvh8.h — the device's acceleration structure: an 8-wide BVH stored as ONE pool of 64-byte elements.  An element is
// either a quantised node ("nodelet": four 16-B words — grid origin, scale exponents, child masks and child base in the
// first, child boxes as 8-bit grid coordinates relative to the node's own origin in the other three) or a Moeller-Trumbore
// triangle record (v0, e0, e1, primitive id, material).  The children of a node — nodelets and triangle records alike, one
// triangle per leaf slot — are contiguous in slot order, so a node needs ONE 32-bit base, and every element is 64-B aligned:
// a 128-B cache line holds exactly two siblings.  It takes the place of accel::mbvh_t (reference src/accel/bvh.hpp:17-49:
// 288-B nodes, 384-B packets) and of the stream traversal in src/kernels/cpu/stream_bvh_kernel.cpp:18-148, redesigned for
// one ray per lane on a 64-lane wavefront:
//   * a node visit is four 16-B loads per lane (the vector L1 charges per lane address, profiles/README.md),
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

// Node origins live on a per-scene grid of 2^18 cells per axis: origin = fma(i, cell, lo) — the SAME fma on the host, in the
// device builder and in the traversal, so the decoded origin is one well-defined float.  The builders pick i so that the
// decoded origin does not exceed the node's true lower corner and quantise the child boxes outwards relative to it.
struct SceneGrid { float lo[3], cell[3]; };
#define PHX_GRID_BITS 18
#define PHX_GRID_MAX ((1u << PHX_GRID_BITS) - 1u)

struct alignas(16) Node8 {
  uint32_t origin_lo, origin_hi;  // bits 0..17 ix | 18..35 iy | 36..53 iz | 54..61 valid mask (bit s: child slot s is in use)
  uint8_t ex, ey, ez, imask;      // grid scale exponents (biased like fp32); bit s of imask: child slot s is an inner node
  uint32_t child_base;            // pool index of the first child; the children of the used slots are contiguous, in slot order
  uint8_t qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node8) == 64, "Node8 must be four 16-byte words");

struct alignas(16) TriRec {  // 64 B: three 16-byte words that the traversal reads + one spare word
  float v0x, v0y, v0z, e0x;
  float e0y, e0z, e1x, e1y;
  float e1z;
  uint32_t prim;      // index in scene_t::triangles() order
  uint32_t material;  // material | smooth << 31 (saves k_shade a dependent load)
  uint32_t pad1;
  uint32_t pad2[4];
};
static_assert(sizeof(TriRec) == 64, "TriRec shares the 64-byte pool with Node8");

union PoolElem { Node8 node; TriRec tri; uint32_t w[16]; };  // host-side view of one pool element
static_assert(sizeof(PoolElem) == 64, "pool elements are 64 bytes");

// One depth limit for every consumer of the tree: the builders refuse deeper trees, k_trace / k_trace_rays size their per-lane
// LDS stacks from the depth (<= 64 levels x 256 lanes x 8 B = 128 KB of the CU's 160 KB).
#define PHX_MAX_BVH_DEPTH 64
// A nodelet staged in LDS keeps an 80-byte stride: with 64 the k-th word of every staged nodelet would fall on 4 of the 16
// 16-byte bank slots (4-way conflicts on a random gather); 5 slots per element spread them over all 16.
#define PHX_NODE_LDS_BYTES 80u

struct Hit { float t, u, v; uint32_t tri; };  // tri = pool index of the TriRec, 0xffffffff = miss

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

PHX_HD void node_origin_encode(Node8& nd, uint32_t ix, uint32_t iy, uint32_t iz, uint32_t valid) {
  const unsigned long long v = (unsigned long long)ix | ((unsigned long long)iy << 18) | ((unsigned long long)iz << 36) | ((unsigned long long)valid << 54);
  nd.origin_lo = (uint32_t)v; nd.origin_hi = (uint32_t)(v >> 32);
}
// origin and valid mask of a node from its first two words
PHX_HD void node_origin_decode(uint32_t w0, uint32_t w1, const SceneGrid& g, float& px, float& py, float& pz, uint32_t& valid) {
  const uint32_t ix = w0 & PHX_GRID_MAX, iy = ((w0 >> 18) | (w1 << 14)) & PHX_GRID_MAX, iz = (w1 >> 4) & PHX_GRID_MAX;
  px = fmaf((float)ix, g.cell[0], g.lo[0]);
  py = fmaf((float)iy, g.cell[1], g.lo[1]);
  pz = fmaf((float)iz, g.cell[2], g.lo[2]);
  valid = (w1 >> 22) & 0xffu;
}
// How far outside a triangle the reference's Moeller-Trumbore (mt_intersect below) can still report a hit: u, v and u + v are tested after
// fp32 rounding of sums that cancel — t = o - v0 of the scene's size times edge vectors of the triangle's — and a ray that misses a 0.01-wide
// facet by 3.4e-6 (exact arithmetic: u + v = 1.00038, 38 x epsilon x the distance) was accepted in round 6; a box that holds the triangle exactly
// then culls a hit the triangle test would report, and the closest hit depends on the tree (one pixel of 2 M x 256 samples differed between the
// device-built and the host-built tree of the closed showroom; profiles/r06_m_*).  Both builders therefore inflate every TRIANGLE's box by
// delta = 2^-18 x (largest extent of the scene + largest coordinate) — 64 x epsilon x the scene's size, ten times the miss that was seen — before
// anything else; node boxes are unions of those.  (No finite delta is a proof: the tolerance grows without bound for grazing rays and slivers.  It is a
// margin, sized on the one violation 10^11 rays produced; k_trace pays +0.3 ... +0.8 % for it, profiles/r06_m_tri_box_inflation_ab.log.)
#ifndef PHX_TRI_BOX_INFLATE
#define PHX_TRI_BOX_INFLATE 1  /* 0: A/B only (the trees of rounds 1-5) */
#endif
PHX_HD float tri_box_inflation(const float* lo, const float* hi) {
  if (!PHX_TRI_BOX_INFLATE) return 0.0f;
  float ext = 0.0f, mag = 0.0f;
  for (int a = 0; a < 3; ++a) {
    ext = fmaxf(ext, hi[a] - lo[a]);
    mag = fmaxf(mag, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
  }
  return (ext + mag) * 3.814697265625e-6f;  // 2^-18
}
// largest grid index whose decoded coordinate does not exceed x (0 when x lies below the grid)
PHX_HD uint32_t grid_index_below(float x, float lo, float cell) {
  double t = floor(((double)x - (double)lo) / (double)cell);
  if (!(t > 0.0)) t = 0.0;
  if (t > (double)PHX_GRID_MAX) t = (double)PHX_GRID_MAX;
  uint32_t i = (uint32_t)t;
  while (i > 0u && fmaf((float)i, cell, lo) > x) --i;
  return i;
}
// the scene grid of a bounding box [lo, hi]: 2^18 - 1 cells of equal size per axis, a positive cell even for flat extents
PHX_HD SceneGrid make_scene_grid(const float* lo, const float* hi) {
  SceneGrid g;
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = lo[a];
    const double ext = (double)hi[a] - (double)lo[a];
    float c = (float)(ext / (double)PHX_GRID_MAX);
    if (!(c > 0.0f) || !(c < 3.0e38f)) c = 1.0e-30f;
    g.cell[a] = c;
  }
  return g;
}

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

struct RayCtx {
  v3 o, d;
  float idx, idy, idz;   // clamped reciprocal direction
  uint32_t oct_inv;      // (dx>=0?4:0)|(dy>=0?2:0)|(dz>=0?1:0)
};
// The reciprocal direction only feeds the conservative box tests, never a result: on the device it is the hardware's v_rcp_f32
// (1 ulp) instead of an IEEE division (10 instructions each, three per ray: k_trace is VALU-issue bound); the slab ends are
// padded by 4 ulp (node_hit8), which covers it.
#ifndef PHX_FAST_RCP
#define PHX_FAST_RCP 1
#endif
PHX_HD float ray_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && PHX_FAST_RCP
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
PHX_HD RayCtx make_ray_ctx(const v3& o, const v3& d) {
  RayCtx r; r.o = o; r.d = d;
  const float big = 1e20f, tiny = 1e-20f;
  r.idx = fabsf(d.x) > tiny ? ray_rcp(d.x) : (d.x < 0.0f ? -big : big);
  r.idy = fabsf(d.y) > tiny ? ray_rcp(d.y) : (d.y < 0.0f ? -big : big);
  r.idz = fabsf(d.z) > tiny ? ray_rcp(d.z) : (d.z < 0.0f ? -big : big);
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
// Conservative: the exit distance is scaled by (1 + 2^-20) before the comparison with the entry distance (8 ulp between
// them; tn >= 0, and a negative exit distance is a miss either way), IEEE maxNum/minNum.  (Two measured-and-rejected encodings of
// the same test — v_pk_fma_f32 plane pairs, fp16 planes through v_perm_b32 + v_fma_mix_f32 — are kept as
// profiles/r06_bvh8_rejected_encodings.patch, not here.)
#ifndef PHX_SIGN_ACCUM
#define PHX_SIGN_ACCUM 1
#endif
#ifndef PHX_NO_NEG_ZERO
#define PHX_NO_NEG_ZERO 1
#endif
#ifndef PHX_PAD_FMA
#define PHX_PAD_FMA 1  /* the padded exit distance minus the entry distance as ONE fma (one instruction less per child; -0.3 ms of 47: profiles/r03_zz_pad_fma_ab.log) */
#endif
#ifndef PHX_BITOP3
#define PHX_BITOP3 1
#endif
PHX_HD uint32_t node_hit8(const uint32_t* w /* 16 words of the node */, float px, float py, float pz, const RayCtx& r, float tmax) {
  const uint32_t e = w[2];
  const float sx = u32_as_f32((e & 0xffu) << 23), sy = u32_as_f32(((e >> 8) & 0xffu) << 23), sz = u32_as_f32(((e >> 16) & 0xffu) << 23);
  const float ax = sx * r.idx, ay = sy * r.idy, az = sz * r.idz;
  const float bx = (px - r.o.x) * r.idx, by = (py - r.o.y) * r.idy, bz = (pz - r.o.z) * r.idz;
  const bool nx = r.idx < 0.0f, ny = r.idy < 0.0f, nz = r.idz < 0.0f;
  const float pad_far = 1.00000095367431640625f;  // 1 + 2^-20
  // words: 4,5 qlox | 6,7 qloy | 8,9 qloz | 10,11 qhix | 12,13 qhiy | 14,15 qhiz
  uint32_t hit8 = 0;
#if defined(__HIP_DEVICE_COMPILE__) && PHX_SIGN_ACCUM
  // Same test, cheaper instructions (scripts/micro/valu_ops.hip: v_mul/v_add/v_sub issue in 2 clocks, v_max/v_min/v_cmp/
  // v_cndmask/v_cvt in 4): the clamps against 0 and tmax become two subtractions, "miss" is the sign bit of
  // (tf - tn) | (tmax - tn) | tf, and a funnel shift collects it — no compare, no select.  Children run 7..0 so that child j
  // ends at bit j.  tf + 0 turns an exit distance of -0 into +0 (t = -0 is a valid Moeller-Trumbore distance).
  uint32_t miss = 0;
  const float tmaxp = tmax * pad_far;
#if PHX_NO_NEG_ZERO
  // b + 0 is never -0, and then no plane distance fma(q, a, b) is (q a = -0 meets b = +0; exact cancellation gives +0): the exit
  // distance needs no "+ 0" per child to keep t = -0 from reading as a miss (three adds per node instead of eight)
  const float bx0 = bx + 0.0f, by0 = by + 0.0f, bz0 = bz + 0.0f;
#define bx bx0
#define by by0
#define bz bz0
#endif
#pragma unroll
  for (int half = 1; half >= 0; --half) {
    const uint32_t nearx = nx ? w[10 + half] : w[4 + half], farx = nx ? w[4 + half] : w[10 + half];
    const uint32_t neary = ny ? w[12 + half] : w[6 + half], fary = ny ? w[6 + half] : w[12 + half];
    const uint32_t nearz = nz ? w[14 + half] : w[8 + half], farz = nz ? w[8 + half] : w[14 + half];
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const int sh = 8 * j;
      const float tnx = fmaf((float)((nearx >> sh) & 0xffu), ax, bx);
      const float tny = fmaf((float)((neary >> sh) & 0xffu), ay, by);
      const float tnz = fmaf((float)((nearz >> sh) & 0xffu), az, bz);
      const float tfx = fmaf((float)((farx >> sh) & 0xffu), ax, bx);
      const float tfy = fmaf((float)((fary >> sh) & 0xffu), ay, by);
      const float tfz = fmaf((float)((farz >> sh) & 0xffu), az, bz);
      const float tn = fmaxf(fmaxf(tnx, tny), tnz);
      const float tf = fminf(fminf(tfx, tfy), tfz);
#if PHX_PAD_FMA && PHX_NO_NEG_ZERO && PHX_BITOP3
      // the three-way OR as v_bitop3_b32 (truth table 0xfe): 2.2 clocks where v_or3_b32 takes 4.1 (profiles/r05_valu_ops.json)
      const uint32_t m = __builtin_amdgcn_bitop3_b32(__float_as_uint(fmaf(tf, pad_far, -tn)), __float_as_uint(tmaxp - tn), __float_as_uint(tf), 0xfe);
#elif PHX_PAD_FMA && PHX_NO_NEG_ZERO
      const uint32_t m = __float_as_uint(fmaf(tf, pad_far, -tn)) | __float_as_uint(tmaxp - tn) | __float_as_uint(tf);
#elif PHX_PAD_FMA
      const uint32_t m = __float_as_uint(fmaf(tf, pad_far, -tn)) | __float_as_uint(tmaxp - tn) | __float_as_uint(tf + 0.0f);  // tf * pad - tn in one instruction
#else
      const uint32_t m = __float_as_uint(tf * pad_far - tn) | __float_as_uint(tmaxp - tn) | __float_as_uint(tf + 0.0f);
#endif
      miss = __builtin_amdgcn_alignbit(miss, m, 31);  // (miss << 1) | (m >> 31)
    }
  }
  hit8 = ~miss & 0xffu;
#if PHX_NO_NEG_ZERO
#undef bx
#undef by
#undef bz
#endif
#else
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const uint32_t nearx = nx ? w[10 + half] : w[4 + half], farx = nx ? w[4 + half] : w[10 + half];
    const uint32_t neary = ny ? w[12 + half] : w[6 + half], fary = ny ? w[6 + half] : w[12 + half];
    const uint32_t nearz = nz ? w[14 + half] : w[8 + half], farz = nz ? w[8 + half] : w[14 + half];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sh = 8 * j;
      const float tnx = fmaf((float)((nearx >> sh) & 0xffu), ax, bx);
      const float tny = fmaf((float)((neary >> sh) & 0xffu), ay, by);
      const float tnz = fmaf((float)((nearz >> sh) & 0xffu), az, bz);
      const float tfx = fmaf((float)((farx >> sh) & 0xffu), ax, bx);
      const float tfy = fmaf((float)((fary >> sh) & 0xffu), ay, by);
      const float tfz = fmaf((float)((farz >> sh) & 0xffu), az, bz);
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
      const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax)) * pad_far;
      if (tn <= tf) hit8 |= 1u << (4 * half + j);  // empty slots have inverted boxes (qlo 255 > qhi 0)
    }
  }
#endif
  return hit8;
}

// Hit mask of a node for one ray: bits 24..31 = hit inner children, XOR-permuted by the ray's octant so that "highest bit
// first" visits them front to back; bits 16..23 = hit leaf children (triangles) by slot — they are tested before the walk
// descends, in slot order: their order changes no result (the closest hit and its tie rule do not depend on it) and walking
// nodelets and triangles in one front-to-back order measured 3 % slower; bits 0..7 = the node's valid mask (the rank of a slot
// among the valid ones is its child's offset from child_base).  `perm8(mask)` is the octant permutation: perm_xor8 in plain
// code, a 2 KB table in LDS inside k_trace (k_trace is VALU-issue bound: the table is 4 % faster, profiles/README.md).
template <typename Perm>
PHX_HD uint32_t node_hitmask(const uint32_t* w, const SceneGrid& g, const RayCtx& r, float tmax, Perm&& perm8) {
  float px, py, pz; uint32_t valid;
  node_origin_decode(w[0], w[1], g, px, py, pz, valid);
  const uint32_t hit8 = node_hit8(w, px, py, pz, r, tmax) & valid;
  const uint32_t imask = w[2] >> 24;
  return (perm8(hit8 & imask) << 24) | ((hit8 & ~imask) << 16) | valid;
}
PHX_HD uint32_t node_hitmask(const uint32_t* w, const SceneGrid& g, const RayCtx& r, float tmax) {
  return node_hitmask(w, g, r, tmax, [&](uint32_t m) { return perm_xor8(m, r.oct_inv); });
}

// Closest-hit (ANY=false) or any-hit (ANY=true) traversal of one ray.  Stack: push(uint32,uint32),
// pop(uint32&,uint32&), empty().  Counters are optional (host-side validation only).
// State of a node group: `base` = child_base of the parent, `hits` = (pending inner hits << 24) | valid mask of the parent.
template <bool ANY, typename Stack>
PHX_HD bool traverse8(const uint32_t* __restrict__ pool /* 16 words per element */, const SceneGrid& grid,
                      const v3& o, const v3& d, float tmax, Hit& hit, Stack& stack,
                      uint32_t* node_visits = nullptr, uint32_t* tri_tests = nullptr) {
  const RayCtx r = make_ray_ctx(o, d);
  hit.t = tmax; hit.u = 0.0f; hit.v = 0.0f; hit.tri = 0xffffffffu;
  uint32_t best_prim = 0;
  uint32_t ng_base = 0, ng_hits = 0x80000000u;  // the root as a one-child group (valid byte 0: its rank is 0 whatever the octant)
  for (;;) {
    // visit the nearest not-yet-visited inner child of the current group
    const uint32_t bit = 31u - (uint32_t)clz32(ng_hits);
    const uint32_t rest = ng_hits & ~(1u << bit);
    if (rest > 0x00ffffffu) stack.push(ng_base, rest);
    const uint32_t slot = (bit - 24u) ^ r.oct_inv;
    const uint32_t ni = ng_base + (uint32_t)popc32(ng_hits & 0xffu & ~(0xffffffffu << slot));
    uint32_t w[16];
    {
      const uint32_t* src = pool + (size_t)ni * 16u;
#if defined(__HIP_DEVICE_COMPILE__)
      const uint4* s4 = reinterpret_cast<const uint4*>(src);
#pragma unroll
      for (int k = 0; k < 4; ++k) { uint4 q = s4[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
#else
      for (int k = 0; k < 16; ++k) w[k] = src[k];
#endif
    }
    if (node_visits) ++*node_visits;
    const uint32_t hm = node_hitmask(w, grid, r, hit.t);
    ng_base = w[3];
    ng_hits = hm & 0xff0000ffu;
    uint32_t th = (hm >> 16) & 0xffu;
    while (th) {
      const uint32_t k = 31u - (uint32_t)clz32(th);
      th &= ~(1u << k);
      const uint32_t ti = ng_base + (uint32_t)popc32(hm & 0xffu & ~(0xffffffffu << k));
      const TriRec T = *reinterpret_cast<const TriRec*>(pool + (size_t)ti * 16u);
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
