// bvh_gpu.hip — device-side construction of the BVH8 of bvh8.h (replace the host
// recursion of the reference builder, src/accel/bvh/binned_sah_builder.hpp:216-281, by a GPU build so that
// preprocess() can rebuild per frame like cpu_t::preprocess does, src/xpu/cpu.cpp:35-44,219).
//
// LBVH pipeline, everything resident in HBM:
//   k_prim_bounds   triangle boxes + centroid bounds of the scene (wave shuffle reduce, ordered-uint atomics)
//   k_morton        63-bit Morton code of the centroid (21 bits per axis) + primitive index
//   rocprim::radix_sort_pairs                      (library radix sort of 8-byte keys: not a hot-path kernel)
//   k_radix_tree    Karras 2012: one thread per internal node of the binary radix tree over the sorted codes
//   k_fit           bottom-up boxes + leaf counts (second arrival at a node continues upwards)
//   k_collapse      per BVH8 node, level by level: greedy surface-area expansion to <= 8 children (single triangles become
//                   leaf slots), octant-order slot assignment, origin on the scene grid, outward quantisation, the
//                   children — nodelets and triangle records — allocated contiguously in the 64-byte pool, in slot
//                   order: the same node semantics as the host builder.
// Traversal results do not depend on which builder made the tree (conservative box tests, bvh8.h).
#include "bvh_gpu.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#ifndef PHX_STUDY_KNOBS
#define PHX_STUDY_KNOBS 0  /* 1: the builder's study knobs (PHX_LBVH_COLLAPSE, PHX_CNODE) are read from the environment; the product library reads none (bvh_build.cpp) */
#endif
static inline const char* gpu_study_knob(const char* name) {
#if PHX_STUDY_KNOBS
  return getenv(name);
#else
  (void)name; return nullptr;
#endif
}
namespace phx {
namespace {

#define GB 256

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  union { uint32_t u; float f; } c; c.u = u; return c.f;
}

struct Box6 { float lo[3], hi[3]; };

// Grid-stride: a wave folds many triangles into its 12 extremes before it touches the 12 shared words — one wave per 64 triangles
// meant 1.9 M atomics on one cache line for 10 M triangles (21 ms at the ~90 atomics/us one line sustains).
__global__ void __launch_bounds__(GB) k_prim_bounds(const float* __restrict__ abc, uint32_t n, Box6* __restrict__ pbox, uint32_t* __restrict__ cb /* 12 ordered uints: centroid lo/hi, geometry lo/hi */) {
  float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float glo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, ghi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = blockIdx.x * GB + threadIdx.x; i < n; i += gridDim.x * GB) {
    const float* t = abc + 9 * (size_t)i;
    Box6 b;
    for (int a = 0; a < 3; ++a) {
      b.lo[a] = fminf(fminf(t[a], t[3 + a]), t[6 + a]);
      b.hi[a] = fmaxf(fmaxf(t[a], t[3 + a]), t[6 + a]);
      const float c = 0.5f * (b.lo[a] + b.hi[a]);
      clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
      glo[a] = fminf(glo[a], b.lo[a]); ghi[a] = fmaxf(ghi[a], b.hi[a]);
    }
    pbox[i] = b;
  }
  for (int a = 0; a < 3; ++a) {
    float lo = clo[a], hi = chi[a];
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    float l2 = glo[a], h2 = ghi[a];
    for (int o = 32; o > 0; o >>= 1) { l2 = fminf(l2, __shfl_xor(l2, o)); h2 = fmaxf(h2, __shfl_xor(h2, o)); }
    if (__lane_id() == 0) { atomicMin(&cb[a], f2ord(lo)); atomicMax(&cb[3 + a], f2ord(hi)); atomicMin(&cb[6 + a], f2ord(l2)); atomicMax(&cb[9 + a], f2ord(h2)); }
  }
}

__device__ __forceinline__ uint64_t spread21(uint32_t x) {  // 21 bits -> every third bit of 63
  uint64_t v = x & 0x1fffffu;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// Extended Morton codes (Vinkler, Bittner, Havran 2017): the SIZE of a primitive is a fourth coordinate of the sort key, so that the
// few huge triangles of a scene (a room's walls around finely tessellated objects) separate from the small ones near the top of the
// tree instead of inflating the box of every node they share a Morton cell with.  The key is "xyzs" fifteen times — 15 bits per
// axis + 15 size bits, size = box diagonal / scene diagonal, linear.  Against plain 63-bit Morton codes (PHX_EMC=0): k_trace -2.4 %
// (Soup 100 k), -2.1 % (1 M), -2.6 % (config 4), -5.5 % (showroom 1 M); against the host's binned-SAH tree: -2.4 / -4.5 % on the
// soups, -3 ... -4 % on the showroom (profiles/r03_x_emc_probe.log ... r03_za_builder_ab.log).  Measured and worse: "xyzxyzs" nine
// times (half the gain), a square-root size scale, a gain of 4 / 16 / 64 on the size, no size bit in the first 1 / 2 / 4 groups,
// the size bit in front of its group (r03_y, r03_z, r03_zj_emc_skip.log, r03_zk_emc_first.log).
#ifndef PHX_EMC
#define PHX_EMC 1
#endif
__global__ void __launch_bounds__(GB) k_morton(const Box6* __restrict__ pbox, uint32_t n, const uint32_t* __restrict__ cb, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * GB + threadIdx.x;
  if (i >= n) return;
  uint32_t q[3];
  float d2 = 0.0f, D2 = 0.0f;
  for (int a = 0; a < 3; ++a) {
    const float lo = ord2f(cb[a]), hi = ord2f(cb[3 + a]);
    const float c = 0.5f * (pbox[i].lo[a] + pbox[i].hi[a]);
    const float ext = hi - lo;
    float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    q[a] = (uint32_t)fminf(u * 2097152.0f, 2097151.0f);
    const float e = pbox[i].hi[a] - pbox[i].lo[a], E = ord2f(cb[9 + a]) - ord2f(cb[6 + a]);
    d2 += e * e; D2 += E * E;
  }
#if PHX_EMC
  const float rel = D2 > 0.0f ? sqrtf(d2 / D2) : 0.0f;  // the box diagonal relative to the scene's
  const uint32_t sz = (uint32_t)fminf(rel * 32768.0f, 32767.0f);
  uint64_t key = 0;
#pragma unroll
  for (int g = 0; g < 15; ++g) {  // group g: bit 20 - g of every axis, then bit 14 - g of the size
    const int bit = 20 - g;
    key = (key << 3) | (uint64_t)(((q[0] >> bit) & 1u) << 2 | ((q[1] >> bit) & 1u) << 1 | ((q[2] >> bit) & 1u));
    key = (key << 1) | (uint64_t)((sz >> (14 - g)) & 1u);
  }
  keys[i] = key;
#else
  keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
#endif
  vals[i] = i;
}

// common-prefix length of sorted keys i and j (ties broken by the index), -1 outside the array
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012), section 3.
// Internal nodes 0..n-2, leaves are encoded as (n-1+k).  Requires n >= 2.
__global__ void __launch_bounds__(GB) k_radix_tree(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                                   uint32_t* __restrict__ parent, uint32_t* __restrict__ first, uint32_t* __restrict__ last) {
  const int i = blockIdx.x * GB + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const uint32_t lc = (lo == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
  const uint32_t rc = (hi == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  left[i] = lc; right[i] = rc;
  parent[lc] = (uint32_t)i; parent[rc] = (uint32_t)i;
  first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
  if (i == 0) parent[0] = 0xffffffffu;
}

// Hand-off between the chains of a bottom-up pass (k_fit, k_collapse_dp).  What one chain writes and another reads inside the
// same launch — a node's box, a node's sub-cost — is stored and loaded with AGENT-SCOPE relaxed atomics (global_store / global_load
// with sc1: write-through to the coherence point, never served from a CU's L1), the storing wave waits for its stores
// (s_waitcnt vmcnt(0)) and only then arrives at the parent's flag with an agent-scope atomic add; the chain whose add comes second
// reads after its add has returned.  No cache maintenance at all: round 2 ran a full __threadfence() — an L2 write-back plus an L1
// invalidate — three times per node, 220 of the 250 ms of a 10 M-triangle build (MI355X_MICROARCH.md, "inter-workgroup visibility":
// hand-off row 1, sc1 stores / sc1 loads, the last adder told by the value its add returned).  Everything else the passes read was
// written by an earlier kernel; everything else they write (cut[], cut_count[]) is read by a later one.
__device__ __forceinline__ void store_handoff(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_handoff(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void store_box(Box6* dst, const Box6& b) {
  float* d = reinterpret_cast<float*>(dst);
  for (int x = 0; x < 3; ++x) { store_handoff(d + x, b.lo[x]); store_handoff(d + 3 + x, b.hi[x]); }
}
__device__ __forceinline__ Box6 load_box(const Box6* src) {
  const float* p = reinterpret_cast<const float*>(src);
  Box6 b;
  for (int x = 0; x < 3; ++x) { b.lo[x] = load_handoff(p + x); b.hi[x] = load_handoff(p + 3 + x); }
  return b;
}

// the leaves' boxes: a triangle's box grown by the triangle test's own tolerance (bvh8.h: tri_box_inflation, from the scene's geometry bounds in cb[6..11])
__global__ void __launch_bounds__(GB) k_leaf_boxes(const Box6* __restrict__ pbox, const uint32_t* __restrict__ sorted, int n, Box6* __restrict__ nbox, const uint32_t* __restrict__ cb) {
  const int k = blockIdx.x * GB + threadIdx.x;
  if (k >= n) return;
  float glo[3], ghi[3];
  for (int a = 0; a < 3; ++a) { glo[a] = ord2f(cb[6 + a]); ghi[a] = ord2f(cb[9 + a]); }
  const float delta = tri_box_inflation(glo, ghi);
  Box6 b = pbox[sorted[k]];
  for (int a = 0; a < 3; ++a) { b.lo[a] -= delta; b.hi[a] += delta; }
  nbox[n - 1 + k] = b;
}
// boxes of the n-1 inner nodes (the leaves' are there: k_leaf_boxes); node ids as above.  flags[] must be zero.
__global__ void __launch_bounds__(GB) k_fit(int n, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                            const uint32_t* __restrict__ parent, uint32_t* __restrict__ flags, Box6* __restrict__ nbox) {
  const int k = blockIdx.x * GB + threadIdx.x;
  if (k >= n) return;
  uint32_t p = parent[(uint32_t)(n - 1 + k)];
  bool wrote = false;
  while (p != 0xffffffffu) {
    if (wrote) stores_done();
    if (atomicAdd(&flags[p], 1u) == 0u) return;  // first arrival: the sibling subtree is not finished yet
    // The loads below are relaxed, like the add: nothing in the memory model orders them behind it.  The hardware cannot run ahead of the
    // branch on the add's result (in-order issue); this keeps the COMPILER from hoisting them above the add.
    asm volatile("" ::: "memory");
    const Box6 a = load_box(&nbox[left[p]]), b = load_box(&nbox[right[p]]);
    Box6 m;
    for (int x = 0; x < 3; ++x) { m.lo[x] = fminf(a.lo[x], b.lo[x]); m.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
    store_box(&nbox[p], m);
    wrote = true;
    p = parent[p];
  }
}

struct Tree2 {
  const uint32_t* left; const uint32_t* right; const uint32_t* first; const uint32_t* last; const Box6* nbox; const uint32_t* sorted;
  int n;
};

__device__ __forceinline__ float area6(const Box6& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
// (an inner node of either binary tree has at least two leaves below it; only "is it a leaf" is ever asked)
__device__ __forceinline__ uint32_t leaf_count(const Tree2& T, uint32_t node) { return node >= (uint32_t)(T.n - 1) ? 1u : 2u; }
__device__ __forceinline__ uint32_t first_prim(const Tree2& T, uint32_t node) { return node - (uint32_t)(T.n - 1); }  // leaves only


// ---- which binary nodes become 8-wide nodes: SAH-optimal collapse on the boxes the traversal really tests ------------------
// The same optimisation as the host builder's (bvh_build.cpp, mode 2): a child's box is stored on its parent's 8-bit grid, i.e.
// inflated by up to two grid units of THAT parent, so the cost of a cut element depends on the 8-wide node m it hangs below:
//   sub(m) = min over cuts K of m's binary subtree, |K| <= 8, of  sum over c in K of  area(box(c) + 2 grid units of m) * (C_NODE |
//   C_TRI) + (sub(c) if c is inner)
// — a local dynamic programme over (descendant, slots), here restricted to cuts within 4 binary levels below m (30 heap
// positions x 7 slot counts).  Bottom-up like k_fit: chains start at the triangles and walk towards the root, the
// second arrival at a node solves it (all its descendants are done by then).  Measured on the host emulation of this tree (Morton
// splits, PHX_HOST_LBVH=1, Soup(1 M), random rays): node visits per ray 29.8 -> 27.8 against the greedy collapse.
// Round 3: the programme of one node is solved by a WAVE, not by a lane.  (One lane per chain kept its 30 x 7 table in LDS — 75 KB
// for 64 threads, two waves per CU, most of their lanes dead after the first arrival: 103 of the 160 ms of a 10 M-triangle build.)
// A wave owns 64 consecutive leaves of the Morton order and walks their chains one after the other; at a node it is the second to
// reach, its lanes fetch the 30 descendants within DP_LEVELS = 4 levels side by side (four dependent rounds of child indices, then
// every box in ONE load instruction), fill the table's leaf level, and solve the levels above it with one lane per (position, slot
// count) pair: 48 / 24 / 12 lanes.  The table is 1.2 KB per wave, so the CU runs its full 32.  Every sum and comparison is the
// one the serial version made, in the same order: sub[], cut[] and therefore the tree are unchanged.
#define DP_WAVES 4  /* waves per workgroup */
#ifndef DP_LEVELS
#define DP_LEVELS 4 /* binary levels below a node within which its cut is searched: 2^(DP_LEVELS+1) - 2 = 30 heap positions, one lane each.
                       5 levels (62 positions) lower the modelled cost (20 931 -> 20 727 at 10 M triangles) and RAISE k_trace's time on
                       every scene tried (+7 % on Soup 1 M, +4 % on the showroom: profiles/r03_zg_dp_levels.log) - deeper trees, one
                       more stack level in LDS - although on the host builder's SAH trees the same knob (PHX_DP_HEAP) helps by 1-3 % */
#endif
#define DP_NPOS (2u << DP_LEVELS) /* heap positions 2 .. DP_NPOS - 1 */
__global__ void __launch_bounds__(64 * DP_WAVES) k_collapse_dp(int n, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, const uint32_t* __restrict__ parent,
                                                             const Box6* __restrict__ nbox, uint32_t* __restrict__ flags, float* __restrict__ sub,
                                                             uint32_t* __restrict__ cut /* 8 per inner node */, uint8_t* __restrict__ cut_count, float CN, float CT) {
  static_assert(DP_NPOS <= 64, "one lane per heap position");
  __shared__ float s_best[DP_WAVES][DP_NPOS * 8];
  __shared__ uint8_t s_split[DP_WAVES][DP_NPOS * 8];
  __shared__ uint32_t s_node[DP_WAVES][DP_NPOS];
  const uint32_t wave = threadIdx.x >> 6, lane = __lane_id();
  float* best = s_best[wave]; uint8_t* split = s_split[wave]; uint32_t* node = s_node[wave];
  auto BEST = [&](uint32_t h, uint32_t j) -> float& { return best[h * 8u + j]; };
  auto SPLIT = [&](uint32_t h, uint32_t j) -> uint8_t& { return split[h * 8u + j]; };
  // LDS traffic between the lanes of ONE wave: the hardware keeps a wave's LDS operations in order; this keeps the compiler from
  // moving them across the step boundaries
  auto step = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  const uint32_t nleaf0 = (uint32_t)(n - 1);  // ids >= n - 1 are triangles
  const uint32_t k0 = (blockIdx.x * DP_WAVES + wave) * 64u;
  for (uint32_t k = k0; k < k0 + 64u && k < (uint32_t)n; ++k) {
    uint32_t m = parent[nleaf0 + k];
    bool wrote = false;
    while (m != 0xffffffffu) {
      if (wrote) stores_done();  // sub[] of the node this wave has just solved
      uint32_t arrived = 0;
      if (lane == 0) arrived = atomicAdd(&flags[m], 1u);
      arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
      if (arrived == 0u) break;  // first arrival: the sibling subtree is not solved yet
      // ---- the descendants of m within four levels, by heap position (2, 3 = m's children)
      if (lane < 2) node[2 + lane] = lane == 0 ? left[m] : right[m];
      step();
#pragma unroll
      for (uint32_t first = 4; first <= DP_NPOS / 2; first <<= 1) {
        if (lane < first) {
          const uint32_t h = first + lane, v = node[h >> 1];
          node[h] = (v != 0xffffffffu && v < nleaf0) ? ((h & 1u) ? right[v] : left[v]) : 0xffffffffu;
        }
        step();
      }
      // ---- one lane per position: its box on m's grid, its cost as ONE slot
      const Box6 mb = nbox[m];
      float g2[3];
      for (int a = 0; a < 3; ++a) {
        const double ext = (double)mb.hi[a] - (double)mb.lo[a];
        int e = -126;
        if (ext > 0.0) e = (int)ceil(log2(ext * 1.00001 / 255.0));
        e = max(-126, min(127, e));
        g2[a] = 2.0f * (float)ldexp(1.0, e);
      }
      if (lane >= 2 && lane < DP_NPOS) {
        const uint32_t h = lane, v = node[h];
        if (v != 0xffffffffu) {
          const Box6 b = nbox[v];
          const float dx = b.hi[0] - b.lo[0] + g2[0], dy = b.hi[1] - b.lo[1] + g2[1], dz = b.hi[2] - b.lo[2] + g2[2];
          const float aq = 2.0f * (dx * dy + dy * dz + dz * dx);
          // a triangle costs aq * CT in any number of slots; an inner node as ONE slot is an 8-wide node of its own; below the
          // fourth level nothing is split, so that is also its cost in more slots
          const float one = v >= nleaf0 ? aq * CT : aq * CN + load_handoff(&sub[v]);
          const uint32_t upto = (v >= nleaf0 || h >= DP_NPOS / 2) ? 7u : 1u;
          for (uint32_t j = 1; j <= upto; ++j) { BEST(h, j) = one; SPLIT(h, j) = 0; }
        }
      }
      step();
      // ---- the levels of the heap above the last, bottom-up: lane -> (position, slot count 2 .. 7), 64 pairs at a time
#pragma unroll
      for (uint32_t first = DP_NPOS / 4; first >= 2; first >>= 1) {
        for (uint32_t item = lane; item < first * 6u; item += 64u) {
          const uint32_t h = first + item / 6u, j = 2u + item % 6u, v = node[h];
          if (v != 0xffffffffu && v < nleaf0) {
            float r = BEST(h, 1); uint8_t sp = 0;
            for (uint32_t kk = 1; kk < j; ++kk) { const float c = BEST(2 * h, kk) + BEST(2 * h + 1, j - kk); if (c < r) { r = c; sp = (uint8_t)kk; } }
            BEST(h, j) = r; SPLIT(h, j) = sp;
          }
        }
        step();
      }
      // ---- m itself: all 8 slots, split between its two children; then the cut that achieves it (depth <= 4: at most 8 pending items)
      if (lane == 0) {
        float r = 3.0e38f; uint32_t bk = 1;
        for (uint32_t kk = 1; kk < 8; ++kk) { const float c = BEST(2, kk) + BEST(3, 8 - kk); if (c < r) { r = c; bk = kk; } }
        store_handoff(&sub[m], r);
        uint32_t sh[8], sj[8]; int top = 0; uint32_t cnt = 0;
        sh[top] = 3; sj[top++] = 8 - bk; sh[top] = 2; sj[top++] = bk;
        while (top > 0) {
          --top; const uint32_t h = sh[top], j = sj[top];
          const uint8_t sp = SPLIT(h, j);
          if (sp == 0) { cut[(size_t)m * 8 + cnt++] = node[h]; continue; }
          sh[top] = 2 * h + 1; sj[top++] = j - sp; sh[top] = 2 * h; sj[top++] = sp;
        }
        cut_count[m] = (uint8_t)cnt;
      }
      step();  // the table is rewritten by the next node
      wrote = true;
      m = parent[m];
    }
  }
}

#define BC_STRIDE 32  /* words between the builder's counters: one cache line each */
// One thread builds one Node8 from BVH2 subtree `qa[e]` into pool element `qb[e]`.
__global__ void __launch_bounds__(64) k_collapse(Tree2 T, SceneGrid grid, const float* __restrict__ abc, const uint32_t* __restrict__ prim_material, const uint32_t* __restrict__ qa,
                                                 const uint32_t* __restrict__ qb, uint32_t count, uint32_t* __restrict__ qa_out, uint32_t* __restrict__ qb_out,
                                                 uint32_t* __restrict__ counters /* x BC_STRIDE words: [0] pool elements, [1] tris, [2] out queue, [3] nodes */, PoolElem* __restrict__ pool,
                                                 const uint32_t* __restrict__ cut, const uint8_t* __restrict__ cut_count, uint32_t* __restrict__ elem_of_prim /* or nullptr */) {
  const uint32_t e = blockIdx.x * 64 + threadIdx.x;
  if (e >= count) return;
  const uint32_t root2 = qa[e], n8 = qb[e];
  uint32_t ch[8]; int nch = 0;
  if (leaf_count(T, root2) <= 1u) { ch[nch++] = root2; }  // degenerate: the whole tree is one triangle
  else if (cut) {  // the optimal cut of k_collapse_dp
    nch = (int)cut_count[root2];
    for (int i = 0; i < nch; ++i) ch[i] = cut[(size_t)root2 * 8 + i];
  } else {
    ch[nch++] = T.left[root2]; ch[nch++] = T.right[root2];
    while (nch < 8) {
      int pick = -1; float best = -1.0f;
      for (int i = 0; i < nch; ++i) {
        if (leaf_count(T, ch[i]) <= 1u) continue;
        const float a = area6(T.nbox[ch[i]]);
        if (a > best) { best = a; pick = i; }
      }
      if (pick < 0) break;
      const uint32_t c = ch[pick];
      ch[pick] = T.left[c]; ch[nch++] = T.right[c];
    }
  }
  const Box6 nb = T.nbox[root2];
  // octant-order slots: greedy minimum of dot(child centre - node centre, slot direction)
  int slot_of[8]; bool used[8] = {false, false, false, false, false, false, false, false}, done[8] = {false, false, false, false, false, false, false, false};
  float cen[8][3];
  for (int i = 0; i < nch; ++i) { const Box6 b = T.nbox[ch[i]]; for (int a = 0; a < 3; ++a) cen[i][a] = 0.5f * (b.lo[a] + b.hi[a]) - 0.5f * (nb.lo[a] + nb.hi[a]); }
  for (int k = 0; k < nch; ++k) {
    int bi = -1, bs = -1; float bc = FLT_MAX;
    for (int i = 0; i < nch; ++i) if (!done[i])
      for (int s = 0; s < 8; ++s) if (!used[s]) {
        const float v = cen[i][0] * ((s & 4) ? -1.0f : 1.0f) + cen[i][1] * ((s & 2) ? -1.0f : 1.0f) + cen[i][2] * ((s & 1) ? -1.0f : 1.0f);
        if (v < bc) { bc = v; bi = i; bs = s; }
      }
    slot_of[bi] = bs; used[bs] = true; done[bi] = true;
  }
  int child_in_slot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  for (int i = 0; i < nch; ++i) child_in_slot[slot_of[i]] = i;

  Node8 nd;
  nd.imask = 0;
  // origin on the scene grid, not above the node's lower corner; everything below is relative to the DECODED origin
  uint32_t gi[3]; float org[3];
  for (int a = 0; a < 3; ++a) { gi[a] = grid_index_below(nb.lo[a], grid.lo[a], grid.cell[a]); org[a] = fmaf((float)gi[a], grid.cell[a], grid.lo[a]); }
  double scale[3]; uint8_t eb[3];
  for (int a = 0; a < 3; ++a) {
    const double ext = (double)nb.hi[a] - (double)org[a];
    int ex = -126;
    if (ext > 0.0) {
      ex = (int)ceil(log2(ext * 1.00001 / 255.0));
      while (ldexp(255.0, ex) < ext * 1.00001) ++ex;
    }
    ex = max(-126, min(127, ex));
    eb[a] = (uint8_t)(ex + 127);
    scale[a] = ldexp(1.0, ex);
  }
  nd.ex = eb[0]; nd.ey = eb[1]; nd.ez = eb[2];
  uint32_t n_inner = 0, n_tri = 0, valid = 0;
  for (int s = 0; s < 8; ++s) {
    const int i = child_in_slot[s];
    uint8_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};
    if (i >= 0) {
      valid |= 1u << s;
      const Box6 b = T.nbox[ch[i]];
      for (int a = 0; a < 3; ++a) {
        double lo = floor(((double)b.lo[a] - (double)org[a]) / scale[a] - 1e-3);
        double hi = ceil(((double)b.hi[a] - (double)org[a]) / scale[a] + 1e-3);
        lo = fmax(0.0, fmin(255.0, lo)); hi = fmax(0.0, fmin(255.0, hi));
        ql[a] = (uint8_t)lo; qh[a] = (uint8_t)hi;
      }
      if (leaf_count(T, ch[i]) <= 1u) ++n_tri;
      else { nd.imask |= (uint8_t)(1u << s); ++n_inner; }
    }
    nd.qlox[s] = ql[0]; nd.qloy[s] = ql[1]; nd.qloz[s] = ql[2]; nd.qhix[s] = qh[0]; nd.qhiy[s] = qh[1]; nd.qhiz[s] = qh[2];
  }
  node_origin_encode(nd, gi[0], gi[1], gi[2], valid);
  // Pool elements and places in the next level's queue are handed out per WAVE: a scan of the lanes' needs (all children | inner
  // children << 16; a wave needs at most 512 of each) and one atomic per counter, each counter on its own 128-byte line — one
  // address sustains ~90 atomics per microsecond, and four per node on one line were a quarter of the 10 M-triangle build.  The
  // lanes beyond `count` have returned; they are the top lanes of the last wave, so every shuffle below reads a live lane.
  const uint32_t lane = threadIdx.x, own = (n_inner + n_tri) | (n_inner << 16);
  const int last_lane = (int)__popcll(__ballot(1)) - 1;
  uint32_t v = own;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, d); if (lane >= (uint32_t)d) v += t; }
  const uint32_t total = (uint32_t)__shfl((int)v, last_lane);
  uint32_t base_elems = 0, base_queue = 0;
  if (lane == 0) {
    const uint32_t all = total & 0xffffu, inner = total >> 16;
    base_elems = atomicAdd(&counters[0], all);
    if (inner) { base_queue = atomicAdd(&counters[2 * BC_STRIDE], inner); atomicAdd(&counters[3 * BC_STRIDE], inner); }
    if (all - inner) atomicAdd(&counters[1 * BC_STRIDE], all - inner);
  }
  base_elems = (uint32_t)__shfl((int)base_elems, 0); base_queue = (uint32_t)__shfl((int)base_queue, 0);
  nd.child_base = base_elems + ((v - own) & 0xffffu);
  const uint32_t qpos = base_queue + ((v - own) >> 16);
  // the children's pool elements, in slot order: triangle records now, nodelets by the next level's threads
  uint32_t rank = 0, r = 0;
  for (int s = 0; s < 8; ++s) {
    const int i = child_in_slot[s];
    if (i < 0) continue;
    const uint32_t ei = nd.child_base + rank++;
    if (nd.imask & (1u << s)) { qa_out[qpos + r] = ch[i]; qb_out[qpos + r] = ei; ++r; }
    else {
      const uint32_t p = T.sorted[first_prim(T, ch[i])];
      const float* t = abc + 9 * (size_t)p;
      TriRec R;
      R.v0x = t[0]; R.v0y = t[1]; R.v0z = t[2];
      R.e0x = t[3] - t[0]; R.e0y = t[4] - t[1]; R.e0z = t[5] - t[2];  // e0 = b - a, e1 = c - a (accel/triangle.hpp:48-50)
      R.e1x = t[6] - t[0]; R.e1y = t[7] - t[1]; R.e1z = t[8] - t[2];
      R.prim = p; R.material = prim_material[p]; R.pad1 = 0; R.pad2[0] = R.pad2[1] = R.pad2[2] = R.pad2[3] = 0;
      pool[ei].tri = R;
      if (elem_of_prim) elem_of_prim[p] = ei;  // the shade kernels find a smooth face's vertex normals by the HIT's pool index (device.cpp: elem_normals)
    }
  }
  pool[n8].node = nd;
}

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::snprintf(err, errlen, "%s: %s", #x, hipGetErrorString(e_)); cleanup(); return 1; } } while (0)

}  // namespace

int build_bvh8_gpu(hipStream_t stream, const float* d_abc, const uint32_t* d_prim_material, uint32_t n, GpuBvh* out, char* err, size_t errlen, uint32_t* d_elem_of_prim) {
  std::memset(out, 0, sizeof(*out)); out->depth = 1;
  void* bufs[24]; int nb = 0;
  PoolElem* pool = nullptr;  // the output: freed by cleanup() unless the build succeeds
  bool keep_output = false;
  auto cleanup = [&]() {
    for (int i = 0; i < nb; ++i) (void)hipFree(bufs[i]);
    nb = 0;
    if (!keep_output && pool) { (void)hipFree(pool); pool = nullptr; }
  };
  auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; bufs[nb++] = p; return p; };
  if (n < 2) { std::snprintf(err, errlen, "device builder needs at least 2 triangles"); return 1; }
  Box6* pbox = (Box6*)dalloc(sizeof(Box6) * n);
  uint32_t* cb = (uint32_t*)dalloc(64);
  uint64_t* keys = (uint64_t*)dalloc(8 * (size_t)n); uint64_t* keys2 = (uint64_t*)dalloc(8 * (size_t)n);
  uint32_t* vals = (uint32_t*)dalloc(4 * (size_t)n); uint32_t* sorted = (uint32_t*)dalloc(4 * (size_t)n);
  uint32_t* left = (uint32_t*)dalloc(4 * (size_t)n); uint32_t* right = (uint32_t*)dalloc(4 * (size_t)n);
  uint32_t* parent = (uint32_t*)dalloc(4 * 2 * (size_t)n); uint32_t* first = (uint32_t*)dalloc(4 * (size_t)n); uint32_t* last = (uint32_t*)dalloc(4 * (size_t)n);
  uint32_t* flags = (uint32_t*)dalloc(4 * (size_t)n);
  Box6* nbox = (Box6*)dalloc(sizeof(Box6) * 2 * (size_t)n);
  uint32_t* queues = (uint32_t*)dalloc(4 * 4 * (size_t)n);
  uint32_t* counters = (uint32_t*)dalloc(4 * BC_STRIDE * sizeof(uint32_t));
  if (!pbox || !cb || !keys || !keys2 || !vals || !sorted || !left || !right || !parent || !first || !last || !flags || !nbox || !queues || !counters) {
    std::snprintf(err, errlen, "hipMalloc failed in the device BVH builder"); cleanup(); return BVH_GPU_RECOVERABLE;
  }
  const uint32_t init_cb[12] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  HCHK(hipMemcpyAsync(cb, init_cb, sizeof(init_cb), hipMemcpyHostToDevice, stream));
  const dim3 g((n + GB - 1) / GB), b(GB);
  hipLaunchKernelGGL(k_prim_bounds, dim3(std::min<uint32_t>((n + GB - 1) / GB, 2048u)), b, 0, stream, d_abc, n, pbox, cb);
  hipLaunchKernelGGL(k_morton, g, b, 0, stream, pbox, n, cb, keys, vals);
  size_t temp_bytes = 0;
  HCHK(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys2, vals, sorted, (size_t)n, 0, 63, stream));
  void* temp = dalloc(temp_bytes);
  if (!temp) { std::snprintf(err, errlen, "hipMalloc failed (sort scratch)"); cleanup(); return BVH_GPU_RECOVERABLE; }
  HCHK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys2, vals, sorted, (size_t)n, 0, 63, stream));
  hipLaunchKernelGGL(k_leaf_boxes, g, b, 0, stream, pbox, sorted, (int)n, nbox, cb);
  // (The binary tree is Karras' radix tree over the extended Morton keys.  PLOC — bottom-up merging of the clusters whose union has the
  // smallest surface area, radius 16 — was built in its place and measured: lower modelled cost (-1 % soups, -9 % showroom) but 15 %
  // MORE node visits and 24 % more triangle tests per ray on the soups, k_trace +17 ... +35 %, +1 ... +2 % on the showroom: merged
  // boxes overlap, and a closest-hit traversal pays for overlap that a surface-area cost does not see.  profiles/r03_zh_ploc_probe.log;
  // the code is in the history at "PLOC (parallel locally-ordered clustering) ...".)
  {
    hipLaunchKernelGGL(k_radix_tree, g, b, 0, stream, keys2, (int)n, left, right, parent, first, last);
    HCHK(hipMemsetAsync(flags, 0, 4 * (size_t)n, stream));
    hipLaunchKernelGGL(k_fit, g, b, 0, stream, (int)n, left, right, parent, flags, nbox);
  }
  HCHK(hipGetLastError());
  // the scene grid of the nodelets' origins: from the bounds of the geometry
  uint32_t h_cb[12];
  HCHK(hipMemcpyAsync(h_cb, cb, sizeof(h_cb), hipMemcpyDeviceToHost, stream));
  HCHK(hipStreamSynchronize(stream));
  float glo[3], ghi[3];
  for (int a = 0; a < 3; ++a) { glo[a] = ord2f(h_cb[6 + a]); ghi[a] = ord2f(h_cb[9 + a]); }
  {  // the leaves' boxes were grown by delta (k_leaf_boxes): so is the grid that has to hold them
    const float delta = tri_box_inflation(glo, ghi);
    for (int a = 0; a < 3; ++a) { glo[a] -= delta; ghi[a] += delta; }
  }
  const SceneGrid grid = make_scene_grid(glo, ghi);

  // output pool: n triangle records + at most n - 1 nodelets (every nodelet has at least two children)
  if (hipMalloc((void**)&pool, sizeof(PoolElem) * 2 * (size_t)n) != hipSuccess) {
    pool = nullptr;
    std::snprintf(err, errlen, "hipMalloc failed (BVH8 pool)"); cleanup(); return BVH_GPU_RECOVERABLE;
  }
  uint32_t* qa[2] = {queues, queues + 2 * (size_t)n}; uint32_t* qb[2] = {queues + (size_t)n, queues + 3 * (size_t)n};
  const uint32_t zero_root[2] = {0u, 0u};
  HCHK(hipMemcpyAsync(qa[0], &zero_root[0], 4, hipMemcpyHostToDevice, stream));
  HCHK(hipMemcpyAsync(qb[0], &zero_root[1], 4, hipMemcpyHostToDevice, stream));
  uint32_t h_counters[4] = {1u, 0u, 0u, 1u};  // element 0 is the root nodelet
  uint32_t h_lines[4 * BC_STRIDE] = {0};
  for (int k = 0; k < 4; ++k) h_lines[k * BC_STRIDE] = h_counters[k];
  HCHK(hipMemcpyAsync(counters, h_lines, sizeof(h_lines), hipMemcpyHostToDevice, stream));
  Tree2 T{left, right, first, last, nbox, sorted, (int)n};
  // optimal collapse (PHX_LBVH_COLLAPSE=0: the greedy surface-area expansion)
  uint32_t* cut = nullptr; uint8_t* cut_count = nullptr; float* sub = nullptr;
  if (!(gpu_study_knob("PHX_LBVH_COLLAPSE") && atoi(gpu_study_knob("PHX_LBVH_COLLAPSE")) == 0)) {
    sub = (float*)dalloc(4 * (size_t)n);
    cut = (uint32_t*)dalloc(4 * 8 * (size_t)n); cut_count = (uint8_t*)dalloc((size_t)n);
    if (!sub || !cut || !cut_count) { std::snprintf(err, errlen, "hipMalloc failed (collapse tables)"); cleanup(); return BVH_GPU_RECOVERABLE; }
    HCHK(hipMemsetAsync(flags, 0, 4 * (size_t)n, stream));  // k_fit is done with its arrival flags
    const float cn = gpu_study_knob("PHX_CNODE") ? (float)atof(gpu_study_knob("PHX_CNODE")) : 1.6f;
    hipLaunchKernelGGL(k_collapse_dp, dim3((n + 64 * DP_WAVES - 1) / (64 * DP_WAVES)), dim3(64 * DP_WAVES), 0, stream, (int)n, left, right, parent, nbox, flags, sub, cut, cut_count, cn, 1.0f);
    HCHK(hipGetLastError());
  }
  uint32_t count = 1, depth = 0; int cur = 0;
  while (count > 0) {
    ++depth;
    hipLaunchKernelGGL(k_collapse, dim3((count + 63) / 64), dim3(64), 0, stream, T, grid, d_abc, d_prim_material, qa[cur], qb[cur], count, qa[cur ^ 1], qb[cur ^ 1],
                       counters, pool, cut, cut_count, d_elem_of_prim);
    HCHK(hipMemcpyAsync(h_lines, counters, sizeof(h_lines), hipMemcpyDeviceToHost, stream));
    HCHK(hipStreamSynchronize(stream));
    for (int k = 0; k < 4; ++k) h_counters[k] = h_lines[k * BC_STRIDE];
    count = h_counters[2];
    const uint32_t z = 0;
    HCHK(hipMemcpyAsync(counters + 2 * BC_STRIDE, &z, 4, hipMemcpyHostToDevice, stream));
    cur ^= 1;
    if (count > 0 && depth >= PHX_MAX_BVH_DEPTH) {
      std::snprintf(err, errlen, "tree too deep: more than %d levels (PHX_MAX_BVH_DEPTH, the traversal stack in LDS)", PHX_MAX_BVH_DEPTH);
      cleanup(); return BVH_GPU_RECOVERABLE;
    }
  }
  HCHK(hipStreamSynchronize(stream));
  if (h_counters[1] != n) { std::snprintf(err, errlen, "device BVH builder lost triangles (%u of %u)", h_counters[1], n); cleanup(); return 1; }
  float cost = 0.0f;
  if (sub) HCHK(hipMemcpy(&cost, sub, 4, hipMemcpyDeviceToHost));
  out->cost = cost;
  keep_output = true;
  out->pool = pool; out->num_elems = h_counters[0]; out->num_tris = h_counters[1]; out->num_nodes = h_counters[3]; out->depth = depth; out->grid = grid;
  cleanup();
  return 0;
}

}  // namespace phx
