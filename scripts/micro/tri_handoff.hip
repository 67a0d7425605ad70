// Micro-benchmark (round 6): k_trace's triangle block runs with ~20 of its 64 lanes active — every lane tests one of
// ITS OWN pending triangles per execution.  The closest hit with the lowest-primitive tie rule does not depend on which lane runs a test, so
// the wave's pending (ray, triangle) pairs could be dealt to all 64 lanes.  What does the hand-off cost?
//   mode 0  today's block: a lane with a pending triangle (probability p) runs the reference's Moeller-Trumbore on its own ray; operands in
//           registers (the triangle record's three loads are the same in both modes and left out of both)
//   mode 1  the hand-off block: every lane owns 0..4 pending pairs (mean set by p); wave-wide exclusive scan of the counts (DPP), owner table
//           through LDS + max-scan, the pair's index inside its owner's groups (4 ds_bpermute of group words + n-th set bit), the owner's
//           ray, tbest and best primitive to the executing lane (8 ds_bpermute), the test on all 64 lanes, the result back through an LDS
//           atomicMin on (t bits << 32 | primitive) per owner, the winner's (u, v, triangle) written to the owner's slot, the owner's read-back
// Output: ns per block and per PAIR tested, for p = 20/64 (mode 0, k_trace's measured lane count) and for mode 1 at full blocks.
// Build: make -C scripts/micro tri_handoff ; run: scripts/micro/tri_handoff
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bvh8.h"

using namespace phx;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
#define DPP(x, ctrl, rows) __builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, true)
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t x) {
  x += (uint32_t)DPP(x, 0x111, 0xf); x += (uint32_t)DPP(x, 0x112, 0xf); x += (uint32_t)DPP(x, 0x114, 0xf); x += (uint32_t)DPP(x, 0x118, 0xf);  // row_shr 1, 2, 4, 8
  x += (uint32_t)DPP(x, 0x142, 0xa); x += (uint32_t)DPP(x, 0x143, 0xc);                                                                        // row_bcast 15, 31
  return x;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
  x = max(x, (uint32_t)DPP(x, 0x111, 0xf)); x = max(x, (uint32_t)DPP(x, 0x112, 0xf)); x = max(x, (uint32_t)DPP(x, 0x114, 0xf)); x = max(x, (uint32_t)DPP(x, 0x118, 0xf));
  x = max(x, (uint32_t)DPP(x, 0x142, 0xa)); x = max(x, (uint32_t)DPP(x, 0x143, 0xc));
  return x;
}
__device__ __forceinline__ uint32_t bperm(uint32_t src_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
__device__ __forceinline__ float bpermf(uint32_t src_lane, float v) { return __uint_as_float(bperm(src_lane, __float_as_uint(v))); }

template <int MODE>
__global__ void __launch_bounds__(256, 8) k(int iters, uint32_t p_256, uint32_t* out, unsigned long long* pairs_out) {
  __shared__ uint32_t own_tab[4][64];
  __shared__ unsigned long long res[4][64];
  __shared__ float4 uvt[4][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = mix(tid + 1u), acc = 0;
  const float fx = (float)(s & 1023u) * (1.0f / 1024.0f) - 0.5f, fy = (float)((s >> 10) & 1023u) * (1.0f / 1024.0f) - 0.5f;
  RayCtx r = make_ray_ctx(v3(0.1f * fx, 0.1f * fy, 0.0f), v3(fx, fy, -0.8f));
  TriRec T;
  T.v0x = fx; T.v0y = fy; T.v0z = -2.0f - 0.1f * fx; T.e0x = 0.3f + 0.01f * fy; T.e0y = 0.01f + 0.02f * fx; T.e0z = 0.02f - 0.01f * fy;
  T.e1x = 0.02f + 0.01f * fx; T.e1y = 0.3f - 0.02f * fy; T.e1z = 0.01f + 0.03f * fx; T.prim = tid;
  float tbest = 3.0e38f, hu = 0.f, hv = 0.f;
  uint32_t hprim = 0, htri = 0xffffffffu;
  uint32_t tg = 0x3355u | (s & 0xff00u), tq = 0x1111u | ((s >> 8) & 0xff00u), tg_base = s >> 8, tq_base = s >> 9;  // group words as k_trace keeps them
  unsigned long long pairs = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const float e = 1.0e-6f * (float)(s >> 24);
    T.v0x += e; T.v0y -= e; T.e0y = -T.e0y; T.e1x += 1.0e-8f;   // the triangle of this iteration (stands for the record a lane has just loaded)
    const uint32_t h = mix(s ^ (lane * 0x9e3779b9u));
    if (MODE == 0) {
      const bool pend = (h & 255u) < p_256;
      pairs += (unsigned long long)__popcll(__ballot(pend));
      if (pend) {
        float us, vs, ds;
        if (mt_intersect(T, r.o, r.d, tbest, hprim, us, vs, ds)) { tbest = ds * 1.0000001f + 1.0f; hu = us; hv = vs; htri = s; hprim = T.prim; }
      }
    } else {
      // pending pairs of this lane: 0..4, mean 4 p (p_256 = 64 -> about one per lane, a full block)
      const uint32_t cnt = ((h & 255u) < p_256 ? 1u : 0u) + (((h >> 8) & 255u) < p_256 ? 1u : 0u) + (((h >> 16) & 255u) < p_256 ? 1u : 0u) + ((h >> 24) < p_256 ? 1u : 0u);
      const uint32_t incl = wave_incl_add(cnt), start = incl - cnt;
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      // who owns pair number `lane`?  owners mark their first pair, a max-scan carries the mark over the owner's other pairs
      own_tab[wave][lane] = 0u;
      res[wave][lane] = ((unsigned long long)__float_as_uint(tbest) << 32) | hprim;   // the owner's current best: (t, primitive) as one ordered key
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (cnt != 0u && start < 64u) own_tab[wave][start] = lane;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t owner = wave_incl_max(own_tab[wave][lane]);
      const bool valid = lane < min(total, 64u);
      pairs += (unsigned long long)min(total, 64u) * (lane == 0 ? 1ull : 0ull);
      // which of the owner's pending triangles: the j-th set bit of its pending masks (two groups)
      const uint32_t j = lane - bperm(owner, start);
      const uint32_t otg = bperm(owner, tg), otq = bperm(owner, tq), otgb = bperm(owner, tg_base), otqb = bperm(owner, tq_base);
      uint32_t m = ((otg >> 8) & 0xffu) | (((otq >> 8) & 0xffu) << 8);
      for (uint32_t q = 0; q < j && q < 4u; ++q) m &= m - 1u;
      const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
      const uint32_t ti = (bit < 8u ? otgb + (uint32_t)__popc(otg & 0xffu & ~(0xffffffffu << bit)) : otqb + (uint32_t)__popc(otq & 0xffu & ~(0xffffffffu << (bit - 8u))));
      // the owner's ray to the executing lane
      const v3 ro(bpermf(owner, r.o.x), bpermf(owner, r.o.y), bpermf(owner, r.o.z)), rd(bpermf(owner, r.d.x), bpermf(owner, r.d.y), bpermf(owner, r.d.z));
      const float rt = bpermf(owner, tbest);
      const uint32_t rp = bperm(owner, hprim);
      if (valid) {
        float us, vs, ds;
        if (mt_intersect(T, ro, rd, rt, rp, us, vs, ds)) {
          const unsigned long long key = ((unsigned long long)__float_as_uint(ds) << 32) | T.prim;
          atomicMin(&res[wave][owner], key);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (res[wave][owner] == key) uvt[wave][owner] = make_float4(us, vs, __uint_as_float(ti), 0.0f);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long best = res[wave][lane];
      if ((uint32_t)(best >> 32) != __float_as_uint(tbest) || (uint32_t)best != hprim) {
        const float4 w = uvt[wave][lane];
        tbest = __uint_as_float((uint32_t)(best >> 32)) * 1.0000001f + 1.0f; hprim = (uint32_t)best; hu = w.x; hv = w.y; htri = __float_as_uint(w.z);
      }
      tg ^= s & 0x0300u; tq ^= (s >> 3) & 0x0100u;
    }
  }
  out[tid] = acc + __float_as_uint(tbest) + __float_as_uint(hu) + __float_as_uint(hv) + htri + hprim;
  if (lane == 0) atomicAdd(pairs_out, pairs);
}

template <int MODE>
static void run(int blocks, int iters, uint32_t p_256, uint32_t* d_out, unsigned long long* d_pairs, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long pairs = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(d_pairs, 0, 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, rep ? iters : iters / 8, p_256, d_out, d_pairs);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) { best = ms; hipMemcpy(&pairs, d_pairs, 8, hipMemcpyDeviceToHost); }
  }
  const double waves = (double)blocks * 4, blocks_run = waves * iters;
  // MODE 0 counts the pending lanes of every wave (lane 0 adds the ballot's popcount); MODE 1 the pairs dealt per block
  const double pairs_per_block = (double)pairs / blocks_run;
  std::printf("mode %d  p %.3f  %-46s %8.3f ms   %6.1f pairs per block   %7.2f ns per block and CU   %6.3f ns per pair and CU\n", MODE, p_256 / 256.0, what, best,
              pairs_per_block, best * 1e6 / (blocks_run / 256.0 /* CUs */), best * 1e6 / (blocks_run / 256.0) / pairs_per_block);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8, iters = 4000;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  uint32_t* d_out; hipMalloc((void**)&d_out, (size_t)blocks * 256 * 4);
  unsigned long long* d_pairs; hipMalloc((void**)&d_pairs, 8);
  run<0>(blocks, iters, 80, d_out, d_pairs, "own-lane tests, 20 of 64 lanes (k_trace today)");
  run<0>(blocks, iters, 128, d_out, d_pairs, "own-lane tests, 32 of 64 lanes");
  run<0>(blocks, iters, 256, d_out, d_pairs, "own-lane tests, all 64 lanes");
  run<1>(blocks, iters, 80, d_out, d_pairs, "hand-off block, ~1.25 pairs per lane (full)");
  run<1>(blocks, iters, 56, d_out, d_pairs, "hand-off block, ~0.9 pairs per lane");
  run<1>(blocks, iters, 24, d_out, d_pairs, "hand-off block, ~0.4 pairs per lane");
  hipFree(d_out); hipFree(d_pairs);
  return 0;
}
