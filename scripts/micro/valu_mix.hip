// Micro-benchmark behind the k_trace roofline: how fast can the chip run k_trace's own arithmetic —
// the 8-box node test (bvh8.h: node_hitmask, octant permutation from an LDS table like the kernel) and the reference's
// Moeller-Trumbore test (bvh8.h: mt_intersect) — when nothing else is in the way: operands in registers, all 64 lanes
// active, no global memory, 8 waves per SIMD.  The two rates are the VALU ceiling k_trace is priced against: a frame that
// needs V node visits and T triangle tests cannot finish faster than V / node_rate + T / tri_rate, however the rays are
// fed.  (k_trace is VALU-issue bound: profiles/README.md.)
// The JSON line carries the hash of the bvh8.h functions this binary was built from (scripts/src_hash.py, -DPHX_SRC_HASH by the Makefile:
// bench.py refuses a peak from other sources) and the shader clock DURING the node-test kernel: s_memtime ticks over the kernel against the
// constant-rate wall clock — meaningful only if s_memtime follows the shader clock on this part, so the capture script also takes
// GRBM_GUI_ACTIVE / 8 / kernel time with rocprofv3 (scripts/capture_valu_mix.sh) and that figure is the one recorded as clock_hz.
// Build: make -C scripts/micro valu_mix
// Run:   scripts/micro/valu_mix   -> one JSON line
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "bvh8.h"

using namespace phx;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: node tests, MODE 1: triangle tests, MODE 2 / 3: the loop skeleton of MODE 0 / 1 alone (what the perturbation of the operands costs)
#ifndef PHX_SRC_HASH
#define PHX_SRC_HASH "unknown"
#endif

template <int MODE>
__global__ void __launch_bounds__(256, 8) k(SceneGrid grid, int iters, uint32_t* out, unsigned long long* ticks) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  __shared__ uint8_t lut[2048];
  for (uint32_t i = threadIdx.x; i < 2048u; i += 256u) lut[i] = (uint8_t)perm_xor8(i & 0xffu, i >> 8);
  __syncthreads();
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = mix(tid + 1u), acc = 0;
  // a plausible ray and node / triangle per lane.  NOTHING of the test may be loop-invariant (the round-5 verdict found the compiler
  // hoisting 8 of the 48 conversions and ~26 of the 192 node-test instructions out of a loop that changed two words per iteration):
  // every one of the sixteen node words, the ray's origin and reciprocal direction and tbest change in every iteration — in MODE 2
  // (the skeleton that is subtracted) exactly as in MODE 0.  scripts/valu_mix_asm.py checks the compiled loop: 48 v_cvt_f32_ubyteN
  // and a VALU count (loop of MODE 0 minus loop of MODE 2) within 5 % of the node test's in kernels.s.
  const float fx = (float)(s & 1023u) * (1.0f / 1024.0f) - 0.5f, fy = (float)((s >> 10) & 1023u) * (1.0f / 1024.0f) - 0.5f;
  RayCtx r = make_ray_ctx(v3(0.1f * fx, 0.1f * fy, 0.0f), v3(fx, fy, -0.8f));
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = mix(s + 17u * i);
  w[2] = (w[2] & 0xff000000u) | 0x007a7a7au;  // scale exponents 2^-5
  TriRec T;  // a triangle per lane: nothing of it is wave-uniform (uniform operands would be computed on the scalar unit)
  T.v0x = fx; T.v0y = fy; T.v0z = -2.0f - 0.1f * fx; T.e0x = 0.3f + 0.01f * fy; T.e0y = 0.01f + 0.02f * fx; T.e0z = 0.02f - 0.01f * fy;
  T.e1x = 0.02f + 0.01f * fx; T.e1y = 0.3f - 0.02f * fy; T.e1z = 0.01f + 0.03f * fx; T.prim = tid;
  float tbest = 3.0e38f;
  // the perturbation of one iteration: one LCG step and two-clock VOP2 instructions (xor / add / mul by constants), identical in a MODE and its skeleton
  auto perturb = [&]() {
    s = s * 1664525u + 1013904223u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i == 2) w[2] ^= s & 0x01010101u;                // the low bit of the three scale exponents and of the inner mask
      else if (i == 3) w[3] += s >> 7;                    // child base
      else w[i] ^= s + 0x9e3779b9u * (uint32_t)(i + 1);   // plane bytes, grid origin
    }
    r.o.x += 1.0e-7f; r.o.y -= 1.0e-7f; r.o.z += 2.0e-7f;
    r.idx *= 1.0000001f; r.idy *= 0.9999999f; r.idz *= 1.0000001f;
    tbest *= 0.9999999f;
  };
  auto perturb_tri = [&]() {
    s = s * 1664525u + 1013904223u;
    const float e = 1.0e-6f * (float)(s >> 24);
    T.v0x += e; T.v0y -= e; T.v0z += 1.0e-7f; T.e0x *= 1.0000001f; T.e0y = -T.e0y; T.e0z *= 0.9999999f; T.e1x += 1.0e-8f; T.e1y *= 1.0000001f; T.e1z -= 1.0e-8f;
    r.o.x += 1.0e-7f; r.o.y -= 1.0e-7f; r.o.z += 2.0e-7f;
    r.d.x *= 1.0000001f; r.d.y *= 0.9999999f; r.d.z *= 1.0000001f;
  };
  // "consume" a register without an instruction: the skeletons keep every perturbed operand alive and add nothing of their own
#define PHX_KEEP_U(x) asm volatile("" : : "v"(x))
#define PHX_KEEP_F(x) asm volatile("" : : "v"(x))
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      perturb();
      const uint32_t hm = node_hitmask(w, grid, r, tbest, [&](uint32_t m) { return (uint32_t)lut[(r.oct_inv << 8) | m]; });
      acc += hm;
    } else if (MODE == 1) {
      perturb_tri();
      float us, vs, ds;
      if (mt_intersect(T, r.o, r.d, tbest, acc, us, vs, ds)) { acc += __float_as_uint(us) ^ __float_as_uint(vs); tbest = ds * 1.0000001f + 1.0f; }
    } else if (MODE == 2) {
      perturb();
#pragma unroll
      for (int i = 0; i < 16; ++i) PHX_KEEP_U(w[i]);
      PHX_KEEP_F(r.o.x); PHX_KEEP_F(r.o.y); PHX_KEEP_F(r.o.z); PHX_KEEP_F(r.idx); PHX_KEEP_F(r.idy); PHX_KEEP_F(r.idz); PHX_KEEP_F(tbest);
      acc += s;
    } else {  // MODE 3: the skeleton of the triangle loop
      perturb_tri();
      PHX_KEEP_F(T.v0x); PHX_KEEP_F(T.v0y); PHX_KEEP_F(T.v0z); PHX_KEEP_F(T.e0x); PHX_KEEP_F(T.e0y); PHX_KEEP_F(T.e0z); PHX_KEEP_F(T.e1x); PHX_KEEP_F(T.e1y); PHX_KEEP_F(T.e1z);
      PHX_KEEP_F(r.o.x); PHX_KEEP_F(r.o.y); PHX_KEEP_F(r.o.z); PHX_KEEP_F(r.d.x); PHX_KEEP_F(r.d.y); PHX_KEEP_F(r.d.z);
      acc += s;
    }
  }
  out[tid] = acc + __float_as_uint(tbest);
  if (tid == 0) { ticks[0] = clock64() - c0; ticks[1] = wall_clock64() - w0; }
}

template <int MODE>
static double run(int grid_blocks, int iters, uint32_t* d_out, unsigned long long* d_ticks) {
  SceneGrid g; for (int a = 0; a < 3; ++a) { g.lo[a] = -1.0f; g.cell[a] = 7.6e-6f; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid_blocks), dim3(256), 0, 0, g, iters / 8, d_out, d_ticks);  // warm-up
  hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid_blocks), dim3(256), 0, 0, g, iters, d_out, d_ticks);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e-3;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 20000;  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
  uint32_t* d_out; hipMalloc((void**)&d_out, (size_t)blocks * 256 * 4);
  unsigned long long* d_ticks; hipMalloc((void**)&d_ticks, 16);
  int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  const double t_skel = run<2>(blocks, iters, d_out, d_ticks), t_skel_tri = run<3>(blocks, iters, d_out, d_ticks), t_tri = run<1>(blocks, iters, d_out, d_ticks), t_node = run<0>(blocks, iters, d_out, d_ticks);
  unsigned long long h_ticks[2] = {0, 0}; hipMemcpy(h_ticks, d_ticks, 16, hipMemcpyDeviceToHost);  // of the last node-test launch
  const double memtime_hz = h_ticks[1] ? (double)h_ticks[0] / (double)h_ticks[1] * wall_khz * 1e3 : 0.0;
  const double lanes = (double)blocks * 256.0;
  // the skeleton (operand perturbation + loop) is subtracted: it is not part of the test being priced
  const double node_rate = lanes * iters / (t_node - t_skel), tri_rate = lanes * iters / (t_tri - t_skel_tri);
  std::printf("{\"device\": \"%s\", \"src_hash\": \"%s\", \"s_memtime_hz_during_node_kernel\": %.4e, \"wall_clock_khz\": %d, \"cus\": %d, \"waves_per_simd\": 8, \"iters\": %d, \"t_skeleton_s\": %.6f, \"t_skeleton_tri_s\": %.6f, \"t_node_s\": %.6f, \"t_tri_s\": %.6f, "
              "\"node_tests_per_s\": %.6e, \"tri_tests_per_s\": %.6e, \"ns_per_node_test_per_cu\": %.4f, \"ns_per_tri_test_per_cu\": %.4f}\n",
              p.gcnArchName, PHX_SRC_HASH, memtime_hz, wall_khz, cus, iters, t_skel, t_skel_tri, t_node, t_tri, node_rate, tri_rate, 1e9 * cus / node_rate, 1e9 * cus / tri_rate);
  hipFree(d_out); hipFree(d_ticks);
  return 0;
}
