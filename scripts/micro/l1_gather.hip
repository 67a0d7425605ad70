// Micro-benchmark behind the k_trace node fetch: what does the CU's vector L1 charge for a
// wave64 gather of 80-byte nodelets?  A: every lane reads the five 16-B words of its own node (5 loads, each lane in a
// different 128-B line).  B: the four lanes of a quad read the first four words of ONE node together (one line per quad
// and instruction, four instructions serve the quad's four nodes) + each lane its own fifth word.
// Build: hipcc -O3 --offload-arch=gfx950 -o l1_gather l1_gather.hip ; run: ./l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ nodes, uint32_t num_nodes, uint32_t stride4, int iters, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u, tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = mix(tid + 1u), acc = 0;
  for (int it = 0; it < iters; ++it) {
    s = mix(s + it);
    const uint32_t ni = s % num_nodes;
    if (MODE == 0) {
      const uint4* p = nodes + (size_t)ni * stride4;
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) { const uint4 v = p[k2]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else if (MODE == 2) {  // half of the lanes idle (every other lane)
      if (lane & 1u) {
        const uint4* p = nodes + (size_t)ni * stride4;
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) { const uint4 v = p[k2]; acc += v.x ^ v.y ^ v.z ^ v.w; }
      }
    } else if (MODE == 3) {  // half of the lanes idle (lower half of the wave)
      if (lane & 32u) {
        const uint4* p = nodes + (size_t)ni * stride4;
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) { const uint4 v = p[k2]; acc += v.x ^ v.y ^ v.z ^ v.w; }
      }
    } else if (MODE == 5) {  // every lane of the wave reads the SAME node (per-wave random): a packet of coherent rays
      const uint32_t nu = __shfl(ni, 0);
      const uint4* p = nodes + (size_t)nu * stride4;
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) { const uint4 v = p[k2]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else if (MODE == 6) {  // groups of 8 neighbouring lanes read the same node
      const uint32_t nu = __shfl(ni, (int)(lane & ~7u));
      const uint4* p = nodes + (size_t)nu * stride4;
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) { const uint4 v = p[k2]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else if (MODE == 4) {  // five 4-byte loads instead of five 16-byte loads
      const uint32_t* p = reinterpret_cast<const uint32_t*>(nodes + (size_t)ni * stride4);
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) { acc += p[4 * k2]; }
    } else {
      const uint32_t i = lane & 3u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // node wanted by quad lane (i + j) & 3, fetched with a quad permute of the index register
        const uint32_t src = (lane & ~3u) | ((i + j) & 3u);
        const uint32_t nj = __shfl(ni, (int)src);
        const uint4 v = nodes[(size_t)nj * stride4 + i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
      const uint4 v = nodes[(size_t)ni * stride4 + 4];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  out[tid] = acc;
}

int main() {
  const int iters = 2000;
  for (uint32_t num_nodes : {200u, 30000u})
  for (uint32_t stride4 : {5u}) {
    std::vector<uint4> h((size_t)num_nodes * stride4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((uint32_t)i, 1, 2, 3);
    uint4* d; uint32_t* out;
    hipMalloc(&d, h.size() * sizeof(uint4)); hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    const int blocks = 256 * 8;  // 8 workgroups of 256 per CU: 32 waves per CU
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 7; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        else if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        else if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, stride4, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double nodes_fetched = (double)blocks * 256 * iters;
        if (rep) printf("%6u nodes stride %3u B  mode %s: %.3f ms, %.1f G nodes/s, %.2f TB/s of node bytes\n", num_nodes, stride4 * 16, (mode == 0 ? "per-lane   " : mode == 1 ? "quad-shared" : mode == 2 ? "odd lanes  " : mode == 3 ? "upper half " : mode == 4 ? "5 x dword  " : mode == 5 ? "same node  " : "same per 8 "), ms,
                        nodes_fetched / ms / 1e6, nodes_fetched * 80 / ms / 1e9);
      }
    }
    hipFree(d); hipFree(out);
  }
  return 0;
}
