// Micro-benchmark (round 4): does the vector-memory path charge a wave-instruction per occupied lane PAIR (lane i, lane i + 32)?
// If so, a lane whose pair partner is not loading can have the partner fetch half of its 64-byte node for free.
// Every wave gathers random 64-byte nodes for a random subset of its lanes ("mem" lanes, probability P_MEM):
//   mode 0  the k_trace pattern: every mem lane loads the four 16-byte words of its own node (4 instructions)
//   mode 1  helped: a mem lane whose partner (lane ^ 32) is not a mem lane loads words 0, 1 itself while the partner loads words 2, 3
//           of the same node (2 instructions); pairs of two mem lanes load 4 words each (instructions 3 and 4 carry only those)
//   mode 2  mode 1 + the exchange: the partner's 8 dwords are moved to the owner with ds_bpermute
// Output: ms and node fetches per second.  Build: hipcc -O3 --offload-arch=gfx950 -o pair_help scripts/micro/pair_help.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ nodes, uint32_t num_nodes, uint32_t pmem_256, int iters, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u, tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = mix(tid + 1u), acc = 0;
  for (int it = 0; it < iters; ++it) {
    s = mix(s + it);
    const uint32_t ni = s % num_nodes;
    const bool mem = ((s >> 20) & 255u) < pmem_256;
    const uint4* p = nodes + (size_t)ni * 4u;
    if (MODE == 0) {
      if (mem) {
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) { const uint4 v = p[k2]; acc += v.x ^ v.y ^ v.z ^ v.w; }
      }
    } else {
      const unsigned long long mm = __ballot(mem);
      const unsigned long long sw = (mm >> 32) | (mm << 32);           // bit i: my partner is a mem lane
      const bool partner_mem = (sw >> lane) & 1ull;
      const bool helped = mem && !partner_mem, helper = !mem && partner_mem, full = mem && partner_mem;
      const uint32_t pni = (uint32_t)__shfl_xor((int)ni, 32);
      const uint4* q = helper ? nodes + (size_t)pni * 4u + 2 : p;     // the helper fetches words 2, 3 of its partner's node
      uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, d = a;
      if (mem || helper) { a = q[0]; b = q[1]; }
      if (full) { c = p[2]; d = p[3]; }
      if (MODE == 2) {
        if (helped || helper) {  // pull the partner's eight dwords (the helper's result is garbage it never uses)
          c.x = __shfl_xor((int)a.x, 32); c.y = __shfl_xor((int)a.y, 32); c.z = __shfl_xor((int)a.z, 32); c.w = __shfl_xor((int)a.w, 32);
          d.x = __shfl_xor((int)b.x, 32); d.y = __shfl_xor((int)b.y, 32); d.z = __shfl_xor((int)b.z, 32); d.w = __shfl_xor((int)b.w, 32);
        }
      }
      if (mem) acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
      else acc += a.x ^ b.y;
    }
  }
  out[tid] = acc;
}

int main() {
  const int iters = 2000;
  for (uint32_t num_nodes : {120000u, 1200000u})
  for (uint32_t pm : {90u, 144u, 200u}) {   // probability of a mem lane x 256: 0.35, 0.56 (k_trace's node block), 0.78
    std::vector<uint4> h((size_t)num_nodes * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((uint32_t)i, 1, 2, 3);
    uint4* d; uint32_t* out;
    hipMalloc(&d, h.size() * sizeof(uint4)); hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    const int blocks = 256 * 8;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, pm, iters, out);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, pm, iters, out);
        else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, num_nodes, pm, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double fetched = (double)blocks * 256 * iters * pm / 256.0;
      printf("%8u nodes  p(mem) %.2f  mode %d (%s): %8.3f ms  %6.1f G nodes/s\n", num_nodes, pm / 256.0, mode, mode == 0 ? "own 4 words      " : mode == 1 ? "partner helps    " : "helps + bpermute ", best, fetched / best / 1e6);
    }
    hipFree(d); hipFree(out);
  }
  return 0;
}
