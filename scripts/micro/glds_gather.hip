// Micro-benchmark (round 6, VERDICT r05 item 4 i): does fetching k_trace's 64-byte nodelets through LDS-DMA relieve the vector-memory
// RETURN path?  k_trace's captures show td_busy 0.97: every lane address brings 16 bytes back through the texture-data unit into VGPRs.
// `global_load_lds_dwordx4` takes a per-lane GLOBAL address like any gather, but its data lands in LDS (wave-uniform base + lane x 16 B,
// one KB per wave-instruction) instead of VGPRs; the lane then reads its four words back with ds_read_b128.
//   mode 0  the k_trace pattern: every "mem" lane loads the four 16-byte words of its own random node into VGPRs (4 x global_load_dwordx4)
//   mode 1  the same four per-lane addresses as 4 x global_load_lds_dwordx4 into the wave's 4 KB LDS slot, s_waitcnt vmcnt(0), 4 x ds_read_b128
//   mode 2  mode 1 without the read-back (what the DMA path alone sustains)
// Tables: 16 KB (L1-resident), 2.4 MB (inside one XCD's L2), 7.7 MB (the 100 k soup's pool), 77 MB (the 1 M soup's pool); p(mem) = 1 and 0.56
// (k_trace's node block: 54 of 64 lanes active, about two thirds of them fetching from memory rather than from the staged top of the tree).
// Output per line: ms, G nodes/s, lane addresses per clock and CU at 2.4 GHz (one per 16-byte word).
// Gate (VERDICT): build the node fetch on LDS-DMA only if mode 1 sustains >= 15 % more lane addresses than mode 0 on the 7.7 MB / 2.4 MB tables.
// Build: make -C scripts/micro glds_gather ; run: scripts/micro/glds_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int MODE>
__global__ void __launch_bounds__(256, 8) k(const uint4* __restrict__ nodes, uint32_t num_nodes, uint32_t pmem_256, int iters, uint32_t* out) {
  __shared__ uint4 slot[4][4][64];  // [wave of the workgroup][word][lane]: 4 KB per wave
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = mix(tid + 1u), acc = 0;
  for (int it = 0; it < iters; ++it) {
    s = mix(s + it);
    const uint32_t ni = s % num_nodes;
    const bool mem = ((s >> 20) & 255u) < pmem_256;
    const uint4* p = nodes + (size_t)ni * 4u;
    if (MODE == 0) {
      if (mem) {
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint4 v = p[w]; acc += v.x ^ v.y ^ v.z ^ v.w; }
      }
    } else {
      if (mem) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(p + w), (lds_ptr_t)&slot[wave][w][0], 16, 0, 0);  // lane l of the wave writes slot[wave][w][l]
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE == 1) {
        if (mem) {
#pragma unroll
          for (int w = 0; w < 4; ++w) { const uint4 v = slot[wave][w][lane]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is read before the next iteration's DMA overwrites it
      } else {
        acc += s;
      }
    }
  }
  if (MODE == 2) { const uint4 v = slot[wave][lane & 3u][lane]; acc += v.x; }
  out[tid] = acc;
}

int main() {
  const int iters = 2000;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = cus * 8;  // 8 workgroups of 256 per CU: 8 waves per SIMD, 128 KB of LDS slots per CU in modes 1 / 2
  uint32_t* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t num_nodes : {256u, 37500u, 120000u, 1200000u})
    for (uint32_t pm : {256u, 144u}) {
      std::vector<uint4> h((size_t)num_nodes * 4);
      for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((uint32_t)i, 1, 2, 3);
      uint4* d; hipMalloc(&d, h.size() * sizeof(uint4)); hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
      double ref_acc = 0;
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
        // modes 0 and 1 must have fetched the same data
        std::vector<uint32_t> ho(1024); hipMemcpy(ho.data(), out, 4096, hipMemcpyDeviceToHost);
        double sum = 0; for (uint32_t v : ho) sum += v;
        if (mode == 0) ref_acc = sum;
        const double fetched = (double)blocks * 256 * iters * (pm / 256.0);
        std::printf("%8u nodes (%7.2f MB)  p(mem) %.2f  mode %d (%s): %8.3f ms  %6.1f G nodes/s  %.2f lane addresses / clock / CU%s\n", num_nodes, num_nodes * 64e-6, pm / 256.0, mode,
                    mode == 0 ? "4 x global_load_dwordx4 -> VGPRs           " : mode == 1 ? "4 x global_load_lds_dwordx4 + 4 x ds_read_b128" : "4 x global_load_lds_dwordx4, no read-back     ",
                    best, fetched / best / 1e6, fetched * 4 / (best * 1e-3) / (cus * 2.4e9), mode == 1 ? (sum == ref_acc ? "  [same data as mode 0]" : "  [DATA DIFFERS FROM MODE 0]") : "");
      }
      hipFree(d);
    }
  hipFree(out);
  return 0;
}
