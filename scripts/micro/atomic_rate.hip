// Micro-benchmark (round 5): how many RETURNING atomics per microsecond the chip sustains on one address, and on several addresses used
// at the same time by different workgroups — the pattern of the shade kernels' queue appends (one returning atomicAdd per workgroup and
// queue, between two barriers).  Resident grid (2 x 1024 threads per CU), every workgroup: K times { thread 0: atomicAdd on counter
// (f(blockIdx) * 32 words), barrier }.   make -C scripts/micro atomic_rate && scripts/micro/atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>

// mode 0: counter = 0; 1: blockIdx & (n - 1); 2: (blockIdx / n_per) % n  (neighbouring workgroups share a counter); 3: like 1, non-returning
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned* ctr, int iters, unsigned n, unsigned* sink) {
  __shared__ unsigned got;
  unsigned acc = 0;
  const unsigned per = gridDim.x / n;
  const unsigned c = MODE == 0 ? 0u : MODE == 2 ? (blockIdx.x / per) % n : blockIdx.x & (n - 1u);
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x == 0) {
      if (MODE == 3) { __hip_atomic_fetch_add(&ctr[c * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got = it; }
      else got = atomicAdd(&ctr[c * 32u], 1u);
    }
    __syncthreads();
    acc += got;
    __syncthreads();
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

template <int MODE>
static double run(unsigned* ctr, unsigned* sink, int blocks, int iters, unsigned n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipMemset(ctr, 0, 64 * 32 * 4);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, ctr, iters / 8, n, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, ctr, iters, n, sink);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)blocks * iters / (ms * 1e3);  // atomics per microsecond
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2, iters = 2000;
  unsigned *ctr, *sink; (void)hipMalloc((void**)&ctr, 64 * 32 * 4); (void)hipMalloc((void**)&sink, blocks * 4);
  std::printf("{\"device\": \"%s\", \"workgroups\": %d, \"unit\": \"returning atomics per microsecond, chip-wide (one per workgroup between two barriers)\",\n", p.gcnArchName, blocks);
  std::printf("  \"one counter\": %.1f,\n", run<0>(ctr, sink, blocks, iters, 1));
  for (unsigned n : {2u, 8u, 16u, 64u}) std::printf("  \"%u counters, workgroup id & %u\": %.1f,\n", n, n - 1, run<1>(ctr, sink, blocks, iters, n));
  for (unsigned n : {2u, 8u}) std::printf("  \"%u counters, neighbouring workgroups share one\": %.1f,\n", n, run<2>(ctr, sink, blocks, iters, n));
  std::printf("  \"8 counters, workgroup id & 7, NON-returning\": %.1f,\n", run<3>(ctr, sink, blocks, iters, 8));
  std::printf("  \"one counter, NON-returning\": %.1f}\n", run<3>(ctr, sink, blocks, iters, 1));
  return 0;
}
