// Micro-benchmark (round 5): does it matter to HBM whether the workgroups that run at the same time stream through ONE contiguous part
// of their arrays or through EIGHT parts 1/8 of the array apart (workgroup b -> part b & 7, block b >> 3)?  Each workgroup reads 1 024
// 16-byte records of one array and writes 1 024 records to each of five others — the shape of k_shade's first launch.
// make -C scripts/micro stream_regions && scripts/micro/stream_regions
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>  // 0 linear; 1 eight interleaved parts; 2 eight parts one after the other
__global__ void __launch_bounds__(1024) k(const float4* in, float4* o0, float4* o1, float4* o2, float4* o3, float4* o4, unsigned n_blocks) {
  const unsigned per = n_blocks / 8u;
  const unsigned blk = MODE == 0 ? blockIdx.x : MODE == 1 ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
  const size_t i = (size_t)blk * 1024u + threadIdx.x;
  const float4 v = in[i];
  o0[i] = make_float4(v.x + 1, v.y, v.z, v.w); o1[i] = make_float4(v.y, v.x, v.z, v.w); o2[i] = make_float4(v.z, v.y, v.x, v.w);
  o3[i] = make_float4(v.w, v.y, v.z, v.x); o4[i] = make_float4(v.x, v.z, v.y, v.w);
}

template <int MODE>
static double run(const float4* in, float4** o, unsigned blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, in, o[0], o[1], o[2], o[3], o[4], blocks);
  (void)hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, in, o[0], o[1], o[2], o[3], o[4], blocks);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return (double)blocks * 1024.0 * 16.0 * 6.0 / (best * 1e-3) / 1e12;  // TB/s moved
}

int main() {
  const unsigned blocks = 230400;  // 236 M records: the bench frame's camera paths
  const size_t bytes = (size_t)blocks * 1024 * 16;
  float4* in; float4* o[5];
  (void)hipMalloc((void**)&in, bytes); (void)hipMemset(in, 0, bytes);
  for (int a = 0; a < 5; ++a) (void)hipMalloc((void**)&o[a], bytes);
  std::printf("{\"records\": %u, \"unit\": \"TB/s (1 array read + 5 written, 16 B records)\", \"linear\": %.3f, \"eight parts, workgroup id & 7\": %.3f}\n",
              blocks * 1024u, run<0>(in, o, blocks), run<1>(in, o, blocks));
  return 0;
}
