// Micro-benchmark, second sheet (round 5): issue cost of the gfx950 instructions a PACKED-fp16 formulation of k_trace's box test
// would be made of (two child planes per instruction), and of a few candidates for the byte -> number step.  Same harness as
// valu_ops.hip: a long unrolled stream of ONE instruction on 8 independent register chains, 8 waves per SIMD, no memory;
// clocks per wave-instruction per SIMD at the nominal 2.4 GHz.
// Build: make -C scripts/micro valu_ops2 ; run: scripts/micro/valu_ops2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP8(X) X X X X X X X X
#define KERNEL(NAME, ASM)                                                                                 \
  __global__ void __launch_bounds__(256, 8) NAME(int iters, unsigned* out, unsigned long long m0) {       \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 * 3, b1 = a0 * 5, b2 = a0 * 7, b3 = a0 * 11; \
    unsigned c0 = 0x3c003c00u + threadIdx.x, c1 = 0x40004000u;                                            \
    for (int it = 0; it < iters; ++it) {                                                                  \
      asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c0), "v"(c1), "s"(m0) : "vcc"); \
    }                                                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;                          \
  }
#define I8_3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
#define I8_2(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define I8_1(OP) OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"

KERNEL(k_pk_fma_f16, I8_3("v_pk_fma_f16"))
KERNEL(k_pk_mul_f16, I8_2("v_pk_mul_f16"))
KERNEL(k_pk_add_f16, I8_2("v_pk_add_f16"))
KERNEL(k_pk_max_f16, I8_2("v_pk_max_f16"))
KERNEL(k_pk_min_f16, I8_2("v_pk_min_f16"))
KERNEL(k_pk_max3_f16, I8_3("v_pk_maximum3_f16"))
KERNEL(k_pk_min3_f16, I8_3("v_pk_minimum3_f16"))
KERNEL(k_max3_f32_new, I8_3("v_maximum3_f32"))
KERNEL(k_cvt_pkrtz, I8_2("v_cvt_pkrtz_f16_f32"))
KERNEL(k_cvt_f32_f16, I8_1("v_cvt_f32_f16"))
KERNEL(k_med3, I8_3("v_med3_f32"))
KERNEL(k_max_i32, I8_2("v_max_i32"))
KERNEL(k_mad_u24, I8_3("v_mad_u32_u24"))
KERNEL(k_mul_u24, I8_2("v_mul_u32_u24"))
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0xfe\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xfe\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xfe\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0xfe\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0xfe\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0xfe\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0xfe\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0xfe\n")
KERNEL(k_pk_max_i16, I8_2("v_pk_max_i16"))
KERNEL(k_pk_mad_u16, I8_3("v_pk_mad_u16"))
KERNEL(k_pk_sub_i16, I8_2("v_pk_sub_i16"))
KERNEL(k_pk_lshr_b16, I8_2("v_pk_lshrrev_b16"))
KERNEL(k_add3, I8_3("v_add3_u32"))
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n")
KERNEL(k_fma_f32_sgpr, "v_fma_f32 %0, %0, s20, %9\n v_fma_f32 %1, %1, s20, %9\n v_fma_f32 %2, %2, s20, %9\n v_fma_f32 %3, %3, s20, %9\n v_fma_f32 %4, %4, s20, %9\n v_fma_f32 %5, %5, s20, %9\n v_fma_f32 %6, %6, s20, %9\n v_fma_f32 %7, %7, s20, %9\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")

// 64-bit destination chains with 32-bit sources: the packed fp8 -> 2 x f32 conversions
#define KERNEL64(NAME, ASM)                                                                               \
  __global__ void __launch_bounds__(256, 8) NAME(int iters, unsigned* out, unsigned long long m0) {       \
    unsigned long long a0 = threadIdx.x * 0x100000001ull + 0x3f8000003f800000ull, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 + 5, b1 = a0 + 7, b2 = a0 + 9, b3 = a0 + 11; \
    unsigned c0 = 0x38404448u + threadIdx.x, c1 = 0x3f800000u;                                            \
    for (int it = 0; it < iters; ++it) {                                                                  \
      asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c0), "v"(c1), "s"(m0) : "vcc"); \
    }                                                                                                     \
    const unsigned long long x = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;                                   \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)x ^ (unsigned)(x >> 32);                              \
  }
KERNEL64(k_cvt_pk_f32_fp8, "v_cvt_pk_f32_fp8 %0, %8\n v_cvt_pk_f32_fp8 %1, %8\n v_cvt_pk_f32_fp8 %2, %8\n v_cvt_pk_f32_fp8 %3, %8\n v_cvt_pk_f32_fp8 %4, %8\n v_cvt_pk_f32_fp8 %5, %8\n v_cvt_pk_f32_fp8 %6, %8\n v_cvt_pk_f32_fp8 %7, %8\n")
KERNEL64(k_cvt_pk_f32_bf8, "v_cvt_pk_f32_bf8 %0, %8\n v_cvt_pk_f32_bf8 %1, %8\n v_cvt_pk_f32_bf8 %2, %8\n v_cvt_pk_f32_bf8 %3, %8\n v_cvt_pk_f32_bf8 %4, %8\n v_cvt_pk_f32_bf8 %5, %8\n v_cvt_pk_f32_bf8 %6, %8\n v_cvt_pk_f32_bf8 %7, %8\n")
KERNEL64(k_cvt_scale_pk_f32_fp8, "v_cvt_scalef32_pk_f32_fp8 %0, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %1, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %2, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %3, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %4, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %5, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %6, %8, %9\n v_cvt_scalef32_pk_f32_fp8 %7, %8, %9\n")

template <typename K>
static double run(K kern, int blocks, int iters, unsigned* d_out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters / 4, d_out, 0x5555555555555555ull);
  (void)hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, d_out, 0x5555555555555555ull);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e-3;
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 2000;
  unsigned* d_out; (void)hipMalloc((void**)&d_out, (size_t)blocks * 256 * 4);
  const double clock = 2.4e9;
  auto report = [&](const char* name, double secs) {
    const double per_simd = 8.0 * iters * 64.0 * 8.0;
    std::printf("  \"%s\": %.3f,\n", name, secs * clock / per_simd);
  };
  std::printf("{\"unit\": \"clocks per wave-instruction per SIMD at 2.4 GHz nominal (8 waves per SIMD, independent chains)\",\n");
#define R(NAME, K) report(NAME, run(K, blocks, iters, d_out))
  R("v_pk_fma_f16 (2 FMAs)", k_pk_fma_f16); R("v_pk_mul_f16", k_pk_mul_f16); R("v_pk_add_f16", k_pk_add_f16);
  R("v_pk_max_f16", k_pk_max_f16); R("v_pk_min_f16", k_pk_min_f16); R("v_pk_maximum3_f16", k_pk_max3_f16); R("v_pk_minimum3_f16", k_pk_min3_f16);
  R("v_maximum3_f32", k_max3_f32_new); R("v_cvt_pkrtz_f16_f32", k_cvt_pkrtz); R("v_cvt_f32_f16", k_cvt_f32_f16); R("v_med3_f32", k_med3);
  R("v_max_i32", k_max_i32); R("v_mad_u32_u24", k_mad_u24); R("v_mul_u32_u24", k_mul_u24); R("v_bitop3_b32", k_bitop3);
  R("v_pk_max_i16", k_pk_max_i16); R("v_pk_mad_u16", k_pk_mad_u16); R("v_pk_sub_i16", k_pk_sub_i16); R("v_pk_lshrrev_b16", k_pk_lshr_b16);
  R("v_add3_u32", k_add3); R("v_lshl_add_u32", k_lshl_add); R("v_fma_f32 (one SGPR operand)", k_fma_f32_sgpr); R("v_mov_b32_dpp quad_perm", k_mov_dpp);
  R("v_cvt_pk_f32_fp8 (2 values)", k_cvt_pk_f32_fp8); R("v_cvt_pk_f32_bf8 (2 values)", k_cvt_pk_f32_bf8); R("v_cvt_scalef32_pk_f32_fp8 (2 values)", k_cvt_scale_pk_f32_fp8);
  std::printf("  \"device\": \"%s\"}\n", p.gcnArchName);
  (void)hipFree(d_out);
  return 0;
}
