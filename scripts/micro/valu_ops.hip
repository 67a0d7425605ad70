// Micro-benchmark: issue cost of the VALU instructions k_trace's node test is made of (gfx950).  Every kernel runs a long
// unrolled stream of ONE instruction on 8 independent register chains, 8 waves per SIMD, no memory; the result is clocks per
// wave-instruction per SIMD (2.0 = the full rate of a wave64 on a SIMD-32).  It tells which formulations of the box test are
// cheap — instruction COUNT alone mis-predicted the fp16-plane experiment (bvh8.h) by 4 %.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_ops scripts/micro/valu_ops.hip ; run: ./valu_ops
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP8(X) X X X X X X X X
#define BODY(ASM)                                                                                         \
  for (int it = 0; it < iters; ++it) {                                                                    \
    asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c0), "v"(c1), "s"(m0) : "vcc"); \
  }

#define KERNEL(NAME, ASM)                                                                                 \
  __global__ void __launch_bounds__(256, 8) NAME(int iters, unsigned* out, unsigned long long m0) {       \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 * 3, b1 = a0 * 5, b2 = a0 * 7, b3 = a0 * 11; \
    unsigned c0 = 0x3f800123u + threadIdx.x, c1 = 0x40000321u;                                            \
    BODY(ASM)                                                                                             \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;                          \
  }

// each ASM string = 8 instructions on the 8 chains (operands %0..%7 chains, %8 %9 constants, %10 an SGPR pair)
KERNEL(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte2 %6, %6\n v_cvt_f32_ubyte3 %7, %7\n")
KERNEL(k_max3, "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
KERNEL(k_max, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[1,0,0]\n")
KERNEL(k_cmp_vcc_cndmask, "v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_le_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %9, %10\n v_cndmask_b32_e64 %1, %1, %9, %10\n v_cndmask_b32_e64 %2, %2, %9, %10\n v_cndmask_b32_e64 %3, %3, %9, %10\n v_cndmask_b32_e64 %4, %4, %9, %10\n v_cndmask_b32_e64 %5, %5, %9, %10\n v_cndmask_b32_e64 %6, %6, %9, %10\n v_cndmask_b32_e64 %7, %7, %9, %10\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n")
KERNEL(k_or3, "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %8, %9\n v_or3_b32 %5, %5, %8, %9\n v_or3_b32 %6, %6, %8, %9\n v_or3_b32 %7, %7, %8, %9\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8\n v_bfe_u32 %6, %6, 8, 8\n v_bfe_u32 %7, %7, 8, 8\n")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
KERNEL(k_cmp_sgpr, "v_cmp_le_f32_e64 s[20:21], %0, %8\n v_cmp_le_f32_e64 s[22:23], %1, %8\n v_cmp_le_f32_e64 s[24:25], %2, %8\n v_cmp_le_f32_e64 s[26:27], %3, %8\n v_cmp_le_f32_e64 s[20:21], %4, %8\n v_cmp_le_f32_e64 s[22:23], %5, %8\n v_cmp_le_f32_e64 s[24:25], %6, %8\n v_cmp_le_f32_e64 s[26:27], %7, %8\n")

KERNEL(k_or, "v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n")
KERNEL(k_or_sdwa, "v_or_b32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n")
KERNEL(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
KERNEL(k_sub_f32, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
KERNEL(k_min3, "v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n")
KERNEL(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n")
KERNEL(k_mul_sdwa, "v_mul_f32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n")
KERNEL(k_max_sdwa, "v_max_f32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_max_f32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n")

#define KERNEL64(NAME, ASM)                                                                               \
  __global__ void __launch_bounds__(256, 8) NAME(int iters, unsigned* out, unsigned long long m0) {       \
    unsigned long long a0 = threadIdx.x * 0x100000001ull + 0x3f8000003f800000ull, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 + 5, b1 = a0 + 7, b2 = a0 + 9, b3 = a0 + 11; \
    unsigned long long c0 = 0x3f8001233f800123ull, c1 = 0x4000032140000321ull;                              \
    for (int it = 0; it < iters; ++it) {                                                                  \
      asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c0), "v"(c1), "s"(m0) : "vcc"); \
    }                                                                                                     \
    const unsigned long long x = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;                                   \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)x ^ (unsigned)(x >> 32);                              \
  }
KERNEL64(k_pk_fma, "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
KERNEL64(k_pk_mul, "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
KERNEL64(k_pk_add, "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")

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
  // per SIMD: 8 waves x iters x (64 asm strings of 8 instructions)
  auto report = [&](const char* name, double secs, double asm_strings_per_iter) {
    const double per_simd = 8.0 * iters * asm_strings_per_iter * 8.0;
    std::printf("  \"%s\": %.3f,\n", name, secs * clock / per_simd);
  };
  std::printf("{\"unit\": \"clocks per wave-instruction per SIMD at 2.4 GHz nominal (8 waves per SIMD, independent chains)\",\n");
  report("v_fma_f32", run(k_fma, blocks, iters, d_out), 64);
  report("v_mul_f32", run(k_mul, blocks, iters, d_out), 64);
  report("v_cvt_f32_ubyteN", run(k_cvt_ubyte, blocks, iters, d_out), 64);
  report("v_max3_f32", run(k_max3, blocks, iters, d_out), 64);
  report("v_max_f32", run(k_max, blocks, iters, d_out), 64);
  report("v_perm_b32", run(k_perm, blocks, iters, d_out), 64);
  report("v_fma_mix_f32", run(k_fma_mix, blocks, iters, d_out), 64);
  report("v_cmp(vcc)+v_cndmask pair, per instruction", run(k_cmp_vcc_cndmask, blocks, iters, d_out), 64);
  report("v_cndmask_b32_e64 (SGPR mask)", run(k_cndmask_sgpr, blocks, iters, d_out), 64);
  report("v_cmp_le_f32_e64 (to SGPR)", run(k_cmp_sgpr, blocks, iters, d_out), 64);
  report("v_and_or_b32", run(k_and_or, blocks, iters, d_out), 64);
  report("v_or3_b32", run(k_or3, blocks, iters, d_out), 64);
  report("v_rcp_f32", run(k_rcp, blocks, iters, d_out), 64);
  report("v_bfe_u32", run(k_bfe, blocks, iters, d_out), 64);
  report("v_add_u32", run(k_add_u32, blocks, iters, d_out), 64);
  report("v_or_b32", run(k_or, blocks, iters, d_out), 64);
  report("v_or_b32_sdwa (src0 byte select)", run(k_or_sdwa, blocks, iters, d_out), 64);
  report("v_cvt_f32_u32_sdwa (byte select)", run(k_cvt_sdwa, blocks, iters, d_out), 64);
  report("v_fmac_f32 (VOP2)", run(k_fmac, blocks, iters, d_out), 64);
  report("v_sub_f32", run(k_sub_f32, blocks, iters, d_out), 64);
  report("v_min3_f32", run(k_min3, blocks, iters, d_out), 64);
  report("v_alignbit_b32", run(k_alignbit, blocks, iters, d_out), 64);
  report("v_and_b32", run(k_and, blocks, iters, d_out), 64);
  report("v_lshrrev_b32", run(k_lshr, blocks, iters, d_out), 64);
  report("v_lshl_or_b32", run(k_lshl_or, blocks, iters, d_out), 64);
  report("v_mul_f32_sdwa", run(k_mul_sdwa, blocks, iters, d_out), 64);
  report("v_max_f32_sdwa", run(k_max_sdwa, blocks, iters, d_out), 64);
  report("v_pk_fma_f32 (2 FMAs)", run(k_pk_fma, blocks, iters, d_out), 64);
  report("v_pk_mul_f32 (2 muls)", run(k_pk_mul, blocks, iters, d_out), 64);
  report("v_pk_add_f32 (2 adds)", run(k_pk_add, blocks, iters, d_out), 64);
  std::printf("  \"device\": \"%s\"}\n", p.gcnArchName);
  (void)hipFree(d_out);
  return 0;
}
