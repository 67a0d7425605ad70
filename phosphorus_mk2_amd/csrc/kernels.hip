// kernels.hip — hand-written gfx950 kernels of the wavefront path tracer (see kernels.h for the
// HBM layout).  One ray / path per lane, 64-lane wavefronts.
//
//   camera_ray  camera::perspective_kernel_t (reference src/kernels/cpu/camera.hpp:80-159) + spt::state_t::reset: a device
//               function — primary rays are rebuilt on the fly by k_trace_primary and the first k_shade of a pass, never stored
//   k_trace_primary  the same trace for the camera rays of a pass, the one coherent launch of a frame: a wave walks the tree once
//               for a packet of 64..256 rays (interval slab test per node, the reference's triangle test per ray)
//   k_trace     stream_mbvh_kernel_t::trace (src/kernels/cpu/stream_bvh_kernel.cpp:18-161) re-thought per lane:
//               BVH8 nodelets, (base,mask) group stack in LDS, closest-hit and any-hit rays in one persistent launch,
//               lanes refilled from chunks each wave pulls for itself
//   k_shade     deferred_shading_kernel_t (deferred_shading_kernel.hpp:20-72) + light_sampler_t (spt.hpp:95-149)
//               + integrator_t (spt.hpp:161-328) fused; survivors and shadow rays are appended to their
//               queues with __ballot/__popcll compaction (one global atomic per 512-thread workgroup)
//   k_film      the per-sample film add of tile_renderer_t::render_tile (src/xpu/cpu.cpp:175-198), in sample order
#include "kernels.h"

#include <algorithm>
#include <cstddef>
#include <cstdlib>

namespace phx {

#define PHX_BLOCK 256
#ifndef PHX_PERM_LUT
#define PHX_PERM_LUT 1  /* 1: the octant permutations of the hit masks come from a 2 KB table in LDS instead of 2 x 15 VALU instructions */
#endif
#ifndef PHX_COUNT
#define PHX_COUNT 0  /* 1: instrumented build that counts node visits and triangle tests (bench.py's device-layout byte model) */
#endif
#ifndef PHX_SHADE_TIMING
#define PHX_SHADE_TIMING 0  /* probe builds only: s_memtime around k_shade_g's sort phase and shading rounds, summed into DevStats fields the count build uses (block_append2 below tests it too) */
#endif
#ifndef PHX_SHADE_PREFETCH
#define PHX_SHADE_PREFETCH 2  /* k_shade_g: 1 = hit record and ray of the next round are requested before this round's append; 2 = and the path state (a queue record since round 6) right after it — the hit triangle's 16-byte shade record goes with stage 1, its index being in LDS (PHX_SHADE_TRI_LDS) — (35.9 / 35.3 / 34.8 ms for 0 / 1 / 2: profiles/r05_c_shade_prefetch_ab.log) */
#endif

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// Stream compaction: append-slot allocation for the threads of a workgroup that `want` one.
// __ballot/__popcll give the rank inside the wave, the waves' counts are summed through LDS, and ONE
// global atomic per workgroup reserves the slots (a returning atomic on one address sustains only
// ~90 ops/us chip-wide — per-wave atomics made k_shade atomic-bound).  All threads of the block must call.
#ifndef PHX_SHADE_BLOCK_D
#define PHX_SHADE_BLOCK_D 1024  /* Lambert-only k_shade: threads per workgroup.  The kernel is bound by the atomics on the two queue counters (one per
                                    workgroup and queue): 256 threads 28.3 ms, 512 16.0 ms, 1024 11.3 ms per frame (profiles/README.md) */
#endif
#ifndef PHX_SHADE_BLOCK_G
#define PHX_SHADE_BLOCK_G 512  /* general k_shade (measured on the 16-recipe stand-in: 256: +4 % shade time, 128: +29 %, 64: +144 % — one atomic per workgroup and append) */
#endif
#ifndef PHX_SHADE_WAVES_G
#define PHX_SHADE_WAVES_G 4    /* general k_shade: waves per SIMD the register allocator must leave room for */
#endif
// Both queues of a workgroup are appended in ONE round: the wave counts of the two predicates go to LDS, the first lanes of
// waves 0 and 1 each prefix one of them and issue its atomic — the two round trips to the counters overlap instead of following
// each other (two barriers and one atomic latency per workgroup instead of four and two).
template <int SHADE_BLOCK>
__device__ __forceinline__ void block_append2(bool want_a, uint32_t* counter_a, bool want_b, uint32_t* counter_b, uint32_t* lds /* [2 * (waves + 1)] */,
                                              uint32_t& at_a, uint32_t& at_b, unsigned long long* probe = nullptr /* PHX_SHADE_TIMING: ticks of the atomics' round trips, their number */) {
  static_assert(SHADE_BLOCK >= 128 && SHADE_BLOCK % 64 == 0, "block_append2: wave 0 sums queue A, wave 1 queue B");
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, nwaves = SHADE_BLOCK >> 6;
  const unsigned long long mask_a = __ballot(want_a), mask_b = __ballot(want_b);
  uint32_t* cnt_a = lds; uint32_t* cnt_b = lds + nwaves + 1;
  if (lane == 0) { cnt_a[wave] = (uint32_t)__popcll(mask_a); cnt_b[wave] = (uint32_t)__popcll(mask_b); }
  __syncthreads();
  if (lane == 0 && wave < 2) {
    uint32_t* cnt = wave == 0 ? cnt_a : cnt_b;
    uint32_t total = 0;
#pragma nounroll
    for (uint32_t w = 0; w < nwaves; ++w) { const uint32_t c = cnt[w]; cnt[w] = total; total += c; }
#if PHX_SHADE_TIMING
    const long long ta_ = clock64();
#endif
    cnt[nwaves] = total ? atomicAdd(wave == 0 ? counter_a : counter_b, total) : 0u;
#if PHX_SHADE_TIMING
    static_assert(offsetof(DevStats, tri_pending_lane_iters) == offsetof(DevStats, idle_lane_iters) + sizeof(unsigned long long), "probe[0] / probe[1] are DevStats::idle_lane_iters / tri_pending_lane_iters");
    if (probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); atomicAdd(&probe[0], (unsigned long long)(clock64() - ta_)); atomicAdd(&probe[1], 1ull); }
#endif
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  at_a = cnt_a[nwaves] + cnt_a[wave] + (uint32_t)__popcll(mask_a & below);
  at_b = cnt_b[nwaves] + cnt_b[wave] + (uint32_t)__popcll(mask_b & below);
}

// ---- camera rays ------------------------------------------------------------------------------------
// camera::perspective_kernel_t (kernels/cpu/camera.hpp:80-159).  Primary rays are never stored: the first
// k_trace of a pass and the first k_shade both rebuild the ray of path `path` from the pixel table and the jitter table
// (about 40 instructions) instead of writing and re-reading 32 B per path.
// LENS (camera_t::aperture_radius != 0, set by the Blender importer when depth of field is on): camera.hpp:140-147 and
// simd::concentric_sample_disc (math/simd/sampling.hpp:8-32) AS WRITTEN (SURVEY A-21): the raw samples in [0, 1) are mapped, not 2 u - 1;
// the constants named pi_o_2 / pi_o_4 hold 2 / pi and 4 / pi; select(m, l, r) returns r where m is set (float8.hpp:103-105).  The two lens
// samples are the spare dimensions 6 and 7 of the path's step 0 (the reference: a table of 1024 per sample index, sampling.cpp:108-109).
// A sample with a zero coordinate gives a non-finite angle and a NaN ray, which hits nothing, there and here.
template <bool LENS>
__device__ __forceinline__ void camera_ray(const DevScene& sc, const PassBuffers& pb, uint32_t path, uint32_t sample0, v3& p, v3& w) {
  const uint32_t pix = path / pb.num_samples, s = path - pix * pb.num_samples;  // pixel-major: a wave = 64 samples of one pixel
  const uint32_t xy = pb.pix_xy[pix];
  const float sx = (float)(xy & 0xffffu), sy = (float)(xy >> 16);
  const float2 jit = pb.jitter[sample0 + s];
  const float ndcy = 0.5f - (-0.5f + sy) * sc.stepy;
  const float ndcx = (-0.5f + sx) * sc.stepx - 0.5f;
  v3 d(jit.x, jit.y, -1.0f);
  d.x = (ndcx + d.x * sc.stepx) * sc.ratio * sc.zoom;
  d.y = (ndcy + d.y * sc.stepy) * sc.zoom;
  const float ool = 1.0f / sqrtf(sdot(d, d));
  d = v3(d.x * ool, d.y * ool, d.z * ool);
  const float* M = sc.cam_m;
  if constexpr (LENS) {
    const uint32_t key = path_key(pb.seed, (xy >> 16) * sc.width + (xy & 0xffffu), sample0 + s);
    const float ux = draw_f32(key, DIM_LENS_U), uy = draw_f32(key, DIM_LENS_V);
    const float c2 = (float)(2.0f / M_PI), c4 = (float)(4.0f / M_PI);
    const bool x_gt_y = fabsf(ux) > fabsf(uy);
    const float r = x_gt_y ? uy : ux;
    const float theta1 = c4 * (uy / ux), theta2 = c2 - c4 * (ux / uy);
    const sincosf_t sc_t = sincosf_(x_gt_y ? theta2 : theta1);
    const float lx = (r * sc_t.c) * sc.aperture_radius, ly = (r * sc_t.s) * sc.aperture_radius;
    const float ft = fabsf(sc.focal_distance / d.z);
    d = v3(d.x * ft - lx, d.y * ft - ly, d.z * ft - 0.0f);
    const float ool2 = 1.0f / sqrtf(sdot(d, d));
    d = v3(d.x * ool2, d.y * ool2, d.z * ool2);
    { float t = lx * M[0]; t = fmaf(ly, M[4], t); t = fmaf(0.0f, M[8], t); p.x = t + M[12]; }
    { float t = lx * M[1]; t = fmaf(ly, M[5], t); t = fmaf(0.0f, M[9], t); p.y = t + M[13]; }
    { float t = lx * M[2]; t = fmaf(ly, M[6], t); t = fmaf(0.0f, M[10], t); p.z = t + M[14]; }
  } else {
  { float t = 0.0f * M[0]; t = fmaf(0.0f, M[4], t); t = fmaf(0.0f, M[8], t); p.x = t + M[12]; }
  { float t = 0.0f * M[1]; t = fmaf(0.0f, M[5], t); t = fmaf(0.0f, M[9], t); p.y = t + M[13]; }
  { float t = 0.0f * M[2]; t = fmaf(0.0f, M[6], t); t = fmaf(0.0f, M[10], t); p.z = t + M[14]; }
  }
  { float t = d.x * M[0]; t = fmaf(d.y, M[4], t); w.x = fmaf(d.z, M[8], t); }
  { float t = d.x * M[1]; t = fmaf(d.y, M[5], t); w.y = fmaf(d.z, M[9], t); }
  { float t = d.x * M[2]; t = fmaf(d.y, M[6], t); w.z = fmaf(d.z, M[10], t); }
}

__device__ __forceinline__ void zero_cursors(uint32_t* counters) {
#pragma unroll
  for (uint32_t k = 0; k < 2u * CNT_SEGS; ++k) counters[CNT_CURSOR + k * CNT_STRIDE] = 0;
}
// start of a pass: queue 0 "holds" the num_pixels x num_samples camera rays (state_t::reset, spt.hpp:39-49, is implicit:
// the first k_shade writes beta, depth and radiance of every path without reading them)
__global__ void k_begin_pass(PassBuffers pb, uint32_t num_samples) {
  const uint32_t npaths = pb.num_pixels * num_samples;
  pb.counters[0] = npaths; pb.counters[CNT_STRIDE] = 0; pb.counters[CNT_SHADOW] = 0; pb.counters[CNT_SHADOW + CNT_STRIDE] = 0;
  zero_cursors(pb.counters);
  atomicAdd(&pb.stats->camera_samples, (unsigned long long)npaths);
}

// ---- trace ----------------------------------------------------------------------------------------
template <int LEVELS>
struct LdsStack {
  uint2* base;  // &lds[threadIdx.x]; entry k lives at base[k * PHX_BLOCK]: conflict-free 8-byte accesses
  int sp;
  __device__ __forceinline__ void push(uint32_t b, uint32_t h) { base[sp * PHX_BLOCK] = make_uint2(b, h); ++sp; }
  __device__ __forceinline__ void pop(uint32_t& b, uint32_t& h) { --sp; const uint2 e = base[sp * PHX_BLOCK]; b = e.x; h = e.y; }
  __device__ __forceinline__ bool empty() const { return sp == 0; }
};

// ---- wave-level streaming traversal -------------------------------------------------------------------
// A wave64 that runs one ray per lane to completion wastes most of its lanes: ray lengths differ widely
// (measured: ~20 % VALU lane utilisation with the plain per-lane loop).  Here every lane is a small state
// machine and the wave iterates "one node visit, one triangle test, pop/finish" with all lanes in whatever
// state they are in; a lane whose ray has finished is REFILLED from the workgroup's range of the queue as
// soon as REFILL_MIN lanes are idle (cursor in LDS: one ds_add per refill, no global atomics).
// Visiting order, box test and triangle test are those of traverse8 (bvh8.h), so results are identical.
// One stream serves BOTH queues: lanes are refilled from the workgroup's shadow-ray range first, then from
// its closest-hit range; the any-hit / closest-hit distinction is a per-lane flag, so a launch has a single
// drain phase (the tail where rays run out and lanes idle) instead of one per queue.
#ifndef PHX_STEPS_PER_REFILL
#define PHX_STEPS_PER_REFILL 1
#endif
// Sensitivity probes (profiles/README.md), never in the product build: extra FMAs / extra 16-byte loads per node visit.
// The chunks of the persistent launch are handed out in two levels.  A WORKGROUP takes 16 chunks' worth of
// consecutive rays from the global cursor at a time; its waves take their chunks from that range through a 64-bit word in LDS
// (next | end << 32: one ds_add returns a consistent pair).  The wave that finds the range used up fetches the next one
// (try-lock in LDS, re-check under the lock); waves that lose the race go on traversing and ask again at their next refill.
// Sixteen waves of one CU then work on neighbouring pixels, and the global cursor takes 1/16 of the atomics.
// (Round 2 also carried a static XCD-aware split of the queues and per-wave global chunks; both lost to this scheme on every
// workload and were removed in round 3 — the history is in EXPERIMENTS.md, Part B section 3.)
#ifndef PHX_WG_CHUNKS
#define PHX_WG_CHUNKS 16u  /* chunks in a workgroup's range (fewer when the queue is too short to give every workgroup four ranges) */
#endif
#ifndef PHX_XCD_SEGMENTS
#define PHX_XCD_SEGMENTS 1  /* a workgroup takes its ranges from its XCD's eighth of the queue first */
#endif
#ifndef PHX_SPILL_FROM_LEVELS
#define PHX_SPILL_FROM_LEVELS 10u  /* trees with this many stack levels or more keep only PHX_SPILL_LDS_LEVELS of them in LDS */
#endif
#ifndef PHX_SPILL_LDS_LEVELS
#define PHX_SPILL_LDS_LEVELS 7u
#endif
#ifndef PHX_STACK_PACKED
// Deep trees (the SPILL plan: >= 10 stack levels, 7 of them in LDS) keep 5-byte stack entries in LDS — (child base : 24 | pending inner hits : 8)
// as a dword + the node's valid mask as a byte, four levels of a lane sharing one dword — instead of 8-byte ones; the 21 KB this frees per
// workgroup stage 256 more nodelets (281 -> 537).  Measured (profiles/r06_f_packed_stack_ab.log): k_trace -2.3 % on BASELINE config 4 (10 M
// triangles), -4.0 % on the closed showroom (depth 13); trees whose stacks fit LDS gain nothing from more staged nodelets (100 k: +0.7 %, the
// second LDS access per push / pop; 1 M: +-0) and keep 8-byte entries.  Pools of < 2^24 elements only (1 GB; larger ones: 8-byte entries).
// 0 = never, 1 = the SPILL plan, 2 = every plan (A/B).
#define PHX_STACK_PACKED 1
#endif
#ifndef PHX_PROBE_VALU
#define PHX_PROBE_VALU 0
#endif
#ifndef PHX_PROBE_VMEM
#define PHX_PROBE_VMEM 0
#endif
#ifndef PHX_PROBE_VMEM_DWORD
#define PHX_PROBE_VMEM_DWORD 0
#endif
#define PHX_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))  /* wave-uniform by construction: keep it in an SGPR */
#ifndef PHX_TRACE_WATCHDOG
#define PHX_TRACE_WATCHDOG (1u << 22)  /* loop iterations after which a k_trace wave exits with DevStats::watchdog set (a launch needs ~1e4): every wave reaches its exit */
#endif
struct DynQueue {            // the launch is persistent and every WAVE pulls chunks of both queues on its own
  uint32_t n0, n1, c0, c1;   // queue lengths and chunk sizes (0 shadow, 1 closest), wave-uniform
  uint32_t r0, r1;           // rays in a workgroup's range
  uint32_t num_waves;
  uint32_t* cursor;          // pb.counters + CNT_CURSOR: two global cursors, zeroed by the kernel that filled the queues
};
template <int BLOCK, bool SPILL /* the stack's deep levels live in HBM */, bool PACKED /* 5-byte stack entries in LDS */>
__device__ __forceinline__ void trace_stream(const DevScene& sc, const PassBuffers& pb, int q,
                                             uint32_t* cursor /* LDS: the workgroup's ranges */, uint2* stack_base, uint8_t* stack_valid /* PHX_STACK_PACKED: this thread's dword of the valid-mask rows */, uint32_t lds_levels,
                                             uint2* spill_base /* this thread's column of sc.stack_spill */, uint32_t refill_min,
                                             const uint4* __restrict__ top /* nodelets staged in LDS */, uint32_t ntop,
                                             const DynQueue dq, const uint8_t* __restrict__ perm_lut) {
  const uint32_t lane = __lane_id();
  bool active = false, any = false;
  uint32_t phase = 0;  // wave-uniform: 0 = shadow range, 1 = closest range, 2 = drained
  RayCtx r; r.o = v3(0.f); r.d = v3(0.f); r.idx = r.idy = r.idz = 0.f; r.oct_inv = 0;
  float tbest = 0.f, hu = 0.f, hv = 0.f;
  uint32_t htri = 0xffffffffu, hprim = 0, idx = 0, ng_base = 0, ng_hits = 0, path = 0;
  // triangle group of the lane: tg_base = pool index of the first child of the node the triangles hang below; tg = that node's
  // valid mask (bits 0..7) | hit leaf slots still to test (8..15); (tq_base, tq) is a second such group, waiting
  constexpr uint32_t TG_PENDING = 0xff00u;
  uint32_t tg_base = 0, tg = 0, tq_base = 0, tq = 0;  // tq: the second group (valid | pending << 8), 0 = none
  int sp = 0;
  const float4* __restrict__ ro = pb.ro[q];
  const float4* __restrict__ rd = pb.rd[q];
  uint32_t dlo = 0, dhi = 0;  // the wave's current chunk [dlo, dhi)
#if PHX_COUNT
  uint32_t cnt_lds[2] = {0, 0}, cnt_mem[2] = {0, 0}, cnt_tri[2] = {0, 0};  // instrumented build: this lane's traversal work
  uint32_t cnt_iter = 0, cnt_nb = 0, cnt_tb = 0, cnt_refill = 0;              // ... and the wave's (wave-uniform)
  uint32_t cnt_idle = 0, cnt_pend = 0;
  uint32_t cnt_push[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // pending (ray, triangle) pairs of the WAVE when its triangle block runs — each lane tests one of its own per execution; a block that
  // handed pairs to idle lanes could test up to 64 (profiles/r06_tri_handoff.md) — summed, and as a histogram
  uint32_t cnt_pairs = 0, cnt_pairs_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  uint32_t guard = 0;  // wave-uniform (an SGPR): iterations of this wave
  for (;;) {
    if (++guard > PHX_TRACE_WATCHDOG) { if (lane == 0) atomicAdd(&pb.stats->watchdog, 1ull); break; }
    // ---- refill idle lanes from the workgroup's cursors
    const unsigned long long idle = __ballot(!active);
    if (phase < 2u && (uint32_t)__popcll(idle) >= refill_min) {
#if PHX_COUNT
      ++cnt_refill;
#endif
      const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
      uint32_t hi, base;
      {
        if (dlo >= dhi) {  // chunk used up: take the next one out of the workgroup's range
          const uint32_t c = phase == 0u ? dq.c0 : dq.c1, qn = phase == 0u ? dq.n0 : dq.n1;
          uint32_t got_lo = 0, got_hi = 0, st = 1;  // st: 0 a chunk, 1 ask again later, 2 this queue is exhausted
          if (lane == leader) {
            unsigned long long* pack = reinterpret_cast<unsigned long long*>(cursor) + phase;
            uint32_t* fetching = cursor + 4 + phase; uint32_t* done = cursor + 6 + phase;
            const unsigned long long old = __hip_atomic_fetch_add(pack, (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t s0 = (uint32_t)old, e0 = (uint32_t)(old >> 32);
            if (s0 < e0) { got_lo = s0; got_hi = min(s0 + c, e0); st = 0; }
            else if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) st = 2;
            else if (atomicCAS(fetching, 0u, 1u) == 0u) {
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
              const unsigned long long cur = __hip_atomic_load(pack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if ((uint32_t)cur < (uint32_t)(cur >> 32)) st = 1;  // another wave refilled the range in the meantime
              else if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) st = 2;
              else {
                const uint32_t cwg = phase == 0u ? dq.r0 : dq.r1;
#if PHX_XCD_SEGMENTS
                // the next range of this XCD's segment of the queue; when that is used up, of the next segment that has one (cursor[8 + phase])
                uint32_t nb = qn, seg_end = qn, sgm = cursor[8 + phase];
                for (uint32_t tries = 0; tries < CNT_SEGS; ++tries, sgm = (sgm + 1u) & (CNT_SEGS - 1u)) {
                  const uint32_t lo_s = (uint32_t)((unsigned long long)qn * sgm / CNT_SEGS), hi_s = (uint32_t)((unsigned long long)qn * (sgm + 1u) / CNT_SEGS);
                  const uint32_t homes = (gridDim.x + CNT_SEGS - 1u - sgm) / CNT_SEGS;  // workgroups whose first range lies in this segment
                  const unsigned long long at = (unsigned long long)lo_s + (unsigned long long)homes * cwg + atomicAdd(&dq.cursor[(phase * CNT_SEGS + sgm) * CNT_STRIDE], cwg);
                  if (at < hi_s) { nb = (uint32_t)at; seg_end = hi_s; break; }
                }
                cursor[8 + phase] = sgm;
                if (nb >= seg_end) { __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); st = 2; }
                else {
                  const uint32_t end = min(nb + cwg, seg_end);
#else
                const uint32_t nb = atomicAdd(&dq.cursor[phase * CNT_SEGS * CNT_STRIDE], cwg) + gridDim.x * cwg;
                if (nb >= qn) { __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); st = 2; }
                else {
                  const uint32_t end = min(nb + cwg, qn);
#endif
                  got_lo = nb; got_hi = min(nb + c, end); st = 0;  // the fetching wave keeps the range's first chunk
                  __hip_atomic_store(pack, (unsigned long long)(nb + c) | ((unsigned long long)end << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              __hip_atomic_store(fetching, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          st = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)st, (int)leader));
          if (st == 2u) {  // this queue is exhausted
            phase = PHX_UNI(phase + 1u);
            if (phase < 2u || __ballot(active) != 0ull) { if (phase < 2u) continue; } else break;
          } else if (st == 0u) {
            dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)got_lo, (int)leader));
            dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)got_hi, (int)leader));
          } else if (__ballot(active) == 0ull) { __builtin_amdgcn_s_sleep(4); continue; }  // nothing to do but wait for the fetching wave (the watchdog counts these sleeps too: a separate counter for them cost 2 % of the kernel, profiles/r04_t_spin_ab.log)
        }
        if (phase >= 2u) { hi = 0; base = 0; }
        else {
          const uint32_t take = PHX_UNI(min((uint32_t)__popcll(idle), dhi - dlo));
          base = dlo; hi = dlo + take; dlo = PHX_UNI(dlo + take);
        }
      }
      if (!active) {
        // rank of this idle lane in pair-major order (0, 32, 1, 33, ...): consecutive rays go to the two lanes of a pair
        const uint32_t l5 = lane & 31u, below = (1u << l5) - 1u, ilo = (uint32_t)idle, ihi = (uint32_t)(idle >> 32);
        const uint32_t my = base + (uint32_t)__popc(ilo & below) + (uint32_t)__popc(ihi & below) + (lane >= 32u ? ((ilo >> l5) & 1u) : 0u);
        if (my < hi) {
          {
            float4 a, b;
            if (phase == 0u) { a = pb.so[my]; b = pb.sd[my]; } else { a = ro[my]; b = rd[my]; }
            r = make_ray_ctx(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z));
            tbest = phase == 0u ? b.w : FLT_MAX; path = f2u(a.w);  // a closest-hit ray always starts at FLT_MAX (spt.hpp:301-305): the queue's fourth word carries the path's RNG key instead
          }
          hu = 0.f; hv = 0.f; htri = 0xffffffffu; hprim = 0; idx = my;
          ng_base = 0; ng_hits = 0x80000000u; tg = 0; tq = 0; sp = 0;  // the root as a one-child group (bvh8.h: traverse8)
          any = phase == 0u;
          active = true;
        }
      }
      if (phase < 2u && __ballot(active) == 0ull) continue;  // nothing in flight yet: go fetch from the next range
    }
    if (!__ballot(active)) break;
#if PHX_COUNT
    ++cnt_iter;
    cnt_idle += (uint32_t)__popcll(__ballot(!active));
#endif
    // ---- one node visit.  A lane's node work (ng_base, ng_hits, the stack) and its triangle work (tg_base, tg: the hit leaf
    // slots of ONE visited node; a second such group waits in the lane's LDS slot) are independent: a lane whose last node left
    // triangles to test goes on to its next node in the same iteration, and sits the node block out only when both triangle
    // groups are taken.  (Until round 4 a lane with pending triangles sat out: 8.6 of 64 lanes in every iteration.)  Triangles
    // are then tested LATER relative to the node visits than in traverse8 — against a tbest that is at most smaller — and the
    // boxes are culled against a tbest that is at most larger: the lane meets a superset of traverse8's triangles, and the closest
    // hit with its tie rule does not depend on which superset (bvh8.h).
    const bool node_ok = active && ng_hits > 0x00ffffffu && (any ? !(tg & TG_PENDING) : tq == 0u);  // an any-hit ray tests what it has first: a hit ends it
#if PHX_COUNT
    if (__ballot(node_ok)) ++cnt_nb;
    cnt_pend += (uint32_t)__popcll(__ballot(active && !node_ok && (tg & TG_PENDING)));
#endif
    if (node_ok) {
      const uint32_t bit = 31u - (uint32_t)__clz((int)ng_hits);
      const uint32_t rest = ng_hits & ~(1u << bit);
      if (rest > 0x00ffffffu) {
#if PHX_COUNT
        ++cnt_push[sp < 7 ? sp : 7];
#endif
        // SPILL: the top lds_levels entries of the stack live in LDS, the rare deeper ones in HBM (sc.stack_spill)
        if (!SPILL || (uint32_t)sp < lds_levels) {
          if constexpr (PACKED) {
            reinterpret_cast<uint32_t*>(stack_base)[sp * BLOCK] = (ng_base & 0x00ffffffu) | (rest & 0xff000000u);
            stack_valid[((uint32_t)sp >> 2) * (BLOCK * 4u) + ((uint32_t)sp & 3u)] = (uint8_t)rest;   // four levels share a lane's dword: the access pattern of a 4-byte column
          } else stack_base[sp * BLOCK] = make_uint2(ng_base, rest);
        }
        else spill_base[(size_t)((uint32_t)sp - lds_levels) * sc.spill_stride] = make_uint2(ng_base, rest);
        ++sp;
      }
      const uint32_t slot = (bit - 24u) ^ r.oct_inv;
      const uint32_t ni = ng_base + (uint32_t)__popc(ng_hits & 0xffu & ~(0xffffffffu << slot));
      uint32_t w[16];
#if PHX_COUNT
      if (ni < ntop) ++cnt_lds[any ? 1 : 0]; else ++cnt_mem[any ? 1 : 0];
#endif
      // top-of-tree nodelets are staged in LDS: four ds_read_b128 instead of four L1 requests per lane.  The LDS lanes go
      // first: both groups write the same registers, so the second group waits for the first one's data — a few dozen cycles
      // for LDS, several hundred for the L1/L2 path if it went first.
      const bool in_lds = ni < ntop;
      {
        const uint4* s4 = top + (in_lds ? ni : 0u) * (PHX_NODE_LDS_BYTES / 16u);  // every lane reads (element 0 when its node is not staged)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint4 v = s4[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
      }
      if (!in_lds) {
        const uint4* s4 = reinterpret_cast<const uint4*>(sc.pool) + (size_t)ni * 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint4 v = s4[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
      }
#if PHX_PROBE_VMEM
      // sensitivity probe (never in the product build): PHX_PROBE_VMEM more 16-byte loads per node visit, from the neighbouring element
      if (!in_lds) {
        const uint4* s4 = reinterpret_cast<const uint4*>(sc.pool) + (size_t)(ni ^ 1u) * 4u;
#pragma unroll
        for (int k = 0; k < PHX_PROBE_VMEM; ++k) {
#if PHX_PROBE_VMEM_DWORD
          w[2] ^= s4[k].x & (refill_min >> 31);   // same addresses, a quarter of the bytes
#else
          const uint4 v = s4[k]; w[2] ^= (v.x ^ v.y ^ v.z ^ v.w) & (refill_min >> 31);
#endif
        }
      }
#endif
#if PHX_PERM_LUT
      uint32_t hm = node_hitmask(w, sc.grid, r, tbest, [&](uint32_t m) { return (uint32_t)perm_lut[(r.oct_inv << 8) | m]; });
#else
      uint32_t hm = node_hitmask(w, sc.grid, r, tbest);
#endif
#if PHX_PROBE_VALU
      {  // sensitivity probe (never in the product build): PHX_PROBE_VALU more v_fma_f32 per node visit
        float x = tbest;
#pragma unroll
        for (int k = 0; k < PHX_PROBE_VALU; ++k) x = __builtin_fmaf(x, 0.99999f, 1.0e-3f);
        hm ^= f2u(x) & (refill_min >> 31);
      }
#endif
      ng_base = w[3];                // the children of the node just visited: nodelets and triangle records, in slot order
      ng_hits = hm & 0xff0000ffu;    // pending inner children | valid mask
      const uint32_t tnew = (hm >> 8) & TG_PENDING;  // hit leaf slots of this node -> bits 8..15
      if (tnew) {
        const uint32_t g = (hm & 0xffu) | tnew;       // valid mask | pending triangles
        if (!(tg & TG_PENDING)) { tg_base = w[3]; tg = g; }
        else { tq_base = w[3]; tq = g; }
      }
    }
    // ---- one triangle test per lane that has one pending
    const bool tpend = active && (tg & TG_PENDING);
#if PHX_COUNT
    if (__ballot(tpend)) {
      ++cnt_tb;
      const uint32_t mine = active ? (uint32_t)__popc((tg >> 8) & 0xffu) + (uint32_t)__popc((tq >> 8) & 0xffu) : 0u;  // <= 16
      uint32_t pairs = 0;
      for (uint32_t b = 0; b < 5u; ++b) pairs += (uint32_t)__popcll(__ballot((mine >> b) & 1u)) << b;
      cnt_pairs += pairs;
      ++cnt_pairs_hist[pairs <= 8u ? 0 : pairs <= 16u ? 1 : pairs <= 24u ? 2 : pairs <= 32u ? 3 : pairs <= 48u ? 4 : pairs <= 64u ? 5 : pairs <= 96u ? 6 : 7];
    }
#endif
    if (tpend) {
      const uint32_t k = 23u - (uint32_t)__clz((int)(tg & TG_PENDING));  // slot of the highest pending triangle (bits 8..15 -> 7..0)
      tg &= ~(0x100u << k);
      const uint32_t ti = tg_base + (uint32_t)__popc(tg & 0xffu & ~(0xffffffffu << k));
      const uint4* t4 = reinterpret_cast<const uint4*>(sc.pool) + (size_t)ti * 4u;
      const uint4 t0 = t4[0], t1 = t4[1], t2 = t4[2];  // three of the record's four words: v0, e0, e1, prim
      TriRec T;
      T.v0x = u2f(t0.x); T.v0y = u2f(t0.y); T.v0z = u2f(t0.z); T.e0x = u2f(t0.w);
      T.e0y = u2f(t1.x); T.e0z = u2f(t1.y); T.e1x = u2f(t1.z); T.e1y = u2f(t1.w);
      T.e1z = u2f(t2.x); T.prim = t2.y;
      float us, vs, ds;
#if PHX_COUNT
      ++cnt_tri[any ? 1 : 0];
#endif
      if (mt_intersect(T, r.o, r.d, tbest, hprim, us, vs, ds)) {
        tbest = ds; hu = us; hv = vs; htri = ti; hprim = T.prim;
        if (any) active = false;  // occluded: nothing to add
      }
      if (!(tg & TG_PENDING) && tq != 0u) { tg_base = tq_base; tg = tq; tq = 0u; }  // this group is done and another one waits
    }
    // ---- pop the next node group; a ray is finished when neither nodes nor triangles are left
    if (active && ng_hits <= 0x00ffffffu) {
      if (sp == 0) {
        if (!(tg & TG_PENDING)) {
          if (any) {  // unoccluded: out += beta * li (spt.hpp:184-186); one shadow ray per path and step
            const float4 cc = pb.sc[idx];
            float4 rr = pb.pr[path];
            rr.x += cc.x; rr.y += cc.y; rr.z += cc.z;
            pb.pr[path] = rr;
          } else {
            pb.hit[idx] = make_float4(tbest, hu, hv, u2f(htri));
          }
          active = false;
        }
      } else {
        --sp;
        uint2 e;
        if (!SPILL || (uint32_t)sp < lds_levels) {
          if constexpr (PACKED) {
            const uint32_t word = reinterpret_cast<uint32_t*>(stack_base)[sp * BLOCK];
            const uint32_t vb = stack_valid[((uint32_t)sp >> 2) * (BLOCK * 4u) + ((uint32_t)sp & 3u)];
            e = make_uint2(word & 0x00ffffffu, (word & 0xff000000u) | vb);
          } else e = stack_base[sp * BLOCK];
        }
        else e = spill_base[(size_t)((uint32_t)sp - lds_levels) * sc.spill_stride];
        ng_base = e.x; ng_hits = e.y;
      }
    }
  }
#if PHX_COUNT
  for (int k = 0; k < 2; ++k) {
    atomicAdd(&pb.stats->node_visits_lds[k], (unsigned long long)cnt_lds[k]);
    atomicAdd(&pb.stats->node_visits_mem[k], (unsigned long long)cnt_mem[k]);
    atomicAdd(&pb.stats->tri_tests[k], (unsigned long long)cnt_tri[k]);
  }
  for (int k = 0; k < 8; ++k) if (cnt_push[k]) atomicAdd(&pb.stats->stack_pushes[k], (unsigned long long)cnt_push[k]);
  if (lane == 0) {
    atomicAdd(&pb.stats->tri_block_execs, (unsigned long long)cnt_tb);
    atomicAdd(&pb.stats->wave_iters, (unsigned long long)cnt_iter);
    atomicAdd(&pb.stats->node_block_execs, (unsigned long long)cnt_nb);
    atomicAdd(&pb.stats->refills, (unsigned long long)cnt_refill);
    atomicAdd(&pb.stats->idle_lane_iters, (unsigned long long)cnt_idle);
    atomicAdd(&pb.stats->tri_pending_lane_iters, (unsigned long long)cnt_pend);
    atomicAdd(&pb.stats->tri_pairs_pending, (unsigned long long)cnt_pairs);
    for (int k = 0; k < 8; ++k) if (cnt_pairs_hist[k]) atomicAdd(&pb.stats->tri_pairs_hist[k], (unsigned long long)cnt_pairs_hist[k]);
  }
#endif
}

// Trace kernel: closest-hit rays of ray queue `q` (if do_closest) and any-hit rays of the shadow queue `sq`
// filled by the previous k_shade (if do_shadow) in ONE launch, so that late bounces with few rays still fill
// the chip.  Queue lengths are only known on the device.
//   The launch is persistent — exactly the resident workgroups — and work is handed out in two levels:
//   a workgroup takes a range of 16 x 64 consecutive rays from a global cursor (its first range is its own by
//   position), its waves take 64-ray chunks out of that range through a word in LDS.  Fine chunks keep the drain phase short,
//   and the global cursor sees one returning atomic per 1024 rays instead of one per chunk (which the wave has to wait for:
//   with per-wave global chunks 512 rays were the optimum, 64-ray chunks were atomic-bound).  A wave
//   drains once per launch, not once per slice, and a slow image region is shared by everyone.
// Dynamic LDS layout: [ntop pool elements x 80 B][levels x BLOCK stack entries x 8 B][12 cursor words][2 KB octant table].
template <int BLOCK, bool SPILL = false, bool PACKED = false>
__global__ void __launch_bounds__(BLOCK, 8) k_trace(DevScene sc, PassBuffers pb, int q, int sq, int do_closest, int do_shadow, uint32_t refill_min,
                                                 uint32_t ntop, uint32_t levels, uint32_t min_chunks, uint32_t target_chunks) {
  extern __shared__ uint4 smem[];
  uint4* top = smem;
  uint2* stack = reinterpret_cast<uint2*>(smem + ntop * (PHX_NODE_LDS_BYTES / 16u));
  // PACKED: [levels x BLOCK dwords][ceil(levels / 4) x BLOCK dwords of valid bytes]: 4 + 1 bytes per entry
  uint32_t* stack_words = reinterpret_cast<uint32_t*>(stack);
  uint32_t* valid_rows = stack_words + levels * BLOCK;
  uint32_t* cursor = PACKED ? valid_rows + ((levels + 3u) >> 2) * BLOCK : reinterpret_cast<uint32_t*>(stack + levels * BLOCK);
  uint8_t* perm_lut = reinterpret_cast<uint8_t*>(cursor + 12);  // PHX_PERM_LUT: 8 octants x 256 masks (cursor: 12 words)
  const uint32_t n_closest = do_closest ? pb.counters[q * CNT_STRIDE] : 0u;
  const uint32_t n_shadow = do_shadow ? pb.counters[CNT_SHADOW + sq * CNT_STRIDE] : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the queues k_shade will append to next: the other ray queue and the other shadow queue
    if (do_closest) { pb.counters[(q ^ 1) * CNT_STRIDE] = 0; pb.counters[CNT_SHADOW + (sq ^ 1) * CNT_STRIDE] = 0; }
    if (n_closest) atomicAdd(&pb.stats->rays_closest, (unsigned long long)n_closest);
    if (n_shadow) atomicAdd(&pb.stats->rays_shadow, (unsigned long long)n_shadow);
    // the host sizes the grids of the NEXT steps' shade launches by this length (queues never grow from one step to the next): a k_shade
    // grid sized for the queue's capacity launches 230 k workgroups at every step, nearly all of them empty from the third step on
    if (do_closest && pb.qlen_out) { __hip_atomic_store(pb.qlen_out, n_closest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
  if (n_shadow == 0u && n_closest == 0u) return;
  DynQueue dq{};
  dq.n0 = n_shadow; dq.n1 = n_closest;
  dq.num_waves = gridDim.x * (BLOCK / 64);
  dq.cursor = pb.counters + CNT_CURSOR;
  // ~min_chunks chunks per wave for mid-size queues, 64 .. target_chunks*64 rays each
  auto chunk_of = [&](uint32_t n) { return min(max((n / (dq.num_waves * max(min_chunks, 1u)) + 63u) & ~63u, 64u), target_chunks * 64u); };
  dq.c0 = chunk_of(n_shadow); dq.c1 = chunk_of(n_closest);
  // a workgroup's range: PHX_WG_CHUNKS chunks, fewer when the queue could not give every workgroup four such ranges
  auto range_of = [&](uint32_t n, uint32_t c) { return c * min(max(n / (gridDim.x * c * 4u), 1u), PHX_WG_CHUNKS); };
  dq.r0 = range_of(n_shadow, dq.c0); dq.r1 = range_of(n_closest, dq.c1);
  if (threadIdx.x == 0) {  // the workgroup's first range is its own by position; read by the waves after the barrier below
    unsigned long long* pack = reinterpret_cast<unsigned long long*>(cursor);
    for (uint32_t p = 0; p < 2u; ++p) {
      const uint32_t cwg = p == 0u ? dq.r0 : dq.r1, qn = p == 0u ? n_shadow : n_closest;
#if PHX_XCD_SEGMENTS
      const uint32_t sgm = blockIdx.x & (CNT_SEGS - 1u), kth = blockIdx.x / CNT_SEGS;  // the kth range of this XCD's segment
      const uint32_t lo_s = (uint32_t)((unsigned long long)qn * sgm / CNT_SEGS), hi_s = (uint32_t)((unsigned long long)qn * (sgm + 1u) / CNT_SEGS);
      const uint32_t lo = (uint32_t)min((unsigned long long)lo_s + (unsigned long long)kth * cwg, (unsigned long long)hi_s), hi = min(lo + cwg, hi_s);
      cursor[8 + p] = sgm;
#else
      const uint32_t lo = min(blockIdx.x * cwg, qn), hi = min(lo + cwg, qn);
#endif
      pack[p] = (unsigned long long)lo | ((unsigned long long)hi << 32);
      cursor[4 + p] = 0u; cursor[6 + p] = 0u;
    }
  }
  // stage the top of the tree (the pool is stored breadth first: its first ntop elements ARE the top levels); a staged element
  // keeps an 80-byte stride in LDS (PHX_NODE_LDS_BYTES, bvh8.h)
  const uint4* g4 = reinterpret_cast<const uint4*>(sc.pool);
  for (uint32_t i = threadIdx.x; i < ntop * 4u; i += BLOCK) top[(i >> 2) * (PHX_NODE_LDS_BYTES / 16u) + (i & 3u)] = g4[i];
#if PHX_PERM_LUT
  for (uint32_t i = threadIdx.x; i < 2048u; i += BLOCK) perm_lut[i] = (uint8_t)perm_xor8(i & 0xffu, i >> 8);
#endif
  __syncthreads();
  if constexpr (PACKED)
    trace_stream<BLOCK, SPILL, true>(sc, pb, q, cursor, reinterpret_cast<uint2*>(stack_words + threadIdx.x), reinterpret_cast<uint8_t*>(valid_rows + threadIdx.x), levels,
                                     sc.stack_spill + (size_t)blockIdx.x * BLOCK + threadIdx.x, refill_min, top, ntop, dq, perm_lut);
  else
    trace_stream<BLOCK, SPILL, false>(sc, pb, q, cursor, stack + threadIdx.x, nullptr, levels, sc.stack_spill + (size_t)blockIdx.x * BLOCK + threadIdx.x,
                                      refill_min, top, ntop, dq, perm_lut);
}

// stage-level hook (phx_dev_trace): one ray per lane run to completion with the plain traverse8 loop of bvh8.h.  The per-lane
// stack lives in dynamic LDS sized by the depth of the tree (<= PHX_MAX_BVH_DEPTH levels x 256 lanes x 8 B = 128 KB).
template <bool ANY>
__global__ void __launch_bounds__(PHX_BLOCK) k_trace_rays(DevScene sc, uint32_t n, const float4* ro, const float4* rd, float4* hit) {
  extern __shared__ uint2 ray_stack[];
  const uint32_t i = blockIdx.x * PHX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const float4 a = ro[i], b = rd[i];
  LdsStack<0> st{ray_stack + threadIdx.x, 0};
  Hit h;
  traverse8<ANY>(sc.pool, sc.grid, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), b.w, h, st);
  hit[i] = make_float4(h.t, h.u, h.v, u2f(h.tri == 0xffffffffu ? 0xffffffffu : sc.tris[h.tri].prim));  // primitive in scene_t::triangles() order
}

// ---- primary rays as packets ---------------------------------------------------------------------------------------------------
// The camera rays of a pass are the one coherent launch of a frame: path ids are pixel-major, so the 64 rays of a wave are 64
// samples of ONE pixel (or of a few neighbouring ones when a pass holds fewer than 64 samples per pixel) and walk the same nodes.
// k_trace runs them like any other rays — every lane tests all eight child boxes of every node it visits, ~290 VALU instructions
// per visit, and the launch sits at the ALU cost of that: 12.9 ms of the 72 ms bench frame.  Here a wave walks the tree ONCE for
// its 64 rays: a node is tested against the packet's interval of origins and reciprocal directions — eight lanes, one child box
// each, interval arithmetic on the same slab test — and every lane runs the reference's triangle test on each triangle the packet
// reaches.  The packet's bounds contain what every lane would get from the same expression with its own ray (a product of two intervals
// contains every member's product, rounding included, because rounding is monotonic); k_trace's node_hit8 evaluates the slab
// distances in another order, so the two differ by roundings, which 2^-16 of the distance on both ends is there to cover (node_hit8
// itself pads by 2^-20).  The lanes then meet a superset of the triangles they meet in k_trace, and the closest hit with its
// lowest-primitive tie rule does not depend on which superset (bvh8.h); that the films are identical bit for bit is tested, not assumed
// (EXPERIMENTS.md Part B section 4).  A wave whose rays do not share a direction octant (a pixel on one of the film's axes) falls back to the
// per-lane walk of bvh8.h.
// wave-wide min / max through DPP (four shifts inside the rows of 16, two row broadcasts; the result lands in lane 63): 13 instructions,
// where six __shfl_xor rounds would be twelve trips through the LDS crossbar
template <bool MAX>
__device__ __forceinline__ float wave_reduce_f(float x) {
#define PHX_DPP_STEP(ctrl, rows) { const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), ctrl, rows, 0xf, false)); x = MAX ? fmaxf(x, t) : fminf(x, t); }
  PHX_DPP_STEP(0x111, 0xf) PHX_DPP_STEP(0x112, 0xf) PHX_DPP_STEP(0x114, 0xf) PHX_DPP_STEP(0x118, 0xf)  // row_shr:1, 2, 4, 8
  PHX_DPP_STEP(0x142, 0xa) PHX_DPP_STEP(0x143, 0xc)                                                      // row_bcast:15 -> rows 1, 3; row_bcast:31 -> rows 2, 3
#undef PHX_DPP_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ float wave_min_f(float x) { return wave_reduce_f<false>(x); }
__device__ __forceinline__ float wave_max_f(float x) { return wave_reduce_f<true>(x); }
// What a node test needs of the packet, per axis (wave-uniform).  With s = +1 / -1 the common sign of the rays' direction on the axis,
// a ray's slab distance to the plane at c is (s (c - o)) * |1/d|: both factors are monotonic in what varies over the packet, so
//   entry  >=  fma(s, c_near, kn) * (that >= 0 ? bmin : bmax),   kn = -s * (the origin coordinate that makes s (c - o) smallest)
//   exit   <=  fma(s, c_far,  kf) * (that >= 0 ? bmax : bmin),   kf = -s * (the one that makes it largest)
// and, rounding being monotonic, the bounds hold for the rounded values every lane would compute from its own ray.
struct PacketBounds { float s[3], kn[3], kf[3], bmin[3], bmax[3]; };
__device__ __forceinline__ void packet_axis(PacketBounds& B, int a, float o_lo, float o_hi, float ia_lo, float ia_hi /* this lane's rays: origin and |1/d| ranges */,
                                            bool down /* the rays run towards -axis */) {
  const float omin = wave_min_f(o_lo), omax = wave_max_f(o_hi);
  B.bmin[a] = wave_min_f(ia_lo); B.bmax[a] = wave_max_f(ia_hi);
  B.s[a] = down ? -1.0f : 1.0f;
  B.kn[a] = down ? omin : -omax;   // s = -1: s (c - o) = o - c is smallest at omin, and -s * omin = omin
  B.kf[a] = down ? omax : -omin;
}
typedef unsigned int phx_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) phx_u4* phx_const_u4p;  // "constant" address space: a wave-uniform address is read through the scalar cache
// bit s of the result: child slot s of the node may be hit by some ray of the packet (every lane returns the same mask)
__device__ __forceinline__ uint32_t packet_node_hit8(const uint32_t* w, const SceneGrid& g, const PacketBounds& B, uint32_t oct_inv, float tmax_wave, uint32_t& valid) {
  float px, py, pz;
  node_origin_decode(w[0], w[1], g, px, py, pz, valid);
  const uint32_t e = w[2];
  const float sx = u32_as_f32((e & 0xffu) << 23), sy = u32_as_f32(((e >> 8) & 0xffu) << 23), sz = u32_as_f32(((e >> 16) & 0xffu) << 23);
  // lane j (mod 8) takes child slot j: byte j of each of the six 8-byte plane rows (one 64-bit shift each; a select between the two
  // words of a row would be turned into an indexed read of w[], i.e. scratch memory)
  const uint32_t sh = (__lane_id() & 7u) * 8u;
  auto plane = [&](int row) { return (float)((uint32_t)((((unsigned long long)w[row + 1] << 32) | (unsigned long long)w[row]) >> sh) & 0xffu); };
  const float lox = fmaf(plane(4), sx, px), hix = fmaf(plane(10), sx, px);
  const float loy = fmaf(plane(6), sy, py), hiy = fmaf(plane(12), sy, py);
  const float loz = fmaf(plane(8), sz, pz), hiz = fmaf(plane(14), sz, pz);
  const bool nx = !(oct_inv & 4u), ny = !(oct_inv & 2u), nz = !(oct_inv & 1u);  // the packet's rays all run down this axis
  const float anx = fmaf(B.s[0], nx ? hix : lox, B.kn[0]), afx = fmaf(B.s[0], nx ? lox : hix, B.kf[0]);
  const float any_ = fmaf(B.s[1], ny ? hiy : loy, B.kn[1]), afy = fmaf(B.s[1], ny ? loy : hiy, B.kf[1]);
  const float anz = fmaf(B.s[2], nz ? hiz : loz, B.kn[2]), afz = fmaf(B.s[2], nz ? loz : hiz, B.kf[2]);
  const float tnx = anx * (anx >= 0.0f ? B.bmin[0] : B.bmax[0]), tfx = afx * (afx >= 0.0f ? B.bmax[0] : B.bmin[0]);
  const float tny = any_ * (any_ >= 0.0f ? B.bmin[1] : B.bmax[1]), tfy = afy * (afy >= 0.0f ? B.bmax[1] : B.bmin[1]);
  const float tnz = anz * (anz >= 0.0f ? B.bmin[2] : B.bmax[2]), tfz = afz * (afz >= 0.0f ? B.bmax[2] : B.bmin[2]);
  float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
  const float pad = 1.52587890625e-05f;  // 2^-16 of the distance on both ends: the lanes' own tests (node_hit8) differ from this one by roundings
  tn = tn - fabsf(tn) * pad; tf = tf + fabsf(tf) * pad;
  const bool hit = tn <= tf && tf >= 0.0f && tn <= tmax_wave;
  return (uint32_t)__ballot(hit) & 0xffu & valid;  // empty slots carry inverted boxes AND are masked out
}

#define PHX_PRIMARY_BLOCK 256
#ifndef PHX_PRIMARY_WAVES
#define PHX_PRIMARY_WAVES 5  /* waves per SIMD the register allocator leaves room for: 95 VGPRs with 4 rays per lane (4: 98; 6: 80 + 48 B of scratch); 3.65 -> 3.35 ms */
#endif
// RPL rays per lane: a packet is 64 x RPL consecutive paths (ray k of lane l is path base + 64 k + l).  The node tests are per packet,
// so they are shared by more rays; the host picks RPL so that a packet stays inside one pixel's samples (launch_trace_primary).
template <int RPL, bool LENS = false /* thin-lens camera: camera_ray<true> */>
__global__ void __launch_bounds__(PHX_PRIMARY_BLOCK) __attribute__((amdgpu_waves_per_eu(PHX_PRIMARY_WAVES, 8))) k_trace_primary(DevScene sc, PassBuffers pb, uint32_t npaths, uint32_t sample0, int q, int sq) {
  extern __shared__ uint2 primary_lds[];  // [4 waves x PHX_MAX_BVH_DEPTH] one shared stack per wave, then [levels x 256] per-lane stacks of the fallback walk
  uint2* lane_stacks = primary_lds + (PHX_PRIMARY_BLOCK / 64) * PHX_MAX_BVH_DEPTH;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // what k_trace does at the start of a step (no shadow rays yet in step 0)
    pb.counters[(q ^ 1) * CNT_STRIDE] = 0; pb.counters[CNT_SHADOW + (sq ^ 1) * CNT_STRIDE] = 0;
    atomicAdd(&pb.stats->rays_closest, (unsigned long long)npaths);
  }
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
  // (Dealing the workgroups to the XCDs in bands of the image — XCD x the x-th eighth of the packets — is SLOWER here: 3.35 -> 3.76 ms,
  // profiles/r04_o_xcd_primary_ab.log: bands of the image cost differently, and a plain grid has no way to steal.)
  const uint32_t base = (blockIdx.x * (PHX_PRIMARY_BLOCK / 64) + wave) * (64u * RPL);
  if (base >= npaths) return;
  uint32_t idx[RPL];
  RayCtx r[RPL];
  float tbest[RPL], hu[RPL], hv[RPL];
  uint32_t htri[RPL], hprim[RPL];
  bool mixed = false;
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    idx[k] = base + 64u * k + lane;
    v3 co, cd;
    camera_ray<LENS>(sc, pb, min(idx[k], npaths - 1u), sample0, co, cd);  // rays past the end repeat the last one and write nothing
    r[k] = make_ray_ctx(co, cd);
    tbest[k] = FLT_MAX; hu[k] = 0.0f; hv[k] = 0.0f; htri[k] = 0xffffffffu; hprim[k] = 0;
  }
  const uint32_t oct = PHX_UNI(r[0].oct_inv);
#pragma unroll
  for (int k = 0; k < RPL; ++k) mixed = mixed || r[k].oct_inv != oct;
  if (__ballot(mixed) != 0ull) {
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
      LdsStack<0> st{lane_stacks + threadIdx.x, 0};
      Hit h;
      // a NaN ray (LENS: a lens sample with a zero coordinate) hits nothing, but the conservative box test ignores NaNs: no walk of the whole tree
      if constexpr (LENS) { const float chk = (r[k].o.x + r[k].o.y + r[k].o.z) + (r[k].d.x + r[k].d.y + r[k].d.z); if (!(fabsf(chk) <= FLT_MAX)) continue; }
      traverse8<false>(sc.pool, sc.grid, r[k].o, r[k].d, FLT_MAX, h, st);
      tbest[k] = h.t; hu[k] = h.u; hv[k] = h.v; htri[k] = h.tri;
    }
#if PHX_COUNT
    if (lane == 0) { atomicAdd(&pb.stats->primary_packets, 1ull); atomicAdd(&pb.stats->primary_fallbacks, 1ull); }
#endif
  } else {
    PacketBounds B;
    {
      v3 olo = r[0].o, ohi = r[0].o, ilo(fabsf(r[0].idx), fabsf(r[0].idy), fabsf(r[0].idz)), ihi = ilo;
#pragma unroll
      for (int k = 1; k < RPL; ++k) {
        olo = v3(fminf(olo.x, r[k].o.x), fminf(olo.y, r[k].o.y), fminf(olo.z, r[k].o.z)); ohi = v3(fmaxf(ohi.x, r[k].o.x), fmaxf(ohi.y, r[k].o.y), fmaxf(ohi.z, r[k].o.z));
        const v3 ia(fabsf(r[k].idx), fabsf(r[k].idy), fabsf(r[k].idz));
        ilo = v3(fminf(ilo.x, ia.x), fminf(ilo.y, ia.y), fminf(ilo.z, ia.z)); ihi = v3(fmaxf(ihi.x, ia.x), fmaxf(ihi.y, ia.y), fmaxf(ihi.z, ia.z));
      }
      packet_axis(B, 0, olo.x, ohi.x, ilo.x, ihi.x, !(oct & 4u)); packet_axis(B, 1, olo.y, ohi.y, ilo.y, ihi.y, !(oct & 2u)); packet_axis(B, 2, olo.z, ohi.z, ilo.z, ihi.z, !(oct & 1u));
    }
    float tmax_wave = FLT_MAX;  // the farthest closest-hit distance of the packet so far
    uint32_t ng_base = 0, ng_hits = 0x80000000u;  // the root as a one-child group (bvh8.h: traverse8)
    uint32_t sp = 0, guard = 0;
#if PHX_COUNT
    uint32_t cnt_nodes = 0, cnt_tris = 0, cnt_lanes = 0;
#endif
    uint2* stack = primary_lds + wave * PHX_MAX_BVH_DEPTH;
    for (;;) {
      if (++guard > PHX_TRACE_WATCHDOG) { if (lane == 0) atomicAdd(&pb.stats->watchdog, 1ull); break; }
      const uint32_t bit = 31u - (uint32_t)__clz((int)ng_hits);
      const uint32_t rest = ng_hits & ~(1u << bit);
      if (rest > 0x00ffffffu) { if (lane == 0) stack[sp] = make_uint2(ng_base, rest); ++sp; }
      const uint32_t slot = (bit - 24u) ^ oct;
      const uint32_t ni = PHX_UNI(ng_base + (uint32_t)__popc(ng_hits & 0xffu & ~(0xffffffffu << slot)));
      uint32_t w[16];
      {
        const phx_const_u4p s4 = (phx_const_u4p)(sc.pool) + (size_t)ni * 4u;  // the pool is never written while a frame runs
#pragma unroll
        for (int k = 0; k < 4; ++k) { const phx_u4 v = s4[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
      }
      uint32_t valid;
#if PHX_COUNT
      ++cnt_nodes;
#endif
      const uint32_t hit8 = packet_node_hit8(w, sc.grid, B, oct, tmax_wave, valid);
      const uint32_t imask = w[2] >> 24;
      ng_base = PHX_UNI(w[3]);
      ng_hits = PHX_UNI((perm_xor8(hit8 & imask, oct) << 24) | valid);
      uint32_t th = PHX_UNI(hit8 & ~imask);
      while (th != 0u) {
        const uint32_t k = 31u - (uint32_t)__clz((int)th);
        th &= ~(1u << k);
        const uint32_t ti = PHX_UNI(ng_base + (uint32_t)__popc(valid & ~(0xffffffffu << k)));
        const phx_const_u4p t4 = (phx_const_u4p)(sc.pool) + (size_t)ti * 4u;
        const phx_u4 t0 = t4[0], t1 = t4[1], t2 = t4[2];
        TriRec T;
        T.v0x = u2f(t0.x); T.v0y = u2f(t0.y); T.v0z = u2f(t0.z); T.e0x = u2f(t0.w);
        T.e0y = u2f(t1.x); T.e0z = u2f(t1.y); T.e1x = u2f(t1.z); T.e1y = u2f(t1.w);
        T.e1z = u2f(t2.x); T.prim = t2.y;
        bool any_better = false;
        float tworst = 0.0f;
#pragma unroll
        for (int m = 0; m < RPL; ++m) {
          float us, vs, ds;
          const bool better = mt_intersect(T, r[m].o, r[m].d, tbest[m], hprim[m], us, vs, ds);
          if (better) { tbest[m] = ds; hu[m] = us; hv[m] = vs; htri[m] = ti; hprim[m] = T.prim; }
          any_better = any_better || better;
          tworst = fmaxf(tworst, tbest[m]);
        }
#if PHX_COUNT
        ++cnt_tris; cnt_lanes += (uint32_t)__popcll(__ballot(any_better));
#endif
        if (__ballot(any_better) != 0ull) tmax_wave = wave_max_f(tworst);
      }
      if (ng_hits <= 0x00ffffffu) {
        if (sp == 0u) break;
        --sp;
        const uint2 e = stack[sp];
        ng_base = PHX_UNI(e.x); ng_hits = PHX_UNI(e.y);
      }
    }
#if PHX_COUNT
    if (lane == 0) {
      atomicAdd(&pb.stats->primary_packets, 1ull);
      atomicAdd(&pb.stats->primary_node_tests, (unsigned long long)cnt_nodes);
      atomicAdd(&pb.stats->primary_tri_tests, (unsigned long long)cnt_tris);
      atomicAdd(&pb.stats->primary_tri_lanes_hit, (unsigned long long)cnt_lanes);
    }
#endif
  }
#pragma unroll
  for (int k = 0; k < RPL; ++k)
    if (idx[k] < npaths) pb.hit[idx[k]] = make_float4(tbest[k], hu[k], hv[k], u2f(htri[k]));
}

// ---- shade + next-event estimation + integrate ------------------------------------------------------
__device__ __forceinline__ v3 shading_normal(const DevScene& sc, uint32_t elem /* pool index of the triangle record */, bool smooth, const v3& e0, const v3& e1, float u, float v) {
  if (smooth) {  // mesh_t::shading_parameters, src/mesh.cpp:187-199
    const float w = 1 - u - v;
    const float* pn = sc.elem_normals + 9 * (size_t)elem;
    const v3 n0(pn[0], pn[1], pn[2]), n1(pn[3], pn[4], pn[5]), n2(pn[6], pn[7], pn[8]);
    return normalize_inplace(w * n0 + u * n1 + v * n2);
  }
  return normalize_inplace(cross(e0, e1));  // (v1-v0) x (v2-v0), never flipped (mesh.cpp:201-215)
}
// k_shade_g: the three vertex normals of the hit's pool element are requested TOGETHER WITH its shade record — whether the face is smooth
// is a bit of that record — and used or dropped when it has landed (flat faces of a scene with smooth ones pay a 36-byte gather for nothing;
// scenes without smooth faces have no table and no request)
struct VertexNormals { v3 n0, n1, n2; };
__device__ __forceinline__ VertexNormals request_vertex_normals(const DevScene& sc, uint32_t elem) {
  VertexNormals N{v3(0.0f), v3(0.0f), v3(0.0f)};
  if (sc.elem_normals) {
    const float* pn = sc.elem_normals + 9 * (size_t)elem;
    N.n0 = v3(pn[0], pn[1], pn[2]); N.n1 = v3(pn[3], pn[4], pn[5]); N.n2 = v3(pn[6], pn[7], pn[8]);
  }
  return N;
}
__device__ __forceinline__ v3 shading_normal(const VertexNormals& N, bool smooth, const v3& geometric /* DevScene::elem_shade */, float u, float v) {
  if (smooth) { const float w = 1 - u - v; return normalize_inplace(w * N.n0 + u * N.n1 + v * N.n2); }  // the same expression as above
  return geometric;
}

__device__ __forceinline__ float luminance(const v3& c) {  // color::y, src/utils/color.hpp:13-16
  return (float)0.212671 * c.x + (float)0.715160 * c.y + (float)0.072169 * c.z;
}

// The Lambert-only scenes (the soups, the Cornell box): shade + NEE + integrate of one path per thread in one go; HBM-stream bound.
// Scenes with other closures go through k_shade_g below.
template <int MATS /* 1 Lambert lobes only; 2 at most one Lambert lobe per material (DevScene::diffuse_only) */,
          bool FIRST /* queue q = the camera rays of this pass: nothing to read but the hit */, bool LENS = false /* FIRST: thin-lens camera */>
__global__ void __launch_bounds__(PHX_SHADE_BLOCK_D) __attribute__((amdgpu_waves_per_eu(4, 8))) k_shade(DevScene sc, PassBuffers pb, int q, int sq, uint32_t sample0) {
  static_assert(MATS == 1 || MATS == 2, "general closures: k_shade_g");
  constexpr bool DIFFUSE_ONLY = true;
  constexpr int MAXL = MATS == 2 ? 1 : 8;
  constexpr int PHX_SHADE_BLOCK = PHX_SHADE_BLOCK_D;
  __shared__ uint32_t lds_cnt[2 * ((PHX_SHADE_BLOCK >> 6) + 1)];
  // a k_trace wave that hit its watchdog left rays untraced: their hit records are whatever the buffer held (an earlier step's, an earlier
  // scene's) — the frame is reported as failed anyway (device.cpp), so nothing is shaded, nothing appended, and the queues of the following
  // steps are empty
  if (pb.stats->watchdog | pb.stats->ring_watchdog) return;
  const uint32_t count = pb.counters[q * CNT_STRIDE];
  const uint32_t i = blockIdx.x * PHX_SHADE_BLOCK + threadIdx.x;
  if (i == 0) zero_cursors(pb.counters);  // the next k_trace pulls its chunks from here
  // One block per workgroup, grid sized for the queue's capacity.  (A fixed grid walking the queue in a loop — what k_shade_g does —
  // costs this kernel its occupancy: whatever is loop-invariant gets hoisted and held in registers, 51-61 VGPRs become 73-82, and
  // 1024-thread workgroups need <= 64 to run two per CU.)
  if (blockIdx.x * PHX_SHADE_BLOCK >= count) return;
  const bool live = i < count;
  bool alive = false, want_shadow = false, masked = false;
  uint32_t path = 0, next_specular = 0;
  v3 nxt_o, nxt_d, sh_o, sh_d, contrib, nxt_beta;
  uint32_t nxt_depth = 0, key = 0;
  float sh_t = 0.0f;
  if (live) {
    float4 a, b, bd;
    const float4 h = pb.hit[i];
    if (FIRST) {
      v3 co, cd;
      camera_ray<LENS>(sc, pb, i, sample0, co, cd);
      a = make_float4(co.x, co.y, co.z, u2f(i)); b = make_float4(cd.x, cd.y, cd.z, FLT_MAX);
      bd = make_float4(1.0f, 1.0f, 1.0f, u2f(0u));  // state_t::reset: beta = 1, depth = 0
    } else {
      a = pb.ro[q][i]; b = pb.rd[q][i]; bd = pb.qs[q][i];
    }
    const uint32_t pbits = f2u(a.w);
    path = pbits & 0x7fffffffu;
    const bool specular = (pbits >> 31) != 0;
    v3 beta(bd.x, bd.y, bd.z);
    // radiance is only read-modified-written when this step adds something: out += beta * e with e == 0 and a
    // finite beta leaves `out` unchanged bit for bit (out is never -0), so the 32 B of traffic are skipped
    v3 add_e(0.0f); bool add_rad = false;
    const bool beta_finite = isfinite(bd.x) && isfinite(bd.y) && isfinite(bd.z);
    uint32_t depth = f2u(bd.w);
    // the path's RNG key: computed once, by the first shade of the pass, then carried in the ray record's fourth word (a closest-hit ray's
    // tmax is always FLT_MAX) — a dependent gather (pixel table), an integer division and four hash rounds less per later step
    if (FIRST) {
      const uint32_t pix = path / pb.num_samples, s = path - pix * pb.num_samples;
      const uint32_t xy = pb.pix_xy[pix];
      key = path_key(pb.seed, (xy >> 16) * sc.width + (xy & 0xffffu), sample0 + s);
    } else {
      key = f2u(b.w);
    }
    const uint32_t tri = f2u(h.w);
    const v3 o(a.x, a.y, a.z), d(b.x, b.y, b.z);
    if (tri != 0xffffffffu) {
      const float4 S = sc.elem_shade[tri];  // (geometric normal, material | smooth << 31): 16 bytes of the hit triangle instead of its 64-byte record
      const uint32_t pm = f2u(S.w);
      const v3 p = o + d * h.x;            // hits.p = p + wi*d
      const v3 wo = -d;                    // hits.wi = -wi
      const v3 n = (pm >> 31) != 0 ? shading_normal(sc, tri, true, v3(0.0f), v3(0.0f), h.y, h.z) : v3(S.x, S.y, S.z);
      // material_t::evaluate (material.cpp:419-458): the closure list at this hit.  A constant recipe is read from the table; a
      // material with a hit-dependent weight (glass: Fresnel-driven mix) gets its weights resolved for (n, hits.wi) first.
      const DevMaterial* mp = &sc.materials[pm & 0x7fffffffu];
      DevMaterial mh;
      if (MATS == 2) {  // the whole material in two 16-byte loads; only the fields named here are ever read (bsdf_f / bsdf_sample <true, 1>)
        const DevMatLite ml = sc.mat_lite[pm & 0x7fffffffu];
        mh.num_lobes = ml.lobes_flags & 0xffu; mh.lobes[0].flags = ml.lobes_flags >> 8; mh.lobes[0].type = L_DIFFUSE;
        mh.lobes[0].wx = ml.wx; mh.lobes[0].wy = ml.wy; mh.lobes[0].wz = ml.wz;
        mh.ex = ml.ex; mh.ey = ml.ey; mh.ez = ml.ez; mh.sheen_L5 = 0.0f;
        mp = &mh;
      }
      const DevMaterial& m = *mp;
      if (pb.pn && (FIRST || depth == 0)) pb.pn[path] = make_float4(n.x, n.y, n.z, 1.0f);
      const v3 e(m.ex, m.ey, m.ez);
      if (depth == 0 || specular) { add_e = e; add_rad = true; }  // spt.hpp:177-179
      // ---- next-event estimation: sampler_t::fresh_light_samples + light_sampler_t (sampling.cpp:160-179, spt.hpp:95-149)
      {
        const uint32_t b0 = depth * DIMS_PER_STEP;
        const float pick = draw_f32(key, b0 + DIM_LIGHT_PICK), lu = draw_f32(key, b0 + DIM_LIGHT_U), lv = draw_f32(key, b0 + DIM_LIGHT_V);
        const float nlf = (float)sc.num_lights;
        uint32_t l = (uint32_t)floorf(pick * nlf);
        if (l > sc.num_lights - 1) l = sc.num_lights - 1;
        const DevLight L = sc.lights[l];
        const float numf = (float)L.num_tris;
        uint32_t ti = (uint32_t)floorf(lu * numf);  // uniform by index (light.cpp:55), pdf = 1/area
        if (ti > L.num_tris - 1) ti = L.num_tris - 1;
        const float remapped = fminf(lu * numf - (float)ti, 1.0f - FLT_EPSILON);
        const DevLightTri LT = sc.light_tris[L.first_tri + ti];
        const float x = sqrtf(remapped);
        const float bu = 1 - x, bv = lv * x;     // triangle_t::sample, mesh.cpp:318-324
        const v3 la(LT.ax, LT.ay, LT.az), lb(LT.bx, LT.by, LT.bz), lc(LT.cx, LT.cy, LT.cz);
        const v3 P = bu * la + bv * lb + (1 - bu - bv) * lc;
        const float lpdf = L.lpdf;  // (1.0f / L.area) / nlf, evaluated at preprocess
        sh_o = v3(p.x + n.x * 0.0001f, p.y + n.y * 0.0001f, p.z + n.z * 0.0001f);
        v3 wl = P - sh_o;
        const float l2 = sdot(wl, wl);
        const float dist = sqrtf(l2) - 0.0001f;
        const float oolen = 1.0f / sqrtf(l2);
        wl = v3(wl.x * oolen, wl.y * oolen, wl.z * oolen);
        if (sdot(n, wl) >= 0.0f) {
          // li(), spt.hpp:212-255 — evaluated before the occlusion test; added by k_trace_shadow if unoccluded
          const v3 f = bsdf_f<DIFFUSE_ONLY, MAXL>(m, n, wl, wo);
          // the light's normal at the sampled point (mesh_t::shading_parameters on the light triangle) and its emission
          const v3 ln = LT.smooth ? shading_normal(sc, LT.prim, true, lb - la, lc - la, bu, bv) : v3(LT.nx, LT.ny, LT.nz);
          const v3 le(L.ex, L.ey, L.ez);
          const float pdf = lpdf * dist * dist / fabsf(dot(ln, -wl));
          const v3 li = ((le * 4.0f) * f) * (1.0f / pdf);
          contrib = beta * li;
          sh_d = wl; sh_t = dist;
          want_shadow = true;
        } else {
          masked = true;
        }
      }
      // ---- integrate: ++depth, russian roulette, bsdf sampling (spt.hpp:188-190, 257-328)
      depth += 1;
      float wgt = 1.0f;
      alive = depth < sc.max_depth;
      if (alive && depth >= 3) {
        const float qq = fmaxf(0.05f, 1.0f - luminance(beta));
        const float xi = draw_f32(key, (depth - 1u) * DIMS_PER_STEP + DIM_RR);
        alive = xi >= qq;
        if (alive) wgt = (1.0f / (1.0f - qq));
      }
      beta = beta * wgt;
      if (alive) {
        const uint32_t b1 = (depth - 1u) * DIMS_PER_STEP;
        v3 sampled; float pdf; uint32_t fl;
        const v3 f = bsdf_sample<DIFFUSE_ONLY, MAXL>(m, n, draw_f32(key, b1 + DIM_BSDF_U), draw_f32(key, b1 + DIM_BSDF_V), wo, sampled, pdf, fl);
        if ((f.x == 0.0f && f.y == 0.0f && f.z == 0.0f) || pdf == 0.0f) {
          alive = false;
        } else {
          const float weight = dot(n, sampled);
          beta = beta * (f * (fabsf(weight) / pdf));
          const float off = (weight < 0.0f) ? -0.0001f : 0.0001f;
          nxt_o = p + n * off;
          nxt_d = sampled;
          next_specular = (fl & B_SPECULAR) ? 1u : 0u;
        }
      }
    } else {
      // miss: environment lighting (deferred_shading_kernel.hpp:65-70, spt.hpp:199-202)
      v3 e(0.0f);
      if (sc.env_material >= 0) { const DevMaterial& m = sc.materials[sc.env_material]; e = v3(m.ex, m.ey, m.ez); }
      add_e = e; add_rad = true;
      if (FIRST && pb.pn) pb.pn[path] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      masked = true;  // a miss still occupies a (MASKED|SHADOW) slot in the reference's shadow stream (spt.hpp:138-141)
    }
    nxt_beta = beta; nxt_depth = depth;  // only the next k_shade of a surviving path reads them: they leave with the ray below
    if (FIRST) {  // r = 0 + beta * e: every path's radiance is written here, nothing is read
      const v3 rad = v3(0.0f) + v3(bd.x, bd.y, bd.z) * add_e;
      pb.pr[path] = make_float4(rad.x, rad.y, rad.z, 0.0f);
    } else if (add_rad && !(beta_finite && add_e.x == 0.0f && add_e.y == 0.0f && add_e.z == 0.0f)) {
      const float4 rr = pb.pr[path];
      const v3 rad = v3(rr.x, rr.y, rr.z) + v3(bd.x, bd.y, bd.z) * add_e;  // beta as it was when the ray arrived
      pb.pr[path] = make_float4(rad.x, rad.y, rad.z, 0.0f);
    }
  }
  // ---- stream compaction: survivors -> next ray queue, unmasked NEE rays -> shadow queue
  uint32_t no, ns;
  // (the rays leave in thread order.  Sorting the workgroup's survivors by direction octant and / or the Morton cell of their origin
  // before the append — 64 or 512 bins through LDS — bought k_trace 0.5 ms of 60 and cost this kernel 0.8-1.8 ms of 10 on every
  // workload, config 4 included: profiles/r03_g_bin_octant_ab.log, r03_i_bin_morton_ab.log; removed after commit "Ray-coherence experiments")
  block_append2<PHX_SHADE_BLOCK>(alive, &pb.counters[(q ^ 1) * CNT_STRIDE], want_shadow, &pb.counters[CNT_SHADOW + sq * CNT_STRIDE], lds_cnt, no, ns);
  if (alive) {
    pb.ro[q ^ 1][no] = make_float4(nxt_o.x, nxt_o.y, nxt_o.z, u2f(path | (next_specular << 31)));
    pb.rd[q ^ 1][no] = make_float4(nxt_d.x, nxt_d.y, nxt_d.z, u2f(key));  // (d, the path's RNG key)
    pb.qs[q ^ 1][no] = make_float4(nxt_beta.x, nxt_beta.y, nxt_beta.z, u2f(nxt_depth));  // the path's state travels with its ray
  }
  if (want_shadow) {
    pb.so[ns] = make_float4(sh_o.x, sh_o.y, sh_o.z, u2f(path));
    pb.sd[ns] = make_float4(sh_d.x, sh_d.y, sh_d.z, sh_t);
    pb.sc[ns] = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
  }
  (void)masked;  // rays_masked = rays_closest - rays_shadow: every shaded slot yields a shadow ray or a masked slot
}

// ---- general closures: shade + NEE + integrate with the hits of a workgroup sorted by material ----------------------------------
// k_shade_g replaces the round-2 k_shade<0/3> (128 VGPRs, 160-700 B of scratch per lane, 4 waves per SIMD: the weakest kernel of the
// repo).  What changed, and why:
//   * no per-hit copy of the material.  A glass hit used to resolve its closure weights into a private 576-byte DevMaterial (scratch
//     memory, and every material read a flat load); now bsdf_f / bsdf_sample resolve a lobe's weight where they use it (bsdf.h).
//   * the two halves of a step (light_sampler_t and integrator_t, spt.hpp:95-149 / 161-328) share the hit's tangent frame, built once.
//     (For most of round 3 the NEE ray was appended to its queue BEFORE roulette and BSDF sampling started, so that its ten registers
//     were dead by then; once the kernel stood at 4 waves per SIMD whatever it did, one append for both queues per round won.)
//   * a workgroup sorts a WINDOW of BLOCK x ITEMS hits by material, not BLOCK: with 16 recipes assigned round-robin a 512-hit
//     bucket sort left 2-3 materials in every wave; a window of 4096 leaves most waves with one (deferred_shading_kernel_t buckets
//     a 1024-slot stream per material for the same reason, deferred_shading_kernel.hpp:9-33, 63).
//   * a fixed grid walks the queue (see k_shade).
#ifndef PHX_SHADE_ITEMS_G
#define PHX_SHADE_ITEMS_G 8
#endif
#define PHX_SHADE_BUCKETS 64  /* sort key = material mod 64; then the misses; slots past the end of the queue go last */
#ifndef PHX_SHADE_KEY_PROBE
#define PHX_SHADE_KEY_PROBE 0
#endif
// probe builds (-DPHX_SHADE_TIMING=1, scripts/shade_phase_probe.py): s_memtime at the phase boundaries of a shading round, per wave.
// PHX_PHASE(n) closes phase n: everything the wave has in flight is waited for first, so that a phase is charged the latency of what it
// asked for (loads issued in a phase and consumed later would otherwise be billed to the consumer).
#if PHX_SHADE_TIMING
#define PHX_PHASE_DECL long long ph_t = clock64(); unsigned long long ph_acc[6] = {0, 0, 0, 0, 0, 0};
#define PHX_PHASE(n) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long now_ = clock64(); ph_acc[n] += (unsigned long long)(now_ - ph_t); ph_t = now_; }
#else
#define PHX_PHASE_DECL
#define PHX_PHASE(n)
#endif
#ifndef PHX_SHADE_PREFETCH_PERHIT
#define PHX_SHADE_PREFETCH_PERHIT 0  /* the same two stages in the per-hit (glass) instantiations: measured worthless in round 5 (profiles/r05_c_shade_prefetch_glass_ab.log) */
#endif
#ifndef PHX_SHADE_TRI_LDS
#define PHX_SHADE_TRI_LDS 1  /* the sort phase keeps each hit's pool index beside the permutation (16 KB of LDS), so that a round requests the hit triangle's shade record (and, where nothing is prefetched, its vertex normals) WITH the hit record and the ray instead of after the hit record has landed */
#endif
#ifndef PHX_SHADE_FIRST_SORT
#define PHX_SHADE_FIRST_SORT 0  /* 1: the camera entries' launch sorts its windows by material like every other.  0: it does not — the 64 lanes of a wave are 64 samples of ONE pixel there, which see one material (two or three on a silhouette) without any sorting */
#endif
#ifndef PHX_SCALAR_F_PERHIT
#define PHX_SCALAR_F_PERHIT 1  /* the per-hit (glass) instantiations read the recipe through the scalar cache too: with the ring append the kernel has the registers (127 / 123 VGPRs, no scratch; round 5: 16 B of scratch): closed showroom -4.4 %, glass showroom -2.9 % shade time (profiles/r06_i_perhit_knobs_ab.log) */
#endif
#ifndef PHX_SCALAR_F
#define PHX_SCALAR_F 1  /* bsdf_f's lobe loop reads the recipe through the scalar cache: -0.6 % shade time, 128 -> 121 VGPRs */
#endif
#ifndef PHX_SCALAR_S
#define PHX_SCALAR_S 0  /* bsdf_sample picks its lobe per lane: through the scalar path it is 2-3 % slower (profiles/r04_j_shade_scalar_ab.log) */
#endif
// The lanes that reach a closure evaluation, one distinct material of the wave at a time (after the window's sort by material most
// waves hold one): the material's index is made wave-uniform (v_readlane), its recipe is addressed in the CONSTANT address space, and
// every read with a uniform address — the lobe loops of bsdf_f and bsdf_sample — becomes an s_load into SGPRs instead of a
// vector load into 64 copies (round 3 made the address uniform without the address space: the compiler cannot prove the table is
// read-only then and keeps the vector loads; profiles/r03_v_unimat_ab.log).  `body` runs under `mine`; `cm` is the ConstMat&.
#define PHX_FOR_EACH_MATERIAL_OF_THE_WAVE(mat, cm, body)                                                                      \
  for (unsigned long long todo_ = __ballot(true); todo_ != 0ull;) {                                                           \
    uint32_t m0_ = (uint32_t)__builtin_amdgcn_readlane((int)(mat), (int)(__ffsll((long long)todo_) - 1));                     \
    const bool mine_ = (mat) == m0_;                                                                                          \
    /* inside `mine_` the optimiser knows mat == m0_ and would address the table with the per-lane register: hide the SGPR */ \
    asm volatile("" : "+s"(m0_));                                                                                             \
    if (mine_) { ConstMat& cm = *((ConstMat*)(sc.materials) + m0_); body; }                                                    \
    todo_ &= ~__ballot(mine_);                                                                                                \
  }
// ---- k_shade_g's append: a workgroup-local ring in LDS, no barrier and no global atomic inside a shading round -------------------------
// Until round 5 every round of 512 entries ended in block_append2: barrier, the workgroup's two atomics on the queue counters, barrier —
// 24 % of a wave's time (profiles/r05_c_shade_phases.md: the atomics' round trip ~2 200 clocks, ~5 000 waiting for the slowest of the eight
// waves, every round).  Now a WAVE reserves its slots in a ring of 2 x PHX_RING_BLK entries with one ds_add_rtn, writes its records there and
// adds its count to the block's commit counter; the wave whose commit completes a block of PHX_RING_BLK entries flushes it: ONE global
// atomic for exactly PHX_RING_BLK slots and coalesced 16-byte stores.  A block is written again only after its flush (`flushed`, per
// buffer), which the writers of the block after next wait for — they wait for waves with LOWER ring positions only, so there is no cycle —
// and what is left in the ring when the workgroup has shaded its last window goes out with an exact count.  The queues stay dense; their
// ORDER changes (blocks of 256 in completion order), which no result depends on (one closest-hit ray and at most one shadow ray per path
// and step; the film add is per path).  LDS: 16 KB (survivors, 32 B) + 24 KB (NEE rays, 48 B) per workgroup, two workgroups per CU.
// Ordering: the LDS executes the DS instructions of one wave in issue order, so "records, then commit" and "reads, then release" need no
// fence in hardware; s_waitcnt lgkmcnt(0) + a compiler barrier keep the compiler (and any doubt) out.  NOT __builtin_amdgcn_fence: a
// workgroup-scope fence also waits for the wave's global loads — the next round's records, requested right before the append.
#ifndef PHX_SHADE_RING
#define PHX_SHADE_RING 1
#endif
#ifndef PHX_SHADE_DYN_SLICES
#define PHX_SHADE_DYN_SLICES 1  /* with the ring: a wave takes the next 64 sorted slots of the window from an LDS counter instead of slots [k x BLOCK + 64 w, + 64) of round k (nothing orders the waves inside a window any more, so the fast ones take more) */
#endif
#ifndef PHX_RING_BLK
#define PHX_RING_BLK 256u  /* entries per flush = per global atomic (the counters sustain ~80 returning atomics per us and address: 256 keeps the kernel near 50) */
#endif
#ifndef PHX_RING_SPINS
#define PHX_RING_SPINS (1u << 19)
#endif
struct RingCtl { uint32_t head, committed[2], flushed[2], dead; };  // dead: a wait of this workgroup timed out — its later appends are dropped, the frame fails
#define PHX_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int NREC>
__device__ __forceinline__ void ring_flush(const float4* ring /* [NREC][2 x BLK] */, uint32_t first, uint32_t n, uint32_t* gcounter, float4* g0, float4* g1, float4* g2) {
  const uint32_t lane = __lane_id();
  uint32_t gb = 0;
  if (lane == 0) gb = atomicAdd(gcounter, n);
  gb = PHX_UNI(gb);
  for (uint32_t k = lane; k < n; k += 64u) {
    g0[gb + k] = ring[first + k];
    g1[gb + k] = ring[2u * PHX_RING_BLK + first + k];
    if (NREC == 3) g2[gb + k] = ring[4u * PHX_RING_BLK + first + k];
  }
}
// The two queues are appended ONE AFTER THE OTHER, each completely (reserve, records, commit, flush if this wave completed a block) before the
// next begins.  Interleaving them — both reservations, then both waits, both sets of records, both commits: three LDS round trips instead of
// six — was built in round 6 and DEADLOCKED on the closed showroom: a wave then holds an uncommitted reservation in one ring while it waits
// for a block of the other, and the two rings do not order the waves alike — inside a burst of eight waves (512 entries = the whole ring)
// wave 1 can be ahead of wave 3 in ring A and behind it in ring B, each waiting for the block the other has not committed (state dumped at
// the time-out: profiles/r06_g_ring_merged_deadlock.log).  One ring at a time, a wave waits only for LOWER positions of the SAME ring and
// holds nothing else: a total order, no cycle.
template <int NREC>
__device__ __forceinline__ void ring_append(bool want, const float4& r0, const float4& r1, const float4& r2, RingCtl* ctl, float4* ring,
                                            uint32_t* gcounter, float4* g0, float4* g1, float4* g2, unsigned long long* watchdog) {
  constexpr uint32_t BLK = PHX_RING_BLK;
  static_assert((BLK & (BLK - 1u)) == 0u && BLK >= 64u, "a wave's reservation spans at most two blocks");
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return;  // wave-uniform
  const uint32_t lane = __lane_id(), n = (uint32_t)__popcll(mask);
  uint32_t pos = 0;
  if (lane == 0) pos = __hip_atomic_fetch_add(&ctl->head, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  pos = PHX_UNI(pos);
  const uint32_t gen0 = pos / BLK, gen1 = (pos + n - 1u) / BLK;  // generation g lives in buffer g & 1; it is that buffer's (g >> 1)-th use
  // (a wait is rare: the block after next fills a whole round later than this one's flush starts.  Every wait is bounded — PHX_RING_SPINS
  // sleeps of 64 clocks, ~30 ms — after which the wave counts itself in DevStats::watchdog and goes on: the frame is then reported as failed
  // (device.cpp), exactly as for k_trace's watchdog; no wave can spin for ever.)
  uint32_t spins = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t gen = h ? gen1 : gen0;
    if (h && gen1 == gen0) break;
    while (PHX_UNI(__hip_atomic_load(&ctl->flushed[gen & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < (gen >> 1)) {
      if (++spins > PHX_RING_SPINS || PHX_UNI(__hip_atomic_load(&ctl->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
        if (lane == 0) { atomicAdd(watchdog, 1ull); __hip_atomic_store(&ctl->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  asm volatile("" ::: "memory");
  if (want) {
    const uint32_t idx = (pos + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) & (2u * BLK - 1u);
    ring[idx] = r0; ring[2u * BLK + idx] = r1;
    if (NREC == 3) ring[4u * BLK + idx] = r2;
  }
  PHX_LDS_ORDER();  // the records are in LDS before the commit
  const uint32_t n0 = min(n, (gen0 + 1u) * BLK - pos), n1 = n - n0;
  uint32_t c0 = 0, c1 = 0;
  if (lane == 0) {
    c0 = __hip_atomic_fetch_add(&ctl->committed[gen0 & 1u], n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (n1) c1 = __hip_atomic_fetch_add(&ctl->committed[gen1 & 1u], n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  c0 = PHX_UNI(c0); c1 = PHX_UNI(c1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t gen = h ? gen1 : gen0;
    if (h ? (n1 != 0u && c1 + n1 == BLK) : (c0 + n0 == BLK)) {  // this wave's commit completed the block: it flushes it
      asm volatile("" ::: "memory");
      ring_flush<NREC>(ring, (gen & 1u) * BLK, BLK, gcounter, g0, g1, g2);
      PHX_LDS_ORDER();  // the block has been read before it is handed back
      if (lane == 0) {
        __hip_atomic_store(&ctl->committed[gen & 1u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PHX_LDS_ORDER();
        __hip_atomic_fetch_add(&ctl->flushed[gen & 1u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}
template <bool PERHIT /* some material's closure weights depend on the hit (glass) */, bool FIRST, bool LENS = false /* FIRST: thin-lens camera */>
__global__ void __launch_bounds__(PHX_SHADE_BLOCK_G) __attribute__((amdgpu_waves_per_eu(PHX_SHADE_WAVES_G, 8))) k_shade_g(DevScene sc, PassBuffers pb, int q, int sq, uint32_t sample0) {
  constexpr int BLOCK = PHX_SHADE_BLOCK_G, ITEMS = PHX_SHADE_ITEMS_G, WINDOW = BLOCK * ITEMS, NB = PHX_SHADE_BUCKETS;
  static_assert(WINDOW <= 65536 && BLOCK >= NB + 2 && NB == 64, "perm holds 16-bit positions; one wave scans the NB material buckets");
#if PHX_SHADE_RING
  __shared__ float4 ring_a[3 * 2 * PHX_RING_BLK];  // survivors: (o, path | SPECULAR << 31), (d, RNG key), (beta, depth)
  __shared__ float4 ring_b[3 * 2 * PHX_RING_BLK];  // NEE rays: (o, path), (d, tmax), (beta * Li)
  __shared__ RingCtl ring_ctl[2];
  __shared__ uint32_t slice_next;  // PHX_SHADE_DYN_SLICES: the window's next 64-slot slice
  if (threadIdx.x < sizeof(ring_ctl) / 4u) reinterpret_cast<uint32_t*>(ring_ctl)[threadIdx.x] = 0u;  // visible after the first barrier every thread reaches
#else
  __shared__ uint32_t lds_sr[2 * ((BLOCK >> 6) + 1)];
#endif
  __shared__ uint32_t bucket[NB + 2];  // [material mod NB], [NB] misses, [NB + 1] slots past the end of the queue
  __shared__ uint16_t perm[WINDOW];
  if (pb.stats->watchdog | pb.stats->ring_watchdog) return;  // (k_shade above: a step whose trace did not finish is not shaded; the frame fails)
  constexpr bool TRI_LDS = PHX_SHADE_TRI_LDS != 0;
  __shared__ uint32_t tri_sorted[TRI_LDS ? WINDOW : 1];  // the hit's pool index (0xffffffff = miss) at its sorted position
  const uint32_t count = pb.counters[q * CNT_STRIDE];
  if (blockIdx.x == 0 && threadIdx.x == 0) zero_cursors(pb.counters);  // the next k_trace pulls its chunks from here
  PHX_PHASE_DECL
#if PHX_SHADE_TIMING
  unsigned long long ph_rounds = 0, ph_windows = 0;
#endif
  // The sort keys of a window: two dependent gathers per hit — the hit record's triangle, that triangle's material.  (Fetching them one
  // window AHEAD — the triangles while the last round of the previous window waits in its append, the materials right after it, parked in
  // 4 KB of LDS — changed nothing: 34.6-35.1 ms either way, profiles/r05_c_shade_prefetch3_ab.log.)  0xfffffffe = slot past the end of the queue.
  auto request_tris = [&](uint32_t b, uint32_t (&tri)[ITEMS]) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t i = b + k * BLOCK + threadIdx.x;
      tri[k] = i < count ? f2u(pb.hit[i].w) : 0xfffffffeu;
    }
  };
  auto request_keys = [&](const uint32_t (&tri)[ITEMS], uint32_t (&key)[ITEMS]) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
#if PHX_SHADE_KEY_PROBE
      key[k] = tri[k] == 0xfffffffeu ? NB + 1u : tri[k] != 0xffffffffu ? 0u : (uint32_t)NB;  // probe builds only: what the material gather of the sort phase costs (one-material scenes)
#else
      key[k] = tri[k] == 0xfffffffeu ? NB + 1u : tri[k] != 0xffffffffu ? (f2u(sc.elem_shade[tri[k]].w) & (NB - 1u)) : (uint32_t)NB;  // the material word of the 16-byte shade record (four to a sector; the shading rounds read the same records)
#endif
    }
  };
  for (uint32_t base = blockIdx.x * WINDOW; base < count; base += gridDim.x * WINDOW) {
    PHX_PHASE(5)  // (the barrier at the end of the previous window, loop overhead)
    // ---- counting sort of the window by material, through LDS
    if (threadIdx.x < NB + 2) bucket[threadIdx.x] = 0;
#if PHX_SHADE_RING && PHX_SHADE_DYN_SLICES
    if (threadIdx.x == 0) slice_next = 0u;
#endif
    __syncthreads();
    uint32_t keys[ITEMS], tri_[ITEMS];
    request_tris(base, tri_);
    if constexpr (FIRST && !PHX_SHADE_FIRST_SORT) {
      // the camera entries: queue order IS pixel order (a wave = 64 samples of one pixel: one material, a few on a silhouette) — no sort, the
      // identity permutation; the slots past the end of the queue are the window's last ones as they are after a sort
      if constexpr (TRI_LDS) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) tri_sorted[k * BLOCK + threadIdx.x] = tri_[k];
      }
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) perm[k * BLOCK + threadIdx.x] = (uint16_t)(k * BLOCK + threadIdx.x);
    } else {
    request_keys(tri_, keys);
    uint32_t ranks[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) ranks[k] = atomicAdd(&bucket[keys[k]], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the bucket counts by one wave
      const uint32_t c = bucket[threadIdx.x];
      uint32_t incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += up; }
      bucket[threadIdx.x] = incl - c;
      if (threadIdx.x == 63) { const uint32_t misses = bucket[NB]; bucket[NB] = incl; bucket[NB + 1] = incl + misses; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t at = bucket[keys[k]] + ranks[k];
      perm[at] = (uint16_t)(k * BLOCK + threadIdx.x);
      if constexpr (TRI_LDS) tri_sorted[at] = tri_[k];
    }
    }
    __syncthreads();
    PHX_PHASE(0)  // the window's sort by material
#if PHX_SHADE_TIMING
    ++ph_windows;
#endif
    // ---- the window in sorted order: wave w of round k shades sorted positions [k * BLOCK + 64 w, + 64)
    // The records of round k + 1 — hit, ray — are REQUESTED right before round k's append (PHX_SHADE_PREFETCH): the phase probe of round 5
    // (profiles/r05_c_shade_phases.md) found a wave waiting 27 % of its time for exactly these loads and 24 % in the append (two barriers
    // and the round trip of the workgroup's atomics); at the append a thread holds almost nothing but its outputs, so the twelve registers of
    // the next records cost no occupancy there, and the two waits overlap.  (Round 3 requested them at the START of round k and held them
    // across the closure code: 5-12 VGPRs where the kernel has none to spare, 1.2 of 42.7 ms: profiles/r03_q_prefetch_ab.log.)
    uint32_t next_i = 0, next_tri = 0xffffffffu; bool next_live = false;
    float4 next_h = make_float4(0.f, 0.f, 0.f, 0.f), next_a = next_h, next_b = next_h, next_S = next_h;
    constexpr bool DYN = PHX_SHADE_RING && PHX_SHADE_DYN_SLICES;
    // DYN: slices of 64 sorted slots, handed out by an LDS counter; the live slots are a prefix of the sorted window, so are the live slices
    const uint32_t nslices = (min((uint32_t)WINDOW, count - base) + 63u) >> 6;
    auto take_slice = [&]() -> uint32_t {
#if PHX_SHADE_RING && PHX_SHADE_DYN_SLICES
      uint32_t s_ = 0;
      if (__lane_id() == 0) s_ = __hip_atomic_fetch_add(&slice_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return PHX_UNI(s_);
#else
      return 0u;
#endif
    };
    auto sorted_slot = [&](uint32_t k) { return DYN ? k * 64u + __lane_id() : k * BLOCK + threadIdx.x; };  // k: slice (DYN) or round
    auto request_round = [&](uint32_t k) {
      next_live = false;
      if (DYN ? k < nslices : (k < (uint32_t)ITEMS && base + k * BLOCK < count)) {
        next_i = base + perm[sorted_slot(k)];
        next_live = next_i < count;
        if (next_live) {
          next_h = pb.hit[next_i];
          if (!FIRST) { next_a = pb.ro[q][next_i]; next_b = pb.rd[q][next_i]; }
          if constexpr (TRI_LDS) {  // the shade record no longer waits for the hit record: its index is in LDS (four registers across the append)
            next_tri = tri_sorted[sorted_slot(k)];
            if (next_tri != 0xffffffffu) next_S = sc.elem_shade[next_tri];
          }
        }
      }
    };
    // second stage (PHX_SHADE_PREFETCH 2): requested right after the append, travelling while this round's queue entries are stored: the path
    // state — since round 6 a queue record like the ray, no longer a gather by path id: four registers that need not live across the append —
    // and, in builds without PHX_SHADE_TRI_LDS, the shade record, whose index then has to wait for the hit record
    float4 next_bd = make_float4(1.0f, 1.0f, 1.0f, u2f(0u));  // FIRST: state_t::reset: beta = 1, depth = 0
    auto request_round_dependents = [&]() {
      if (next_live) {
        if (!FIRST) next_bd = pb.qs[q][next_i];
        const uint32_t tri = f2u(next_h.w);
        if constexpr (!TRI_LDS) { if (tri != 0xffffffffu) next_S = sc.elem_shade[tri]; }
      }
    };
    // (with per-hit closure weights — glass — either stage costs the kernel 16 B of scratch and buys nothing: 41.7-42.1 ms with, 41.9-42.2
    // without on the glass showroom, profiles/r05_c_shade_prefetch_glass_ab.log: those instantiations request where they consume)
    constexpr bool STAGE1 = PHX_SHADE_PREFETCH >= 1 && (!PERHIT || PHX_SHADE_PREFETCH_PERHIT >= 1), STAGE2 = PHX_SHADE_PREFETCH >= 2 && (!PERHIT || PHX_SHADE_PREFETCH_PERHIT >= 2);
    uint32_t k = DYN ? take_slice() : 0u, k_next = 0u;
    if constexpr (STAGE1) request_round(k);
    if constexpr (STAGE2) request_round_dependents();
    for (;; k = k_next) {
      // wave-uniform (DYN) / workgroup-uniform: the slots past the end of the queue sort behind every live one
      if (DYN ? k >= nslices : (k >= (uint32_t)ITEMS || base + k * BLOCK >= count)) break;
      // (instantiations without a stage request each record where it is consumed, exactly as the round-4 kernel did)
      if constexpr (STAGE1 && !STAGE2) request_round_dependents();
      const uint32_t i = STAGE1 ? next_i : base + perm[sorted_slot(k)];
      const bool live = STAGE1 ? next_live : i < count;
      // Live ranges are kept short on purpose (the kernel is register-bound: 128 VGPRs as one block of code): radiance and the
      // normals channel are written as soon as the hit is known; the light's record is re-read after the closure evaluation instead
      // of being held across it.
      bool alive = false, want_shadow = false, hit_surface = false;
      uint32_t path = 0, depth = 0, key = 0;
      v3 p, n, wo, beta;
      uint32_t mat = 0;  // the hit's material: an INDEX — bsdf_f / bsdf_sample read the recipe through the scalar cache, one distinct material of the wave at a time
      if (live) {
        float4 a, b, bd;
        float4 h;
        if constexpr (STAGE1) h = next_h; else h = pb.hit[i];
        // TRI_LDS: the hit's pool index comes from the sort phase (LDS), so its shade record and vertex normals are requested HERE, beside the
        // hit record and the ray, not after the hit record has landed (one memory round trip less on the critical path of a round)
        uint32_t tri_early = 0xffffffffu; float4 S_early = make_float4(0.f, 0.f, 0.f, 0.f); VertexNormals VN_early{v3(0.0f), v3(0.0f), v3(0.0f)};
        if constexpr (TRI_LDS && STAGE1) {
          tri_early = next_tri; S_early = next_S;
          if (tri_early != 0xffffffffu) VN_early = request_vertex_normals(sc, tri_early);
        } else if constexpr (TRI_LDS) {
          tri_early = tri_sorted[sorted_slot(k)];
          if (tri_early != 0xffffffffu) { S_early = sc.elem_shade[tri_early]; VN_early = request_vertex_normals(sc, tri_early); }
        }
        if (FIRST) {
          v3 co, cd;
          camera_ray<LENS>(sc, pb, i, sample0, co, cd);
          a = make_float4(co.x, co.y, co.z, u2f(i)); b = make_float4(cd.x, cd.y, cd.z, FLT_MAX);
          bd = make_float4(1.0f, 1.0f, 1.0f, u2f(0u));  // state_t::reset: beta = 1, depth = 0
        } else {
          if constexpr (STAGE1) { a = next_a; b = next_b; bd = next_bd; } else { a = pb.ro[q][i]; b = pb.rd[q][i]; bd = pb.qs[q][i]; }
        }
        const uint32_t pbits = f2u(a.w);
        path = pbits & 0x7fffffffu;
        const bool specular = (pbits >> 31) != 0;
        beta = v3(bd.x, bd.y, bd.z);
        depth = f2u(bd.w);
        if (FIRST) {  // the path's RNG key: computed here once, then carried in the ray record's fourth word (k_shade above)
          const uint32_t pix = path / pb.num_samples, s = path - pix * pb.num_samples;
          const uint32_t xy = pb.pix_xy[pix];
          key = path_key(pb.seed, (xy >> 16) * sc.width + (xy & 0xffffu), sample0 + s);
        } else {
          key = f2u(b.w);
        }
        const uint32_t tri = TRI_LDS ? tri_early : f2u(h.w);
        const v3 o(a.x, a.y, a.z), d(b.x, b.y, b.z);
        v3 add_e(0.0f); bool add_rad = false;
        if (tri != 0xffffffffu) {
          hit_surface = true;
          VertexNormals VN;
          float4 S;  // (geometric normal, material | smooth << 31): DevScene::elem_shade
          if constexpr (TRI_LDS) { VN = VN_early; S = S_early; }
          else { VN = request_vertex_normals(sc, tri); if constexpr (STAGE1) S = next_S; else S = sc.elem_shade[tri]; }
          const uint32_t pm = f2u(S.w);
          p = o + d * h.x;            // hits.p = p + wi*d
          wo = -d;                    // hits.wi = -wi
          n = shading_normal(VN, (pm >> 31) != 0, v3(S.x, S.y, S.z), h.y, h.z);
          mat = pm & 0x7fffffffu;  // material_t::evaluate (material.cpp:419-458): the closure recipe at this hit
          if (pb.pn && (FIRST || depth == 0)) pb.pn[path] = make_float4(n.x, n.y, n.z, 1.0f);
          if (depth == 0 || specular) { const DevMaterial& m = sc.materials[mat]; add_e = v3(m.ex, m.ey, m.ez); add_rad = true; }  // spt.hpp:177-179
        } else {
          // miss: environment lighting (deferred_shading_kernel.hpp:65-70, spt.hpp:199-202); the slot is MASKED|SHADOW in the
          // reference's shadow stream (spt.hpp:138-141): rays_masked = rays_closest - rays_shadow
          if (sc.env_material >= 0) { const DevMaterial& m = sc.materials[sc.env_material]; add_e = v3(m.ex, m.ey, m.ez); }
          add_rad = true;
          if (FIRST && pb.pn) pb.pn[path] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        // out += beta * e with the beta the ray arrived with.  Radiance is only read-modified-written when this step adds something:
        // with e == 0 and a finite beta `out` is unchanged bit for bit (it is never -0)
        if (FIRST) {  // r = 0 + beta * e: every path's radiance is written here, nothing is read
          const v3 rad = v3(0.0f) + beta * add_e;
          pb.pr[path] = make_float4(rad.x, rad.y, rad.z, 0.0f);
        } else if (add_rad && !(isfinite(beta.x) && isfinite(beta.y) && isfinite(beta.z) && add_e.x == 0.0f && add_e.y == 0.0f && add_e.z == 0.0f)) {
          const float4 rr = pb.pr[path];
          const v3 rad = v3(rr.x, rr.y, rr.z) + beta * add_e;
          pb.pr[path] = make_float4(rad.x, rad.y, rad.z, 0.0f);
        }
      }
      PHX_PHASE(1)  // hit record, ray, path state, shade record, normals: requested and landed; emission added
      // the hit's tangent frame (orthogonal_base_t): once per hit, for the NEE evaluation and the BSDF sample
      const Frame fr(hit_surface ? n : v3(0.0f, 1.0f, 0.0f));
      // ---- next-event estimation: sampler_t::fresh_light_samples + light_sampler_t (sampling.cpp:160-179, spt.hpp:95-149)
      v3 sh_o, sh_d, contrib; float sh_t = 0.0f;  // the NEE ray waits in registers for the survivor: both queues are appended in one go below
      {
        if (hit_surface) {
          const uint32_t b0 = depth * DIMS_PER_STEP;
          const float pick = draw_f32(key, b0 + DIM_LIGHT_PICK), lu = draw_f32(key, b0 + DIM_LIGHT_U), lv = draw_f32(key, b0 + DIM_LIGHT_V);
          const float nlf = (float)sc.num_lights;
          uint32_t l = (uint32_t)floorf(pick * nlf);
          if (l > sc.num_lights - 1) l = sc.num_lights - 1;
          const uint32_t ltris = sc.lights[l].num_tris;
          const float numf = (float)ltris;
          uint32_t ti = (uint32_t)floorf(lu * numf);  // uniform by index (light.cpp:55), pdf = 1/area
          if (ti > ltris - 1) ti = ltris - 1;
          const float remapped = fminf(lu * numf - (float)ti, 1.0f - FLT_EPSILON);
          uint32_t lt = sc.lights[l].first_tri + ti;
          const float x = sqrtf(remapped);
          const float bu = 1 - x, bv = lv * x;     // triangle_t::sample, mesh.cpp:318-324
          {
            const DevLightTri& LT = sc.light_tris[lt];
            const v3 la(LT.ax, LT.ay, LT.az), lb(LT.bx, LT.by, LT.bz), lc(LT.cx, LT.cy, LT.cz);
            const v3 P = bu * la + bv * lb + (1 - bu - bv) * lc;
            sh_o = v3(p.x + n.x * 0.0001f, p.y + n.y * 0.0001f, p.z + n.z * 0.0001f);
            sh_d = P - sh_o;
          }
          const float l2 = sdot(sh_d, sh_d);
          sh_t = sqrtf(l2) - 0.0001f;
          const float oolen = 1.0f / sqrtf(l2);
          sh_d = v3(sh_d.x * oolen, sh_d.y * oolen, sh_d.z * oolen);
          if (sdot(n, sh_d) >= 0.0f) {
            // li(), spt.hpp:212-255 — evaluated before the occlusion test; k_trace adds it if the ray is unoccluded
            v3 f(0.0f);
            if constexpr (PHX_SCALAR_F && (!PERHIT || PHX_SCALAR_F_PERHIT)) {
              PHX_FOR_EACH_MATERIAL_OF_THE_WAVE(mat, cm, f = (bsdf_f<false, 8, PERHIT>(cm, n, fr, sh_d, wo)));
            } else {
              f = bsdf_f<false, 8, PERHIT>(sc.materials[mat], n, fr, sh_d, wo);
            }
            // the light's record again (L1-resident), behind an empty asm so that the first read is not kept alive across bsdf_f
            asm volatile("" : "+v"(l), "+v"(lt));
            const DevLight& L = sc.lights[l];
            const DevLightTri& LT = sc.light_tris[lt];
            const v3 ln = LT.smooth ? shading_normal(sc, LT.prim, true, v3(LT.bx - LT.ax, LT.by - LT.ay, LT.bz - LT.az), v3(LT.cx - LT.ax, LT.cy - LT.ay, LT.cz - LT.az), bu, bv)
                                    : v3(LT.nx, LT.ny, LT.nz);
            const v3 le(L.ex, L.ey, L.ez);
            const float pdf = L.lpdf * sh_t * sh_t / fabsf(dot(ln, -sh_d));
            const v3 li = ((le * 4.0f) * f) * (1.0f / pdf);
            contrib = beta * li;
            want_shadow = true;
          }
        }
      }
      PHX_PHASE(2)  // next-event estimation: light sample, bsdf_f, li
      // ---- integrate: ++depth, russian roulette, bsdf sampling (spt.hpp:188-190, 257-328)
      {
        v3 nxt_d; uint32_t next_specular = 0; float off = 0.0f;
        if (hit_surface) {
          depth += 1;
          float wgt = 1.0f;
          alive = depth < sc.max_depth;
          if (alive && depth >= 3) {
            const float qq = fmaxf(0.05f, 1.0f - luminance(beta));
            const float xi = draw_f32(key, (depth - 1u) * DIMS_PER_STEP + DIM_RR);
            alive = xi >= qq;
            if (alive) wgt = (1.0f / (1.0f - qq));
          }
          beta = beta * wgt;
          if (alive) {
            const uint32_t b1 = (depth - 1u) * DIMS_PER_STEP;
            float pdf; uint32_t fl;
            const float u1 = draw_f32(key, b1 + DIM_BSDF_U), u2 = draw_f32(key, b1 + DIM_BSDF_V);
#if PHX_SCALAR_S
            v3 f(0.0f);
            PHX_FOR_EACH_MATERIAL_OF_THE_WAVE(mat, cm, f = (bsdf_sample<false, 8, PERHIT>(cm, n, fr, u1, u2, wo, nxt_d, pdf, fl)));
#else
            const v3 f = bsdf_sample<false, 8, PERHIT>(sc.materials[mat], n, fr, u1, u2, wo, nxt_d, pdf, fl);
#endif
            if ((f.x == 0.0f && f.y == 0.0f && f.z == 0.0f) || pdf == 0.0f) {
              alive = false;
            } else {
              const float weight = dot(n, nxt_d);
              beta = beta * (f * (fabsf(weight) / pdf));
              off = (weight < 0.0f) ? -0.0001f : 0.0001f;
              next_specular = (fl & B_SPECULAR) ? 1u : 0u;
            }
          }
        }
        // both queues in one go: two barriers and two concurrent atomics per round (appending the NEE ray before roulette and sampling
        // — shorter live ranges, four barriers, two atomics in a row — was right while the kernel fought for occupancy; at 4 waves per SIMD
        // either way, the 10 registers are free and the round trip is not: 43.1 -> 41.5 ms, profiles/r03_zzc_append2_ab.log)
        PHX_PHASE(3)  // roulette, bsdf_sample
        k_next = DYN ? take_slice() : k + 1u;
        if constexpr (STAGE1) request_round(k_next);  // in flight across the append (the append waits for LDS traffic only)
#if PHX_SHADE_RING
        {
          const v3 nxt_o = p + n * off;
          ring_append<3>(want_shadow, make_float4(sh_o.x, sh_o.y, sh_o.z, u2f(path)), make_float4(sh_d.x, sh_d.y, sh_d.z, sh_t), make_float4(contrib.x, contrib.y, contrib.z, 0.0f),
                         &ring_ctl[1], ring_b, &pb.counters[CNT_SHADOW + sq * CNT_STRIDE], pb.so, pb.sd, pb.sc, &pb.stats->ring_watchdog);
          ring_append<3>(alive, make_float4(nxt_o.x, nxt_o.y, nxt_o.z, u2f(path | (next_specular << 31))), make_float4(nxt_d.x, nxt_d.y, nxt_d.z, u2f(key)),
                         make_float4(beta.x, beta.y, beta.z, u2f(depth)),  // the path's state travels with its ray: only the next shade of a surviving path reads it
                         &ring_ctl[0], ring_a, &pb.counters[(q ^ 1) * CNT_STRIDE], pb.ro[q ^ 1], pb.rd[q ^ 1], pb.qs[q ^ 1], &pb.stats->ring_watchdog);
        }
        PHX_PHASE(4)  // the append: slot reservation, records to LDS, commit; for one wave in BLK / 64 rounds the flush of a block
        if constexpr (STAGE2) request_round_dependents();
#else
        uint32_t no, ns;
        #if PHX_SHADE_TIMING
        block_append2<BLOCK>(alive, &pb.counters[(q ^ 1) * CNT_STRIDE], want_shadow, &pb.counters[CNT_SHADOW + sq * CNT_STRIDE], lds_sr, no, ns, &pb.stats->idle_lane_iters);
#else
        block_append2<BLOCK>(alive, &pb.counters[(q ^ 1) * CNT_STRIDE], want_shadow, &pb.counters[CNT_SHADOW + sq * CNT_STRIDE], lds_sr, no, ns);
#endif
        PHX_PHASE(4)  // the append: two barriers and the workgroup's two atomics on the queue counters
        if constexpr (STAGE2) request_round_dependents();
        if (want_shadow) {
          pb.so[ns] = make_float4(sh_o.x, sh_o.y, sh_o.z, u2f(path));
          pb.sd[ns] = make_float4(sh_d.x, sh_d.y, sh_d.z, sh_t);
          pb.sc[ns] = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
        }
        if (alive) {
          const v3 nxt_o = p + n * off;
          pb.ro[q ^ 1][no] = make_float4(nxt_o.x, nxt_o.y, nxt_o.z, u2f(path | (next_specular << 31)));
          pb.rd[q ^ 1][no] = make_float4(nxt_d.x, nxt_d.y, nxt_d.z, u2f(key));
          pb.qs[q ^ 1][no] = make_float4(beta.x, beta.y, beta.z, u2f(depth));
        }
#endif
        PHX_PHASE(5)  // the stores of the two queue entries (waited for: the probe charges them here, the product build does not wait)
#if PHX_SHADE_TIMING
        ++ph_rounds;
#endif
      }
    }
    __syncthreads();  // perm and bucket are rewritten by the next window
  }
#if PHX_SHADE_RING
  // what the workgroup's last windows left in the rings: every complete block has been flushed by the wave that completed it (before that
  // wave reached the barrier above); the open block goes out with its exact count, the survivors' by wave 0, the NEE rays' by wave 1
  __syncthreads();
  if (threadIdx.x < 128u) {
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t head = ring_ctl[w].head, left = head & (PHX_RING_BLK - 1u), first = ((head / PHX_RING_BLK) & 1u) * PHX_RING_BLK;
    // (a ring that timed out is NOT flushed: a wave that gave up has reserved slots it never wrote, and whatever LDS held there — records of two
    // blocks ago, or nothing at all — must not reach a queue, where a path id is an index.  The frame is reported as failed anyway.  The
    // ring-watchdog twin library faulted the GPU on exactly this until the guard was added: tests/test_gpu_parity.py::test_append_ring_timeout_…)
    if (left && !ring_ctl[w].dead) {
      if (w == 0u) ring_flush<3>(ring_a, first, left, &pb.counters[(q ^ 1) * CNT_STRIDE], pb.ro[q ^ 1], pb.rd[q ^ 1], pb.qs[q ^ 1]);
      else ring_flush<3>(ring_b, first, left, &pb.counters[CNT_SHADOW + sq * CNT_STRIDE], pb.so, pb.sd, pb.sc);
    }
  }
#endif
#if PHX_SHADE_TIMING
  if ((threadIdx.x & 63u) == 0u) {  // probe build only: s_memtime ticks per phase, summed over the waves (DevStats fields of the count build)
    for (int k = 0; k < 6; ++k) atomicAdd(&pb.stats->stack_pushes[k], ph_acc[k]);
    atomicAdd(&pb.stats->stack_pushes[6], ph_rounds); atomicAdd(&pb.stats->stack_pushes[7], ph_windows);
  }
#endif
}

// ---- film -------------------------------------------------------------------------------------------
// channels.primary->add(x, y, r * (1.0f / (spp * pps))) per sample, IN SAMPLE ORDER (cpu.cpp:175-198): the sum of a pixel is
// a serial chain, so it cannot be a lane-parallel reduction.  Radiance is stored pixel-major (pix * S + s): a 256-thread
// block moves 64 pixels x 16 samples at a time through LDS — read as 256-B runs of one pixel's samples, summed by the 64
// threads that own a pixel each (row pitch 17 float4: conflict-free) — instead of every thread striding 16*S bytes.
#define PHX_FILM_PIX 64
#define PHX_FILM_SMP 16
__global__ void __launch_bounds__(256) k_film(PassBuffers pb, uint32_t num_samples, float inv) {
  __shared__ float4 tile[PHX_FILM_PIX * (PHX_FILM_SMP + 1)];
  const uint32_t pix0 = blockIdx.x * PHX_FILM_PIX;
  const uint32_t my_pix = pix0 + threadIdx.x;
  const bool owner = threadIdx.x < PHX_FILM_PIX && my_pix < pb.num_pixels;
  float r = 0.0f, g = 0.0f, b = 0.0f;
  float* out = pb.acc + (size_t)my_pix * pb.xstride;
  if (owner) { r = out[0]; g = out[1]; b = out[2]; }
  float nx = 0.0f, ny = 0.0f, nz = 0.0f; bool have_n = false;
  const int nsrc = pb.pn ? 2 : 1;
  for (uint32_t s0 = 0; s0 < num_samples; s0 += PHX_FILM_SMP) {
    for (int src = 0; src < nsrc; ++src) {
      const float4* __restrict__ in = src ? pb.pn : pb.pr;
      __syncthreads();  // the previous tile has been consumed
#pragma unroll
      for (int k = 0; k < PHX_FILM_PIX * PHX_FILM_SMP / 256; ++k) {
        const uint32_t e = k * 256 + threadIdx.x, p = e / PHX_FILM_SMP, sm = e % PHX_FILM_SMP;
        if (pix0 + p < pb.num_pixels && s0 + sm < num_samples) tile[p * (PHX_FILM_SMP + 1) + sm] = in[(size_t)(pix0 + p) * num_samples + s0 + sm];
      }
      __syncthreads();
      if (owner) {
        const uint32_t cnt = min((uint32_t)PHX_FILM_SMP, num_samples - s0);
        for (uint32_t sm = 0; sm < cnt; ++sm) {
          const float4 c = tile[threadIdx.x * (PHX_FILM_SMP + 1) + sm];
          if (src == 0) { r += c.x * inv; g += c.y * inv; b += c.z * inv; }
          else if (c.w != 0.0f) { nx = c.x; ny = c.y; nz = c.z; have_n = true; }  // the last sample with a primary hit wins
        }
      }
    }
  }
  if (owner) {
    out[0] = r; out[1] = g; out[2] = b;
    if (have_n) { out[pb.normals_offset] = nx; out[pb.normals_offset + 1] = ny; out[pb.normals_offset + 2] = nz; }
  }
}

// batch render buffer -> full-frame device film (film_t::add_tile for a device-resident sink)
__global__ void __launch_bounds__(PHX_BLOCK) k_scatter_film(PassBuffers pb, float* film, uint32_t film_width) {
  const uint32_t pix = blockIdx.x * PHX_BLOCK + threadIdx.x;
  if (pix >= pb.num_pixels) return;
  const uint32_t xy = pb.pix_xy[pix];
  const float* src = pb.acc + (size_t)pix * pb.xstride;
  float* dst = film + ((size_t)(xy >> 16) * film_width + (xy & 0xffffu)) * pb.xstride;
  for (uint32_t c = 0; c < pb.xstride; ++c) dst[c] = src[c];
}

// ---- KAT kernels ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_bsdf_f(const DevMaterial* mat, uint32_t n, const float* n3, const float* wi3, const float* wo3, float* f3) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const v3 nn(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]), view(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);  // f(wi = to the light, wo = hits.wi)
  const v3 f = bsdf_f<false, 8, true>(*mat, nn, v3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), view);
  f3[3 * i] = f.x; f3[3 * i + 1] = f.y; f3[3 * i + 2] = f.z;
}
__global__ void __launch_bounds__(64) k_bsdf_sample(const DevMaterial* mat, uint32_t n, const float* n3, const float* wi3, const float* u2,
                              float* wo3, float* f3, float* pdf, uint32_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  v3 wo; float p; uint32_t fl;
  const v3 nn(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]), view(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);  // sample(u, wi = hits.wi)
  v3 f = bsdf_sample<false, 8, true>(*mat, nn, u2[2 * i], u2[2 * i + 1], view, wo, p, fl);
  if (p == 0.0f) { wo = v3(0.0f); f = v3(0.0f); fl = 0; }
  wo3[3 * i] = wo.x; wo3[3 * i + 1] = wo.y; wo3[3 * i + 2] = wo.z;
  f3[3 * i] = f.x; f3[3 * i + 1] = f.y; f3[3 * i + 2] = f.z; pdf[i] = p; flags[i] = fl;
}

// ---- launches ---------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint32_t n) { return (n + PHX_BLOCK - 1) / PHX_BLOCK; }
void launch_begin_pass(hipStream_t stream, const PassBuffers& pb, uint32_t num_samples) {
  hipLaunchKernelGGL(k_begin_pass, dim3(1), dim3(1), 0, stream, pb, num_samples);
}
namespace {
struct TraceEnv {
  int grid_mul, wg_cap;
  uint32_t refill, block_env, ntop_env, min_chunks, target_chunks, lds_levels_env;
};
const TraceEnv& trace_env() {
  static const TraceEnv e = [] {
    // knobs are experiment switches, but a typo must not become a launch shape the kernels were never built for: negative values
    // fall back to the default and the workgroup size has to be one the kernels are instantiated for
    auto geti = [](const char* n, int d) { const char* v = getenv(n); const int x = v ? atoi(v) : d; return x < 0 ? d : x; };
    TraceEnv t;
    t.refill = (uint32_t)geti("PHX_REFILL", 12); t.grid_mul = std::max(1, geti("PHX_TRACE_DYN_GRID", 1));
    // 1024-thread workgroups: the CU's LDS holds two copies of the top of the tree instead of eight, so each copy is 4x larger
    t.block_env = (uint32_t)geti("PHX_TRACE_BLOCK", 0); t.ntop_env = (uint32_t)geti("PHX_NTOP", 0);
    if (t.block_env != 0 && t.block_env != 256 && t.block_env != 512 && t.block_env != 1024) t.block_env = 0;
    t.lds_levels_env = (uint32_t)geti("PHX_LDS_LEVELS", 0);  // stack levels kept in LDS (0 = chosen by trace_plan)
    if (t.lds_levels_env > PHX_MAX_BVH_DEPTH) t.lds_levels_env = 0;
    if (t.refill < 1 || t.refill > 64) t.refill = 12;
    t.min_chunks = (uint32_t)geti("PHX_MIN_CHUNKS", 8);
    t.wg_cap = geti("PHX_TRACE_WG_CAP", 0);  // experiment: at most this many k_trace workgroups per CU (leaves wave slots to another stream)
    // 64-ray chunks out of a workgroup's range of 16 (k_trace ms per frame at 100 k: 60.8 — with one global atomic per WAVE and
    // chunk the best was 63.7 at 512 rays: 128 rays 75.7, 256 66.7, 384 64.4, 768 63.9, 1024 64.2; profiles/r02_n_knob_sweep.log,
    // r02_o_wg_cursor.log)
    t.target_chunks = (uint32_t)std::max(1, geti("PHX_TARGET_CHUNKS", 1));
    return t;
  }();
  return e;
}
template <typename F>
void for_each_trace_kernel(F&& f) {
  f(reinterpret_cast<const void*>(&k_trace<256>)); f(reinterpret_cast<const void*>(&k_trace<512>)); f(reinterpret_cast<const void*>(&k_trace<1024>));
  f(reinterpret_cast<const void*>(&k_trace<1024, true>)); f(reinterpret_cast<const void*>(&k_trace<1024, true, true>));
#if PHX_STACK_PACKED >= 2
  f(reinterpret_cast<const void*>(&k_trace<256, false, true>)); f(reinterpret_cast<const void*>(&k_trace<512, false, true>)); f(reinterpret_cast<const void*>(&k_trace<1024, false, true>));
#endif
  f(reinterpret_cast<const void*>(&k_trace_rays<true>)); f(reinterpret_cast<const void*>(&k_trace_rays<false>));
  f(reinterpret_cast<const void*>(&k_trace_primary<1>)); f(reinterpret_cast<const void*>(&k_trace_primary<2>)); f(reinterpret_cast<const void*>(&k_trace_primary<4>));
  f(reinterpret_cast<const void*>(&k_trace_primary<1, true>)); f(reinterpret_cast<const void*>(&k_trace_primary<2, true>)); f(reinterpret_cast<const void*>(&k_trace_primary<4, true>));
}
}  // namespace

bool launch_counts_traversal_work() { return PHX_COUNT != 0; }

// Allow every traversal kernel the CU's full 160 KB of LDS as dynamic shared memory ON THE CURRENT DEVICE.  The attribute is
// per device and per kernel, so the host calls this once for each phx_device it makes (device.cpp: phx_dev_make).
hipError_t init_kernels_on_current_device() {
  hipError_t rc = hipSuccess;
  for_each_trace_kernel([&](const void* k) {
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess && rc == hipSuccess) rc = e;
  });
  return rc;
}

// What a k_trace launch on this scene looks like: workgroup size, nodelets staged in LDS, stack levels, LDS bytes.
TracePlan trace_plan(const DevScene& sc) {
  const TraceEnv& E = trace_env();
  TracePlan P{};
  // stack entries needed = BVH depth - 1: one pending sibling group per level above the deepest node (the root "group" has a
  // single member and the deepest nodes have no inner children); tests/test_host_bvh8.py checks the bound
  P.levels = std::max(2u, sc.stack_levels > 1u ? sc.stack_levels - 1u : 1u);
  // ... of which the top lds_levels live in LDS and the rest in HBM (k_trace<1024, SPILL>).  A lane's stack is rarely deep
  // (100 k triangles, 8 levels: 6.5 pushes per ray, 0.011 of them land at depth 5 or below; 1 M, 9 levels: 0.013 at depth 6 or
  // below), but the test on every push and pop costs what the freed LDS buys back on trees whose stacks fit (profiles/README.md):
  // spilling is for trees so deep that the stacks alone would force smaller workgroups and fewer resident waves.
  P.lds_levels = P.levels;
  if (!E.block_env) {
    const uint32_t want = E.lds_levels_env ? E.lds_levels_env : (P.levels >= PHX_SPILL_FROM_LEVELS ? PHX_SPILL_LDS_LEVELS : P.levels);
    P.lds_levels = std::max(2u, std::min(P.levels, want));
  }
  const bool spill = P.lds_levels < P.levels;
  // nodelets staged in LDS: whatever the per-lane stacks leave of the workgroup's share of the CU's 160 KB at full occupancy
  // (32 waves per CU); 9 (root + one level) when the stacks alone do not fit, and occupancy then follows from the LDS
  // 5-byte stack entries (PHX_STACK_PACKED): for the SPILL plan, when the pool's indices fit 24 bits
  P.packed = (PHX_STACK_PACKED >= 2 || (PHX_STACK_PACKED == 1 && spill)) && sc.num_elems < (1u << 24) ? 1u : 0u;
  auto stack_bytes = [&](uint32_t lv, uint32_t blk) { return P.packed ? (lv + ((lv + 3u) >> 2)) * blk * 4u : lv * blk * 8u; };
  auto plan = [&](uint32_t blk, uint32_t& ntop_out, uint32_t& lds_out) {  // -> workgroups per CU for this block size
    uint32_t ntop_req = E.ntop_env;
    if (!ntop_req) {
      const uint32_t share = 160u * 1024u / (2048u / blk), stacks = stack_bytes(P.lds_levels, blk) + 48u + (PHX_PERM_LUT ? 2048u : 0u);
      ntop_req = share > stacks + 9u * PHX_NODE_LDS_BYTES ? (share - stacks) / PHX_NODE_LDS_BYTES : 9u;
    }
    ntop_out = std::min(ntop_req, sc.num_elems);
    lds_out = ntop_out * PHX_NODE_LDS_BYTES + stack_bytes(P.lds_levels, blk) + 48u + (PHX_PERM_LUT ? 2048u : 0u);
    return std::min(160u * 1024u / lds_out, 2048u / blk);
  };
  // 1024-thread workgroups share one copy of the staged nodelets among 16 waves; a deep tree (levels >= 10: the stacks alone
  // exceed the CU's LDS at full occupancy) is better served by smaller workgroups, whose LDS granularity wastes less
  P.block = E.block_env ? E.block_env : 1024u;
  if (E.block_env || spill) P.wg_per_cu = plan(P.block, P.ntop, P.lds_bytes);  // spilling exists for 1024-thread workgroups only
  else {
    uint32_t best_waves = 0;
    for (uint32_t blk = 1024u; blk >= 256u; blk >>= 1) {
      uint32_t nt, l; const uint32_t wgs = plan(blk, nt, l);
      if (wgs * (blk / 64u) > best_waves) { best_waves = wgs * (blk / 64u); P.block = blk; P.ntop = nt; P.lds_bytes = l; P.wg_per_cu = wgs; }
    }
  }
  if (E.wg_cap > 0 && P.wg_per_cu > (uint32_t)E.wg_cap) P.wg_per_cu = (uint32_t)E.wg_cap;
  if (P.wg_per_cu == 0) P.wg_per_cu = 1;  // deeper than the LDS can hold even with 256 threads: the launch will report the error
  P.spill_threads = spill ? sc.num_cus * P.wg_per_cu * (uint32_t)E.grid_mul * P.block : 0u;
  return P;
}

void launch_trace(hipStream_t stream, const DevScene& sc, const PassBuffers& pb, int q, int sq, int do_closest, int do_shadow, uint32_t capacity) {
  const TraceEnv& E = trace_env();
  const TracePlan P = trace_plan(sc);
  const uint32_t block = P.block, ntop = P.ntop, lds = P.lds_bytes, levels = P.lds_levels;
  uint32_t grid = sc.num_cus * P.wg_per_cu * (uint32_t)E.grid_mul;  // persistent: the resident workgroups
  const uint32_t need = (((capacity + block - 1) / block + 7u) / 8u) * 8u;
  grid = std::max(8u, std::min(grid, need));
  const dim3 g(grid), b(block);
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, g, b, lds, stream, sc, pb, q, sq, do_closest, do_shadow, E.refill, ntop, levels, E.min_chunks, E.target_chunks);
  };
  if (P.lds_levels < P.levels) { if (P.packed) go(&k_trace<1024, true, true>); else go(&k_trace<1024, true>); }  // deep tree: 1024-thread workgroups, the stack's deep levels in HBM
#if PHX_STACK_PACKED >= 2
  else if (P.packed && block == 256) go(&k_trace<256, false, true>);
  else if (P.packed && block == 512) go(&k_trace<512, false, true>);
  else if (P.packed) go(&k_trace<1024, false, true>);
#endif
  else if (block == 256) go(&k_trace<256>);
  else if (block == 512) go(&k_trace<512>);
  else go(&k_trace<1024>);
}
// k_shade / k_shade_g walk the queue with a fixed grid: PHX_SHADE_GRID workgroups per resident slot (never more than the queue's
// capacity needs)
static uint32_t shade_grid(const DevScene& sc, uint32_t capacity, uint32_t per_wg, uint32_t block) {
  static const int mul = [] { const char* v = getenv("PHX_SHADE_GRID"); const int x = v ? atoi(v) : 4; return x < 1 ? 4 : x; }();
  const uint32_t resident = sc.num_cus * (2048u / block);
  const uint32_t need = (capacity + per_wg - 1) / per_wg;
  return std::max(1u, std::min(need, resident * (uint32_t)mul));
}
void launch_shade(hipStream_t stream, const DevScene& sc, const PassBuffers& pb, int q, int sq, uint32_t capacity, uint32_t sample0, int camera_rays) {
  const bool lens = camera_rays && sc.aperture_radius != 0.0f;  // camera_t::is_pinhole, entities/camera.hpp:37
  if (sc.diffuse_only) {
    const dim3 g((capacity + PHX_SHADE_BLOCK_D - 1) / PHX_SHADE_BLOCK_D), b(PHX_SHADE_BLOCK_D);
    if (sc.diffuse_only == 2) {
      if (lens) hipLaunchKernelGGL((k_shade<2, true, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
      else if (camera_rays) hipLaunchKernelGGL((k_shade<2, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
      else hipLaunchKernelGGL((k_shade<2, false>), g, b, 0, stream, sc, pb, q, sq, sample0);
    } else {
      if (lens) hipLaunchKernelGGL((k_shade<1, true, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
      else if (camera_rays) hipLaunchKernelGGL((k_shade<1, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
      else hipLaunchKernelGGL((k_shade<1, false>), g, b, 0, stream, sc, pb, q, sq, sample0);
    }
    return;
  }
  const dim3 g(shade_grid(sc, capacity, PHX_SHADE_BLOCK_G * PHX_SHADE_ITEMS_G, PHX_SHADE_BLOCK_G)), b(PHX_SHADE_BLOCK_G);
  if (sc.any_per_hit) {
    if (lens) hipLaunchKernelGGL((k_shade_g<true, true, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
    else if (camera_rays) hipLaunchKernelGGL((k_shade_g<true, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
    else hipLaunchKernelGGL((k_shade_g<true, false>), g, b, 0, stream, sc, pb, q, sq, sample0);
  } else {
    if (lens) hipLaunchKernelGGL((k_shade_g<false, true, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
    else if (camera_rays) hipLaunchKernelGGL((k_shade_g<false, true>), g, b, 0, stream, sc, pb, q, sq, sample0);
    else hipLaunchKernelGGL((k_shade_g<false, false>), g, b, 0, stream, sc, pb, q, sq, sample0);
  }
}
void launch_trace_primary(hipStream_t stream, const DevScene& sc, const PassBuffers& pb, uint32_t npaths, uint32_t sample0, int q, int sq) {
  static const int rpl_env = [] { const char* v = getenv("PHX_PRIMARY_RPL"); return v ? atoi(v) : 0; }();  // experiment: 1, 2 or 4 rays per lane
  const size_t lds = ((size_t)std::max(2u, std::min(sc.stack_levels, (uint32_t)PHX_MAX_BVH_DEPTH)) * PHX_PRIMARY_BLOCK + (PHX_PRIMARY_BLOCK / 64) * PHX_MAX_BVH_DEPTH) * sizeof(uint2);
  // path ids are pixel-major (pixel * samples_of_this_pass + sample): 256 consecutive paths are samples of ONE pixel when the pass holds
  // a multiple of 256 samples per pixel; otherwise 128 paths, i.e. the samples of a few pixels next to each other in a tile row
  // (measured, profiles/r03_zo_primary_packets.log: 4 rays per lane beat 2 only inside one pixel; 2 beat 1 from 16 samples per pixel up)
  uint32_t rpl = pb.num_samples % 256u == 0 ? 4u : pb.num_samples >= 16u ? 2u : 1u;
  if (rpl_env == 1 || rpl_env == 2 || rpl_env == 4) rpl = (uint32_t)rpl_env;
  const dim3 g((npaths + PHX_PRIMARY_BLOCK * rpl - 1) / (PHX_PRIMARY_BLOCK * rpl)), b(PHX_PRIMARY_BLOCK);
  if (sc.aperture_radius != 0.0f) {  // thin lens: the rays of a packet leave from a disc, not a point (the packet's bounds know origins apart)
    if (rpl == 4) hipLaunchKernelGGL((k_trace_primary<4, true>), g, b, lds, stream, sc, pb, npaths, sample0, q, sq);
    else if (rpl == 2) hipLaunchKernelGGL((k_trace_primary<2, true>), g, b, lds, stream, sc, pb, npaths, sample0, q, sq);
    else hipLaunchKernelGGL((k_trace_primary<1, true>), g, b, lds, stream, sc, pb, npaths, sample0, q, sq);
    return;
  }
  if (rpl == 4) hipLaunchKernelGGL(k_trace_primary<4>, g, b, lds, stream, sc, pb, npaths, sample0, q, sq);
  else if (rpl == 2) hipLaunchKernelGGL(k_trace_primary<2>, g, b, lds, stream, sc, pb, npaths, sample0, q, sq);
  else hipLaunchKernelGGL(k_trace_primary<1>, g, b, lds, stream, sc, pb, npaths, sample0, q, sq);
}
namespace {
__global__ void k_build_shade_recs(const TriRec* __restrict__ tris, const uint32_t* __restrict__ elem_of_prim, float4* __restrict__ elem_shade, uint32_t n) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t e = elem_of_prim[p];
  if (e == 0xffffffffu) return;
  const TriRec T = tris[e];
  const v3 gn = shading_normal(DevScene{}, e, false, v3(T.e0x, T.e0y, T.e0z), v3(T.e1x, T.e1y, T.e1z), 0.0f, 0.0f);  // the flat face's normal, by the expression the shade kernels used per hit
  elem_shade[e] = make_float4(gn.x, gn.y, gn.z, u2f(T.material));
}
__global__ void k_permute_normals(const float* __restrict__ prim_normals, const uint32_t* __restrict__ elem_of_prim, float* __restrict__ elem_normals, uint32_t n) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t e = elem_of_prim[p];
  if (e == 0xffffffffu) return;
#pragma unroll
  for (int k = 0; k < 9; ++k) elem_normals[9 * (size_t)e + k] = prim_normals[9 * (size_t)p + k];
}
__global__ void k_remap_light_tris(DevLightTri* __restrict__ lt, uint32_t n, const uint32_t* __restrict__ elem_of_prim) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && lt[i].smooth) lt[i].prim = elem_of_prim[lt[i].prim];
}
}  // namespace
void launch_build_shade_recs(hipStream_t stream, const TriRec* tris, const uint32_t* elem_of_prim, float4* elem_shade, uint32_t num_prims) {
  if (num_prims) hipLaunchKernelGGL(k_build_shade_recs, dim3((num_prims + 255) / 256), dim3(256), 0, stream, tris, elem_of_prim, elem_shade, num_prims);
}
void launch_permute_normals(hipStream_t stream, const float* prim_normals, const uint32_t* elem_of_prim, float* elem_normals, uint32_t num_prims) {
  if (num_prims) hipLaunchKernelGGL(k_permute_normals, dim3((num_prims + 255) / 256), dim3(256), 0, stream, prim_normals, elem_of_prim, elem_normals, num_prims);
}
void launch_remap_light_tris(hipStream_t stream, DevLightTri* light_tris, uint32_t num_light_tris, const uint32_t* elem_of_prim) {
  if (num_light_tris) hipLaunchKernelGGL(k_remap_light_tris, dim3((num_light_tris + 255) / 256), dim3(256), 0, stream, light_tris, num_light_tris, elem_of_prim);
}
void launch_film(hipStream_t stream, const PassBuffers& pb, uint32_t num_samples, float inv) {
  hipLaunchKernelGGL(k_film, dim3((pb.num_pixels + PHX_FILM_PIX - 1) / PHX_FILM_PIX), dim3(256), 0, stream, pb, num_samples, inv);
}
void launch_scatter_film(hipStream_t stream, const PassBuffers& pb, float* device_film, uint32_t film_width) {
  hipLaunchKernelGGL(k_scatter_film, dim3(blocks_for(pb.num_pixels)), dim3(PHX_BLOCK), 0, stream, pb, device_film, film_width);
}
void launch_trace_rays(hipStream_t stream, const DevScene& sc, uint32_t n, const float4* ro, const float4* rd, float4* hit, int any) {
  const dim3 g(blocks_for(n)), b(PHX_BLOCK);
  const size_t lds = (size_t)std::max(2u, std::min(sc.stack_levels, (uint32_t)PHX_MAX_BVH_DEPTH)) * PHX_BLOCK * sizeof(uint2);
  if (any) hipLaunchKernelGGL((k_trace_rays<true>), g, b, lds, stream, sc, n, ro, rd, hit);
  else hipLaunchKernelGGL((k_trace_rays<false>), g, b, lds, stream, sc, n, ro, rd, hit);
}
void launch_bsdf_f(hipStream_t stream, const DevMaterial* mat, uint32_t n, const float* n3, const float* wi3, const float* wo3, float* f3) {
  hipLaunchKernelGGL(k_bsdf_f, dim3((n + 63) / 64), dim3(64), 0, stream, mat, n, n3, wi3, wo3, f3);
}
void launch_bsdf_sample(hipStream_t stream, const DevMaterial* mat, uint32_t n, const float* n3, const float* wi3, const float* u2,
                        float* wo3, float* f3, float* pdf, uint32_t* flags) {
  hipLaunchKernelGGL(k_bsdf_sample, dim3((n + 63) / 64), dim3(64), 0, stream, mat, n, n3, wi3, u2, wo3, f3, pdf, flags);
}

}  // namespace phx
