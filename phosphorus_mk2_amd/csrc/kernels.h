// kernels.h — device-side data layout of the wavefront path tracer and the launch entry points
// implemented in kernels.hip.  One "pass" carries P pixels x S samples = npaths paths through
// generate -> [trace closest -> shade/NEE/integrate -> trace shadow] x depth -> film.
//
// HBM layout (all records 16 B so that a wave reads/writes 1 KiB per instruction):
//   ray queue (x2, ping-pong)  ro[i] = (o.xyz, path | SPECULAR<<31)   rd[i] = (d.xyz, tmax)
//   hit records                hit[i] = (t, u, v, triangle-record index)
//   shadow queue               so[i] = (o.xyz, path)  sd[i] = (d.xyz, tmax)  sc[i] = (beta*Li rgb, 0)
//   path state (by path id)    pb[path] = (beta.rgb, depth)   pr[path] = (radiance.rgb, 0)
//   path id = sample_in_pass * P + pixel_in_batch  (neighbouring lanes = neighbouring pixels)
// These take the place of ray_t<1024> / interaction_t<1024> / active_t<1024> / spt::state_t<1024>
// (reference src/state.hpp:40-282, src/kernels/cpu/spt.hpp:23-64): the interaction record is never
// materialised — shade, NEE and integrate are one kernel.
#pragma once
#include "bsdf.h"
#include "bvh8.h"

#include <hip/hip_runtime.h>

namespace phx {

// Light table (light_t::make_area, src/light.cpp:10-45).  What k_shade would otherwise fetch through three more dependent loads per
// shaded hit is resolved once at preprocess, with the same fp32 operations the kernel would use: the pick pdf (1/area)/nlights
// (spt.hpp:95-149), the light material's emission (material_t::evaluate's e) and, per triangle, the geometric normal of a flat face.
struct DevLight { uint32_t first_tri, num_tris; float area; uint32_t material; float lpdf, ex, ey, ez; };                 // 32 B
struct DevLightTri { float ax, ay, az, bx, by, bz, cx, cy, cz, nx, ny, nz; uint32_t prim, smooth, mesh_mat, face; };   // 64 B
// A material of a scene in which every material is at most ONE Lambert lobe (the soups, the Cornell box): 32 B instead of 544
struct DevMatLite { float wx, wy, wz; uint32_t lobes_flags /* num_lobes | flags << 8 */; float ex, ey, ez; uint32_t pad; };

struct DevScene {
  const uint32_t* pool;           // the BVH8 pool: 16 words per element, element 0 = root nodelet (bvh8.h)
  const TriRec* tris;             // the same pool seen as triangle records (hit records carry pool indices)
  SceneGrid grid;                 // grid of the nodelets' origins
  const uint32_t* prim_material;  // per primitive (scene_t::triangles() order): material | smooth << 31
  const float4* elem_shade;       // per POOL ELEMENT (indexed like `tris`): what shading needs of a hit triangle in 16 bytes — its geometric normal
                                  // normalize((v1-v0) x (v2-v0)) (mesh.cpp:201-215, computed once at preprocess by the shade kernels' own expression) and
                                  // material | smooth << 31 — instead of the 64-byte triangle record (round 6)
  const float* elem_normals;      // 9 floats (n0,n1,n2) per POOL ELEMENT — indexed like `tris`, by the hit's pool index, so that the normals are requested
                                  // WITH the triangle record, not after it (round 6; entries of nodelets are never read) — or nullptr when no face is smooth
  const DevMaterial* materials;
  const DevMatLite* mat_lite;     // diffuse_only == 2: the same table, 32 B per material
  const DevLight* lights;
  const DevLightTri* light_tris;
  uint32_t num_lights;
  int32_t env_material;
  float cam_m[16];
  float zoom, stepx, stepy, ratio;
  uint32_t width, height;
  uint32_t max_depth;
  uint32_t stack_levels;          // BVH depth
  uint32_t num_elems;             // pool elements (stored breadth first: a prefix of the pool is the top of the tree)
  uint32_t num_cus;               // compute units of the device (persistent grid sizing)
  uint32_t diffuse_only;          // 1: every lobe of every material is Lambert (k_shade<1>); 2: and no material has more than one (k_shade<2>)
  uint32_t any_per_hit;           // some material's closure weights depend on the hit (glass): k_shade<0>; none: k_shade<3>
  uint2* stack_spill;             // k_trace<., SPILL>: stack entries below the levels kept in LDS, [level - lds_levels][thread of the grid]
  uint32_t spill_stride;          // threads of the largest k_trace grid (0 = every level is in LDS)
  float aperture_radius, focal_distance;  // camera_t (entities/camera.hpp:24-29): thin lens when aperture_radius != 0 (camera.hpp:140-147)
};

// counters (x CNT_STRIDE words): [0],[1] ray-queue lengths (ping-pong); [2],[3] shadow-queue lengths (by step parity); [4],[5] chunk cursors
// every counter on its own 128-byte line: the queue-length counters take one atomic per k_shade workgroup and one address (line)
// sustains ~90 atomics/us — two counters on one line halve what each gets
// [4 + p * CNT_SEGS + s]: the range cursor of queue p (0 shadow, 1 closest) for SEGMENT s of the queue — the queue is cut in CNT_SEGS equal
// parts, one per XCD (workgroup ids go round the XCDs: id & 7), so that an XCD's L2 serves one part of the image
enum { CNT_STRIDE = 32, CNT_SHADOW = 2 * CNT_STRIDE, CNT_CURSOR = 4 * CNT_STRIDE, CNT_SEGS = 8, CNT_WORDS = (4 + 2 * CNT_SEGS) * CNT_STRIDE };
struct DevStats {
  unsigned long long rays_closest, rays_shadow, rays_masked, camera_samples;
  // instrumented build only (-DPHX_COUNT=1, `make variant NAME=count`): traversal work, [0] closest-hit rays, [1] shadow rays
  unsigned long long node_visits_lds[2], node_visits_mem[2], tri_tests[2];
  // wave-level: loop iterations, executions of the node block / the triangle block (a block runs when ANY lane needs it)
  unsigned long long wave_iters, node_block_execs, tri_block_execs, refills;
  unsigned long long idle_lane_iters, tri_pending_lane_iters;
  // k_trace_primary (instrumented build): packets walked, packets that fell back to the per-lane walk, node tests and triangle tests
  // per PACKET (one test serves the 64 rays), and lanes whose ray was improved by a triangle test (of 64 per test)
  unsigned long long primary_packets, primary_fallbacks, primary_node_tests, primary_tri_tests, primary_tri_lanes_hit;
  unsigned long long watchdog;         // k_trace waves that gave up after PHX_TRACE_WATCHDOG iterations: 0, or the frame is reported as failed
  unsigned long long stack_pushes[8];  // instrumented: pushes onto the per-lane group stack by the depth they land at (7 = 7 and deeper)
  // instrumented: pending (ray, triangle) pairs of the wave at its triangle-block executions — their sum, and executions by the number of
  // pairs: <= 8, <= 16, <= 24, <= 32, <= 48, <= 64, <= 96, more
  unsigned long long tri_pairs_pending, tri_pairs_hist[8];
  unsigned long long ring_watchdog;    // k_shade_g waves whose wait on the append ring timed out (PHX_RING_SPINS): 0, or the frame is reported as failed
};

struct PassBuffers {
  float4* ro[2]; float4* rd[2];
  float4* hit;
  float4* so; float4* sd; float4* sc;
  float4* qs[2];              // path state (beta, depth) of the queue entry: it travels WITH the ray (round 6; before: an array by path id, a dependent gather)
  float4* pr;
  float4* pn;                 // primary normal + hit flag per path (only when the normals channel is on)
  uint32_t* counters;         // CNT_WORDS
  DevStats* stats;
  const uint32_t* pix_xy;     // per pixel of the batch: x | y << 16 (film coordinates)
  const float2* jitter;       // per spp index: film jitter shared by all pixels (src/sampling.cpp:98-112)
  float* acc;                 // batch render buffer: tile after tile, interleaved xstride floats per pixel
  uint32_t num_pixels;        // P
  uint32_t num_samples;       // samples of this pass; path id = pixel_in_batch * num_samples + sample_in_pass
  uint32_t xstride;
  uint32_t normals_offset;    // offset of channel "normals" inside a pixel, 0 = off
  uint64_t seed;
  uint32_t* qlen_out;         // k_trace: host-visible word that receives the length of the ray queue it traces (device.cpp sizes the next shade grids by it), or nullptr
};

// shape of a k_trace launch on a scene (reported through phx_stats so that tests can assert which plan a tree ran with)
struct TracePlan { uint32_t block, ntop, levels, lds_bytes, wg_per_cu, lds_levels, spill_threads, packed /* 5-byte stack entries in LDS */; };
TracePlan trace_plan(const DevScene& sc);
// per device, once: lets the traversal kernels use the CU's full 160 KB of LDS as dynamic shared memory
hipError_t init_kernels_on_current_device();

bool launch_counts_traversal_work();  // true in the instrumented build (PHX_COUNT)

// launches (all asynchronous on `stream`)
// start of a pass: queue 0 stands for the num_pixels x num_samples camera rays, which are rebuilt on the fly (camera_ray)
void launch_begin_pass(hipStream_t stream, const PassBuffers& pb, uint32_t num_samples);
// one persistent launch: closest-hit rays of queue q (do_closest) + any-hit rays of shadow queue sq (do_shadow)
void launch_trace(hipStream_t stream, const DevScene& sc, const PassBuffers& pb, int q, int sq, int do_closest, int do_shadow, uint32_t capacity);
// step 0 of a pass: the npaths camera rays (never stored: rebuilt from the pixel and jitter tables), walked as packets by k_trace_primary;
// writes the hit records of queue q and does k_trace's start-of-step bookkeeping
void launch_trace_primary(hipStream_t stream, const DevScene& sc, const PassBuffers& pb, uint32_t npaths, uint32_t sample0, int q, int sq);
// shades queue q, appends survivors to queue q^1 and NEE rays to shadow queue sq
void launch_shade(hipStream_t stream, const DevScene& sc, const PassBuffers& pb, int q, int sq, uint32_t capacity, uint32_t sample0, int camera_rays);
void launch_film(hipStream_t stream, const PassBuffers& pb, uint32_t num_samples, float inv_spp_pps);
// preprocess: vertex normals from scene_t::triangles() order to pool-element order (elem_of_prim: the builders' map), and the smooth light
// triangles' `prim` from primitive to pool element (they look their normals up in the same table)
void launch_build_shade_recs(hipStream_t stream, const TriRec* tris, const uint32_t* elem_of_prim, float4* elem_shade, uint32_t num_prims);
void launch_permute_normals(hipStream_t stream, const float* prim_normals, const uint32_t* elem_of_prim, float* elem_normals, uint32_t num_prims);
void launch_remap_light_tris(hipStream_t stream, DevLightTri* light_tris, uint32_t num_light_tris, const uint32_t* elem_of_prim);
void launch_scatter_film(hipStream_t stream, const PassBuffers& pb, float* device_film, uint32_t film_width);

// stage-level entry points (synchronous helpers for the parity tests)
void launch_trace_rays(hipStream_t stream, const DevScene& sc, uint32_t n, const float4* ro, const float4* rd, float4* hit, int any);
void launch_bsdf_f(hipStream_t stream, const DevMaterial* mat, uint32_t n, const float* n3, const float* wi3, const float* wo3, float* f3);
void launch_bsdf_sample(hipStream_t stream, const DevMaterial* mat, uint32_t n, const float* n3, const float* wi3, const float* u2,
                        float* wo3, float* f3, float* pdf, uint32_t* flags);

}  // namespace phx
