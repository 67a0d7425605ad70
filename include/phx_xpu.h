/*
 * phx_xpu.h — C ABI of the MI355X (gfx950) path-tracing device for phosphorus.
 *
 * This is the drop-in boundary: a shared library (libphx_hip.so) exporting exactly
 * what a `hip_t : xpu_t` backend of the reference needs.  The reference interface it
 * replaces is `struct xpu_t` (reference src/xpu.hpp:12-40):
 *
 *     virtual void preprocess(const scene_t&)            -> phx_dev_preprocess
 *     virtual void start(const scene_t&, frame_state_t&) -> phx_dev_start
 *     virtual void join()                                -> phx_dev_join
 *     static T* make(const parsed_options_t&)            -> phx_dev_make   (src/xpu/cpu.hpp:35, src/xpu/cuda.hpp:12)
 *     virtual ~xpu_t()                                   -> phx_dev_destroy
 *     static discover(const parsed_options_t&)           -> phx_discover   (src/xpu.cpp:7-9)
 *
 * Plain C types only: pointers + sizes, no C++/torch types, int status codes instead of the
 * reference's exceptions (std::runtime_error in utils/allocator.hpp:37-39 and kernels/cpu/spt.hpp:230).
 * All arrays passed to phx_dev_preprocess are copied; the caller may free them afterwards
 * (the reference device also rebuilds its own BVH in preprocess, src/xpu/cpu.cpp:35-44,219).
 */
#ifndef PHX_XPU_H
#define PHX_XPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------- */
enum {
  PHX_OK            = 0,
  PHX_ERR_ARG       = 1, /* bad argument / malformed scene */
  PHX_ERR_DEVICE    = 2, /* HIP error; see phx_last_error() */
  PHX_ERR_NO_DEVICE = 3, /* no gfx950 device visible */
  PHX_ERR_STATE     = 4, /* call order violated (start before preprocess, ...) */
  PHX_ERR_OOM       = 5  /* device arena exhausted ("Out of memory", utils/allocator.hpp:37) */
};

/* ---- ray / interaction flag bits (reference src/state.hpp:33-36) ------------------- */
enum { PHX_HIT = 1, PHX_MASKED = 2, PHX_SHADOW = 4, PHX_SPECULAR = 8 };

/* ---- closure ids (reference src/bsdf.hpp:14-24) and lobe flags (src/bsdf/params.hpp:12-16) */
enum {
  PHX_LOBE_EMISSIVE = 0, PHX_LOBE_DIFFUSE = 1, PHX_LOBE_OREN_NAYAR = 2, PHX_LOBE_REFLECTION = 4,
  PHX_LOBE_REFRACTION = 8, PHX_LOBE_MICROFACET = 16, PHX_LOBE_SHEEN = 32, PHX_LOBE_BACKGROUND = 64,
  PHX_LOBE_TRANSPARENT = 128
};
enum { PHX_BSDF_DIFFUSE = 1, PHX_BSDF_GLOSSY = 2, PHX_BSDF_SPECULAR = 4, PHX_BSDF_REFLECT = 8, PHX_BSDF_TRANSMIT = 16 };
#define PHX_MAX_LOBES 8 /* bsdf_t::MaxLobes, src/bsdf.hpp:9 */

/* ---- options: parsed_options_t (reference src/options.hpp:6-43) + device knobs ------- */
typedef struct phx_options {
  uint32_t samples_per_pixel; /* default 16 */
  uint32_t paths_per_sample;  /* default 16; only scales the film by 1/pps (src/xpu/cpu.cpp:191) */
  uint32_t path_depth;        /* default 9 */
  uint32_t single_threaded;   /* honoured by CPU devices only */
  uint32_t host_only;         /* --no-gpu: phx_discover returns 0 devices */
  uint32_t render_normals;
  uint32_t verbose;
  /* device additions (0 = auto) */
  int32_t  device_ordinal;    /* HIP device index; -1 = current device */
  uint32_t samples_in_flight; /* samples of one pixel carried per wavefront pass */
  uint32_t tiles_per_batch;   /* tiles pulled from the queue per pass */
  uint32_t bvh_builder;       /* PHX_BVH_AUTO (default), PHX_BVH_DEVICE_LBVH or PHX_BVH_HOST_SAH: where preprocess builds the tree */
  uint32_t reserved[5];
} phx_options;
/* AUTO = the device builder (LBVH over extended Morton codes + the optimal 8-wide collapse) for every scene with at least 64
 * triangles: 1 M triangles in 12 ms and 10 M in 40 ms against 0.6 s / 7 s of the host's binned-SAH builder, and since round 3 its
 * trees trace as fast or faster on every scene measured (profiles/r03_z_emc_probe.log).  cpu_t::preprocess rebuilds its accelerator
 * on every call (src/xpu/cpu.cpp:35-44), so the build time is part of the interface's cost.  HOST_SAH stays selectable, and AUTO falls back to it
 * when the device build fails (phx_stats.bvh_built_on_device = 0); an explicit DEVICE_LBVH request fails with PHX_ERR_DEVICE instead. */
enum { PHX_BVH_AUTO = 0, PHX_BVH_DEVICE_LBVH = 1, PHX_BVH_HOST_SAH = 2 };

/* ---- scene: what the device reads through scene_t (src/scene.hpp:14-50) --------------- */

/* One closure of a flattened closure tree (the output contract of material_t::evaluate,
 * src/material.cpp:218-305,419-458, with constant inputs; shading normal = interpolated N). */
typedef struct phx_lobe {
  uint32_t type;      /* PHX_LOBE_* */
  float    weight[3]; /* accumulated colour weight */
  float    alpha;     /* OrenNayar: sigma (treated as degrees, src/bsdf/params.hpp:38) */
  float    eta;       /* Reflection / Refraction / Microfacet */
  float    xalpha;    /* Microfacet roughness inputs BEFORE precompute() (params.hpp:86-99) */
  float    yalpha;
  uint32_t refract;   /* Microfacet: 1 = transmissive */
  float    r;         /* Sheen roughness */
  /* Per-hit closure weight — the one hit-dependent input the reference's own node shaders produce: a mix_closure_node whose `fac`
   * is driven by fresnel_dielectric_node (Blender's glass node: plugins/blender/blender/shader.hpp:306-335,
   * src/shaders/fresnel_dielectric_node.osl:16-20, mix_closure_node.osl:20).  At every hit
   *   fac = fresnel_dielectric(dot(I, N), backfacing ? 1 / max(1e-5, fac_ior) : max(1e-5, fac_ior))      (src/shaders/fresnel.h)
   * and the closure's weight is (pre_weight * term) * weight with term = fac (PHX_FAC_MIX_B) or 1 - fac (PHX_FAC_MIX_A), in the
   * order material_t::eval_closure multiplies down the tree (src/material.cpp:218-305); a closure whose weight comes out all
   * zero is not there at that hit (OSL: closure * 0 is the null closure).  PHX_FAC_NONE: weight is the whole constant weight. */
  uint32_t fac_mode;
  float    fac_ior;
  float    pre_weight[3]; /* product of the constant weights ABOVE the hit-dependent factor in the closure tree */
  uint32_t pad;
} phx_lobe;
enum { PHX_FAC_NONE = 0, PHX_FAC_MIX_B = 1, PHX_FAC_MIX_A = 2 };

typedef struct phx_material {
  uint32_t num_lobes;   /* 0 for pure emitters (diffuse_emitter_node.osl) */
  uint32_t is_emitter;  /* material_t::is_emitter, src/material.cpp:487 */
  float    emission[3]; /* weight of the emission()/background() closure: hits.e */
  uint32_t pad[3];
  phx_lobe lobes[PHX_MAX_LOBES];
} phx_material;

/* mesh_t::face_set_t, src/mesh.hpp:26-41 */
typedef struct phx_face_set {
  uint32_t        material;  /* index into phx_scene.materials */
  uint32_t        num_faces;
  const uint32_t* faces;     /* face indices (not multiplied by 3) */
} phx_face_set;

/* mesh_t, src/mesh.hpp:14-138; mesh id = index in phx_scene.meshes (src/scene.cpp:79-82) */
typedef struct phx_mesh {
  const float*    vertices;  /* xyz per vertex */
  uint32_t        num_vertices;
  const float*    normals;   /* xyz; per vertex, or per face-corner when !NORMALS_PER_VERTEX */
  uint32_t        num_normals;
  const uint32_t* faces;     /* 3 vertex indices per face */
  uint32_t        num_faces;
  const uint8_t*  smooth;    /* per face: 1 = interpolate normals, 0 = geometric normal (mesh.cpp:187-206) */
  uint32_t        flags;     /* PHX_MESH_* */
  uint32_t        num_sets;
  const phx_face_set* sets;
} phx_mesh;
enum { PHX_MESH_UV_PER_VERTEX = 1, PHX_MESH_NORMALS_PER_VERTEX = 2 }; /* mesh_t::flags_t, src/mesh.hpp:20-23 */

/* camera_t, src/entities/camera.hpp:10-39.  aperture_radius == 0: pinhole (camera_t::is_pinhole); otherwise the thin lens of
 * src/kernels/cpu/camera.hpp:140-147 with the disc mapping of src/math/simd/sampling.hpp:8-32 as the reference wrote it (SURVEY A-21:
 * raw samples, swapped select) and focal_distance the distance of the plane in focus.  Both must be finite. */
typedef struct phx_camera {
  float    to_world[16]; /* Imath::M44f x[i][j], row-vector convention v' = v * M */
  float    fov;
  float    focal_distance;
  float    aperture_radius;
  uint32_t film_width;
  uint32_t film_height;
} phx_camera;

typedef struct phx_scene {
  uint32_t            num_meshes;
  const phx_mesh*     meshes;
  uint32_t            num_materials;
  const phx_material* materials;
  int32_t             environment_material; /* -1 = none (scene_t::environment, src/scene.cpp:126) */
  phx_camera          camera;
} phx_scene;

/* ---- frame: frame_state_t {sampler, tiles, film} (src/state.hpp:18-31) --------------- */
typedef struct phx_tile { uint32_t x, y, w, h; } phx_tile; /* job::tiles_t::tile_t, src/jobs/tiles.hpp:12-19 */

/* job::tiles_t::next (src/jobs/tiles.hpp:40-47): returns 1 and fills *out, or 0 when drained.
 * Called from the device's driver thread; must be thread safe (devices share one queue,
 * src/core.cpp:103-108). */
typedef int (*phx_next_tile_fn)(void* user, phx_tile* out);

/* film_t<>::add_tile (src/film.hpp:12-15): `buffer` is the tile's interleaved fp32
 * render_buffer_t (src/buffer.hpp:8-97): pixel (x,y) at buffer[y*ystride + x*xstride + c].
 * Called from the device's driver thread. */
typedef void (*phx_add_tile_fn)(void* user, int32_t x, int32_t y, int32_t w, int32_t h,
                                const float* buffer, uint32_t xstride, uint32_t ystride);

typedef struct phx_frame {
  void*            tiles_user;
  phx_next_tile_fn next_tile;
  void*            film_user;
  phx_add_tile_fn  add_tile;        /* may be NULL when device_film is set */
  uint64_t         sampler_seed;    /* seed of the counter-based sampler (replaces sampler_t's mt19937) */
  uint32_t         primary_components; /* components of channel "primary": 3 or 4 (3 are written, buffer.cpp:25-30) */
  uint32_t         normals_channel;    /* 1: append channel "normals" x3 (cpu.cpp:194-196) */
  /* optional: accumulate straight into a full-frame device-resident film (W*H*xstride fp32 in HBM,
   * e.g. a torch tensor's data_ptr) — used for the multi-GPU film reduce. */
  float*           device_film;
  /* optional: a full-frame HOST film (W*H*xstride fp32) the device fills tile by tile itself — the same
   * effect as an add_tile callback that copies into a frame buffer (film::file_t, src/film/file.cpp:27-41)
   * without a foreign-function call per tile. */
  float*           host_film;
  uint32_t         reserved[2];
} phx_frame;

/* ---- statistics ------------------------------------------------------------------------ */
typedef struct phx_stats {
  uint64_t camera_samples;   /* primary rays generated */
  uint64_t rays_closest;     /* non-masked slots presented to closest-hit trace */
  uint64_t rays_shadow;      /* non-masked shadow rays traced */
  uint64_t rays_masked;      /* shadow rays masked before trace (src/kernels/cpu/spt.hpp:138-141) */
  uint64_t tiles;
  uint64_t trace_launches;   /* k_trace launches (each traces closest-hit + pending shadow rays) */
  double   trace_ms;         /* sum of HIP-event durations of the k_trace launches */
  double   closest_ms;       /* = trace_ms (kept for ABI stability) */
  double   shadow_ms;        /* 0: shadow rays are traced inside k_trace */
  double   shade_ms;         /* generate + shade/NEE + integrate + film kernels */
  double   frame_ms;         /* wall time start..join on the host */
  uint64_t bvh_nodes;
  uint64_t bvh_bytes;
  uint64_t triangles;
  double   preprocess_ms;    /* wall time of the last phx_dev_preprocess */
  double   bvh_build_ms;     /* of which: tree construction (host SAH, or device LBVH incl. its upload of the triangles) */
  /* the k_trace launch plan of the preprocessed scene (kernels.hip: trace_plan) */
  uint64_t trace_block;      /* threads per k_trace workgroup: 256, 512 or 1024 */
  uint64_t trace_ntop;       /* top-of-tree elements staged in LDS by every workgroup */
  uint64_t trace_levels;     /* per-lane stack entries (= tree depth - 1, at least 2) */
  uint64_t trace_waves_per_cu; /* resident k_trace waves per compute unit with that LDS footprint */
  uint64_t bvh_depth;        /* levels of the 8-wide tree (<= 64) */
  uint64_t paths_in_flight;  /* paths carried per wavefront pass in the last frame (pixels of a batch x samples) */
  /* traversal work of the last frame — filled only by the instrumented build (make variant NAME=count EXTRA=-DPHX_COUNT=1),
   * 0 otherwise; [0] closest-hit rays, [1] shadow rays */
  uint64_t node_visits_lds[2]; /* nodelets read from the top of the tree staged in LDS */
  uint64_t node_visits_mem[2]; /* nodelets read through the vector L1 (4 x 16 B per lane) */
  uint64_t tri_tests[2];       /* triangle records read through the vector L1 (3 x 16 B per lane) and tested */
  uint64_t instrumented;       /* 1 if this library counts them */
  uint64_t wave_iters;         /* instrumented: k_trace loop iterations summed over waves */
  uint64_t node_block_execs;   /* instrumented: iterations in which at least one lane visited a node */
  uint64_t tri_block_execs;    /* instrumented: iterations in which at least one lane tested a triangle */
  uint64_t refills;            /* instrumented: refill rounds summed over waves */
  uint64_t idle_lane_iters;    /* instrumented: lanes without a ray at the node block, summed over iterations */
  uint64_t tri_pending_lane_iters; /* instrumented: lanes that sit out the node block because triangles of their last node are pending */
  uint64_t stack_pushes[8];    /* instrumented: pushes onto the per-lane stack of pending sibling groups, by the depth they land at (7 = 7 and deeper) */
  double   bvh_cost_model;     /* modelled traversal cost of the tree in use (the optimal collapse's objective; area units) */
  uint64_t trace_lds_levels;   /* ... of which this many live in LDS (deep trees: the rest spills to HBM) */
  uint64_t bvh_built_on_device; /* 1: the tree in use was built on the device */
  double   shade_kernel_ms;    /* of shade_ms: the shade/NEE/integrate launches alone (k_shade or k_shade_g), HIP events */
  uint64_t shade_launches;     /* how many of them */
  uint64_t shade_general;      /* 1: the scene has non-Lambert closures and runs k_shade_g; 0: k_shade (Lambert only) */
  double   primary_ms;         /* of trace_ms: the camera-ray launches (k_trace_primary: one packet walk per 64 rays), HIP events; closest_ms
                                  holds the k_trace launches alone */
  uint64_t primary_launches;   /* how many of them (one per pass) */
  uint64_t primary_packets;    /* instrumented: packets of 64 camera rays walked by k_trace_primary ... */
  uint64_t primary_fallbacks;  /* ... of which this many had rays in more than one direction octant and took the per-lane walk */
  uint64_t primary_node_tests; /* instrumented: node tests per packet (one test serves its 64 rays), summed */
  uint64_t primary_tri_tests;  /* instrumented: triangles a packet reached (each is tested by its 64 lanes), summed */
  uint64_t primary_tri_lanes_hit; /* instrumented: lanes whose closest hit a triangle test improved, summed */
  uint64_t device_bytes;       /* HBM this device object holds right now: tree, scene tables, ray / hit / shadow queues, path state, batch buffer */
  uint64_t tri_pairs_pending;  /* instrumented: pending (ray, triangle) pairs of a wave at its triangle-block executions, summed (each execution tests one per pending lane) */
  uint64_t tri_pairs_hist[8];  /* instrumented: those executions by the wave's pending pairs: <= 8, 16, 24, 32, 48, 64, 96, more */
  uint64_t trace_stack_packed; /* 1: k_trace keeps 5-byte stack entries in LDS (deep trees: more of the tree is staged), 0: 8-byte entries */
} phx_stats;

typedef struct phx_device phx_device; /* opaque */

/* ---- the xpu_t surface ----------------------------------------------------------------- */
/* xpu_t::discover (src/xpu.cpp:7-9): number of usable gfx950 devices (0 when options->host_only). */
int         phx_discover(const phx_options* options, int* num_devices);
/* T::make(const parsed_options_t&) (src/xpu/cpu.hpp:35).  NULL on failure (see phx_last_error). */
phx_device* phx_dev_make(const phx_options* options);
/* xpu_t::preprocess (src/xpu.hpp:20; cpu.cpp:219 -> details_t::reset :35): flatten + upload scene, build BVH. */
int         phx_dev_preprocess(phx_device* dev, const phx_scene* scene);
/* xpu_t::start (src/xpu.hpp:26; cpu.cpp:223-238): non-blocking; spawns the driver thread. */
int         phx_dev_start(phx_device* dev, const phx_frame* frame);
/* xpu_t::join (src/xpu.hpp:32; cpu.cpp:240): blocks; returns the frame's status. */
int         phx_dev_join(phx_device* dev);
/* ~xpu_t.  A frame that was started and not joined is JOINED here: destroy blocks until that frame has ended, and the frame's
 * next_tile / add_tile callbacks may still be called while it does — whatever they use must outlive this call.  The device's driver
 * thread (one per device, asleep between frames) ends with it; a device that is never destroyed leaves that thread asleep at exit. */
void        phx_dev_destroy(phx_device* dev);

/* last error message of the calling thread (a frame's error is handed to the thread that calls phx_dev_join) */
const char* phx_last_error(void);
int         phx_dev_get_stats(const phx_device* dev, phx_stats* out);

/* ---- native tile queue: job::tiles_t (src/jobs/tiles.hpp:10-90) ----------------------- */
/* make(): row-major tile_size x tile_size tiles with edge remainders (tiles.hpp:49-89);
 * rank/world shard the queue for multi-GPU: tile (tx, ty) belongs to rank (tx + s*ty) % world, s the smallest odd
 * number >= 3 coprime to world (diagonals over the film: balanced whatever the row length). */
typedef struct phx_tiles phx_tiles;
phx_tiles*  phx_tiles_make(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t rank, uint32_t world);
int         phx_tiles_next(void* tiles /* phx_tiles* */, phx_tile* out); /* a phx_next_tile_fn */
uint32_t    phx_tiles_count(const phx_tiles* tiles);
void        phx_tiles_reset(phx_tiles* tiles);
void        phx_tiles_free(phx_tiles* tiles);

/* ---- stage-level entry points (used by the parity tests; all synchronous) ------------- */
/* Closest-hit (shadow==0) or any-hit (shadow==1) trace of n host rays against the device BVH:
 * the device equivalent of stream_mbvh_kernel_t::trace (kernels/cpu/stream_bvh_kernel.cpp:159).
 * o,d: xyz per ray; tmax per ray.  Outputs: t (hit distance, or tmax on miss), u,v, prim
 * (index into scene_t::triangles() order, 0xffffffff on miss; for any-hit 0/1-style hit flag in hit[]). */
int phx_dev_trace(phx_device* dev, uint32_t n, const float* o, const float* d, const float* tmax,
                  int shadow, float* t, float* u, float* v, uint32_t* prim, uint8_t* hit);

/* bsdf_t::f (src/bsdf.cpp:113-131) and bsdf_t::sample (:133-248) evaluated on the device for
 * material `material` with shading normal n[i]: KAT hooks.  Vectors are xyz per item. */
int phx_dev_bsdf_f(phx_device* dev, uint32_t material, uint32_t n_items, const float* n,
                   const float* wi, const float* wo, float* f_out);
int phx_dev_bsdf_sample(phx_device* dev, uint32_t material, uint32_t n_items, const float* n,
                        const float* wi, const float* u2, float* wo_out, float* f_out,
                        float* pdf_out, uint32_t* flags_out);

/* The acceleration structure of the preprocessed scene as the traversal kernels read it (a parity hook: the tests check that every box the
 * device builder stored contains what hangs below it).  Copies min(capacity, size) bytes of the pool of 64-byte elements (csrc/bvh8.h:
 * element 0 = root nodelet) to `out`, stores the pool's size in *bytes and the per-scene grid of the nodelets' origins in grid6
 * (lo.xyz, cell.xyz; origin = fma(i, cell, lo)). */
int phx_dev_copy_bvh(phx_device* dev, void* out, uint64_t capacity, uint64_t* bytes, float* grid6);

#ifdef __cplusplus
}
#endif
#endif /* PHX_XPU_H */
