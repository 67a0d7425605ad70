// device.cpp — host side of the gfx950 device: the C ABI of include/phx_xpu.h.
//
// Mirrors cpu_t (reference src/xpu/cpu.cpp:208-248): preprocess() flattens the scene and builds the
// accelerator (cpu.cpp:35-44), start() spawns a driver thread that drains the shared tile queue
// (cpu.cpp:223-238) — in batches of many tiles, each carried through the wavefront kernels of
// kernels.hip — and hands finished tiles to the film sink (cpu.cpp:201), join() waits for it.
// There is no CPU rendering path in this library: without a gfx950 device every entry point fails.
#include "../../include/phx_xpu.h"
#include "bvh_build.h"
#include "bvh_gpu.h"
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace phx;

#ifndef PHX_TEST_HOOKS
#define PHX_TEST_HOOKS 0  /* 1 only in the twin library libphx_hip_hooks.so (fault injection for the tests); never in the product build */
#endif

namespace {

// Errors: every thread has its own last-error string; a driver thread's error is also kept with its device (phx_device::
// frame_error) and handed to the thread that calls phx_dev_join, so two devices rendering at once never share a buffer.
thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(PHX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Entry points run on the CALLER's thread: they switch to their device and put the caller's current device back on the way out,
// so a host that mixes several phx_devices with its own HIP / torch allocations never finds itself on another GPU.
struct DeviceScope {
  int prev = -1; bool ok = false;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <typename T>
struct DevBuf {
  T* p = nullptr; size_t n = 0;
  int alloc(size_t count) {
    if (count <= n && p) return PHX_OK;
    release();
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return fail(PHX_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    n = count;
    return PHX_OK;
  }
  int upload(const std::vector<T>& h) {
    int rc = alloc(h.size()); if (rc) return rc;
    if (!h.empty()) HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return PHX_OK;
  }
  void adopt(T* ptr, size_t count) { release(); p = ptr; n = count; }  // take ownership of a hipMalloc'd array
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
  size_t bytes() const { return p ? std::max<size_t>(n, 1) * sizeof(T) : 0; }
  ~DevBuf() { release(); }
};

}  // namespace

struct phx_tiles {
  std::vector<phx_tile> tiles;
  std::atomic<uint32_t> cursor{0};
};

struct phx_device {
  phx_options opt{};
  int hip_device = 0;
  hipStream_t stream = nullptr;
  bool preprocessed = false;

  // scene
  DevBuf<PoolElem> d_pool; DevBuf<uint32_t> d_prim_material; DevBuf<float> d_elem_normals; DevBuf<float4> d_elem_shade; DevBuf<uint2> d_spill;
  DevBuf<DevMaterial> d_materials; DevBuf<DevMatLite> d_mat_lite; DevBuf<DevLight> d_lights; DevBuf<DevLightTri> d_light_tris;
  DevScene scene{};
  uint32_t num_materials = 0;
  uint64_t bvh_nodes = 0, bvh_bytes = 0, num_triangles = 0;
  double preprocess_ms = 0, bvh_build_ms = 0;

  // pass buffers
  DevBuf<float4> ro[2], rd[2], qs[2], hit, so, sd, sc, pr, pn;
  DevBuf<uint32_t> counters; DevBuf<DevStats> dstats; DevBuf<uint32_t> pix_xy; DevBuf<float2> jitter; DevBuf<float> acc;
  uint64_t jitter_seed = 0; uint32_t jitter_spp = 0;  // what the jitter table on the device was made for
  float* h_acc = nullptr; size_t h_acc_n = 0;  // pinned staging for add_tile
  uint32_t* h_qlen = nullptr; size_t h_qlen_n = 0;  // pinned, device-visible: queue lengths published by k_trace, one word per (pass, step) of a batch (enqueue_batch)
  std::vector<phx_tile> pix_xy_tiles;          // the tiles pix_xy currently describes

  // frame
  phx_frame frame{};
  // ONE driver thread per device, kept across frames (cpu_t spawns its workers per frame, src/xpu/cpu.cpp:223-238; here a frame of a rank
  // of eight lasts 9 ms and a thread start costs 50-70 us of it): phx_dev_start hands it the frame, phx_dev_join waits for `frame_done`
  std::thread driver;
  std::mutex mu; std::condition_variable cv;
  bool frame_pending = false, frame_done = false, quit = false;
  bool running = false;
  int frame_status = PHX_OK;
  std::string frame_error;  // g_error of the driver thread, copied when its frame ends
  phx_stats stats{};
  TracePlan plan{};
  uint64_t paths_in_flight = 0;
  double bvh_cost_model = 0; uint32_t bvh_built_on_device = 0;
  struct Timed { size_t begin, end; int kind; };  // (event before, event behind, kind 0 k_trace / 2 begin-pass, film / 3 shade / 4 k_trace_primary)
  // The launches of a batch and the HIP events between them.  (Round 5 captured a batch's launches — memset, [begin-pass, camera rays,
  // shade, (trace, shade) x (depth - 1), trace, film] per pass, film scatter — as ONE hipGraph, cached by a hash of the kernel arguments:
  // rank 0 of 8 of the bench frame 9.12-9.27 ms with direct launches and events, 9.08-9.37 without events, 9.14-9.16 as a graph — nothing,
  // the fixed costs of a short frame are k_trace's drain tails on the GPU, not the host's enqueue; and an event recorded by a graph's
  // event-record node cannot be read with hipEventElapsedTime on ROCm 7.2.  profiles/r05_b_graph_rank_share.log, r05_b_graph.patch.)
  struct BatchLaunches {
    std::vector<hipEvent_t> events; size_t events_used = 0;
    std::vector<Timed> timed;
  };
  BatchLaunches direct;
  bool kernel_timing = true;        // per-kernel HIP events (phx_stats::closest_ms ...); PHX_KERNEL_TIMING=0 launches without them (probe)
  std::chrono::steady_clock::time_point t_start, t_enq, t_sync;  // host timing probe (PHX_HOST_TIMING)

  ~phx_device() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    if (driver.joinable()) driver.join();
    for (auto e : direct.events) (void)hipEventDestroy(e);
    if (h_acc) (void)hipHostFree(h_acc);
    if (h_qlen) (void)hipHostFree(h_qlen);
    if (stream) (void)hipStreamDestroy(stream);
  }

  static int next_event(BatchLaunches& g, hipEvent_t* out) {
    if (g.events_used == g.events.size()) {
      hipEvent_t e; HIPCHK(hipEventCreate(&e));
      g.events.push_back(e);
    }
    *out = g.events[g.events_used++];
    return PHX_OK;
  }
  // HBM held by this device object (phx_stats::device_bytes)
  uint64_t device_bytes() const {
    uint64_t b = d_pool.bytes() + d_prim_material.bytes() + d_elem_normals.bytes() + d_elem_shade.bytes() + d_spill.bytes() + d_materials.bytes() + d_mat_lite.bytes() +
                 d_lights.bytes() + d_light_tris.bytes() + hit.bytes() + so.bytes() + sd.bytes() + sc.bytes() + pr.bytes() + pn.bytes() +
                 counters.bytes() + dstats.bytes() + pix_xy.bytes() + jitter.bytes() + acc.bytes();
    for (int q = 0; q < 2; ++q) b += ro[q].bytes() + rd[q].bytes() + qs[q].bytes();
    return b;
  }
  // paths in flight this device may carry: up to 512 M (about 95 GB of queues + state: sized for 288 GB of HBM), but never more than 60 % of
  // what the device has free right now plus what this object already holds for queues (another device object, torch or RCCL may share the GPU)
  mutable uint64_t budget_bytes = 0;  // hipMemGetInfo costs ~0.1 ms a call: asked once per preprocess, not twice per frame
  uint64_t path_budget(size_t path_bytes) const {
    static const uint64_t cap_m = [] { const char* v = std::getenv("PHX_PATH_BUDGET_M"); const long x = v ? std::atol(v) : 0; return x >= 1 && x <= 1536 ? (uint64_t)x : 512ull; }();  // knob: millions of paths (128 / 256 / 512: config 4 on one GPU 6 987 / 7 182 / 7 314 Mrays/s, profiles/r04_x_budget.log)
    if (!budget_bytes) {
      budget_bytes = ~0ull;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        size_t held = 0;
        for (int q = 0; q < 2; ++q) held += (ro[q].n + rd[q].n + qs[q].n) * sizeof(float4);
        held += (hit.n + so.n + sd.n + sc.n + pr.n + pn.n) * sizeof(float4);
        budget_bytes = (uint64_t)((double)(free_b + held) * 0.6);
      }
    }
    return std::min<uint64_t>(cap_m << 20, budget_bytes / path_bytes);
  }
  void driver_loop();
  int run_frame();
  int enqueue_batch(BatchLaunches& g, const PassBuffers& B0, uint32_t P, uint32_t S, uint32_t xs);
  int render_batch(const std::vector<phx_tile>& tiles, const std::vector<float2>& jit);
};

// ---- scene flattening --------------------------------------------------------------------------------
namespace {

// microfacet_t::roughness_to_alpha + precompute (src/bsdf/params.hpp:86-99), with the device's logf_
float roughness_to_alpha(float roughness) {
  roughness = std::max(roughness, (float)1e-5);
  float x = logf_(roughness);
  return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

int bake_material(const phx_material& m, float sheen_L5, DevMaterial& out) {
  std::memset(&out, 0, sizeof(out));
  out.is_emitter = m.is_emitter; out.ex = m.emission[0]; out.ey = m.emission[1]; out.ez = m.emission[2];
  out.sheen_L5 = sheen_L5;
  if (m.num_lobes > PHX_MAX_LOBES) return 1;
  uint32_t k = 0;
  for (uint32_t i = 0; i < m.num_lobes; ++i) {
    const phx_lobe& s = m.lobes[i];
    DevLobe& l = out.lobes[k];
    l.type = s.type; l.wx = s.weight[0]; l.wy = s.weight[1]; l.wz = s.weight[2];
    l.fac_mode = s.fac_mode; l.fac_ior = s.fac_ior; l.px = s.pre_weight[0]; l.py = s.pre_weight[1]; l.pz = s.pre_weight[2];
    if (s.fac_mode > PHX_FAC_MIX_A) return 1;
    if (s.fac_mode != PHX_FAC_NONE) out.per_hit = 1;
    switch (s.type) {
      case PHX_LOBE_DIFFUSE: l.flags = B_REFLECT | B_DIFFUSE; break;
      case PHX_LOBE_OREN_NAYAR: {  // oren_nayar_t::precompute, params.hpp:36-43
        l.flags = B_REFLECT | B_DIFFUSE;
        const float sg = (float)((double)s.alpha * (kPiD / (double)180.0f));
        const float s2 = sg * sg;
        l.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
        l.b = 0.45f * s2 / (s2 + 0.09f);
        break;
      }
      case PHX_LOBE_REFLECTION: l.flags = B_REFLECT | B_SPECULAR; l.eta = s.eta; break;
      case PHX_LOBE_REFRACTION: l.flags = B_TRANSMIT | B_SPECULAR; l.eta = s.eta; break;
      case PHX_LOBE_MICROFACET:
        l.flags = s.refract ? B_TRANSMIT : B_REFLECT;  // src/bsdf.hpp:70-72
        l.eta = s.eta; l.refract = s.refract;
        l.xalpha = std::min(1.0f, std::max(0.0001f, roughness_to_alpha(s.xalpha)));
        l.yalpha = std::min(1.0f, std::max(0.0001f, roughness_to_alpha(s.yalpha)));
        break;
      case PHX_LOBE_SHEEN: l.flags = B_REFLECT | B_GLOSSY; l.r = s.r; break;
      case PHX_LOBE_TRANSPARENT: l.flags = B_TRANSMIT; break;  // src/material.cpp:98-103
      case PHX_LOBE_EMISSIVE: case PHX_LOBE_BACKGROUND: continue;  // not lobes (material.cpp:240-245)
      default: return 1;
    }
    ++k;
  }
  out.num_lobes = k;
  return 0;
}

// No C++ exception may cross the C ABI (the reference's own std::runtime_error cases become status codes): every entry
// point that allocates or spawns runs its body through guarded().
template <typename F>
int guarded(F&& body) {
  try { return body(); }
  catch (const std::bad_alloc&) { return fail(PHX_ERR_OOM, "host memory allocation failed"); }
  catch (const std::exception& e) { return fail(PHX_ERR_DEVICE, std::string("unexpected exception: ") + e.what()); }
  catch (...) { return fail(PHX_ERR_DEVICE, "unexpected exception"); }
}

}  // namespace

extern "C" {

const char* phx_last_error(void) { return g_error.c_str(); }  // last error of the CALLING thread

int phx_discover(const phx_options* options, int* num_devices) {
  if (!num_devices) return fail(PHX_ERR_ARG, "phx_discover: null out pointer");
  *num_devices = 0;
  if (options && options->host_only) return PHX_OK;  // --no-gpu (src/core.cpp:49-52)
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(PHX_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  int usable = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  *num_devices = usable;
  if (!usable) return fail(PHX_ERR_NO_DEVICE, "no gfx950 (MI355X) device visible");
  return PHX_OK;
}

phx_device* phx_dev_make(const phx_options* options) {
  if (!options) { fail(PHX_ERR_ARG, "phx_dev_make: null options"); return nullptr; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); fail(PHX_ERR_NO_DEVICE, "no HIP device: the gfx950 path cannot run"); return nullptr; }
  int dev = options->device_ordinal;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= n) { fail(PHX_ERR_ARG, "device_ordinal out of range"); return nullptr; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess || std::strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    fail(PHX_ERR_NO_DEVICE, std::string("device is not gfx950: ") + p.gcnArchName);
    return nullptr;
  }
  DeviceScope on(dev);  // the caller's current device is restored when this returns
  if (!on.ok) { fail(PHX_ERR_DEVICE, "hipSetDevice failed"); return nullptr; }
  phx_device* d = new (std::nothrow) phx_device();
  if (!d) { fail(PHX_ERR_OOM, "host memory allocation failed"); return nullptr; }
  d->opt = *options; d->hip_device = dev;
  if (d->opt.samples_per_pixel == 0) d->opt.samples_per_pixel = 16;
  if (d->opt.paths_per_sample == 0) d->opt.paths_per_sample = 1;
  if (d->opt.path_depth == 0) d->opt.path_depth = 9;
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { delete d; fail(PHX_ERR_DEVICE, "hipStreamCreate failed"); return nullptr; }
  // dynamic-LDS limit of the traversal kernels: a per-device, per-kernel attribute, so it is set for every device made
  const hipError_t ae = init_kernels_on_current_device();
  if (ae != hipSuccess) { delete d; fail(PHX_ERR_DEVICE, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(ae)); return nullptr; }
  return d;
}

void phx_dev_destroy(phx_device* dev) {
  if (!dev) return;
  DeviceScope on(dev->hip_device);
  delete dev;
}

static int preprocess_impl(phx_device* d, const phx_scene* s);
int phx_dev_preprocess(phx_device* d, const phx_scene* s) { return guarded([&]() { return preprocess_impl(d, s); }); }
static int preprocess_impl(phx_device* d, const phx_scene* s) {
  if (!d || !s) return fail(PHX_ERR_ARG, "preprocess: null argument");
  if (d->running) return fail(PHX_ERR_STATE, "preprocess while a frame is running");
  if (!s->meshes || !s->materials || s->num_materials == 0) return fail(PHX_ERR_ARG, "scene without meshes/materials");
  // (camera_t's constructor leaves focal_distance uninitialised, entities/camera.hpp:31-36: it means something only behind a lens)
  if (!(std::fabs(s->camera.aperture_radius) <= FLT_MAX) || (s->camera.aperture_radius != 0.0f && !(std::fabs(s->camera.focal_distance) <= FLT_MAX)))
    return fail(PHX_ERR_ARG, "camera: aperture radius / focal distance not finite");
  if (s->camera.film_width == 0 || s->camera.film_height == 0 || s->camera.film_width > 65535 || s->camera.film_height > 65535)
    return fail(PHX_ERR_ARG, "film size out of range");
  if (s->environment_material >= (int32_t)s->num_materials) return fail(PHX_ERR_ARG, "environment material out of range");
  DeviceScope on(d->hip_device);
  if (!on.ok) return fail(PHX_ERR_DEVICE, "hipSetDevice failed");
  const auto t_pre0 = std::chrono::steady_clock::now();

  // triangles in scene_t::triangles() order: mesh order x face-set order (scene.cpp:58-62, mesh.cpp:118-128)
  std::vector<float> abc; std::vector<uint32_t> prim_material; std::vector<float> prim_normals;
  std::vector<DevLight> lights; std::vector<DevLightTri> light_tris;
  bool any_smooth = false;
  for (uint32_t mi = 0; mi < s->num_meshes; ++mi) {
    const phx_mesh& m = s->meshes[mi];
    for (uint32_t f = 0; f < m.num_faces; ++f) if (m.smooth && m.smooth[f]) any_smooth = true;
  }
  for (uint32_t mi = 0; mi < s->num_meshes; ++mi) {
    const phx_mesh& m = s->meshes[mi];
    if (!m.vertices || !m.faces || (m.num_sets && !m.sets)) return fail(PHX_ERR_ARG, "mesh with null arrays");
    for (uint32_t si = 0; si < m.num_sets; ++si) {
      const phx_face_set& fs = m.sets[si];
      if (fs.material >= s->num_materials) return fail(PHX_ERR_ARG, "face set material out of range");
      const bool emitter = s->materials[fs.material].is_emitter != 0;
      DevLight L{(uint32_t)light_tris.size(), 0, 0.0f, fs.material, 0.0f, 0.0f, 0.0f, 0.0f};
      for (uint32_t k = 0; k < fs.num_faces; ++k) {
        const uint32_t f = fs.faces[k];
        if (f >= m.num_faces) return fail(PHX_ERR_ARG, "face index out of range");
        const uint32_t ia = m.faces[3 * f], ib = m.faces[3 * f + 1], ic = m.faces[3 * f + 2];
        if (ia >= m.num_vertices || ib >= m.num_vertices || ic >= m.num_vertices) return fail(PHX_ERR_ARG, "vertex index out of range");
        const uint32_t prim = (uint32_t)prim_material.size();
        const float* a = m.vertices + 3 * (size_t)ia; const float* b = m.vertices + 3 * (size_t)ib; const float* c = m.vertices + 3 * (size_t)ic;
        abc.insert(abc.end(), a, a + 3); abc.insert(abc.end(), b, b + 3); abc.insert(abc.end(), c, c + 3);
        const bool smooth = m.smooth && m.smooth[f];
        prim_material.push_back(fs.material | (smooth ? 0x80000000u : 0u));
        if (any_smooth) {
          uint32_t na = ia, nb = ib, nc = ic;
          if (!(m.flags & PHX_MESH_NORMALS_PER_VERTEX)) { na = 3 * f; nb = 3 * f + 1; nc = 3 * f + 2; }  // mesh.cpp:188-192
          float nn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          if (smooth) {
            if (!m.normals || na >= m.num_normals || nb >= m.num_normals || nc >= m.num_normals) return fail(PHX_ERR_ARG, "normal index out of range");
            std::memcpy(nn, m.normals + 3 * (size_t)na, 12); std::memcpy(nn + 3, m.normals + 3 * (size_t)nb, 12); std::memcpy(nn + 6, m.normals + 3 * (size_t)nc, 12);
          }
          prim_normals.insert(prim_normals.end(), nn, nn + 9);
        }
        if (emitter) {  // mesh_t::preprocess -> light_t::make_area (mesh.cpp:108-116), area_light_t (light.cpp:10-45)
          const v3 ab(b[0] - a[0], b[1] - a[1], b[2] - a[2]), ac(c[0] - a[0], c[1] - a[1], c[2] - a[2]);
          const v3 gn = normalize_inplace(cross(ab, ac));  // the flat face's normal as k_shade's shading_normal would compute it per sample
          DevLightTri T{a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], gn.x, gn.y, gn.z, prim, smooth ? 1u : 0u, mi | (fs.material << 16), 3 * f};
          light_tris.push_back(T);
          L.area += 0.5f * length(cross(ab, ac));  // triangle_t::area, mesh.cpp:293-300; summed in face order (light.cpp:36-39)
          L.num_tris++;
        }
      }
      if (emitter && L.num_tris) lights.push_back(L);
    }
  }
  if (prim_material.empty()) return fail(PHX_ERR_ARG, "scene has no triangles");
  if (lights.empty()) return fail(PHX_ERR_ARG, "scene has no emissive face set (reference underflows nlights-1, SURVEY A-19)");

  // sheen_L5: the first sheen lobe of the material table (bsdf.h)
  float L5 = 0.0f; bool have = false;
  for (uint32_t i = 0; i < s->num_materials && !have; ++i)
    for (uint32_t k = 0; k < s->materials[i].num_lobes && k < PHX_MAX_LOBES; ++k)
      if (s->materials[i].lobes[k].type == PHX_LOBE_SHEEN) { L5 = sheen_L(0.5f, s->materials[i].lobes[k].r); have = true; break; }
  std::vector<DevMaterial> mats(s->num_materials);
  for (uint32_t i = 0; i < s->num_materials; ++i)
    if (bake_material(s->materials[i], L5, mats[i])) return fail(PHX_ERR_ARG, "material with an unknown closure id");

  {  // per light: the pick pdf and the emission of its material, as k_shade evaluated them per sample until round 2
    const float nlf = (float)lights.size();
    for (auto& L : lights) {
      L.lpdf = (1.0f / L.area) / nlf;
      if (L.material >= s->num_materials) return fail(PHX_ERR_ARG, "light with a material index out of range");
      L.ex = mats[L.material].ex; L.ey = mats[L.material].ey; L.ez = mats[L.material].ez;
    }
  }
  int rc;
  if ((rc = d->d_prim_material.upload(prim_material))) return rc;
  const auto t_bvh0 = std::chrono::steady_clock::now();
  // PHX_BVH_AUTO: the device builder (bvh_gpu.hip) unless the scene is tiny.  Round 2 kept the host's binned SAH for scenes up to 2 M
  // triangles because its trees traced 3-5 % faster on mesh-like scenes; with extended Morton codes (the size of a primitive as a
  // fourth coordinate) the device trees are as fast or faster everywhere measured — soups -2 ... -7 % k_trace time, the showroom
  // -3 % — and they are built in milliseconds (profiles/r03_z_emc_probe.log, r03_za_builder_ab.log).
  const uint32_t builder = d->opt.bvh_builder;
  const uint32_t ntri = (uint32_t)prim_material.size();
  if (builder > PHX_BVH_HOST_SAH) return fail(PHX_ERR_ARG, "unknown bvh_builder");
  bool want_host = builder == PHX_BVH_HOST_SAH || (builder == PHX_BVH_AUTO && ntri < 64u);
  GpuBvh g{};
  DevBuf<uint32_t> d_elem_of_prim;  // pool index of every primitive's triangle record (the shade records and the normals table are laid out by it)
  if (!want_host) {
    // the triangles go up once (36 B each); the tree is built and stays in HBM (bvh_gpu.hip)
    DevBuf<float> d_abc;
    char msg[256] = {0};
    rc = d_abc.upload(abc);
    int brc = rc ? (rc == PHX_ERR_OOM ? (int)BVH_GPU_RECOVERABLE : 1) : 0;
    if (rc) std::snprintf(msg, sizeof(msg), "%s", g_error.c_str());
#if PHX_TEST_HOOKS
    // test hook, compiled into the twin library libphx_hip_hooks.so only (tests/test_gpu_parity.py): makes the device build report a
    // recoverable or a fatal failure
    if (const char* how = std::getenv("PHX_TEST_FAIL_DEVICE_BUILD")) {
      brc = std::strcmp(how, "fatal") == 0 ? 1 : (int)BVH_GPU_RECOVERABLE;
      std::snprintf(msg, sizeof(msg), "forced %s failure (PHX_TEST_FAIL_DEVICE_BUILD)", brc == 1 ? "fatal" : "recoverable");
    }
#endif
    if (!brc && d_elem_of_prim.alloc(ntri)) brc = (int)BVH_GPU_RECOVERABLE, std::snprintf(msg, sizeof(msg), "%s", g_error.c_str());
    if (!brc) brc = build_bvh8_gpu(d->stream, d_abc.p, d->d_prim_material.p, ntri, &g, msg, sizeof(msg), d_elem_of_prim.p);
    if (brc) {
      // An explicit DEVICE_LBVH request fails loudly, and so does AUTO when the device builder reports anything but a RECOVERABLE cause
      // (a HIP error from a launch or a sync, lost triangles: bugs that a silent 0.6-7 s host build would hide).  Under AUTO a device
      // build that cannot get its scratch memory, or meets a tree deeper than its tables, falls back to the host's binned-SAH builder
      // — which handled every scene before the device builder became the default — and says so: on stderr, in phx_last_error of this
      // thread (a successful call leaves the text in place) and in phx_stats::bvh_built_on_device.
      if (builder != PHX_BVH_AUTO || brc != (int)BVH_GPU_RECOVERABLE) return fail(rc ? rc : PHX_ERR_DEVICE, std::string("device BVH build: ") + msg);
      (void)hipGetLastError();  // a failed hipMalloc leaves its error behind
      g_error = std::string("device BVH build fell back to the host builder: ") + msg;
      std::fprintf(stderr, "libphx_hip: %s\n", g_error.c_str());
      want_host = true;
    }
  }
  uint32_t bvh_depth = 0; size_t bvh_node_count = 0, bvh_elems = 0;
  SceneGrid bvh_grid{};
  if (want_host) {
    Bvh8 bvh;
    const int threads = (int)std::max(1u, std::thread::hardware_concurrency());
    build_bvh8(abc.data(), ntri, bvh, threads, prim_material.data());
    if ((rc = d->d_pool.upload(bvh.pool))) return rc;
    if ((rc = d_elem_of_prim.upload(bvh.elem_of_prim))) return rc;
    bvh_depth = bvh.depth; bvh_node_count = bvh.num_nodes; bvh_elems = bvh.pool.size(); bvh_grid = bvh.grid;
    d->bvh_cost_model = bvh.cost; d->bvh_built_on_device = 0;
  } else {
    d->d_pool.adopt(g.pool, g.num_elems);
    bvh_depth = g.depth; bvh_node_count = g.num_nodes; bvh_elems = g.num_elems; bvh_grid = g.grid;
    d->bvh_cost_model = g.cost; d->bvh_built_on_device = 1;
  }
  const auto t_bvh1 = std::chrono::steady_clock::now();
  // k_trace / k_trace_rays keep one pending sibling group per level and lane in LDS (bvh8.h: PHX_MAX_BVH_DEPTH)
  if (bvh_depth > PHX_MAX_BVH_DEPTH)
    return fail(PHX_ERR_ARG, "tree too deep: " + std::to_string(bvh_depth) + " levels, the traversal stack in LDS holds " + std::to_string(PHX_MAX_BVH_DEPTH));
  if ((rc = d->d_materials.upload(mats))) return rc;
  if ((rc = d->d_lights.upload(lights))) return rc;
  if ((rc = d->d_light_tris.upload(light_tris))) return rc;
  // what shading reads of a hit triangle, 16 bytes per POOL ELEMENT: geometric normal + material word
  if ((rc = d->d_elem_shade.alloc(bvh_elems))) return rc;
  launch_build_shade_recs(d->stream, reinterpret_cast<const TriRec*>(d->d_pool.p), d_elem_of_prim.p, d->d_elem_shade.p, ntri);
  HIPCHK(hipGetLastError());
  if (any_smooth) {
    // vertex normals by POOL ELEMENT (the index a hit record carries), so that the shade kernels request them with the triangle record and not
    // after it; the smooth light triangles' `prim` becomes a pool index too (shading_normal on the light's face, spt.hpp:212-255)
    DevBuf<float> d_prim_normals;
    if ((rc = d_prim_normals.upload(prim_normals))) return rc;
    if ((rc = d->d_elem_normals.alloc(9 * bvh_elems))) return rc;
    launch_permute_normals(d->stream, d_prim_normals.p, d_elem_of_prim.p, d->d_elem_normals.p, ntri);
    launch_remap_light_tris(d->stream, d->d_light_tris.p, (uint32_t)light_tris.size(), d_elem_of_prim.p);
    HIPCHK(hipGetLastError());
  } else {
    d->d_elem_normals.release();
  }
  HIPCHK(hipStreamSynchronize(d->stream));  // d_elem_of_prim (and the normals in primitive order) go out of scope below

  DevScene& sc = d->scene;
  sc.pool = reinterpret_cast<const uint32_t*>(d->d_pool.p);
  sc.tris = reinterpret_cast<const TriRec*>(d->d_pool.p);
  sc.grid = bvh_grid;
  sc.prim_material = d->d_prim_material.p;
  sc.elem_normals = any_smooth ? d->d_elem_normals.p : nullptr;
  sc.elem_shade = d->d_elem_shade.p;
  sc.materials = d->d_materials.p;
  sc.lights = d->d_lights.p; sc.light_tris = d->d_light_tris.p; sc.num_lights = (uint32_t)lights.size();
  sc.env_material = s->environment_material;
  std::memcpy(sc.cam_m, s->camera.to_world, sizeof(sc.cam_m));
  sc.zoom = 1.12f * std::tan(s->camera.fov * 0.5f);  // camera.hpp:113
  sc.stepx = 1.0f / (float)s->camera.film_width; sc.stepy = 1.0f / (float)s->camera.film_height;
  sc.ratio = (float)s->camera.film_width / (float)s->camera.film_height;
  sc.width = s->camera.film_width; sc.height = s->camera.film_height;
  sc.aperture_radius = s->camera.aperture_radius; sc.focal_distance = s->camera.focal_distance;  // thin lens iff aperture_radius != 0 (camera_t::is_pinhole)
  sc.max_depth = d->opt.path_depth;
  sc.stack_levels = bvh_depth;
  sc.num_elems = (uint32_t)bvh_elems;
  {
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, d->hip_device));
    sc.num_cus = (uint32_t)prop.multiProcessorCount;
  }
  sc.diffuse_only = 1;
  for (auto& m : mats) { if (m.per_hit) sc.diffuse_only = 0; for (uint32_t k = 0; k < m.num_lobes; ++k) if (m.lobes[k].type != L_DIFFUSE) sc.diffuse_only = 0; }
  sc.mat_lite = nullptr;
  sc.any_per_hit = 0;
  for (auto& m : mats) if (m.per_hit) sc.any_per_hit = 1;
  if (sc.diffuse_only) {  // at most one Lambert lobe everywhere (the soups, the Cornell box): a 32-byte material table for k_shade<2>
    bool single = true;
    for (auto& m : mats) single = single && m.num_lobes <= 1;
    if (single) {
      std::vector<DevMatLite> lite(mats.size());
      for (size_t i = 0; i < mats.size(); ++i) {
        const DevMaterial& m = mats[i];
        lite[i] = DevMatLite{m.lobes[0].wx, m.lobes[0].wy, m.lobes[0].wz, m.num_lobes | (m.lobes[0].flags << 8), m.ex, m.ey, m.ez, 0u};
        if (m.num_lobes == 0) { lite[i].wx = lite[i].wy = lite[i].wz = 0.0f; lite[i].lobes_flags = 0; }
      }
      if ((rc = d->d_mat_lite.upload(lite))) return rc;
      sc.mat_lite = d->d_mat_lite.p;
      sc.diffuse_only = 2;
    }
  }
  d->num_materials = s->num_materials;
  d->bvh_nodes = bvh_node_count;
  d->bvh_bytes = bvh_elems * sizeof(PoolElem);
  d->bvh_build_ms = std::chrono::duration<double, std::milli>(t_bvh1 - t_bvh0).count();
  d->preprocess_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pre0).count();
  d->num_triangles = prim_material.size();
  sc.stack_spill = nullptr; sc.spill_stride = 0;
  d->plan = trace_plan(sc);
  if (d->plan.spill_threads) {  // stack levels below the ones k_trace keeps in LDS (kernels.hip: trace_plan)
    if ((rc = d->d_spill.alloc((size_t)d->plan.spill_threads * (d->plan.levels - d->plan.lds_levels)))) return rc;
    sc.stack_spill = d->d_spill.p; sc.spill_stride = d->plan.spill_threads;
  }
  d->budget_bytes = 0;  // the next frame asks the device again how much memory is free
  d->preprocessed = true;
  return PHX_OK;
}

int phx_dev_start(phx_device* d, const phx_frame* f) {
  if (!d || !f) return fail(PHX_ERR_ARG, "start: null argument");
  if (!d->preprocessed) return fail(PHX_ERR_STATE, "start before preprocess");
  if (d->running) return fail(PHX_ERR_STATE, "start while a frame is running");
  if (!f->next_tile) return fail(PHX_ERR_ARG, "frame without a tile queue");
  if (!f->add_tile && !f->device_film && !f->host_film) return fail(PHX_ERR_ARG, "frame without a film sink");
  if (f->primary_components != 3 && f->primary_components != 4) return fail(PHX_ERR_ARG, "primary channel must have 3 or 4 components");
  d->frame = *f;
  {
    static const int timing_env = [] { const char* v = std::getenv("PHX_KERNEL_TIMING"); return v ? std::atoi(v) : -1; }();
    d->kernel_timing = timing_env >= 0 ? timing_env != 0 : true;
  }
  d->t_start = std::chrono::steady_clock::now();
  const int rc = guarded([&]() {
    if (!d->driver.joinable()) d->driver = std::thread([d]() { d->driver_loop(); });  // the first frame of this device starts its driver
    return (int)PHX_OK;
  });
  if (rc != PHX_OK) return rc;  // the driver thread could not be created
  {
    std::lock_guard<std::mutex> lk(d->mu);
    d->running = true; d->frame_status = PHX_OK; d->frame_done = false; d->frame_pending = true;
  }
  d->cv.notify_all();
  return PHX_OK;
}

int phx_dev_join(phx_device* d) {
  if (!d) return fail(PHX_ERR_ARG, "join: null device");
  if (!d->running) return fail(PHX_ERR_STATE, "join without start");
  {
    std::unique_lock<std::mutex> lk(d->mu);
    d->cv.wait(lk, [d]() { return d->frame_done; });
    d->running = false;
  }
  if (d->frame_status != PHX_OK) g_error = d->frame_error;
  return d->frame_status;
}

int phx_dev_get_stats(const phx_device* d, phx_stats* out) {
  if (!d || !out) return fail(PHX_ERR_ARG, "get_stats: null argument");
  *out = d->stats;
  out->bvh_nodes = d->bvh_nodes; out->bvh_bytes = d->bvh_bytes; out->triangles = d->num_triangles;
  out->preprocess_ms = d->preprocess_ms; out->bvh_build_ms = d->bvh_build_ms;
  out->trace_block = d->plan.block; out->trace_ntop = d->plan.ntop; out->trace_levels = d->plan.levels; out->trace_lds_levels = d->plan.lds_levels; out->trace_stack_packed = d->plan.packed;
  out->trace_waves_per_cu = (uint64_t)d->plan.wg_per_cu * (d->plan.block / 64u); out->bvh_depth = d->scene.stack_levels;
  out->paths_in_flight = d->paths_in_flight;
  out->bvh_cost_model = d->bvh_cost_model; out->bvh_built_on_device = d->bvh_built_on_device;
  out->shade_general = d->scene.diffuse_only ? 0 : 1;
  out->device_bytes = d->device_bytes();
  return PHX_OK;
}

uint32_t phx_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(phx_options); case 1: return sizeof(phx_lobe); case 2: return sizeof(phx_material);
    case 3: return sizeof(phx_face_set); case 4: return sizeof(phx_mesh); case 5: return sizeof(phx_camera);
    case 6: return sizeof(phx_scene); case 7: return sizeof(phx_tile); case 8: return sizeof(phx_frame);
    case 9: return sizeof(phx_stats);
    default: return 0;
  }
}

// ---- tile queue: job::tiles_t (src/jobs/tiles.hpp:10-90) ------------------------------------------------
phx_tiles* phx_tiles_make(uint32_t width, uint32_t height, uint32_t ts, uint32_t rank, uint32_t world) {
  if (ts == 0 || world == 0 || rank >= world) { fail(PHX_ERR_ARG, "phx_tiles_make: bad arguments"); return nullptr; }
  uint32_t ht = width / ts, vt = height / ts;
  const uint32_t rh = height - ts * vt, rw = width - ts * ht;
  if (rh > 0) vt++;
  if (rw > 0) ht++;
  phx_tiles* q = new (std::nothrow) phx_tiles();
  if (!q) { fail(PHX_ERR_OOM, "host memory allocation failed"); return nullptr; }
  try {
  // owner of tile (x, y) = (x + s*y) mod world, s the smallest odd number >= 3 coprime to world: every rank's tiles run in
  // diagonals over the whole film.  (Plain "tile id mod world" degenerates into vertical stripes whenever the row length is
  // a multiple of world — 40 tiles per row at 1280 px — and the stripes under the light cost 8 % more rays than the mean.)
  uint32_t s = 3;
  for (;; s += 2) { uint32_t a = s, b = world; while (b) { const uint32_t t = a % b; a = b; b = t; } if (a == 1) break; }
  for (uint32_t y = 0; y < vt; ++y)
    for (uint32_t x = 0; x < ht; ++x) {
      uint32_t tw = ts, th = ts;
      if (y == vt - 1 && rh > 0) th = rh;
      if (x == ht - 1 && rw > 0) tw = rw;
      if ((x + s * y) % world == rank) q->tiles.push_back(phx_tile{x * ts, y * ts, tw, th});
    }
  } catch (...) { delete q; fail(PHX_ERR_OOM, "host memory allocation failed"); return nullptr; }
  return q;
}
int phx_tiles_next(void* tiles, phx_tile* out) {
  phx_tiles* q = (phx_tiles*)tiles;
  const uint32_t t = q->cursor++;
  if (t < q->tiles.size()) { *out = q->tiles[t]; return 1; }
  return 0;
}
uint32_t phx_tiles_count(const phx_tiles* q) { return q ? (uint32_t)q->tiles.size() : 0; }
void phx_tiles_reset(phx_tiles* q) { if (q) q->cursor = 0; }
void phx_tiles_free(phx_tiles* q) { delete q; }

// ---- stage-level entry points ----------------------------------------------------------------------------
static int dev_trace_impl(phx_device* d, uint32_t n, const float* o, const float* dir, const float* tmax, int shadow,
                  float* t, float* u, float* v, uint32_t* prim, uint8_t* hit);
int phx_dev_trace(phx_device* d, uint32_t n, const float* o, const float* dir, const float* tmax, int shadow,
                  float* t, float* u, float* v, uint32_t* prim, uint8_t* hit) { return guarded([&]() { return dev_trace_impl(d, n, o, dir, tmax, shadow, t, u, v, prim, hit); }); }
static int dev_trace_impl(phx_device* d, uint32_t n, const float* o, const float* dir, const float* tmax, int shadow,
                  float* t, float* u, float* v, uint32_t* prim, uint8_t* hit) {
  if (!d || !d->preprocessed) return fail(PHX_ERR_STATE, "trace before preprocess");
  if (n == 0) return PHX_OK;
  DeviceScope on(d->hip_device);
  if (!on.ok) return fail(PHX_ERR_DEVICE, "hipSetDevice failed");
  std::vector<float4> ro(n), rd(n), hh(n);
  for (uint32_t i = 0; i < n; ++i) {
    ro[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], 0.0f);
    rd[i] = make_float4(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], tmax[i]);
  }
  DevBuf<float4> a, b, c; int rc;
  if ((rc = a.upload(ro)) || (rc = b.upload(rd)) || (rc = c.alloc(n))) return rc;
  launch_trace_rays(d->stream, d->scene, n, a.p, b.p, c.p, shadow);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipMemcpy(hh.data(), c.p, n * sizeof(float4), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t tri; std::memcpy(&tri, &hh[i].w, 4);
    if (t) t[i] = hh[i].x;
    if (u) u[i] = hh[i].y;
    if (v) v[i] = hh[i].z;
    if (prim) prim[i] = tri;  // k_trace_rays stores the primitive id (scene_t::triangles() order), 0xffffffff on a miss
    if (hit) hit[i] = tri != 0xffffffffu;
  }
  return PHX_OK;
}

static int kat_upload(const float* src, size_t n, DevBuf<float>& dst) {
  int rc = dst.alloc(n); if (rc) return rc;
  HIPCHK(hipMemcpy(dst.p, src, n * sizeof(float), hipMemcpyHostToDevice));
  return PHX_OK;
}

static int dev_bsdf_f_impl(phx_device* d, uint32_t material, uint32_t n, const float* n3, const float* wi3, const float* wo3, float* f3);
int phx_dev_bsdf_f(phx_device* d, uint32_t material, uint32_t n, const float* n3, const float* wi3, const float* wo3, float* f3) { return guarded([&]() { return dev_bsdf_f_impl(d, material, n, n3, wi3, wo3, f3); }); }
static int dev_bsdf_f_impl(phx_device* d, uint32_t material, uint32_t n, const float* n3, const float* wi3, const float* wo3, float* f3) {
  if (!d || !d->preprocessed) return fail(PHX_ERR_STATE, "bsdf_f before preprocess");
  if (material >= d->num_materials) return fail(PHX_ERR_ARG, "material out of range");
  if (n == 0) return PHX_OK;
  DeviceScope on(d->hip_device);
  if (!on.ok) return fail(PHX_ERR_DEVICE, "hipSetDevice failed");
  DevBuf<float> a, b, c, o; int rc;
  if ((rc = kat_upload(n3, 3 * (size_t)n, a)) || (rc = kat_upload(wi3, 3 * (size_t)n, b)) || (rc = kat_upload(wo3, 3 * (size_t)n, c)) || (rc = o.alloc(3 * (size_t)n))) return rc;
  launch_bsdf_f(d->stream, d->d_materials.p + material, n, a.p, b.p, c.p, o.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipMemcpy(f3, o.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return PHX_OK;
}

static int dev_bsdf_sample_impl(phx_device* d, uint32_t material, uint32_t n, const float* n3, const float* wi3, const float* u2,
                        float* wo3, float* f3, float* pdf, uint32_t* flags);
int phx_dev_bsdf_sample(phx_device* d, uint32_t material, uint32_t n, const float* n3, const float* wi3, const float* u2,
                        float* wo3, float* f3, float* pdf, uint32_t* flags) { return guarded([&]() { return dev_bsdf_sample_impl(d, material, n, n3, wi3, u2, wo3, f3, pdf, flags); }); }
static int dev_bsdf_sample_impl(phx_device* d, uint32_t material, uint32_t n, const float* n3, const float* wi3, const float* u2,
                        float* wo3, float* f3, float* pdf, uint32_t* flags) {
  if (!d || !d->preprocessed) return fail(PHX_ERR_STATE, "bsdf_sample before preprocess");
  if (material >= d->num_materials) return fail(PHX_ERR_ARG, "material out of range");
  if (n == 0) return PHX_OK;
  DeviceScope on(d->hip_device);
  if (!on.ok) return fail(PHX_ERR_DEVICE, "hipSetDevice failed");
  DevBuf<float> a, b, c, wo, f, p; DevBuf<uint32_t> fl; int rc;
  if ((rc = kat_upload(n3, 3 * (size_t)n, a)) || (rc = kat_upload(wi3, 3 * (size_t)n, b)) || (rc = kat_upload(u2, 2 * (size_t)n, c)) ||
      (rc = wo.alloc(3 * (size_t)n)) || (rc = f.alloc(3 * (size_t)n)) || (rc = p.alloc(n)) || (rc = fl.alloc(n))) return rc;
  launch_bsdf_sample(d->stream, d->d_materials.p + material, n, a.p, b.p, c.p, wo.p, f.p, p.p, fl.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipMemcpy(wo3, wo.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(f3, f.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(pdf, p.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(flags, fl.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PHX_OK;
}

int phx_dev_copy_bvh(phx_device* d, void* out, uint64_t capacity, uint64_t* bytes, float* grid6) {
  if (!d || !d->preprocessed) return fail(PHX_ERR_STATE, "copy_bvh before preprocess");
  if (!bytes) return fail(PHX_ERR_ARG, "copy_bvh: null size pointer");
  DeviceScope on(d->hip_device);
  if (!on.ok) return fail(PHX_ERR_DEVICE, "hipSetDevice failed");
  *bytes = d->bvh_bytes;
  if (grid6) for (int a = 0; a < 3; ++a) { grid6[a] = d->scene.grid.lo[a]; grid6[3 + a] = d->scene.grid.cell[a]; }
  const uint64_t n = std::min<uint64_t>(capacity, d->bvh_bytes);
  if (out && n) HIPCHK(hipMemcpy(out, d->d_pool.p, n, hipMemcpyDeviceToHost));
  return PHX_OK;
}

}  // extern "C"

// ---- the frame driver ------------------------------------------------------------------------------------
void phx_device::driver_loop() {
  (void)hipSetDevice(hip_device);
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this]() { return frame_pending || quit; });
      if (!frame_pending) return;  // quit, and no frame handed over: a frame that WAS started is rendered first (phx_dev_destroy joins it, include/phx_xpu.h)
      frame_pending = false;
    }
    const int st = guarded([this]() { return run_frame(); });
    {
      std::lock_guard<std::mutex> lk(mu);
      frame_status = st;
      frame_error = st != PHX_OK ? g_error : std::string();
      frame_done = true;
    }
    cv.notify_all();
  }
}

int phx_device::run_frame() {
  HIPCHK(hipSetDevice(hip_device));
  const auto t0 = std::chrono::steady_clock::now();
  std::memset(&stats, 0, sizeof(stats));
  int rc;
  if ((rc = dstats.alloc(1)) || (rc = counters.alloc(CNT_WORDS))) return rc;
  HIPCHK(hipMemsetAsync(dstats.p, 0, sizeof(DevStats), stream));
  HIPCHK(hipMemsetAsync(counters.p, 0, CNT_WORDS * sizeof(uint32_t), stream));

  // per-spp film jitter shared by all pixels: sampler_t::preprocess (sampling.cpp:98-112) with
  // sample::stratified_2d (math/sampling.hpp:65-77), numbers from the counter sampler
  const uint32_t spp = opt.samples_per_pixel;
  std::vector<float2> jit(spp, make_float2(0.0f, 0.0f));
  {
    const uint32_t spd = (uint32_t)std::lroundf(std::sqrt((float)spp));
    const float step = 1.0f / (float)spd;
    float dy = 0.0f;
    for (uint32_t i = 0; i < spd; ++i, dy += step) {
      float dx = 0.0f;
      for (uint32_t j = 0; j < spd; ++j, dx += step) {
        const uint32_t cell = j * spd + i;
        const uint32_t key = path_key(frame.sampler_seed, FILM_JITTER_STREAM, cell);
        const float a = dx + draw_f32(key, 0) * step;
        const float b = dy + draw_f32(key, 1) * step;
        if (cell < spp) jit[cell] = make_float2(a, b);
      }
    }
  }
  if (jitter_seed != frame.sampler_seed || jitter_spp != spp || !jitter.p) {  // a frame loop presents the same seed again and again
    if ((rc = jitter.upload(jit))) return rc;
    jitter_seed = frame.sampler_seed; jitter_spp = spp;
  }

  // drain the shared tile queue in batches (cpu.cpp:233-234 pulls one tile at a time).  A batch holds as many pixels as can carry ALL
  // their samples in one pass (P x spp <= the path budget, at most 8 M pixels): path ids are pixel-major, so a pass with every sample of a
  // pixel keeps the 64 lanes of a wave — and the 256 rays of a camera-ray packet — on ONE pixel.  (Round 3 took 8 M pixels whatever the spp:
  // the 3840x2160, 256-spp frame of BASELINE config 4 went through as 8 passes of 32 samples; as 8 batches of 256 samples it is 6.8 %
  // faster — camera rays 67.9 -> 36.0 ms, k_trace -4 %: profiles/r04_v_batch_probe.log.)
  const size_t path_bytes = 176u + (frame.normals_channel ? 16u : 0u);
  // (divided by the samples a pass will really carry: with an explicit samples_in_flight only P x S paths are ever in flight)
  const uint32_t pass_samples = std::max(1u, std::min(opt.samples_per_pixel, opt.samples_in_flight ? opt.samples_in_flight : opt.samples_per_pixel));
  const uint64_t pixel_cap = std::min<uint64_t>(8u << 20, std::max<uint64_t>(path_budget(path_bytes) / pass_samples, 64u << 10));
  // A batch never EXCEEDS the cap (the tile that would is kept for the next batch): its paths then fit the queues the first full batch
  // allocated, whatever mix of whole and edge tiles it holds.  (Until round 5 a batch overshot by up to one tile; a 4 096-spp batch that grew
  // from 131 072 to 131 584 pixels re-allocated 86 GB of queues — 2.6 s of hipFree — in the middle of a frame: profiles/r05_f_c5_batch_probe.log.)
  phx_tile held{}; bool have_held = false;
  for (;;) {
    std::vector<phx_tile> tiles;
    uint64_t px = 0;
    while ((opt.tiles_per_batch == 0 || tiles.size() < opt.tiles_per_batch) && px < pixel_cap) {
      phx_tile t;
      if (have_held) { t = held; have_held = false; }
      else if (!frame.next_tile(frame.tiles_user, &t)) break;
      if (t.w == 0 || t.h == 0 || (uint64_t)t.x + t.w > scene.width || (uint64_t)t.y + t.h > scene.height) return fail(PHX_ERR_ARG, "tile outside the film");
      if (!tiles.empty() && px + (uint64_t)t.w * t.h > pixel_cap) { held = t; have_held = true; break; }
      tiles.push_back(t); px += (uint64_t)t.w * t.h;
    }
    if (tiles.empty()) break;
    {
      // the batch's tiles in Morton order of their film position: path ids are pixel-major, so the batch's queue — and with it each
      // XCD's eighth of it (k_trace's segments) — covers a compact block of the film instead of a row of tiles 3840 pixels wide
      static const bool morton = [] { const char* v = std::getenv("PHX_TILE_MORTON"); return v ? std::atoi(v) != 0 : true; }();
      if (morton) {
        auto spread = [](uint32_t v) { v &= 0xffffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v; };
        auto key = [&](const phx_tile& t) { return spread(t.x >> 5) | (spread(t.y >> 5) << 1); };
        std::stable_sort(tiles.begin(), tiles.end(), [&](const phx_tile& a, const phx_tile& b) { return key(a) < key(b); });
      }
    }
    if ((rc = render_batch(tiles, jit))) return rc;
    stats.tiles += tiles.size();
  }
  t_enq = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(stream));
  t_sync = std::chrono::steady_clock::now();
  DevStats ds;
  HIPCHK(hipMemcpy(&ds, dstats.p, sizeof(ds), hipMemcpyDeviceToHost));
  stats.camera_samples = ds.camera_samples; stats.rays_closest = ds.rays_closest; stats.rays_shadow = ds.rays_shadow;
  stats.rays_masked = ds.rays_closest - ds.rays_shadow;  // every shaded slot is a shadow ray or a masked slot (spt.hpp:138-141)
  for (int k = 0; k < 2; ++k) { stats.node_visits_lds[k] = ds.node_visits_lds[k]; stats.node_visits_mem[k] = ds.node_visits_mem[k]; stats.tri_tests[k] = ds.tri_tests[k]; }
  stats.instrumented = launch_counts_traversal_work() ? 1 : 0;
  stats.wave_iters = ds.wave_iters; stats.node_block_execs = ds.node_block_execs; stats.tri_block_execs = ds.tri_block_execs; stats.refills = ds.refills;
  stats.idle_lane_iters = ds.idle_lane_iters; stats.tri_pending_lane_iters = ds.tri_pending_lane_iters;
  for (int k = 0; k < 8; ++k) { stats.stack_pushes[k] = ds.stack_pushes[k]; stats.tri_pairs_hist[k] = ds.tri_pairs_hist[k]; }
  stats.tri_pairs_pending = ds.tri_pairs_pending;
  stats.primary_packets = ds.primary_packets; stats.primary_fallbacks = ds.primary_fallbacks; stats.primary_node_tests = ds.primary_node_tests;
  stats.primary_tri_tests = ds.primary_tri_tests; stats.primary_tri_lanes_hit = ds.primary_tri_lanes_hit;
  if (ds.watchdog) return fail(PHX_ERR_DEVICE, "k_trace: " + std::to_string(ds.watchdog) + " wave(s) hit the iteration watchdog: the frame is incomplete");
  if (ds.ring_watchdog) return fail(PHX_ERR_DEVICE, "k_shade_g: " + std::to_string(ds.ring_watchdog) + " wave(s) timed out waiting for a block of the append ring: the frame is incomplete");
  stats.trace_ms = stats.closest_ms + stats.shadow_ms + stats.primary_ms;
  stats.frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  static const bool host_timing = std::getenv("PHX_HOST_TIMING") != nullptr;
  if (host_timing) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "host timing: start->thread %.3f  thread->sync-begin (enqueue etc.) %.3f  sync wait %.3f  stats %.3f | kernels on the GPU %.3f ms | start->done %.3f\n",
                 ms(t_start, t0), ms(t0, t_enq), ms(t_enq, t_sync), ms(t_sync, std::chrono::steady_clock::now()), stats.trace_ms + stats.shade_ms, ms(t_start, std::chrono::steady_clock::now()));
  }
  return PHX_OK;
}

int phx_device::render_batch(const std::vector<phx_tile>& tiles, const std::vector<float2>& jit) {
  (void)jit;
  int rc;
  static const bool host_timing = std::getenv("PHX_HOST_TIMING") != nullptr;  // probe: where a batch's host time goes
  const auto tb0 = std::chrono::steady_clock::now();
  auto tb_alloc = tb0, tb_enq = tb0;
  uint32_t P = 0;
  for (auto& t : tiles) P += t.w * t.h;
  const uint32_t spp = opt.samples_per_pixel;
  const uint32_t xs = frame.primary_components + (frame.normals_channel ? 3u : 0u);
  // bytes of queues + state per path in flight: ray queues 2 x 32, hit 16, shadow queue 48, path state 32 (+ 16 with normals)
  const size_t path_bytes = 176u + (frame.normals_channel ? 16u : 0u);
  // paths in flight: the device's budget (path_budget: up to 512 M paths).  Deep bounces keep only a few percent of the paths alive, so
  // many paths per pass are what keeps late launches full; a batch that cannot carry all its samples at once splits the spp range
  // into equal passes.
  auto pick_samples = [&]() -> uint32_t {
    if (opt.samples_in_flight) return opt.samples_in_flight;
    const uint64_t budget = std::max<uint64_t>(path_budget(path_bytes), P);
    // (no slack: run_frame never hands over a batch above its pixel cap = budget / samples of a pass, so P x spp <= budget holds for every
    // batch it sized; a caller's own tile list — or the cap's floor of 64 k pixels — may exceed it, and then the spp range is split)
    const uint32_t smax = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(budget / P, 0x7ffffff0ull / P));
    const uint32_t npasses = (spp + smax - 1) / smax;
    return (spp + npasses - 1) / npasses;
  };
  uint32_t S = std::min(pick_samples(), spp);
  // The budget is a cached reading of the device's free memory (hipMemGetInfo costs ~0.1 ms).  It is only trusted while the queues this
  // object already holds are large enough: a batch that has to GROW them asks the device again — another device object, torch films or
  // RCCL buffers may have taken memory since the reading was made.
  if ((size_t)P * S > hit.n && !opt.samples_in_flight) { budget_bytes = 0; S = std::min(pick_samples(), spp); }
  if ((size_t)P * S >= 0x7fffffffull) return fail(PHX_ERR_ARG, "too many paths in flight");

  // pixel table of the batch; a frame loop presents the same tiles again and again, so the upload is skipped when nothing changed
  if (tiles.size() != pix_xy_tiles.size() || std::memcmp(tiles.data(), pix_xy_tiles.data(), tiles.size() * sizeof(phx_tile)) != 0) {
    std::vector<uint32_t> xy(P);
    uint32_t k = 0;
    for (auto& t : tiles)
      for (uint32_t y = 0; y < t.h; ++y)
        for (uint32_t x = 0; x < t.w; ++x) xy[k++] = (t.x + x) | ((t.y + y) << 16);
    pix_xy_tiles.clear();
    if ((rc = pix_xy.upload(xy))) return rc;
    pix_xy_tiles = tiles;
  }
  if ((rc = acc.alloc((size_t)P * xs))) return rc;
  // queues + state; when the device cannot hold S samples per pixel in flight, back off to half as many (more, shorter passes)
  size_t npaths = 0;
  for (;;) {
    npaths = (size_t)P * S;
    rc = PHX_OK;
    for (int q = 0; q < 2 && !rc; ++q) if ((rc = ro[q].alloc(npaths)) || (rc = rd[q].alloc(npaths)) || (rc = qs[q].alloc(npaths))) break;
    if (!rc) (void)((rc = hit.alloc(npaths)) || (rc = so.alloc(npaths)) || (rc = sd.alloc(npaths)) || (rc = sc.alloc(npaths)) ||
                    (rc = pr.alloc(npaths)) || (frame.normals_channel && (rc = pn.alloc(npaths))));
    if (!rc) break;
    if (rc != PHX_ERR_OOM || S == 1) return rc;
    (void)hipGetLastError();
    for (int q = 0; q < 2; ++q) { ro[q].release(); rd[q].release(); qs[q].release(); }
    hit.release(); so.release(); sd.release(); sc.release(); pr.release(); pn.release();
    budget_bytes = 0;  // the reading was stale: the next path_budget() — of this batch, of the next one, of the next frame — asks the device again
    S = std::min((S + 1) / 2, std::min(pick_samples(), spp));
  }
  paths_in_flight = npaths;
  tb_alloc = std::chrono::steady_clock::now();

  PassBuffers B{};
  for (int q = 0; q < 2; ++q) { B.ro[q] = ro[q].p; B.rd[q] = rd[q].p; B.qs[q] = qs[q].p; }
  B.hit = hit.p; B.so = so.p; B.sd = sd.p; B.sc = sc.p; B.pr = pr.p;
  B.pn = frame.normals_channel ? pn.p : nullptr;
  B.counters = counters.p; B.stats = dstats.p; B.pix_xy = pix_xy.p; B.jitter = jitter.p; B.acc = acc.p;
  B.num_pixels = P; B.xstride = xs; B.normals_offset = frame.normals_channel ? frame.primary_components : 0;
  B.seed = frame.sampler_seed;

  BatchLaunches* G = &direct;
  direct.events_used = 0; direct.timed.clear();
  if ((rc = enqueue_batch(direct, B, P, S, xs))) return rc;
  tb_enq = std::chrono::steady_clock::now();
  if (frame.add_tile || frame.host_film) {
    const size_t nfl = (size_t)P * xs;
    if (nfl > h_acc_n) {
      if (h_acc) (void)hipHostFree(h_acc);
      HIPCHK(hipHostMalloc((void**)&h_acc, nfl * sizeof(float), hipHostMallocDefault));
      h_acc_n = nfl;
    }
    HIPCHK(hipMemcpyAsync(h_acc, acc.p, nfl * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<size_t> offs(tiles.size() + 1, 0);
    for (size_t i = 0; i < tiles.size(); ++i) offs[i + 1] = offs[i] + (size_t)tiles[i].w * tiles[i].h * xs;
    if (frame.add_tile)
      for (size_t i = 0; i < tiles.size(); ++i) {
        const phx_tile& t = tiles[i];
        frame.add_tile(frame.film_user, (int32_t)t.x, (int32_t)t.y, (int32_t)t.w, (int32_t)t.h, h_acc + offs[i], xs, xs * t.w);
      }
    if (frame.host_film) {  // what an add_tile that copies into a frame buffer does, spread over a few host threads
      auto copy_range = [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
          const phx_tile& t = tiles[i];
          for (uint32_t y = 0; y < t.h; ++y)
            std::memcpy(frame.host_film + ((size_t)(t.y + y) * scene.width + t.x) * xs, h_acc + offs[i] + (size_t)y * t.w * xs, (size_t)t.w * xs * sizeof(float));
        }
      };
      const size_t nthreads = std::min<size_t>({8, std::max(1u, std::thread::hardware_concurrency() / 2), (nfl * sizeof(float)) >> 20});
      if (nthreads <= 1) copy_range(0, tiles.size());
      else {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < nthreads; ++k) pool.emplace_back(copy_range, tiles.size() * k / nthreads, tiles.size() * (k + 1) / nthreads);
        for (auto& th : pool) th.join();
      }
    }
  } else {
    HIPCHK(hipStreamSynchronize(stream));
  }
  if (host_timing) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "batch of %u px x %u samples: tables + allocation %.3f ms, enqueue %.3f, wait + sink %.3f\n", P, S, ms(tb0, tb_alloc), ms(tb_alloc, tb_enq),
                 ms(tb_enq, std::chrono::steady_clock::now()));
  }
  // the batch is complete: its launches' HIP-event times (the next batch records the same events again)
  for (auto& te : G->timed) {
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, G->events[te.begin], G->events[te.end]));
    if (te.kind == 0) { stats.closest_ms += ms; stats.trace_launches++; }  // k_trace: closest + shadow rays in one launch
    else if (te.kind == 4) { stats.primary_ms += ms; stats.primary_launches++; }  // k_trace_primary: the camera rays of a pass
    else stats.shade_ms += ms;
    if (te.kind == 3) { stats.shade_kernel_ms += ms; stats.shade_launches++; }
  }
  return PHX_OK;
}

// The launches of one batch, in stream order.
int phx_device::enqueue_batch(BatchLaunches& g, const PassBuffers& B0, uint32_t P, uint32_t S, uint32_t xs) {
  int rc;
  PassBuffers B = B0;
  const uint32_t spp = opt.samples_per_pixel;
  HIPCHK(hipMemsetAsync(acc.p, 0, (size_t)P * xs * sizeof(float), stream));
  const float inv = 1.0f / (float)(spp * opt.paths_per_sample);  // cpu.cpp:191
  // One event BETWEEN two launches serves as the end of the first and the start of the second (an event record is a packet of its own in
  // the queue: 42 of them per pass made the gaps between the 21 launches longer than the launches need).  A launch's time then includes
  // the few microseconds since the previous kernel ended.
  long last_end = -1;  // index of the event recorded behind the previous timed launch of this batch
  auto timed_launch = [&](int kind, auto&& fn) -> int {
    hipEvent_t e; int r;
    if (!kernel_timing) { fn(); return PHX_OK; }
    if (last_end < 0) { if ((r = next_event(g, &e))) return r; HIPCHK(hipEventRecord(e, stream)); last_end = (long)g.events_used - 1; }
    const size_t begin = (size_t)last_end;
    fn();
    if ((r = next_event(g, &e))) return r;
    HIPCHK(hipEventRecord(e, stream));
    last_end = (long)g.events_used - 1;
    g.timed.push_back({begin, (size_t)last_end, kind});
    return PHX_OK;
  };
  // Shade grids sized by the queue, not by its capacity.  k_shade runs one workgroup per 1024 entries and the host does not know the queue
  // lengths (they stay on the device): a grid for the capacity is 230 k workgroups at EVERY step of the bench frame, nearly all of them
  // empty from the third step on — 0.25 ms per launch for nothing.  k_trace of step b publishes the length of the queue it traces in pinned
  // host memory; the queue of a later step is never longer, so the shade launch of step b + 1 is sized by it.  The host waits for that word
  // before it enqueues the launch — while the device still has k_trace(b), k_shade(b) and k_trace(b + 1) in its queue: it never runs dry
  // (measured: k_shade 8.6 -> 7.5 ms on the bench frame, profiles/r06_v_grid_by_queue_ab.log).
  // (PHX_SHADE_GRID_BY_QUEUE=0: grids for the capacity, everything enqueued at once, as before.)
  static const bool grid_by_queue = [] { const char* v = std::getenv("PHX_SHADE_GRID_BY_QUEUE"); return !(v && v[0] == '0'); }();
  const uint32_t depth = opt.path_depth, npasses = (spp + S - 1) / S;
  if (grid_by_queue) {
    const size_t need = (size_t)npasses * depth;
    if (need > h_qlen_n) {
      if (h_qlen) (void)hipHostFree(h_qlen);
      h_qlen = nullptr; h_qlen_n = 0;
      HIPCHK(hipHostMalloc((void**)&h_qlen, need * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
      h_qlen_n = need;
    }
    for (size_t k = 0; k < need; ++k) __atomic_store_n(&h_qlen[k], 0xffffffffu, __ATOMIC_RELAXED);  // the previous batch has been joined (render_batch)
  }
  auto queue_bound = [&](uint32_t pass, uint32_t step, uint32_t cap) -> uint32_t {  // length of the ray queue k_trace(step) of this pass traced, once it has started
    const volatile uint32_t* w = &h_qlen[(size_t)pass * depth + step];
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
      const uint32_t v = __atomic_load_n(const_cast<const uint32_t*>(w), __ATOMIC_ACQUIRE);
      if (v != 0xffffffffu) return std::min(v, cap);
      // never for ever: a device that has stopped (fault, watchdog) is found by the synchronisation behind the batch — size for the capacity and go on
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return cap;
    }
  };
  uint32_t pass = 0;
  for (uint32_t s0 = 0; s0 < spp; s0 += S, ++pass) {
    const uint32_t ns = std::min(S, spp - s0);
    const uint32_t cap = P * ns;
    B.num_samples = ns;
    B.qlen_out = nullptr;
    if ((rc = timed_launch(2, [&]() { launch_begin_pass(stream, B, ns); }))) return rc;
    int q = 0;
    for (uint32_t bounce = 0; bounce < opt.path_depth; ++bounce) {  // a path takes at most path_depth steps (spt.hpp:314)
      // step `bounce`: closest-hit rays of this step + the shadow rays k_shade produced in the previous step
      const int sq_read = (int)((bounce + 1) & 1), sq_write = (int)(bounce & 1);
      if (bounce == 0) {  // the camera rays: one packet walk per 64 x n of them
        if ((rc = timed_launch(4, [&]() { launch_trace_primary(stream, scene, B, cap, s0, q, sq_read); }))) return rc;
      } else {
        B.qlen_out = grid_by_queue ? h_qlen + (size_t)pass * depth + bounce : nullptr;
        if ((rc = timed_launch(0, [&]() { launch_trace(stream, scene, B, q, sq_read, 1, 1, cap); }))) return rc;
        B.qlen_out = nullptr;
      }
      // step 0: the capacity.  Step 1: the length k_trace(1) — the longest launch of a pass, enqueued just above — publishes as it starts.
      // Later steps: the length published one step earlier (the device then still has two launches queued while the host waits).
      const uint32_t shade_cap = (grid_by_queue && bounce >= 1) ? std::max(queue_bound(pass, bounce == 1 ? 1u : bounce - 1, cap), 1u) : cap;
      if ((rc = timed_launch(3, [&]() { launch_shade(stream, scene, B, q, sq_write, shade_cap, s0, bounce == 0); }))) return rc;
      q ^= 1;
    }
    if ((rc = timed_launch(0, [&]() { launch_trace(stream, scene, B, q, (int)((opt.path_depth - 1) & 1), 0, 1, cap); }))) return rc;
    if ((rc = timed_launch(2, [&]() { launch_film(stream, B, ns, inv); }))) return rc;
    HIPCHK(hipGetLastError());
  }
  // film_t<>::add_tile (film.hpp:12-15)
  if (frame.device_film) {
    launch_scatter_film(stream, B, frame.device_film, scene.width);
    HIPCHK(hipGetLastError());
  }
  return PHX_OK;
}
