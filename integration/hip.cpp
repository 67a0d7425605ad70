// integration/hip.cpp — see hip.hpp.  Flattens scene_t through its public accessors into phx_scene,
// forwards xpu_t's three virtuals to the C ABI, and adapts job::tiles_t / film_t<> to the two callbacks.
#include "hip.hpp"

#include "buffer.hpp"
#include "light.hpp"
#include "material.hpp"
#include "mesh.hpp"
#include "options.hpp"
#include "scene.hpp"
#include "state.hpp"

#include <stdexcept>
#include <vector>

namespace {
  phx_options to_options(const parsed_options_t& o, int ordinal = -1) {
    phx_options out = {};
    out.samples_per_pixel = o.samples_per_pixel;
    out.paths_per_sample  = o.paths_per_sample;
    out.path_depth        = o.path_depth;
    out.single_threaded   = o.single_threaded;
    out.render_normals    = o.render_normals;
    out.verbose           = o.verbose;
    out.host_only         = o.host_only;
    out.device_ordinal    = ordinal;
    return out;
  }

  // job::tiles_t::next  ->  phx_next_tile_fn
  int next_tile(void* user, phx_tile* out) {
    job::tiles_t::tile_t t;
    if (!static_cast<job::tiles_t*>(user)->next(t)) return 0;
    out->x = t.x; out->y = t.y; out->w = t.w; out->h = t.h;
    return 1;
  }

  // phx_add_tile_fn  ->  film_t<>::add_tile: wrap the device's tile buffer in a render_buffer_t view
  typedef hip_t::sink_t sink_t;
  void add_tile(void* user, int32_t x, int32_t y, int32_t w, int32_t h, const float* data, uint32_t xstride, uint32_t ystride) {
    sink_t* sink = static_cast<sink_t*>(user);
    render_buffer_t view(sink->format);
    view.buffer = const_cast<float*>(data);
    view.width = w; view.height = h; view.ystride = ystride;  // xstride was summed by the constructor
    sink->film->add_tile(Imath::V2i(x, y), Imath::V2i(w, h), view);
  }
}

hip_t::hip_t(const parsed_options_t& options, int ordinal) : device(nullptr), ordinal(ordinal) {
  const phx_options o = to_options(options, ordinal);
  device = phx_dev_make(&o);
  if (!device) throw std::runtime_error(phx_last_error());
}

hip_t::~hip_t() { phx_dev_destroy(device); }

hip_t* hip_t::make(const parsed_options_t& options, int ordinal) { return new hip_t(options, ordinal); }

int hip_t::count(const parsed_options_t& options) {
  const phx_options o = to_options(options);
  int n = 0;
  return phx_discover(&o, &n) == PHX_OK ? n : 0;
}

void hip_t::preprocess(const scene_t& scene) {
  // The closure recipe of a material is what material_t::evaluate would flatten for constant inputs
  // (material.cpp:218-305).  `bake_closures` is NOT defined anywhere in this repository: it is the one piece the
  // maintainer supplies inside the reference tree, where OSL is available — run each material's shader group once on a
  // neutral ShaderGlobals and record the closure tree (ids = bsdf_t::type_t, weights, parameter structs), INTEGRATION.md §3.
  // Until it exists this binding is INCOMPLETE: it links only together with that function.
  extern void bake_closures(const material_t* m, phx_material* out);

  std::vector<phx_material> materials(scene.num_materials());
  for (uint32_t i = 0; i < scene.num_materials(); ++i) {
    materials[i] = {};
    bake_closures(scene.material(i), &materials[i]);
    materials[i].is_emitter = scene.material(i)->is_emitter();
  }
  std::vector<phx_mesh> meshes(scene.num_meshes());
  std::vector<std::vector<phx_face_set>> sets(scene.num_meshes());
  std::vector<std::vector<uint8_t>> smooth(scene.num_meshes());
  for (uint32_t i = 0; i < scene.num_meshes(); ++i) {
    const mesh_t* m = scene.mesh(i);
    phx_mesh& out = meshes[i];
    out = {};
    out.vertices  = reinterpret_cast<const float*>(m->vertices);  out.num_vertices = m->num_vertices();   // accessor to add: details->vertices.size()
    out.normals   = reinterpret_cast<const float*>(m->normals);   out.num_normals  = m->num_normals();
    out.faces     = m->faces;                                      out.num_faces    = m->num_faces;
    smooth[i].resize(m->num_faces);
    for (uint32_t f = 0; f < m->num_faces; ++f) smooth[i][f] = m->is_smooth(f);                           // accessor to add: details->smooth[f]
    out.smooth    = smooth[i].data();
    out.flags     = m->flags;
    for (uint32_t s = 0; s < m->num_sets(); ++s)                                                          // accessor to add: details->sets.size()
      sets[i].push_back(phx_face_set{ m->sets[s].material, m->sets[s].num_faces, m->sets[s].faces });
    out.num_sets  = sets[i].size();
    out.sets      = sets[i].data();
  }
  phx_scene flat = {};
  flat.num_meshes = meshes.size();       flat.meshes = meshes.data();
  flat.num_materials = materials.size(); flat.materials = materials.data();
  flat.environment_material = scene.environment() ? (int32_t) scene.environment()->matid() : -1;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) flat.camera.to_world[4*r+c] = scene.camera.to_world.x[r][c];
  flat.camera.fov = scene.camera.fov;
  flat.camera.focal_distance = scene.camera.focal_distance;
  flat.camera.aperture_radius = scene.camera.aperture_radius;
  flat.camera.film_width = scene.camera.film.width;
  flat.camera.film_height = scene.camera.film.height;
  if (phx_dev_preprocess(device, &flat) != PHX_OK) throw std::runtime_error(phx_last_error());
}

void hip_t::start(const scene_t&, frame_state_t& state) {
  sink.film = state.film; sink.format = state.tiles->format;
  phx_frame f = {};
  f.tiles_user = state.tiles; f.next_tile = next_tile;
  f.film_user = &sink;        f.add_tile = add_tile;
  f.sampler_seed = 5489;      // std::mt19937's default seed, the only seed the reference ever uses
  f.primary_components = state.tiles->format.channels[0].components;
  f.normals_channel = state.tiles->format.channels.size() > 1;
  if (phx_dev_start(device, &f) != PHX_OK) throw std::runtime_error(phx_last_error());
}

void hip_t::join() {
  if (phx_dev_join(device) != PHX_OK) throw std::runtime_error(phx_last_error());
}
