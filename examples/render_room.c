/* A host written in plain C against include/phx_xpu.h — the same call sequence the reference's session_t::render makes
 * through xpu_t (plugins/blender/session.cpp:73-94: discover -> preprocess -> tiles_t::make -> start -> join), with no
 * Python, no C++ and no torch in the process.  Renders a small room (floor, back wall, emissive ceiling panel, two
 * tilted triangles) and writes the raw fp32 film (H x W x 4) to argv[1].
 *   make -C examples
 *   ./render_room film.f32 [host-bvh|device-bvh|auto-bvh] [N]
 * With N > 1 it is the reference's multi-device mechanism (src/core.cpp:103-115): N devices — one per GPU that phx_discover
 * reports, wrapping around when the box has fewer — are all started on ONE tile queue and ONE film, and joined in turn.
 * No collective: whoever renders a tile writes it into the shared film.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "phx_xpu.h"

#define W 128
#define H 96

static int fail(const char* what) {
  fprintf(stderr, "%s: %s\n", what, phx_last_error());
  return 1;
}

int main(int argc, char** argv) {
  static const float vertices[] = {
      /* floor y = -1 */ -2, -1, -1, 2, -1, -1, 2, -1, -5, -2, -1, -5,
      /* back wall z = -5 */ -2, -1, -5, 2, -1, -5, 2, 2, -5, -2, 2, -5,
      /* ceiling panel y = 1.5, facing down */ -1, 1.5f, -2, -1, 1.5f, -4, 1, 1.5f, -4, 1, 1.5f, -2,
      /* two tilted triangles */ -0.8f, -0.9f, -3.2f, 0.9f, -0.6f, -3.6f, 0.1f, 0.7f, -3.0f, -1.5f, -0.2f, -4.2f, -0.6f, 0.9f, -4.4f, -1.7f, 1.1f, -3.9f};
  static const uint32_t faces[] = {0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7, 8, 9, 10, 8, 10, 11, 12, 13, 14, 15, 16, 17};
  static const uint32_t grey_faces[] = {0, 1, 2, 3}, light_faces[] = {4, 5}, red_faces[] = {6, 7};

  phx_material materials[3];
  memset(materials, 0, sizeof(materials));
  materials[0].num_lobes = 1; /* diffuse_bsdf_node, Cs = 0.73 */
  materials[0].lobes[0].type = PHX_LOBE_DIFFUSE;
  materials[0].lobes[0].weight[0] = materials[0].lobes[0].weight[1] = materials[0].lobes[0].weight[2] = 0.73f;
  materials[1].is_emitter = 1; /* diffuse_emitter_node */
  materials[1].emission[0] = 17.0f; materials[1].emission[1] = 12.0f; materials[1].emission[2] = 4.0f;
  materials[2].num_lobes = 1;
  materials[2].lobes[0].type = PHX_LOBE_DIFFUSE;
  materials[2].lobes[0].weight[0] = 0.63f; materials[2].lobes[0].weight[1] = 0.065f; materials[2].lobes[0].weight[2] = 0.05f;

  phx_face_set sets[3] = {{0, 4, grey_faces}, {1, 2, light_faces}, {2, 2, red_faces}};
  phx_mesh mesh;
  memset(&mesh, 0, sizeof(mesh));
  mesh.vertices = vertices; mesh.num_vertices = 18;
  mesh.faces = faces; mesh.num_faces = 8;
  mesh.flags = PHX_MESH_UV_PER_VERTEX | PHX_MESH_NORMALS_PER_VERTEX;
  mesh.num_sets = 3; mesh.sets = sets;

  phx_scene scene;
  memset(&scene, 0, sizeof(scene));
  scene.num_meshes = 1; scene.meshes = &mesh;
  scene.num_materials = 3; scene.materials = materials;
  scene.environment_material = -1;
  for (int i = 0; i < 4; ++i) scene.camera.to_world[5 * i] = 1.0f; /* camera at the origin looking down -z */
  scene.camera.fov = 1.2f; scene.camera.focal_distance = 1.0f;
  scene.camera.film_width = W; scene.camera.film_height = H;

  phx_options opt;
  memset(&opt, 0, sizeof(opt));
  opt.samples_per_pixel = 8; opt.paths_per_sample = 1; opt.path_depth = 5; opt.device_ordinal = -1;
  if (argc > 2 && strcmp(argv[2], "device-bvh") == 0) opt.bvh_builder = PHX_BVH_DEVICE_LBVH;
  if (argc > 2 && strcmp(argv[2], "host-bvh") == 0) opt.bvh_builder = PHX_BVH_HOST_SAH;

  int n = 0;
  if (phx_discover(&opt, &n) != PHX_OK || n < 1) return fail("phx_discover");
  int num_devices = argc > 3 ? atoi(argv[3]) : 1;
  if (num_devices < 1 || num_devices > 64) num_devices = 1;
  phx_device* devs[64];
  for (int i = 0; i < num_devices; ++i) { /* xpu_t::discover: one device object per GPU (src/xpu.cpp:7-9) */
    opt.device_ordinal = num_devices > 1 ? i % n : -1;
    devs[i] = phx_dev_make(&opt);
    if (!devs[i]) return fail("phx_dev_make");
    if (phx_dev_preprocess(devs[i], &scene) != PHX_OK) return fail("phx_dev_preprocess"); /* every device holds its own copy + BVH */
  }

  float* film = (float*)calloc((size_t)W * H * 4, sizeof(float));
  phx_tiles* tiles = phx_tiles_make(W, H, 32, 0, 1); /* ONE job::tiles_t for all devices */
  phx_frame frame;
  memset(&frame, 0, sizeof(frame));
  frame.tiles_user = tiles; frame.next_tile = phx_tiles_next;
  frame.sampler_seed = 7; frame.primary_components = 4;
  frame.host_film = film;                             /* ONE film: tiles are disjoint, so no lock and no reduce */
  for (int i = 0; i < num_devices; ++i)
    if (phx_dev_start(devs[i], &frame) != PHX_OK) return fail("phx_dev_start");
  for (int i = 0; i < num_devices; ++i)
    if (phx_dev_join(devs[i]) != PHX_OK) return fail("phx_dev_join");

  phx_stats st, sum;
  memset(&sum, 0, sizeof(sum));
  for (int i = 0; i < num_devices; ++i) {
    phx_dev_get_stats(devs[i], &st);
    sum.tiles += st.tiles; sum.camera_samples += st.camera_samples; sum.rays_closest += st.rays_closest; sum.rays_shadow += st.rays_shadow;
    sum.bvh_nodes = st.bvh_nodes;
    if (num_devices > 1) printf("device %d (GPU %d): tiles %llu\n", i, i % n, (unsigned long long)st.tiles);
  }
  st = sum;
  printf("tiles %llu camera_samples %llu rays %llu+%llu bvh_nodes %llu\n", (unsigned long long)st.tiles, (unsigned long long)st.camera_samples,
         (unsigned long long)st.rays_closest, (unsigned long long)st.rays_shadow, (unsigned long long)st.bvh_nodes);
  if (argc > 1) {
    FILE* f = fopen(argv[1], "wb");
    if (!f || fwrite(film, sizeof(float), (size_t)W * H * 4, f) != (size_t)W * H * 4) { fprintf(stderr, "cannot write %s\n", argv[1]); return 1; }
    fclose(f);
  }
  phx_tiles_free(tiles);
  for (int i = 0; i < num_devices; ++i) phx_dev_destroy(devs[i]);
  free(film);
  return 0;
}
