// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// oscene.h: scene_t / mesh_t / triangle_t / light_t of the reference, rebuilt from the flat
// phx_scene handed over the C ABI.  Follows src/scene.cpp:58-62 (triangle order = mesh order x
// face-set order), src/mesh.cpp:108-128 (emitter face sets become area lights, in mesh order),
// src/mesh.cpp:154-258 (shading_parameters), src/mesh.cpp:269-337 (triangle_t),
// src/light.cpp:30-71 (area_light_t).
#pragma once
#include "../include/phx_xpu.h"
#include "ovec.h"

#include <cmath>
#include <limits>
#include <vector>

namespace orc {

struct mesh_t {
  uint32_t id;
  std::vector<V3> vertices, normals;
  std::vector<uint32_t> faces;  // 3 per face
  std::vector<uint8_t> smooth;
  uint32_t flags;
  struct set_t { uint32_t material; std::vector<uint32_t> faces; };
  std::vector<set_t> sets;
  bool normals_per_vertex() const { return (flags & PHX_MESH_NORMALS_PER_VERTEX) != 0; }

  // mesh_t::shading_parameters, src/mesh.cpp:169-258 (normal part; tangents/uvs feed only OSL
  // texture nodes, which the closure-recipe materials do not have)
  V3 shading_normal(uint32_t face, float u, float v) const {
    const float w = 1 - u - v;
    const uint32_t a = faces[face], b = faces[face + 1], c = faces[face + 2];
    uint32_t na = a, nb = b, nc = c;
    V3 n;
    if (smooth[face / 3]) {
      if (!normals_per_vertex()) { na = face; nb = face + 1; nc = face + 2; }
      const V3& n0 = normals[na]; const V3& n1 = normals[nb]; const V3& n2 = normals[nc];
      n = (w * n0 + u * n1 + v * n2).normalize();
    } else {
      const V3 v0 = vertices[a], v1 = vertices[b], v2 = vertices[c];
      n = (v1 - v0).cross(v2 - v0).normalize();
    }
    return n;  // not flipped towards the viewer (mesh.cpp:209-215 is commented out)
  }
};

// triangle_t, src/triangle.hpp:12 + src/mesh.cpp:269-337
struct tri_ref_t {
  const mesh_t* mesh;
  uint32_t set;
  uint32_t face;  // 3 * face index
  uint32_t meshid() const { return mesh->id; }
  uint32_t matid() const { return mesh->sets[set].material; }
  const V3& a() const { return mesh->vertices[mesh->faces[face]]; }
  const V3& b() const { return mesh->vertices[mesh->faces[face + 1]]; }
  const V3& c() const { return mesh->vertices[mesh->faces[face + 2]]; }
  Box3 bounds() const { Box3 bb; bb.extendBy(a()); bb.extendBy(b()); bb.extendBy(c()); return bb; }
  float area() const {
    const V3 ab = b() - a(), ac = c() - a();
    return 0.5f * ab.cross(ac).length();
  }
  // triangle_t::barycentric_to_point, mesh.cpp:314-316
  V3 barycentric_to_point(const V2& uv) const { return uv.x * a() + uv.y * b() + (1 - uv.x - uv.y) * c(); }
  // triangle_t::sample, mesh.cpp:318-324
  static V2 sample(const V2& uv) {
    const float x = std::sqrt(uv.x);
    return V2(1 - x, uv.y * x);
  }
};

struct light_sample_t { V3 p; V2 uv; uint32_t mesh; uint32_t face; float pdf; float area; };

// area_light_t, src/light.cpp:10-71
struct area_light_t {
  const mesh_t* mesh; uint32_t set; uint32_t matid;
  std::vector<tri_ref_t> triangles;
  float area = 0.0f;
  void preprocess() { for (auto& t : triangles) area += t.area(); }  // cdf computed but unused (light.cpp:30-45)
  void sample(const V2& uv, light_sample_t& out) const {
    const size_t num = triangles.size();
    const size_t i = std::min((size_t)std::floor(uv.x * num), num - 1);
    const float one_minus_epsilon = 1.0f - std::numeric_limits<float>::epsilon();
    const float remapped = std::min(uv.x * num - i, one_minus_epsilon);
    const tri_ref_t& t = triangles[i];
    const V2 bary = tri_ref_t::sample(V2(remapped, uv.y));
    out.p = t.barycentric_to_point(bary);
    out.uv = bary;
    out.pdf = 1.0f / area;
    out.mesh = t.meshid() | (t.matid() << 16);
    out.face = t.face;
    out.area = area;
  }
};

struct scene_t {
  std::vector<mesh_t> meshes;
  std::vector<phx_material> materials;
  std::vector<area_light_t> lights;
  int32_t env_material = -1;
  phx_camera camera;

  bool load(const phx_scene* s) {
    if (!s || !s->meshes || !s->materials) return false;
    camera = s->camera;
    env_material = s->environment_material;
    materials.assign(s->materials, s->materials + s->num_materials);
    meshes.resize(s->num_meshes);
    for (uint32_t i = 0; i < s->num_meshes; ++i) {
      const phx_mesh& m = s->meshes[i];
      mesh_t& o = meshes[i];
      o.id = i;  // scene_t::add(mesh_t*), scene.cpp:79-82
      o.flags = m.flags;
      o.vertices.resize(m.num_vertices);
      for (uint32_t k = 0; k < m.num_vertices; ++k) o.vertices[k] = V3(m.vertices[3 * k], m.vertices[3 * k + 1], m.vertices[3 * k + 2]);
      o.normals.resize(m.num_normals);
      for (uint32_t k = 0; k < m.num_normals; ++k) o.normals[k] = V3(m.normals[3 * k], m.normals[3 * k + 1], m.normals[3 * k + 2]);
      o.faces.assign(m.faces, m.faces + 3 * (size_t)m.num_faces);
      o.smooth.assign(m.smooth, m.smooth + m.num_faces);
      o.sets.resize(m.num_sets);
      for (uint32_t k = 0; k < m.num_sets; ++k) {
        if (m.sets[k].material >= s->num_materials) return false;
        o.sets[k].material = m.sets[k].material;
        o.sets[k].faces.assign(m.sets[k].faces, m.sets[k].faces + m.sets[k].num_faces);
      }
    }
    // scene_t::preprocess, scene.cpp:48-56: meshes first (emitter sets -> lights), then lights
    for (auto& m : meshes)
      for (uint32_t k = 0; k < m.sets.size(); ++k)
        if (materials[m.sets[k].material].is_emitter) {
          area_light_t l; l.mesh = &m; l.set = k; l.matid = m.sets[k].material;
          for (uint32_t f : m.sets[k].faces) l.triangles.push_back(tri_ref_t{&m, k, f * 3});
          lights.push_back(l);
        }
    for (auto& l : lights) l.preprocess();
    return true;
  }

  // scene_t::triangles, scene.cpp:58-62 -> mesh_t::triangles, mesh.cpp:118-128
  void triangles(std::vector<tri_ref_t>& out) const {
    for (auto& m : meshes)
      for (uint32_t k = 0; k < m.sets.size(); ++k)
        for (uint32_t f : m.sets[k].faces) out.push_back(tri_ref_t{&m, k, f * 3});
  }
};

}  // namespace orc
