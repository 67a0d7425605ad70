// bvh_build.h — host-side construction of the device BVH (bvh8.h).  Takes the place of the
// reference's recursive 8-ary binned-SAH builder (src/accel/bvh/binned_sah_builder.hpp:216-281,
// adapter src/accel/bvh.cpp:22-79): binned SAH binary build (multi-threaded over subtrees),
// greedy surface-area collapse to 8-wide nodes, octant-order slot assignment, outward quantisation.
#pragma once
#include "bvh8.h"

#include <vector>

namespace phx {

struct Bvh8 {
  std::vector<Node8> nodes;
  std::vector<TriRec> tris;
  uint32_t depth = 0;  // levels of Node8 (root = 1): bounds the traversal stack
};

// tri_abc: 9 floats per primitive (a, b, c), in scene_t::triangles() order.
void build_bvh8(const float* tri_abc, uint32_t num_prims, Bvh8& out, int num_threads);

}  // namespace phx
