// bvh_build.h — host-side construction of the device BVH (bvh8.h).  Takes the place of the
// reference's recursive 8-ary binned-SAH builder (src/accel/bvh/binned_sah_builder.hpp:216-281,
// adapter src/accel/bvh.cpp:22-79): binned SAH binary build down to single triangles (multi-threaded over
// subtrees), greedy surface-area collapse to 8-wide nodes, octant-order slot assignment, outward quantisation,
// breadth-first layout of the 64-byte pool (a node's children — nodelets and triangle records — are contiguous).
#pragma once
#include "bvh8.h"

#include <vector>

namespace phx {

struct Bvh8 {
  std::vector<PoolElem> pool;  // element 0 is the root nodelet; breadth first, so a prefix of the pool is the top of the tree
  SceneGrid grid{};            // the grid the nodelets' origins are stored on
  uint32_t num_nodes = 0, num_tris = 0;
  uint32_t depth = 0;          // levels of nodelets (root = 1): bounds the traversal stack
  std::vector<uint32_t> elem_of_prim;  // pool index of every primitive's triangle record
  float cost = 0.0f;           // modelled traversal cost of the collapse (sub(root) of the optimal-collapse programme; 0 if greedy)
};

// tri_abc: 9 floats per primitive (a, b, c), in scene_t::triangles() order.
// prim_material: optional per-primitive material word copied into the triangle records.
void build_bvh8(const float* tri_abc, uint32_t num_prims, Bvh8& out, int num_threads, const uint32_t* prim_material = nullptr);

// SAH cost of a finished pool on its quantised child boxes (expected box tests x c_node + triangle tests x c_tri per random ray)
double bvh8_sah_cost(const Bvh8& b, float c_node, float c_tri);

}  // namespace phx
