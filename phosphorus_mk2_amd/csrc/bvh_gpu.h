// bvh_gpu.h — device-side LBVH build of the BVH8 pool of bvh8.h (see bvh_gpu.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "bvh8.h"

namespace phx {

struct GpuBvh {
  PoolElem* pool;      // hipMalloc'd (2 x triangles elements reserved), owned by the caller after a successful build
  uint32_t num_elems;  // elements in use: nodelets + triangle records (material word already filled)
  uint32_t num_nodes, num_tris, depth;
  SceneGrid grid;      // the grid the nodelets' origins are stored on
  float cost;          // modelled traversal cost of the collapse (sub(root) of k_collapse_dp; 0 if greedy): comparable with Bvh8::cost
};

// d_abc: 9 floats per primitive (a, b, c) in scene_t::triangles() order, device memory.
// d_prim_material: per-primitive material word (material | smooth << 31), device memory.
// Returns 0 on success; on failure writes a message to err and leaves *out empty: BVH_GPU_RECOVERABLE when the cause is one a host
// build can step around (the builder's scratch or pool did not fit the device's free memory; the tree is deeper than the traversal's
// tables), 1 for everything else — a HIP error from a launch or a sync, or a builder that lost triangles: bugs, never to be hidden.
enum { BVH_GPU_RECOVERABLE = 2 };
// d_elem_of_prim (optional, n words of device memory): receives the pool index of every primitive's triangle record.
int build_bvh8_gpu(hipStream_t stream, const float* d_abc, const uint32_t* d_prim_material, uint32_t n, GpuBvh* out, char* err, size_t errlen, uint32_t* d_elem_of_prim = nullptr);

}  // namespace phx
