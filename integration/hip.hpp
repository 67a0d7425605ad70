// integration/hip.hpp — drop this file and hip.cpp into the reference tree as src/xpu/hip.{hpp,cpp}
// (next to src/xpu/cpu.hpp and the empty src/xpu/cuda.hpp stub it supersedes) and link libphx_hip.so.
// It compiles only inside the reference tree (it includes the reference's own headers); it is the
// complete reference-side binding of the C ABI in include/phx_xpu.h.
#pragma once

#include "xpu.hpp"

#include <phx_xpu.h>

struct frame_state_t;
struct parsed_options_t;
struct scene_t;

/* MI355X (gfx950) device behind the xpu_t interface */
struct hip_t : public xpu_t {
  phx_device* device;
  frame_state_t* frame;  // valid between start() and join()

  hip_t(const parsed_options_t& options);
  ~hip_t();

  void preprocess(const scene_t& scene);
  void start(const scene_t& scene, frame_state_t& state);
  void join();

  static hip_t* make(const parsed_options_t& options);
};
