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

#include "buffer.hpp"
#include "film.hpp"

/* One MI355X (gfx950) behind the xpu_t interface.  xpu_t::discover makes one hip_t per GPU (ordinal 0 .. n-1); like every
 * other xpu_t they all drain the frame's one job::tiles_t and add their tiles to the frame's one film (src/core.cpp:103-115). */
struct hip_t : public xpu_t {
  phx_device* device;
  int ordinal;
  struct sink_t { film_t<>* film; render_buffer_t::descriptor_t format; } sink;  // what the add_tile callback needs, alive start..join

  hip_t(const parsed_options_t& options, int ordinal);
  ~hip_t();

  void preprocess(const scene_t& scene);
  void start(const scene_t& scene, frame_state_t& state);
  void join();

  static hip_t* make(const parsed_options_t& options, int ordinal);
  static int count(const parsed_options_t& options);  // usable gfx950 devices (0 with --no-gpu)
};
