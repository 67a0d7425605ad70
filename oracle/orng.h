// ORACLE — test infrastructure only.  Never linked into or called by the product path.
//
// orng.h: the two random-number services of the restatement.
//
//  (1) seq_rng_t — the reference's sampler_t::details_t (src/sampling.cpp:43-76): ONE
//      std::mt19937 (default seed 5489, never reseeded) behind
//      std::uniform_real_distribution<float>(0,1); sample() = one draw, sample2() = {draw, draw}
//      left to right.  Same libstdc++ on both boxes, so the stream is reproduced exactly
//      (first values 0.81472367, 0.135477006, 0.905791938 — SURVEY A-5).
//
//  (2) counter RNG — the sampler the HIP device uses (a GPU cannot replay a data-dependent
//      sequential stream, SURVEY §7 hard part 1): every random number is a pure function of
//      (seed, pixel, sample, dimension).  Integer-only, so CPU and GPU agree bit for bit.
#pragma once
#include <cstdint>
#include <random>

namespace orc {

struct seq_rng_t {
  std::mt19937 gen;
  std::uniform_real_distribution<float> dis;
  uint64_t draws = 0;
  seq_rng_t() : dis(0.0f, 1.0f) {}
  float sample() { ++draws; return dis(gen); }
};

// ---- counter RNG ---------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// one key per (pixel, sample) path; pixel = y*W+x of the film, sample = spp index
inline uint32_t path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t k = mix32((uint32_t)seed ^ 0x85ebca6bu);
  k = mix32(k + pixel);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k + sample * 0x9e3779b1u);
  return k;
}
inline uint32_t draw_u32(uint32_t key, uint32_t dim) { return mix32(key + (dim + 1u) * 0x9e3779b9u); }
// [0,1) with 24 random bits — exact in fp32
inline float draw_f32(uint32_t key, uint32_t dim) { return (float)(draw_u32(key, dim) >> 8) * (1.0f / 16777216.0f); }

// dimensions of one path step (bounce b uses 8*b + DIM_*)
enum { DIM_LIGHT_PICK = 0, DIM_LIGHT_U = 1, DIM_LIGHT_V = 2, DIM_RR = 3, DIM_BSDF_U = 4, DIM_BSDF_V = 5, DIMS_PER_STEP = 8,
       DIM_LENS_U = 6, DIM_LENS_V = 7 /* the two spare dimensions of step 0: the thin lens */ };
// pseudo pixel id used for the per-spp film jitter table
static const uint32_t FILM_JITTER_STREAM = 0xffffffffu;

}  // namespace orc
