// Test helper (tests only): runs the device's BVH8 builder and the shared host/device traversal
// template on the CPU so that topology-independent closest-hit parity can be checked without a GPU.
#include "../../phosphorus_mk2_amd/csrc/bvh_build.h"
#include <cstdio>
using namespace phx;
struct HostStack {
  uint32_t b[64], h[64]; int sp = 0; int max_sp = 0;
  void push(uint32_t x, uint32_t y) { b[sp] = x; h[sp] = y; ++sp; if (sp > max_sp) max_sp = sp; }
  void pop(uint32_t& x, uint32_t& y) { --sp; x = b[sp]; y = h[sp]; }
  bool empty() const { return sp == 0; }
};
extern "C" {
void* hb8_build(const float* tri_abc, uint32_t n, int threads) { Bvh8* b = new Bvh8(); build_bvh8(tri_abc, n, *b, threads); return b; }
double hb8_cost(void* h, float cn, float ct) { return bvh8_sah_cost(*(Bvh8*)h, cn, ct); }
void hb8_free(void* h) { delete (Bvh8*)h; }
void hb8_info(void* h, uint64_t* out) { Bvh8* b = (Bvh8*)h; out[0] = b->num_nodes; out[1] = b->num_tris; out[2] = b->depth; }
// returns max stack depth used
int hb8_trace(void* h, uint32_t n, const float* o, const float* d, const float* tmax, int any, float* t, float* u, float* v, uint32_t* prim, uint64_t* counters) {
  Bvh8* b = (Bvh8*)h; int max_sp = 0; uint64_t nv = 0, tt = 0;
  for (uint32_t i = 0; i < n; ++i) {
    HostStack st; Hit hit; uint32_t a = 0, c = 0;
    v3 oo(o[3*i], o[3*i+1], o[3*i+2]), dd(d[3*i], d[3*i+1], d[3*i+2]);
    if (any) traverse8<true>((const uint32_t*)b->pool.data(), b->grid, oo, dd, tmax[i], hit, st, &a, &c);
    else traverse8<false>((const uint32_t*)b->pool.data(), b->grid, oo, dd, tmax[i], hit, st, &a, &c);
    t[i] = hit.t; u[i] = hit.u; v[i] = hit.v; prim[i] = hit.tri == 0xffffffffu ? 0xffffffffu : b->pool[hit.tri].tri.prim;
    if (st.max_sp > max_sp) max_sp = st.max_sp; nv += a; tt += c;
  }
  if (counters) { counters[0] = nv; counters[1] = tt; }
  return max_sp;
}
}
