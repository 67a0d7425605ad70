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

// Study only (scripts/width_study.py): an any-hit walk that visits the hit inner children of a node by DECREASING box area (a proxy for
// "most likely to hold an occluder") instead of the ray's octant order.  Any-hit results do not depend on the order, the work does.
// order: 0 = octant order (what traverse8<true> does), 1 = largest child first, 2 = smallest first.  counters: node visits, triangle tests.
int hb8_trace_any_ordered(void* h, uint32_t n, const float* o, const float* d, const float* tmax, int order, uint8_t* occluded, uint64_t* counters) {
  Bvh8* b = (Bvh8*)h; uint64_t nv = 0, tt = 0;
  const uint32_t* pool = (const uint32_t*)b->pool.data();
  for (uint32_t i = 0; i < n; ++i) {
    const v3 oo(o[3*i], o[3*i+1], o[3*i+2]), dd(d[3*i], d[3*i+1], d[3*i+2]);
    const RayCtx r = make_ray_ctx(oo, dd);
    uint32_t stack[512]; int sp = 0; stack[sp++] = 0; bool hit_any = false;
    while (sp > 0 && !hit_any) {
      const uint32_t ni = stack[--sp];
      const uint32_t* w = pool + (size_t)ni * 16u; ++nv;
      const uint32_t hm = node_hitmask(w, b->grid, r, tmax[i], [](uint32_t m) { return m; });  // inner hits by SLOT (no octant permutation)
      const uint32_t base = w[3], valid = hm & 0xffu;
      uint32_t th = (hm >> 16) & 0xffu;
      while (th && !hit_any) {
        const uint32_t k = 31u - (uint32_t)clz32(th); th &= ~(1u << k);
        const uint32_t ti = base + (uint32_t)popc32(valid & ~(0xffffffffu << k));
        const TriRec T = *reinterpret_cast<const TriRec*>(pool + (size_t)ti * 16u);
        float us, vs, ds; ++tt;
        if (mt_intersect(T, oo, dd, tmax[i], 0u, us, vs, ds)) hit_any = true;
      }
      if (hit_any) break;
      uint32_t ih = hm >> 24;  // hit inner children by slot
      struct C { float key; uint32_t idx; } c[8]; int nc = 0;
      const Node8& nd = b->pool[ni].node;
      while (ih) {
        const uint32_t s = 31u - (uint32_t)clz32(ih); ih &= ~(1u << s);
        const float ex = std::ldexp(1.0f, (int)nd.ex - 127), ey = std::ldexp(1.0f, (int)nd.ey - 127), ez = std::ldexp(1.0f, (int)nd.ez - 127);
        const float dx = (float)(nd.qhix[s] - nd.qlox[s]) * ex, dy = (float)(nd.qhiy[s] - nd.qloy[s]) * ey, dz = (float)(nd.qhiz[s] - nd.qloz[s]) * ez;
        float key;
        if (order == 0) key = (float)(s ^ r.oct_inv);                 // highest (slot ^ octant) first, like the product walk
        else { key = dx * dy + dy * dz + dz * dx; if (order == 2) key = -key; }
        c[nc++] = C{key, base + (uint32_t)popc32(valid & ~(0xffffffffu << s))};
      }
      // push in increasing key order so that the largest key is popped first
      for (int a = 1; a < nc; ++a) { C x = c[a]; int j = a - 1; while (j >= 0 && c[j].key > x.key) { c[j + 1] = c[j]; --j; } c[j + 1] = x; }
      for (int a = 0; a < nc; ++a) if (sp < 512) stack[sp++] = c[a].idx;
    }
    occluded[i] = hit_any;
  }
  if (counters) { counters[0] = nv; counters[1] = tt; }
  return 0;
}
}
