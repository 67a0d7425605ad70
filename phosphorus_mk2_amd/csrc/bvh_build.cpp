// bvh_build.cpp — see bvh_build.h
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <thread>

namespace phx {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() { for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
  void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
  void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
  float area() const {
    if (hi[0] < lo[0]) return 0.0f;
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};

struct Node2 {
  Box box;
  uint32_t left, right;   // children (inner)
  uint32_t first, count;  // leaf range in the index array (count > 0 => leaf)
};

const int BINS = 16;
const int MAX_LEAF = 3;       // triangles per leaf child slot (unary count fits meta's 3 bits)
static float C_TRAV = 0.15f;  // cost of one child-slot box test relative to one triangle test (measured: 0.15 traces 1.4 % faster than 0.35)

struct Builder2 {
  const Box* pbox;
  const float* centroid;  // 3 per prim
  std::vector<uint32_t> idx;
  std::vector<Node2> nodes;
  std::atomic<uint32_t> next{0};
  std::atomic<int> spare_threads{0};

  uint32_t alloc() { return next.fetch_add(1); }

  void build(uint32_t ni, uint32_t first, uint32_t count) {
    Node2& nd = nodes[ni];
    Box b, cb; b.reset(); cb.reset();
    for (uint32_t i = first; i < first + count; ++i) { b.grow(pbox[idx[i]]); cb.grow(centroid + 3 * (size_t)idx[i]); }
    nd.box = b; nd.first = first; nd.count = 0; nd.left = nd.right = 0;
    if (count == 1) { nd.count = 1; return; }
    // binned SAH over the three axes
    float best_cost = FLT_MAX; int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; ++a) {
      const float ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0.0f)) continue;
      Box bb[BINS]; uint32_t bc[BINS];
      for (int k = 0; k < BINS; ++k) { bb[k].reset(); bc[k] = 0; }
      const float scale = (float)BINS / ext;
      for (uint32_t i = first; i < first + count; ++i) {
        const uint32_t p = idx[i];
        int k = (int)((centroid[3 * (size_t)p + a] - cb.lo[a]) * scale);
        k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
        bb[k].grow(pbox[p]); bc[k]++;
      }
      float right_area[BINS]; uint32_t right_cnt[BINS];
      Box acc; acc.reset(); uint32_t c = 0;
      for (int k = BINS - 1; k > 0; --k) { acc.grow(bb[k]); c += bc[k]; right_area[k] = acc.area(); right_cnt[k] = c; }
      acc.reset(); c = 0;
      for (int k = 0; k < BINS - 1; ++k) {
        acc.grow(bb[k]); c += bc[k];
        if (c == 0 || right_cnt[k + 1] == 0) continue;
        const float cost = acc.area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
      }
    }
    const float leaf_cost = b.area() * (float)count;
    if (count <= (uint32_t)MAX_LEAF) {
      if (best_axis < 0 || C_TRAV * 2.0f * b.area() + best_cost >= leaf_cost) { nd.count = count; return; }
    }
    uint32_t mid;
    if (best_axis >= 0) {
      const float ext = cb.hi[best_axis] - cb.lo[best_axis];
      const float scale = (float)BINS / ext; const float lo = cb.lo[best_axis]; const int a = best_axis; const int bin = best_bin;
      const float* cen = centroid;
      auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [=](uint32_t p) {
        int k = (int)((cen[3 * (size_t)p + a] - lo) * scale);
        k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
        return k <= bin;
      });
      mid = (uint32_t)(it - idx.begin());
    } else {
      mid = first + count / 2;  // coincident centroids: split by index
    }
    if (mid == first || mid == first + count) mid = first + count / 2;
    const uint32_t l = alloc(), r = alloc();
    nodes[ni].left = l; nodes[ni].right = r;
    const uint32_t lc = mid - first, rc = count - lc;
    if (count > 32768 && spare_threads.fetch_sub(1) > 0) {
      auto fut = std::async(std::launch::async, [this, l, first, lc]() { build(l, first, lc); });
      build(r, mid, rc);
      fut.get();
      spare_threads.fetch_add(1);
    } else {
      if (count > 32768) spare_threads.fetch_add(1);  // undo the failed reservation
      build(l, first, lc);
      build(r, mid, rc);
    }
  }
};

// direction favoured by slot s: the ray octant for which slot s is visited first
inline void slot_dir(int s, float* d) { d[0] = (s & 4) ? -1.0f : 1.0f; d[1] = (s & 2) ? -1.0f : 1.0f; d[2] = (s & 1) ? -1.0f : 1.0f; }

}  // namespace

void build_bvh8(const float* tri_abc, uint32_t n, Bvh8& out, int num_threads) {
  if (const char* e = getenv("PHX_CTRAV")) C_TRAV = (float)atof(e);
  out.nodes.clear(); out.tris.clear(); out.depth = 1;
  if (n == 0) {  // a root that hits nothing
    Node8 root; std::memset(&root, 0, sizeof(root));
    root.ex = root.ey = root.ez = 127;
    for (int i = 0; i < 8; ++i) { root.qlox[i] = root.qloy[i] = root.qloz[i] = 255; root.qhix[i] = root.qhiy[i] = root.qhiz[i] = 0; }
    out.nodes.push_back(root);
    return;
  }
  std::vector<Box> pbox(n);
  std::vector<float> cen(3 * (size_t)n);
  for (uint32_t i = 0; i < n; ++i) {
    Box b; b.reset();
    b.grow(tri_abc + 9 * (size_t)i); b.grow(tri_abc + 9 * (size_t)i + 3); b.grow(tri_abc + 9 * (size_t)i + 6);
    pbox[i] = b;
    for (int a = 0; a < 3; ++a) cen[3 * (size_t)i + a] = 0.5f * (b.lo[a] + b.hi[a]);
  }
  Builder2 B;
  B.pbox = pbox.data(); B.centroid = cen.data();
  B.idx.resize(n);
  for (uint32_t i = 0; i < n; ++i) B.idx[i] = i;
  B.nodes.resize(2 * (size_t)n);
  B.next = 1;
  B.spare_threads = std::max(0, num_threads - 1);
  B.build(0, 0, n);

  // ---- collapse to 8-wide, breadth first --------------------------------------------------------
  struct Work { uint32_t n2; uint32_t n8; uint32_t depth; };
  std::deque<Work> queue;
  out.nodes.reserve((size_t)n / 4 + 16);
  out.tris.reserve(n);
  out.nodes.emplace_back();
  queue.push_back(Work{0, 0, 1});
  while (!queue.empty()) {
    const Work wk = queue.front(); queue.pop_front();
    out.depth = std::max(out.depth, wk.depth);
    const Node2& r = B.nodes[wk.n2];
    uint32_t ch[8]; int nch = 0;
    if (r.count > 0) { ch[nch++] = wk.n2; }  // degenerate: the whole tree is one leaf
    else {
      ch[nch++] = r.left; ch[nch++] = r.right;
      while (nch < 8) {
        int pick = -1; float best = -1.0f;
        for (int i = 0; i < nch; ++i) {
          const Node2& c = B.nodes[ch[i]];
          if (c.count > 0) continue;
          const float a = c.box.area();
          if (a > best) { best = a; pick = i; }
        }
        if (pick < 0) break;
        const Node2& c = B.nodes[ch[pick]];
        ch[pick] = c.left; ch[nch++] = c.right;
      }
    }
    // octant-order slot assignment: greedy minimum of dot(centroid_child - centroid_node, slot_dir)
    const Box nb = r.box;
    float cost[8][8];
    for (int i = 0; i < nch; ++i) {
      const Box& cb = B.nodes[ch[i]].box;
      for (int s = 0; s < 8; ++s) {
        float d[3]; slot_dir(s, d);
        float v = 0.0f;
        for (int a = 0; a < 3; ++a) v += (0.5f * (cb.lo[a] + cb.hi[a]) - 0.5f * (nb.lo[a] + nb.hi[a])) * d[a];
        cost[i][s] = v;
      }
    }
    int slot_of[8]; bool slot_used[8] = {false, false, false, false, false, false, false, false}; bool child_done[8] = {false, false, false, false, false, false, false, false};
    for (int k = 0; k < nch; ++k) {
      int bi = -1, bs = -1; float bc = FLT_MAX;
      for (int i = 0; i < nch; ++i) if (!child_done[i])
        for (int s = 0; s < 8; ++s) if (!slot_used[s] && cost[i][s] < bc) { bc = cost[i][s]; bi = i; bs = s; }
      slot_of[bi] = bs; slot_used[bs] = true; child_done[bi] = true;
    }
    int child_in_slot[8]; for (int s = 0; s < 8; ++s) child_in_slot[s] = -1;
    for (int i = 0; i < nch; ++i) child_in_slot[slot_of[i]] = i;

    Node8 nd; std::memset(&nd, 0, sizeof(nd));
    nd.px = nb.lo[0]; nd.py = nb.lo[1]; nd.pz = nb.lo[2];
    float scale[3]; uint8_t eb[3];
    for (int a = 0; a < 3; ++a) {
      const float ext = nb.hi[a] - nb.lo[a];
      int e = -126;
      if (ext > 0.0f) {
        e = (int)std::ceil(std::log2((double)ext * 1.00001 / 255.0));
        while (std::ldexp(255.0, e) < (double)ext * 1.00001) ++e;
      }
      e = std::max(-126, std::min(127, e));
      eb[a] = (uint8_t)(e + 127);
      scale[a] = (float)std::ldexp(1.0, e);
    }
    nd.ex = eb[0]; nd.ey = eb[1]; nd.ez = eb[2];
    nd.child_base = (uint32_t)out.nodes.size();
    nd.tri_base = (uint32_t)out.tris.size();
    for (int s = 0; s < 8; ++s) {
      const int i = child_in_slot[s];
      if (i < 0) {  // empty slot: inverted box, never hit
        nd.qlox[s] = nd.qloy[s] = nd.qloz[s] = 255; nd.qhix[s] = nd.qhiy[s] = nd.qhiz[s] = 0;
        continue;
      }
      const Node2& c = B.nodes[ch[i]];
      uint8_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz}; uint8_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
      for (int a = 0; a < 3; ++a) {
        // outward rounding with 1e-3 grid units of slack against the decode's rounding error
        double lo = std::floor(((double)c.box.lo[a] - (double)nb.lo[a]) / (double)scale[a] - 1e-3);
        double hi = std::ceil(((double)c.box.hi[a] - (double)nb.lo[a]) / (double)scale[a] + 1e-3);
        lo = std::max(0.0, std::min(255.0, lo)); hi = std::max(0.0, std::min(255.0, hi));
        qlo[a][s] = (uint8_t)lo; qhi[a][s] = (uint8_t)hi;
      }
      if (c.count > 0) {
        for (uint32_t k = 0; k < c.count; ++k) nd.tmask |= 1u << (s + 8 * (int)k);
      } else {
        nd.imask |= (uint8_t)(1u << s);
        const uint32_t n8 = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        queue.push_back(Work{ch[i], n8, wk.depth + 1});
      }
    }
    // triangle records in tmask bit order: bit (s + 8*j) = j-th triangle of leaf slot s
    for (int bit = 0; bit < 24; ++bit) {
      if (!(nd.tmask & (1u << bit))) continue;
      const int s = bit & 7, k = bit >> 3;
      const Node2& c = B.nodes[ch[child_in_slot[s]]];
      const uint32_t p = B.idx[c.first + (uint32_t)k];
      const float* t = tri_abc + 9 * (size_t)p;
      TriRec T; std::memset(&T, 0, sizeof(T));
      T.v0x = t[0]; T.v0y = t[1]; T.v0z = t[2];
      T.e0x = t[3] - t[0]; T.e0y = t[4] - t[1]; T.e0z = t[5] - t[2];  // e0 = b - a, e1 = c - a (triangle.hpp:48-50)
      T.e1x = t[6] - t[0]; T.e1y = t[7] - t[1]; T.e1z = t[8] - t[2];
      T.prim = p;
      out.tris.push_back(T);
    }
    out.nodes[wk.n8] = nd;
  }
}

}  // namespace phx
