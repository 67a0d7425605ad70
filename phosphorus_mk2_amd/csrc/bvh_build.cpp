// bvh_build.cpp — see bvh_build.h
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <thread>

// Study knobs of the builder (tree shape experiments: scripts/width_study.py, scripts/cnode_probe.sh) are read from the environment ONLY in
// builds made with -DPHX_STUDY_KNOBS=1 (tests/native/libhost_bvh8.so, `make variant NAME=study EXTRA=-DPHX_STUDY_KNOBS=1`): the product
// library reads none of them, so a stray variable cannot change its trees.
#ifndef PHX_STUDY_KNOBS
#define PHX_STUDY_KNOBS 0
#endif
static inline const char* study_knob(const char* name) {
#if PHX_STUDY_KNOBS
  return getenv(name);
#else
  (void)name; return nullptr;
#endif
}
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
// Leaf slots hold exactly ONE triangle (the pool addresses a child by its rank among the used slots, bvh8.h), so the binary
// tree is built down to single triangles; measured on the device builder: 1-triangle slots trace 23 % faster than 3.

struct Builder2 {
  const Box* pbox;
  const float* centroid;  // 3 per prim
  std::vector<uint32_t> idx;
  std::vector<Node2> nodes;
  const uint64_t* morton = nullptr;  // experiment (PHX_HOST_LBVH=1): idx is sorted by these codes and nodes split at the highest differing bit
  std::atomic<uint32_t> next{0};
  std::atomic<int> spare_threads{0};

  uint32_t alloc() { return next.fetch_add(1); }

  void build(uint32_t ni, uint32_t first, uint32_t count) {
    Node2& nd = nodes[ni];
    Box b, cb; b.reset(); cb.reset();
    for (uint32_t i = first; i < first + count; ++i) { b.grow(pbox[idx[i]]); cb.grow(centroid + 3 * (size_t)idx[i]); }
    nd.box = b; nd.first = first; nd.count = 0; nd.left = nd.right = 0;
    if (count == 1) { nd.count = 1; return; }
    if (morton) {  // the split of a linear BVH (what csrc/bvh_gpu.hip builds): where the highest differing bit of the sorted codes flips
      const uint64_t a = morton[first], z = morton[first + count - 1];
      uint32_t mid = first + count / 2;
      if (a != z) {
        const int bit = 63 - __builtin_clzll(a ^ z);
        uint32_t lo = first, hi = first + count - 1;  // last index whose bit is 0
        while (lo < hi) { const uint32_t m = (lo + hi + 1) / 2; if ((morton[m] >> bit) & 1ull) hi = m - 1; else lo = m; }
        mid = lo + 1;
      }
      const uint32_t l = alloc(), r = alloc();
      nodes[ni].left = l; nodes[ni].right = r;
      build(l, first, mid - first); build(r, mid, first + count - mid);
      return;
    }
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

// Surface-area cost of a finished pool, on the boxes the traversal really tests (the QUANTISED child boxes, as decoded from the
// nodelets): sum over all children of area(child box) x (C_NODE for a nodelet, C_TRI for a triangle), relative to the root box.
double bvh8_sah_cost(const Bvh8& b, float c_node, float c_tri) {
  if (b.pool.empty()) return 0.0;
  double total = 0.0, root_area = 0.0;
  std::vector<uint32_t> todo; todo.push_back(0);
  for (size_t qi = 0; qi < todo.size(); ++qi) {
    const uint32_t* w = b.pool[todo[qi]].w;
    float px, py, pz; uint32_t valid;
    node_origin_decode(w[0], w[1], b.grid, px, py, pz, valid);
    const uint32_t e = w[2], imask = e >> 24;
    const double sx = std::ldexp(1.0, (int)(e & 0xffu) - 127), sy = std::ldexp(1.0, (int)((e >> 8) & 0xffu) - 127), sz = std::ldexp(1.0, (int)((e >> 16) & 0xffu) - 127);
    const Node8& nd = b.pool[todo[qi]].node;
    uint32_t rank = 0;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int s = 0; s < 8; ++s) {
      if (!(valid & (1u << s))) continue;
      const double dx = (nd.qhix[s] - nd.qlox[s]) * sx, dy = (nd.qhiy[s] - nd.qloy[s]) * sy, dz = (nd.qhiz[s] - nd.qloz[s]) * sz;
      const double area = 2.0 * (dx * dy + dy * dz + dz * dx);
      const bool inner = (imask >> s) & 1u;
      total += area * (inner ? c_node : c_tri);
      if (inner) todo.push_back(nd.child_base + rank);
      ++rank;
      lo[0] = std::min(lo[0], px + nd.qlox[s] * sx); hi[0] = std::max(hi[0], px + nd.qhix[s] * sx);
      lo[1] = std::min(lo[1], py + nd.qloy[s] * sy); hi[1] = std::max(hi[1], py + nd.qhiy[s] * sy);
      lo[2] = std::min(lo[2], pz + nd.qloz[s] * sz); hi[2] = std::max(hi[2], pz + nd.qhiz[s] * sz);
    }
    if (qi == 0 && rank) { const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; root_area = 2.0 * (dx * dy + dy * dz + dz * dx); }
  }
  return root_area > 0.0 ? c_node + total / root_area : total;
}

void build_bvh8(const float* tri_abc, uint32_t n, Bvh8& out, int num_threads, const uint32_t* prim_material) {
  out.pool.clear(); out.elem_of_prim.assign(n, 0xffffffffu); out.depth = 1; out.num_nodes = 1; out.num_tris = 0;
  auto empty_node = [](Node8& nd) {
    std::memset(&nd, 0, sizeof(nd));
    nd.ex = nd.ey = nd.ez = 127;
    for (int i = 0; i < 8; ++i) { nd.qlox[i] = nd.qloy[i] = nd.qloz[i] = 255; nd.qhix[i] = nd.qhiy[i] = nd.qhiz[i] = 0; }
  };
  if (n == 0) {  // a root that hits nothing
    const float z[3] = {0.0f, 0.0f, 0.0f};
    out.grid = make_scene_grid(z, z);
    PoolElem root; empty_node(root.node);
    out.pool.push_back(root);
    return;
  }
  std::vector<Box> pbox(n);
  std::vector<float> cen(3 * (size_t)n);
  Box scene; scene.reset();
  for (uint32_t i = 0; i < n; ++i) {
    Box b; b.reset();
    b.grow(tri_abc + 9 * (size_t)i); b.grow(tri_abc + 9 * (size_t)i + 3); b.grow(tri_abc + 9 * (size_t)i + 6);
    pbox[i] = b;
    scene.grow(b);
    for (int a = 0; a < 3; ++a) cen[3 * (size_t)i + a] = 0.5f * (b.lo[a] + b.hi[a]);
  }
  {  // every triangle's box grows by the triangle test's own tolerance (bvh8.h: tri_box_inflation); the scene's with them
    const float delta = tri_box_inflation(scene.lo, scene.hi);
    for (uint32_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { pbox[i].lo[a] -= delta; pbox[i].hi[a] += delta; }
    for (int a = 0; a < 3; ++a) { scene.lo[a] -= delta; scene.hi[a] += delta; }
  }
  out.grid = make_scene_grid(scene.lo, scene.hi);
  Builder2 B;
  B.pbox = pbox.data(); B.centroid = cen.data();
  B.idx.resize(n);
  for (uint32_t i = 0; i < n; ++i) B.idx[i] = i;
  B.nodes.resize(2 * (size_t)n);
  B.next = 1;
  B.spare_threads = std::max(0, num_threads - 1);
  std::vector<uint64_t> codes;
  if (study_knob("PHX_HOST_LBVH") && atoi(study_knob("PHX_HOST_LBVH"))) {  // experiment: the device builder's binary tree, on the host
    Box cbox; cbox.reset();
    for (uint32_t i = 0; i < n; ++i) cbox.grow(cen.data() + 3 * (size_t)i);
    auto spread = [](uint64_t v) { uint64_t r = 0; for (int b = 0; b < 21; ++b) r |= ((v >> b) & 1ull) << (3 * b); return r; };
    std::vector<uint64_t> key(n);
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t q[3];
      for (int a = 0; a < 3; ++a) {
        const float ext = cbox.hi[a] - cbox.lo[a];
        float u = ext > 0.0f ? (cen[3 * (size_t)i + a] - cbox.lo[a]) / ext : 0.0f;
        u = std::min(std::max(u, 0.0f), 1.0f);
        q[a] = (uint64_t)std::min(u * 2097152.0f, 2097151.0f);
      }
      key[i] = (spread(q[0]) << 2) | (spread(q[1]) << 1) | spread(q[2]);
    }
    std::sort(B.idx.begin(), B.idx.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
    codes.resize(n);
    for (uint32_t i = 0; i < n; ++i) codes[i] = key[B.idx[i]];
    B.morton = codes.data();
  }
  B.build(0, 0, n);

  // ---- which binary nodes become 8-wide nodes: SAH-optimal collapse by dynamic programming (Ylitie, Karras, Laine: "Efficient
  // Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", 2017, section 3).  cost(n, i) = cheapest way to hang the
  // subtree of binary node n below an 8-wide node using at most i of its child slots; a slot holds one triangle or one 8-wide
  // node: cost(n, 1) = area(n) * C_NODE + distribute(n, 8) for an inner n, area(n) * C_TRI for a triangle;
  // distribute(n, j) = min over k of cost(left, k) + cost(right, j - k); cost(n, i) = min(distribute(n, i), cost(n, i - 1)).
  // Children are allocated after their parents (alloc()), so decreasing index order is bottom-up.  The greedy top-down collapse
  // it replaces left the 8-wide nodes 55 % full (4.4 children); a node visit costs ~2.4 triangle tests on the device.
  // Measured on the device (Soup frames, k_trace ms per frame, greedy -> this optimum): 100 k triangles 75.3 -> 77.8 (it hangs
  // small triangles below large nodes, whose 8-bit grids inflate their boxes: +13 % triangle tests on that frame), 300 k 89.2 ->
  // 82.7, 1 M 98.3 -> 93.6, 3 M 109.2 -> 100.1, 10 M 250.9 -> 246.0.  Mode 2 below optimises on the boxes the traversal really
  // tests and wins or ties everywhere (greedy / mode 1 / mode 2: 100 k 67.8 / 70.9 / 67.0 ms, 300 k 84.3 / 76.4 / 76.6, 1 M 95.6 /
  // 89.7 / 87.5): it is the default.  PHX_COLLAPSE=0 / 1 / 2 forces greedy / optimal on true boxes / optimal on quantised boxes.
  static const int collapse_env = study_knob("PHX_COLLAPSE") ? atoi(study_knob("PHX_COLLAPSE")) : -1;
  const int collapse_mode = collapse_env < 0 ? 2 : collapse_env;
  const bool use_dp = collapse_mode == 1;
  const float C_NODE = 2.4f, C_TRI = 1.0f;  // with one triangle per leaf slot only the sum of the nodes' areas is left to minimise
  std::vector<float> dp_cost; std::vector<uint8_t> dp_split;
  if (use_dp) {
    const uint32_t nn = B.next.load();
    dp_cost.assign((size_t)nn * 9, 0.0f); dp_split.assign((size_t)nn * 9, 0);
    for (uint32_t v = nn; v-- > 0;) {
      const Node2& nd = B.nodes[v];
      float* c = &dp_cost[(size_t)v * 9]; uint8_t* sp = &dp_split[(size_t)v * 9];
      const float area = nd.box.area();
      if (nd.count > 0) { for (int i = 1; i <= 8; ++i) c[i] = area * C_TRI; continue; }
      const float* cl = &dp_cost[(size_t)nd.left * 9]; const float* cr = &dp_cost[(size_t)nd.right * 9];
      float dist[9];
      for (int j = 2; j <= 8; ++j) {
        float best = FLT_MAX; int bk = 1;
        for (int k = 1; k < j; ++k) { const float t = cl[std::min(k, 7)] + cr[std::min(j - k, 7)]; if (t < best) { best = t; bk = k; } }
        dist[j] = best; sp[j] = (uint8_t)bk;
      }
      c[1] = area * C_NODE + dist[8];
      for (int i = 2; i <= 7; ++i) { if (dist[i] < c[i - 1]) c[i] = dist[i]; else { c[i] = c[i - 1]; sp[i] = 0; } }
      c[8] = dist[8];
    }
  }

  // ---- mode 2: the same optimisation on the boxes the traversal really tests.  A child's box is stored on its parent's 8-bit
  // grid, so it is inflated by up to two grid units per axis of THAT parent: a small triangle hung below a large node is tested
  // far more often than its own area says (what made the classic optimum lose at 100 k triangles).  The cost of a cut element
  // therefore depends on the 8-wide node m it hangs below: sub(m) = min over cuts K of m's binary subtree, |K| <= 8, of the sum
  // over c in K of area(box(c) inflated by m's grid) * (C_NODE | C_TRI) + (sub(c) if c is inner).  A cut of <= 8 elements lies
  // within 7 binary levels below m, so sub(m) is an exact local dynamic programme over (descendant, slots) given the sub() of the
  // descendants: bottom-up over the binary tree (children have larger indices than their parents).
  struct Cut { uint32_t node[8]; uint8_t count; };
  std::vector<Cut> cuts;
  if (collapse_mode == 2) {
    const uint32_t nn = B.next.load();
    std::vector<float> sub(nn, 0.0f);
    cuts.resize(nn);
    const float CN = study_knob("PHX_CNODE") ? (float)atof(study_knob("PHX_CNODE")) : 1.6f, CT = 1.0f;  // measured VALU time per node visit : per triangle test
    // experiment knob (16 = cuts within 4 levels), clamped: the DP tables below hold 256 heap positions and a position h < limit
    // looks at its children 2h and 2h+1
    static const uint32_t dp_heap_limit = (uint32_t)std::min(128, std::max(2, study_knob("PHX_DP_HEAP") ? atoi(study_knob("PHX_DP_HEAP")) : 128));
    // study knob (scripts/width_study.py, read per build): at most this many children per node — 4 gives the tree a 4-wide collapse of the
    // same binary tree would have, in the same 64-byte nodelets (half of their slots empty): node visits and triangle tests of a
    // narrower layout can be counted with the unchanged traversal
    const int width = std::min(8, std::max(2, study_knob("PHX_WIDTH") ? atoi(study_knob("PHX_WIDTH")) : 8));
    struct Local {
      float best[256][9]; uint8_t split[256][9]; uint8_t done[256][9]; uint32_t node[256];
    };
    auto solve = [&](uint32_t m, Local& L) {
      const Node2& M = B.nodes[m];
      float g2[3];  // two grid units of m per axis
      for (int a = 0; a < 3; ++a) {
        const double ext = (double)M.box.hi[a] - (double)M.box.lo[a];
        int e = -126;
        if (ext > 0.0) { e = (int)std::ceil(std::log2(ext * 1.00001 / 255.0)); }
        e = std::max(-126, std::min(127, e));
        g2[a] = 2.0f * (float)std::ldexp(1.0, e);
      }
      auto areaq = [&](const Box& b) {
        const float dx = b.hi[0] - b.lo[0] + g2[0], dy = b.hi[1] - b.lo[1] + g2[1], dz = b.hi[2] - b.lo[2] + g2[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
      };
      std::memset(L.done, 0, sizeof(L.done));
      // best(h, j): cheapest way to hang the binary node at heap position h below m using <= j slots
      std::function<float(uint32_t, uint32_t, int)> best = [&](uint32_t h, uint32_t v, int j) -> float {
        if (L.done[h][j]) return L.best[h][j];
        L.node[h] = v;
        const Node2& nd = B.nodes[v];
        float r; uint8_t sp = 0;
        if (nd.count > 0) r = areaq(nd.box) * CT;
        else {
          r = areaq(nd.box) * CN + sub[v];  // as ONE slot: an 8-wide node of its own
          if (j > 1 && h < dp_heap_limit) {
            for (int k = 1; k < j; ++k) {
              const float t = best(2 * h, nd.left, k) + best(2 * h + 1, nd.right, j - k);
              if (t < r) { r = t; sp = (uint8_t)k; }
            }
          }
        }
        L.best[h][j] = r; L.split[h][j] = sp; L.done[h][j] = 1;
        return r;
      };
      float r = FLT_MAX; int bk = 1;
      for (int k = 1; k < width; ++k) {
        const float t = best(2, M.left, k) + best(3, M.right, width - k);
        if (t < r) { r = t; bk = k; }
      }
      sub[m] = r;
      // the cut that achieves it
      Cut& C = cuts[m]; C.count = 0;
      struct Item { uint32_t h; int j; };
      Item st[16]; int top = 0;
      st[top++] = Item{3, width - bk}; st[top++] = Item{2, bk};
      while (top > 0) {
        const Item it = st[--top];
        const uint8_t k = L.split[it.h][it.j];
        if (k == 0) { C.node[C.count++] = L.node[it.h]; continue; }
        st[top++] = Item{2 * it.h + 1, it.j - (int)k}; st[top++] = Item{2 * it.h, (int)k};
      }
    };
    // bottom-up: decreasing index order, in parallel over index ranges level by level would need heights; the plain loop is
    // 0.3 s per 100 k triangles, so large inputs are split over threads by subtree (a subtree's nodes only depend on themselves)
    std::function<void(uint32_t)> run = [&](uint32_t v) {
      const Node2& nd = B.nodes[v];
      if (nd.count > 0) return;
      // subtree size is not stored: use the primitive range of the build (first/count live in leaves only) -> recurse, fork near the top
      run(nd.left); run(nd.right);
      static thread_local Local L;
      solve(v, L);
    };
    // fork the top levels: collect subtree roots at depth ~ log2(threads) + 2, run them on threads, then finish the top serially
    std::vector<uint32_t> roots{0}, top_nodes;
    const int want = std::max(1, num_threads) * 4;
    while ((int)roots.size() < want) {
      std::vector<uint32_t> next_roots; bool grew = false;
      for (uint32_t v : roots) {
        const Node2& nd = B.nodes[v];
        if (nd.count > 0) { next_roots.push_back(v); continue; }
        top_nodes.push_back(v); next_roots.push_back(nd.left); next_roots.push_back(nd.right); grew = true;
      }
      roots.swap(next_roots);
      if (!grew) break;
    }
    std::atomic<size_t> cursor{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < std::max(1, num_threads); ++t)
      pool.emplace_back([&]() { for (;;) { const size_t i = cursor++; if (i >= roots.size()) break; run(roots[i]); } });
    for (auto& th : pool) th.join();
    std::sort(top_nodes.begin(), top_nodes.end(), [](uint32_t a, uint32_t b) { return a > b; });  // children before parents
    static thread_local Local Ltop;
    for (uint32_t v : top_nodes) solve(v, Ltop);
    out.cost = sub[0];
  }

  // ---- collapse to 8-wide, breadth first --------------------------------------------------------
  struct Work { uint32_t n2; uint32_t n8; uint32_t depth; };
  std::deque<Work> queue;
  out.pool.reserve((size_t)n + (size_t)n / 3 + 16);
  out.pool.emplace_back();
  queue.push_back(Work{0, 0, 1});
  // (breadth first: k_trace stages the first elements of the pool in LDS, so they must be the top levels; a depth-first order
  // below the staged part was measured at 1 M and 4 M triangles and changes the trace time by < 1 %, profiles/README.md)
  while (!queue.empty()) {
    const Work wk = queue.front(); queue.pop_front();
    out.depth = std::max(out.depth, wk.depth);
    const Node2& r = B.nodes[wk.n2];
    uint32_t ch[8]; int nch = 0;
    if (r.count > 0) { ch[nch++] = wk.n2; }  // degenerate: the whole tree is one triangle
    else if (collapse_mode == 2) {
      const Cut& C = cuts[wk.n2];
      for (int i = 0; i < C.count; ++i) ch[nch++] = C.node[i];
    } else if (use_dp) {
      // the SAH-optimal cut of this binary subtree into <= 8 children (collapse_dp above): follow the recorded decisions
      struct Item { uint32_t node; int slots; };
      Item st[16]; int sp = 0;
      const int k = dp_split[(size_t)wk.n2 * 9 + 8];
      st[sp++] = Item{r.right, 8 - k}; st[sp++] = Item{r.left, k};
      while (sp > 0) {
        const Item it = st[--sp];
        const Node2& c = B.nodes[it.node];
        if (c.count > 0 || it.slots == 1) { ch[nch++] = it.node; continue; }
        int i = it.slots;
        while (i > 1 && dp_split[(size_t)it.node * 9 + i] == 0) --i;  // 0: "as good with one slot fewer"
        if (i == 1) { ch[nch++] = it.node; continue; }
        const int kk = dp_split[(size_t)it.node * 9 + i];
        st[sp++] = Item{c.right, i - kk}; st[sp++] = Item{c.left, kk};
      }
    } else {
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

    Node8 nd; empty_node(nd);
    // origin on the scene grid, not above the node's lower corner; everything below is relative to the DECODED origin
    uint32_t gi[3]; float org[3];
    for (int a = 0; a < 3; ++a) { gi[a] = grid_index_below(nb.lo[a], out.grid.lo[a], out.grid.cell[a]); org[a] = fmaf((float)gi[a], out.grid.cell[a], out.grid.lo[a]); }
    float scale[3]; uint8_t eb[3];
    for (int a = 0; a < 3; ++a) {
      const double ext = (double)nb.hi[a] - (double)org[a];
      int e = -126;
      if (ext > 0.0) {
        e = (int)std::ceil(std::log2(ext * 1.00001 / 255.0));
        while (std::ldexp(255.0, e) < ext * 1.00001) ++e;
      }
      e = std::max(-126, std::min(127, e));
      eb[a] = (uint8_t)(e + 127);
      scale[a] = (float)std::ldexp(1.0, e);
    }
    nd.ex = eb[0]; nd.ey = eb[1]; nd.ez = eb[2];
    nd.child_base = (uint32_t)out.pool.size();
    uint32_t valid = 0;
    for (int s = 0; s < 8; ++s) {
      const int i = child_in_slot[s];
      if (i < 0) continue;  // empty slot: inverted box, never hit, not in the valid mask
      valid |= 1u << s;
      const Node2& c = B.nodes[ch[i]];
      uint8_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz}; uint8_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
      for (int a = 0; a < 3; ++a) {
        // outward rounding with 1e-3 grid units of slack against the decode's rounding error
        double lo = std::floor(((double)c.box.lo[a] - (double)org[a]) / (double)scale[a] - 1e-3);
        double hi = std::ceil(((double)c.box.hi[a] - (double)org[a]) / (double)scale[a] + 1e-3);
        lo = std::max(0.0, std::min(255.0, lo)); hi = std::max(0.0, std::min(255.0, hi));
        qlo[a][s] = (uint8_t)lo; qhi[a][s] = (uint8_t)hi;
      }
      // the child's pool element, in slot order: a triangle record now, or a nodelet filled in when its turn comes
      const uint32_t ei = (uint32_t)out.pool.size();
      out.pool.emplace_back();
      if (c.count > 0) {
        const uint32_t p = B.idx[c.first];
        const float* t = tri_abc + 9 * (size_t)p;
        TriRec T; std::memset(&T, 0, sizeof(T));
        T.v0x = t[0]; T.v0y = t[1]; T.v0z = t[2];
        T.e0x = t[3] - t[0]; T.e0y = t[4] - t[1]; T.e0z = t[5] - t[2];  // e0 = b - a, e1 = c - a (triangle.hpp:48-50)
        T.e1x = t[6] - t[0]; T.e1y = t[7] - t[1]; T.e1z = t[8] - t[2];
        T.prim = p;
        T.material = prim_material ? prim_material[p] : 0u;
        out.pool[ei].tri = T;
        out.elem_of_prim[p] = ei;
        ++out.num_tris;
      } else {
        nd.imask |= (uint8_t)(1u << s);
        ++out.num_nodes;
        queue.push_back(Work{ch[i], ei, wk.depth + 1});
      }
    }
    node_origin_encode(nd, gi[0], gi[1], gi[2], valid);
    out.pool[wk.n8].node = nd;
  }
}

}  // namespace phx
