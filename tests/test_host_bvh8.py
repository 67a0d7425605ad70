"""The device's BVH8 builder and traversal template (phosphorus_mk2_amd/csrc/bvh8.h, bvh_build.cpp) compiled
for the CPU by tests/native/host_bvh8.cpp: closest hits must equal the oracle's on the reference-layout BVH
and brute force, bit for bit — the conservative box test makes the result independent of tree topology."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits_equal, random_rays, tri_abc
from phosphorus_mk2_amd import abi


def fp(a):
    return a.ctypes.data_as(abi.f32p)


def trace(lib, h, o, d, tm, any_hit=False):
    n = len(tm)
    t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); p = np.zeros(n, np.uint32)
    ctr = (C.c_uint64 * 2)()
    sp = lib.hb8_trace(h, n, fp(o), fp(d), fp(tm), 1 if any_hit else 0, fp(t), fp(u), fp(v), p.ctypes.data_as(abi.u32p), ctr)
    return {"t": t, "u": u, "v": v, "prim": p, "max_stack": sp, "node_visits": ctr[0], "tri_tests": ctr[1]}


@pytest.mark.parametrize("kind,n", [("cornell", 12), ("soup", 7), ("soup", 500), ("soup", 20000)])
def test_bvh8_equals_oracle(host_bvh8, orc, kind, n):
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(32, 32) if kind == "cornell" else scenes.soup(n, width=32, height=32)
    abc = tri_abc(sc)
    h = host_bvh8.hb8_build(fp(abc), len(abc), 4)
    info = (C.c_uint64 * 3)(); host_bvh8.hb8_info(h, info)
    assert info[1] == len(abc)  # every triangle stored exactly once
    o, d, tm = random_rays(6000, 17)
    g = trace(host_bvh8, h, o, d, tm)
    assert g["max_stack"] <= max(0, info[2] - 1) + 0  # one pending group per level
    O = orc.Oracle(sc, spp=1)
    r = O.trace(o, d, tm, brute=True)
    assert np.array_equal(g["prim"], r["prim"]) and bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    if O.bvh_info()["nodes"]:
        r2 = O.trace(o, d, tm)
        assert np.array_equal(g["prim"], r2["prim"])
    tm2 = np.full(len(tm), 0.4, np.float32)
    a = trace(host_bvh8, h, o, d, tm2, any_hit=True)
    b = O.trace(o, d, tm2, shadow=True, brute=True)
    assert np.array_equal(a["prim"] != 0xffffffff, b["hit"])
    host_bvh8.hb8_free(h)


def stress_rays(sc, n, seed):
    """random rays + rays aimed at vertices, edge midpoints and centroids of the scene's own triangles (ties, grazing hits)"""
    abc = tri_abc(sc).reshape(-1, 3, 3)
    rng = np.random.default_rng(seed)
    o, d, tm = random_rays(n, seed)
    k = n // 2
    pick = rng.integers(0, len(abc), k)
    w = rng.dirichlet((1, 1, 1), k).astype(np.float32)
    w[: k // 3] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k // 3)]       # exactly a vertex
    w[k // 3: 2 * k // 3, 2] = 0; w[k // 3: 2 * k // 3, :2] = 0.5                # exactly an edge midpoint
    target = (abc[pick] * w[:, :, None]).sum(1)
    dd = target - o[:k]
    ln = np.linalg.norm(dd, axis=1, keepdims=True)
    ok = (ln[:, 0] > 0) & np.isfinite(ln[:, 0])
    d[:k][ok] = (dd[ok] / ln[ok]).astype(np.float32)
    return o, d, tm


def check_hits_modulo_ties(g, r, min_ties=0):
    """g: BVH8 traversal (host or device), r: the oracle.  Distances must agree bit for bit everywhere.  Where two triangles
    are hit at bitwise the same distance the reference keeps whichever its own packet order meets first (strict d < tmax),
    the BVH8 path the lowest primitive index: there the primitive may differ, and then it must be the lower one."""
    assert bits_equal(g["t"], r["t"])
    same = g["prim"] == r["prim"]
    assert bits_equal(g["u"][same], r["u"][same]) and bits_equal(g["v"][same], r["v"][same])
    assert (g["prim"][~same] < r["prim"][~same]).all()
    assert (~same).sum() >= min_ties


def test_stress_geometry_equals_brute_force(host_bvh8, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.stress()
    abc = tri_abc(sc)
    h = host_bvh8.hb8_build(fp(abc), len(abc), 4)
    o, d, tm = stress_rays(sc, 8000, 3)
    g = trace(host_bvh8, h, o, d, tm)
    r = orc.Oracle(sc, spp=1).trace(o, d, tm, brute=True)
    check_hits_modulo_ties(g, r, min_ties=100)
    assert (g["prim"] != 0xffffffff).sum() > 2000
    orc.set_tie_rule(1)  # the oracle with the device's tie rule: the primitive agrees too
    try:
        O = orc.Oracle(sc, spp=1)
        for brute in (True, False):
            r = O.trace(o, d, tm, brute=brute)
            assert np.array_equal(g["prim"], r["prim"]) and bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    finally:
        orc.set_tie_rule(0)
    host_bvh8.hb8_free(h)


def test_showroom_meshes_equal_brute_force(host_bvh8, orc):
    """indexed meshes with shared vertices, pole slivers of zero area and very different triangle sizes (scenes.showroom)"""
    from phosphorus_mk2_amd import scenes
    sc = scenes.showroom(20000, width=32, height=32)
    abc = tri_abc(sc)
    h = host_bvh8.hb8_build(fp(abc), len(abc), 4)
    o, d, tm = stress_rays(sc, 6000, 11)
    g = trace(host_bvh8, h, o, d, tm)
    orc.set_tie_rule(1)
    try:
        r = orc.Oracle(sc, spp=1).trace(o, d, tm, brute=True)
    finally:
        orc.set_tie_rule(0)
    assert (g["prim"] != 0xffffffff).sum() > 3000
    assert np.array_equal(g["prim"], r["prim"]) and bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    host_bvh8.hb8_free(h)


def test_degenerate_inputs(host_bvh8):
    # empty scene: a root that hits nothing; one triangle; coincident triangles (identical centroids)
    h = host_bvh8.hb8_build(fp(np.zeros((0, 9), np.float32)), 0, 1)
    o, d, tm = random_rays(10, 1)
    assert (trace(host_bvh8, h, o, d, tm)["prim"] == 0xffffffff).all()
    host_bvh8.hb8_free(h)
    one = np.array([[-1, -1, -3, 1, -1, -3, 0, 1, -3]], np.float32)
    h = host_bvh8.hb8_build(fp(one), 1, 1)
    r = trace(host_bvh8, h, np.zeros((1, 3), np.float32), np.array([[0, 0, -1]], np.float32), np.array([1e30], np.float32))
    assert r["prim"][0] == 0 and abs(r["t"][0] - 3) < 1e-6
    host_bvh8.hb8_free(h)
    same = np.repeat(one, 50, axis=0)
    h = host_bvh8.hb8_build(fp(same), 50, 2)
    r = trace(host_bvh8, h, np.zeros((1, 3), np.float32), np.array([[0, 0, -1]], np.float32), np.array([1e30], np.float32))
    assert r["prim"][0] == 0 and abs(r["t"][0] - 3) < 1e-6  # exact tie: the lowest primitive index wins
    host_bvh8.hb8_free(h)
    # axis-aligned flat geometry (zero extent on one axis), rays parallel to the plane
    quad = np.array([[-1, 0, -1, 1, 0, -1, 1, 0, -3], [-1, 0, -1, 1, 0, -3, -1, 0, -3]] * 6, np.float32)
    h = host_bvh8.hb8_build(fp(quad), len(quad), 1)
    r = trace(host_bvh8, h, np.array([[0, 1, -2], [0, 0.5, 0]], np.float32), np.array([[0, -1, 0], [0, 0, -1]], np.float32), np.array([1e30, 1e30], np.float32))
    assert r["prim"][0] != 0xffffffff and r["prim"][1] == 0xffffffff
    host_bvh8.hb8_free(h)


def test_any_hit_answers_do_not_depend_on_the_visiting_order(host_bvh8):
    """the study walk of scripts/width_study.py (tests/native/host_bvh8.cpp: hb8_trace_any_ordered): an any-hit ray is occluded or it is
    not, whichever child of a node is visited first — octant order (the product's), largest first, smallest first — only the work differs;
    and the PHX_WIDTH study knob (a 4-wide collapse of the same binary tree) changes the tree, not a single closest hit"""
    import os
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(20000, width=32, height=32)
    abc = tri_abc(sc)
    h = host_bvh8.hb8_build(fp(abc), len(abc), 4)
    o, d, _ = random_rays(8000, 23)
    tm = np.full(len(o), 0.3, np.float32)
    ref = trace(host_bvh8, h, o, d, tm, any_hit=True)
    want = ref["prim"] != 0xffffffff
    work = {}
    for order in (0, 1, 2):
        occ = np.zeros(len(o), np.uint8); ctr = (C.c_uint64 * 2)()
        host_bvh8.hb8_trace_any_ordered(h, len(o), fp(o), fp(d), fp(tm), order, occ.ctypes.data_as(C.POINTER(C.c_uint8)), ctr)
        assert np.array_equal(occ.astype(bool), want), order
        work[order] = (ctr[0], ctr[1])
    assert 0.2 < want.mean() < 0.98 and work[0][0] > 0 and work[1] != work[0]
    full = trace(host_bvh8, h, o, d, np.full(len(o), np.finfo(np.float32).max, np.float32))
    info8 = (C.c_uint64 * 3)(); host_bvh8.hb8_info(h, info8)
    host_bvh8.hb8_free(h)
    os.environ["PHX_WIDTH"] = "4"
    try:
        h4 = host_bvh8.hb8_build(fp(abc), len(abc), 4)
    finally:
        del os.environ["PHX_WIDTH"]
    info4 = (C.c_uint64 * 3)(); host_bvh8.hb8_info(h4, info4)
    narrow = trace(host_bvh8, h4, o, d, np.full(len(o), np.finfo(np.float32).max, np.float32))
    host_bvh8.hb8_free(h4)
    assert info4[0] > 1.5 * info8[0] and info4[1] == info8[1] == len(abc) and info4[2] > info8[2]  # more, narrower nodes; every triangle once
    assert np.array_equal(full["prim"], narrow["prim"]) and bits_equal(full["t"], narrow["t"]) and narrow["node_visits"] > full["node_visits"]
