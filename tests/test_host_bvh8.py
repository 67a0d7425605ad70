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
    assert r["prim"][0] < 50 and abs(r["t"][0] - 3) < 1e-6
    host_bvh8.hb8_free(h)
    # axis-aligned flat geometry (zero extent on one axis), rays parallel to the plane
    quad = np.array([[-1, 0, -1, 1, 0, -1, 1, 0, -3], [-1, 0, -1, 1, 0, -3, -1, 0, -3]] * 6, np.float32)
    h = host_bvh8.hb8_build(fp(quad), len(quad), 1)
    r = trace(host_bvh8, h, np.array([[0, 1, -2], [0, 0.5, 0]], np.float32), np.array([[0, -1, 0], [0, 0, -1]], np.float32), np.array([1e30, 1e30], np.float32))
    assert r["prim"][0] != 0xffffffff and r["prim"][1] == 0xffffffff
    host_bvh8.hb8_free(h)
