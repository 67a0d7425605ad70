"""The oracle against everything that pins it:
  (1) vectors produced by the reference's own object code for the dependency-free headers
      (tests/golden/ref_subset_vectors.npz, generator: tests/golden/make_ref_subset_vectors.py), and the
      live oracle/_ref library when it is present;
  (2) statistics of the real reference recorded by the survey in this container:
      mt19937 head, Cornell rays per camera sample, BVH visits per ray;
  (3) self-consistency: MBVH-RS stream traversal == brute force, literal == conservative slab test.
Everything else of the oracle is "parity unpinned" (see oracle/orender.cpp header)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, bits_equal, random_rays

G = np.load(os.path.join(ROOT, "tests", "golden", "ref_subset_vectors.npz"))


def fp(a):
    from phosphorus_mk2_amd import abi
    return a.ctypes.data_as(abi.f32p)


def up(a):
    from phosphorus_mk2_amd import abi
    return a.ctypes.data_as(abi.u32p)


def test_fresnel_matches_reference_object_code(orc):
    lib = orc.load()
    out = np.zeros_like(G["fresnel"])
    lib.orc_fresnel_dielectric(len(out), fp(np.ascontiguousarray(G["cosi"])), fp(np.ascontiguousarray(G["eta"])), fp(out))
    assert bits_equal(out, G["fresnel"])


def test_radians_matches_reference_object_code(orc):
    lib = orc.load()
    lib.orc_radians.argtypes = [C.c_uint32, orc.abi.f32p, orc.abi.f32p]
    out = np.zeros_like(G["rad"])
    lib.orc_radians(len(out), fp(np.ascontiguousarray(G["deg"])), fp(out))
    assert bits_equal(out, G["rad"])


def test_simd_wrapper_semantics_match_reference_object_code(orc):
    lib = orc.load()
    lib.orc_simd_select.argtypes = [C.c_uint32, orc.abi.u32p, orc.abi.f32p, orc.abi.f32p, orc.abi.f32p]
    lib.orc_simd_minmax.argtypes = [C.c_uint32, C.c_int, orc.abi.f32p, orc.abi.f32p, orc.abi.f32p]
    lib.orc_simd_cmp.argtypes = [C.c_uint32, C.c_int, orc.abi.f32p, orc.abi.f32p, orc.abi.u32p]
    l = np.ascontiguousarray(G["l"]).reshape(-1); r = np.ascontiguousarray(G["r"]).reshape(-1)
    m = np.ascontiguousarray(G["mask"]).reshape(-1).view(np.uint32)
    out = np.zeros_like(l)
    lib.orc_simd_select(len(l), up(m), fp(l), fp(r), fp(out))
    assert bits_equal(out, G["select"].reshape(-1))
    for k in range(2):
        lib.orc_simd_minmax(len(l), k, fp(l), fp(r), fp(out))
        assert bits_equal(out, G["minmax"][k].reshape(-1))
    bits = np.zeros(len(l), np.uint32)
    for op in range(4):
        lib.orc_simd_cmp(len(l), op, fp(l), fp(r), up(bits))
        assert np.array_equal(bits, G["cmp"][op].reshape(-1))
    lib.orc_bscf.restype = C.c_uint64; lib.orc_bscf.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    for v, i, rest in zip(G["bscf_in"], G["bscf_idx"], G["bscf_rest"]):
        rr = C.c_uint64()
        assert lib.orc_bscf(int(v), C.byref(rr)) == int(i) and rr.value == int(rest)


def test_live_ref_subset_if_present(orc):
    so = os.path.join(ROOT, "oracle", "_ref", "libphx_ref_subset.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    ref = C.CDLL(so)
    rng = np.random.default_rng(7)
    cosi = rng.uniform(-1, 1, 4096).astype(np.float32); eta = rng.uniform(0.3, 3.0, 4096).astype(np.float32)
    a = np.zeros(4096, np.float32); b = np.zeros(4096, np.float32)
    ref.ref_fresnel_dielectric(4096, fp(cosi), fp(eta), fp(a))
    orc.load().orc_fresnel_dielectric(4096, fp(cosi), fp(eta), fp(b))
    assert bits_equal(a, b)
    # the reference's RCPPS wrapper vs the oracle's approx mode: same instruction on the same CPU
    x = rng.uniform(0.01, 100, 8).astype(np.float32); y = np.zeros(8, np.float32)
    ref.ref_rcp8(fp(x), fp(y))
    assert np.allclose(y, 1.0 / x, rtol=4e-4)


def test_mt19937_stream_head(orc):
    """SURVEY A-5: std::mt19937 (seed 5489) through uniform_real_distribution<float>(0,1)."""
    out = np.zeros(3, np.float32)
    orc.load().orc_mt19937_head(3, fp(out))
    assert np.allclose(out, [0.81472367, 0.135477006, 0.905791938], rtol=0, atol=1e-8)


def test_cornell_statistics_match_survey_run_of_the_reference(orc):
    """The survey ran the real reference on its Cornell box (256x256, 16 spp, 1 thread): 2.57 M closest +
    1.90 M shadow rays, 0.67 M masked, 30.6 M RNG draws, 1.00 node + 1.39 packet visits per ray.
    The restatement in reference RNG order reproduces those within the scene-description uncertainty
    (the survey's quad winding/order is not recorded): 2 % on ray counts."""
    from phosphorus_mk2_amd import scenes
    O = orc.Oracle(scenes.cornell(256, 256), spp=16)
    assert O.bvh_info() == {"nodes": 1, "packets": 3, "triangles": 12}
    _, st = O.render(rng=orc.RNG_SEQ, slab_literal=1)
    assert st["camera_samples"] == 256 * 256 * 16
    assert abs(st["rays_closest"] / 2.57e6 - 1) < 0.02
    assert abs(st["rays_shadow"] / 1.90e6 - 1) < 0.02
    assert abs(st["rays_masked"] / 0.67e6 - 1) < 0.02
    assert abs(st["rng_draws"] / 30.6e6 - 1) < 0.02
    rays = st["rays_closest"] + st["rays_shadow"]
    assert (st["node_visits_closest"] + st["node_visits_shadow"]) == rays  # exactly 1.00 node visit per ray
    assert 1.2 < (st["packet_visits_closest"] + st["packet_visits_shadow"]) / rays < 1.5


@pytest.mark.parametrize("n", [64, 3000])
def test_stream_traversal_equals_brute_force(orc, n):
    from phosphorus_mk2_amd import scenes
    O = orc.Oracle(scenes.soup(n, width=32, height=32), spp=1)
    o, d, tm = random_rays(4000, 3)
    a = O.trace(o, d, tm); b = O.trace(o, d, tm, brute=True); c = O.trace(o, d, tm, slab_literal=1)
    assert np.array_equal(a["prim"], b["prim"]) and bits_equal(a["t"], b["t"]) and bits_equal(a["u"], b["u"])
    assert np.array_equal(a["prim"], c["prim"])  # the literal slab test loses no hit on these rays
    s1 = O.trace(o, d, np.full(len(tm), 0.5, np.float32), shadow=True)
    s2 = O.trace(o, d, np.full(len(tm), 0.5, np.float32), shadow=True, brute=True)
    assert np.array_equal(s1["hit"], s2["hit"])


def test_reference_builder_invariants(orc):
    """binned_sah_builder.hpp: every primitive lands in exactly one packet slot, leaves hold <= 255
    primitives (uint8_t num, SURVEY A-13), child boxes enclose their packets' triangles."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(5000, width=32, height=32)
    O = orc.Oracle(sc, spp=1)
    dump = O.bvh_dump()
    prims = dump["packet_prims"][dump["packet_prims"] != 0xffffffff]
    assert len(prims) == sc.num_triangles and len(np.unique(prims)) == sc.num_triangles
    leaf = dump["flags"] == 1
    assert dump["num"][leaf].max() <= 255
    assert (dump["packet_num"] >= 1).all() and (dump["packet_num"] <= 8).all()
    info = O.bvh_info()
    assert info["triangles"] == sc.num_triangles


def test_scene_without_root_node_hits_nothing(orc):
    """SURVEY A-13: fewer than 8 triangles leave the reference BVH without a root."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(32, 32); sc.meshes = sc.meshes[:1] + sc.meshes[5:]
    O = orc.Oracle(sc, spp=1)
    assert O.bvh_info()["nodes"] == 0
    o, d, tm = random_rays(100, 1)
    assert not O.trace(o, d, tm)["hit"].any()
