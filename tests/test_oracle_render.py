"""Image-level checks of the oracle (CPU only): reference-order RNG vs counter RNG agree statistically,
thread count / tile subsets / literal-vs-conservative slab test do not change the counter-mode image,
and the committed regression film still reproduces."""
import os

import numpy as np
import pytest

from conftest import ROOT, bits_equal


def test_sequential_and_counter_modes_agree_statistically(orc):
    from phosphorus_mk2_amd import scenes
    O = orc.Oracle(scenes.cornell(64, 64), spp=64)
    a, sa = O.render(rng=orc.RNG_SEQ)
    b, sb = O.render(rng=orc.RNG_COUNTER, seed=11, threads=8)
    assert np.isfinite(a).all() and np.isfinite(b).all()
    # same estimator, different random numbers: block means agree to Monte-Carlo noise
    blk = lambda f: np.clip(f[..., :3], 0, 4).reshape(8, 8, 8, 8, 3).mean(axis=(1, 3))
    rel = np.abs(blk(a) - blk(b)) / (0.5 * (blk(a) + blk(b)) + 0.02)
    assert rel.mean() < 0.08
    for k in ("rays_closest", "rays_shadow", "rays_masked"):
        assert abs(sa[k] / sb[k] - 1) < 0.02


def test_counter_mode_is_independent_of_threads_tiles_and_slab_mode(orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(2000, width=96, height=80)
    O = orc.Oracle(sc, spp=4)
    a, sa = O.render(rng=orc.RNG_COUNTER, seed=5, threads=1)
    b, sb = O.render(rng=orc.RNG_COUNTER, seed=5, threads=8)
    assert bits_equal(a, b) and sa["rays_closest"] == sb["rays_closest"]
    c, _ = O.render(rng=orc.RNG_COUNTER, seed=5, threads=4, slab_literal=1)
    assert bits_equal(a, c)
    tiles = [(32, 0, 32, 32), (64, 64, 32, 16)]
    d, _ = O.render(rng=orc.RNG_COUNTER, seed=5, threads=2, tiles=tiles)
    for (x, y, w, h) in tiles:
        assert bits_equal(d[y:y + h, x:x + w], a[y:y + h, x:x + w])
    assert d[0:32, 0:32].max() == 0  # tiles not requested stay untouched
    e, _ = O.render(rng=orc.RNG_COUNTER, seed=6, threads=8)
    assert not bits_equal(a, e)


def test_sample_range_partitions_the_estimate(orc):
    """render(samples [0,2)) + render(samples [2,4)) == render(all 4) up to fp32 re-association."""
    from phosphorus_mk2_amd import scenes
    O = orc.Oracle(scenes.cornell(32, 32), spp=4)
    full, _ = O.render(rng=orc.RNG_COUNTER, seed=3, threads=4)
    lo, _ = O.render(rng=orc.RNG_COUNTER, seed=3, threads=4, sample_begin=0, sample_end=2)
    hi, _ = O.render(rng=orc.RNG_COUNTER, seed=3, threads=4, sample_begin=2, sample_end=4)
    assert np.allclose(lo + hi, full, rtol=1e-6, atol=1e-7)


def test_paths_per_sample_only_scales(orc):
    """SURVEY A-3: pps only multiplies the film by 1/pps (src/xpu/cpu.cpp:191)."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(32, 32)
    a, _ = orc.Oracle(sc, spp=4, pps=1).render(rng=orc.RNG_COUNTER, seed=3, threads=4)
    b, _ = orc.Oracle(sc, spp=4, pps=4).render(rng=orc.RNG_COUNTER, seed=3, threads=4)
    assert np.allclose(a / 4, b, rtol=1e-6)


def test_smooth_scene_sanity(orc):
    """interpolated normals are unit length, both lights contribute, sequential and counter modes agree"""
    from phosphorus_mk2_amd import scenes
    for pv in (True, False):
        sc = scenes.smooth_blobs(48, 32, per_vertex=pv)
        O = orc.Oracle(sc, spp=16)
        a, sa, n = O.render(rng=orc.RNG_COUNTER, seed=2, threads=8, normals=True)
        ln = np.linalg.norm(n, axis=-1)
        assert np.isfinite(a).all() and np.abs(ln[ln > 0] - 1).max() < 1e-5
        b, sb = O.render(rng=orc.RNG_SEQ)
        assert abs(np.clip(a[..., :3], 0, 4).mean() / np.clip(b[..., :3], 0, 4).mean() - 1) < 0.1
    pick = np.linspace(0.01, 0.99, 50).astype(np.float32)
    p, uv, pdf, mesh, face = O.light_sample(pick, np.full((50, 2), 0.5, np.float32))
    assert set(np.unique(mesh & 0xffff)) == {4, 5}  # both emissive meshes get picked


def test_regression_film(orc):
    """tests/golden/oracle_cornell_32x32_spp4.npy: written by tests/golden/make_oracle_pins.py from THIS
    oracle (a regression pin, not a reference-derived vector)."""
    from phosphorus_mk2_amd import scenes
    ref = np.load(os.path.join(ROOT, "tests", "golden", "oracle_cornell_32x32_spp4.npy"))
    a, _ = orc.Oracle(scenes.cornell(32, 32), spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=2)
    assert bits_equal(a[..., :3], ref)


@pytest.mark.parametrize("scene_name", ["cornell", "soup", "stress"])
def test_simd_and_scalar_restatements_agree(orc, scene_name):
    """obvh.h traces 8 child boxes / 8 triangles per AVX2 instruction by default; modes_t::scalar spells the same arithmetic
    out one lane at a time.  Films, ray counts, BVH visit counters and ray-dump traces must be identical, in the
    conservative and in the literal slab test."""
    from conftest import bits_equal, random_rays
    from phosphorus_mk2_amd import scenes
    sc = {"cornell": lambda: scenes.cornell(48, 48), "soup": lambda: scenes.soup(3000, width=64, height=48), "stress": lambda: scenes.stress(width=48, height=48)}[scene_name]()
    o, d, tm = random_rays(4000, 21)
    d[:50] = np.eye(3, dtype=np.float32)[np.arange(50) % 3] * np.float32(-1.0)  # axis-parallel rays: 0 * inf in the slab test
    out = {}
    try:
        for scalar in (1, 0):
            orc.set_scalar(scalar)
            O = orc.Oracle(sc, spp=3, pps=1, depth=6)
            for literal in (0, 1):
                img, st = O.render(rng=orc.RNG_COUNTER, seed=5, threads=2, slab_literal=literal)
                tr = O.trace(o, d, tm, slab_literal=literal)
                sh = O.trace(o, d, np.full(len(tm), 0.5, np.float32), shadow=True, slab_literal=literal)
                out[(scalar, literal)] = (img.copy(), {k: v for k, v in st.items() if k != "seconds"}, tr, sh)
    finally:
        orc.set_scalar(0)
    for literal in (0, 1):
        a, b = out[(1, literal)], out[(0, literal)]
        assert bits_equal(a[0], b[0]) and a[1] == b[1]
        for k in ("t", "u", "v"):
            assert bits_equal(a[2][k], b[2][k])
        assert np.array_equal(a[2]["prim"], b[2]["prim"]) and np.array_equal(a[3]["hit"], b[3]["hit"])
        assert a[2]["node_visits"] == b[2]["node_visits"] and a[2]["packet_visits"] == b[2]["packet_visits"]


def test_traversal_finds_the_hit_on_a_facet_edge_that_brute_force_finds(orc):
    """Round 6: ONE pixel of the closed showroom differed between device and oracle (L2 5e-5).  Replaying the pixel's rays stage by stage
    (scripts/pixel_replay_probe.py, profiles/r06_l_oracle_slab_miss.json) showed the DEVICE agreeing with the oracle's brute-force test of all
    triangles (t 1.4738580, primitive 427989) and the oracle's BVH traversal returning the neighbouring facet of the mirror sphere (t 1.4738613):
    the ray runs along the facets' shared edge, Moeller-Trumbore accepts it a few 1e-7 outside the nearer facet, and the "conservative" slab test —
    padded for ITS rounding only — rejected that facet's leaf.  The slab test now inflates the boxes for the triangle test's tolerance too
    (oracle/obvh.h: slab_inflation); this ray pins it, in both restatements."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.bmw_showroom(500_000, width=64, height=64)
    o = np.array([[1.9998998641967773, -0.019932806491851807, -4.015493392944336]], np.float32)
    d = np.array([[-0.40463271737098694, 0.10292842984199524, 0.9086683988571167]], np.float32)
    tm = np.array([np.finfo(np.float32).max], np.float32)
    orc.set_tie_rule(1)
    try:
        O = orc.Oracle(sc, spp=1)
        b = O.trace(o, d, tm, brute=True)
        for scalar in (0, 1):
            orc.set_scalar(scalar)
            try:
                t = O.trace(o, d, tm)
            finally:
                orc.set_scalar(0)
            assert int(t["prim"][0]) == int(b["prim"][0]) == 427989 and bits_equal(t["t"], b["t"]), (scalar, t["t"], t["prim"], b["t"], b["prim"])
        O.close()
    finally:
        orc.set_tie_rule(0)
