"""GPU parity tests: everything goes through the C ABI (libphx_hip.so) and is compared with the CPU
oracle on the same seeded inputs.  Bar: bit-exact for ids / flags / counts, and — because the device
and the oracle share one definition of every arithmetic step — bit-exact for fp32 values too; the
image gates additionally state the north-star tolerance (per-pixel L2 < 1e-4)."""
import numpy as np
import pytest

from conftest import aim_camera, bits_equal, max_pixel_l2, random_rays

pytestmark = pytest.mark.gpu

L2_TOL = 1e-4  # north_star: per-pixel L2 < 1e-4 vs CPU reference at fixed seed


@pytest.fixture(scope="module")
def xpu():
    from phosphorus_mk2_amd import xpu
    xpu.load_library()
    return xpu


def _device(xpu, scene, spp=4, depth=9):
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=depth))
    dev.preprocess(scene)
    return dev


@pytest.mark.parametrize("name,n", [("cornell", 0), ("soup", 3000), ("soup", 100000)])
def test_trace_matches_oracle(xpu, orc, name, n):
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(64, 64) if name == "cornell" else scenes.soup(n, width=64, height=64)
    dev = _device(xpu, sc)
    O = orc.Oracle(sc, spp=1)
    o, d, tm = random_rays(50000, 11)
    g = dev.trace(o, d, tm)
    r = O.trace(o, d, tm)  # MBVH-RS restatement on the reference-layout BVH
    assert np.array_equal(g["prim"], r["prim"])
    assert bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    nb = 3000
    rb = O.trace(o[:nb], d[:nb], tm[:nb], brute=True)  # linear_mbvh_kernel_t semantics
    assert np.array_equal(g["prim"][:nb], rb["prim"]) and bits_equal(g["t"][:nb], rb["t"])
    # any-hit (shadow) rays with finite length
    tm2 = np.full(len(tm), 0.7, np.float32)
    gs = dev.trace(o, d, tm2, shadow=True)
    rs = O.trace(o, d, tm2, shadow=True)
    assert np.array_equal(gs["hit"], rs["hit"])
    assert 0 < gs["hit"].sum() < len(tm)
    dev.close()


def test_trace_edge_cases(xpu, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(64, 64)
    dev = _device(xpu, sc)
    O = orc.Oracle(sc, spp=1)
    # axis-parallel directions (zero components), rays starting on surfaces, zero-length and huge tmax; the first ray runs
    # exactly into the shared edge (the diagonal) of the back wall's two triangles — the commonest tie of real meshes
    o = np.array([[0, 0, 0], [0.1, 0.3, 0], [0, 0, -2.5], [0.5, -1.0, -2.5], [0, 0, -2.5], [0, 0, -2.5], [0, 0.99, -2.5], [5, 5, 5]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 1, 0], [0, 1, 0], [1, 0, 0], [0, -1, 0], [0, -1, 0], [1, 0, 0]], np.float32)
    tm = np.array([3.4e38, 3.4e38, 3.4e38, 3.4e38, 0.0, 1e-6, 3.4e38, 3.4e38], np.float32)
    g = dev.trace(o, d, tm)
    orc.set_tie_rule(1)  # both triangles of the wall are hit at bitwise the same distance: the lower primitive index wins
    try:
        r = O.trace(o, d, tm, brute=True); rs = O.trace(o, d, tm)
    finally:
        orc.set_tie_rule(0)
    assert np.array_equal(g["prim"], r["prim"]) and bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    assert np.array_equal(g["prim"], rs["prim"]) and bits_equal(g["t"], rs["t"])
    assert g["prim"][0] == 4 and g["t"][0] == np.float32(3.5)  # back wall = mesh 2 = primitives 4 and 5; the tie goes to 4
    r0 = O.trace(o, d, tm, brute=True)  # the reference's rule (first met wins) may pick the other triangle of the tie, nothing else
    assert bits_equal(g["t"], r0["t"]) and np.array_equal(g["prim"][1:], r0["prim"][1:]) and r0["prim"][0] in (4, 5)
    assert dev.trace(o[:0], d[:0], tm[:0])["t"].shape == (0,)  # empty input
    dev.close()


def test_bsdf_known_answers(xpu, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.multi_material_soup(64, width=32, height=32)
    dev = _device(xpu, sc)
    O = orc.Oracle(sc, spp=1)
    rng = np.random.default_rng(5)
    k = 4096
    def unit(n):
        v = rng.normal(size=(n, 3)); return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    n, wi, wo = unit(k), unit(k), unit(k)
    # a few special configurations: normal-aligned, grazing, equal components (ONB second branch)
    n[:4] = [[0, 1, 0], [0, 0, 1], [0.57735026, 0.57735026, 0.57735026], [1, 0, 0]]
    wi[:2] = [[0, 1, 0], [0, 0, 1]]
    u2 = rng.random((k, 2)).astype(np.float32)
    u2[:3] = [[0, 0], [0.99999994, 0.99999994], [0.5, 0.5]]
    for m in range(len(sc.materials) - 1):
        fg = dev.bsdf_f(m, n, wi, wo); fo = O.bsdf_f(m, n, wi, wo)
        assert bits_equal(fg, fo), f"bsdf_f material {m}"
        wg, f2g, pg, flg = dev.bsdf_sample(m, n, wi, u2)
        wo_, f2o, po, flo = O.bsdf_sample(m, n, wi, u2)
        assert np.array_equal(flg, flo), f"sample flags material {m}"
        assert bits_equal(pg, po), f"sample pdf material {m}"
        assert bits_equal(wg, wo_) and bits_equal(f2g, f2o), f"sample wo/f material {m}"
    # the input tuple of the reference's scratch src/test.cpp:8-14 (rough refraction, eta 1.1)
    dev.close()


def test_sheen_is_nan_for_a_direction_on_the_normal_on_the_device_as_in_the_oracle(xpu, orc):
    """Where the non-finite pixels of the many-sample records come from (9 of 8.3 M at 4 096 spp, profiles/r06_nonfinite_probe_*.json): a direction
    that coincides with the shading normal has cos(theta) = n.n = 1 + 1 ulp for one unit normal in five to twenty (by how the normal was rounded), and the sheen lobe's
    Lambda = exp(2 L(0.5) - L(1 - cos theta)) (src/bsdf/sheen.hpp:51-64) raises the negative 1 - cos(theta) to a fractional power: NaN.  The
    reference has no guard; the restatement and the device reproduce it — the SAME inputs are NaN on both sides, everything else is bit-equal."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.multi_material_soup(64, width=32, height=32)
    dev = _device(xpu, sc)
    O = orc.Oracle(sc, spp=1)
    rng = np.random.default_rng(11)
    k = 8192
    n = rng.normal(size=(k, 3)); n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    wo = rng.normal(size=(k, 3)); wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    wo = np.where((wo * n).sum(1, keepdims=True) < 0, -wo, wo).astype(np.float32)
    for m in (6, 9):  # the sheen recipe and the three-lobe recipe that contains one (scenes.closure_zoo)
        fg, fo = dev.bsdf_f(m, n, n.copy(), wo), O.bsdf_f(m, n, n.copy(), wo)
        nan_g, nan_o = ~np.isfinite(fg).all(1), ~np.isfinite(fo).all(1)
        assert np.array_equal(nan_g, nan_o) and 0.01 < nan_g.mean() < 0.5, (m, nan_g.mean(), nan_o.mean())
        assert bits_equal(fg[~nan_g], fo[~nan_o]), m
    dev.close()


def _render_both(xpu, orc, sc, spp, seed, depth=9, threads=8, normals=False, **kw):
    film, st = xpu.render(sc, spp=spp, pps=1, depth=depth, seed=seed, normals=normals, **kw)
    O = orc.Oracle(sc, spp=spp, pps=1, depth=depth)
    res = O.render(rng=orc.RNG_COUNTER, seed=seed, threads=threads, normals=normals)
    return film, st, res


def test_render_cornell_matches_oracle(xpu, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(64, 64)
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=16, seed=3)
    assert st["camera_samples"] == ost["camera_samples"] == 64 * 64 * 16
    assert st["rays_closest"] == ost["rays_closest"]
    assert st["rays_shadow"] == ost["rays_shadow"]
    assert st["rays_masked"] == ost["rays_masked"]
    assert max_pixel_l2(film, ref) < L2_TOL
    assert bits_equal(film[..., :3], ref[..., :3])
    assert film[..., :3].max() > 0.1 and np.isfinite(film).all()


@pytest.mark.parametrize("spp,flight", [(1, 0), (3, 0), (16, 0), (48, 0), (256, 0), (96, 32), (40, 16)])
def test_camera_ray_packets_of_every_shape_match_oracle(xpu, orc, spp, flight):
    """k_trace_primary walks the camera rays of a pass as packets of 64 / 128 / 256 consecutive paths (1, 2 or 4 rays per lane, by the
    samples per pixel of the pass); a packet may be the samples of one pixel, span several pixels of a tile row, straddle two tiles,
    end in the middle of a wave, or hold rays of several direction octants (the camera looks down -z: the pixels on the film's axes),
    which sends it down the per-lane walk.  Every shape must give the oracle's film bit for bit, a ragged film included."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(3000, width=72, height=40)  # 32-pixel tiles: ragged on both axes
    kw = {"samples_in_flight": flight} if flight else {}
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=spp, seed=11, **kw)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert st["primary_launches"] == (1 if not flight else -(-spp // flight)) and st["primary_ms"] > 0
    assert bits_equal(film[..., :3], ref[..., :3])


@pytest.mark.parametrize("depth,pps,components", [(1, 1, 4), (2, 3, 4), (4, 1, 3)])
def test_depth_pps_and_channel_options_match_oracle(xpu, orc, depth, pps, components):
    """parsed_options_t: path_depth (spt.hpp:314 a path takes at most `depth` steps), paths_per_sample (only the 1/(spp*pps)
    film scale, cpu.cpp:191) and a 3-component primary channel, each against the oracle"""
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(48, 40)
    opts = xpu.Options(samples_per_pixel=5, paths_per_sample=pps, path_depth=depth, samples_in_flight=2)  # 3 passes: 2 + 2 + 1
    dev = xpu.HipDevice.make(opts)
    dev.preprocess(sc)
    film = xpu.Film(48, 40, components)
    dev.start(sc, xpu.FrameState(8, xpu.Tiles.make(48, 40, 32), film))
    dev.join()
    st = dev.stats()
    dev.close()
    # At this film size a few camera rays run exactly into the box's creases, where two triangles with different normals are hit
    # at the same distance: the comparison is exact under the device's tie rule, and under the reference's rule (first met in
    # ITS tree) only those few pixels may differ.
    orc.set_tie_rule(1)
    try:
        ref, ost = orc.Oracle(sc, spp=5, pps=pps, depth=depth).render(rng=orc.RNG_COUNTER, seed=8, threads=4)
    finally:
        orc.set_tie_rule(0)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert film.data.shape == (40, 48, components) and bits_equal(film.data[..., :3], ref[..., :3])
    ref0, _ = orc.Oracle(sc, spp=5, pps=pps, depth=depth).render(rng=orc.RNG_COUNTER, seed=8, threads=4)
    assert (film.data[..., :3] != ref0[..., :3]).any(-1).sum() <= 4


def test_render_edge_tiles_and_ragged_film(xpu, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(3000, width=96, height=80)  # 80 = 2*32 + 16: a 16-row edge band (SURVEY A-1/A-2)
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=4, seed=9)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert bits_equal(film[..., :3], ref[..., :3])


def test_render_all_closures_matches_oracle(xpu, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.multi_material_soup(4000, width=64, height=64)
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=16, seed=21)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    fin = np.isfinite(ref[..., :3]).all(axis=-1)
    assert np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1))
    assert max_pixel_l2(film[fin], ref[fin]) < L2_TOL
    assert bits_equal(film[..., :3][fin], ref[..., :3][fin])


def test_lambert_only_materials_with_several_lobes_match_oracle(xpu, orc):
    """k_shade has three material paths: any closure, Lambert lobes only, and at most ONE Lambert lobe per material (a 32-byte
    material record; the soups and the Cornell box).  This scene has Lambert-only materials with two and three lobes — lobe pick
    by floor(u * lobes), f summed over lobes, pdf averaged (bsdf.cpp:133-248) — so it runs the middle one."""
    from phosphorus_mk2_amd import abi, scenes
    D = abi.LOBE_DIFFUSE
    mats = [scenes.MaterialDesc([scenes.LobeDesc(D, (0.4, 0.3, 0.2)), scenes.LobeDesc(D, (0.2, 0.3, 0.4))]),
            scenes.MaterialDesc([scenes.LobeDesc(D, (0.3, 0.1, 0.1)), scenes.LobeDesc(D, (0.1, 0.3, 0.1)), scenes.LobeDesc(D, (0.1, 0.1, 0.3))]),
            scenes.diffuse(0.73, 0.73, 0.73)]
    sc = scenes.soup(4000, width=96, height=64, materials=mats)
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=16, seed=9)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"]
    assert max_pixel_l2(film, ref) < L2_TOL and bits_equal(film[..., :3], ref[..., :3])
    assert film[..., :3].max() > 0.05


def test_glass_per_hit_closures_match_oracle(xpu, orc):
    """Blender's glass node (mix(refraction, glossy, fresnel_dielectric(I.N, backfacing ? 1/IoR : IoR)),
    plugins/blender/blender/shader.hpp:306-335): closure weights evaluated at every hit (bsdf.h: material_at_hit), closures
    with an all-zero weight dropped (total internal reflection leaves ONE lobe).  Known answers and a film, bit for bit."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.glass_blobs(160, 96)
    dev = _device(xpu, sc)
    O = orc.Oracle(sc, spp=1)
    rng = np.random.default_rng(9)
    k = 4096
    def unit(m):
        v = rng.normal(size=(m, 3)); return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    n, wi, wo = unit(k), unit(k), unit(k)
    n[:2] = [[0, 1, 0], [0, 1, 0]]; wi[:2] = [[0.9, -0.43588990, 0.0], [0.0, 1.0, 0.0]]  # total internal reflection from inside; normal incidence
    u2 = rng.random((k, 2)).astype(np.float32)
    for m in (1, 2):  # sharp glass, frosted glass
        assert bits_equal(dev.bsdf_f(m, n, wi, wo), O.bsdf_f(m, n, wi, wo)), f"glass f, material {m}"
        wg, fg, pg, flg = dev.bsdf_sample(m, n, wi, u2); w0, f0, p0, fl0 = O.bsdf_sample(m, n, wi, u2)
        assert np.array_equal(flg, fl0) and bits_equal(pg, p0) and bits_equal(wg, w0) and bits_equal(fg, f0), f"glass sample, material {m}"
    dev.close()
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=16, seed=5)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"]
    fin = np.isfinite(ref[..., :3]).all(axis=-1)
    assert fin.mean() > 0.99 and np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1))
    assert max_pixel_l2(film[fin], ref[fin]) < L2_TOL and bits_equal(film[..., :3][fin], ref[..., :3][fin])
    # the glass changes the picture: the same scene with Lambert blobs is a different film
    plain, _ = xpu.render(scenes.smooth_blobs(160, 96), spp=16, seed=5)
    assert not bits_equal(plain, film)


def test_general_closures_at_film_size(xpu, orc):
    """the general k_shade (all seven lobe models, 16 closure recipes: the declared stand-in of the BMW configs) on a 640x360
    film with an edge band (360 = 11 * 32 + 8), whole frame against the oracle"""
    from phosphorus_mk2_amd import scenes
    sc = scenes.multi_material_soup(20000, width=640, height=360)
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=4, seed=17, threads=16)
    assert st["camera_samples"] == 640 * 360 * 4
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"]
    fin = np.isfinite(ref[..., :3]).all(axis=-1)
    assert fin.mean() > 0.999 and np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1))
    assert max_pixel_l2(film[fin], ref[fin]) < L2_TOL
    assert bits_equal(film[..., :3][fin], ref[..., :3][fin])


@pytest.mark.parametrize("per_vertex", [True, False])
def test_smooth_normals_and_several_lights(xpu, orc, per_vertex):
    """interpolated normals (per vertex / per face corner, mesh.cpp:187-199), two area lights of 2 and 4
    triangles (light + triangle pick by index), a two-lobe material: the general (non diffuse-only) k_shade."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.smooth_blobs(per_vertex=per_vertex)
    film, st, (ref, ost, nref) = _render_both(xpu, orc, sc, spp=9, seed=13, normals=True)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert max_pixel_l2(film, ref) < L2_TOL
    assert bits_equal(film[..., :3], ref[..., :3]) and bits_equal(film[..., 4:7], nref)
    assert np.abs(np.linalg.norm(nref, axis=-1)[nref.any(axis=-1)] - 1).max() < 1e-5


def test_normals_channel_and_env_light(xpu, orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(64, 64)
    sc.materials.append(scenes.MaterialDesc(lobes=[], emission=(0.3, 0.4, 0.5)))  # background closure
    sc.environment_material = len(sc.materials) - 1
    sc.meshes = sc.meshes[:1] + sc.meshes[2:]  # no ceiling: paths escape to the environment (>= 8 triangles: SURVEY A-13)
    film, st, (ref, ost, nref) = _render_both(xpu, orc, sc, spp=4, seed=2, normals=True)
    assert bits_equal(film[..., :3], ref[..., :3])
    assert bits_equal(film[..., 4:7], nref)
    assert np.allclose(film[0, 0, :3], (0.3, 0.4, 0.5))  # the corner pixel looks past the box


def test_invariances(xpu):
    """size-independent properties: samples in flight, tile batching, callback vs native queue,
    rank sharding all leave the film bit-identical; a different seed does not."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(5000, width=128, height=96)
    base, st = xpu.render(sc, spp=9, seed=4)
    a, _ = xpu.render(sc, spp=9, seed=4, samples_in_flight=1)
    b, _ = xpu.render(sc, spp=9, seed=4, samples_in_flight=4, tiles_per_batch=5)
    c, _ = xpu.render(sc, spp=9, seed=4, callback_tiles=True)
    d, _ = xpu.render(sc, spp=9, seed=4, native_sink=True)  # phx_frame.host_film instead of add_tile callbacks
    assert bits_equal(base, a) and bits_equal(base, b) and bits_equal(base, c) and bits_equal(base, d)
    r0, _ = xpu.render(sc, spp=9, seed=4, rank=0, world=2)
    r1, _ = xpu.render(sc, spp=9, seed=4, rank=1, world=2)
    assert bits_equal(r0 + r1, base)  # disjoint tiles: the film reduce is exact
    assert (r0[..., :3].sum(axis=-1) > 0).sum() > 0 and (r1[..., :3].sum(axis=-1) > 0).sum() > 0
    other, _ = xpu.render(sc, spp=9, seed=5)
    assert not bits_equal(base, other)


@pytest.mark.parametrize("name,n", [("cornell", 0), ("soup", 17), ("soup", 100000), ("blobs", 0)])
def test_device_built_bvh_matches_host_built(xpu, orc, name, n):
    """phx_options.bvh_builder = PHX_BVH_DEVICE_LBVH: the tree is built on the GPU (bvh_gpu.hip).  The box
    tests are conservative, so hits, ray counts and the film must not depend on which builder made the tree."""
    from phosphorus_mk2_amd import scenes
    sc = {"cornell": lambda: scenes.cornell(64, 64), "soup": lambda: scenes.soup(n, width=64, height=64),
          "blobs": lambda: scenes.smooth_blobs(64, 64)}[name]()
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4, paths_per_sample=1, bvh_builder="device"))
    dev.preprocess(sc)
    O = orc.Oracle(sc, spp=1)
    o, d, tm = random_rays(40000, 23)
    g = dev.trace(o, d, tm)
    r = O.trace(o, d, tm)
    assert np.array_equal(g["prim"], r["prim"])
    assert bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    tm2 = np.full(len(tm), 0.7, np.float32)
    assert np.array_equal(dev.trace(o, d, tm2, shadow=True)["hit"], O.trace(o, d, tm2, shadow=True)["hit"])
    st = dev.stats()
    assert st["bvh_nodes"] > 0 and st["triangles"] == O.bvh_info()["triangles"] and st["bvh_build_ms"] > 0
    dev.close()
    host, hst = xpu.render(sc, spp=8, seed=6)
    devb, dst = xpu.render(sc, spp=8, seed=6, bvh_builder="device")
    assert bits_equal(host, devb)
    for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked"):
        assert hst[k] == dst[k]
    ref, _ = orc.Oracle(sc, spp=8, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=6, threads=8)
    assert max_pixel_l2(devb, ref) < L2_TOL and bits_equal(devb[..., :3], ref[..., :3])


def _check_bvh_containment(pool, glo, gcell):
    """walk the 8-wide tree breadth first (numpy, a level at a time), then bottom-up: the box a nodelet stores for a child must contain
    every TRIANGLE that hangs below that child (a child's own stored boxes live on the child's grid and may stick out of the parent's
    box by a grid unit of the child: the invariant is about the geometry).  -> (nodelets, triangles, deepest level)"""
    u8 = pool.view(np.uint8).reshape(-1, 64)
    n_el = len(pool)
    true_lo = np.full((n_el, 3), np.inf); true_hi = np.full((n_el, 3), -np.inf)
    slot = np.arange(8)
    levels = []
    level = np.array([0], np.int64)
    nodes = tris = 0
    while len(level):
        w = pool[level].astype(np.uint64)
        ix = w[:, 0] & 0x3ffff; iy = ((w[:, 0] >> 18) | (w[:, 1] << 14)) & 0x3ffff; iz = (w[:, 1] >> 4) & 0x3ffff
        valid = ((w[:, 1] >> 22) & 0xff).astype(np.uint32); imask = (w[:, 2] >> 24).astype(np.uint32); base = w[:, 3].astype(np.int64)
        org = np.stack([ix, iy, iz], 1).astype(np.float64) * gcell.astype(np.float64) + glo.astype(np.float64)  # fma(i, cell, lo) up to one rounding
        org = org.astype(np.float32).astype(np.float64)
        e = np.stack([w[:, 2] & 0xff, (w[:, 2] >> 8) & 0xff, (w[:, 2] >> 16) & 0xff], 1).astype(np.int64)
        scale = np.ldexp(1.0, e - 127)
        q = u8[level][:, 16:].reshape(-1, 6, 8).astype(np.float64)  # rows: lox loy loz hix hiy hiz, columns: slots
        lo = org[:, :, None] + q[:, 0:3, :] * scale[:, :, None]; hi = org[:, :, None] + q[:, 3:6, :] * scale[:, :, None]  # [n, axis, slot]
        vbit = ((valid[:, None] >> slot) & 1).astype(bool); ibit = ((imask[:, None] >> slot) & 1).astype(bool)
        assert not (ibit & ~vbit).any()
        child = base[:, None] + np.cumsum(vbit, 1) - vbit
        assert child[vbit].max() < n_el
        tn, ts = np.nonzero(vbit & ~ibit)
        if len(tn):  # triangle records: a, a + e0, a + e1 (the builder's fp32 v0, e0, e1: b and c up to a rounding)
            ti = child[tn, ts]
            rec = pool[ti].view(np.float32).astype(np.float64)
            a3 = rec[:, 0:3]; b3 = a3 + rec[:, 3:6]; c3 = a3 + rec[:, 6:9]
            true_lo[ti] = np.minimum(np.minimum(a3, b3), c3); true_hi[ti] = np.maximum(np.maximum(a3, b3), c3)
            tris += len(tn)
        nodes += len(level)
        levels.append((level, child, vbit, lo, hi))
        cn, cs = np.nonzero(ibit)
        level = child[cn, cs]
    for level, child, vbit, lo, hi in reversed(levels):  # bottom-up: true bounds of every subtree, checked against the stored boxes
        cl = np.where(vbit[:, :, None], true_lo[np.where(vbit, child, 0)], np.inf); ch = np.where(vbit[:, :, None], true_hi[np.where(vbit, child, 0)], -np.inf)  # [n, slot, axis]
        assert np.isfinite(cl[vbit]).all() and np.isfinite(ch[vbit]).all()
        eps = 1e-5 * (1.0 + np.abs(cl) + np.abs(ch))
        slo = np.transpose(lo, (0, 2, 1)); shi = np.transpose(hi, (0, 2, 1))
        ok = (slo <= cl + eps) & (shi >= ch - eps)
        assert ok[vbit].all(), "geometry sticks out of the box its ancestor stores for it"
        true_lo[level] = cl.min(1); true_hi[level] = ch.max(1)
    return nodes, tris, len(levels)


@pytest.mark.parametrize("name,n", [("soup", 1000000), ("showroom", 200000)])
def test_device_built_tree_boxes_contain_their_subtrees(xpu, name, n):
    """the device builder's bottom-up passes (k_fit, k_collapse_dp: chains hand boxes and costs to one another inside ONE launch through
    agent-scope stores / loads and an arrival counter, bvh_gpu.hip) under real load: in the finished 8-wide tree every stored box must
    contain the whole subtree below it — a box read before its sibling chain had written it would not.  Read back through
    phx_dev_copy_bvh and walked on the host."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(n, width=64, height=64) if name == "soup" else scenes.showroom(n, width=64, height=64)
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=1, paths_per_sample=1, bvh_builder="device"))
    for _ in range(2):  # twice: the hand-offs are timing dependent
        dev.preprocess(sc)
        pool, glo, gcell = dev.bvh_pool()
        st = dev.stats()
        nodes, tris, depth = _check_bvh_containment(pool, glo, gcell)
        assert nodes == st["bvh_nodes"] and tris == st["triangles"] and depth == st["bvh_depth"] and len(pool) == nodes + tris
    dev.close()


def test_device_builder_handoff_under_load(xpu):
    """The bottom-up passes of the device builder (k_fit, k_collapse_dp) hand boxes and sub-costs from chain to chain inside one
    launch with agent-scope stores / loads and no cache maintenance (bvh_gpu.hip: store_handoff / load_handoff).  A stale read there
    would change a box or a cut, hence the tree: the same scene built over and over — uneven trees, other work on the GPU in between
    — must give the same node count, depth and modelled cost every time, and the film of the first build."""
    from phosphorus_mk2_amd import scenes
    for sc in (scenes.showroom(300000, width=96, height=64), scenes.soup(400000, width=96, height=64)):
        seen, films = set(), []
        for rep in range(10):
            dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=2, paths_per_sample=1, bvh_builder="device"))
            dev.preprocess(sc)
            st = dev.stats()
            seen.add((st["bvh_nodes"], st["bvh_depth"], st["bvh_cost_model"], st["bvh_bytes"]))
            if rep in (0, 9):
                tiles = xpu.Tiles.make(96, 64, 32); film = xpu.Film(96, 64, 4)
                dev.start(sc, xpu.FrameState(3, tiles, film, native_sink=True)); dev.join()
                films.append(film.data.copy())
            dev.close()
        assert len(seen) == 1, seen
        assert bits_equal(films[0], films[1])


def test_device_builder_handoff_beside_a_running_render(xpu):
    """The same hand-offs under UNEVEN load, caches warm with other work: while another device object on the same GPU renders frame
    after frame (its own stream), the builder makes the same tree every time.  (The guide: test a hand-off under uneven load — idle
    chips and cold caches hide stale reads.)"""
    import threading
    from phosphorus_mk2_amd import scenes
    busy = scenes.soup(60000, width=256, height=192)
    worker = xpu.HipDevice.make(xpu.Options(samples_per_pixel=16, paths_per_sample=1))
    worker.preprocess(busy)
    stop = threading.Event(); frames = [0]

    def render_loop():
        tiles = xpu.Tiles.make(256, 192, 32); film = xpu.Film(256, 192, 4)
        while not stop.is_set():
            tiles.reset()
            worker.start(busy, xpu.FrameState(1, tiles, film, native_sink=True)); worker.join()
            frames[0] += 1

    t = threading.Thread(target=render_loop); t.start()
    try:
        sc = scenes.showroom(500000, width=64, height=64)
        seen = set()
        for rep in range(12):
            dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=1, paths_per_sample=1, bvh_builder="device"))
            dev.preprocess(sc)
            st = dev.stats()
            seen.add((st["bvh_nodes"], st["bvh_depth"], st["bvh_cost_model"], st["bvh_bytes"]))
            dev.close()
    finally:
        stop.set(); t.join(); worker.close()
    assert len(seen) == 1, seen
    assert frames[0] >= 2  # the render really ran beside the builds


@pytest.mark.parametrize("builder", ["host", "device"])
def test_stress_geometry(xpu, orc, builder):
    """zero-area, coincident (exact distance ties), 2^-20-sized, 1e4-sized and flat triangles: both builders must give the
    oracle's distances bit for bit, ties resolved to the lowest primitive index, and the oracle's film"""
    from phosphorus_mk2_amd import scenes
    from test_host_bvh8 import check_hits_modulo_ties, stress_rays
    sc = scenes.stress()
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4, paths_per_sample=1, bvh_builder=builder))
    dev.preprocess(sc)
    O = orc.Oracle(sc, spp=1)
    o, d, tm = stress_rays(sc, 8000, 5)
    g = dev.trace(o, d, tm)
    check_hits_modulo_ties(g, O.trace(o, d, tm, brute=True), min_ties=100)
    orc.set_tie_rule(1)  # with the device's tie rule the oracle agrees on the primitive as well, brute force and stream traversal
    try:
        for brute in (True, False):
            r = O.trace(o, d, tm, brute=brute)
            assert np.array_equal(g["prim"], r["prim"]) and bits_equal(g["t"], r["t"]) and bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    finally:
        orc.set_tie_rule(0)
    tm2 = np.full(len(tm), 0.6, np.float32)
    assert np.array_equal(dev.trace(o, d, tm2, shadow=True)["hit"], O.trace(o, d, tm2, shadow=True, brute=True)["hit"])
    dev.close()
    film, st = xpu.render(sc, spp=8, seed=2, bvh_builder=builder)
    ref, ost = orc.Oracle(sc, spp=8, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=2, threads=8)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert max_pixel_l2(film, ref) < L2_TOL and bits_equal(film[..., :3], ref[..., :3])


@pytest.mark.parametrize("builder", ["host", "device"])
def test_showroom_meshes_match_oracle(xpu, orc, builder):
    """connected, indexed meshes with shared vertices (every ray that leaves a sphere's surface starts ON an edge or vertex of
    its neighbours), zero-area pole slivers, three orders of magnitude of triangle size, inside a closed room: closest and
    shadow rays bit for bit against brute force, then the film — with glass and glossy spheres — against the oracle"""
    from phosphorus_mk2_amd import scenes
    from test_host_bvh8 import stress_rays
    sc = scenes.showroom(20000, width=160, height=96, materials=[scenes.diffuse(0.6, 0.3, 0.2), scenes.glass(1.45), scenes.closure_zoo()[4]])
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4, paths_per_sample=1, bvh_builder=builder))
    dev.preprocess(sc)
    O = orc.Oracle(sc, spp=1)
    o, d, tm = stress_rays(sc, 6000, 11)
    g = dev.trace(o, d, tm)
    orc.set_tie_rule(1)
    try:
        r = O.trace(o, d, tm, brute=True)
    finally:
        orc.set_tie_rule(0)
    assert g["hit"].sum() > 3000
    assert np.array_equal(g["hit"], r["hit"]) and np.array_equal(g["prim"], r["prim"]) and bits_equal(g["t"], r["t"])
    assert bits_equal(g["u"], r["u"]) and bits_equal(g["v"], r["v"])
    tm2 = np.full(len(tm), 0.8, np.float32)
    assert np.array_equal(dev.trace(o, d, tm2, shadow=True)["hit"], O.trace(o, d, tm2, shadow=True, brute=True)["hit"])
    dev.close()
    film, st = xpu.render(sc, spp=8, seed=6, bvh_builder=builder)
    ref, ost = orc.Oracle(sc, spp=8, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=6, threads=8)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"]
    assert max_pixel_l2(film, ref) < L2_TOL and bits_equal(film[..., :3], ref[..., :3])


def _room_scene():
    """the scene examples/render_room.c builds in C, through the Python mirror"""
    from phosphorus_mk2_amd import abi, scenes
    v = np.array([[-2, -1, -1], [2, -1, -1], [2, -1, -5], [-2, -1, -5], [-2, -1, -5], [2, -1, -5], [2, 2, -5], [-2, 2, -5],
                  [-1, 1.5, -2], [-1, 1.5, -4], [1, 1.5, -4], [1, 1.5, -2], [-0.8, -0.9, -3.2], [0.9, -0.6, -3.6], [0.1, 0.7, -3.0],
                  [-1.5, -0.2, -4.2], [-0.6, 0.9, -4.4], [-1.7, 1.1, -3.9]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11], [12, 13, 14], [15, 16, 17]], np.uint32)
    mats = [scenes.MaterialDesc([scenes.LobeDesc(abi.LOBE_DIFFUSE, (0.73, 0.73, 0.73))]),
            scenes.MaterialDesc([], emission=(17.0, 12.0, 4.0), is_emitter=True),
            scenes.MaterialDesc([scenes.LobeDesc(abi.LOBE_DIFFUSE, (0.63, 0.065, 0.05))])]
    mesh = scenes.MeshDesc(v, f, [(0, [0, 1, 2, 3]), (1, [4, 5]), (2, [6, 7])])
    return scenes.SceneDesc([mesh], mats, scenes.CameraDesc(128, 96, fov=1.2))


@pytest.mark.parametrize("builder,ndev", [("host", 1), ("device", 1), ("host", 3)])
def test_plain_c_host_renders_the_same_film(xpu, orc, builder, ndev, tmp_path):
    """examples/render_room.c drives the C ABI from C (no Python, no torch in that process); its film must equal the
    film of the same scene rendered through the ctypes mirror, and the oracle's."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "render_room")
    if not os.path.exists(exe):
        subprocess.run(["make", "-B", "-C", os.path.join(root, "examples")], check=True)  # -B: a stale binary built against an older header may have travelled with the snapshot
    out = str(tmp_path / "room.f32")
    # ndev > 1: the C host makes that many devices (ordinal i % GPUs of the box), starts them all on ONE queue and ONE film
    r = subprocess.run([exe, out, f"{builder}-bvh"] + ([str(ndev)] if ndev > 1 else []), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    if ndev > 1:
        per_dev = [int(l.split("tiles")[1]) for l in r.stdout.splitlines() if l.startswith("device ")]
        assert len(per_dev) == ndev and sum(per_dev) == 12  # 128x96 film: 4 x 3 tiles, every tile rendered exactly once
    c_film = np.fromfile(out, np.float32).reshape(96, 128, 4)
    sc = _room_scene()
    film, st = xpu.render(sc, spp=8, pps=1, depth=5, seed=7, native_sink=True, bvh_builder=builder)
    assert bits_equal(c_film, film)
    assert f"rays {st['rays_closest']}+{st['rays_shadow']}" in r.stdout
    ref, _ = orc.Oracle(sc, spp=8, pps=1, depth=5).render(rng=orc.RNG_COUNTER, seed=7, threads=4)
    assert max_pixel_l2(c_film, ref) < L2_TOL and bits_equal(c_film[..., :3], ref[..., :3])
    assert c_film[..., :3].max() > 0.05


def test_loaded_yaml_obj_scene_matches_oracle(xpu, orc):
    """scene ingestion -> preprocess -> render: baked shader graphs (glossy GGX, emitter, background), OBJ mesh with
    per-face-corner normals, look-at camera, environment material"""
    import os
    from conftest import ROOT
    from phosphorus_mk2_amd import sceneio
    sc = sceneio.load_scene(os.path.join(ROOT, "tests", "golden", "room", "scene.yaml"))
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=16, seed=5)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert bits_equal(film[..., :3], ref[..., :3]) and film[..., :3].max() > 0.1


def test_hybrid_cpu_gpu_share_one_tile_queue(xpu, orc):
    """The reference's hybrid mechanism: several xpu_t devices drain ONE job::tiles_t (src/core.cpp:103-108).
    Here the gfx950 device and a CPU worker (the oracle, standing in for cpu_t) pull from the same queue;
    whoever renders a tile, the film is the same bit for bit."""
    import threading
    from phosphorus_mk2_amd import dist, scenes
    sc = scenes.cornell(160, 128)
    all_tiles = dist.shard_tiles(160, 128, 32, 0, 1)
    queue = xpu.CallbackTiles(all_tiles)
    film = xpu.Film(160, 128, 4)
    O = orc.Oracle(sc, spp=4)
    cpu_tiles = []

    def cpu_device():
        while True:
            t = queue.next()
            if t is None:
                return
            part, _ = O.render(rng=orc.RNG_COUNTER, seed=6, threads=1, tiles=[t])
            x, y, w, h = t
            with film._lock:
                film.data[y:y + h, x:x + w, :] = part[y:y + h, x:x + w, :]
            cpu_tiles.append(t)

    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4, paths_per_sample=1, tiles_per_batch=2))
    dev.preprocess(sc)
    worker = threading.Thread(target=cpu_device)
    dev.start(sc, xpu.FrameState(6, queue, film))
    worker.start()
    dev.join(); worker.join()
    gpu_tiles = dev.stats()["tiles"]
    assert gpu_tiles + len(cpu_tiles) == len(all_tiles) and gpu_tiles > 0
    ref, _ = O.render(rng=orc.RNG_COUNTER, seed=6, threads=4)
    assert bits_equal(film.data[..., :3], ref[..., :3])
    dev.close()


def test_two_devices_drain_one_tile_queue(xpu, orc):
    """The reference's multi-device mechanism (src/core.cpp:103-115): every xpu_t of discover() drains the SAME job::tiles_t
    and adds its tiles to the SAME film.  Two phx_device objects (both on this box's one GPU, ordinal 0) share one native
    queue and one film; whoever renders a tile, the film equals the one-device film and the oracle's, bit for bit."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(5000, width=320, height=208)  # 70 tiles, a 16-row edge band
    one, st1 = xpu.render(sc, spp=9, seed=4)
    opts = xpu.Options(samples_per_pixel=9, paths_per_sample=1, tiles_per_batch=3, device_ordinal=0)
    devs = [xpu.HipDevice.make(opts), xpu.HipDevice.make(opts)]
    for dv in devs:
        dv.preprocess(sc)
    for sink in ("callback", "native"):
        film, sts = xpu.render_on(devs, sc, seed=4, native_sink=(sink == "native"))
        assert sum(s["tiles"] for s in sts) == 70
        for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked"):
            assert sum(s[k] for s in sts) == st1[k]
        assert bits_equal(film, one)
    # discover(): one device object per GPU of the box, each on its own ordinal
    found = xpu.HipDevice.discover(xpu.Options(samples_per_pixel=9, paths_per_sample=1))
    assert len(found) >= 1 and all(isinstance(d, xpu.HipDevice) for d in found)
    for d in found:
        d.preprocess(sc)
    film, sts = xpu.render_on(found, sc, seed=4)
    assert bits_equal(film, one) and sum(s["tiles"] for s in sts) == 70
    for d in found:
        d.close()
    for dv in devs:
        dv.close()
    ref, _ = orc.Oracle(sc, spp=9).render(rng=orc.RNG_COUNTER, seed=4, threads=8)
    assert bits_equal(one[..., :3], ref[..., :3])


def _random_scene(seed):
    """a scene nobody designed: 2-4 meshes (indexed vertices shared between faces, smooth and flat faces mixed, per-vertex or
    per-corner normals), 3-6 materials drawn from the closure zoo and the glass node, 1-2 emissive face sets, an optional
    environment, a camera that is not the identity — on a third of the scenes with a thin lens —, a film whose width and height are not multiples of
    the tile size"""
    from phosphorus_mk2_amd import abi, scenes
    rng = np.random.default_rng(seed)
    zoo = scenes.closure_zoo()
    mats = [zoo[int(k)] for k in rng.choice(len(zoo), int(rng.integers(2, 5)), replace=False)]
    if rng.random() < 0.7:
        mats.append(scenes.glass(float(rng.uniform(1.2, 1.8)), float(rng.choice([0.0, 0.15])), tuple(rng.uniform(0.8, 1.0, 3)), (1.0, 1.0, 1.0)))
    n_surface = len(mats)
    mats.append(scenes.emitter(*scenes.LE)); mats.append(scenes.emitter(3.0, 4.0, 5.0))
    meshes = []
    for mi in range(int(rng.integers(2, 5))):
        g = int(rng.integers(3, 9))  # a g x g grid of vertices bent into a bumpy sheet: every inner vertex is shared by six faces
        u, v = np.meshgrid(np.linspace(-1, 1, g), np.linspace(-1, 1, g))
        c = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(-3.2, -1.8)])
        ax = rng.normal(size=(3, 3)); ax, _ = np.linalg.qr(ax)
        p = (u[..., None] * ax[0] + v[..., None] * ax[1]) * rng.uniform(0.3, 0.9) + ax[2] * 0.15 * np.sin(3 * u + mi)[..., None] * np.cos(2 * v)[..., None] + c
        verts = p.reshape(-1, 3).astype(np.float32)
        idx = np.arange(g * g).reshape(g, g)
        f = np.concatenate([np.stack([idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:]], -1).reshape(-1, 3), np.stack([idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 3)]).astype(np.uint32)
        nrm = np.zeros_like(verts)
        fn = np.cross(verts[f[:, 1]] - verts[f[:, 0]], verts[f[:, 2]] - verts[f[:, 0]])
        for k in range(3):
            np.add.at(nrm, f[:, k], fn)
        nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)).astype(np.float32)
        smooth = (rng.random(len(f)) < 0.6).astype(np.uint8)
        order = rng.permutation(len(f)); cut = int(rng.integers(1, len(f)))
        sets = [(int(rng.integers(0, n_surface)), order[:cut].astype(np.uint32)), (int(rng.integers(0, n_surface)), order[cut:].astype(np.uint32))]
        if rng.random() < 0.5:
            meshes.append(scenes.MeshDesc(verts, f, sets, normals=nrm, smooth=smooth))
        else:  # one normal per face corner (mesh.cpp:188-192)
            meshes.append(scenes.MeshDesc(verts, f, sets, normals=nrm[f.reshape(-1)], smooth=smooth, flags=abi.MESH_UV_PER_VERTEX))
    meshes.append(scenes._quad((-1.5, 1.6, -1.5), (-1.5, 1.6, -3.5), (1.5, 1.6, -3.5), (1.5, 1.6, -1.5), n_surface))
    if rng.random() < 0.5:
        meshes.append(scenes._quad((-1.9, -0.5, -1.5), (-1.9, 0.5, -1.5), (-1.9, 0.5, -2.5), (-1.9, -0.5, -2.5), n_surface + 1))
    sc = scenes.SceneDesc(meshes, mats, scenes.CameraDesc(int(rng.integers(5, 14)) * 8, int(rng.integers(30, 90)), fov=float(rng.uniform(0.9, 1.9))))
    if rng.random() < 0.5:
        sc.materials.append(scenes.MaterialDesc(lobes=[], emission=(0.2, 0.25, 0.3))); sc.environment_material = len(sc.materials) - 1
    th = rng.uniform(-0.3, 0.3); M = np.eye(4, dtype=np.float32)
    M[0, 0] = np.cos(th); M[0, 2] = -np.sin(th); M[2, 0] = np.sin(th); M[2, 2] = np.cos(th); M[3, :3] = rng.uniform(-0.2, 0.2, 3)  # row-vector convention
    sc.camera.to_world = M
    if rng.random() < 0.35:  # a thin lens on a third of the scenes (drawn last: the scenes of earlier rounds keep their geometry)
        sc.camera.aperture_radius, sc.camera.focal_distance = float(rng.uniform(0.005, 0.08)), float(rng.uniform(1.5, 3.5))
    return sc


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_match_oracle(xpu, orc, seed):
    """eight scenes nobody designed (shared vertices, mixed smooth / flat faces, zoo + glass materials, one or two lights, optional
    environment, rotated and shifted camera, ragged film), both builders, odd spp and depth: counts and film bit for bit"""
    sc = _random_scene(1000 + seed)
    spp, depth = [1, 4, 9, 16][seed % 4], [9, 3, 6, 9][(seed // 2) % 4]
    builder = "device" if seed % 2 else "host"
    film, st = xpu.render(sc, spp=spp, pps=1, depth=depth, seed=seed, normals=True, bvh_builder=builder, samples_in_flight=[0, 3][seed % 2])
    ref, ost, nrm = orc.Oracle(sc, spp=spp, pps=1, depth=depth).render(rng=orc.RNG_COUNTER, seed=seed, threads=8, normals=True)
    for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    fin = np.isfinite(ref[..., :3]).all(axis=-1)
    assert fin.mean() > 0.98 and np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1))
    assert max_pixel_l2(film[fin], ref[fin]) < L2_TOL and bits_equal(film[..., :3][fin], ref[..., :3][fin])
    assert bits_equal(film[..., 4:7], nrm)


def test_differential_fuzz_of_sixty_random_scenes(xpu):
    """scripts/fuzz_parity.py (3 000 scenes in profiles/r02_fuzz_parity_3000.json) on sixty fresh seeds: random scenes x random
    options, ray counts + film + normals bit for bit against the oracle under the device's tie rule"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "60", "70000"], capture_output=True, text=True, timeout=600)
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and out["failed"] == 0 and out["scenes"] == 60, (out, p.stderr[-2000:])


def test_instrumented_build_counts_the_same_frame(xpu, tmp_path):
    """libphx_hip_count.so (the same sources with -DPHX_COUNT=1: bench.py's roofline reads node visits and triangle tests from it)
    must render the very same film, and its counters must add up: every ray visits the root, every wave iteration runs a block."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    lib = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_count.so")
    if not os.path.exists(lib):
        pytest.skip("instrumented variant not built (make -C phosphorus_mk2_amd/csrc variant NAME=count EXTRA=-DPHX_COUNT=1)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "count_work.py"), "--triangles", "20000", "--width", "320", "--height", "192", "--spp", "16"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    w = json.loads(r.stdout.strip().splitlines()[-1])
    from phosphorus_mk2_amd import scenes
    import hashlib
    film, st = xpu.render(scenes.soup(20000, width=320, height=192), spp=16, seed=1)
    assert hashlib.sha1(film.tobytes()).hexdigest() == w["film_sha1"]  # same film, bit for bit
    assert st["instrumented"] == 0 and st["node_visits_mem"] == [0, 0]  # the product library counts nothing
    assert w["closest"]["rays"] + w["primary"]["rays"] == st["rays_closest"] and w["shadow"]["rays"] == st["rays_shadow"]
    assert w["primary"]["rays"] == st["camera_samples"] and w["primary"]["rays"] <= w["primary"]["packets"] * 256 < 8 * w["primary"]["rays"]  # camera rays: k_trace_primary, 64 / 128 / 256 per packet
    assert 1.0 <= w["primary"]["node_tests_per_packet"] < 1000 and w["primary"]["fallback_packets"] < 0.05 * w["primary"]["packets"]
    for k in ("closest", "shadow"):
        assert w[k]["node_visits_lds_per_ray"] >= 1.0  # the root is staged in LDS
        assert 1.0 <= w[k]["node_visits_lds_per_ray"] + w[k]["node_visits_mem_per_ray"] < 100 and 0 < w[k]["tri_tests_per_ray"] < 100
    assert w["wave"]["node_block_execs"] <= w["wave"]["iterations"] and 1 <= w["wave"]["lanes_per_node_block"] <= 64


def test_error_behaviour(xpu):
    from phosphorus_mk2_amd import scenes
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4, paths_per_sample=1))
    sc = scenes.cornell(32, 32)
    with pytest.raises(xpu.DeviceError):  # start before preprocess
        dev._scene = sc
        dev.start(sc, xpu.FrameState(1, xpu.Tiles.make(32, 32), xpu.Film(32, 32)))
    dark = scenes.cornell(32, 32); dark.meshes = dark.meshes[:5]  # no emissive face set (SURVEY A-19)
    with pytest.raises(xpu.DeviceError):
        dev.preprocess(dark)
    lens = scenes.cornell(32, 32); lens.camera.aperture_radius = float("inf")  # a finite aperture is the thin lens (test_thin_lens_*)
    with pytest.raises(xpu.DeviceError):
        dev.preprocess(lens)
    pin = scenes.cornell(32, 32); pin.camera.focal_distance = float("nan")  # camera_t() leaves it uninitialised: it means nothing without a lens
    dev.preprocess(pin)
    lens.camera.aperture_radius = 0.1; lens.camera.focal_distance = float("nan")
    with pytest.raises(xpu.DeviceError):
        dev.preprocess(lens)
    assert xpu.HipDevice.discover(xpu.Options(host_only=True)) == []
    dev.close()


def test_one_device_many_scenes_and_frames(xpu):
    """xpu_t::preprocess is called again for every scene (cpu.cpp:35-44 rebuilds the accelerator) and start/join once per
    frame: a device that has rendered other scenes, film sizes and tile sets must give what a fresh device gives."""
    from phosphorus_mk2_amd import scenes
    a, b, c = scenes.cornell(96, 64), scenes.soup(4000, width=64, height=96), scenes.multi_material_soup(2000, width=64, height=64)
    fresh = {id(s): xpu.render(s, spp=5, seed=3)[0] for s in (a, b, c)}
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=5, paths_per_sample=1))
    for s in (a, b, c, b, a):
        dev.preprocess(s)
        W, H = s.camera.width, s.camera.height
        for rep in range(2):  # the second frame reuses the cached pixel table of the same tiles
            film = xpu.Film(W, H, 4)
            dev.start(s, xpu.FrameState(3, xpu.Tiles.make(W, H, 32), film))
            dev.join()
            assert bits_equal(film.data, fresh[id(s)])
        half = xpu.Film(W, H, 4)  # other tiles on the same device: one rank's share of two
        dev.start(s, xpu.FrameState(3, xpu.Tiles.make(W, H, 32, 1, 2), half))
        dev.join()
        mask = half.data[..., :3].sum(-1) != 0
        assert mask.any() and bits_equal(half.data[mask], fresh[id(s)][mask])
    dev.close()


def test_full_size_properties(xpu, orc):
    """BASELINE config #2 (100k soup, 1280x720) at reduced spp: ray accounting, and the WHOLE frame — all 920 tiles — against
    the oracle, bit for bit (the AVX2 oracle traces the 8.8 M rays of 4 spp in about a second on 16 threads)."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(100000)
    film, st = xpu.render(sc, spp=4, seed=1)
    assert film.shape == (720, 1280, 4) and np.isfinite(film).all()
    assert st["camera_samples"] == 1280 * 720 * 4
    assert st["rays_closest"] >= st["camera_samples"] and st["rays_shadow"] + st["rays_masked"] <= st["rays_closest"]
    # (a) the oracle with the device's tie rule (equal-distance hits go to the lowest primitive index): everything is exact
    orc.set_tie_rule(1)
    try:
        ref, ost = orc.Oracle(sc, spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=16)
    finally:
        orc.set_tie_rule(0)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"]
    assert max_pixel_l2(film, ref) < L2_TOL and bits_equal(film[..., :3], ref[..., :3])
    # (b) the oracle as the reference resolves ties (first met wins, i.e. by the layout of ITS tree): the film is still within
    # the north-star tolerance and the ray counts differ by at most a few rays in millions
    ref0, ost0 = orc.Oracle(sc, spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=16)
    assert max_pixel_l2(film, ref0) < L2_TOL
    d0 = film[..., :3].astype(np.float64) - ref0[..., :3].astype(np.float64)
    assert int((np.sqrt((d0 * d0).sum(-1)) > L2_TOL).sum()) == 0  # pixels above the gate under the REFERENCE's tie rule: 0 on this frame (README: 1 / 18 on others)
    for k in ("rays_closest", "rays_shadow", "rays_masked"):
        assert abs(st[k] - ost0[k]) <= 4, (k, st[k], ost0[k])


def test_full_size_frame_with_a_lit_edge_band(xpu, orc):
    """BASELINE config #2's film (1280x720 = 22 tile rows + a 16-row edge band) with the camera aimed so that the band and the last
    column look INTO the cloud (conftest.aim_camera: with the identity camera rows 704-720 see nothing): the whole frame
    against the oracle bit for bit, and the band itself must be lit and must have traced shadow rays."""
    from conftest import aim_camera
    from phosphorus_mk2_amd import scenes
    sc = aim_camera(scenes.soup(100000), 0.5, 0.3)
    film, st = xpu.render(sc, spp=4, seed=5)
    O = orc.Oracle(sc, spp=4)
    orc.set_tie_rule(1)
    try:
        ref, ost = O.render(rng=orc.RNG_COUNTER, seed=5, threads=16)
        band = [(x, 704, 32, 16) for x in range(0, 1280, 32)]
        _, bst = O.render(rng=orc.RNG_COUNTER, seed=5, threads=16, tiles=band)
    finally:
        orc.set_tie_rule(0)
        O.close()
    for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert bits_equal(film[..., :3], ref[..., :3])
    assert bst["rays_closest"] > 1.3 * 1280 * 16 * 4 and bst["rays_shadow"] > 0.2 * 1280 * 16 * 4, bst  # the band's rays hit the cloud
    assert (film[704:, 640:, :3].sum(-1) > 0).mean() > 0.08 and (film[:, 1248:, :3].sum(-1) > 0).mean() > 0.2  # band and last column are lit


def test_host_and_device_trees_give_the_same_film_in_the_closed_room(xpu):
    """Results must not depend on the tree.  Round 6 found the ONE case in ~1e11 rays where they did: pixel (1685, 541) of this very frame differed
    between the device-built and the host-built tree — a ray along a facet edge that Moeller-Trumbore accepts 3.4e-6 outside the facet (38 x epsilon x
    the distance), where the host tree's exact child box culled it.  Both builders now grow every triangle's box by the test's own tolerance
    (bvh8.h: tri_box_inflation); this frame — 2 G rays per tree — pins it."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.bmw_showroom(500_000, width=1920, height=1080)
    films, stats = {}, {}
    for b in ("device", "host"):
        films[b], stats[b] = xpu.render(sc, spp=256, seed=1, native_sink=True, bvh_builder=b)
    for k in ("rays_closest", "rays_shadow", "rays_masked"):
        assert stats["device"][k] == stats["host"][k], k
    assert stats["device"]["bvh_built_on_device"] == 1 and stats["host"]["bvh_built_on_device"] == 0
    a, c = films["device"][..., :3], films["host"][..., :3]
    fin = np.isfinite(a).all(-1)
    assert np.array_equal(fin, np.isfinite(c).all(-1)) and fin.mean() > 0.99999
    assert int((a[fin].view(np.uint32) != c[fin].view(np.uint32)).any(-1).sum()) == 0
    assert bits_equal(a[541, 1685], c[541, 1685])  # the pixel that differed


def test_closed_room_whole_frame_matches_oracle(xpu, orc):
    """scenes.bmw_showroom(500 000) — closed room, mesh spheres, 16 recipes + glass, 7 rays per camera sample, a deep tree (the 5-byte-stack plan of
    k_trace, the per-hit kernels of k_shade_g) — whole 1280x720 frame at 32 spp against the oracle: counts and every pixel, bit for bit.  (Round 6:
    this frame had ONE pixel off by L2 5e-5; the device was right — the oracle's traversal missed a hit on a facet edge, see
    tests/test_oracle_render.py::test_traversal_finds_the_hit_on_a_facet_edge_that_brute_force_finds.)"""
    from phosphorus_mk2_amd import scenes
    sc = scenes.bmw_showroom(500_000, width=1280, height=720)
    film, st = xpu.render(sc, spp=32, seed=1, native_sink=True)
    assert st["shade_general"] == 1 and st["trace_stack_packed"] == 1 and st["trace_lds_levels"] < st["trace_levels"]
    O = orc.Oracle(sc, spp=32)
    orc.set_tie_rule(1)
    try:
        ref, ost = O.render(rng=orc.RNG_COUNTER, seed=1, threads=16)
    finally:
        orc.set_tie_rule(0)
        O.close()
    for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    fin = np.isfinite(ref[..., :3]).all(-1)
    assert fin.mean() > 0.99999 and np.array_equal(fin, np.isfinite(film[..., :3]).all(-1))
    assert int((film[..., :3][fin].view(np.uint32) != ref[..., :3][fin].view(np.uint32)).any(-1).sum()) == 0
    assert st["rays_closest"] + st["rays_shadow"] > 6 * st["camera_samples"]


def test_tie_rule_deviation_on_the_1M_soup_is_one_pixel(xpu, orc):
    """The documented deviation, kept measurable: the device gives equal-distance hits to the lowest primitive index, the reference to
    whichever triangle ITS traversal of ITS tree meets first (src/accel/triangle.hpp:166-179).  On Soup(1 M) at 64 spp that costs
    exactly ONE pixel of 921 600 the north-star gate (L2 0.131: one camera ray whose tie goes the other way — a different material
    behind it); under the device's rule the whole film is exact.  The COUNT is pinned so that a regression to many is caught."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(1_000_000, seed=1234, width=1280, height=720)
    film, st = xpu.render(sc, spp=64, seed=1, native_sink=True)
    O = orc.Oracle(sc, spp=64, pps=1, depth=9)
    try:
        orc.set_tie_rule(1)
        try:
            ref, ost = O.render(rng=orc.RNG_COUNTER, seed=1, threads=16)
        finally:
            orc.set_tie_rule(0)
        ref0, ost0 = O.render(rng=orc.RNG_COUNTER, seed=1, threads=16)
    finally:
        O.close()
    for k in ("rays_closest", "rays_shadow", "rays_masked"):
        assert st[k] == ost[k], k
        assert abs(st[k] - ost0[k]) <= 4, (k, st[k], ost0[k])
    assert bits_equal(film[..., :3], ref[..., :3])
    d0 = film[..., :3].astype(np.float64) - ref0[..., :3].astype(np.float64)
    l2 = np.sqrt((d0 * d0).sum(-1))
    assert int((l2 > L2_TOL).sum()) == 1 and 0.05 < float(l2.max()) < 0.3, (int((l2 > L2_TOL).sum()), float(l2.max()))
    assert int((l2 > 0).sum()) <= 4


def test_baseline_config_1_at_its_real_size(xpu, orc):
    """BASELINE.json configs[0]: the Cornell box (12 triangles, 1 area light), 256x256, 64 spp, depth 9 — the reference's own
    CPU-runnable case, whole frame, device against oracle: ray counts exact and the film bit for bit under the device's tie rule;
    under the reference's first-met rule the one exact-tie pixel of this scene (the camera ray through the back wall's diagonal,
    profiles/r03_zzb_full_parity_cornell.json: L2 4.8e-5) stays below the north-star gate."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.cornell(256, 256)
    film, st = xpu.render(sc, spp=64, seed=1, native_sink=True)
    orc.set_tie_rule(1)
    try:
        ref, ost = orc.Oracle(sc, spp=64, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=1, threads=16)
    finally:
        orc.set_tie_rule(0)
    assert st["camera_samples"] == ost["camera_samples"] == 256 * 256 * 64
    for k in ("rays_closest", "rays_shadow", "rays_masked"):
        assert st[k] == ost[k], k
    assert max_pixel_l2(film, ref) < L2_TOL and bits_equal(film[..., :3], ref[..., :3]) and film[..., :3].max() > 0.1
    ref0, _ = orc.Oracle(sc, spp=64, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=1, threads=16)
    d0 = film[..., :3].astype(np.float64) - ref0[..., :3].astype(np.float64)
    l2 = np.sqrt((d0 * d0).sum(-1))
    assert int((l2 > L2_TOL).sum()) == 0 and int((l2 > 0).sum()) <= 2


def test_auto_builder_falls_back_to_the_host_only_for_recoverable_failures(xpu, orc):
    """PHX_BVH_AUTO builds on the device; if that build cannot get its memory or meets a tree deeper than its tables (a RECOVERABLE
    failure) preprocess must not fail: the host's binned-SAH builder takes over, says so on stderr and in phx_stats, and the film is the
    same.  Any other failure of the device builder (a HIP error, lost triangles) fails preprocess even under AUTO, and an explicit
    DEVICE_LBVH request always fails loudly.  The fault injection lives in a twin library (libphx_hip_hooks.so, -DPHX_TEST_HOOKS=1):
    the product library does not read PHX_TEST_FAIL_DEVICE_BUILD."""
    import os, subprocess, sys, hashlib
    from conftest import ROOT
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(5000, width=64, height=64)
    os.environ["PHX_TEST_FAIL_DEVICE_BUILD"] = "recoverable"
    try:
        want, wst = xpu.render(sc, spp=4, seed=2)  # the product library ignores the variable
    finally:
        del os.environ["PHX_TEST_FAIL_DEVICE_BUILD"]
    assert wst["bvh_built_on_device"] == 1
    hooks = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_hooks.so")
    assert os.path.exists(hooks), "build() makes the fault-injection twin"
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from phosphorus_mk2_amd import scenes, xpu\n"
            "sc = scenes.soup(5000, width=64, height=64)\n"
            "for builder in ('auto', 'device'):\n"
            "    try:\n"
            "        film, st = xpu.render(sc, spp=4, seed=2, bvh_builder=builder)\n"
            "        print(builder, 'RENDERED', st['bvh_built_on_device'], st['rays_closest'], hashlib.sha1(film.tobytes()).hexdigest())\n"
            "    except xpu.DeviceError as e:\n"
            "        print(builder, 'FAILED:', e)\n") % ROOT
    out = {}
    for how in ("recoverable", "fatal"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180, env=dict(os.environ, PHX_LIB=hooks, PHX_TEST_FAIL_DEVICE_BUILD=how))
        assert r.returncode == 0, (r.stdout, r.stderr[-800:])
        out[how] = ({l.split()[0]: l for l in r.stdout.splitlines() if l.split() and l.split()[0] in ("auto", "device")}, r.stderr)
    rec, rec_err = out["recoverable"]
    assert rec["auto"].split()[1:4] == ["RENDERED", "0", str(wst["rays_closest"])] and rec["auto"].split()[4] == hashlib.sha1(want.tobytes()).hexdigest()
    assert "fell back to the host builder" in rec_err and "FAILED:" in rec["device"] and "device BVH build" in rec["device"]
    fat, fat_err = out["fatal"]
    assert "FAILED:" in fat["auto"] and "device BVH build" in fat["auto"] and "FAILED:" in fat["device"] and "fell back" not in fat_err
    again, ast_ = xpu.render(sc, spp=4, seed=2)
    assert ast_["bvh_built_on_device"] == 1 and bits_equal(again, want)


def test_destroying_a_device_with_a_frame_in_flight_joins_it(xpu, orc):
    """phx_dev_destroy after phx_dev_start WITHOUT phx_dev_join (include/phx_xpu.h: destroy joins): the call returns when the frame has
    ended, every tile has been delivered through the add_tile callback by then, and the film is the one a joined frame gives."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(20000, width=160, height=96)
    ref, _ = xpu.render(sc, spp=16, seed=3)
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=16, paths_per_sample=1, path_depth=9))
    dev.preprocess(sc)
    film = xpu.Film(160, 96, 4)
    dev.start(sc, xpu.FrameState(3, xpu.Tiles.make(160, 96, 32), film))  # the Python add_tile callback: fires from the driver thread
    dev.close()                                                          # no join()
    assert bits_equal(film.data[..., :3], ref[..., :3])
    dev2 = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4, paths_per_sample=1, path_depth=9))  # and the process can go on making devices
    dev2.preprocess(sc); dev2.close()


def test_watchdog_fails_the_frame_instead_of_hanging(xpu, orc):
    """every k_trace wave reaches its exit: a wave that iterates longer than PHX_TRACE_WATCHDOG leaves its loop and raises
    DevStats::watchdog, and phx_dev_join reports the frame as failed (PHX_ERR_DEVICE) — never a hang that takes the GPU down.
    libphx_hip_wd.so is the same library built with a watchdog of 8 iterations (__graft_entry__.build): any real frame trips it.
    Afterwards the device is intact: the product library renders the oracle's film.  (The per-lane fallback walks of
    k_trace_primary and k_trace_rays are plain bounded tree walks and carry no watchdog.)"""
    import os, subprocess, sys
    from conftest import ROOT
    from phosphorus_mk2_amd import scenes
    wd = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_wd.so")
    assert os.path.exists(wd), "build() makes the watchdog twin"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from phosphorus_mk2_amd import scenes, xpu\n"
            "try:\n"
            "    xpu.render(scenes.soup(3000, width=64, height=64), spp=4, seed=2)\n"
            "    print('RENDERED')\n"
            "except xpu.DeviceError as e:\n"
            "    print('FAILED:', e)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, PHX_LIB=wd))
    assert r.returncode == 0 and "FAILED:" in r.stdout and "watchdog" in r.stdout and "RENDERED" not in r.stdout, (r.stdout, r.stderr[-500:])
    sc = scenes.soup(3000, width=64, height=64)
    film, st = xpu.render(sc, spp=4, seed=2)
    ref, ost = orc.Oracle(sc, spp=4, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=2, threads=8)
    assert st["rays_closest"] == ost["rays_closest"] and bits_equal(film[..., :3], ref[..., :3])


def test_append_ring_timeout_fails_the_frame_instead_of_hanging(xpu, orc):
    """k_shade_g's waves wait for a block of the workgroup's append ring only a bounded time (PHX_RING_SPINS): a wave that gives up raises
    DevStats::ring_watchdog, the workgroup drops its later records, and phx_dev_join reports the frame as FAILED (PHX_ERR_DEVICE) — never a hang, never a film
    with holes handed over as good (round 6: this is how an append variant that could deadlock showed up as a log line, profiles/r06_g_ring_merged_deadlock.log).
    libphx_hip_ringwd.so is the same library with a patience of zero and a ring of 2 x 64 entries (__graft_entry__.build): any busy workgroup trips it.
    Afterwards the device is intact: the product library renders the oracle's film of the same scene."""
    import os, subprocess, sys
    from conftest import ROOT
    from phosphorus_mk2_amd import scenes
    wd = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_ringwd.so")
    assert os.path.exists(wd), "build() makes the ring-watchdog twin"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from phosphorus_mk2_amd import scenes, xpu\n"
            "try:\n"
            "    xpu.render(scenes.multi_material_soup(3000, width=128, height=128), spp=16, seed=2)\n"
            "    print('RENDERED')\n"
            "except xpu.DeviceError as e:\n"
            "    print('FAILED:', e)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, PHX_LIB=wd))
    assert r.returncode == 0 and "FAILED:" in r.stdout and "append ring" in r.stdout and "RENDERED" not in r.stdout, (r.stdout, r.stderr[-500:])
    sc = scenes.multi_material_soup(3000, width=128, height=128)
    film, st = xpu.render(sc, spp=16, seed=2)
    orc.set_tie_rule(1)
    try:
        ref, ost = orc.Oracle(sc, spp=16, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=2, threads=8)
    finally:
        orc.set_tie_rule(0)
    fin = np.isfinite(ref[..., :3]).all(-1)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and bits_equal(film[..., :3][fin], ref[..., :3][fin])


def test_shade_grids_sized_by_the_queue_render_the_same_film(xpu):
    """enqueue_batch sizes the shade launches of step b + 1 by the queue length k_trace(b) publishes in pinned host memory (a grid for the
    capacity is 230 k mostly empty workgroups per launch on the bench frame); PHX_SHADE_GRID_BY_QUEUE=0 (read once per process) enqueues
    everything at once with grids for the capacity, as before.  Same film, same counts — in several passes too, and on the general kernel"""
    import hashlib, os, subprocess, sys
    from conftest import ROOT
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from phosphorus_mk2_amd import scenes, xpu\n"
            "for sc, kw in ((scenes.soup(20000, width=320, height=200), dict(spp=24, samples_in_flight=9)), (scenes.multi_material_soup(4000, width=96, height=64), dict(spp=9))):\n"
            "    film, st = xpu.render(sc, seed=4, depth=7, **kw)\n"
            "    print('R', hashlib.sha1(film.tobytes()).hexdigest(), st['rays_closest'], st['rays_shadow'], st['shade_launches'])\n") % ROOT
    runs = []
    for knob in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180, env=dict(os.environ, PHX_SHADE_GRID_BY_QUEUE=knob))
        assert r.returncode == 0, r.stderr[-500:]
        runs.append([l.split()[1:] for l in r.stdout.splitlines() if l.startswith("R ")])
    assert len(runs[0]) == 2 and runs[0] == runs[1], runs
    assert int(runs[0][0][3]) == 3 * 7  # three passes (9 + 9 + 6 samples) of seven steps


def test_frames_without_kernel_timing_render_the_same_film(xpu):
    """PHX_KERNEL_TIMING=0 (a probe knob, read once per process): no HIP events between the launches — phx_stats carries no kernel times,
    the film and the ray counts are the ones of a timed frame"""
    import hashlib, os, subprocess, sys
    from conftest import ROOT
    from phosphorus_mk2_amd import scenes
    want, wst = xpu.render(scenes.soup(3000, width=96, height=64), spp=9, seed=4)
    assert wst["closest_ms"] > 0 and wst["shade_kernel_ms"] > 0 and wst["trace_launches"] == 9
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from phosphorus_mk2_amd import scenes, xpu\n"
            "film, st = xpu.render(scenes.soup(3000, width=96, height=64), spp=9, seed=4)\n"
            "print('R', hashlib.sha1(film.tobytes()).hexdigest(), st['rays_closest'], st['rays_shadow'], st['closest_ms'], st['shade_ms'], st['trace_launches'])\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, PHX_KERNEL_TIMING="0"))
    assert r.returncode == 0, r.stderr[-500:]
    f = [l for l in r.stdout.splitlines() if l.startswith("R ")][0].split()
    assert f[1] == hashlib.sha1(want.tobytes()).hexdigest() and int(f[2]) == wst["rays_closest"] and int(f[3]) == wst["rays_shadow"]
    assert float(f[4]) == 0.0 and float(f[5]) == 0.0 and int(f[6]) == 0


def test_4k_film_in_several_batches(xpu, orc):
    """BASELINE config #4 shape of the film (3840x2160: more pixels than one tile batch holds, so the frame is rendered in
    several batches) with the normals channel on; tiles from every batch are compared with the oracle bit for bit."""
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(20000, width=3840, height=2160)
    film, st = xpu.render(sc, spp=2, seed=9, normals=True, native_sink=True)
    assert film.shape == (2160, 3840, 7) and np.isfinite(film).all()
    assert st["camera_samples"] == 3840 * 2160 * 2 and st["tiles"] == 120 * 68
    O = orc.Oracle(sc, spp=2)
    tiles = [(32 * x, 32 * y, 32, 32 if y < 67 else 16) for y in (0, 33, 34, 66, 67) for x in (0, 59, 119)]
    img, _, nrm = O.render(rng=orc.RNG_COUNTER, seed=9, threads=8, tiles=tiles, normals=True)
    for (x, y, w, h) in tiles:
        assert bits_equal(film[y:y + h, x:x + w, :3], img[y:y + h, x:x + w, :3])
        assert bits_equal(film[y:y + h, x:x + w, 4:7], nrm[y:y + h, x:x + w, :])


def _with_lens(sc, aperture, focal, yaw=0.0, pitch=0.0):
    if yaw or pitch:
        aim_camera(sc, yaw, pitch)
    sc.camera.aperture_radius, sc.camera.focal_distance = aperture, focal
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("spp,flight", [(3, 0), (16, 0), (256, 0), (40, 16)])
def test_thin_lens_camera_matches_oracle(xpu, orc, spp, flight):
    """camera_t::aperture_radius != 0 (the Blender importer sets it when depth of field is on): camera_ray<LENS> in k_trace_primary<RPL, true>
    — 1, 2 and 4 rays per lane; a packet's rays leave from a disc, not a point — and in the first k_shade of a pass, against the oracle's
    restatement of camera.hpp:140-147 (tests/test_thin_lens.py), bit for bit, on a ragged film with a turned camera"""
    from phosphorus_mk2_amd import scenes
    sc = _with_lens(scenes.soup(3000, width=72, height=40), 0.05, 2.5, 0.2, -0.1)
    kw = {"samples_in_flight": flight} if flight else {}
    film, st, (ref, ost) = _render_both(xpu, orc, sc, spp=spp, seed=11, **kw)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"]
    assert bits_equal(film[..., :3], ref[..., :3]) and film[..., :3].max() > 0.05
    pin, _ = xpu.render(_with_lens(scenes.soup(3000, width=72, height=40), 0.0, 2.5, 0.2, -0.1), spp=spp, seed=11, **kw)
    assert not bits_equal(pin[..., :3], film[..., :3])  # the lens is really on


@pytest.mark.gpu
def test_thin_lens_camera_in_every_shade_kernel(xpu, orc):
    """the first shade kernel of a pass rebuilds the camera ray: k_shade<1>, k_shade<2>, k_shade_g<false> and k_shade_g<true> each have a
    thin-lens instantiation.  Cornell box (one Lambert lobe), several Lambert lobes, the 16 closure recipes, glass — with the normals
    channel (the primary hit's shading normal) on the first"""
    from phosphorus_mk2_amd import abi, scenes
    D = abi.LOBE_DIFFUSE
    mats = [scenes.MaterialDesc([scenes.LobeDesc(D, (0.4, 0.3, 0.2)), scenes.LobeDesc(D, (0.2, 0.3, 0.4))]), scenes.diffuse(0.73, 0.73, 0.73)]
    cases = [("cornell", _with_lens(scenes.cornell(64, 48), 0.02, 1.5), 16),
             ("lobes", _with_lens(scenes.soup(4000, width=96, height=64, materials=mats), 0.04, 2.2), 8),
             ("zoo", _with_lens(scenes.multi_material_soup(4000, width=64, height=64), 0.04, 2.2, 0.1, 0.1), 16),
             ("glass", _with_lens(scenes.glass_blobs(96, 64), 0.03, 2.0), 16)]
    for name, sc, spp in cases:
        normals = name == "cornell"
        film, st, res = _render_both(xpu, orc, sc, spp=spp, seed=4, normals=normals)
        ref, ost = res[0], res[1]
        assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_masked"] == ost["rays_masked"], name
        fin = np.isfinite(ref[..., :3]).all(axis=-1)
        assert fin.mean() > 0.99 and np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1)), name
        assert bits_equal(film[..., :3][fin], ref[..., :3][fin]) and film[..., :3][fin].max() > 0.05, name
        if normals:
            assert bits_equal(film[..., 4:7], res[2]) and np.abs(res[2]).max() > 0.5, name


@pytest.mark.gpu
def test_thin_lens_zero_lens_sample_is_a_ray_that_hits_nothing(xpu, orc):
    """a lens sample with a zero coordinate: non-finite angle, NaN ray (tests/test_thin_lens.py finds the pixels that own one by inverting
    the counter RNG).  The device's conservative box test ignores NaNs, so the packet walk must not let such a ray in and the per-lane
    walk must not start: the ray misses, the frame is finite and equal to the oracle's, in both walks (the tile on the film's axes has
    packets of mixed octants)"""
    from test_thin_lens import pixels_with_a_zero_lens_sample
    from phosphorus_mk2_amd import scenes
    W = H = 1024
    found = pixels_with_a_zero_lens_sample(1, W, H, 64)
    assert len(found) >= 2
    tiles = sorted({(x // 32 * 32, y // 32 * 32, 32, 32) for x, y, _, _ in found[:4]})
    sc = _with_lens(scenes.soup(20000, width=W, height=H), 0.05, 2.5)
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=64, paths_per_sample=1, path_depth=9))
    dev.preprocess(sc)
    film = xpu.Film(W, H, 4)
    dev.start(sc, xpu.FrameState(1, xpu.CallbackTiles(tiles), film)); dev.join()
    st = dev.stats(); dev.close()
    ref, ost = orc.Oracle(sc, spp=64).render(rng=orc.RNG_COUNTER, seed=1, threads=8, tiles=tiles)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert np.isfinite(film.data).all() and bits_equal(film.data[..., :3], ref[..., :3])
    # the same pixels through the per-lane walk: aim the camera so that the film's axes (mixed direction octants) cross the first such tile
    x, y, _, _ = found[0]
    sc2 = _with_lens(scenes.soup(20000, width=W, height=H), 0.05, 2.5)
    # pixel (x, y) looks along (fx, fy, -1) with the identity camera; turning the camera by the opposite angles puts the world's -z axis there
    zoom = 1.12 * np.tan(1.9 / 2)
    fx, fy = ((x + 0.5) / W - 0.5) * zoom, (0.5 - (y + 0.5) / H) * zoom
    aim_camera(sc2, np.arctan(fx), -np.arctan(fy))
    film2, st2 = xpu.Film(W, H, 4), None
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=64, paths_per_sample=1, path_depth=9))
    dev.preprocess(sc2)
    dev.start(sc2, xpu.FrameState(1, xpu.CallbackTiles(tiles[:1]), film2)); dev.join()
    st2 = dev.stats(); dev.close()
    ref2, ost2 = orc.Oracle(sc2, spp=64).render(rng=orc.RNG_COUNTER, seed=1, threads=8, tiles=tiles[:1])
    assert st2["rays_closest"] == ost2["rays_closest"] and st2["rays_shadow"] == ost2["rays_shadow"]
    assert np.isfinite(film2.data).all() and bits_equal(film2.data[..., :3], ref2[..., :3])
