"""The thin-lens camera (camera_t::aperture_radius != 0; reference src/kernels/cpu/camera.hpp:140-147 with
simd::concentric_sample_disc, src/math/simd/sampling.hpp:8-32, restated AS WRITTEN — SURVEY A-21): CPU checks of the oracle's
restatement.  The device is compared with it in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import aim_camera, bits_equal

M32 = 0xFFFFFFFF


def mix32(x):
    x &= M32
    x ^= x >> 16; x = (x * 0x7FEB352D) & M32
    x ^= x >> 15; x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def unmix32(x):
    x &= M32
    x ^= x >> 16; x = (x * pow(0x846CA68B, -1, 1 << 32)) & M32
    x ^= (x >> 15) ^ (x >> 30); x = (x * pow(0x7FEB352D, -1, 1 << 32)) & M32
    x ^= x >> 16
    return x


def path_key(seed, pixel, sample):  # csrc/phx_math.h / oracle/orng.h
    k = mix32((seed & M32) ^ 0x85EBCA6B)
    k = mix32(k + pixel)
    k = mix32(k ^ (seed >> 32))
    return mix32(k + sample * 0x9E3779B1)


def draw_f32(key, dim):
    return np.float32(mix32(key + (dim + 1) * 0x9E3779B9) >> 8) * np.float32(1.0 / 16777216.0)


def pixels_with_a_zero_lens_sample(seed, width, height, spp):
    """(x, y, sample, dim) whose lens draw is exactly 0: mix32 is a bijection, so the 256 keys whose draw in dimension 6 / 7 is zero can be
    walked back through path_key to the pixel that owns them, for every sample index"""
    base = mix32((seed & M32) ^ 0x85EBCA6B)
    out = []
    for dim in (6, 7):
        for v in range(256):
            key = (unmix32(v) - (dim + 1) * 0x9E3779B9) & M32
            for s in range(spp):
                k3 = (unmix32(key) - s * 0x9E3779B1) & M32
                k2 = unmix32(k3) ^ (seed >> 32)
                pixel = (unmix32(k2) - base) & M32
                if pixel < width * height:
                    assert path_key(seed, pixel, s) == key and draw_f32(key, dim) == 0.0
                    out.append((pixel % width, pixel // width, s, dim))
    return out


def lens_scene(width=96, height=80, aperture=0.05, focal=2.5, n=3000):
    from phosphorus_mk2_amd import scenes
    sc = aim_camera(scenes.soup(n, width=width, height=height), 0.2, -0.1)
    sc.camera.aperture_radius, sc.camera.focal_distance = aperture, focal
    return sc


def test_mix32_inverse_and_counter_draws_match_the_oracle(orc):
    rng = np.random.default_rng(3)
    for x in rng.integers(0, 1 << 32, 64):
        assert unmix32(mix32(int(x))) == int(x)
    out = np.zeros(8, np.float32)
    orc.load().orc_counter_rng(77, 1234, 5, 8, out.ctypes.data_as(orc.abi.f32p))
    assert [draw_f32(path_key(77, 1234, 5), d) for d in range(8)] == list(out)


def test_zero_aperture_is_the_pinhole_camera(orc):
    sc = lens_scene(aperture=0.0)
    O = orc.Oracle(sc, spp=4)
    o, d = O.camera_rays((32, 32, 32, 32), 2, seed=9)
    assert np.all(o == 0.0)
    sc2 = lens_scene(aperture=0.0, focal=7.0)  # the focal distance means nothing without an aperture
    o2, d2 = orc.Oracle(sc2, spp=4).camera_rays((32, 32, 32, 32), 2, seed=9)
    assert bits_equal(d, d2) and bits_equal(o, o2)


def test_lens_rays_leave_the_aperture_disc_and_meet_in_the_focal_plane(orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(500, width=64, height=64)  # identity camera: camera space = world space
    sc.camera.aperture_radius, sc.camera.focal_distance = 0.08, 3.0
    pin = scenes.soup(500, width=64, height=64)
    tile = (32, 0, 32, 32)
    for s in range(4):
        o, d = orc.Oracle(sc, spp=4).camera_rays(tile, s, seed=5)
        _, dp = orc.Oracle(pin, spp=4).camera_rays(tile, s, seed=5)
        assert np.all(o[:, 2] == 0.0) and np.hypot(o[:, 0], o[:, 1]).max() <= 0.08 * (1 + 1e-6)
        assert np.hypot(o[:, 0], o[:, 1]).max() > 0.04 and len(np.unique(o[:, 0])) > 1000  # one lens sample per slot
        # the point a pinhole ray reaches at z = -focal ... is where the lens ray of the same slot goes through
        focus = dp * (3.0 / np.abs(dp[:, 2:3]))
        hit = o + d * ((-3.0 - o[:, 2:3]) / d[:, 2:3])
        assert np.abs(hit - focus).max() < 2e-5
        assert np.abs(np.linalg.norm(d.astype(np.float64), axis=1) - 1).max() < 1e-6


def test_lens_mapping_is_the_one_the_reference_wrote(orc):
    """concentric_sample_disc as written: raw samples (not 2 u - 1), theta = 4/pi * (y / x) resp. 2/pi - 4/pi * (x / y), and the simd
    select() that returns its THIRD argument where the mask is set: radius and angle are those of the branch the names do not suggest"""
    from phosphorus_mk2_amd import scenes
    W = H = 64
    sc = scenes.soup(500, width=W, height=H)
    sc.camera.aperture_radius, sc.camera.focal_distance = 0.25, 2.0
    seed, s, tile = 21, 1, (0, 32, 32, 32)
    o, _ = orc.Oracle(sc, spp=4).camera_rays(tile, s, seed=seed)
    for k in (0, 17, 500, 1023):
        x, y = tile[0] + k % 32, tile[1] + k // 32
        key = path_key(seed, y * W + x, s)
        ux, uy = float(draw_f32(key, 6)), float(draw_f32(key, 7))
        gt = abs(ux) > abs(uy)
        r = uy if gt else ux
        th = (2 / np.pi - 4 / np.pi * (ux / uy)) if gt else 4 / np.pi * (uy / ux)
        assert abs(o[k, 0] - r * np.cos(th) * 0.25) < 2e-6 and abs(o[k, 1] - r * np.sin(th) * 0.25) < 2e-6


def test_thin_lens_render_defocuses_and_keeps_the_counter_mode_invariants(orc):
    sc = lens_scene()
    O = orc.Oracle(sc, spp=16)
    a, sa = O.render(rng=orc.RNG_COUNTER, seed=5, threads=1)
    b, sb = O.render(rng=orc.RNG_COUNTER, seed=5, threads=8)
    assert bits_equal(a, b) and sa["rays_closest"] == sb["rays_closest"] and np.isfinite(a).all()
    tiles = [(64, 64, 32, 16), (0, 32, 32, 32)]
    c, _ = O.render(rng=orc.RNG_COUNTER, seed=5, threads=2, tiles=tiles)
    for (x, y, w, h) in tiles:
        assert bits_equal(c[y:y + h, x:x + w], a[y:y + h, x:x + w])
    p, _ = orc.Oracle(lens_scene(aperture=0.0), spp=16).render(rng=orc.RNG_COUNTER, seed=5, threads=8)
    assert not bits_equal(a, p)
    assert abs(a[..., :3].mean() / p[..., :3].mean() - 1) < 0.1  # the same scene, blurred
    # the sequential (reference-order mt19937) mode takes its lens samples from the table sampler_t::preprocess draws
    q, sq = O.render(rng=orc.RNG_SEQ)
    assert np.isfinite(q).all() and abs(q[..., :3].mean() / a[..., :3].mean() - 1) < 0.1


def test_a_zero_lens_sample_gives_a_ray_that_hits_nothing(orc):
    """u.x == 0 (or u.y == 0) makes the angle non-finite (y / 0 resp. 0 / 0): the reference's ray is NaN and misses everything"""
    W = H = 1024
    found = pixels_with_a_zero_lens_sample(1, W, H, 64)
    assert found, "no pixel of this film owns a zero lens draw under seed 1: pick another seed"
    x, y, s, dim = found[0]
    from phosphorus_mk2_amd import scenes
    sc = scenes.soup(2000, width=W, height=H)
    sc.camera.aperture_radius, sc.camera.focal_distance = 0.05, 2.5
    O = orc.Oracle(sc, spp=64)
    tile = (x // 32 * 32, y // 32 * 32, 32, 32)
    o, d = O.camera_rays(tile, s, seed=1)
    k = (y - tile[1]) * 32 + (x - tile[0])
    assert np.isnan(o[k]).any() and np.isnan(d[k]).all()
    assert np.isfinite(np.delete(o, k, axis=0)).all()
    film, st = O.render(rng=orc.RNG_COUNTER, seed=1, threads=8, tiles=[tile])
    assert np.isfinite(film).all()
