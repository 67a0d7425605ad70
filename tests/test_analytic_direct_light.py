"""A check that shares no code with the device OR with the CPU restatement: direct lighting of a Lambert plane under a
rectangular lamp has a closed form.  A depth-1 render (camera ray -> floor -> one next-event sample, spt.hpp:95-149, 212-255) must
converge to

    L(x) = (rho / pi) * (4 L_e) * G(x),     G(x) = integral over the lamp of cos(theta_x) cos(theta_L) / d^2 dA

— including the reference's unexplained factor 4 on L_e (src/kernels/cpu/spt.hpp:212-255, SURVEY A-7) and its uniform-by-area light
sampling with pdf = 1/A * d^2 / |n_L . w| (light.cpp:55-67).  For a point and a parallel rectangle G is elementary (the
differential-area-to-rectangle view factor times pi); the formula itself is checked against brute-force quadrature here.
Tolerances are Monte-Carlo: 256 spp, the estimator's relative spread over the lamp is < 0.5, so a pixel is good to a few percent
and the film mean to a few 1e-4."""
import math

import numpy as np
import pytest

H_LAMP, CAM_Y, RHO, FOV, W, SPP = 1.0, 0.6, 0.5, 1.9, 32, 256
LAMP = (-0.5, 0.5, -0.4, 0.6)  # x0, x1, z0, z1 (off-centre on purpose)
LE = (3.0, 2.0, 1.0)


def corner(a, b, h):
    """pi x view factor of a rectangle [0,a] x [0,b] at height h seen from the point below its corner (odd in a and in b)"""
    ra, rb = np.sqrt(a * a + h * h), np.sqrt(b * b + h * h)
    return 0.5 * (a / ra * np.arctan(b / ra) + b / rb * np.arctan(a / rb))


def G_closed(x, z):
    x0, x1, z0, z1 = LAMP
    return corner(x1 - x, z1 - z, H_LAMP) - corner(x0 - x, z1 - z, H_LAMP) - corner(x1 - x, z0 - z, H_LAMP) + corner(x0 - x, z0 - z, H_LAMP)


def G_quadrature(x, z, n=400):
    x0, x1, z0, z1 = LAMP
    xs = x0 + (np.arange(n) + 0.5) * (x1 - x0) / n; zs = z0 + (np.arange(n) + 0.5) * (z1 - z0) / n
    X, Z = np.meshgrid(xs, zs)
    d2 = (X - x) ** 2 + (Z - z) ** 2 + H_LAMP ** 2
    return float((H_LAMP * H_LAMP / (d2 * d2)).sum() * (x1 - x0) * (z1 - z0) / (n * n))  # cos cos / d^2 = h^2 / d^4


def scene():
    from phosphorus_mk2_amd import scenes as S
    mats = [S.diffuse(RHO, RHO, RHO), S.emitter(*LE)]
    meshes = []
    for (xa, xb) in ((-2.0, 0.0), (0.0, 2.0)):      # the floor, y = 0, four quads (the reference's builder wants >= 8 triangles), n = +y
        for (za, zb) in ((0.0, -2.0), (2.0, 0.0)):  # (front, back): front has the larger z
            meshes.append(S._quad((xa, 0.0, za), (xb, 0.0, za), (xb, 0.0, zb), (xa, 0.0, zb), 0))
    x0, x1, z0, z1 = LAMP
    meshes.append(S._quad((x0, H_LAMP, z1), (x0, H_LAMP, z0), (x1, H_LAMP, z0), (x1, H_LAMP, z1), 1))  # the lamp, n = -y
    # camera at (0, CAM_Y, 0) looking down -y: camera x -> world x, camera y -> world -z, camera z -> world +y (rows, Imath v * M)
    M = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, CAM_Y, 0, 1]], np.float32)
    return S.SceneDesc(meshes, mats, S.CameraDesc(W, W, FOV, to_world=M), name="lamp_over_plane")


def expected_film():
    """closed form at every pixel centre (camera::perspective_kernel_t's mapping, camera.hpp:80-159, with the jitter at 0.5)"""
    zoom = 1.12 * math.tan(FOV / 2)
    px = np.arange(W, dtype=np.float64)
    fx = (px / W - 0.5) * zoom                      # ((sx - .5)/W - .5 + .5/W) * (W/H) * zoom, W = H
    fy = (0.5 - (-0.5 + px) / W + 0.5 / W) * zoom   # d.y = (ndcy + jit.y / H) * zoom with ndcy = 0.5 - (-0.5 + sy) / H
    X = CAM_Y * fx[None, :] * np.ones((W, 1)); Z = -CAM_Y * fy[:, None] * np.ones((1, W))
    G = G_closed(X, Z)
    return np.stack([RHO / math.pi * 4.0 * le * G for le in LE], -1)


def check(film):
    exp = expected_film()
    got = film[..., :3].astype(np.float64)
    ratio = got / exp
    assert abs(ratio.mean() - 1.0) < 3e-3, ratio.mean()           # the film mean: 1024 pixels x 256 samples
    assert np.abs(ratio - 1.0).max() < 0.12 and ratio.std() < 0.03, (np.abs(ratio - 1.0).max(), ratio.std())  # per pixel: sigma = 2.2 % at 256 spp
    # the three channels are the same estimate scaled by L_e
    assert np.allclose(got[..., 0] / LE[0], got[..., 2] / LE[2], rtol=1e-5)


def test_closed_form_is_the_integral():
    for (x, z) in ((0.0, 0.0), (0.3, -0.2), (-0.45, 0.4), (0.9, 0.9)):
        assert abs(G_closed(x, z) - G_quadrature(x, z)) < 2e-5 * G_quadrature(x, z) + 1e-7


def test_oracle_depth_1_render_converges_to_the_closed_form(orc):
    film, st = orc.Oracle(scene(), spp=SPP, pps=1, depth=1).render(rng=orc.RNG_COUNTER, seed=3, threads=4)
    assert st["rays_shadow"] == st["rays_closest"] == W * W * SPP  # every camera ray hits the floor and sees the lamp's plane
    check(film)


@pytest.mark.gpu
def test_device_depth_1_render_converges_to_the_closed_form():
    from phosphorus_mk2_amd import xpu
    film, st = xpu.render(scene(), spp=SPP, pps=1, depth=1, seed=3, native_sink=True)
    assert st["rays_shadow"] == st["rays_closest"] == W * W * SPP
    check(film)
