import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    from oracle import oracle as o
    o.load()
    return o


@pytest.fixture(scope="session")
def host_bvh8():
    """tests/native/libhost_bvh8.so: the device's BVH8 builder + traversal template compiled for the CPU."""
    import ctypes as C
    from phosphorus_mk2_amd import abi
    d = os.path.join(ROOT, "tests", "native")
    so = os.path.join(d, "libhost_bvh8.so")
    src = [os.path.join(d, "host_bvh8.cpp"), os.path.join(ROOT, "phosphorus_mk2_amd", "csrc", "bvh_build.cpp")]
    hdr = [os.path.join(ROOT, "phosphorus_mk2_amd", "csrc", h) for h in ("bvh8.h", "bvh_build.h", "phx_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src + hdr):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-march=haswell", "-mfma", "-ffp-contract=off", "-pthread",
                        "-DPHX_STUDY_KNOBS=1",  # the builder's study knobs (PHX_WIDTH ...) exist in this test / study library only
                        "-o", so] + src, check=True)
    lib = C.CDLL(so)
    lib.hb8_build.restype = C.c_void_p; lib.hb8_build.argtypes = [abi.f32p, C.c_uint32, C.c_int]
    lib.hb8_free.argtypes = [C.c_void_p]
    lib.hb8_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hb8_trace.argtypes = [C.c_void_p, C.c_uint32, abi.f32p, abi.f32p, abi.f32p, C.c_int, abi.f32p, abi.f32p, abi.f32p, abi.u32p,
                              C.POINTER(C.c_uint64)]
    lib.hb8_trace.restype = C.c_int
    lib.hb8_trace_any_ordered.argtypes = [C.c_void_p, C.c_uint32, abi.f32p, abi.f32p, abi.f32p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    return lib


def tri_abc(scene):
    """triangles in scene_t::triangles() order as (n, 9) float32"""
    out = []
    for m in scene.meshes:
        for _, faces in m.sets:
            out.append(m.vertices[m.faces[faces]].reshape(-1, 9))
    return np.ascontiguousarray(np.concatenate(out), np.float32)


def random_rays(n, seed, inside=True):
    rng = np.random.default_rng(seed)
    o = np.zeros((n, 3), np.float32)
    if inside:
        o[:, :2] = rng.uniform(-0.9, 0.9, (n, 2)); o[:, 2] = rng.uniform(-3.4, -1.6, n)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32), np.full(n, np.finfo(np.float32).max, np.float32)


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def max_pixel_l2(a, b):
    d = (a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    return float(np.sqrt((d * d).sum(axis=-1)).max())


def aim_camera(scene, yaw=0.0, pitch=0.0):
    """turn the camera `yaw` radians to the LEFT about y, then tilt it `pitch` radians UP about its x axis (CameraDesc.to_world, Imath
    row-vector convention: the view direction (0, 0, -1) becomes (-sin yaw cos pitch, sin pitch, -cos yaw cos pitch)).  The soups'
    identity camera (fov 1.9) sees y / |z| >= 0.75 in the bottom edge band of a 720 / 1080 / 2160-row film and |x / z| up to 1.39 in its
    last column, while the triangle cloud ends at 0.667: every edge-band and last-column tile is black with it.  Looking
    left and up moves the cloud to the lower right of the film, where the partial tiles are."""
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Rx = np.array([[1, 0, 0], [0, cp, sp], [0, -sp, cp]], np.float64)
    Ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]], np.float64)
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] = (Rx @ Ry).astype(np.float32)
    scene.camera.to_world = M
    return scene


def oracle_render_per_tile(O, tiles, **kw):
    """O.render() one tile at a time -> (film with every tile filled in, [stats of each tile]): a test can then require of EACH compared tile
    that it traced shadow rays (an all-miss tile compares black with black)"""
    film, stats = None, []
    for (x, y, w, h) in tiles:
        f, st = O.render(tiles=[(x, y, w, h)], **kw)
        film = f.copy() if film is None else film
        film[y:y + h, x:x + w] = f[y:y + h, x:x + w]
        stats.append(st)
    return film, stats
