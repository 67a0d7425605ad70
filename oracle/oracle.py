"""ORACLE — test infrastructure only.  ctypes loader for oracle/liboracle.so (the CPU restatement).

Importable ONLY from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (phosphorus_mk2_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from phosphorus_mk2_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
F_HIT, F_MASKED, F_SHADOW, F_SPECULAR = 1, 2, 4, 8
RNG_SEQ, RNG_COUNTER = 0, 1


class RenderArgs(C.Structure):
    _fields_ = [
        ("rng_mode", C.c_int32), ("rcp_approx", C.c_int32), ("slab_literal", C.c_int32), ("num_threads", C.c_int32),
        ("seed", C.c_uint64), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
        ("num_tiles", C.c_uint32), ("tiles", C.POINTER(abi.Tile)),
    ]


class OStats(C.Structure):
    _fields_ = [
        ("camera_samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_masked", C.c_uint64),
        ("node_visits_closest", C.c_uint64), ("packet_visits_closest", C.c_uint64),
        ("node_visits_shadow", C.c_uint64), ("packet_visits_shadow", C.c_uint64),
        ("rng_draws", C.c_uint64), ("seconds", C.c_double), ("bvh_nodes", C.c_uint64), ("bvh_packets", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(libm=False, quiet=True):
    target = "libm" if libm else "all"
    subprocess.run(["make", "-C", HERE, target], check=True, stdout=subprocess.DEVNULL if quiet else None)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, stdout=subprocess.DEVNULL if quiet else None)


_libs = {}


def load(libm=False):
    name = "liboracle_libm.so" if libm else "liboracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build(libm)
    lib = C.CDLL(path)
    vp, f32p, u32p = C.c_void_p, abi.f32p, abi.u32p
    lib.orc_create.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Options)]; lib.orc_create.restype = vp
    lib.orc_destroy.argtypes = [vp]; lib.orc_destroy.restype = None
    lib.orc_bvh_info.argtypes = [vp, C.POINTER(C.c_uint64)] + [C.POINTER(C.c_uint64)] * 2; lib.orc_bvh_info.restype = C.c_int
    lib.orc_render.argtypes = [vp, C.POINTER(RenderArgs), f32p, f32p, C.POINTER(OStats)]; lib.orc_render.restype = C.c_int
    lib.orc_bench.argtypes = [vp, C.POINTER(RenderArgs), C.c_double, f32p, C.POINTER(OStats), u32p]; lib.orc_bench.restype = C.c_int
    lib.orc_jitter_table.argtypes = [C.c_uint64, C.c_uint32, f32p]; lib.orc_jitter_table.restype = C.c_int
    lib.orc_camera_rays.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(abi.Tile), C.c_uint32, f32p, f32p]; lib.orc_camera_rays.restype = C.c_int
    lib.orc_trace.argtypes = [vp, C.c_uint32, f32p, f32p, f32p, u32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, u32p, u32p,
                              C.POINTER(C.c_uint64)]
    lib.orc_trace.restype = C.c_int
    lib.orc_bsdf_f.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, f32p, f32p, f32p]; lib.orc_bsdf_f.restype = C.c_int
    lib.orc_bsdf_sample.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, f32p, f32p, f32p, f32p, f32p, u32p]
    lib.orc_bsdf_sample.restype = C.c_int
    lib.orc_fresnel_dielectric.argtypes = [C.c_uint32, f32p, f32p, f32p]
    lib.orc_onb.argtypes = [C.c_uint32, f32p, f32p]
    lib.orc_cosine_weighted.argtypes = [C.c_uint32, f32p, f32p, f32p]
    lib.orc_sincos.argtypes = [C.c_uint32, f32p, f32p, f32p]
    lib.orc_exp_log_pow.argtypes = [C.c_uint32, f32p, f32p, f32p, f32p, f32p]
    lib.orc_mt19937_head.argtypes = [C.c_uint32, f32p]
    lib.orc_counter_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    lib.orc_light_sample.argtypes = [vp, C.c_uint32, f32p, f32p, f32p, f32p, f32p, u32p, u32p]; lib.orc_light_sample.restype = C.c_int
    lib.orc_probe_soup.argtypes = [C.c_uint32, f32p]; lib.orc_probe_soup.restype = C.c_int
    lib.orc_bvh_dump.argtypes = [vp, f32p, u32p, u32p, u32p, u32p, u32p]; lib.orc_bvh_dump.restype = C.c_int
    _libs[name] = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(abi.f32p)


def _up(a):
    return a.ctypes.data_as(abi.u32p)


def make_options(spp=16, pps=1, depth=9, **kw):
    o = abi.Options()
    o.samples_per_pixel, o.paths_per_sample, o.path_depth = spp, pps, depth
    o.device_ordinal = -1
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def set_scalar(on):
    """1: trace one child box / one triangle at a time; 0 (default): the same arithmetic on 8 AVX2 lanes.  Bit-identical
    (tests/test_oracle_render.py::test_simd_and_scalar_restatements_agree); applies to tracers created afterwards."""
    load().orc_set_scalar(1 if on else 0)


def set_tie_rule(lowest_prim):
    """0 (default, the reference): of two triangles hit at bitwise the same distance the one the traversal meets first wins;
    1: the one with the lower primitive index wins — the device's rule (bvh8.h).  Applies to tracers created afterwards."""
    load().orc_set_tie_rule(1 if lowest_prim else 0)


def set_debug_nonfinite(on):
    """diagnostic: the renderer prints to stderr where a non-finite value enters a path (li() / the throughput update); results unchanged"""
    load().orc_set_debug_nonfinite(1 if on else 0)


def set_debug_pixel(x, y):
    """diagnostic: the renderer prints to stderr the shadow ray of every step of every sample of film pixel (x, y) and its occlusion (x < 0: off)"""
    load().orc_set_debug_pixel(int(x), int(y))


def probe_soup(n):
    """(n, 9) float32 triangles of the soup the survey's probe rendered with the real reference:
    std::mt19937(1234) + std::uniform_real_distribution<float>(-1, 1), 12 draws per triangle."""
    abc = np.zeros((n, 9), np.float32)
    load().orc_probe_soup(n, _fp(abc))
    return abc


class Oracle:
    """One scene loaded into the CPU restatement (reference-layout BVH built on creation)."""

    def __init__(self, scene_desc, spp=16, pps=1, depth=9, libm=False):
        self.lib = load(libm)
        self.desc = scene_desc
        self.scene, self._keep = scene_desc.pack()
        self.opt = make_options(spp, pps, depth)
        self.h = self.lib.orc_create(C.byref(self.scene), C.byref(self.opt))
        if not self.h:
            raise RuntimeError("orc_create failed")
        self.W, self.H = scene_desc.camera.width, scene_desc.camera.height

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.orc_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"nodes": a.value, "packets": b.value, "triangles": c.value}

    def camera_rays(self, tile, sample, seed=1):
        """origins and directions [w * h, 3] of the camera rays of one tile (x, y, w, h) at one sample index (counter RNG)"""
        t = abi.Tile(*tile)
        o = np.zeros((tile[2] * tile[3], 3), np.float32); d = np.zeros_like(o)
        rc = self.lib.orc_camera_rays(self.h, seed, C.byref(t), sample, _fp(o), _fp(d))
        if rc != 0:
            raise RuntimeError(f"orc_camera_rays failed: {rc}")
        return o, d

    def render(self, rng=RNG_COUNTER, seed=1, threads=1, tiles=None, sample_begin=0, sample_end=0, slab_literal=0,
               rcp_approx=0, normals=False):
        film = np.zeros((self.H, self.W, 4), np.float32)
        nrm = np.zeros((self.H, self.W, 3), np.float32) if normals else None
        args = RenderArgs()
        args.rng_mode, args.rcp_approx, args.slab_literal, args.num_threads = rng, rcp_approx, slab_literal, threads
        args.seed, args.sample_begin, args.sample_end = seed, sample_begin, sample_end
        keep = None
        if tiles is not None:
            keep = (abi.Tile * len(tiles))(*[abi.Tile(*t) for t in tiles])
            args.num_tiles, args.tiles = len(tiles), keep
        st = OStats()
        rc = self.lib.orc_render(self.h, C.byref(args), _fp(film), _fp(nrm) if normals else None, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"orc_render failed: {rc}")
        return (film, st.as_dict(), nrm) if normals else (film, st.as_dict())

    def bench(self, threads, min_seconds, seed=1, tiles=None, sample_begin=0, sample_end=0):
        """CPU-baseline timing: a warm pool of `threads` workers renders the tile list round after round (counter RNG) for at
        least `min_seconds`; thread start-up and per-thread stream construction are outside the clock.  -> stats dict."""
        film = np.zeros((self.H, self.W, 4), np.float32)
        args = RenderArgs()
        args.rng_mode, args.num_threads, args.seed = RNG_COUNTER, threads, seed
        args.sample_begin, args.sample_end = sample_begin, sample_end
        keep = None
        if tiles is not None:
            keep = (abi.Tile * len(tiles))(*[abi.Tile(*t) for t in tiles])
            args.num_tiles, args.tiles = len(tiles), keep
        st = OStats(); rounds = C.c_uint32(0)
        rc = self.lib.orc_bench(self.h, C.byref(args), float(min_seconds), _fp(film), C.byref(st), C.byref(rounds))
        if rc != 0:
            raise RuntimeError(f"orc_bench failed: {rc}")
        d = st.as_dict(); d["rounds"] = rounds.value
        return d

    def trace(self, o, d, tmax, shadow=False, brute=False, slab_literal=0, rcp_approx=0):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(tmax)
        fl = np.full(n, F_SHADOW if shadow else 0, np.uint32)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32); flo = np.zeros(n, np.uint32)
        ctr = (C.c_uint64 * 3)()
        self.lib.orc_trace(self.h, n, _fp(o), _fp(d), _fp(tmax), _up(fl), 1 if brute else 0, slab_literal, rcp_approx,
                           _fp(t), _fp(u), _fp(v), _up(prim), _up(flo), ctr)
        return {"t": t, "u": u, "v": v, "prim": prim, "hit": (flo & F_HIT) != 0,
                "rays": ctr[0], "node_visits": ctr[1], "packet_visits": ctr[2]}

    def bsdf_f(self, material, n, wi, wo):
        n = np.ascontiguousarray(n, np.float32); wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32)
        out = np.zeros_like(wi)
        rc = self.lib.orc_bsdf_f(self.h, material, len(wi), _fp(n), _fp(wi), _fp(wo), _fp(out))
        assert rc == 0
        return out

    def bsdf_sample(self, material, n, wi, u2):
        n = np.ascontiguousarray(n, np.float32); wi = np.ascontiguousarray(wi, np.float32); u2 = np.ascontiguousarray(u2, np.float32)
        k = len(wi)
        wo = np.zeros((k, 3), np.float32); f = np.zeros((k, 3), np.float32); pdf = np.zeros(k, np.float32); fl = np.zeros(k, np.uint32)
        rc = self.lib.orc_bsdf_sample(self.h, material, k, _fp(n), _fp(wi), _fp(u2), _fp(wo), _fp(f), _fp(pdf), _up(fl))
        assert rc == 0
        return wo, f, pdf, fl

    def light_sample(self, pick, u2):
        pick = np.ascontiguousarray(pick, np.float32); u2 = np.ascontiguousarray(u2, np.float32)
        k = len(pick)
        p = np.zeros((k, 3), np.float32); uv = np.zeros((k, 2), np.float32); pdf = np.zeros(k, np.float32)
        mesh = np.zeros(k, np.uint32); face = np.zeros(k, np.uint32)
        rc = self.lib.orc_light_sample(self.h, k, _fp(pick), _fp(u2), _fp(p), _fp(uv), _fp(pdf), _up(mesh), _up(face))
        assert rc == 0
        return p, uv, pdf, mesh, face

    def bvh_dump(self):
        info = self.bvh_info()
        nn, npk = info["nodes"], info["packets"]
        b = np.zeros((nn, 48), np.float32); off = np.zeros((nn, 8), np.uint32); fl = np.zeros((nn, 8), np.uint32)
        num = np.zeros((nn, 8), np.uint32); pn = np.zeros(npk, np.uint32); pp = np.zeros((npk, 8), np.uint32)
        self.lib.orc_bvh_dump(self.h, _fp(b), _up(off), _up(fl), _up(num), _up(pn), _up(pp))
        return {"bounds": b, "offset": off, "flags": fl, "num": num, "packet_num": pn, "packet_prims": pp}
