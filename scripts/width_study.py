#!/usr/bin/env python3
"""k_trace data-structure study (CPU only; round-5 verdict, item 3): before writing a kernel, what would another layout of the SAME
binary tree cost per ray?  Uses the host harness (tests/native/host_bvh8.cpp: the device's builder, box test and triangle test compiled
for the CPU) on bounce rays of the bench scenes:
  (i)   today's BVH8, 64-byte nodelets: 4 lane addresses per node visit
  (ii)  a 4-wide collapse of the same binary tree (PHX_WIDTH=4; counted in the same nodelets, half of their slots empty) priced as
        32-byte nodelets: 2 lane addresses per visit, half the slab arithmetic
  (iii) BVH8 with the any-hit rays visiting the hit children of a node largest-first (or smallest-first) instead of in octant order
Lane addresses per ray = addresses per node visit x node visits FROM MEMORY + 3 x triangle tests; VALU clocks per ray from the instruction
budget of profiles/r04_ktrace_budget.md (node block 737 clocks for 8 children, triangle block 256).
    python scripts/width_study.py [--triangles 100000 1000000] [--rays 200000]
Rays: closest-hit rays = cosine-weighted bounce rays leaving the camera rays' hit points (what k_trace's first launch sees);
any-hit rays = from those hit points to uniform points on the light, tmax = distance - 1e-4 (the NEE rays of the same step)."""
import argparse, ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from phosphorus_mk2_amd import abi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--triangles", type=int, nargs="+", default=[100000, 1000000])
ap.add_argument("--rays", type=int, default=200000)
ap.add_argument("--ntop", type=int, default=281, help="nodelets of the top of the tree served from LDS (281 at 100 k): visits to pool elements below this index cost no lane address")
a = ap.parse_args()


def load():
    d = os.path.join(ROOT, "tests", "native")
    so = os.path.join(d, "libhost_bvh8_study.so")
    src = [os.path.join(d, "host_bvh8.cpp"), os.path.join(ROOT, "phosphorus_mk2_amd", "csrc", "bvh_build.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-march=haswell", "-mfma", "-ffp-contract=off", "-pthread", "-o", so] + src, check=True)
    lib = C.CDLL(so)
    lib.hb8_build.restype = C.c_void_p; lib.hb8_build.argtypes = [abi.f32p, C.c_uint32, C.c_int]
    lib.hb8_free.argtypes = [C.c_void_p]; lib.hb8_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hb8_trace.argtypes = [C.c_void_p, C.c_uint32, abi.f32p, abi.f32p, abi.f32p, C.c_int, abi.f32p, abi.f32p, abi.f32p, abi.u32p, C.POINTER(C.c_uint64)]
    lib.hb8_trace_any_ordered.argtypes = [C.c_void_p, C.c_uint32, abi.f32p, abi.f32p, abi.f32p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    return lib


def fp(x):
    return x.ctypes.data_as(abi.f32p)


def tri_abc(sc):
    """triangles in scene_t::triangles() order as (n, 9) float32"""
    out = []
    for m in sc.meshes:
        for _, faces in m.sets:
            out.append(m.vertices[m.faces[faces]].reshape(-1, 9))
    return np.ascontiguousarray(np.concatenate(out), np.float32)


def trace(lib, h, o, d, tm, any_hit=False):
    n = len(tm)
    t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); p = np.zeros(n, np.uint32)
    ctr = (C.c_uint64 * 2)()
    lib.hb8_trace(h, n, fp(o), fp(d), fp(tm), 1 if any_hit else 0, fp(t), fp(u), fp(v), p.ctypes.data_as(abi.u32p), ctr)
    return t, p, ctr[0] / n, ctr[1] / n


def any_ordered(lib, h, o, d, tm, order):
    n = len(tm); occ = np.zeros(n, np.uint8); ctr = (C.c_uint64 * 2)()
    lib.hb8_trace_any_ordered(h, n, fp(o), fp(d), fp(tm), order, occ.ctypes.data_as(C.POINTER(C.c_uint8)), ctr)
    return occ, ctr[0] / n, ctr[1] / n


def bounce_rays(lib, h, abc, n, seed=5):
    """camera rays of the bench frame -> their hits -> (cosine-weighted bounce rays, rays towards the light quad)"""
    rng = np.random.default_rng(seed)
    W, H = 1280, 720
    zoom = 1.12 * np.tan(1.9 / 2)
    x = (rng.random(n) - 0.5) * (W / H) * zoom; y = (rng.random(n) - 0.5) * zoom
    d = np.stack([x, y, -np.ones(n)], 1); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.zeros((n, 3), np.float32); d = d.astype(np.float32); tm = np.full(n, np.finfo(np.float32).max, np.float32)
    t, prim, _, _ = trace(lib, h, o, d, tm)
    ok = prim != 0xffffffff
    o, d, t, prim = o[ok], d[ok], t[ok], prim[ok]
    T = abc[prim]
    nrm = np.cross(T[:, 3:6] - T[:, 0:3], T[:, 6:9] - T[:, 0:3]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    p = o + d * t[:, None]
    # cosine-weighted direction around nrm (either side: the reference never flips the normal either)
    u1, u2 = rng.random(len(p)), rng.random(len(p))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    tx = np.cross(nrm, np.where(np.abs(nrm[:, :1]) < 0.9, [[1.0, 0, 0]], [[0, 1.0, 0]])); tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(nrm, tx)
    bd = tx * (r * np.cos(phi))[:, None] + ty * (r * np.sin(phi))[:, None] + nrm * np.sqrt(np.maximum(0, 1 - u1))[:, None]
    bo = p + nrm * 1e-4
    return (bo.astype(np.float32), bd.astype(np.float32), np.full(len(p), np.finfo(np.float32).max, np.float32)), bo.astype(np.float32)


lib = load()
CLK_NODE8, CLK_TRI = 737.0, 256.0          # profiles/r04_ktrace_budget.md: VALU clocks of the node block (8 children) and of the triangle block
CLK_NODE4 = 37 + 31 + 14 + 117 + 514 / 2 + 25  # the same block with the slab arithmetic of four children
rows = []
for ntri in a.triangles:
    sc = scenes.soup(ntri, seed=1234, width=1280, height=720)
    abc = tri_abc(sc)
    light = abc[-2:].reshape(-1, 3)  # the emissive quad is the last mesh of the soup scenes
    lo, hi = light.min(0), light.max(0)
    res = {}
    for width in (8, 4):
        os.environ["PHX_WIDTH"] = str(width)
        t0 = time.time(); h = lib.hb8_build(fp(abc), len(abc), 8); tb = time.time() - t0
        info = (C.c_uint64 * 3)(); lib.hb8_info(h, info)
        if width == 8:
            (bo, bd, btm), so = bounce_rays(lib, h, abc, a.rays)
            rng = np.random.default_rng(9)
            P = lo + rng.random((len(so), 3)) * (hi - lo)
            sd = P - so; dist = np.linalg.norm(sd, axis=1); sd = (sd / dist[:, None]).astype(np.float32); stm = (dist - 1e-4).astype(np.float32)
        _, _, nv_c, tt_c = trace(lib, h, bo, bd, btm)
        _, _, nv_s, tt_s = trace(lib, h, so, sd, stm, any_hit=True)
        res[width] = dict(nodes=info[0], depth=info[2], build_s=tb, closest=(nv_c, tt_c), shadow=(nv_s, tt_s))
        if width == 8:
            ordered = {}
            for order, name in ((0, "octant order"), (1, "largest child first"), (2, "smallest child first")):
                occ, nv, tt = any_ordered(lib, h, so, sd, stm, order)
                ordered[name] = (nv, tt, float(occ.mean()))
            res["ordered"] = ordered
        lib.hb8_free(h)
    rows.append((ntri, len(bo), res))

print("## k_trace data-structure study (scripts/width_study.py, host harness)\n")
for ntri, nr, res in rows:
    print(f"### Soup({ntri}), {nr} bounce rays + {nr} NEE rays from the camera rays' hit points\n")
    print("| layout | nodelets | depth | node visits / closest ray | triangle tests / closest ray | node visits / any-hit ray | triangle tests / any-hit ray | lane addresses / closest ray (all visits from memory) | VALU clocks / closest ray |")
    print("|---|---|---|---|---|---|---|---|---|")
    for width, addr, clk, name in ((8, 4, CLK_NODE8, "(i) BVH8, 64-B nodelets (today)"), (4, 2, CLK_NODE4, "(ii) 4-wide collapse, priced as 32-B nodelets")):
        r = res[width]
        nv, tt = r["closest"]; nvs, tts = r["shadow"]
        print(f"| {name} | {r['nodes']} | {r['depth']} | {nv:.2f} | {tt:.2f} | {nvs:.2f} | {tts:.2f} | {addr * nv + 3 * tt:.1f} | {clk * nv + CLK_TRI * tt:.0f} |")
    r8, r4 = res[8], res[4]
    a8 = 4 * r8["closest"][0] + 3 * r8["closest"][1]; a4 = 2 * r4["closest"][0] + 3 * r4["closest"][1]
    c8 = CLK_NODE8 * r8["closest"][0] + CLK_TRI * r8["closest"][1]; c4 = CLK_NODE4 * r4["closest"][0] + CLK_TRI * r4["closest"][1]
    print(f"\n(ii) against (i): lane addresses {100 * (a4 / a8 - 1):+.1f} %, VALU clocks {100 * (c4 / c8 - 1):+.1f} %  (gate: >= 12 % fewer lane addresses at <= equal VALU clocks)\n")
    print("| (iii) any-hit visiting order (BVH8) | node visits / any-hit ray | triangle tests / any-hit ray | occluded |")
    print("|---|---|---|---|")
    for name, (nv, tt, occ) in res["ordered"].items():
        print(f"| {name} | {nv:.2f} | {tt:.2f} | {occ:.3f} |")
    o0, o1 = res["ordered"]["octant order"], res["ordered"]["largest child first"]
    print(f"\nlargest-first against octant order: node visits {100 * (o1[0] / o0[0] - 1):+.1f} %, triangle tests {100 * (o1[1] / o0[1] - 1):+.1f} % of the any-hit rays' work\n")
