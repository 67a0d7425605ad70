#!/usr/bin/env python3
"""How much traversal work is overlap?  The bench soup with every triangle cut into 2 or 4 smaller triangles (longest-edge /
midpoint subdivision: the same surface, tighter boxes): node visits and triangle tests per ray (instrumented build) and k_trace
time.  An upper bound on what storing a triangle in several leaves under clipped boxes (reference splitting) could buy.
python scripts/presplit_probe.py [triangles]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
tri = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
mode = sys.argv[2] if len(sys.argv) > 2 else "time"
if mode == "count":
    os.environ["PHX_LIB"] = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_count.so")
from phosphorus_mk2_amd import scenes, xpu

def subdivide(sc, pieces):
    m = sc.meshes[0]
    v = np.asarray(m.vertices, np.float32).reshape(-1, 3, 3)
    if pieces == 1:
        return sc
    if pieces == 2:  # split the longest edge at its midpoint
        e = np.stack([np.linalg.norm(v[:, 1] - v[:, 0], axis=1), np.linalg.norm(v[:, 2] - v[:, 1], axis=1), np.linalg.norm(v[:, 0] - v[:, 2], axis=1)], 1)
        k = e.argmax(1); idx = np.arange(len(v))
        a = v[idx, k]; b = v[idx, (k + 1) % 3]; c = v[idx, (k + 2) % 3]; mid = (0.5 * (a + b)).astype(np.float32)
        out = np.concatenate([np.stack([a, mid, c], 1), np.stack([mid, b, c], 1)])
    else:  # four: the three edge midpoints
        a, b, c = v[:, 0], v[:, 1], v[:, 2]
        ab, bc, ca = (0.5 * (a + b)).astype(np.float32), (0.5 * (b + c)).astype(np.float32), (0.5 * (c + a)).astype(np.float32)
        out = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
    n = len(out)
    mesh = scenes.MeshDesc(vertices=out.reshape(-1, 3), faces=np.arange(3 * n, dtype=np.uint32).reshape(n, 3), sets=[(0, np.arange(n, dtype=np.uint32))])
    return scenes.SceneDesc([mesh] + list(sc.meshes[1:]), sc.materials, sc.camera, name=f"{sc.name}x{pieces}")

for pieces in (1, 2, 4):
    sc = subdivide(scenes.soup(tri), pieces)
    film, st = xpu.render(sc, spp=64, pps=1, depth=9, seed=1, native_sink=True)
    rays = st["rays_closest"] + st["rays_shadow"]
    row = {"pieces": pieces, "triangles": sc.num_triangles, "rays": rays, "trace_ms": st["trace_ms"], "bvh_MB": st["bvh_bytes"] / 1e6, "bvh_depth": st["bvh_depth"]}
    if mode == "count":
        nv = sum(st["node_visits_lds"]) + sum(st["node_visits_mem"]); tt = sum(st["tri_tests"])
        row.update({"node_visits_per_ray": nv / rays, "tri_tests_per_ray": tt / rays})
    print(json.dumps(row), flush=True)
