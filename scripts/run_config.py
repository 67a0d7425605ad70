#!/usr/bin/env python3
"""Run one BASELINE.json-style configuration (or one rank's shard of it) through the device and print a JSON
summary: a functional check of the large configs on a single GPU.
  python scripts/run_config.py --scene soup --triangles 10000000 --width 3840 --height 2160 --spp 16 --world 8 --rank 0
  python scripts/run_config.py --scene zoo --triangles 500000 --width 1920 --height 1080 --spp 64
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from phosphorus_mk2_amd import scenes, xpu  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--scene", default="soup", choices=["soup", "zoo", "cornell", "showroom", "glassroom", "bmwroom", "bmwroom_cg"])
p.add_argument("--triangles", type=int, default=100000)
p.add_argument("--width", type=int, default=1280)
p.add_argument("--height", type=int, default=720)
p.add_argument("--spp", type=int, default=16)
p.add_argument("--world", type=int, default=1)
p.add_argument("--rank", type=int, default=0)
p.add_argument("--frames", type=int, default=2)
p.add_argument("--builder", default="auto", choices=["auto", "host", "device"])
a = p.parse_args()
t0 = time.time()
if a.scene == "soup":
    sc = scenes.soup(a.triangles, width=a.width, height=a.height)
elif a.scene == "zoo":
    sc = scenes.multi_material_soup(a.triangles, width=a.width, height=a.height)
elif a.scene == "showroom":
    sc = scenes.showroom(a.triangles, width=a.width, height=a.height)
elif a.scene == "glassroom":  # the showroom with Lambert, glass (per-hit Fresnel mix: k_shade_g<true>) and a glossy recipe on its spheres
    sc = scenes.showroom(a.triangles, width=a.width, height=a.height, materials=[scenes.diffuse(0.6, 0.3, 0.2), scenes.glass(1.45), scenes.closure_zoo()[4]])
elif a.scene == "bmwroom_cg":  # A/B only: the same room with the two glass materials as CONSTANT mixes (no per-hit weights: k_shade_g<false>)
    sc = scenes.bmw_showroom(a.triangles, width=a.width, height=a.height, per_hit_glass=False)
elif a.scene == "bmwroom":  # the closed showroom with the 16 recipes + sharp and frosted glass: the mesh-geometry stand-in for BASELINE configs 3 / 5
    sc = scenes.bmw_showroom(a.triangles, width=a.width, height=a.height)
else:
    sc = scenes.cornell(a.width, a.height)
t_scene = time.time() - t0
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=a.spp, paths_per_sample=1, path_depth=9, bvh_builder=a.builder))
t0 = time.time()
dev.preprocess(sc)
t_pre = time.time() - t0
film = xpu.Film(a.width, a.height, 4)
tiles = xpu.Tiles.make(a.width, a.height, 32, a.rank, a.world)
best = None
for _ in range(a.frames):
    tiles.reset(); film.data[:] = 0
    t0 = time.time()
    dev.start(sc, xpu.FrameState(1, tiles, film, native_sink=True)); dev.join()
    dt = time.time() - t0
    st = dev.stats()
    best = dt if best is None else min(best, dt)
rays = st["rays_closest"] + st["rays_shadow"]
print(json.dumps({"scene": sc.name, "triangles": sc.num_triangles, "film": [a.width, a.height], "spp": a.spp, "rank": a.rank, "world": a.world,
                  "tiles": len(tiles), "scene_gen_s": t_scene, "preprocess_s": t_pre, "frame_s": best, "rays": rays, "Mrays_per_s": rays / best / 1e6,
                  "builder": a.builder, "bvh_cost_model": st["bvh_cost_model"], "bvh_built_on_device": st["bvh_built_on_device"], "bvh_depth": st["bvh_depth"], "bvh_build_ms": st["bvh_build_ms"], "plan": [st["trace_block"], st["trace_ntop"], st["trace_levels"], st["trace_lds_levels"], st["trace_waves_per_cu"]], "trace_ms": st["trace_ms"], "primary_ms": st["primary_ms"], "shade_ms": st["shade_ms"], "bvh_MB": st["bvh_bytes"] / 1e6, "film_mean": float(film.data[..., :3].mean()), "film_sha1": __import__("hashlib").sha1(film.data.tobytes()).hexdigest()[:16], "k_trace_ms": st["closest_ms"], "shade_kernel_ms": st["shade_kernel_ms"], "other_ms": st["shade_ms"] - st["shade_kernel_ms"],
                  "finite": bool(np.isfinite(film.data).all())}))
dev.close()
