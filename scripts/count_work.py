#!/usr/bin/env python3
"""Traversal work per ray from the instrumented build (libphx_hip_count.so: `make -C phosphorus_mk2_amd/csrc variant NAME=count
EXTRA=-DPHX_COUNT=1`): node visits served from LDS / through the vector L1, triangle tests, and how full the wave's node and
triangle blocks run.  python scripts/count_work.py [--triangles N --width W --height H --spp S --builder host|device]"""
import argparse, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PHX_LIB"] = os.environ.get("PHX_COUNT_LIB") or os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_count.so")
from phosphorus_mk2_amd import scenes, xpu  # noqa: E402
p = argparse.ArgumentParser()
p.add_argument("--triangles", type=int, default=100000); p.add_argument("--width", type=int, default=1280); p.add_argument("--height", type=int, default=720)
p.add_argument("--spp", type=int, default=256); p.add_argument("--builder", default="auto")
a = p.parse_args()
sc = scenes.soup(a.triangles, width=a.width, height=a.height)
film, st = xpu.render(sc, spp=a.spp, pps=1, depth=9, seed=1, native_sink=True, bvh_builder=a.builder)
assert st["instrumented"] == 1
out = {"film_sha1": hashlib.sha1(film.tobytes()).hexdigest(), "triangles": a.triangles, "film": [a.width, a.height], "spp": a.spp, "builder": a.builder, "bvh_bytes": st["bvh_bytes"], "bvh_nodes": st["bvh_nodes"],
       "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]}}
# camera rays are walked as packets by k_trace_primary (unless PHX_PRIMARY_PACKETS=0): they are not k_trace's work
primary_rays = st["camera_samples"] if st["primary_launches"] else 0
pk = max(1, st["primary_packets"] - st["primary_fallbacks"])
out["primary"] = {"rays": primary_rays, "packets": st["primary_packets"], "fallback_packets": st["primary_fallbacks"],
                  "node_tests_per_packet": st["primary_node_tests"] / pk, "tri_tests_per_packet": st["primary_tri_tests"] / pk,
                  "lanes_improved_per_tri_test": st["primary_tri_lanes_hit"] / max(1, st["primary_tri_tests"])}
for k, name in ((0, "closest"), (1, "shadow")):
    rays = st["rays_closest"] - primary_rays if k == 0 else st["rays_shadow"]
    out[name] = {"rays": rays, "node_visits_lds_per_ray": st["node_visits_lds"][k] / rays, "node_visits_mem_per_ray": st["node_visits_mem"][k] / rays,
                 "tri_tests_per_ray": st["tri_tests"][k] / rays}
rays = st["rays_closest"] - primary_rays + st["rays_shadow"]
nv = sum(st["node_visits_lds"]) + sum(st["node_visits_mem"]); tt = sum(st["tri_tests"])
out["wave"] = {"iterations": st["wave_iters"], "node_block_execs": st["node_block_execs"], "tri_block_execs": st["tri_block_execs"], "refills": st["refills"], "idle_lanes_per_iteration": st["idle_lane_iters"] / max(1, st["wave_iters"]), "tri_pending_lanes_per_iteration": st["tri_pending_lane_iters"] / max(1, st["wave_iters"]),
               "lanes_per_node_block": nv / max(1, st["node_block_execs"]), "lanes_per_tri_block": tt / max(1, st["tri_block_execs"]),
               "iterations_per_ray_x64": st["wave_iters"] * 64 / rays,
               "stack_pushes_per_ray_by_depth": [x / rays for x in st["stack_pushes"]],
               # pending (ray, triangle) pairs of the wave when its triangle block runs (it tests one per pending lane): profiles/r06_tri_handoff.md
               "pending_pairs_per_tri_block": st["tri_pairs_pending"] / max(1, st["tri_block_execs"]),
               "tri_blocks_by_pending_pairs": dict(zip(("<=8", "<=16", "<=24", "<=32", "<=48", "<=64", "<=96", ">96"), [x / max(1, st["tri_block_execs"]) for x in st["tri_pairs_hist"]]))}
print(json.dumps(out))
