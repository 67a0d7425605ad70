#!/usr/bin/env python3
"""Rays per bounce of the bench frame (depth d minus depth d-1) next to k_trace's launch times (profiles/r02_n_100k_pmc.json):
how efficient are the late, small launches?  python scripts/rays_per_bounce.py [triangles]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phosphorus_mk2_amd import scenes, xpu
tri = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sc = scenes.soup(tri)
prev = (0, 0); rows = []
for depth in range(1, 10):
    film, st = xpu.render(sc, spp=256, pps=1, depth=depth, seed=1, native_sink=True)
    cur = (st["rays_closest"], st["rays_shadow"])
    rows.append({"depth": depth, "closest": cur[0] - prev[0], "shadow": cur[1] - prev[1], "trace_ms_total": st["trace_ms"], "shade_ms_total": st["shade_ms"]})
    prev = cur
# launch b traces the closest rays of step b and the shadow rays made by step b-1
pm = json.load(open(os.path.join(ROOT, "profiles", "r02_n_100k_pmc.json")))["k_trace_launch_ms"] if tri == 100000 else None
out = []
for b in range(10):
    closest = rows[b]["closest"] if b < 9 else 0
    shadow = rows[b - 1]["shadow"] if b >= 1 else 0
    e = {"launch": b, "closest_rays": closest, "shadow_rays": shadow}
    if pm: e["ms"] = pm[b]; e["Grays_per_s"] = (closest + shadow) / pm[b] / 1e6
    out.append(e)
print(json.dumps({"triangles": tri, "launches": out, "by_depth": rows}))
