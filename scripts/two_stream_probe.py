#!/usr/bin/env python3
"""Experiment: does overlapping one device object's k_shade (memory-bound) with another's k_trace (VALU-bound) on the SAME GPU pay?
N phx_devices share one tile queue (the reference's multi-device mechanism) on one GPU; PHX_TRACE_WG_CAP caps the persistent k_trace at
that many workgroups per CU so that the other device's kernels find free wave slots.  python scripts/two_stream_probe.py [triangles]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phosphorus_mk2_amd import scenes, xpu
tri = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ndev = int(os.environ.get("PROBE_DEVICES", "2"))
world = int(os.environ.get("PROBE_WORLD", "1"))  # the tiles of rank 0 of `world` only
sc = scenes.soup(tri)
W, H = sc.camera.width, sc.camera.height
ntiles = (((W + 31) // 32) * ((H + 31) // 32) + world - 1) // world
opts = xpu.Options(samples_per_pixel=256, paths_per_sample=1, path_depth=9, device_ordinal=0, tiles_per_batch=(ntiles + ndev - 1) // ndev)
devs = [xpu.HipDevice.make(opts) for _ in range(ndev)]
for d in devs:
    d.preprocess(sc)
best = None
for rep in range(5):
    t0 = time.perf_counter()
    film, sts = xpu.render_on(devs, sc, seed=1, world=world)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
rays = sum(s["rays_closest"] + s["rays_shadow"] for s in sts)
print(json.dumps({"devices": ndev, "world": world, "wg_cap": os.environ.get("PHX_TRACE_WG_CAP"), "triangles": tri, "frame_ms": best * 1e3, "Mrays_per_s": rays / best / 1e6,
                  "tiles": [s["tiles"] for s in sts], "trace_ms": [round(s["trace_ms"], 1) for s in sts], "shade_ms": [round(s["shade_ms"], 1) for s in sts]}))
for d in devs:
    d.close()
