#!/usr/bin/env python3
"""Run ON the GPU box: rank 0's share of the bench frame for world sizes 1 / 2 / 4 / 8 on ONE GPU, the way bench.py runs a rank
(device film in HBM, one frame in flight; no RCCL, no other ranks: NOT a scaling result) — wall time per frame, kernel times,
and what is left outside the kernels.  python scripts/rank_share_probe.py [--frames 12] [--triangles 100000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phosphorus_mk2_amd import scenes, xpu
p = argparse.ArgumentParser(); p.add_argument("--frames", type=int, default=12); p.add_argument("--triangles", type=int, default=100000)
p.add_argument("--width", type=int, default=1280); p.add_argument("--height", type=int, default=720); p.add_argument("--spp", type=int, default=256)
a = p.parse_args()
sc = scenes.soup(a.triangles, width=a.width, height=a.height)
base = None
for world in (1, 2, 4, 8):
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=a.spp, paths_per_sample=1, path_depth=9, device_ordinal=0)); dev.preprocess(sc)
    tiles = xpu.Tiles.make(a.width, a.height, 32, 0, world)
    film = torch.zeros((a.height, a.width, 4), dtype=torch.float32, device="cuda")
    ms = {"primary": 0.0, "trace": 0.0, "shade": 0.0, "other": 0.0}
    for i in range(2 + a.frames):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        tiles.reset(); dev.start(sc, xpu.FrameState(1, tiles, None, device_film_ptr=film.data_ptr())); dev.join()
        if i >= 2:
            st = dev.stats(); ms["primary"] += st["primary_ms"]; ms["trace"] += st["closest_ms"]; ms["shade"] += st["shade_kernel_ms"]; ms["other"] += st["shade_ms"] - st["shade_kernel_ms"]
    wall = (time.perf_counter() - t0) * 1e3 / a.frames
    k = {n: v / a.frames for n, v in ms.items()}; ksum = sum(k.values())
    base = base or wall
    print(f"world {world}: rank 0 of {world}: {len(tiles):4d} tiles  wall {wall:6.2f} ms per frame ({base / world / wall * 100:5.1f} % of 1/{world} of the one-rank frame)  "
          f"kernels {ksum:6.2f} (primary {k['primary']:.2f} trace {k['trace']:.2f} shade {k['shade']:.2f} other {k['other']:.2f})  outside the kernels {wall - ksum:5.2f} ms", flush=True)
    dev.close(); del film
