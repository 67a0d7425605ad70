#!/usr/bin/env python3
"""Run ON the GPU box: two device objects on ONE GPU render ALTERNATE frames (two frames in flight, each on its own stream), so that the
drain of one frame's launches overlaps the other frame's launches.  python scripts/pipeline_probe.py [world]  (rank 0's share of the bench frame)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phosphorus_mk2_amd import scenes, xpu
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H, N = 1280, 720, 16
sc = scenes.soup(100000, width=W, height=H)
def make():
    d = xpu.HipDevice.make(xpu.Options(samples_per_pixel=256, paths_per_sample=1, path_depth=9))
    d.preprocess(sc)
    return d, xpu.Tiles.make(W, H, 32, 0, world), torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for ndev in tuple(int(x) for x in os.environ.get("PROBE_DEVICES", "1,2").split(",")):
    devs = [make() for _ in range(ndev)]
    def start(i):
        d, t, f = devs[i % ndev]
        t.reset(); d.start(sc, xpu.FrameState(1, t, None, device_film_ptr=f.data_ptr()))
    rays = 0
    for rep in range(2):  # warm-up pass, then the timed one
        torch.cuda.synchronize(); t0 = time.perf_counter(); rays = 0
        for i in range(N + ndev):  # frame i starts once frame i - ndev (same device) has been joined; the last ndev iterations only join
            if i >= ndev:  # the frame started ndev frames ago must be done before its device is reused
                d = devs[i % ndev][0]; d.join(); st = d.stats(); rays += st["rays_closest"] + st["rays_shadow"]
            if i < N:
                start(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"world {world} devices {ndev}: {dt / N * 1e3:.3f} ms per frame over {N} frames, film_mean {float(devs[0][2][..., :3].mean()):.9g}")
    for d, _, _ in devs: d.close()
