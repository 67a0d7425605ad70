"""Load balance of the static tile shard (tile i -> rank i % world): render every rank's share of the BASELINE frame on ONE GPU,
one after the other, and print the spread of the per-rank frame times (the job's time is the slowest rank's)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phosphorus_mk2_amd import scenes, xpu

xpu.load_library()
W, H, SPP = 1280, 720, 256
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = scenes.soup(100000, seed=1234, width=W, height=H)
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=SPP, paths_per_sample=1, path_depth=9))
dev.preprocess(scene)
film = xpu.Film(W, H, 4, False)
times, rays = [], []
for rank in range(world):
    best = 1e9
    for rep in range(2):
        fs = xpu.FrameState(1, xpu.Tiles.make(W, H, 32, rank, world), film, native_sink=True)
        t0 = time.perf_counter(); dev.start(scene, fs); dev.join(); best = min(best, (time.perf_counter() - t0) * 1e3)
    st = dev.stats()
    times.append(best); rays.append(st["rays_closest"] + st["rays_shadow"])
    print(f"rank {rank}: {best:6.2f} ms  {rays[-1] / 1e6:7.2f} M rays", flush=True)
print(f"world {world}: max {max(times):.2f} ms, mean {np.mean(times):.2f} ms, max/mean {max(times) / np.mean(times):.3f}; rays max/mean {max(rays) / np.mean(rays):.3f}")
dev.close()
