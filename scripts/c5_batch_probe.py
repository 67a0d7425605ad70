"""Run ON the GPU box with PHX_HOST_TIMING=1: the config-5 frame (3840x2160, 4096 spp) on every 8th tile — eight batches of 130 k pixels —
twice; the library prints where each batch's host time goes.  python scripts/c5_batch_probe.py [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phosphorus_mk2_amd import scenes, xpu
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scenes.multi_material_soup(500000, seed=1234, width=3840, height=2160)
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=4096, paths_per_sample=1, path_depth=9, device_ordinal=0)); dev.preprocess(sc)
film = torch.zeros((2160, 3840, 4), dtype=torch.float32, device="cuda")
for rep in range(2):
    tiles = xpu.Tiles.make(3840, 2160, 32, 0, world)
    t0 = time.perf_counter(); dev.start(sc, xpu.FrameState(1, tiles, None, device_film_ptr=film.data_ptr())); dev.join(); dt = time.perf_counter() - t0
    st = dev.stats()
    print(f"frame {rep}: {len(tiles)} tiles wall {dt * 1e3:.1f} ms, kernels {st['trace_ms'] + st['shade_ms']:.1f} ms, frame_ms {st['frame_ms']:.1f}", flush=True)
dev.close()
