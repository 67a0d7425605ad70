#!/usr/bin/env python3
"""Run ON the GPU box: the same frame with different tile-batch sizes (phx_options.tiles_per_batch).  A batch of P pixels carries
S = min(spp, 256 M / P) samples per pass; path ids are pixel-major, so S decides how many samples of ONE pixel sit side by side in a wave.
  python scripts/batch_probe.py --triangles 10000000 --width 3840 --height 2160 --spp 256 --tiles 0 1024 2048 4096"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phosphorus_mk2_amd import scenes, xpu
p = argparse.ArgumentParser()
p.add_argument("--scene", default="soup"); p.add_argument("--triangles", type=int, default=10000000)
p.add_argument("--width", type=int, default=3840); p.add_argument("--height", type=int, default=2160); p.add_argument("--spp", type=int, default=256)
p.add_argument("--tiles", type=int, nargs="+", default=[0, 1024, 2048, 4096]); p.add_argument("--frames", type=int, default=2)
a = p.parse_args()
sc = scenes.multi_material_soup(a.triangles, width=a.width, height=a.height) if a.scene == "zoo" else scenes.soup(a.triangles, width=a.width, height=a.height)
for tpb in a.tiles:
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=a.spp, paths_per_sample=1, path_depth=9, tiles_per_batch=tpb))
    dev.preprocess(sc)
    film = xpu.Film(a.width, a.height, 4); tiles = xpu.Tiles.make(a.width, a.height, 32)
    best = None
    for _ in range(a.frames):
        tiles.reset(); t0 = time.time()
        dev.start(sc, xpu.FrameState(1, tiles, film, native_sink=True)); dev.join()
        dt = time.time() - t0; best = dt if best is None else min(best, dt)
    st = dev.stats(); rays = st["rays_closest"] + st["rays_shadow"]
    print(f"tiles_per_batch {tpb:5d}: {rays / best / 1e6:6.0f} Mrays/s  frame {best * 1e3:7.1f} ms  primary {st['primary_ms']:6.1f}  k_trace {st['closest_ms']:7.1f}  shade {st['shade_kernel_ms']:6.1f}  "
          f"other {st['shade_ms'] - st['shade_kernel_ms']:5.1f}  paths in flight {st['paths_in_flight'] / 1e6:5.0f} M  launches {st['trace_launches']}  film {hashlib.sha1(film.data.tobytes()).hexdigest()[:12]}", flush=True)
    dev.close()
