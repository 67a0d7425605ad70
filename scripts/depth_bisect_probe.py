#!/usr/bin/env python3
"""At which path depth does a pixel start to differ between device and oracle?  Run ON the GPU box.
    python scripts/depth_bisect_probe.py bmwroom:500000 1280 720 32 237 191 [seed]
The sampler is keyed by (pixel, sample, depth dimension), so a frame rendered with path_depth = k is the depth-9 frame cut after k steps: the
smallest k at which the pixel differs names the step, and the oracle's per-sample increments from k - 1 to k name the sample."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phosphorus_mk2_amd import scenes, xpu
from oracle import oracle as orc

what, W, H, spp, px, py = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 1
kind, n = what.split(":")
sc = {"bmwroom": lambda: scenes.bmw_showroom(int(n), width=W, height=H), "zoo": lambda: scenes.multi_material_soup(int(n), width=W, height=H)}[kind]()
tile = [(px // 32 * 32, py // 32 * 32, min(32, W - px // 32 * 32), min(32, H - py // 32 * 32))]
strip = [(px // 8 * 8, py, min(8, W - px // 8 * 8), 1)]
orc.set_tie_rule(1)
out = {"pixel": [px, py], "depths": []}
prev = None
for k in range(1, 10):
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=k))
    dev.preprocess(sc)
    f = xpu.Film(W, H, 4)
    dev.start(sc, xpu.FrameState(seed, xpu.CallbackTiles(tile), f)); dev.join(); dev.close()
    O = orc.Oracle(sc, spp=spp, pps=1, depth=k)
    r, _ = O.render(rng=orc.RNG_COUNTER, seed=seed, threads=4, tiles=tile)
    a, b = f.data[py, px, :3], r[py, px, :3]
    tile_diff = int((f.data[..., :3].view(np.uint32) != r[..., :3].view(np.uint32)).any(-1).sum())
    per = np.array([O.render(rng=orc.RNG_COUNTER, seed=seed, threads=1, tiles=strip, sample_begin=s, sample_end=s + 1)[0][py, px, :3] for s in range(spp)], np.float64) * spp
    rec = {"path_depth": k, "pixel_equal": bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), "pixels_differing_in_the_tile": tile_diff,
           "device_minus_oracle_x_spp": ((a.astype(np.float64) - b.astype(np.float64)) * spp).tolist()}
    if prev is not None:
        inc = per - prev
        rec["oracle_per_sample_increment_red"] = [float(v) for v in inc[:, 0]]
    prev = per
    out["depths"].append(rec)
    O.close()
orc.set_tie_rule(0)
print(json.dumps(out))
