#!/usr/bin/env python3
"""Which pixels differ between the device and the oracle (device tie rule), and in which sample?  Run ON the GPU box.
    python scripts/pixel_diff_probe.py bmwroom:500000 1280 720 32 [seed]          (PHX_LIB selects another build)
Renders the frame on both sides, lists the differing pixels (values and bit patterns), then — the device has no sample-range
option, but a pass of ONE sample per pixel (samples_per_pixel = s + 1, film scaled back) would change the jitter table — renders
each differing pixel's 32x32 tile again on the device alone (is the difference reproducible in a one-tile frame?) and the
oracle's per-sample contributions of the pixel, so that the sample whose value is off by the difference can be named."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from phosphorus_mk2_amd import scenes, xpu  # noqa: E402
from oracle import oracle as orc  # noqa: E402

what, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kind, n = what.split(":")
sc = {"bmwroom": lambda: scenes.bmw_showroom(int(n), width=W, height=H), "zoo": lambda: scenes.multi_material_soup(int(n), width=W, height=H),
      "bmwroom_cg": lambda: scenes.bmw_showroom(int(n), width=W, height=H, per_hit_glass=False)}[kind]()
film, st = xpu.render(sc, spp=spp, pps=1, depth=9, seed=seed, native_sink=True)
orc.set_tie_rule(1)
O = orc.Oracle(sc, spp=spp, pps=1, depth=9)
ref, ost = O.render(rng=orc.RNG_COUNTER, seed=seed, threads=16)
a, b = film[..., :3], ref[..., :3]
bad = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(-1))
out = {"lib": os.environ.get("PHX_LIB", "libphx_hip.so"), "scene": sc.name, "film": [W, H], "spp": spp, "rays_equal": all(st[k] == ost[k] for k in ("rays_closest", "rays_shadow", "rays_masked")),
       "pixels_differing": int(len(bad)), "pixels": []}
for y, x in bad[:8]:
    y, x = int(y), int(x)
    rec = {"xy": [x, y], "device": [float(v) for v in a[y, x]], "oracle": [float(v) for v in b[y, x]], "device_bits": [hex(int(v)) for v in a[y, x].view(np.uint32)],
           "oracle_bits": [hex(int(v)) for v in b[y, x].view(np.uint32)]}
    tile = [(x // 32 * 32, y // 32 * 32, min(32, W - x // 32 * 32), min(32, H - y // 32 * 32))]
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=9))
    dev.preprocess(sc)
    f2 = xpu.Film(W, H, 4)
    dev.start(sc, xpu.FrameState(seed, xpu.CallbackTiles(tile), f2)); dev.join(); dev.close()
    rec["device_one_tile_frame"] = [float(v) for v in f2.data[y, x, :3]]
    rec["device_one_tile_equals_full_frame"] = bool(np.array_equal(f2.data[y, x, :3].view(np.uint32), a[y, x].view(np.uint32)))
    strip = [(x // 8 * 8, y, min(8, W - x // 8 * 8), 1)]
    per = []
    for s in range(spp):
        f, _ = O.render(rng=orc.RNG_COUNTER, seed=seed, threads=1, tiles=strip, sample_begin=s, sample_end=s + 1)
        per.append([float(v) for v in f[y, x, :3]])
    rec["oracle_per_sample_x_spp"] = [[v * spp for v in p] for p in per]
    d = (np.asarray(rec["device"], np.float64) - np.asarray(rec["oracle"], np.float64)) * spp
    rec["difference_x_spp"] = [float(v) for v in d]
    out["pixels"].append(rec)
orc.set_tie_rule(0)
print(json.dumps(out))
