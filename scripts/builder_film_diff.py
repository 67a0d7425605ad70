#!/usr/bin/env python3
"""Do the device-built and the host-built tree give the same film?  Run ON the GPU box.
    python scripts/builder_film_diff.py bmwroom:500000 1920 1080 256
Lists the pixels whose bits differ (results must not depend on the tree: the box test is conservative, the tie rule is by primitive)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phosphorus_mk2_amd import scenes, xpu
what, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kind, n = what.split(":")
sc = {"bmwroom": lambda: scenes.bmw_showroom(int(n), width=W, height=H), "zoo": lambda: scenes.multi_material_soup(int(n), width=W, height=H),
      "soup": lambda: scenes.soup(int(n), width=W, height=H), "cornell": lambda: scenes.cornell(W, H),
      "glassroom": lambda: scenes.showroom(int(n), width=W, height=H, materials=[scenes.diffuse(0.6, 0.3, 0.2), scenes.glass(1.45), scenes.closure_zoo()[4]])}[kind]()
films, stats = {}, {}
for b in ("device", "host"):
    films[b], stats[b] = xpu.render(sc, spp=spp, pps=1, depth=9, seed=1, native_sink=True, bvh_builder=b)
a, c = films["device"][..., :3], films["host"][..., :3]
bad = np.argwhere((a.view(np.uint32) != c.view(np.uint32)).any(-1))
out = {"scene": sc.name, "film": [W, H], "spp": spp, "rays": {b: [stats[b][k] for k in ("rays_closest", "rays_shadow", "rays_masked")] for b in films},
       "pixels_differing": int(len(bad)), "nonfinite": {b: int((~np.isfinite(films[b][..., :3]).all(-1)).sum()) for b in films},
       "pixels": [{"xy": [int(x), int(y)], "device_tree": [float(v) for v in a[y, x]], "host_tree": [float(v) for v in c[y, x]]} for y, x in bad[:12]]}
print(json.dumps(out))
