import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from phosphorus_mk2_amd import xpu, scenes
from conftest import random_rays
xpu.load_library()
sc = scenes.soup(100000, width=64, height=64)
devs = []
for b in ("host", "device"):
    d = xpu.HipDevice.make(xpu.Options(samples_per_pixel=1, paths_per_sample=1, bvh_builder=b))
    d.preprocess(sc); devs.append(d)
tot = 0; bad = 0
for it in range(40):
    o, d, tm = random_rays(2000000, 100 + it)
    a = devs[0].trace(o, d, tm); b = devs[1].trace(o, d, tm)
    m = (a["prim"] != b["prim"]) | (a["t"].view(np.uint32) != b["t"].view(np.uint32))
    tot += len(tm); bad += int(m.sum())
    for i in np.nonzero(m)[0][:5]:
        print("mismatch", it, i, a["prim"][i], b["prim"][i], a["t"][i], b["t"][i], a["t"][i] == b["t"][i], flush=True)
print("rays", tot, "mismatches", bad)
