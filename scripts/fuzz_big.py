#!/usr/bin/env python3
"""The fuzzer's big brother: a few dozen LARGE random scenes (soup / showroom, 100 k .. 3 M triangles, random closure recipes,
small ragged films, both builders) against the oracle under the device's tie rule.  Deep trees: stack levels 8..10, both k_trace
plans.  python scripts/fuzz_big.py [N] [first_seed]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from phosphorus_mk2_amd import scenes, xpu
from oracle import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
first = int(sys.argv[2]) if len(sys.argv) > 2 else 900000
bad, rows, t0 = [], [], time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    ntri = int(1.0e5 * 30.0 ** rng.random())
    zoo = scenes.closure_zoo() + [scenes.glass(float(rng.uniform(1.2, 1.8)), float(rng.choice([0.0, 0.15])))]
    mats = [zoo[int(k)] for k in rng.choice(len(zoo), int(rng.integers(1, 5)), replace=False)]
    W, H = int(rng.integers(8, 25)) * 8, int(rng.integers(40, 130))
    sc = scenes.soup(ntri, seed=seed, width=W, height=H, materials=mats) if rng.random() < 0.6 else scenes.showroom(ntri, seed=seed, width=W, height=H, materials=mats)
    spp, depth = int(rng.choice([4, 9, 16])), int(rng.choice([3, 9]))
    builder = str(rng.choice(["host", "device"]))
    film, st = xpu.render(sc, spp=spp, pps=1, depth=depth, seed=seed, normals=True, bvh_builder=builder, native_sink=True)
    orc.set_tie_rule(1)
    try:
        ref, ost, nrm = orc.Oracle(sc, spp=spp, pps=1, depth=depth).render(rng=orc.RNG_COUNTER, seed=seed, threads=16, normals=True)
    finally:
        orc.set_tie_rule(0)
    why = [k for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked") if st[k] != ost[k]]
    fin = np.isfinite(ref[..., :3]).all(axis=-1)
    if not np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1)): why.append("finite mask")
    elif not np.array_equal(film[..., :3][fin].view(np.uint32), ref[..., :3][fin].view(np.uint32)): why.append("film")
    if not np.array_equal(film[..., 4:7].view(np.uint32), nrm.view(np.uint32)): why.append("normals")
    row = {"seed": seed, "scene": sc.name, "film": [W, H], "spp": spp, "depth": depth, "builder": builder, "bvh_depth": st["bvh_depth"],
           "plan": [st["trace_block"], st["trace_ntop"], st["trace_levels"]], "rays": st["rays_closest"] + st["rays_shadow"], "differs": why}
    rows.append(row)
    if why: bad.append(row)
    print(f"{seed - first + 1}/{n} {sc.name} depth {st['bvh_depth']} {builder}: {'FAIL ' + str(why) if why else 'ok'} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
print(json.dumps({"scenes": n, "failed": len(bad), "rows": rows, "seconds": time.time() - t0}))
sys.exit(1 if bad else 0)
