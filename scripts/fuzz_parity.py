#!/usr/bin/env python3
"""Differential fuzzing of the device against the CPU oracle: N scenes nobody designed (tests/test_gpu_parity.py::_random_scene:
shared vertices, smooth / flat faces, closure zoo + glass, one or two lights, optional environment, rotated camera, ragged
film; every fourth scene a soup or showroom of 64..60 000 triangles with random closure recipes: deep trees) x random options (spp, depth, builder, samples in flight, tiles per batch, callback tiles, seed).  Ray counts, film and
normals channel must agree bit for bit.  python scripts/fuzz_parity.py [N] [first_seed]  ->  one JSON line"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from phosphorus_mk2_amd import xpu
from oracle import oracle as orc
from test_gpu_parity import _random_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad, ties, rays, t0 = [], [], 0, time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed ^ 0x5bd1e995)
    if seed % 4 == 3:  # every fourth scene is a deep tree: a soup or the showroom with 64..60 000 triangles and random closure recipes
        from phosphorus_mk2_amd import scenes
        ntri = int(64 * (60000 / 64) ** rng.random())
        zoo = scenes.closure_zoo() + [scenes.glass(float(rng.uniform(1.2, 1.8)), float(rng.choice([0.0, 0.15])))]
        mats = [zoo[int(k)] for k in rng.choice(len(zoo), int(rng.integers(1, 6)), replace=False)]
        W, H = int(rng.integers(3, 17)) * 8, int(rng.integers(20, 97))
        sc = scenes.soup(ntri, seed=seed, width=W, height=H, materials=mats) if rng.random() < 0.6 else scenes.showroom(ntri, seed=seed, width=W, height=H, materials=mats)
    else:
        sc = _random_scene(seed)
    spp = int(rng.choice([1, 2, 3, 4, 7, 9, 16, 25])); depth = int(rng.choice([1, 2, 3, 5, 9, 12]))
    kw = dict(bvh_builder=str(rng.choice(["host", "device", "auto"])), samples_in_flight=int(rng.choice([0, 1, 3])),
              tiles_per_batch=int(rng.choice([0, 1, 2, 5])), callback_tiles=bool(rng.random() < 0.3), native_sink=bool(rng.random() < 0.5))
    fseed = int(rng.integers(0, 1 << 30))
    if seed % 4 == 3 and rng.random() < 0.35:  # a thin lens on a third of the deep-tree scenes too (_random_scene draws its own)
        sc.camera.aperture_radius, sc.camera.focal_distance = float(rng.uniform(0.005, 0.08)), float(rng.uniform(1.5, 3.5))
    try:
        film, st = xpu.render(sc, spp=spp, pps=1, depth=depth, seed=fseed, normals=True, **kw)
        orc.set_tie_rule(1)
        try:
            ref, ost, nrm = orc.Oracle(sc, spp=spp, pps=1, depth=depth).render(rng=orc.RNG_COUNTER, seed=fseed, threads=8, normals=True)
        finally:
            orc.set_tie_rule(0)
        ref0, ost0, nrm0 = orc.Oracle(sc, spp=spp, pps=1, depth=depth).render(rng=orc.RNG_COUNTER, seed=fseed, threads=8, normals=True)
        if not np.array_equal(ref0.view(np.uint32), ref.view(np.uint32)) or any(ost0[k] != ost[k] for k in ("rays_closest", "rays_shadow", "rays_masked")):
            ties.append(seed)
        why = [k for k in ("camera_samples", "rays_closest", "rays_shadow", "rays_masked") if st[k] != ost[k]]
        fin = np.isfinite(ref[..., :3]).all(axis=-1)
        if not np.array_equal(fin, np.isfinite(film[..., :3]).all(axis=-1)): why.append("finite mask")
        elif not np.array_equal(film[..., :3][fin].view(np.uint32), ref[..., :3][fin].view(np.uint32)): why.append("film")
        if not np.array_equal(film[..., 4:7].view(np.uint32), nrm.view(np.uint32)): why.append("normals")
        rays += st["rays_closest"] + st["rays_shadow"]
    except Exception as e:  # a device error is a finding too
        why = [f"exception: {e}"]
    if (seed - first) % 250 == 249:  # a long run must keep writing (the GPU pool kills silent commands)
        print(f"{seed - first + 1} scenes, {len(bad)} failed, {rays} rays, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    if why:
        bad.append({"seed": seed, "spp": spp, "depth": depth, "options": kw, "film_seed": fseed, "differs": why})
print(json.dumps({"scenes": n, "first_seed": first, "failed": len(bad), "failures": bad[:20], "scenes_where_the_reference_tie_rule_differs": ties, "rays_compared": rays, "seconds": time.time() - t0}))
sys.exit(1 if bad else 0)
