#!/usr/bin/env python3
"""One pixel differs between device and oracle: which of its rays, which closure evaluation?  Run ON the GPU box.
    python scripts/pixel_replay_probe.py bmwroom:500000 1280 720 32 237 191 [seed]
The oracle renders the pixel's strip with its diagnostic hook on (oracle.set_debug_pixel) and prints the shadow ray of every step of every
sample with the occlusion ITS traversal found; the same rays then go through the device's stage-level trace (phx_dev_trace, any-hit), the
oracle's traversal again and the oracle's BRUTE-FORCE test of every triangle (linear_mbvh_kernel_t semantics): who is right where they differ?"""
import contextlib
import json
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from phosphorus_mk2_amd import scenes, xpu  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@contextlib.contextmanager
def captured_stderr():
    sys.stderr.flush()
    saved = os.dup(2)
    tmp = tempfile.TemporaryFile(mode="w+b")
    os.dup2(tmp.fileno(), 2)
    box = {}
    try:
        yield box
    finally:
        os.dup2(saved, 2); os.close(saved)
        tmp.seek(0); box["text"] = tmp.read().decode(errors="replace"); tmp.close()


what, W, H, spp, px, py = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 1
kind, n = what.split(":")
sc = {"bmwroom": lambda: scenes.bmw_showroom(int(n), width=W, height=H), "zoo": lambda: scenes.multi_material_soup(int(n), width=W, height=H)}[kind]()
orc.set_tie_rule(1)
O = orc.Oracle(sc, spp=spp, pps=1, depth=9)
orc.set_debug_pixel(px, py)
with captured_stderr() as cap:
    O.render(rng=orc.RNG_COUNTER, seed=seed, threads=1, tiles=[(px // 8 * 8, py, min(8, W - px // 8 * 8), 1)])
orc.set_debug_pixel(-1, -1)
pat = re.compile(r"orc dbg pixel: sample (\d+) depth (\d+) hit (\d) shadow o (\S+) (\S+) (\S+) d (\S+) (\S+) (\S+) tmax (\S+) flags (\S+) occluded (\d) material (-?\d+) n (\S+) (\S+) (\S+) view (\S+) (\S+) (\S+) beta (\S+) (\S+) (\S+)")
pat_s = re.compile(r"orc dbg sample: sample (\d+) depth (\d+) material (-?\d+) n (\S+) (\S+) (\S+) view (\S+) (\S+) (\S+) u (\S+) (\S+) -> f (\S+) (\S+) (\S+) pdf (\S+) dir (\S+) (\S+) (\S+) flags (\S+)")
srows = [m.groups() for m in map(pat_s.match, cap["text"].splitlines()) if m]
rows = [m.groups() for m in map(pat.match, cap["text"].splitlines()) if m]
rows = [r for r in rows if r[2] == "1" and not (int(r[10], 16) & 2)]  # hits whose shadow ray was not masked (obvh.h: F_HIT = 1, F_MASKED = 2, F_SHADOW = 4)
o = np.array([[float(r[3]), float(r[4]), float(r[5])] for r in rows], np.float32)
d = np.array([[float(r[6]), float(r[7]), float(r[8])] for r in rows], np.float32)
tm = np.array([float(r[9]) for r in rows], np.float32)
occ_render = np.array([int(r[11]) for r in rows], bool)
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=9, bvh_builder=os.environ.get("PHX_PROBE_BUILDER", "auto")))
dev.preprocess(sc)
g = dev.trace(o, d, tm, shadow=True)
dev.close()
t = O.trace(o, d, tm, shadow=True)
b = O.trace(o, d, tm, shadow=True, brute=True)
out = {"scene": sc.name, "builder": os.environ.get("PHX_PROBE_BUILDER", "auto"), "pixel": [px, py], "spp": spp, "shadow_rays": len(rows), "oracle_trace_equals_its_render": bool(np.array_equal(t["hit"], occ_render)),
       "device_vs_oracle_mismatches": int((g["hit"] != t["hit"]).sum()), "device_vs_brute_mismatches": int((g["hit"] != b["hit"]).sum()),
       "oracle_vs_brute_mismatches": int((t["hit"] != b["hit"]).sum()), "rays": []}
for i in np.argwhere((g["hit"] != t["hit"]) | (g["hit"] != b["hit"]) | (t["hit"] != occ_render)).ravel():
    i = int(i)
    out["rays"].append({"sample": int(rows[i][0]), "depth": int(rows[i][1]), "o": [float(v) for v in o[i]], "d": [float(v) for v in d[i]], "tmax": float(tm[i]),
                        "oracle_render_occluded": bool(occ_render[i]), "flags_after_the_render's_trace": rows[i][10], "material_of_the_shaded_hit": int(rows[i][12]), "device_occluded": bool(g["hit"][i]), "oracle_traversal_occluded": bool(t["hit"][i]), "oracle_brute_force_occluded": bool(b["hit"][i]),
                        "device_t": float(g["t"][i]), "brute_t": float(b["t"][i]), "device_prim": int(g["prim"][i]), "brute_prim": int(b["prim"][i])})
# the closure evaluations of the same steps, replayed on both sides: li()'s f (light direction = the shadow ray's) and the sampled continuation
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=9, bvh_builder=os.environ.get("PHX_PROBE_BUILDER", "auto")))
dev.preprocess(sc)
out["f_mismatches"], out["sample_mismatches"] = [], []
for mat in sorted({int(r[12]) for r in rows}):
    idx = [i for i, r in enumerate(rows) if int(r[12]) == mat]
    nn = np.array([[float(rows[i][13]), float(rows[i][14]), float(rows[i][15])] for i in idx], np.float32)
    vw = np.array([[float(rows[i][16]), float(rows[i][17]), float(rows[i][18])] for i in idx], np.float32)
    fg, fo = dev.bsdf_f(mat, nn, d[idx], vw), O.bsdf_f(mat, nn, d[idx], vw)
    for k in np.argwhere((fg.view(np.uint32) != fo.view(np.uint32)).any(1)).ravel():
        i = idx[int(k)]
        out["f_mismatches"].append({"material": mat, "sample": int(rows[i][0]), "depth": int(rows[i][1]), "n": nn[k].tolist(), "wi": d[i].tolist(), "wo": vw[k].tolist(), "beta": [float(rows[i][19]), float(rows[i][20]), float(rows[i][21])],
                                    "device_f": fg[k].tolist(), "oracle_f": fo[k].tolist()})
for mat in sorted({int(r[2]) for r in srows}):
    idx = [i for i, r in enumerate(srows) if int(r[2]) == mat]
    nn = np.array([[float(srows[i][3]), float(srows[i][4]), float(srows[i][5])] for i in idx], np.float32)
    vw = np.array([[float(srows[i][6]), float(srows[i][7]), float(srows[i][8])] for i in idx], np.float32)
    u2 = np.array([[float(srows[i][9]), float(srows[i][10])] for i in idx], np.float32)
    wg, f2g, pg, flg = dev.bsdf_sample(mat, nn, vw, u2)
    wo_, f2o, po, flo = O.bsdf_sample(mat, nn, vw, u2)
    neq = (wg.view(np.uint32) != wo_.view(np.uint32)).any(1) | (f2g.view(np.uint32) != f2o.view(np.uint32)).any(1) | (pg.view(np.uint32) != po.view(np.uint32)) | (flg != flo)
    for k in np.argwhere(neq).ravel():
        i = idx[int(k)]
        out["sample_mismatches"].append({"material": mat, "sample": int(srows[i][0]), "depth": int(srows[i][1]), "n": nn[k].tolist(), "view": vw[k].tolist(), "u": u2[k].tolist(),
                                         "device": {"dir": wg[k].tolist(), "f": f2g[k].tolist(), "pdf": float(pg[k]), "flags": int(flg[k])}, "oracle": {"dir": wo_[k].tolist(), "f": f2o[k].tolist(), "pdf": float(po[k]), "flags": int(flo[k])}})
# the continuation rays (closest hit): origin = the step's hit point offset along +-n (spt.hpp:301), direction = the sampled one
all_rows = {(int(r[0]), int(r[1])): r for r in [m.groups() for m in map(pat.match, cap["text"].splitlines()) if m]}
co, cd, ck = [], [], []
for r in srows:
    smp, dep = int(r[0]), int(r[1])  # sample_bsdf prints the depth AFTER ++depth: the hit is the one of step dep - 1
    h = all_rows.get((smp, dep - 1))
    if h is None or float(r[14]) == 0.0:
        continue
    nn = np.array([float(h[13]), float(h[14]), float(h[15])], np.float32)
    so = np.array([float(h[3]), float(h[4]), float(h[5])], np.float32)       # p + 1e-4 n
    dd = np.array([float(r[15]), float(r[16]), float(r[17])], np.float32)
    hp = so - nn * np.float32(1e-4)
    co.append(hp + nn * np.float32(-1e-4 if float(np.dot(nn, dd)) < 0 else 1e-4)); cd.append(dd); ck.append((smp, dep))
if co:
    co, cd = np.array(co, np.float32), np.array(cd, np.float32); ctm = np.full(len(co), np.finfo(np.float32).max, np.float32)
    g2, t2, b2 = dev.trace(co, cd, ctm), O.trace(co, cd, ctm), O.trace(co, cd, ctm, brute=True)
    out["closest_rays_replayed"] = len(co)
    out["closest_mismatches"] = []
    for i in range(len(co)):
        if not (g2["prim"][i] == t2["prim"][i] == b2["prim"][i] and g2["t"][i].view(np.uint32) == t2["t"][i].view(np.uint32) == b2["t"][i].view(np.uint32)):
            out["closest_mismatches"].append({"sample": ck[i][0], "ray_into_depth": ck[i][1], "o": co[i].tolist(), "d": cd[i].tolist(),
                                              "device": [float(g2["t"][i]), int(g2["prim"][i])], "oracle_traversal": [float(t2["t"][i]), int(t2["prim"][i])], "oracle_brute_force": [float(b2["t"][i]), int(b2["prim"][i])]})
dev.close()
out["closure_evaluations_replayed"] = {"f": len(rows), "sample": len(srows)}
orc.set_tie_rule(0)
print(json.dumps(out))
