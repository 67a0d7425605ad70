"""The parity gate at FULL size: BASELINE config #2 (Soup(100 000), 1280x720, 256 spp, depth 9) rendered by the device and by
the CPU oracle (all host threads), compared pixel by pixel.  Prints the max per-pixel L2 with the reference's tie rule and
with the device's (lowest primitive index on exact distance ties), and the ray counts.  ~1 min on the GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phosphorus_mk2_amd import scenes, xpu
from oracle import oracle as orc

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
what = sys.argv[2] if len(sys.argv) > 2 else "100000"   # triangle count of the soup, "cornell" (BASELINE config #1: 256x256), "zoo:N" (BMW stand-in, 1920x1080), "showroom:N", "bmwroom:N"
builder = sys.argv[3] if len(sys.argv) > 3 else "auto"
if what == "cornell":
    sc = scenes.cornell(256, 256)
elif what.startswith("zoo:"):
    sc = scenes.multi_material_soup(int(what[4:]), width=1920, height=1080)
elif what.startswith("bmwroom:"):  # the closed mesh room with the 16 recipes + sharp and frosted glass (BASELINE configs 3 / 5 on mesh geometry), 1280x720
    sc = scenes.bmw_showroom(int(what[8:]), width=1280, height=720)
elif what.startswith("showroom:"):
    sc = scenes.showroom(int(what[9:]), width=1280, height=720, materials=[scenes.diffuse(0.6, 0.3, 0.2), scenes.glass(1.45), scenes.closure_zoo()[4]])
else:
    sc = scenes.soup(int(what), seed=1234, width=1280, height=720)
if os.environ.get("PHX_LENS"):  # "aperture_radius,focal_distance": the thin-lens camera on the same scene
    sc.camera.aperture_radius, sc.camera.focal_distance = (float(v) for v in os.environ["PHX_LENS"].split(","))
import bench
threads = max(1, int(bench.host_cpus()[2]))  # the CPU share of this job: more oracle threads than that lose to context switches
t0 = time.time(); film, st = xpu.render(sc, spp=spp, pps=1, depth=9, seed=1, native_sink=True, bvh_builder=builder); t_gpu = time.time() - t0
out = {"scene": sc.name, "lens": [sc.camera.aperture_radius, sc.camera.focal_distance], "spp": spp, "builder": builder, "gpu_s": t_gpu, "gpu_rays": [st["rays_closest"], st["rays_shadow"], st["rays_masked"]], "oracle_threads": threads}
for rule in (1, 0):
    orc.set_tie_rule(rule)
    t0 = time.time()
    ref, ost = orc.Oracle(sc, spp=spp, pps=1, depth=9).render(rng=orc.RNG_COUNTER, seed=1, threads=threads)
    d = film[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)
    out["device_tie_rule" if rule else "reference_tie_rule"] = {
        "oracle_s": time.time() - t0, "oracle_rays": [ost["rays_closest"], ost["rays_shadow"], ost["rays_masked"]],
        "max_pixel_l2": float(np.sqrt((d * d).sum(-1)).max()),
        "pixels_differing": int((film[..., :3].view(np.uint32) != ref[..., :3].view(np.uint32)).any(-1).sum())}
    print(json.dumps(out), flush=True)
orc.set_tie_rule(0)
