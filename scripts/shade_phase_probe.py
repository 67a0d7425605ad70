"""Run ON the GPU box after `make -C phosphorus_mk2_amd/csrc variant NAME=shtime EXTRA=-DPHX_SHADE_TIMING=1`: where k_shade_g's wave time goes.
s_memtime at the phase boundaries of every shading round, per wave, everything in flight waited for at each boundary (so a phase is charged the
latency of what it asked for); summed over waves and launches.  Prints a markdown table.  (Round 4's two-phase version: sort 16 %, rounds 84 %.)
    python scripts/shade_phase_probe.py [width height spp [room]]"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["PHX_LIB"] = os.environ.get("PHX_PROBE_LIB") or os.path.join(os.getcwd(), "phosphorus_mk2_amd", "libphx_hip_shtime.so")  # PHX_PROBE_LIB: another probe build (e.g. the block_append2 kernel, -DPHX_SHADE_RING=0)
from phosphorus_mk2_amd import scenes, xpu
W, H, SPP = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 256)))
ROOM = len(sys.argv) >= 5 and sys.argv[4] == "room"  # the closed mesh room with the per-hit glass (BASELINE configs 3 / 5 on mesh geometry) instead of the stand-in soup
sc = scenes.bmw_showroom(500000, width=W, height=H) if ROOM else scenes.multi_material_soup(500000, width=W, height=H)
film, st = xpu.render(sc, spp=SPP, seed=1, native_sink=True)
ph = list(st["stack_pushes"])
names = ["sort of the window by material (hit record + material gather, LDS histogram, scan, scatter of the permutation)",
         "loads landed: permuted index, hit record, ray, path state, triangle record, normals; emission added",
         "next-event estimation: light sample, bsdf_f, li",
         "roulette + bsdf_sample + path-state store",
         "append (ring build: slot reservation in LDS, records to LDS, commit, the occasional flush of a block; -DPHX_SHADE_RING=0: two barriers + the workgroup's two atomics)",
         "stores of the next ray / shadow ray (waited for) + end-of-window barrier"]
tot = float(sum(ph[:6]))
print(f"k_shade_g phase probe: {sc.name} {W}x{H} {SPP} spp; shade kernel {st['shade_kernel_ms']:.2f} ms (probe build: every phase boundary waits for everything in flight), "
      f"{ph[7]} wave-windows, {ph[6]} wave-rounds, {st['rays_closest']} entries shaded\n")
print("| phase | share of wave time | ticks per wave-round |")
print("|---|---|---|")
for n, t in zip(names, ph[:6]):
    print(f"| {n} | {t / tot:.3f} | {t / max(1, ph[6]):.0f} |")
print(f"\nround trip of one queue-counter atomic (wave 0 / 1, lane 0; s_memtime around the returning atomicAdd): {st['idle_lane_iters'] / max(1, st['tri_pending_lane_iters']):.0f} ticks over {st['tri_pending_lane_iters']} atomics")
