#!/usr/bin/env python3
"""bench.py — Mrays/s of the gfx950 path-tracing device on BASELINE.json's configuration.

A step = one whole frame (xpu_t::start ... join, the reference's own "Rendering time" bracket,
src/core.cpp:158-177) of the synthetic workload:
  N=1 : Soup(100k) 1280x720 256 spp, depth 9, pps 1   (BASELINE.json configs[1])
  N>1 : the same frame, its 32x32 tiles interleaved over the ranks (tile (tx, ty) -> rank (tx + 3 ty) % N), every rank
        accumulating into its own zero-initialised device film, one RCCL reduce(sum) of the film to
        rank 0 inside the timed region ("strong" scaling: total work is fixed).
value = rays traced by ALL ranks (closest-hit + non-masked shadow rays) / max-over-ranks time.
Scene upload and BVH build (xpu_t::preprocess) happen before the timed region: inputs are HBM-resident.

Extra objects on the JSON line (N = 1 only):
  value_host_film  the same frame timed through the host film sink (the xpu_t boundary's add_tile: 14.7 MB over PCIe per frame),
               the rate the reference's own start...join bracket would see; `value` keeps the film in HBM.
  roofline     of the dominant kernel, k_trace (closest-hit rays of a step + shadow rays of the previous step in one persistent
               launch; time = HIP events on the device's own stream around every launch of the timed steps).
               `frac` is WORK-based (bound "valu"): the node visits and triangle tests the frame needs (counted by the instrumented
               build, libphx_hip_count.so, on the same frame) priced at the rate the chip runs k_trace's own arithmetic with nothing
               else in the way (scripts/micro/valu_mix.hip, profiles/r*_valu_mix.json) = minimum ALU time / k_trace time measured
               in THIS run.  `diagnostics` are counter rates, each taken from ONE committed capture (profiles/r*_<tag>_pmc.json)
               and divided by the kernel time of THAT capture's --kernel-trace --stats pass (kernel_ms_stats_pass), never by this
               run's time:
                 valu_issue    VALU wave-instructions issued vs 256 CUs x 4 SIMDs x 1 per 2 clocks (rewards wasted instructions:
                               a diagnostic, not the fraction)
                 vector_l1     vector-L1 lane addresses vs the measured 1.7 per clock and CU (scripts/micro/l1_gather.hip)
                 l2            L1->L2 read requests x 64 B vs 34.5 TB/s
                 hbm           FETCH_SIZE x 2 + WRITE_SIZE (separate PMC passes, MI355X_MICROARCH.md) vs 8 TB/s; `traffic` = those
                               bytes per launch
               `algorithmic_ref_layout` keeps SURVEY 8(d)'s figure (bytes per ray in the REFERENCE's 288-B node / 384-B packet
               layout, V_n and V_l from the CPU restatement's counters): a work-normalised rate, not a bound.
  cpu_baseline the CPU restatement (oracle/, kind "port") on this box's host cores: a warm thread pool renders tiles of the same
               frame (counter RNG) for >= 10 s; thread start-up and per-thread stream construction are outside the clock.
  secondary    the same measurement on Soup(1 M) (the north star's target scene), on the whole BASELINE config-4 frame
               (Soup(10 M), 3840x2160, 256 spp) on this one GPU, and on the declared stand-ins for BASELINE configs 3 and 5 (no BMW
               scene ships with the reference): the 16-recipe multi_material_soup(500 000) at 1920x1080, 256 spp and at 3840x2160,
               64 of 4096 spp — those two carry a roofline of the general-closure shade kernel, k_shade_g.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth (MI355X_MICROARCH.md, L2)
CLOCK_HZ = 2.4e9        # max clock
CUS = 256
L1_ADDR_PER_CLK_CU = 1.7  # measured ceiling of the vector L1: lane addresses per clock and CU (scripts/micro/l1_gather.hip)
COUNT_LIB = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_count.so")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--triangles", type=int, default=100000)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--height", type=int, default=720)
    p.add_argument("--spp", type=int, default=256)
    p.add_argument("--depth", type=int, default=9)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--samples-in-flight", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline, the instrumented pass and the secondary workloads")
    p.add_argument("--no-secondary", action="store_true", help="skip the Soup(1 M) and config-4 records")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="length of the headline CPU baseline run")
    p.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel rendered per tile visit in the CPU baseline")
    p.add_argument("--bvh-builder", choices=["auto", "host", "device"], default="auto",
                   help="auto (host binned SAH up to 2 M triangles, device LBVH above), host, device")
    p.add_argument("--force-dist", action="store_true", help="use torch.distributed + the film reduce even at N=1")
    p.add_argument("--one-sink", action="store_true", help="do not time the frames a second time through the other film sink (profiling captures: one frame per run)")
    p.add_argument("--host-film", action="store_true", help="N=1: hand the frame to the host film sink (PCIe inside the timed region) instead of a device film")
    return p.parse_args()


# ---- host CPUs ----------------------------------------------------------------------------------------------
def host_cpus():
    """-> (hardware threads visible, physical cores, CPU share of this process): what os.cpu_count() hides"""
    logical = os.cpu_count() or 1
    try:
        logical = min(logical, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except Exception:
        pass
    physical = min(len(cores), logical) if cores else logical
    share = float(logical)
    try:  # cgroup v2 CPU quota of the container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            share = min(share, float(q) / float(per))
    except Exception:
        pass
    return logical, physical, share


def frame_tiles(W, H):
    return [(x, y, min(32, W - x), min(32, H - y)) for y in range(0, H, 32) for x in range(0, W, 32)]


def cpu_baseline(scene, args, seconds=None, thread_counts=None):
    """The CPU restatement on this box's host cores: a warm pool of N workers pulls 32x32 tiles of the same frame from one
    atomic cursor (the reference's threading unit, src/xpu/cpu.cpp:223-238) for at least `seconds`, counter RNG,
    `cpu_spp` samples per tile visit.  Returns (cpu_baseline object, reference-layout visits per ray)."""
    from oracle import oracle as orc
    seconds = args.cpu_seconds if seconds is None else seconds
    logical, physical, share = host_cpus()
    O = orc.Oracle(scene, spp=args.spp, pps=1, depth=args.depth)
    tiles = frame_tiles(args.width, args.height)
    tiles = tiles[::7] + tiles[1::7] + tiles[2::7] + tiles[3::7] + tiles[4::7] + tiles[5::7] + tiles[6::7]  # spread over the film
    share_threads = max(1, min(logical, int(round(share))))
    if thread_counts is None:
        # 1, 8, the container's CPU share, 64, the physical cores, all hardware threads; the two candidates for the headline
        # (the CPU share, and 64 threads — over-subscribing a quota usually wins) run for the full `seconds`
        plan = {1: 0.3, 8: 0.3, share_threads: 1.0, 64: 1.0, physical: 0.4, logical: 0.4}
        thread_counts = {nt: max(w, plan.get(nt, 0)) for nt, w in plan.items() if 1 <= nt <= logical}
    elif not isinstance(thread_counts, dict):
        thread_counts = {nt: 1.0 for nt in thread_counts}
    runs = []
    for nt in sorted(thread_counts):
        secs = max(2.0 if seconds >= 2.0 else seconds, seconds * thread_counts[nt])
        st = O.bench(nt, secs, seed=args.seed, tiles=tiles, sample_end=args.cpu_spp)
        rays = st["rays_closest"] + st["rays_shadow"]
        runs.append({"threads": nt, "Mrays_per_s": rays / st["seconds"] / 1e6, "seconds": st["seconds"], "rays": rays, "stats": st})
    one = next((r for r in runs if r["threads"] == 1), None)
    for r in runs:
        r["efficiency_vs_1_thread"] = (r["Mrays_per_s"] / (one["Mrays_per_s"] * r["threads"])) if one else None
    full = [r for r in runs if r["seconds"] >= 0.99 * seconds] or runs  # the headline comes from a run of the full length
    best = max(full, key=lambda r: r["Mrays_per_s"])
    out = {
        "value": best["Mrays_per_s"], "unit": "Mrays/s", "cores": best["threads"], "kind": "port",
        "sample": f"32x32 tiles of the same frame x {args.cpu_spp} spp per visit, counter RNG, warm pool of {best['threads']} threads for "
                  f"{best['seconds']:.1f} s ({best['rays']} rays); construction and thread start-up are outside the clock",
        "host": {"hardware_threads": logical, "physical_cores": physical, "cpu_share": share},
        "scaling": [{k: r[k] for k in ("threads", "Mrays_per_s", "seconds", "rays", "efficiency_vs_1_thread")} for r in runs],
        # context, not measured here: the AVX2 reference itself traced its own 100 k soup at 1.30 Mrays/s on one thread and 3.57 Mrays/s
        # on eight in the survey's container; the port traces with the same 8-lane AVX2 arithmetic (oracle/obvh.h)
        "reference_context": "BASELINE.md: reference AVX2 build, its 100 k soup 1280x720 4 spp: 1.30 Mrays/s (1 thread), 3.57 Mrays/s (8 threads)",
    }
    st = best["stats"]
    vn_c = st["node_visits_closest"] / max(1, st["rays_closest"]); vl_c = st["packet_visits_closest"] / max(1, st["rays_closest"])
    vn_s = st["node_visits_shadow"] / max(1, st["rays_shadow"]); vl_s = st["packet_visits_shadow"] / max(1, st["rays_shadow"])
    visits = {"closest": (vn_c, vl_c), "shadow": (vn_s, vl_s), "ref_bvh_nodes": st["bvh_nodes"], "ref_bvh_packets": st["bvh_packets"]}
    O.close()
    return out, visits


# ---- committed measurements bench.py cannot take itself -------------------------------------------------------
def workload_tag(kind, triangles, width, height, depth=9):
    """name of the committed PMC capture of this workload (profiles/r*_<tag>_pmc.json), None if there is none"""
    if depth != 9:
        return None
    return {("soup", 100000, 1280, 720): "100k", ("soup", 1000000, 1280, 720): "1M", ("soup", 10000000, 3840, 2160): "c4",
            ("zoo", 500000, 1920, 1080): "zoo", ("zoo", 500000, 3840, 2160): "zoo4k"}.get((kind, triangles, width, height))


def committed_valu_peak():
    """node tests / triangle tests per second of the chip running k_trace's own arithmetic alone (scripts/micro/valu_mix.hip)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_mix.json")))
    if not files:
        return None, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def count_work(triangles, width, height, spp, builder):
    """node visits / triangle tests of one frame from the instrumented build, in a child process (one library per process)"""
    if not os.path.exists(COUNT_LIB):
        return None
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "count_work.py"), "--triangles", str(triangles), "--width", str(width),
           "--height", str(height), "--spp", str(spp), "--builder", builder]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr.strip().splitlines()[-1] if r.stderr.strip() else f"rc {r.returncode}"}
    return json.loads(r.stdout.strip().splitlines()[-1])


def load_capture(tag):
    """the newest committed capture profiles/r*_<tag>_pmc.json (scripts/summarize_profile.py) -> (dict, relative path)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc.json"))) if tag else []  # rNN_<letter>_...: the name orders them
    if not files:
        return None, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def capture_diagnostics(cap, src, kernel, units_in_capture):
    """counter rates of `kernel` from ONE capture: every count is divided by the kernel's time in that capture's own
    --kernel-trace --stats pass.  units_in_capture = rays (k_trace) or shaded hits (k_shade*) of the captured command, if known."""
    e = cap["kernels"].get(kernel)
    kt = (cap.get("kernel_ms_stats_pass") or {}).get(kernel)
    if kt is None and kernel == "k_trace" and cap.get("k_trace_launch_ms"):
        kt = {"launches": len(cap["k_trace_launch_ms"]), "total_ms": sum(cap["k_trace_launch_ms"])}
    if not e or not kt:
        return None
    c, n = e["counters"], e["launches"]
    t = kt["total_ms"] * 1e-3 * (n / kt["launches"])  # seconds of the n launches the PMC passes summed over
    D = {"source": src, "kernel_ms_in_capture": kt["total_ms"], "launches_in_capture": kt["launches"], "pmc_launches": n}
    if "SQ_INSTS_VALU" in c:
        a = c["SQ_INSTS_VALU"] / t
        D["valu_issue"] = {"achieved": a / 1e9, "peak": CUS * 4 * CLOCK_HZ / 2 / 1e9, "unit": "G wave-instructions/s", "frac": a / (CUS * 4 * CLOCK_HZ / 2),
                           "lane_utilisation": e.get("valu_lane_utilisation"), "wave_cycles_waiting_frac": e.get("wave_cycles_waiting_frac")}
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        a = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / t
        D["vector_l1"] = {"achieved": a / 1e9, "peak": L1_ADDR_PER_CLK_CU * CUS * CLOCK_HZ / 1e9, "unit": "G lane addresses/s",
                          "frac": a / (L1_ADDR_PER_CLK_CU * CUS * CLOCK_HZ), "l1_hit_rate": e.get("l1_hit_rate"),
                          "ta_busy_frac": e.get("ta_busy_frac"), "td_busy_frac": e.get("td_busy_frac")}
    if "TCP_TCC_READ_REQ_sum" in c:
        a = c["TCP_TCC_READ_REQ_sum"] * 64.0 / t
        D["l2"] = {"achieved": a / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": a / 1e9 / L2_PEAK_GBS, "l2_hit_rate": e.get("l2_hit_rate")}
    if "hbm_bytes_per_launch_corrected" in e:
        a = e["hbm_bytes_per_launch_corrected"] * n / t
        D["hbm"] = {"achieved": a / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / 1e9 / HBM_PEAK_GBS,
                    "bytes_per_launch": e["hbm_bytes_per_launch_corrected"], "bytes_per_launch_uncorrected": e.get("hbm_bytes_per_launch_raw")}
        if units_in_capture:
            D["hbm"]["bytes_per_unit"] = e["hbm_bytes_per_launch_corrected"] * n / units_in_capture
    return D


def roofline(acc, steps, work, tag, ref_visits):
    """k_trace.  acc: rays and k_trace ms summed over `steps` timed frames of THIS run; work: count_work() of one frame (instrumented
    build, same frame); tag: which committed capture holds this workload's counters.  The camera rays are k_trace_primary's (its own
    kernel, its own HIP events): rays, work and time here are k_trace's alone."""
    nl = max(1, acc["launches"])
    t_frame = acc["closest_ms"] * 1e-3 / steps          # k_trace seconds per frame (HIP events, this run)
    t_launch = acc["closest_ms"] * 1e-3 / nl
    rays_c, rays_s = (acc["closest"] - acc.get("primary_rays", 0)) / steps, acc["shadow"] / steps
    roof = {"kernel": "k_trace", "launches_per_step": nl / steps, "avg_launch_ms": t_launch * 1e3,
            "rays_per_launch": (rays_c + rays_s) * steps / nl, "kernel_rays_per_s": (rays_c + rays_s) / t_frame,
            "bound": "valu", "achieved": None, "peak": None, "unit": "G node-visit equivalents/s", "frac": None, "traffic": None}
    peak, peak_src = committed_valu_peak()
    if work and "error" not in work and peak:
        nv = sum(work[k]["rays"] * (work[k]["node_visits_lds_per_ray"] + work[k]["node_visits_mem_per_ray"]) for k in ("closest", "shadow"))
        tt = sum(work[k]["rays"] * work[k]["tri_tests_per_ray"] for k in ("closest", "shadow"))
        t_min = nv / peak["node_tests_per_s"] + tt / peak["tri_tests_per_s"]
        roof.update({"achieved": nv / t_frame / 1e9, "peak": nv / t_min / 1e9, "frac": t_min / t_frame})
        roof["work"] = {"node_visits_per_frame": nv, "tri_tests_per_frame": tt, "min_alu_ms_per_frame": t_min * 1e3, "k_trace_ms_per_frame": t_frame * 1e3,
                        "peak_node_tests_per_s": peak["node_tests_per_s"], "peak_tri_tests_per_s": peak["tri_tests_per_s"], "peak_source": peak_src,
                        "lanes_per_node_block": work["wave"]["lanes_per_node_block"], "lanes_per_tri_block": work["wave"]["lanes_per_tri_block"],
                        "what": "node visits + triangle tests of this frame (instrumented build, this run) priced at the chip's rate for k_trace's own "
                                "arithmetic alone (all 64 lanes active, operands in registers) = minimum ALU time / k_trace time of this run"}
        # bytes this kernel's own layout moves per ray: 64-B nodelets through the L1 (LDS-staged ones are free), 48 B of a triangle
        # record, ray in (32 B closest, 48 B shadow incl. its beta*Li), result out (16 B hit record; 32 B radiance read-modify-write)
        dl = {}
        for k, stream in (("closest", 32 + 16), ("shadow", 48 + 32)):
            dl[k] = work[k]["node_visits_mem_per_ray"] * 64 + work[k]["tri_tests_per_ray"] * 48 + stream
        dev_bytes = rays_c * dl["closest"] + rays_s * dl["shadow"]
        roof["device_layout"] = {"bytes_per_ray": dl, "GBps_through_L1": dev_bytes / t_frame / 1e9,
                                 "visits_per_ray": {k: {x: work[k][x] for x in ("node_visits_lds_per_ray", "node_visits_mem_per_ray", "tri_tests_per_ray")} for k in ("closest", "shadow")}}
    elif work and "error" in work:
        roof["instrumented_pass_error"] = work["error"]
    cap, src = load_capture(tag)
    if cap:
        D = capture_diagnostics(cap, src, "k_trace", rays_c + rays_s)  # the captures render this same frame once
        if D:
            roof["diagnostics"] = D
            if "hbm" in D:
                roof["traffic"] = D["hbm"]["bytes_per_launch"]
    if ref_visits:
        (vn, vl), (vns, vls) = ref_visits["closest"], ref_visits["shadow"]
        b_ray, b_shadow = 56.0 + vn * 288.0 + vl * 384.0, 36.0 + vns * 288.0 + vls * 384.0
        roof["algorithmic_ref_layout"] = {
            "GBps": (rays_c * b_ray + rays_s * b_shadow) / t_frame / 1e9, "bytes_per_ray": {"closest": b_ray, "shadow": b_shadow},
            "visits_per_ray": {"closest": {"nodes": vn, "leaf_packets": vl}, "shadow": {"nodes": vns, "leaf_packets": vls}},
            "note": "SURVEY 8(d): bytes per ray in the REFERENCE's 288-B node / 384-B packet layout (CPU restatement's counters) over k_trace "
                    "time: a work-normalised rate, not a fraction of any ceiling (the device's tree is 10x smaller and L2-resident)"}
    return roof


# algorithmic bytes per shaded queue entry in the device's layout: in — hit 16, ray 32, path state 16,
# triangle record 64 (one 64-B element of the pool); out — next ray 32 + path state 16 for a survivor, 48 for an NEE ray
SHADE_BYTES_IN, SHADE_BYTES_SURVIVOR, SHADE_BYTES_NEE = 16 + 32 + 16 + 64, 32 + 16, 48


def shade_roofline(acc, steps, st, tag):
    """the shade/NEE/integrate kernel (k_shade_g for general closures): HBM-stream roofline.  One launch shades every entry of a
    ray queue: algorithmic bytes = entries x (hit + ray + state + triangle record) + survivors x (next ray + state) + NEE rays x 48,
    all counted by the device in this run."""
    nl = max(1, acc["shade_launches"])
    t = acc["shade_kernel_ms"] * 1e-3
    entries, nee = acc["closest"], acc["shadow"]
    survivors = max(0, acc["closest"] - acc["camera"])  # every closest-hit ray after the camera rays is a survivor of a shade launch
    alg = entries * SHADE_BYTES_IN + survivors * SHADE_BYTES_SURVIVOR + nee * SHADE_BYTES_NEE
    kernel = "k_shade_g" if st.get("shade_general") else "k_shade"
    roof = {"kernel": kernel, "bound": "hbm", "achieved": alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "launches_per_step": nl / steps, "avg_launch_ms": t * 1e3 / nl, "entries_per_launch": entries / nl,
            "algorithmic_bytes_per_launch": alg / nl, "algorithmic_bytes_per_entry": alg / max(1, entries),
            "shaded_entries_per_s": entries / t}
    cap, src = load_capture(tag)
    if cap:
        D = capture_diagnostics(cap, src, kernel, None)
        if D:
            roof["diagnostics"] = D
            if "hbm" in D:
                roof["traffic"] = D["hbm"]["bytes_per_launch"]
    res = os.path.join(ROOT, "profiles")
    files = sorted(glob.glob(os.path.join(res, "r*_kernel_resources.txt")))
    if files:
        for line in open(files[-1]):
            if line.startswith(kernel + "<") or line.startswith(kernel + " "):
                f = line.split()
                roof.setdefault("resources", {"source": os.path.relpath(files[-1], ROOT), "variants": []})["variants"].append(
                    {"kernel": " ".join(f[:-6]), "vgprs": int(f[-6]), "scratch_bytes": int(f[-3]), "waves_per_simd": int(f[-1])})
    return roof


def new_acc():
    return {"closest": 0, "shadow": 0, "camera": 0, "closest_ms": 0.0, "shade_ms": 0.0, "shade_kernel_ms": 0.0, "shade_launches": 0, "launches": 0, "frame_ms": 0.0,
            "primary_ms": 0.0, "primary_launches": 0, "primary_rays": 0}


def kernel_ms(acc, steps):
    """HIP-event milliseconds per step by kernel: k_trace_primary (camera rays), k_trace (every other ray), k_shade / k_shade_g, the rest"""
    return {"primary": acc["primary_ms"] / steps, "trace": acc["closest_ms"] / steps, "shade": acc["shade_kernel_ms"] / steps,
            "begin_pass_film": (acc["shade_ms"] - acc["shade_kernel_ms"]) / steps}


def primary_record(acc, steps, work):
    """k_trace_primary: the camera rays of every pass, one packet walk per 64-256 rays (its work is not k_trace's: reported beside the roofline)"""
    if not acc["primary_launches"]:
        return None
    t = acc["primary_ms"] * 1e-3
    rec = {"kernel": "k_trace_primary", "launches_per_step": acc["primary_launches"] / steps, "ms_per_step": acc["primary_ms"] / steps,
           "rays_per_step": acc["primary_rays"] / steps, "rays_per_s": acc["primary_rays"] / t}
    if work and "error" not in work and work.get("primary"):
        rec["packets"] = work["primary"]
    return rec


def add_stats(acc, st):
    acc["closest"] += st["rays_closest"]; acc["shadow"] += st["rays_shadow"]; acc["camera"] += st["camera_samples"]
    acc["closest_ms"] += st["closest_ms"]; acc["shade_ms"] += st["shade_ms"]; acc["launches"] += st["trace_launches"]
    acc["shade_kernel_ms"] += st["shade_kernel_ms"]; acc["shade_launches"] += st["shade_launches"]; acc["frame_ms"] += st["frame_ms"]
    acc["primary_ms"] += st["primary_ms"]; acc["primary_launches"] += st["primary_launches"]
    acc["primary_rays"] += st["camera_samples"] if st["primary_launches"] else 0


def make_scene(scenes, kind, triangles, width, height):
    return scenes.multi_material_soup(triangles, seed=1234, width=width, height=height) if kind == "zoo" else scenes.soup(triangles, seed=1234, width=width, height=height)


def run_workload(xpu, scenes, kind, triangles, width, height, spp, depth, seed, builder, steps, warmup, samples_in_flight=0):
    """one device, one scene, `steps` timed frames with the film in HBM, then `steps` more through the host film sink
    -> (value Mrays/s, ms per step, acc, last stats, preprocess s, scene, film, value through the host film)"""
    scene = make_scene(scenes, kind, triangles, width, height)
    import torch
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=depth, samples_in_flight=samples_in_flight,
                                         bvh_builder=builder, device_ordinal=torch.cuda.current_device()))
    t0 = time.time(); dev.preprocess(scene); pre = time.time() - t0
    tiles = xpu.Tiles.make(width, height, 32)
    film_dev = torch.zeros((height, width, 4), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    acc = new_acc()
    st = None
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        tiles.reset()
        dev.start(scene, xpu.FrameState(seed, tiles, None, device_film_ptr=film_dev.data_ptr())); dev.join()
        st = dev.stats()
        if i >= warmup:
            add_stats(acc, st)
    elapsed = time.perf_counter() - t0
    film = film_dev.cpu().numpy()
    del film_dev
    host = xpu.Film(width, height, 4)
    rays_h = 0
    for i in range(1 + steps):  # one warm-up frame sizes the pinned staging buffer
        if i == 1:
            t0 = time.perf_counter()
        tiles.reset()
        dev.start(scene, xpu.FrameState(seed, tiles, host, native_sink=True)); dev.join()
        if i >= 1:
            h = dev.stats(); rays_h += h["rays_closest"] + h["rays_shadow"]
    value_host = rays_h / (time.perf_counter() - t0) / 1e6
    dev.close()
    return (acc["closest"] + acc["shadow"]) / elapsed / 1e6, elapsed * 1e3 / steps, acc, st, pre, scene, film, value_host


def secondary_record(xpu, scenes, name, kind, triangles, width, height, spp, args, cpu_seconds):
    like = argparse.Namespace(triangles=triangles, width=width, height=height, depth=args.depth, spp=spp, seed=args.seed, cpu_spp=args.cpu_spp,
                              cpu_seconds=cpu_seconds)
    value, ms, acc, st, pre, scene, film, value_host = run_workload(xpu, scenes, kind, triangles, width, height, spp, args.depth, args.seed, "auto", steps=2, warmup=1)
    rec = {"workload": name, "value": value, "value_host_film": value_host, "unit": "Mrays/s", "ms_per_step": ms, "steps": 2, "rays_per_step": (acc["closest"] + acc["shadow"]) / 2,
           "bvh_bytes": st["bvh_bytes"], "bvh_build_ms": st["bvh_build_ms"], "preprocess_s": pre, "paths_in_flight": st["paths_in_flight"],
           "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]},
           "kernel_ms_per_step": kernel_ms(acc, 2), "film_finite": bool(np.isfinite(film).all())}
    tag = workload_tag(kind, triangles, width, height, args.depth)
    if kind == "zoo":
        rec["roofline"] = shade_roofline(acc, 2, st, tag)
        rec["roofline_k_trace"] = roofline(acc, 2, None, tag, None)
        rec["primary"] = primary_record(acc, 2, None)
        return rec
    work = count_work(triangles, width, height, spp, "auto")
    ref_visits = None
    if cpu_seconds > 0:
        logical, _, share = host_cpus()
        base, ref_visits = cpu_baseline(scene, like, seconds=cpu_seconds, thread_counts={1: 0.2, max(1, min(logical, int(round(share)))): 1.0, min(64, logical): 0.5})
        rec["cpu_baseline"] = base
        rec["gpu_over_cpu"] = value / base["value"]
    rec["roofline"] = roofline(acc, 2, work, tag, ref_visits)
    rec["primary"] = primary_record(acc, 2, work)
    return rec


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a ONE-GPU box (PHX_BENCH_REHEARSAL=1, never set by the driver): every rank uses GPU 0 and the
    # film reduce runs on gloo, because two RCCL ranks cannot share a device.  Everything else — tile shard, device films, the reduce,
    # max-over-ranks timing, the rays summed over ranks — is the code the real N-GPU run takes.  The number it prints is not a result.
    rehearsal = os.environ.get("PHX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    use_dist = world > 1 or args.force_dist
    dist = None
    # torch first: libphx_hip.so then binds to the HIP runtime torch already loaded (one runtime per process).  torch is plumbing
    # here: the film's device memory and, at N > 1, the process group.
    import torch
    torch.cuda.set_device(local_rank)
    if use_dist:
        from phosphorus_mk2_amd import dist as pdist
        dist = pdist.init_process_group("gloo" if rehearsal else "nccl", rank, world, None if rehearsal else torch.device("cuda", local_rank))
    from phosphorus_mk2_amd import scenes, xpu
    xpu.load_library()  # raises if the HIP extension is missing: no fallback

    scene = scenes.soup(args.triangles, seed=1234, width=args.width, height=args.height)
    opts = xpu.Options(samples_per_pixel=args.spp, paths_per_sample=1, path_depth=args.depth,
                       device_ordinal=local_rank, samples_in_flight=args.samples_in_flight,
                       bvh_builder=args.bvh_builder)
    dev = xpu.HipDevice.make(opts)
    t0 = time.time()
    dev.preprocess(scene)  # flatten + BVH build + upload: outside the timed region
    preprocess_s = time.time() - t0
    W, H = args.width, args.height
    tiles = xpu.Tiles.make(W, H, 32, rank, world)
    # The film stays in HBM inside the timed region (`value` is HBM-resident in, HBM-resident out); --host-film times the frame
    # through the host film sink instead — 14.7 MB over PCIe per frame, what a host that hands over a frame buffer sees: the
    # PCIe-inclusive rate DESIGN.md section 5 quotes beside `value`.
    film_host = None
    film_dev = None
    if use_dist or not args.host_film:
        film_dev = torch.zeros((H, W, 4), dtype=torch.float32, device=torch.device("cuda", local_rank))
    else:
        film_host = xpu.Film(W, H, 4)
    # N > 1: TWO frames in flight per rank.  A rank's share of a frame is 21 short launches, and every launch ends with a drain in which
    # the chip empties (17 % of the share at N = 8); a second device object on the same GPU — its own stream, queues
    # and tree — renders the NEXT frame meanwhile, so one frame's drains are filled by the other frame's launches (frames are
    # independent; every frame of the timed region starts after the opening barrier and is complete before the closing one).  The
    # film reduce of a finished frame (RCCL's stream) runs beside both.  Four films in turn; pending[b] = the reduce still reading film b.
    IN_FLIGHT = 2 if use_dist else 1
    devs, tile_sets = [dev], [tiles]
    if use_dist:
        dev2 = xpu.HipDevice.make(opts); dev2.preprocess(scene)
        devs.append(dev2); tile_sets.append(xpu.Tiles.make(W, H, 32, rank, world))
    films = [film_dev] + [torch.zeros_like(film_dev) for _ in range(3)] if use_dist else [film_dev]
    pending = [None] * len(films)
    frame_no = [0]

    def barrier():
        if use_dist:
            for b in range(len(films)):  # every film reduce in flight belongs to the frames before the barrier
                if pending[b] is not None:
                    pending[b].wait(); pending[b] = None
            dist.barrier()
        torch.cuda.synchronize()

    def start_frame(i):
        """N > 1: frame i goes to device i mod 2 and film i mod 4"""
        b = i % len(films)
        if pending[b] is not None:
            pending[b].wait(); pending[b] = None  # the reduce of four frames ago has read this film (orders torch's stream behind it)
        films[b].zero_()
        cleared = torch.cuda.Event(); cleared.record(); cleared.synchronize()  # the device renders on its own stream: wait for the zeros only
        tile_sets[i % IN_FLIGHT].reset()
        devs[i % IN_FLIGHT].start(scene, xpu.FrameState(args.seed, tile_sets[i % IN_FLIGHT], None, device_film_ptr=films[b].data_ptr()))

    def finish_frame(i):
        d = devs[i % IN_FLIGHT]
        d.join()  # join() synchronises the device's stream: the film is complete
        pending[i % len(films)] = pdist.reduce_film(films[i % len(films)], dst=0, async_op=True)  # the single film collective (RCCL over xGMI)
        return d.stats()

    def run_frames(n, acc=None):
        """n frames, at most IN_FLIGHT of them at a time; returns the stats of the last one"""
        st, first = None, frame_no[0]
        for i in range(first, first + n):
            if i - first >= IN_FLIGHT:
                st = finish_frame(i - IN_FLIGHT)
                if acc is not None:
                    add_stats(acc, st)
            start_frame(i)
        for i in range(max(first, first + n - IN_FLIGHT), first + n):
            st = finish_frame(i)
            if acc is not None:
                add_stats(acc, st)
        frame_no[0] = first + n
        return st

    def step():
        tiles.reset()
        if film_dev is not None:
            # no clearing: at world 1 the device's tiles cover (and overwrite) every pixel of the film
            dev.start(scene, xpu.FrameState(args.seed, tiles, None, device_film_ptr=film_dev.data_ptr()))
            dev.join()  # join() synchronises the device's stream
        else:
            dev.start(scene, xpu.FrameState(args.seed, tiles, film_host, native_sink=True))
            dev.join()
        return dev.stats()

    acc = new_acc()
    if use_dist:
        if args.warmup:
            run_frames(args.warmup)
        barrier()
        t0 = time.perf_counter()
        st = run_frames(args.steps, acc)
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st = step()
            add_stats(acc, st)
        barrier()
        elapsed = time.perf_counter() - t0
    # the same frames once more through the OTHER film sink (N = 1): `value` keeps the film in HBM, `value_host_film` hands every
    # frame to a host frame buffer as xpu_t's add_tile does (the reference's own start...join bracket, src/core.cpp:158-177)
    value_other = None
    if not use_dist and not args.one_sink:
        other_dev = film_dev is None
        if other_dev:
            film_other = torch.zeros((H, W, 4), dtype=torch.float32, device=torch.device("cuda", local_rank))
        else:
            film_other = xpu.Film(W, H, 4)
        rays_o = 0
        for i in range(1 + args.steps):
            if i == 1:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            tiles.reset()
            if other_dev:
                dev.start(scene, xpu.FrameState(args.seed, tiles, None, device_film_ptr=film_other.data_ptr()))
            else:
                dev.start(scene, xpu.FrameState(args.seed, tiles, film_other, native_sink=True))
            dev.join()
            if i >= 1:
                so = dev.stats(); rays_o += so["rays_closest"] + so["rays_shadow"]
        value_other = rays_o / (time.perf_counter() - t1) / 1e6
    # ... and once more with TWO frames in flight (a second device object, its own stream), as every rank of an N > 1 run does: the N = 1
    # number to hold a multi-GPU `value` against.  `value` itself stays one frame at a time, so that the HIP-event kernel times the
    # roofline is made of are those of kernels that had the chip to themselves.
    value_two_in_flight = None
    if not use_dist and not args.one_sink and film_dev is not None:
        dev2 = xpu.HipDevice.make(opts); dev2.preprocess(scene)
        pair = [(dev, tiles, film_dev), (dev2, xpu.Tiles.make(W, H, 32, rank, world), torch.zeros_like(film_dev))]
        def go(i):
            d, t, f = pair[i & 1]
            t.reset(); d.start(scene, xpu.FrameState(args.seed, t, None, device_film_ptr=f.data_ptr()))
        def done(i):
            d = pair[i & 1][0]
            d.join(); sp = d.stats()
            return sp["rays_closest"] + sp["rays_shadow"]
        go(0); go(1); done(0); done(1)  # untimed: both devices have rendered a frame
        torch.cuda.synchronize(); t2 = time.perf_counter()
        rays_p = 0
        for i in range(args.steps):
            if i >= 2:
                rays_p += done(i - 2)
            go(i)
        for i in range(max(0, args.steps - 2), args.steps):
            rays_p += done(i)
        torch.cuda.synchronize()
        value_two_in_flight = rays_p / (time.perf_counter() - t2) / 1e6
        dev2.close()
    rays_local = acc["closest"] + acc["shadow"]
    if use_dist:
        elapsed = pdist.max_over_ranks(elapsed, "cpu" if rehearsal else "cuda")
        rays_total = pdist.sum_over_ranks(rays_local, "cpu" if rehearsal else "cuda")
    else:
        rays_total = float(rays_local)

    if rank == 0:
        if use_dist:
            film_dev = films[(frame_no[0] - 1) % len(films)]  # the film of the last frame, reduced onto this rank
        film = film_dev.cpu().numpy() if film_dev is not None else film_host.data
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_total / elapsed / 1e6
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", **({"rehearsal": "all ranks on GPU 0, gloo reduce: not a scaling result"} if rehearsal else {}),
            "value_film": "host (PCIe inside the timed region, --host-film)" if film_dev is None else "hbm (device film: nothing crosses PCIe inside a step)",
            "value_host_film": value if film_dev is None else value_other, "value_hbm_film": value_other if film_dev is None else value,
            "value_two_frames_in_flight": value_two_in_flight,
            "config": {"workload": f"Soup({args.triangles}, seed 1234) {W}x{H} {args.spp} spp depth {args.depth} pps 1, "
                                   "1 emissive quad, Lambert 0.73 (BASELINE.json configs[1])",
                       "tiles": "32x32, tile (tx, ty) -> rank (tx + 3 ty) % n_gpus", "film_collective": "reduce(sum) to rank 0, asynchronous, beside the rendering of the next frames" if use_dist else "none",
                       "frames_in_flight": IN_FLIGHT, **({"frames_in_flight_note": "N > 1: two device objects per rank render alternate frames on their own streams (one frame's "
                                                          "drain phases are filled by the other's launches); kernel_ms_per_step are HIP-event times of overlapped kernels"} if use_dist else {}),
                       "rays_per_step": rays_total / args.steps, "camera_samples_per_step": W * H * args.spp,
                       "preprocess_s": preprocess_s, "bvh_builder": args.bvh_builder, "bvh_build_ms": st["bvh_build_ms"],
                       "bvh_bytes": st["bvh_bytes"], "paths_in_flight": st["paths_in_flight"],
                       "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]},
                       "film_mean": float(film[..., :3].mean()), "film_finite": bool(np.isfinite(film).all())},
        }
        for d_ in devs:
            d_.close()
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            work = count_work(args.triangles, W, H, args.spp, args.bvh_builder)
            base, ref_visits = cpu_baseline(scene, args)
            out["cpu_baseline"] = base
            out["config"]["gpu_over_cpu"] = value / base["value"]
            out["roofline"] = roofline(acc, args.steps, work, workload_tag("soup", args.triangles, W, H, args.depth), ref_visits)
            out["primary"] = primary_record(acc, args.steps, work)
            if not args.no_secondary and (args.triangles, W, H, args.spp) == (100000, 1280, 720, 256):
                sec = []
                sec.append(secondary_record(xpu, scenes, "Soup(1000000, seed 1234) 1280x720 256 spp depth 9 (north star's target scene)",
                                            "soup", 1000000, 1280, 720, 256, args, cpu_seconds=args.cpu_seconds))
                sec.append(secondary_record(xpu, scenes, "Soup(10000000, seed 1234) 3840x2160 256 spp depth 9: the whole BASELINE config-4 frame on ONE GPU",
                                            "soup", 10000000, 3840, 2160, 256, args, cpu_seconds=0))
                sec.append(secondary_record(xpu, scenes, "stand-in for BASELINE config 3 (no BMW scene ships with the reference): multi_material_soup(500000), 16 closure "
                                            "recipes over all 7 lobe models, 1920x1080, 256 of 1024 spp, depth 9, whole frame on one GPU",
                                            "zoo", 500000, 1920, 1080, 256, args, cpu_seconds=0))
                sec.append(secondary_record(xpu, scenes, "stand-in for BASELINE config 5: the same 16-recipe scene at 3840x2160, 64 of 4096 spp, depth 9, whole frame on one GPU "
                                            "(the shading-bound regime: k_shade_g)",
                                            "zoo", 500000, 3840, 2160, 64, args, cpu_seconds=0))
                out["secondary"] = sec
        else:
            out["cpu_baseline"] = None
            out["roofline"] = roofline(acc, args.steps, None, workload_tag("soup", args.triangles, W, H, args.depth), None) if world == 1 else {
                "bound": None, "kernel": "k_trace", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "note": "roofline and cpu_baseline are reported at N = 1"}
        out["config"]["kernel_ms_per_step"] = {**kernel_ms(acc, args.steps), "frame": acc["frame_ms"] / args.steps}
        print(json.dumps(out))
    else:
        for d_ in devs:
            d_.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
