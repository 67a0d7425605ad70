#!/usr/bin/env python3
"""bench.py — Mrays/s of the gfx950 path-tracing device on BASELINE.json's configuration.

    python bench.py --gpus N --steps K --warmup W

A step = one whole frame (xpu_t::start ... join, the reference's own "Rendering time" bracket, src/core.cpp:158-177) of the
synthetic workload Soup(100k) 1280x720 256 spp, depth 9, pps 1 (BASELINE.json configs[1]).  ONE protocol at every N:
  value   ONE frame in flight.  Its 32x32 tiles are interleaved over the ranks (tile (tx, ty) -> rank (tx + 3 ty) % N), every rank
          renders into its own device film (HBM) and — N > 1 — the films are summed onto rank 0 with ONE reduce(sum) (RCCL over xGMI)
          INSIDE the bracket, before the next frame starts ("strong" scaling: total work is fixed).
          value = rays traced by ALL ranks (closest-hit + non-masked shadow rays) / max-over-ranks time of K frames.
  value_two_frames_in_flight
          the same K frames with a second device object per rank (own stream, queues and tree) rendering alternate frames and the
          film reduce asynchronous beside them: a throughput, not a frame latency; reported at every N, never called `value`.
  value_host_film (N = 1)
          the frame handed to a host frame buffer through the xpu_t boundary's add_tile sink (14.7 MB over PCIe per frame).
Scene upload and BVH build (xpu_t::preprocess) happen before the timed region: inputs are HBM-resident.

--gpus N without a torch.distributed environment starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
process (before this process touches the GPU), relays rank 0's line and exits with the child's code; it fails when fewer than N
devices are visible.  Under the driver's own torchrun launch WORLD_SIZE must equal --gpus.

Output: the LAST stdout line is ONE compact JSON object (< 4 KB: metric, value, roofline, cpu_baseline, four secondary workloads reduced
to value / ms / roofline fraction).  The FULL record — every operand of the roofline, counter diagnostics, CPU scaling table, complete
secondary records — goes to --full-json (default gpurun_out/bench_full.json).  At N = 1 the full record carries:
  roofline     of the dominant kernel, k_trace (closest-hit rays of a step + shadow rays of the previous step in one persistent
               launch; time = HIP events on the device's own stream around every launch of the timed steps).
               `frac` is WORK-based (bound "valu"): the node visits and triangle tests the frame needs (counted by the instrumented
               build, libphx_hip_count.so, on the same frame) priced at the rate the chip runs k_trace's own arithmetic with nothing
               else in the way (scripts/micro/valu_mix.hip, profiles/r*_valu_mix.json) = minimum ALU time / k_trace time measured
               in THIS run.  `diagnostics` are counter rates, each taken from ONE committed capture (profiles/r*_<tag>_pmc.json)
               and divided by the kernel time of THAT capture's --kernel-trace --stats pass, never by this run's time
               (valu_issue, vector_l1, l2, hbm = FETCH_SIZE x 2 + WRITE_SIZE vs 8 TB/s; `traffic` = those bytes per launch).
               `stream_GBps` = the compulsory queue bytes of this kernel (48 B per closest-hit ray, 80 B per shadow ray) over its
               live HIP-event time: the HBM-roofline reading of the same launches (the tree is L2-resident, so it is far from 8 TB/s).
  cpu_baseline the CPU restatement (oracle/, kind "port") on this box's host cores: a warm thread pool renders tiles of the same
               frame (counter RNG) for >= 10 s; thread start-up and per-thread stream construction are outside the clock.
  secondary    the same measurement on Soup(1 M) (the north star's target scene), on the whole BASELINE config-4 frame
               (Soup(10 M), 3840x2160, 256 spp) on this one GPU, and on the declared stand-ins for BASELINE configs 3 and 5 (no BMW
               scene ships with the reference): the 16-recipe multi_material_soup(500 000) at 1920x1080 with 1024 spp and at
               3840x2160 with 4096 spp, both whole frames at BASELINE's sizes — those two carry a roofline of the general-closure
               shade kernel, k_shade_g — and on the same two configurations with MESH geometry in a closed room
               (scenes.bmw_showroom(500 000): 24 tessellated spheres, triangle sizes over three decades, the 16 recipes + sharp and
               frosted glass, 7.3 rays per camera sample where the open soup has 1.9).
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
L1_ADDR_PER_CLK_CU = 1.7  # measured ceiling of the vector L1 for an L1-RESIDENT table: lane addresses per clock and CU (scripts/micro/l1_gather.hip)
# the same gather over larger tables (profiles/r04_l_pair_help.log, r04_m_l1_gather.log): the rate a uniform gather over a pool of the tree's size reaches
GATHER_PER_CLK_CU_BY_TABLE = {"16 KB (L1-resident)": 1.67, "2.4 MB (inside one XCD's 4 MB L2)": 1.36, "7.7 MB (the 100 k soup's pool)": 0.81, "77 MB (the 1 M soup's pool)": 0.38}
COUNT_LIB = os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip_count.so")


def parse(argv=None):
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
    p.add_argument("--one-sink", action="store_true", help="time `value` only: no host-film pass, no two-frames-in-flight pass (profiling captures: one frame per run)")
    p.add_argument("--full-json", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"), help="where the full record goes (the stdout line is the compact one)")
    return p.parse_args(argv)


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
            ("zoo", 500000, 1920, 1080): "zoo", ("zoo", 500000, 3840, 2160): "zoo4k", ("room", 500000, 1920, 1080): "room", ("room", 500000, 3840, 2160): "room"}.get((kind, triangles, width, height))


def priced_source_hash():
    """hash of the bvh8.h functions the roofline prices (scripts/src_hash.py): the tree's, now"""
    from scripts import src_hash
    return src_hash.priced_source_hash()


def committed_valu_peak():
    """node tests / triangle tests per second of the chip running k_trace's own arithmetic alone (scripts/micro/valu_mix.hip)
    -> (peak dict or None, path, reason it was refused or None).  A peak is only used when it was measured on the sources of THIS
    tree: the micro-benchmark records the hash of the functions it timed, and a peak without one, or with another, is refused."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_mix.json")))
    if not files:
        return None, None, "no profiles/r*_valu_mix.json"
    peak, src = json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    try:
        now = priced_source_hash()
    except Exception as e:  # the sources are part of the repo: this is a broken checkout
        return None, src, f"cannot hash csrc/bvh8.h ({e})"
    if peak.get("src_hash") != now:
        return None, src, (f"stale peak: {src} was measured on node/triangle-test sources with hash {peak.get('src_hash')}, the tree's is {now} "
                           "(re-run scripts/capture_valu_mix.sh on the GPU box and commit its output)")
    return peak, src, None


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
    # per-clock ceilings are priced at the clock the kernel really ran at in the capture (GRBM_GUI_ACTIVE is summed over the 8 XCDs:
    # cycles / 8 / kernel time; k_trace runs at ~2.2 GHz, not the 2.4 GHz maximum), falling back to the maximum when the capture has no such pass
    clock = c["GRBM_GUI_ACTIVE"] / 8.0 / t if c.get("GRBM_GUI_ACTIVE") else CLOCK_HZ
    D["clock_hz"] = clock
    D["clock_source"] = "GRBM_GUI_ACTIVE / 8 / kernel time of the capture" if c.get("GRBM_GUI_ACTIVE") else "maximum clock (no GRBM_GUI_ACTIVE pass in the capture)"
    if "SQ_INSTS_VALU" in c:
        a = c["SQ_INSTS_VALU"] / t
        D["valu_issue"] = {"achieved": a / 1e9, "peak": CUS * 4 * clock / 2 / 1e9, "unit": "G wave-instructions/s", "frac": a / (CUS * 4 * clock / 2),
                           "lane_utilisation": e.get("valu_lane_utilisation"), "wave_cycles_waiting_frac": e.get("wave_cycles_waiting_frac")}
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        a = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / t
        D["vector_l1"] = {"achieved": a / 1e9, "peak": L1_ADDR_PER_CLK_CU * CUS * clock / 1e9, "unit": "G lane addresses/s",
                          "frac": a / (L1_ADDR_PER_CLK_CU * CUS * clock), "l1_hit_rate": e.get("l1_hit_rate"),
                          "ta_busy_frac": e.get("ta_busy_frac"), "td_busy_frac": e.get("td_busy_frac"),
                          "uniform_gather_lane_addresses_per_clk_cu_by_table": GATHER_PER_CLK_CU_BY_TABLE}
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


def roofline(acc, steps, work, tag, ref_visits, capture_is_this_frame=True):
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
    # the HBM reading of the same launches: compulsory queue bytes (closest-hit ray 32 in + 16 hit record out; shadow ray 48 in + 32 of
    # radiance read-modify-write) over the live HIP-event time.  The 8 MB tree is L2-resident, so this is far from the 8 TB/s peak.
    stream_bytes = rays_c * 48.0 + rays_s * 80.0
    roof["stream_GBps"] = stream_bytes / t_frame / 1e9
    roof["stream_bytes_per_launch"] = stream_bytes * steps / nl
    roof["stream_frac_of_hbm_peak"] = stream_bytes / t_frame / 1e9 / HBM_PEAK_GBS
    peak, peak_src, peak_refused = committed_valu_peak()
    roof["timing"] = ("HIP events on the device's own stream, ONE event between consecutive launches: a launch's time includes the few "
                      "microseconds since the previous kernel ended")
    cap, src = load_capture(tag)
    # (bytes per ray need the rays OF THE CAPTURE: the soup captures render this same frame once; the zoo / room captures render it at 256 spp)
    D = capture_diagnostics(cap, src, "k_trace", (rays_c + rays_s) if capture_is_this_frame else None) if cap else None
    if work and "error" not in work and not peak:
        roof["frac_unavailable"] = peak_refused
    if work and "error" not in work and peak:
        nv = sum(work[k]["rays"] * (work[k]["node_visits_lds_per_ray"] + work[k]["node_visits_mem_per_ray"]) for k in ("closest", "shadow"))
        tt = sum(work[k]["rays"] * work[k]["tri_tests_per_ray"] for k in ("closest", "shadow"))
        # The peak is a rate PER SECOND measured at the micro-benchmark's shader clock (2.3-2.4 GHz: nothing but VALU work); k_trace itself runs
        # at ~2.2 GHz (GRBM_GUI_ACTIVE of its own capture).  The ceiling of the arithmetic AT THE CLOCK THE KERNEL RUNS AT is the per-clock rate
        # x that clock; without a capture of this workload the micro-benchmark's own clock stands and the record says so.
        k_clock, p_clock = (D or {}).get("clock_hz"), peak.get("clock_hz")
        scale = (k_clock / p_clock) if (k_clock and p_clock and (D or {}).get("clock_source", "").startswith("GRBM")) else 1.0
        node_rate, tri_rate = peak["node_tests_per_s"] * scale, peak["tri_tests_per_s"] * scale
        t_min = nv / node_rate + tt / tri_rate
        roof.update({"achieved": nv / t_frame / 1e9, "peak": nv / t_min / 1e9, "frac": t_min / t_frame})
        roof["work"] = {"node_visits_per_frame": nv, "tri_tests_per_frame": tt, "min_alu_ms_per_frame": t_min * 1e3, "k_trace_ms_per_frame": t_frame * 1e3,
                        "peak_node_tests_per_s": node_rate, "peak_tri_tests_per_s": tri_rate, "peak_source": peak_src,
                        "peak_src_hash": peak.get("src_hash"), "peak_clock_hz": p_clock, "kernel_clock_hz": k_clock if scale != 1.0 else None, "peak_clock_scale": scale,
                        "peak_node_tests_per_s_at_micro_clock": peak["node_tests_per_s"], "peak_tri_tests_per_s_at_micro_clock": peak["tri_tests_per_s"],
                        "peak_asm_check": peak.get("asm_check"),
                        "lanes_per_node_block": work["wave"]["lanes_per_node_block"], "lanes_per_tri_block": work["wave"]["lanes_per_tri_block"],
                        "what": "node visits + triangle tests of this frame (instrumented build, this run) priced at the chip's rate for k_trace's own "
                                "arithmetic alone (all 64 lanes active, operands in registers; the micro-benchmark's loop is checked against the kernel's "
                                "assembly: same 48 conversions, VALU count within 5 %), at the shader clock k_trace itself ran at in its capture "
                                "= minimum ALU time / k_trace time of this run"}
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


# algorithmic bytes per shaded queue entry in the device's layout: in — hit 16, ray 32, path state 16 (a queue
# record beside the ray since round 6), the hit triangle's shade record 16 (geometric normal + material word; until round 6 the 64-B
# triangle record of the pool); out — next ray 32 + path state 16 for a survivor, 48 for an NEE ray.  The camera rays' entries (the
# first launch of a pass) read no ray and no path state — both are rebuilt — and write the path's radiance (16).  Vertex normals of
# smooth faces (36 per hit on mesh scenes) are not counted.
SHADE_BYTES_IN, SHADE_BYTES_IN_CAMERA, SHADE_BYTES_SURVIVOR, SHADE_BYTES_NEE = 16 + 32 + 16 + 16, 16 + 16 + 16, 32 + 16, 48


def shade_roofline(acc, steps, st, tag):
    """the shade/NEE/integrate kernel (k_shade_g for general closures): HBM-stream roofline.  One launch shades every entry of a
    ray queue: algorithmic bytes = entries x (hit + ray + state + shade record) + survivors x (next ray + state) + NEE rays x 48,
    all counted by the device in this run."""
    nl = max(1, acc["shade_launches"])
    t = acc["shade_kernel_ms"] * 1e-3
    entries, nee = acc["closest"], acc["shadow"]
    survivors = max(0, acc["closest"] - acc["camera"])  # every closest-hit ray after the camera rays is a survivor of a shade launch
    alg = acc["camera"] * SHADE_BYTES_IN_CAMERA + survivors * SHADE_BYTES_IN + survivors * SHADE_BYTES_SURVIVOR + nee * SHADE_BYTES_NEE
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
    if kind == "room":  # the closed showroom: mesh geometry, three decades of triangle sizes, 16 recipes + two glass materials (per-hit closure weights)
        return scenes.bmw_showroom(triangles, width=width, height=height)
    return scenes.multi_material_soup(triangles, seed=1234, width=width, height=height) if kind == "zoo" else scenes.soup(triangles, seed=1234, width=width, height=height)


def run_workload(xpu, scenes, kind, triangles, width, height, spp, depth, seed, builder, steps, warmup, samples_in_flight=0, shard=(0, 1), host_pass=True, warmup_shard=None):
    """one device, one scene, `steps` timed frames with the film in HBM, then `steps` more through the host film sink
    -> (value Mrays/s, ms per step, acc, last stats, preprocess s, scene, film, value through the host film)"""
    scene = make_scene(scenes, kind, triangles, width, height)
    import torch
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=depth, samples_in_flight=samples_in_flight,
                                         bvh_builder=builder, device_ordinal=torch.cuda.current_device()))
    t0 = time.time(); dev.preprocess(scene); pre = time.time() - t0
    tiles = xpu.Tiles.make(width, height, 32, shard[0], shard[1])  # shard = (rank, world): every world-th tile of the film (diagonal interleave)
    # warm-up frames may render a SHARE of the film only (a 7 s frame is warmed up by 1/64 of its tiles: queues allocated, kernels loaded)
    warm_tiles = xpu.Tiles.make(width, height, 32, warmup_shard[0], warmup_shard[1]) if warmup_shard else tiles
    film_dev = torch.zeros((height, width, 4), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    acc = new_acc()
    st = None
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        tq = tiles if i >= warmup else warm_tiles
        tq.reset()
        dev.start(scene, xpu.FrameState(seed, tq, None, device_film_ptr=film_dev.data_ptr())); dev.join()
        st = dev.stats()
        if i >= warmup:
            add_stats(acc, st)
    elapsed = time.perf_counter() - t0
    film = film_dev.cpu().numpy()
    del film_dev
    value_host = None
    if host_pass:
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


def secondary_record(xpu, scenes, name, kind, triangles, width, height, spp, args, cpu_seconds, shard=(0, 1), steps=2, warmup=1, host_pass=True, warmup_shard=None):
    like = argparse.Namespace(triangles=triangles, width=width, height=height, depth=args.depth, spp=spp, seed=args.seed, cpu_spp=args.cpu_spp,
                              cpu_seconds=cpu_seconds)
    value, ms, acc, st, pre, scene, film, value_host = run_workload(xpu, scenes, kind, triangles, width, height, spp, args.depth, args.seed, "auto", steps=steps, warmup=warmup, shard=shard,
                                                                    host_pass=host_pass, warmup_shard=warmup_shard)
    rec = {"workload": name, "value": value, "value_host_film": value_host, "unit": "Mrays/s", "ms_per_step": ms, "steps": steps, "warmup": warmup, "rays_per_step": (acc["closest"] + acc["shadow"]) / steps,
           "bvh_bytes": st["bvh_bytes"], "bvh_build_ms": st["bvh_build_ms"], "preprocess_s": pre, "paths_in_flight": st["paths_in_flight"], "hbm_bytes": st["device_bytes"],
           "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]},
           "kernel_ms_per_step": kernel_ms(acc, steps), "film_finite": bool(np.isfinite(film).all()),
           # (general-closure scenes at thousands of samples per pixel: a few pixels of 8 M collect ONE NaN sample, listed here.  Its origin
           # (scripts/nonfinite_probe.py, profiles/r06_nonfinite_probe_*.json: the oracle has the SAME pixels, every finite pixel of their tiles
           # is bit-equal, and the one sample of each is found by bisection): the sheen lobe.  A direction within rounding of the shading normal
           # has cos(theta) = 1 + 1 ulp, and Lambda = exp(2 L(0.5) - L(1 - cos theta)) (src/bsdf/sheen.hpp:51-64) raises the negative
           # 1 - cos(theta) to a fractional power.  The reference has no guard; the restatement and the device reproduce it
           # (tests/test_gpu_parity.py::test_sheen_is_nan_for_a_direction_on_the_normal_...).  NOT "a light seen edge-on", as round 5 wrote.)
           "film_finite_fraction": float(np.isfinite(film).all(-1).mean()),
           "nonfinite_pixels_xy": [[int(x), int(y)] for y, x in np.argwhere(~np.isfinite(film).all(-1))[:32]]}
    tag = workload_tag(kind, triangles, width, height, args.depth)
    if kind in ("zoo", "room"):
        rec["rays_per_camera_sample"] = (acc["closest"] + acc["shadow"]) / max(1, acc["camera"])
        rec["roofline"] = shade_roofline(acc, steps, st, tag)
        rec["roofline_k_trace"] = roofline(acc, steps, None, tag, None, capture_is_this_frame=False)
        rec["primary"] = primary_record(acc, steps, None)
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


# ---- --gpus N: who runs what ---------------------------------------------------------------------------------
def launch_decision(gpus, env, visible_devices, rehearsal=False):
    """What `bench.py --gpus N` does, decided before anything touches the GPU (src/core.cpp:103-115 makes one device per GPU; here it is
    one PROCESS per GPU).  -> ("run", world) render in this process as one rank of `world`;
                              ("spawn", N)  start torch.distributed.run with N ranks as a child process and relay its line;
                              ("error", message)."""
    if gpus < 1:
        return ("error", f"--gpus {gpus}: need at least one GPU")
    ws = env.get("WORLD_SIZE")
    if ws is not None:  # launched by torch.distributed.run (the driver's N > 1 launch, or our own child)
        if int(ws) != gpus:
            return ("error", f"--gpus {gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {gpus}")
        if not rehearsal and visible_devices is not None and visible_devices < gpus:
            return ("error", f"--gpus {gpus} but only {visible_devices} GPU(s) visible: ranks would share a device")
        return ("run", gpus)
    if gpus == 1:
        return ("run", 1)
    if not rehearsal and visible_devices is not None and visible_devices < gpus:
        return ("error", f"--gpus {gpus} but only {visible_devices} GPU(s) visible: ranks would share a device")
    return ("spawn", gpus)


def visible_gpus():
    """number of HIP devices, without initialising the GPU in this process (torch.cuda.device_count() does not, on this image)"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return None


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """python -m torch.distributed.run --nproc-per-node n bench.py <argv> as a CHILD process (never exec: this process may not replace
    itself once a GPU runtime is loaded, and it has not touched the GPU).  Everything the ranks print is relayed; rank 0's JSON line
    is held back and printed LAST.  Returns the child's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    result = None
    for line in child.stdout:
        if line.startswith('{"metric"'):
            result = line.rstrip("\n")
        else:
            sys.stdout.write(line); sys.stdout.flush()
    rc = child.wait()
    if result is not None:
        print(result, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


# ---- the compact line -----------------------------------------------------------------------------------------
LINE_BUDGET = 4096  # bytes of the final stdout line (round 3's 21.9 KB line could not be parsed by the driver)


def _r(x, n=6):
    """numbers to n significant digits: the line is for a reader, the full record keeps every digit"""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    return x


def compact_roofline(rf):
    if not rf:
        return None
    out = {k: _r(rf.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step")}
    hbm = (rf.get("diagnostics") or {}).get("hbm")
    out["hbm_GBps"] = _r(hbm["achieved"]) if hbm else None     # counter bytes of the committed capture / the kernel's time IN that capture
    out["hbm_frac"] = _r(hbm["frac"]) if hbm else None
    if "stream_GBps" in rf:
        out["stream_GBps"] = _r(rf["stream_GBps"])             # compulsory queue bytes / live HIP-event time of this run
    if rf.get("diagnostics"):
        out["counters_from"] = rf["diagnostics"]["source"]
    if rf.get("frac_unavailable"):
        out["frac_unavailable"] = rf["frac_unavailable"][:200]
    if (rf.get("work") or {}).get("peak_source"):
        out["peak_from"] = rf["work"]["peak_source"]
    return out


def compact_line(full, full_path):
    """the <= 4 KB line the driver parses, made from the full record"""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "value_definition", "value_host_film", "value_two_frames_in_flight", "rehearsal")
    out = {k: _r(full[k]) for k in keep if k in full}
    c = full["config"]
    out["config"] = {k: _r(c[k]) for k in ("workload", "tiles", "film_collective", "frames_in_flight", "rays_per_step", "hbm_bytes_per_rank",
                                           "hbm_bytes_per_rank_two_frames_in_flight", "film_mean", "film_finite", "gpu_over_cpu") if k in c}
    if "kernel_ms_per_step" in c:
        out["config"]["kernel_ms_per_step"] = {k: _r(v, 4) for k, v in c["kernel_ms_per_step"].items()}
    out["roofline"] = compact_roofline(full.get("roofline"))
    def by_counters(rf):  # HBM fraction from the committed capture's counters (FETCH x 2 + WRITE over the kernel's time in that capture)
        hbm = ((rf or {}).get("diagnostics") or {}).get("hbm")
        return _r(hbm["frac"], 4) if hbm else None
    rs = full.get("roofline_shade")
    if rs:  # the frame's second kernel: by algorithmic bytes (counts a 64-B triangle record per entry that comes from L2) AND by counters
        out["roofline_shade"] = {"kernel": rs["kernel"], "bound": rs["bound"], "frac": _r(rs["frac"], 4), "frac_by_counters": by_counters(rs)}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "sample")}
        out["cpu_baseline"]["cores_of"] = f"{cb['cores']} threads on a {cb['host']['cpu_share']:g}-CPU share of {cb['host']['hardware_threads']} hardware threads"
    else:
        out["cpu_baseline"] = None
    if full.get("secondary"):
        out["secondary"] = [{"workload": s["workload"][:56], "value": _r(s["value"]), "ms_per_step": _r(s["ms_per_step"], 5),
                             "roofline": {"kernel": s["roofline"]["kernel"], "bound": s["roofline"]["bound"], "frac": _r(s["roofline"]["frac"], 4),
                                          **({"frac_by_counters": by_counters(s["roofline"])} if s["roofline"]["bound"] == "hbm" else {})}}
                            for s in full["secondary"]]
    out["full_record"] = os.path.relpath(full_path, ROOT) if full_path else None
    line = json.dumps(out)
    if len(line) >= LINE_BUDGET:  # never again an unparseable line: drop the optional parts first
        for k in ("secondary", "full_record"):
            out.pop(k, None)
            line = json.dumps(out)
            if len(line) < LINE_BUDGET:
                break
    assert len(line) < LINE_BUDGET, len(line)
    return line


def write_full(full, path):
    if not path:
        return None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            f.write(json.dumps(full) + "\n")
        return path
    except OSError as e:
        print(f"bench.py: full record not written ({e})", file=sys.stderr)
        return None


VALUE_DEFINITION = ("rays of all ranks / max-over-ranks wall time of K frames, ONE frame in flight; inputs (scene, BVH) and the film stay in HBM, "
                    "N > 1: one reduce(sum) of the film to rank 0 inside the bracket; value_host_film = the same frames through the host "
                    "film sink (PCIe, the reference's own start...join bracket); value_two_frames_in_flight = pipelined throughput")


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    # Rehearsal of the N > 1 path on a ONE-GPU box (PHX_BENCH_REHEARSAL=1, never set by the driver): every rank uses GPU 0 and the
    # film reduce runs on gloo, because two RCCL ranks cannot share a device.  Everything else — the launch, tile shard, device films,
    # the reduce, max-over-ranks timing, the rays summed over ranks — is the code the real N-GPU run takes.  Its number is not a result.
    rehearsal = os.environ.get("PHX_BENCH_REHEARSAL") == "1"
    what, arg = launch_decision(args.gpus, os.environ, visible_gpus(), rehearsal)
    if what == "error":
        print("bench.py: " + arg, file=sys.stderr)
        return 2
    if what == "spawn":
        return spawn_ranks(arg, argv)
    world = arg
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    dist = pdist = None
    # torch first: libphx_hip.so then binds to the HIP runtime torch already loaded (one runtime per process).  torch is plumbing
    # here: the film's device memory and, at N > 1, the process group.
    import torch
    torch.cuda.set_device(local_rank)
    cuda = torch.device("cuda", local_rank)
    if use_dist:
        from phosphorus_mk2_amd import dist as pdist
        dist = pdist.init_process_group("gloo" if rehearsal else "nccl", rank, world, None if rehearsal else cuda)
    from phosphorus_mk2_amd import scenes, xpu
    xpu.load_library()  # raises if the HIP extension is missing: no fallback

    scene = scenes.soup(args.triangles, seed=1234, width=args.width, height=args.height)
    opts = xpu.Options(samples_per_pixel=args.spp, paths_per_sample=1, path_depth=args.depth,
                       device_ordinal=local_rank, samples_in_flight=args.samples_in_flight,
                       bvh_builder=args.bvh_builder)
    W, H = args.width, args.height
    t0 = time.time()
    dev = xpu.HipDevice.make(opts)
    dev.preprocess(scene)  # flatten + BVH build + upload: outside the timed region
    preprocess_s = time.time() - t0
    tiles = xpu.Tiles.make(W, H, 32, rank, world)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=cuda)  # the film stays in HBM inside the timed region
    red_dev = "cpu" if rehearsal else "cuda"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- `value`: ONE frame in flight, the reduce inside the bracket — the same protocol at every N -------------------------------
    def one_frame():
        if use_dist:
            # the reduce leaves the SUM in rank 0's film: every rank starts a frame from zeros.  zero_() runs on torch's stream behind the
            # previous frame's reduce; the device renders on its own stream, so wait for the zeros on the host.
            film.zero_()
            ev = torch.cuda.Event(); ev.record(); ev.synchronize()
        tiles.reset()
        dev.start(scene, xpu.FrameState(args.seed, tiles, None, device_film_ptr=film.data_ptr()))
        dev.join()  # join() synchronises the device's stream: this rank's tiles are in its film
        if use_dist:
            pdist.reduce_film(film, dst=0)  # the frame's single collective (RCCL over xGMI); complete before the next zero_() / the barrier
            if rehearsal:
                pass  # gloo reduces synchronously
        return dev.stats()

    acc = new_acc()
    st = None
    for _ in range(args.warmup):
        one_frame()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = one_frame()
        add_stats(acc, st)
    barrier()
    elapsed = time.perf_counter() - t0
    hbm_one = st["device_bytes"] + film.numel() * 4
    film_np = None
    if rank == 0:
        film_np = film.cpu().numpy()  # the last frame of the timed region, reduced onto this rank

    # ---- `value_two_frames_in_flight`: a second device object renders alternate frames, the reduce is asynchronous ----------------
    value_two = None
    hbm_two = None
    if not args.one_sink:
        dev2 = xpu.HipDevice.make(opts); dev2.preprocess(scene)
        devs, tile_sets = [dev, dev2], [tiles, xpu.Tiles.make(W, H, 32, rank, world)]
        films = [film] + [torch.zeros_like(film) for _ in range(3 if use_dist else 1)]  # N > 1: four in turn, a reduce may still read one
        pending = [None] * len(films)

        def go(i):
            b = i % len(films)
            if use_dist:
                if pending[b] is not None:
                    pending[b].wait(); pending[b] = None  # the reduce of four frames ago has read this film
                films[b].zero_()
                ev = torch.cuda.Event(); ev.record(); ev.synchronize()
            tile_sets[i & 1].reset()
            devs[i & 1].start(scene, xpu.FrameState(args.seed, tile_sets[i & 1], None, device_film_ptr=films[b].data_ptr()))

        def done(i):
            d = devs[i & 1]
            d.join()
            if use_dist:
                pending[i % len(films)] = pdist.reduce_film(films[i % len(films)], dst=0, async_op=True)
            sp = d.stats()
            return sp["rays_closest"] + sp["rays_shadow"]

        def drain():
            for b in range(len(films)):
                if pending[b] is not None:
                    pending[b].wait(); pending[b] = None

        def frames(first, n):
            rays = 0
            for i in range(first, first + n):
                if i - first >= 2:
                    rays += done(i - 2)
                go(i)
            for i in range(max(first, first + n - 2), first + n):
                rays += done(i)
            return rays

        frames(0, max(2, args.warmup))  # untimed: both devices have rendered a frame
        drain(); barrier()
        t2 = time.perf_counter()
        rays_p = frames(max(2, args.warmup), args.steps)
        drain(); barrier()
        el2 = time.perf_counter() - t2
        hbm_two = dev.stats()["device_bytes"] + dev2.stats()["device_bytes"] + sum(f.numel() * 4 for f in films)
        if use_dist:
            el2 = pdist.max_over_ranks(el2, red_dev); rays_p = pdist.sum_over_ranks(rays_p, red_dev)
        value_two = rays_p / el2 / 1e6
        dev2.close()
        del films

    # ---- `value_host_film` (N = 1): the same frames through the host film sink ----------------------------------------------------
    value_host = None
    if not use_dist and not args.one_sink:
        host = xpu.Film(W, H, 4)
        rays_o = 0
        for i in range(1 + args.steps):  # one untimed frame sizes the pinned staging buffer
            if i == 1:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            tiles.reset()
            dev.start(scene, xpu.FrameState(args.seed, tiles, host, native_sink=True)); dev.join()
            if i >= 1:
                so = dev.stats(); rays_o += so["rays_closest"] + so["rays_shadow"]
        value_host = rays_o / (time.perf_counter() - t1) / 1e6

    rays_local = acc["closest"] + acc["shadow"]
    if use_dist:
        elapsed = pdist.max_over_ranks(elapsed, red_dev)
        rays_total = pdist.sum_over_ranks(rays_local, red_dev)
        hbm_one = pdist.max_over_ranks(hbm_one, red_dev)
        if hbm_two is not None:
            hbm_two = pdist.max_over_ranks(hbm_two, red_dev)
    else:
        rays_total = float(rays_local)
    dev.close()

    rc = 0
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_total / elapsed / 1e6
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", **({"rehearsal": "all ranks on GPU 0, gloo reduce: not a scaling result"} if rehearsal else {}),
            "value_definition": VALUE_DEFINITION, "value_host_film": value_host, "value_two_frames_in_flight": value_two,
            "config": {"workload": f"Soup({args.triangles}, seed 1234) {W}x{H} {args.spp} spp depth {args.depth} pps 1, "
                                   "1 emissive quad, Lambert 0.73 (BASELINE.json configs[1])",
                       "tiles": "32x32, tile (tx, ty) -> rank (tx + 3 ty) % n_gpus",
                       "film_collective": "reduce(sum) to rank 0, inside the bracket" if use_dist else "none",
                       "frames_in_flight": 1,  # of `value`, at every N
                       "hbm_bytes_per_rank": int(hbm_one), "hbm_bytes_per_rank_two_frames_in_flight": int(hbm_two) if hbm_two is not None else None,
                       "rays_per_step": rays_total / args.steps, "camera_samples_per_step": W * H * args.spp,
                       "preprocess_s": preprocess_s, "bvh_builder": args.bvh_builder, "bvh_build_ms": st["bvh_build_ms"],
                       "bvh_bytes": st["bvh_bytes"], "paths_in_flight": st["paths_in_flight"],
                       "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]},
                       "film_mean": float(film_np[..., :3].mean()), "film_finite": bool(np.isfinite(film_np).all())},
        }
        assert out["n_gpus"] == args.gpus
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            work = count_work(args.triangles, W, H, args.spp, args.bvh_builder)
            base, ref_visits = cpu_baseline(scene, args)
            out["cpu_baseline"] = base
            out["config"]["gpu_over_cpu"] = value / base["value"]
            out["roofline"] = roofline(acc, args.steps, work, workload_tag("soup", args.triangles, W, H, args.depth), ref_visits)
            out["primary"] = primary_record(acc, args.steps, work)
            # the frame's second kernel by time, k_shade (Lambert scenes): HBM-stream roofline by algorithmic bytes, counters from the same capture (full record only)
            out["roofline_shade"] = shade_roofline(acc, args.steps, st, workload_tag("soup", args.triangles, W, H, args.depth))
            if not args.no_secondary and (args.triangles, W, H, args.spp) == (100000, 1280, 720, 256):
                sec = []
                sec.append(secondary_record(xpu, scenes, "Soup(1000000, seed 1234) 1280x720 256 spp depth 9 (north star's target scene)",
                                            "soup", 1000000, 1280, 720, 256, args, cpu_seconds=args.cpu_seconds))
                sec.append(secondary_record(xpu, scenes, "Soup(10000000, seed 1234) 3840x2160 256 spp depth 9: the whole BASELINE config-4 frame on ONE GPU",
                                            "soup", 10000000, 3840, 2160, 256, args, cpu_seconds=0))
                sec.append(secondary_record(xpu, scenes, "BASELINE config 3 at its full size on a stand-in scene (no BMW scene ships with the reference): multi_material_soup(500000), "
                                            "16 closure recipes over all 7 lobe models, 1920x1080, 1024 spp, depth 9, whole frame on one GPU",
                                            "zoo", 500000, 1920, 1080, 1024, args, cpu_seconds=0))
                sec.append(secondary_record(xpu, scenes, "BASELINE config 3 at its full size on MESH geometry: bmw_showroom(500000) — closed room, 24 tessellated spheres, triangle sizes over three decades, "
                                            "the 16 recipes + sharp and frosted glass (per-hit closure weights: k_shade_g<PERHIT>) — 1920x1080, 1024 spp, depth 9, whole frame on one GPU",
                                            "room", 500000, 1920, 1080, 1024, args, cpu_seconds=0))
                sec.append(secondary_record(xpu, scenes, "BASELINE config 5 at its full size on the same stand-in scene: 3840x2160, 4096 spp, depth 9, the WHOLE frame on ONE GPU "
                                            "(34 G camera samples; one timed frame after a warm-up on every 32nd tile — two batches, one of them full-size — no host-film pass: the shading-bound regime, k_shade_g)",
                                            "zoo", 500000, 3840, 2160, 4096, args, cpu_seconds=0, steps=1, warmup=1, host_pass=False, warmup_shard=(0, 32)))
                sec.append(secondary_record(xpu, scenes, "BASELINE config 5 at its full size on MESH geometry: bmw_showroom(500000), 3840x2160, 4096 spp, depth 9, the WHOLE frame on ONE GPU "
                                            "(34 G camera samples, ~250 G rays; one timed frame after a warm-up on every 32nd tile, no host-film pass)",
                                            "room", 500000, 3840, 2160, 4096, args, cpu_seconds=0, steps=1, warmup=1, host_pass=False, warmup_shard=(0, 32)))
                out["secondary"] = sec
        else:
            out["cpu_baseline"] = None
            if world == 1:
                out["roofline"] = roofline(acc, args.steps, None, workload_tag("soup", args.triangles, W, H, args.depth), None)
            else:  # rank 0's own k_trace launches (HIP events of this run): rays per launch, time per launch, compulsory stream bytes per second
                r0 = roofline(acc, args.steps, None, None, None)
                out["roofline"] = {**{k: r0[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms",
                                                         "rays_per_launch", "kernel_rays_per_s", "stream_GBps", "stream_frac_of_hbm_peak")},
                                   "note": "rank 0's share of the frame; the work-based fraction and the counter diagnostics are reported at N = 1"}
        out["config"]["kernel_ms_per_step"] = {**kernel_ms(acc, args.steps), "frame": acc["frame_ms"] / args.steps}
        path = write_full(out, args.full_json)
        print(compact_line(out, path), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
