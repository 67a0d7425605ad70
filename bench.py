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
  roofline     of the dominant kernel, k_trace (closest-hit rays of a step + shadow rays of the previous step in one persistent
               launch; time = HIP events on the device's own stream around every launch of the timed steps).  Every `frac` is a
               real fraction of a stated ceiling:
                 valu          the bound: the node visits and triangle tests the frame needs (counted by the instrumented
                               build, libphx_hip_count.so, on the same frame) priced at the rate the chip runs k_trace's own
                               arithmetic with nothing else in the way (scripts/micro/valu_mix.hip, profiles/r02_valu_mix.json)
                               = minimum ALU time / k_trace time
                 valu_issue    VALU wave-instructions issued (committed PMC pass) vs 256 CUs x 4 SIMDs x 1 per 2 clocks
                 vector_l1     vector-L1 lane addresses (PMC) vs the measured 1.7 per clock and CU (scripts/micro/l1_gather.hip)
                 l2            L1->L2 read requests x 64 B (PMC) vs 34.5 TB/s
                 hbm           FETCH_SIZE x 2 + WRITE_SIZE (separate PMC passes, MI355X_MICROARCH.md) vs 8 TB/s; `traffic` = those
                               bytes per launch
               `algorithmic_ref_layout` keeps SURVEY §8(d)'s figure (bytes per ray in the REFERENCE's 288-B node / 384-B packet
               layout, V_n and V_l from the CPU restatement's counters): a work-normalised rate, not a bound — the device's tree
               is smaller and lives in L2.  `device_layout` = the bytes this kernel's own layout moves per ray.
  cpu_baseline the CPU restatement (oracle/, kind "port") on this box's host cores: a warm thread pool renders tiles of the same
               frame (counter RNG) for >= 10 s; thread start-up and per-thread stream construction are outside the clock; rates
               at 1, 8, 64, the physical cores and all hardware threads are listed.
  secondary    the same measurement on Soup(1 M) (the north star's target scene) and on the whole BASELINE config-4 frame
               (Soup(10 M), 3840x2160, 256 spp) on this one GPU.
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
def workload_tag(triangles, width, height, depth=9):
    return {(100000, 1280, 720): "100k", (1000000, 1280, 720): "1M", (10000000, 3840, 2160): "c4"}.get((triangles, width, height)) if depth == 9 else None


def committed_pmc(args_like):
    """k_trace's PMC sums of one frame of this workload from the newest committed summary (profiles/r*_<tag>_pmc.json, written by
    scripts/summarize_profile.py from separate rocprofv3 passes).  bench.py cannot collect PMCs itself; other workloads get None."""
    tag = workload_tag(args_like.triangles, args_like.width, args_like.height, args_like.depth)
    if tag is None:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    e = d["kernels"].get("k_trace")
    return (e, os.path.relpath(files[-1], ROOT)) if e else (None, None)


def committed_traffic(args_like):
    """HBM bytes per k_trace launch (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) from the newest committed PMC summary"""
    e, src = committed_pmc(args_like)
    if e and "hbm_bytes_per_launch_corrected" in e:
        return e["hbm_bytes_per_launch_corrected"], src
    return None, None


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


def roofline(acc, steps, work, pmc, pmc_src, ref_visits, like):
    """acc: rays and k_trace ms summed over `steps` timed frames; work: count_work() of one frame; pmc: committed PMC entry"""
    nl = max(1, acc["launches"])
    t_frame = acc["closest_ms"] * 1e-3 / steps          # k_trace seconds per frame (HIP events)
    t_launch = acc["closest_ms"] * 1e-3 / nl
    rays_c, rays_s = acc["closest"] / steps, acc["shadow"] / steps
    roof = {"kernel": "k_trace", "launches_per_step": nl / steps, "avg_launch_ms": t_launch * 1e3,
            "rays_per_launch": (acc["closest"] + acc["shadow"]) / nl, "kernel_rays_per_s": (rays_c + rays_s) / t_frame,
            "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "ceilings": {}}
    C = roof["ceilings"]
    peak, peak_src = committed_valu_peak()
    if work and "error" not in work and peak:
        nv = sum(work[k]["rays"] * (work[k]["node_visits_lds_per_ray"] + work[k]["node_visits_mem_per_ray"]) for k in ("closest", "shadow"))
        tt = sum(work[k]["rays"] * work[k]["tri_tests_per_ray"] for k in ("closest", "shadow"))
        t_min = nv / peak["node_tests_per_s"] + tt / peak["tri_tests_per_s"]
        C["valu"] = {"achieved": nv / t_frame / 1e9, "peak": nv / t_min / 1e9, "unit": "G node-visit equivalents/s", "frac": t_min / t_frame,
                     "node_visits_per_frame": nv, "tri_tests_per_frame": tt, "min_alu_ms_per_frame": t_min * 1e3,
                     "peak_node_tests_per_s": peak["node_tests_per_s"], "peak_tri_tests_per_s": peak["tri_tests_per_s"], "peak_source": peak_src,
                     "lanes_per_node_block": work["wave"]["lanes_per_node_block"], "lanes_per_tri_block": work["wave"]["lanes_per_tri_block"],
                     "what": "node visits + triangle tests of this frame (instrumented build) priced at the chip's rate for k_trace's own "
                             "arithmetic alone (all 64 lanes active, operands in registers) = minimum ALU time / k_trace time"}
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
    if pmc:
        c = pmc["counters"]; n = pmc["launches"]
        scale = nl / steps / n  # PMC sums are over one frame of n launches
        if "SQ_INSTS_VALU" in c:
            a = c["SQ_INSTS_VALU"] * scale / t_frame
            C["valu_issue"] = {"achieved": a / 1e9, "peak": CUS * 4 * CLOCK_HZ / 2 / 1e9, "unit": "G wave-instructions/s", "frac": a / (CUS * 4 * CLOCK_HZ / 2),
                               "lane_utilisation": pmc.get("valu_lane_utilisation")}
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            a = c["TCP_TOTAL_CACHE_ACCESSES_sum"] * scale / t_frame
            C["vector_l1"] = {"achieved": a / 1e9, "peak": L1_ADDR_PER_CLK_CU * CUS * CLOCK_HZ / 1e9, "unit": "G lane addresses/s",
                              "frac": a / (L1_ADDR_PER_CLK_CU * CUS * CLOCK_HZ), "per_ray": c["TCP_TOTAL_CACHE_ACCESSES_sum"] * scale / (rays_c + rays_s)}
        if "TCP_TCC_READ_REQ_sum" in c:
            a = c["TCP_TCC_READ_REQ_sum"] * scale * 64.0 / t_frame
            C["l2"] = {"achieved": a / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": a / 1e9 / L2_PEAK_GBS, "l2_hit_rate": pmc.get("l2_hit_rate")}
        if "hbm_bytes_per_launch_corrected" in pmc:
            roof["traffic"] = pmc["hbm_bytes_per_launch_corrected"]
            a = pmc["hbm_bytes_per_launch_corrected"] / t_launch
            C["hbm"] = {"achieved": a / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / 1e9 / HBM_PEAK_GBS,
                        "bytes_per_ray": pmc["hbm_bytes_per_launch_corrected"] * n / (rays_c + rays_s)}
        roof["pmc_source"] = pmc_src
    if C:
        b = max(C, key=lambda k: C[k]["frac"])
        roof.update({"bound": b, "achieved": C[b]["achieved"], "peak": C[b]["peak"], "unit": C[b]["unit"], "frac": C[b]["frac"]})
    if ref_visits:
        (vn, vl), (vns, vls) = ref_visits["closest"], ref_visits["shadow"]
        b_ray, b_shadow = 56.0 + vn * 288.0 + vl * 384.0, 36.0 + vns * 288.0 + vls * 384.0
        roof["algorithmic_ref_layout"] = {
            "GBps": (rays_c * b_ray + rays_s * b_shadow) / t_frame / 1e9, "bytes_per_ray": {"closest": b_ray, "shadow": b_shadow},
            "visits_per_ray": {"closest": {"nodes": vn, "leaf_packets": vl}, "shadow": {"nodes": vns, "leaf_packets": vls}},
            "note": "SURVEY 8(d): bytes per ray in the REFERENCE's 288-B node / 384-B packet layout (CPU restatement's counters) over k_trace "
                    "time: a work-normalised rate, not a fraction of any ceiling (the device's tree is 10x smaller and L2-resident)"}
    return roof


def run_workload(xpu, scenes, triangles, width, height, spp, depth, seed, builder, steps, warmup, samples_in_flight=0):
    """one device, one scene, `steps` timed frames on one GPU -> (value Mrays/s, ms per step, acc, last stats, preprocess s)"""
    scene = scenes.soup(triangles, seed=1234, width=width, height=height)
    import torch
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=spp, paths_per_sample=1, path_depth=depth, samples_in_flight=samples_in_flight,
                                         bvh_builder=builder, device_ordinal=torch.cuda.current_device()))
    t0 = time.time(); dev.preprocess(scene); pre = time.time() - t0
    tiles = xpu.Tiles.make(width, height, 32)
    import torch  # device memory for the film, which stays in HBM inside the timed region (as in main())
    film_dev = torch.zeros((height, width, 4), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    acc = {"closest": 0, "shadow": 0, "closest_ms": 0.0, "shade_ms": 0.0, "launches": 0, "frame_ms": 0.0}
    st = None
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        tiles.reset()
        dev.start(scene, xpu.FrameState(seed, tiles, None, device_film_ptr=film_dev.data_ptr())); dev.join()
        st = dev.stats()
        if i >= warmup:
            acc["closest"] += st["rays_closest"]; acc["shadow"] += st["rays_shadow"]; acc["closest_ms"] += st["closest_ms"]
            acc["shade_ms"] += st["shade_ms"]; acc["launches"] += st["trace_launches"]; acc["frame_ms"] += st["frame_ms"]
    elapsed = time.perf_counter() - t0
    film = film_dev.cpu().numpy()
    del film_dev
    dev.close()
    return (acc["closest"] + acc["shadow"]) / elapsed / 1e6, elapsed * 1e3 / steps, acc, st, pre, scene, film


def secondary_record(xpu, scenes, name, triangles, width, height, spp, args, cpu_seconds):
    like = argparse.Namespace(triangles=triangles, width=width, height=height, depth=args.depth, spp=spp, seed=args.seed, cpu_spp=args.cpu_spp,
                              cpu_seconds=cpu_seconds)
    value, ms, acc, st, pre, scene, film = run_workload(xpu, scenes, triangles, width, height, spp, args.depth, args.seed, "auto", steps=2, warmup=1)
    rec = {"workload": name, "value": value, "unit": "Mrays/s", "ms_per_step": ms, "steps": 2, "rays_per_step": (acc["closest"] + acc["shadow"]) / 2,
           "bvh_bytes": st["bvh_bytes"], "bvh_build_ms": st["bvh_build_ms"], "preprocess_s": pre, "paths_in_flight": st["paths_in_flight"],
           "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]},
           "kernel_ms_per_step": {"trace": acc["closest_ms"] / 2, "shade_gen_film": acc["shade_ms"] / 2}, "film_finite": bool(np.isfinite(film).all())}
    work = count_work(triangles, width, height, spp, "auto")
    pmc, src = committed_pmc(like)
    ref_visits = None
    if cpu_seconds > 0:
        logical, _, share = host_cpus()
        base, ref_visits = cpu_baseline(scene, like, seconds=cpu_seconds, thread_counts={1: 0.2, max(1, min(logical, int(round(share)))): 1.0, min(64, logical): 0.5})
        rec["cpu_baseline"] = base
        rec["gpu_over_cpu"] = value / base["value"]
    rec["roofline"] = roofline(acc, 2, work, pmc, src, ref_visits, like)
    return rec


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    dist = None
    # torch first: libphx_hip.so then binds to the HIP runtime torch already loaded (one runtime per process).  torch is plumbing
    # here: the film's device memory and, at N > 1, the process group.
    import torch
    torch.cuda.set_device(local_rank)
    if use_dist:
        from phosphorus_mk2_amd import dist as pdist
        dist = pdist.init_process_group("nccl", rank, world, torch.device("cuda", local_rank))
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

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        tiles.reset()
        if use_dist:
            film_dev.zero_()
            torch.cuda.synchronize()
            dev.start(scene, xpu.FrameState(args.seed, tiles, None, device_film_ptr=film_dev.data_ptr()))
            dev.join()  # join() synchronises the device's stream
            pdist.reduce_film(film_dev, dst=0)  # the single film collective (RCCL over xGMI)
        elif film_dev is not None:
            # no clearing: at world 1 the device's tiles cover (and overwrite) every pixel of the film
            dev.start(scene, xpu.FrameState(args.seed, tiles, None, device_film_ptr=film_dev.data_ptr()))
            dev.join()  # join() synchronises the device's stream
        else:
            dev.start(scene, xpu.FrameState(args.seed, tiles, film_host, native_sink=True))
            dev.join()
        return dev.stats()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {"closest": 0, "shadow": 0, "closest_ms": 0.0, "shade_ms": 0.0, "launches": 0, "frame_ms": 0.0}
    for _ in range(args.steps):
        st = step()
        acc["closest"] += st["rays_closest"]; acc["shadow"] += st["rays_shadow"]
        acc["closest_ms"] += st["closest_ms"]; acc["shade_ms"] += st["shade_ms"]
        acc["launches"] += st["trace_launches"]; acc["frame_ms"] += st["frame_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    rays_local = acc["closest"] + acc["shadow"]
    if use_dist:
        elapsed = pdist.max_over_ranks(elapsed, "cuda")
        rays_total = pdist.sum_over_ranks(rays_local, "cuda")
    else:
        rays_total = float(rays_local)

    if rank == 0:
        film = film_dev.cpu().numpy() if film_dev is not None else film_host.data
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_total / elapsed / 1e6
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Soup({args.triangles}, seed 1234) {W}x{H} {args.spp} spp depth {args.depth} pps 1, "
                                   "1 emissive quad, Lambert 0.73 (BASELINE.json configs[1])",
                       "tiles": "32x32, tile (tx, ty) -> rank (tx + 3 ty) % n_gpus", "film_collective": "reduce(sum) to rank 0" if use_dist else "none",
                       "rays_per_step": rays_total / args.steps, "camera_samples_per_step": W * H * args.spp,
                       "preprocess_s": preprocess_s, "bvh_builder": args.bvh_builder, "bvh_build_ms": st["bvh_build_ms"],
                       "bvh_bytes": st["bvh_bytes"], "paths_in_flight": st["paths_in_flight"],
                       "plan": {"block": st["trace_block"], "ntop": st["trace_ntop"], "levels": st["trace_levels"]},
                       "film_mean": float(film[..., :3].mean()), "film_finite": bool(np.isfinite(film).all())},
        }
        dev.close()
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            work = count_work(args.triangles, W, H, args.spp, args.bvh_builder)
            base, ref_visits = cpu_baseline(scene, args)
            out["cpu_baseline"] = base
            out["config"]["gpu_over_cpu"] = value / base["value"]
            pmc, src = committed_pmc(args)
            out["roofline"] = roofline(acc, args.steps, work, pmc, src, ref_visits, args)
            if not args.no_secondary and (args.triangles, W, H, args.spp) == (100000, 1280, 720, 256):
                sec = []
                sec.append(secondary_record(xpu, scenes, "Soup(1000000, seed 1234) 1280x720 256 spp depth 9 (north star's target scene)",
                                            1000000, 1280, 720, 256, args, cpu_seconds=args.cpu_seconds))
                sec.append(secondary_record(xpu, scenes, "Soup(10000000, seed 1234) 3840x2160 256 spp depth 9: the whole BASELINE config-4 frame on ONE GPU",
                                            10000000, 3840, 2160, 256, args, cpu_seconds=0))
                out["secondary"] = sec
        else:
            out["cpu_baseline"] = None
            pmc, src = committed_pmc(args) if world == 1 else (None, None)
            out["roofline"] = roofline(acc, args.steps, None, pmc, src, None, args) if world == 1 else {
                "bound": None, "kernel": "k_trace", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "note": "roofline and cpu_baseline are reported at N = 1"}
        out["config"]["kernel_ms_per_step"] = {"trace": acc["closest_ms"] / args.steps, "shade_gen_film": acc["shade_ms"] / args.steps,
                                               "frame": acc["frame_ms"] / args.steps}
        print(json.dumps(out))
    else:
        dev.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
