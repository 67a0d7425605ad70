#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/profile_gpu.sh) into the committed summaries under profiles/:
  profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of one full bench frame
  profiles/<name>_pmc.json           per-kernel PMC sums + HBM traffic per launch of the dominant kernel
HBM traffic follows MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate passes;
on gfx950 FETCH_SIZE counts half the bytes of wide streaming reads, so the corrected figure doubles it (this
kernel's reads are 16-B-per-lane gathers, an access shape the guide calls uncalibrated: both raw and corrected
values are recorded)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")


def kname(s):
    """kernel name without arguments and without template arguments: the instantiations of k_trace / k_shade are summed"""
    return s.split("(")[0].replace("void phx::", "").replace("phx::", "").split("<")[0]


stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
# per k_trace launch: counters keyed by the launch's position among the k_trace dispatches of its pass (0 = the first bounce: the camera rays are k_trace_primary's)
per_launch = collections.defaultdict(lambda: collections.defaultdict(list))
launches = collections.defaultdict(set)
passes = collections.defaultdict(set)  # a counter collected in several passes is averaged over them
mem = os.path.join(root, "gpurun_out", f"mem_{tag}")  # scripts/profile_mem.sh: TA / TCP / TD passes of the same command
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")) + glob.glob(os.path.join(mem, "*", "*", "*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    trace_ids = sorted({int(r["Dispatch_Id"]) for r in rows if kname(r["Kernel_Name"]) == "k_trace"})
    order = {d: i for i, d in enumerate(trace_ids)}
    for r in rows:
        k = kname(r["Kernel_Name"])
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            passes[(k, r["Counter_Name"])].add(os.path.dirname(f))
            launches[k].add((os.path.dirname(f), r["Dispatch_Id"]))
        if k == "k_trace":
            per_launch[order[int(r["Dispatch_Id"])]][r["Counter_Name"]].append((os.path.dirname(f), float(r["Counter_Value"])))
out = {"command": os.environ.get("PROFILE_COMMAND", "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --one-sink " + os.environ.get("BENCH_ARGS", "") + " (scripts/profile_gpu.sh)"),
       "kernels": {}}
for (k, c), ps in passes.items():
    agg[k][c] /= len(ps)
for k, a in agg.items():
    n = len({d for (_, d) in launches[k]})
    e = {"launches": n, "counters": dict(a)}
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        raw = (a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0
        cor = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_per_launch_raw"] = raw / n
        e["hbm_bytes_per_launch_corrected"] = cor / n
    if "TCC_HIT" in a:
        e["l2_hit_rate"] = a["TCC_HIT"] / (a["TCC_HIT"] + a["TCC_MISS"])
    if "SQ_THREAD_CYCLES_VALU" in a and "SQ_INSTS_VALU" in a:
        e["valu_lane_utilisation"] = a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_INSTS_VALU"] * 64.0)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in a and "GRBM_GUI_ACTIVE" in a:
        # vector-L1 path: lane addresses per clock and CU (ceiling measured by scripts/micro/l1_gather.hip: ~1.7, whatever the
        # width of the load), and the busy fractions of the address (TA) and data-return (TD) units
        cu_cycles = a["GRBM_GUI_ACTIVE"] / 8.0 * 256.0  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["l1_lane_accesses_per_clk_per_cu"] = a["TCP_TOTAL_CACHE_ACCESSES_sum"] / cu_cycles
        e["ta_busy_frac"] = a.get("TA_TA_BUSY_sum", 0.0) / cu_cycles
        e["td_busy_frac"] = a.get("TD_TD_BUSY_sum", 0.0) / cu_cycles
        e["l1_hit_rate"] = 1.0 - a.get("TCP_TCC_READ_REQ_sum", 0.0) / a["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "SQ_WAIT_INST_ANY" in a and "SQ_WAVE_CYCLES" in a:
            e["wave_cycles_waiting_frac"] = a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]
    out["kernels"][k] = e
# k_trace launch by launch (a counter summed over its rows within a pass directory, averaged over the passes that collected it)
pl = []
for i in sorted(per_launch):
    c = {}
    for cn, vals in per_launch[i].items():
        by_pass = collections.defaultdict(float)
        for dpath, v in vals:
            by_pass[dpath] += v
        c[cn] = sum(by_pass.values()) / len(by_pass)
    e = {"launch": i, "what": "camera rays" if i == 0 else f"bounce {i} closest-hit rays + bounce {i - 1} shadow rays", "counters": c}
    if "GRBM_GUI_ACTIVE" in c:
        cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c: e["l1_lane_accesses_per_clk_per_cu"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / cu_cycles
        if "TA_TA_BUSY_sum" in c: e["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / cu_cycles
        if "TD_TD_BUSY_sum" in c: e["td_busy_frac"] = c["TD_TD_BUSY_sum"] / cu_cycles
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: e["hbm_bytes_corrected"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    if "TCC_HIT" in c: e["l2_hit_rate"] = c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"])
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_INSTS_VALU" in c: e["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0)
    pl.append(e)
out["k_trace_per_launch"] = pl
# launch durations from the kernel trace of the --stats pass, in dispatch order
kt = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
if kt:
    allrows = list(csv.DictReader(open(kt[0])))
    rows = [r for r in allrows if kname(r["Kernel_Name"]) == "k_trace"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out["k_trace_launch_ms"] = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    # every kernel's time in the --kernel-trace --stats pass of this capture: the duration the PMC sums above are divided by
    # (bench.py: a counter is never combined with the time of another run)
    km = collections.defaultdict(lambda: {"launches": 0, "total_ms": 0.0, "launch_ms": []})
    for r in sorted(allrows, key=lambda r: int(r["Dispatch_Id"])):
        k = kname(r["Kernel_Name"])
        if k.startswith("k_"):
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            km[k]["launches"] += 1; km[k]["total_ms"] += ms; km[k]["launch_ms"].append(round(ms, 4))
    out["kernel_ms_stats_pass"] = km
json.dump(out, open(os.path.join(dst, f"{name}_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: v for x, v in e.items() if x != "counters"} for k, e in out["kernels"].items()}, indent=1))
