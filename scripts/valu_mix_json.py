#!/usr/bin/env python3
"""gpurun_out/valu_mix/ (scripts/capture_valu_mix.sh) -> one JSON object: the line scripts/micro/valu_mix printed plus the shader clock
during its node-test kernel, GRBM_GUI_ACTIVE / 8 / kernel time of the SAME dispatches (rocprofv3 --kernel-trace --pmc)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
rec = json.loads(open(os.path.join(out, "valu_mix.out")).read().strip().splitlines()[-1])
cc = glob.glob(os.path.join(out, "grbm", "*", "*_counter_collection.csv"))
kt = glob.glob(os.path.join(out, "grbm", "*", "*_kernel_trace.csv"))
clock = None
if cc:
    rows = [r for r in csv.DictReader(open(cc[0])) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    times = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            times[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    per_kernel = {}
    for r in rows:
        dur = times.get(r["Dispatch_Id"])
        if dur is None and "Start_Timestamp" in r:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        if dur:
            per_kernel.setdefault(r["Kernel_Name"].split("(")[0], []).append((float(r["Counter_Value"]) / 8.0 / dur, dur))
    detail = {}
    for k, v in per_kernel.items():
        long = [c for c, d in v if d >= 0.5 * max(d for _, d in v)]  # the timed launches, not the short warm-up
        detail[k] = {"clock_hz": sum(long) / len(long), "launches": len(long)}
    rec["clock_by_kernel"] = detail
    node = [d["clock_hz"] for k, d in detail.items() if "<0>" in k or "Li0" in k]
    clock = node[0] if node else (sum(d["clock_hz"] for d in detail.values()) / len(detail) if detail else None)
# what the micro-benchmark's loops compile to, against the node / triangle test inside k_trace (scripts/valu_mix_asm.py, asserted there)
ac = os.path.join(out, "asm_check.json")
if os.path.exists(ac) and open(ac).read().strip():
    a = json.loads(open(ac).read().strip().splitlines()[-1])
    rec["asm_check"] = {"node_test_valu_micro": a["priced_node_test_valu"], "node_test_valu_k_trace": a["ktrace_node_test"]["valu"],
                        "node_test_cvt_ubyte_micro": a["priced_node_test_cvt_ubyte"], "node_test_cvt_ubyte_k_trace": a["ktrace_node_test"]["cvt_ubyte"],
                        "tri_test_valu_micro": a["priced_tri_test_valu"], "tri_test_valu_k_trace": a["ktrace_tri_test"]["valu"],
                        "skeleton_valu": a["micro_loop_skeleton"]["valu"], "skeleton_tri_valu": a["micro_loop_skeleton_tri"]["valu"]}
rec["clock_hz"] = clock
rec["clock_source"] = "GRBM_GUI_ACTIVE / 8 / kernel time of the node-test launches (rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -- scripts/micro/valu_mix)" if clock else None
print(json.dumps(rec))
