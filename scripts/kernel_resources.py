#!/usr/bin/env python3
"""Resource table of every kernel of the device library (VGPRs, SGPRs, scratch, LDS, occupancy, code bytes) from the compiler's
kernel-resource-usage remarks: `python scripts/kernel_resources.py [-DNAME=VALUE ...] > profiles/rNN_kernel_resources.txt`.
Runs `hipcc -S --cuda-device-only` on kernels.hip and bvh_gpu.hip with the flags of csrc/Makefile; needs no GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phosphorus_mk2_amd", "csrc")
FLAGS = "-std=c++17 -O3 -fno-slp-vectorize -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()
rows = []
for src in ("kernels.hip", "bvh_gpu.hip"):
    out = "/tmp/_phx_%s.s" % src.split(".")[0]
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "-S", "--cuda-device-only", "-o", out, src,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
print(f"{'kernel':74s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B':>9s} {'LDS B':>7s} {'waves/SIMD':>10s}")
for r in rows:
    if "rocprim" in r["name"]:
        continue  # the library's radix sort, not ours
    n = re.sub(r"\((?!anonymous).*", "", r["name"]).replace("void ", "").replace("phx::", "").replace("(anonymous namespace)::", "")
    print(f"{n[:74]:74s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('ScratchSize [bytes/lane]', '?'):>9s} "
          f"{r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>10s}")
