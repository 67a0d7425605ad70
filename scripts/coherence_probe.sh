#!/bin/bash
# Run ON the GPU box: how much does k_trace gain from coherent rays?  The camera-ray launch traces the same set of rays twice: in
# pixel-major order (a wave = 64 samples of one pixel) and scrambled (-DPHX_PROBE_SCRAMBLE=1: no two neighbouring lanes share a
# pixel).  depth 1 = the camera-ray launch alone (+ one shade and the shadow-ray launch, whose rays inherit the order).
R=${GRAFT_REPO_ROOT:-$(pwd)}
make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=scramble EXTRA="-DPHX_PROBE_SCRAMBLE=1" > /tmp/build_scramble.log 2>&1 || { tail -5 /tmp/build_scramble.log; exit 1; }
make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=plain EXTRA="" > /tmp/build_plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "100000 1280 720 256" "1000000 1280 720 256" "10000000 3840 2160 32"; do
  set -- $cfg
  for lib in plain scramble; do
    OUT=$R/gpurun_out/prof_coh_${1}_$lib; rm -rf $OUT
    PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_$lib.so timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/run_config.py --scene soup --triangles $1 --width $2 --height $3 --spp $4 --frames 1 > /tmp/coh.log 2>&1
    python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$OUT/*/*_kernel_trace.csv")[0])))
rows=[r for r in rows if "k_trace" in r["Kernel_Name"]]; rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ms=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows]
print("$1 triangles $lib: camera-ray launch %.2f ms, bounce-1 launch %.2f ms, bounce-2 %.2f ms, all k_trace %.1f ms" % (ms[0], ms[1], ms[2], sum(ms)))
PY
  done
done
