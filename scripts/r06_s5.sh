#!/bin/bash
# round 6, GPU session 5: prefetch stages in the per-hit kernels on top of the scalar bsdf_f; 3 000 fuzzed scenes on the round's kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== A/B per-hit kernels"; AB_CFGS="bmwroom:1920:1080:256 glassroom:1280:720:256" bash scripts/ab_scene_libs.sh "product:" "phboth:phboth" "phpf1:phpf1" > gpurun_out/s5_ab_perhit.log 2>&1 || { tail -5 gpurun_out/s5_ab_perhit.log; exit 1; }
cat gpurun_out/s5_ab_perhit.log
echo "== fuzz"; timeout -k 10 900 python3 scripts/fuzz_parity.py 3000 600000 > gpurun_out/s5_fuzz.json 2> gpurun_out/s5_fuzz.err; echo "fuzz rc=$?"; tail -1 gpurun_out/s5_fuzz.json | cut -c1-600
