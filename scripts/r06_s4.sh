#!/bin/bash
# round 6, GPU session 4: packed stack for deep trees (template); window of 8192; prefetch / scalar bsdf_f in the per-hit kernels; constant-glass room; full suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== gate"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closures or glass or random_scenes or showroom or sheen" > gpurun_out/s4_gate.log 2>&1 || { tail -40 gpurun_out/s4_gate.log; exit 1; }
tail -1 gpurun_out/s4_gate.log
echo "== A/B"; bash scripts/ab_scene_libs.sh "product:" "items16:items16" "old:oldappend" > gpurun_out/s4_ab.log 2>&1 || { tail -5 gpurun_out/s4_ab.log; exit 1; }
cat gpurun_out/s4_ab.log
echo "== A/B per-hit kernels"; AB_CFGS="bmwroom:1920:1080:256 glassroom:1280:720:256 bmwroom_cg:1920:1080:256" bash scripts/ab_scene_libs.sh "product:" "phpf2:phpf2" "phsf:phsf" > gpurun_out/s4_ab_perhit.log 2>&1 || { tail -5 gpurun_out/s4_ab_perhit.log; exit 1; }
cat gpurun_out/s4_ab_perhit.log
echo "== phases"; timeout -k 10 200 python3 scripts/shade_phase_probe.py > gpurun_out/s4_phases.log 2>&1; tail -10 gpurun_out/s4_phases.log
echo "== suite"; timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/s4_tests.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/s4_tests.log
