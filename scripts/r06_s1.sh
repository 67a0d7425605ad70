#!/bin/bash
# round 6, GPU session 1: gate the ring-append k_shade_g, re-measure the VALU peak, A/B ring vs block_append2, phase probes, full suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== gate"; timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closures or glass or random_scenes or destroying" > gpurun_out/s1_gate.log 2>&1 || { tail -40 gpurun_out/s1_gate.log; exit 1; }
tail -2 gpurun_out/s1_gate.log
echo "== valu mix"; bash scripts/capture_valu_mix.sh > gpurun_out/s1_valu_mix.log 2>&1; tail -2 gpurun_out/s1_valu_mix.log | cut -c1-600
echo "== A/B"; bash scripts/ab_scene_libs.sh "ring:" "old:oldappend" > gpurun_out/s1_ab_ring.log 2>&1 || { tail -5 gpurun_out/s1_ab_ring.log; exit 1; }
cat gpurun_out/s1_ab_ring.log
echo "== glds gather"; timeout -k 10 300 scripts/micro/glds_gather > gpurun_out/s1_glds_gather.log 2>&1; echo "glds rc=$?"; cat gpurun_out/s1_glds_gather.log
echo "== phases"; timeout -k 10 200 python3 scripts/shade_phase_probe.py > gpurun_out/s1_phases_ring.log 2>&1 && PHX_PROBE_LIB=$R/phosphorus_mk2_amd/libphx_hip_shtime_old.so timeout -k 10 200 python3 scripts/shade_phase_probe.py > gpurun_out/s1_phases_old.log 2>&1
tail -12 gpurun_out/s1_phases_ring.log
echo "== suite"; timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/s1_tests.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/s1_tests.log
