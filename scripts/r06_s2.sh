#!/bin/bash
# round 6, GPU session 2: dynamic slices A/B, triangle hand-off micro-benchmark + pending-pair counts, bench with the new peak
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== gate"; timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closures or glass or random_scenes or destroying" > gpurun_out/s2_gate.log 2>&1 || { tail -40 gpurun_out/s2_gate.log; exit 1; }
tail -1 gpurun_out/s2_gate.log
echo "== A/B"; AB_CFGS="zoo:1920:1080:256 zoo:3840:2160:64 bmwroom:1920:1080:256 glassroom:1280:720:256" bash scripts/ab_scene_libs.sh "dyn:" "static:ringstatic" "old:oldappend" > gpurun_out/s2_ab_dyn.log 2>&1 || { tail -5 gpurun_out/s2_ab_dyn.log; exit 1; }
cat gpurun_out/s2_ab_dyn.log
echo "== tri handoff"; timeout -k 10 300 scripts/micro/tri_handoff > gpurun_out/s2_tri_handoff.log 2>&1; echo "rc=$?"; cat gpurun_out/s2_tri_handoff.log
echo "== pending pairs"; for tri in 100000 1000000; do timeout -k 10 300 python3 scripts/count_work.py --triangles $tri > gpurun_out/s2_count_$tri.json 2> gpurun_out/s2_count_$tri.err || { tail -3 gpurun_out/s2_count_$tri.err; exit 1; }; python3 -c "import json; d=json.load(open('gpurun_out/s2_count_$tri.json')); print($tri, json.dumps(d['wave']))"; done
echo "== bench"; timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/s2_bench.log 2> gpurun_out/s2_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/s2_bench.log | cut -c1-3000
