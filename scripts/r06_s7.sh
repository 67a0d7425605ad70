#!/bin/bash
# round 6, GPU session 7: stack levels kept in LDS with the 5-byte entries (deep trees); then the round's bench record
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== LDS levels (packed stack)"
for rep in 1 2; do for lv in 0 6 8 9; do
  PHX_LDS_LEVELS=$lv python3 scripts/run_config.py --scene soup --triangles 10000000 --width 3840 --height 2160 --spp 64 --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('c4   PHX_LDS_LEVELS=$lv', round(d['Mrays_per_s']), 'k_trace %.1f plan %s' % (d['k_trace_ms'], d['plan']), d['film_sha1'])"
  PHX_LDS_LEVELS=$lv python3 scripts/run_config.py --scene bmwroom --triangles 500000 --width 1920 --height 1080 --spp 256 --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('room PHX_LDS_LEVELS=$lv', round(d['Mrays_per_s']), 'k_trace %.1f plan %s' % (d['k_trace_ms'], d['plan']), d['film_sha1'])"
done; done 2>&1 | tee gpurun_out/s7_lds_levels.log
echo "== bench"; timeout -k 10 1000 python3 bench.py --steps 20 --warmup 5 > gpurun_out/s7_bench.log 2> gpurun_out/s7_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/s7_bench.log | cut -c1-4000
