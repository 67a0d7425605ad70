#!/bin/bash
# round 6, GPU session 3: packed 5-byte stack entries A/B (+ what more staged nodelets buy in memory visits), non-finite pixel probe, the room parity test
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== gate (packed stack lib on the trace / render parity tests)"
PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_packed.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "trace or soup or full_size or stress or showroom or random_scenes" > gpurun_out/s3_gate_packed.log 2>&1 || { tail -40 gpurun_out/s3_gate_packed.log; exit 1; }
tail -1 gpurun_out/s3_gate_packed.log
echo "== A/B 100k / 1M"; bash scripts/ab_libs.sh r06pk "base::" "packed:packed:" > gpurun_out/s3_ab_packed.log 2>&1; cat gpurun_out/s3_ab_packed.log
echo "== A/B c4 + room"
for rep in 1 2; do for lib in "" "_packed"; do
  PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip$lib.so python3 scripts/run_config.py --scene soup --triangles 10000000 --width 3840 --height 2160 --spp 64 --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('c4 lib$lib', round(d['Mrays_per_s']), 'frame %.1f ms k_trace %.1f plan %s' % (d['frame_s']*1e3, d['k_trace_ms'], d['plan']), d['film_sha1'])"
  PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip$lib.so python3 scripts/run_config.py --scene bmwroom --triangles 500000 --width 1920 --height 1080 --spp 256 --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('room lib$lib', round(d['Mrays_per_s']), 'frame %.1f ms k_trace %.1f plan %s' % (d['frame_s']*1e3, d['k_trace_ms'], d['plan']), d['film_sha1'])"
done; done 2>&1 | tee gpurun_out/s3_ab_packed_c4_room.log
echo "== room parity test"; timeout -k 10 600 python3 -m pytest tests/test_gpu_config4.py -x -q -m gpu -k "showroom" > gpurun_out/s3_room_test.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/s3_room_test.log
echo "== non-finite probe"; timeout -k 10 600 python3 scripts/nonfinite_probe.py --scene zoo > gpurun_out/s3_nonfinite_zoo.json 2> gpurun_out/s3_nonfinite_zoo.err; echo "rc=$?"; cat gpurun_out/s3_nonfinite_zoo.json | cut -c1-3000
timeout -k 10 600 python3 scripts/nonfinite_probe.py --scene bmwroom --width 1920 --height 1080 --spp 1024 > gpurun_out/s3_nonfinite_room.json 2> gpurun_out/s3_nonfinite_room.err; echo "rc=$?"; cat gpurun_out/s3_nonfinite_room.json | cut -c1-2000
