#!/bin/bash
# Run ON the GPU box: the stats pass + nine PMC passes (scripts/profile_gpu.sh / profile_cmd.sh) for the six workloads bench.py reports.
#   bash scripts/capture_all.sh r03zzh    ->  gpurun_out/prof_<tag>_{100k,1M,c4,zoo,zoo4k,room}; summarise with scripts/summarize_profile.py
T=${1:-cap}
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scripts/profile_gpu.sh ${T}_100k > $R/gpurun_out/cap_${T}_100k.log 2>&1 &&
BENCH_ARGS="--triangles 1000000" bash $R/scripts/profile_gpu.sh ${T}_1M > $R/gpurun_out/cap_${T}_1M.log 2>&1 &&
BENCH_ARGS="--triangles 10000000 --width 3840 --height 2160" PROFILE_LIMIT=400 bash $R/scripts/profile_gpu.sh ${T}_c4 > $R/gpurun_out/cap_${T}_c4.log 2>&1 &&
bash $R/scripts/profile_cmd.sh ${T}_zoo python3 scripts/run_config.py --scene zoo --triangles 500000 --width 1920 --height 1080 --spp 256 --frames 1 > $R/gpurun_out/cap_${T}_zoo.log 2>&1 &&
bash $R/scripts/profile_cmd.sh ${T}_zoo4k python3 scripts/run_config.py --scene zoo --triangles 500000 --width 3840 --height 2160 --spp 4096 --world 64 --rank 0 --frames 1 > $R/gpurun_out/cap_${T}_zoo4k.log 2>&1 &&
bash $R/scripts/profile_cmd.sh ${T}_room python3 scripts/run_config.py --scene bmwroom --triangles 500000 --width 1920 --height 1080 --spp 256 --frames 1 > $R/gpurun_out/cap_${T}_room.log 2>&1
for w in 100k 1M c4 zoo zoo4k room; do echo "$w: $(tr '\n' ' ' < $R/gpurun_out/cap_${T}_$w.log)"; done
