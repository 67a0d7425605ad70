#!/bin/bash
# Run ON the GPU box (gpurun -- 'bash scripts/profile_gpu.sh <tag>'): rocprofv3 kernel-trace stats of one full bench frame + the
# PMC passes (each in its own run, per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with
# trace domains other than --kernel-trace) on one full frame of the same command.
# Outputs land under gpurun_out/prof_<tag>/; scripts/summarize_profile.py turns them into profiles/*.json.
TAG=${1:-x}
BENCH_ARGS=${BENCH_ARGS:-}   # e.g. BENCH_ARGS='--triangles 1000000' for the 1 M-triangle soup
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
LIMIT=${PROFILE_LIMIT:-300}
timeout -k 10 $LIMIT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --one-sink --full-json $OUT/stats_full.json $BENCH_ARGS > $OUT/stats.log 2>&1; echo "stats rc=$?"
run() { name=$1; shift; timeout -k 10 $LIMIT rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --one-sink --full-json $OUT/${name}_full.json $BENCH_ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcc TCC_HIT TCC_MISS
run td TD_TD_BUSY_sum TD_TC_STALL_sum
echo "profile $TAG done"
