#!/bin/bash
# Run ON the GPU box: memory-pipeline PMC passes (TA / TCP / TD / SQ wait) of one bench frame.  Each pass is its own
# run with at most two counters of one hardware block (more fail with "exceeds the capabilities of the hardware"),
# under its own timeout (rocprofv3 can hang after such a failure).
TAG=${1:-x}
BENCH_ARGS=${BENCH_ARGS:-}   # e.g. BENCH_ARGS='--triangles 1000000' for the 1 M-triangle soup
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/mem_$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?" | tee -a $OUT/progress.log; }
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum
run sq3 SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
run td TD_TD_BUSY_sum TD_TC_STALL_sum
run grbm GRBM_GUI_ACTIVE
echo "mem profile $TAG done"
