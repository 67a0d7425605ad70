#!/bin/bash
# Run ON the GPU box: kernel-trace stats + a few PMC passes of an arbitrary python command (default: the BMW stand-in frame).
#   bash scripts/profile_cmd.sh <tag> python3 scripts/run_config.py --scene zoo --triangles 500000 --width 1920 --height 1080 --spp 256 --frames 1
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD=("$@"); CMD[1]=$R/${CMD[1]}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- "${CMD[@]}" > $OUT/stats.log 2>&1; echo "stats rc=$?"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- "${CMD[@]}" > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
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
