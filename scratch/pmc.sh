#!/bin/bash
# usage: pmc.sh <tag> ; runs several PMC passes on a 1-pass frame (spp 18)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 18 > $R/gpurun_out/pmc_$TAG/$name.log 2>&1; }
mkdir -p $R/gpurun_out/pmc_$TAG
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT
run tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
run tcc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $R/gpurun_out/pmc_$TAG/*
