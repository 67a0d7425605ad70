#!/bin/bash
# Run ON the GPU box: one SQ counter pass of a bench frame per library variant.  bash scripts/pmc_ab.sh "name:flags" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=$name EXTRA="$flags" || exit 1
  export PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_$name.so
  OUT=$R/gpurun_out/pmcab_$name; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq1.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq2.log 2>&1
  python3 - $OUT $name <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[2], {k:"%.4g"%v for k,v in sorted(agg.items())})
PY
done
