R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "" "_base"; do
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"; do
    rm -rf /tmp/pm; PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pm -- python3 $R/scripts/run_config.py --scene soup --triangles 100000 --spp 256 --frames 1 > /tmp/pm.log 2>&1
    python3 - <<PY
import csv,glob,collections
fs=glob.glob('/tmp/pm/*/*_counter_collection.csv')
if not fs: print("lib$lib $pass: no output")
else:
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if 'k_shade' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print("lib$lib", {k:round(v) for k,v in agg.items()})
PY
  done
done
