# Run ON the GPU box: HBM bytes and L2 behaviour of the shade kernel for two libraries (PMC passes on one frame each).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  for pass in ${PMC_PASSES:-"FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS"}; do
    rm -rf /tmp/pm; PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pm -- python3 $R/scripts/run_config.py --scene soup --triangles 100000 --spp 256 --frames 1 > /tmp/pm.log 2>&1
    python3 - <<PY
import csv,glob,collections
fs=glob.glob('/tmp/pm/*/*_counter_collection.csv')
if not fs: print("lib$lib $pass: no output")
else:
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if 'k_shade' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print("lib$lib", dict(agg))
PY
  done
done
