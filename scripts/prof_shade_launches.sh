R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "" "_base"; do
  rm -rf /tmp/pr$lib; PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pr$lib -- python3 $R/scripts/run_config.py --scene soup --triangles 100000 --spp 256 --frames 1 > /tmp/pr$lib.log 2>&1
  echo "== lib$lib"; python3 - <<PY
import csv,glob
f=glob.glob('/tmp/pr$lib/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_shade' in r['Kernel_Name']]
print([round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3) for r in rows])
PY
done
