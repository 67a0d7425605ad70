#!/bin/bash
# Run ON the GPU box: A/B a list of library variants on the 100 k and 1 M soups.
#   bash scripts/ab.sh tag "name1:-DFLAG1=1 -DX=2" "name2:" ...     ("name:" = the default build under another name)
# Prints per variant: Mrays/s and k_trace ms per frame of both soups (interleaved runs: variant order is repeated twice).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_$TAG; mkdir -p $OUT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=$name EXTRA="$flags" > $OUT/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $OUT/build_$name.log; exit 1; }
done
for rep in 1 2; do
  for v in "$@"; do
    name=${v%%:*}
    for tri in ${AB_TRIANGLES:-100000 1000000}; do
      PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_$name.so python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --one-sink --triangles $tri $BENCH_ARGS > $OUT/${name}_${tri}_$rep.json 2> $OUT/${name}_${tri}_$rep.err || { echo "$name $tri failed"; tail -3 $OUT/${name}_${tri}_$rep.err; }
    done
  done
done
python3 - <<PY
import json,glob,os,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*_*_*.json")):
    b=os.path.basename(f)[:-5]; name,tri,rep=b.rsplit("_",2)
    try: d=json.load(open(f))
    except Exception: continue
    k=d["config"]["kernel_ms_per_step"]; res[(name,tri)].append((d["value"], k["trace"], k["shade"], k.get("primary", 0.0)))
for (name,tri),v in sorted(res.items(), key=lambda x:(x[0][1],x[0][0])):
    print(f"{tri:>8} {name:<16} " + "  ".join(f"{a:7.0f} Mrays/s primary {p:5.2f} trace {b:6.2f} shade {c:5.2f} ms" for a,b,c,p in v))
PY
