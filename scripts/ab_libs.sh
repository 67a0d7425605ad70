#!/bin/bash
# Run ON the GPU box: A/B already-built libraries and environment knobs on the 100 k and 1 M soups (and whatever AB_TRIANGLES names).
#   bash scripts/ab_libs.sh tag "label:libname:ENV1=v1 ENV2=v2" ...      libname "" = libphx_hip.so, "base" = libphx_hip_base.so
# Interleaved (the variant list is run twice); prints Mrays/s and kernel ms per frame per variant and scene, and the film mean
# (identical films have identical means; the parity suite is what proves identity).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_$TAG; mkdir -p $OUT
for rep in 1 2; do
  for v in "$@"; do
    IFS=: read -r label lib envs <<< "$v"
    so=$R/phosphorus_mk2_amd/libphx_hip${lib:+_$lib}.so
    for tri in ${AB_TRIANGLES:-100000 1000000}; do
      env PHX_LIB=$so $envs python3 $R/bench.py --steps ${AB_STEPS:-6} --warmup 1 --no-cpu-baseline --one-sink --triangles $tri --full-json $OUT/${label}_${tri}_$rep.full.json $BENCH_ARGS > $OUT/${label}_${tri}_$rep.json 2> $OUT/${label}_${tri}_$rep.err || { echo "$label $tri failed"; tail -3 $OUT/${label}_${tri}_$rep.err; }
    done
  done
done
python3 - <<PY
import json,glob,os,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*_*_[12].json")):
    b=os.path.basename(f)[:-5]; name,tri,rep=b.rsplit("_",2)
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: continue
    k=d["config"]["kernel_ms_per_step"]; res[(name,tri)].append((d["value"], k["trace"], k["shade"], k.get("primary", 0.0), d["config"]["film_mean"]))
for (name,tri),v in sorted(res.items(), key=lambda x:(int(x[0][1]),x[0][0])):
    print(f"{tri:>8} {name:<14} " + "  ".join(f"{a:6.0f} Mrays/s trace {b:6.2f} shade {c:5.2f} prim {p:4.2f}" for a,b,c,p,m in v) + f"  film {v[0][4]:.9g}")
PY
