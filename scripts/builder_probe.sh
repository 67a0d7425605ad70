#!/bin/bash
# Run ON the GPU box: trace time of the host SAH tree against the device LBVH tree (library variants given as "name:flags") on soups,
# the 16-recipe stand-in and the showroom.  bash scripts/builder_probe.sh "emc0:-DPHX_EMC=0" "emc1:-DPHX_EMC=1"
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARIANTS=("$@")
for v in "${VARIANTS[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=$name EXTRA="$flags" > /tmp/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 /tmp/build_$name.log; exit 1; }
done
for cfg in "soup 100000" "soup 1000000" "zoo 500000" "showroom 100000" "showroom 1000000" "showroom 3000000"; do
  set -- $cfg
  python3 $R/scripts/run_config.py --scene $1 --triangles $2 --spp 64 --frames 3 --builder host | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 $2 host          ', 'depth', d['bvh_depth'], 'build %.0f ms' % d['bvh_build_ms'], 'trace %.2f' % d['trace_ms'], round(d['Mrays_per_s']))"
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_$name.so python3 $R/scripts/run_config.py --scene $1 --triangles $2 --spp 64 --frames 3 --builder device | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 $2 device $name', 'depth', d['bvh_depth'], 'build %.0f ms' % d['bvh_build_ms'], 'trace %.2f' % d['trace_ms'], round(d['Mrays_per_s']), 'film %.9g' % d['film_mean'])"
  done
done
