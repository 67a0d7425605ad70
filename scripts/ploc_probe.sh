#!/bin/bash
# Run ON the GPU box: the device builder's binary tree by Karras radix tree (PHX_PLOC=0) against PLOC (PHX_PLOC=1), same library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "cornell 32" "soup 100000" "soup 1000000" "zoo 500000" "showroom 100000" "showroom 1000000" "showroom 3000000" "soup 10000000"; do
  set -- $cfg
  for pl in 0 1; do
    PHX_PLOC=$pl PLOC_R=${PLOC_R:-} python3 $R/scripts/run_config.py --scene $1 --triangles $2 --spp ${PROBE_SPP:-64} --frames 3 --builder device | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 $2 ploc=$pl', 'depth', d['bvh_depth'], 'build %.1f ms' % d['bvh_build_ms'], 'cost %.5g' % d['bvh_cost_model'], 'trace %.2f' % d['trace_ms'], round(d['Mrays_per_s']), 'film %.9g' % d['film_mean'])"
  done
done
