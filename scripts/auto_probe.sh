#!/bin/bash
# Run ON the GPU box: modelled cost vs measured trace time of the two builders, and what PHX_BVH_AUTO picks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "soup 100000" "soup 1000000" "zoo 500000" "showroom 100000" "showroom 1000000" "cornell 32"; do
  set -- $cfg
  for b in host device auto; do
    python3 $R/scripts/run_config.py --scene $1 --triangles $2 --spp 64 --frames 3 --builder $b | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 $2 $b', 'cost %.4g' % d['bvh_cost_model'], 'dev' if d['bvh_built_on_device'] else 'host', 'depth', d['bvh_depth'], 'build %.0f ms' % d['bvh_build_ms'], 'trace %.1f shade %.1f' % (d['trace_ms'], d['shade_ms']), round(d['Mrays_per_s']))"
  done
done
