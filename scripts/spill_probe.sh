#!/bin/bash
# Run ON the GPU box: deep trees against the number of stack levels kept in LDS (the rest spills to HBM; 1024-thread workgroups)
R=${GRAFT_REPO_ROOT:-$(pwd)}
one() { # scene triangles width height spp levels...
  scene=$1; tri=$2; w=$3; h=$4; spp=$5; shift 5
  for L in "$@"; do
    PHX_LDS_LEVELS=$L python3 $R/scripts/run_config.py --scene $scene --triangles $tri --width $w --height $h --spp $spp --frames 2 --builder device | python3 -c "import json,sys; d=json.load(sys.stdin); print('$scene $tri lds_levels $L plan', d['plan'], 'depth', d['bvh_depth'], round(d['Mrays_per_s']), 'trace %.1f shade %.1f' % (d['trace_ms'], d['shade_ms']), 'mean %.6f' % d['film_mean'])"
  done
}
one showroom 2400000 1280 720 64 99 8 7 6 5
one showroom 400000 1280 720 64 99 8 7 6 5
one soup 10000000 3840 2160 32 99 8 7 6
