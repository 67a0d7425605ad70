#!/bin/bash
# Run ON the GPU box: the node-cost constant of the device builder's optimal collapse (PHX_CNODE; triangle cost = 1)
# (the knob exists in a study build only: make -C phosphorus_mk2_amd/csrc variant NAME=study EXTRA=-DPHX_STUDY_KNOBS=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
make -C $R/phosphorus_mk2_amd/csrc variant NAME=study EXTRA=-DPHX_STUDY_KNOBS=1 > /dev/null || exit 1
export PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_study.so
for cfg in "soup 100000" "soup 1000000" "showroom 1000000"; do
  set -- $cfg
  for cn in 1.0 1.3 1.6 2.0 2.5 3.2; do
    PHX_CNODE=$cn python3 $R/scripts/run_config.py --scene $1 --triangles $2 --spp 64 --frames 3 --builder device | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 $2 CN=$cn', 'depth', d['bvh_depth'], 'nodes MB %.1f' % d['bvh_MB'], 'trace %.2f' % d['trace_ms'], round(d['Mrays_per_s']))"
  done
done
