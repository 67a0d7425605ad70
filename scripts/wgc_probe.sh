#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 900 python3 -m pytest $R/tests -x -q -m gpu 2>&1 | tail -2
BENCH_ARGS=--no-secondary timeout -k 10 600 bash $R/scripts/ab.sh wgc2 "percwave:-DPHX_WG_CURSOR=0" "wgcursor:"
bash $R/scripts/scale_probe.sh 2>&1 | tail -5
bash $R/scripts/ab_zoo.sh "base:" 2>&1 | tail -1
python3 $R/scripts/run_config.py --scene soup --triangles 10000000 --width 3840 --height 2160 --spp 64 --frames 2 | python3 -c "import json,sys; d=json.load(sys.stdin); print('c4 64spp', round(d['Mrays_per_s']), 'trace %.1f shade %.1f' % (d['trace_ms'], d['shade_ms']))"
python3 $R/scripts/run_config.py --scene showroom --triangles 1000000 --spp 64 --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('showroom 1M 64spp', round(d['Mrays_per_s']), 'trace %.1f shade %.1f' % (d['trace_ms'], d['shade_ms']))"
