#!/bin/bash
# Run ON the GPU box: A/B library variants on the BMW stand-in frame (general closures: k_shade_g).  bash scripts/ab_zoo.sh "name:flags" ...
# The first variant is checked against the default library's film (bit-identical or the run stops).
R=${GRAFT_REPO_ROOT:-$(pwd)}
SPP=${ZOO_SPP:-256}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=$name EXTRA="$flags" > /tmp/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 /tmp/build_$name.log; exit 1; }
done
for rep in 1 2; do
  for v in "$@"; do
    name=${v%%:*}
    PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_$name.so python3 $R/scripts/run_config.py --scene zoo --triangles 500000 --width 1920 --height 1080 --spp $SPP --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('$name', round(d['Mrays_per_s']), 'frame %.1f ms trace %.1f shade %.1f' % (d['frame_s']*1e3, d['trace_ms'], d['shade_ms']), 'film_mean %.9g' % d['film_mean'])"
  done
done
