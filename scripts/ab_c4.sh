#!/bin/bash
# Run ON the GPU box: A/B library variants on the BASELINE config-4 frame (Soup(10 M), 3840x2160) at 64 spp, one GPU.
#   bash scripts/ab_c4.sh "name:flags" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=$name EXTRA="$flags" > /tmp/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 /tmp/build_$name.log; exit 1; }
done
for v in "$@"; do
  name=${v%%:*}
  PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_$name.so python3 $R/scripts/run_config.py --scene soup --triangles 10000000 --width 3840 --height 2160 --spp ${C4_SPP:-64} --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('$name', round(d['Mrays_per_s']), 'frame %.1f ms trace %.1f shade %.1f' % (d['frame_s']*1e3, d['trace_ms'], d['shade_ms']), 'film_mean %.9g' % d['film_mean'])"
done
