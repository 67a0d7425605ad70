#!/bin/bash
# Run ON the GPU box: A/B already-built libraries on the BMW stand-in frames (general closures: k_shade_g).
#   bash scripts/ab_zoo_libs.sh "label:libname" ...     libname "" = libphx_hip.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in "$@"; do
    IFS=: read -r label lib <<< "$v"
    so=$R/phosphorus_mk2_amd/libphx_hip${lib:+_$lib}.so
    for cfg in "1920 1080 ${ZOO_SPP:-256}" "3840 2160 ${ZOO4K_SPP:-64}"; do
      read -r W H S <<< "$cfg"
      PHX_LIB=$so python3 $R/scripts/run_config.py --scene zoo --triangles 500000 --width $W --height $H --spp $S --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('%-10s %4dx%-4d %5.0f Mrays/s  frame %6.1f ms  k_trace %6.1f  shade %6.2f  film %s' % ('$label', $W, $H, d['Mrays_per_s'], d['frame_s']*1e3, d['k_trace_ms'], d['shade_kernel_ms'], d['film_sha1']))"
    done
  done
done
