#!/bin/bash
# Run ON the GPU box: A/B already-built libraries on general-closure frames, interleaved.
#   AB_CFGS="zoo:1920:1080:256 bmwroom:1920:1080:256" bash scripts/ab_scene_libs.sh "label:libname" ...     libname "" = libphx_hip.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in "$@"; do
    IFS=: read -r label lib <<< "$v"
    so=$R/phosphorus_mk2_amd/libphx_hip${lib:+_$lib}.so
    for cfg in ${AB_CFGS:-zoo:1920:1080:256 zoo:3840:2160:64 bmwroom:1920:1080:256 glassroom:1280:720:256}; do
      IFS=: read -r SC W H S <<< "$cfg"
      PHX_LIB=$so timeout -k 10 300 python3 $R/scripts/run_config.py --scene $SC --triangles ${AB_TRIANGLES:-500000} --width $W --height $H --spp $S --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('%-10s %-9s %4dx%-4d %4d spp %5.0f Mrays/s  frame %7.1f ms  k_trace %7.1f  shade %7.2f  film %s finite %s' % ('$label', '$SC', $W, $H, $S, d['Mrays_per_s'], d['frame_s']*1e3, d['k_trace_ms'], d['shade_kernel_ms'], d['film_sha1'], d['finite']))" || { echo "$label $cfg FAILED"; exit 1; }
    done
  done
done
