#!/bin/bash
# Run ON the GPU box: the queue-wide ray sort (PHX_SORT_RAYS = bits per axis of the origin cell, PHX_SORT_OCT = direction octant in the
# key, PHX_SORT_FROM = first bounce whose rays are sorted) against the unsorted queue on the 100 k / 1 M soups and the config-4 frame.
# Prints per variant: frame ms, k_trace ms, shade ms, "other" ms (begin pass + film + THE SORT LAUNCHES), and the film's hash.
# The sort exists in probe builds only: git apply profiles/r06_k_sort_probe.patch && make -C phosphorus_mk2_amd/csrc variant NAME=sortprobe EXTRA=-DPHX_SORT_PROBE=1
# (PHX_LIB is set below); the product sources do not carry it.
# SORT_SCENES overrides the scene list (round 6: the closed showroom, whose k_trace hits L2 at 0.54).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_sortprobe.so
run() {  # label, env..., then -- args
  label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 $R/scripts/run_config.py "$@" --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('%-22s frame %7.1f ms  k_trace %7.1f  shade %6.1f  other(+sort) %6.1f  %5.0f Mrays/s  film %s' % ('$label', d['frame_s']*1e3, d['k_trace_ms'], d['shade_kernel_ms'], d['other_ms'], d['Mrays_per_s'], d['film_sha1']))"
}
IFS='|' read -r -a SCENES <<< "${SORT_SCENES:---scene soup --triangles 100000 --width 1280 --height 720 --spp 256|--scene soup --triangles 1000000 --width 1280 --height 720 --spp 256|--scene soup --triangles 10000000 --width 3840 --height 2160 --spp ${C4_SPP:-64}}"
for scene in "${SCENES[@]}"; do
  echo "== $scene"
  run "unsorted" PHX_X=0 -- $scene
  for v in ${SORT_VARIANTS:-3_1_1 4_1_1 4_0_1 4_1_2 3_0_1}; do  # bits_octant_firstbounce
    set -- ${v//_/ }
    run "bits $1 oct $2 from $3" PHX_SORT_RAYS=$1 PHX_SORT_OCT=$2 PHX_SORT_FROM=$3 -- $scene
  done
done
