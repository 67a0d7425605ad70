R=$(pwd)
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_sortp.so python3 $R/scripts/run_config.py "$@" --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('%-26s frame %7.1f ms  k_trace %7.1f  shade %6.1f  other(+sort) %6.1f  film %s' % ('$label', d['frame_s']*1e3, d['k_trace_ms'], d['shade_kernel_ms'], d['other_ms'], d['film_sha1']))"; }
for scene in "--scene soup --triangles 100000 --width 1280 --height 720 --spp 256" "--scene soup --triangles 1000000 --width 1280 --height 720 --spp 256"; do
  echo "== $scene"
  run "inherited order" PHX_X=0 -- $scene
  run "path-id bins, from 1" PHX_SORT_RAYS=29 PHX_SORT_FROM=1 -- $scene
  run "path-id bins, from 2" PHX_SORT_RAYS=29 PHX_SORT_FROM=2 -- $scene
done
