R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for lib in "" "_nopf"; do
PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip$lib.so python3 $R/scripts/run_config.py --scene glassroom --triangles 400000 --width 1280 --height 720 --spp 256 --frames 3 | python3 -c "import json,sys; d=json.load(sys.stdin); print('lib$lib', round(d['Mrays_per_s']), 'frame %.1f k_trace %.1f shade %.2f' % (d['frame_s']*1e3, d['k_trace_ms'], d['shade_kernel_ms']), d['film_sha1'])"
done; done
