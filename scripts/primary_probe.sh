#!/bin/bash
# Run ON the GPU box: k_trace_primary with 1 / 2 / 4 rays per lane (PHX_PRIMARY_RPL; unset = the launcher's rule) on the bench frame, the
# 1 M soup, config 4 at 64 spp, the Cornell box, the BMW stand-in and a 16-spp frame, and the packet statistics of the instrumented build.
# (profiles/r03_zo_primary_packets.log also holds the A/B against the per-lane path of the camera rays that k_trace had before.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
one() { python3 -c "import json,sys; d=json.load(sys.stdin); print('$1', 'frame %.2f ms' % (d['frame_s']*1e3), 'trace %.2f (primary %.2f) shade %.2f' % (d['trace_ms'], d.get('primary_ms', 0.0), d['shade_ms']), round(d['Mrays_per_s']), 'film_mean %.9g' % d['film_mean'])"; }
for rep in 1 2; do for rpl in 1 2 4; do
  export PHX_PRIMARY_RPL=$rpl
  python3 $R/scripts/run_config.py --scene soup --triangles 100000 --spp 256 --frames 4 | one "rpl=$rpl soup 100000"
  python3 $R/scripts/run_config.py --scene soup --triangles 1000000 --spp 256 --frames 4 | one "rpl=$rpl soup 1000000"
done; done
for rpl in 1 2 4; do
  export PHX_PRIMARY_RPL=$rpl
  python3 $R/scripts/run_config.py --scene soup --triangles 10000000 --width 3840 --height 2160 --spp 64 --frames 3 | one "rpl=$rpl config 4 (64 spp)"
  python3 $R/scripts/run_config.py --scene cornell --width 1024 --height 1024 --spp 256 --frames 3 | one "rpl=$rpl cornell"
  python3 $R/scripts/run_config.py --scene zoo --triangles 500000 --width 1920 --height 1080 --spp 256 --frames 3 | one "rpl=$rpl zoo"
  python3 $R/scripts/run_config.py --scene soup --triangles 100000 --spp 16 --frames 4 | one "rpl=$rpl soup 100000 16 spp"
  python3 $R/scripts/count_work.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rpl=$rpl', d['primary'])"
done
