#!/bin/bash
# Run ON the GPU box: the tuning knobs of EXPERIMENTS.md Part B section 7 change speed, never results — a slice of the parity suite under each
R=${GRAFT_REPO_ROOT:-$(pwd)}
for e in "PHX_SHADE_GRID=1" "PHX_TRACE_BLOCK=512" "PHX_TRACE_BLOCK=256" "PHX_LDS_LEVELS=3" "PHX_NTOP=9" "PHX_REFILL=1" "PHX_REFILL=64" "PHX_TARGET_CHUNKS=4 PHX_MIN_CHUNKS=1" "PHX_TRACE_DYN_GRID=2"; do
  echo "== $e"; env $e timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_parity.py -x -q -k "render_cornell or random or showroom or trace_matches or closures_at_film or invariances" 2>&1 | tail -1
done
