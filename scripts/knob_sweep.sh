#!/bin/bash
# Run ON the GPU box: the tuning knobs of EXPERIMENTS.md Part B section 7 against the defaults, on the bench frame (100 k) and the 1 M soup
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # label, env assignments...
  label=$1; shift
  for tri in 100000 1000000; do
    env "$@" python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --one-sink --triangles $tri 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label'.ljust(24), '$tri'.rjust(8), round(d['value']), 'trace %.2f shade %.2f' % (d['config']['kernel_ms_per_step']['trace'], d['config']['kernel_ms_per_step']['shade']))"
  done
}
run default PHX_NONE=0
run refill4 PHX_REFILL=4
run refill8 PHX_REFILL=8
run refill16 PHX_REFILL=16
run refill24 PHX_REFILL=24
run refill32 PHX_REFILL=32
run chunks2 PHX_TARGET_CHUNKS=2
run chunks8 PHX_TARGET_CHUNKS=8
run minchunks4 PHX_MIN_CHUNKS=4
run minchunks16 PHX_MIN_CHUNKS=16
run block512 PHX_TRACE_BLOCK=512
run dyngrid2 PHX_TRACE_DYN_GRID=2
run default2 PHX_NONE=0
