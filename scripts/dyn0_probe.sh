R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { label=$1; shift
  for tri in 100000 1000000; do
    env "$@" python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --triangles $tri 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label'.ljust(28), '$tri'.rjust(8), round(d['value']), 'trace %.2f shade %.2f' % (d['config']['kernel_ms_per_step']['trace'], d['config']['kernel_ms_per_step']['shade_gen_film']), d['config']['plan'])"
  done
}
run dyn PHX_NONE=0
run static_b1024_g4 PHX_TRACE_DYN=0 PHX_TRACE_BLOCK=1024
run static_b1024_g8 PHX_TRACE_DYN=0 PHX_TRACE_BLOCK=1024 PHX_TRACE_GRID=8 PHX_TRACE_GRID0=8
run static_b1024_g2 PHX_TRACE_DYN=0 PHX_TRACE_BLOCK=1024 PHX_TRACE_GRID=2 PHX_TRACE_GRID0=4
run static_b256 PHX_TRACE_DYN=0
run dyn2 PHX_NONE=0
