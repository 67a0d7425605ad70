#!/bin/bash
# Run ON the GPU box (one GPU): every piece of bench.py's N > 1 path that one GPU can exercise — the two-rank tests, the RCCL code at world 1
# (--force-dist: two frames in flight, asynchronous film reduce), and a four-rank rehearsal on gloo (all ranks on GPU 0; not a scaling result).
R=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'Mrays/s', '%.2f ms per step' % d['ms_per_step'], 'frames in flight', d['config']['frames_in_flight'], 'film_mean %.9g' % d['config']['film_mean'], 'rays per step', int(d['config']['rays_per_step']))"; }
timeout -k 10 400 python3 -m pytest $R/tests/test_gpu_dist.py -x -q 2>&1 | tail -2
timeout -k 10 200 python3 $R/bench.py --one-sink --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | show "N=1 (one frame at a time)"
timeout -k 10 200 python3 $R/bench.py --force-dist --steps 8 --warmup 2 2> $R/gpurun_out/force_dist.err | show "force-dist (RCCL, world 1)"
PHX_BENCH_REHEARSAL=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 $R/bench.py --gpus 4 --steps 5 --warmup 1 2>/dev/null | show "rehearsal, 4 ranks on GPU 0 (gloo)"
