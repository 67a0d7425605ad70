#!/bin/bash
# Run ON the GPU box: the distributed code path at world 1, then rank 0's share of the bench frame at world 1/2/4/8 (one GPU).
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/bench.py --force-dist --steps 2 --warmup 1 > $R/gpurun_out/force_dist.json 2> $R/gpurun_out/force_dist.err || { echo force-dist failed; tail -5 $R/gpurun_out/force_dist.err; exit 1; }
python3 -c "import json; d=json.loads(open('$R/gpurun_out/force_dist.json').read().strip().splitlines()[-1]); print('force-dist', round(d['value']), d['ms_per_step'], d['config']['film_collective'])"
for w in 1 2 4 8; do
  python3 $R/scripts/run_config.py --scene soup --triangles 100000 --spp 256 --frames 4 --world $w --rank 0 | python3 -c "import json,sys; d=json.load(sys.stdin); print('world $w rank 0: tiles', d['tiles'], 'frame %.2f ms' % (d['frame_s']*1e3), 'trace %.2f shade %.2f' % (d['trace_ms'], d['shade_ms']), round(d['Mrays_per_s']))"
done
