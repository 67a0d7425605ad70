"""Run ON the GPU box with PHX_HOST_TIMING=1: bench.py's config-5 secondary record alone (one warm-up on every 32nd tile, one timed full frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
import bench
from phosphorus_mk2_amd import scenes, xpu
xpu.load_library()
t0 = time.time()
value, ms, acc, st, pre, scene, film, vh = bench.run_workload(xpu, scenes, "zoo", 500000, 3840, 2160, 4096, 9, 1, "auto", steps=1, warmup=1, shard=(0, 1), host_pass=False, warmup_shard=(0, 32))
print("value", value, "ms", ms, "kernels", bench.kernel_ms(acc, 1), "frame_ms", acc["frame_ms"], "total s", time.time() - t0)
