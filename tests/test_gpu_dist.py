"""The N > 1 path with the REAL device (one GPU is what a test box has): two processes, both on GPU 0, each renders the tiles
job::tiles_t::make gives its rank (src/jobs/tiles.hpp:40-89 + the rank interleave) on the HIP device into a zero-initialised film;
the films are summed onto rank 0 with one reduce (gloo, on the host copies — two RCCL ranks cannot share one GPU) and must equal
the one-process film bit for bit, ray totals included (src/core.cpp:103-115: devices share nothing but the tile queue and the film).
No 1 -> 8 curve is measured here or anywhere in this repo's own runs: that needs an 8-GPU node."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, bits_equal

pytestmark = pytest.mark.gpu
W, H, SPP, SEED = 160, 112, 9, 11


def _scene():
    from phosphorus_mk2_amd import scenes
    return scenes.multi_material_soup(3000, width=W, height=H)  # general closures: k_shade_g on both ranks


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    from phosphorus_mk2_amd import dist as pdist
    from phosphorus_mk2_amd import xpu
    pdist.init_process_group("gloo", rank, world)
    sc = _scene()
    dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=SPP, paths_per_sample=1, path_depth=9, device_ordinal=0))
    dev.preprocess(sc)
    tiles = xpu.Tiles.make(W, H, 32, rank, world)
    film = xpu.Film(W, H, 4)  # zero-initialised
    dev.start(sc, xpu.FrameState(SEED, tiles, film, native_sink=True)); dev.join()
    st = dev.stats()
    dev.close()
    t = torch.from_numpy(film.data)
    pdist.reduce_film(t, dst=0)  # the frame's single collective
    closest = pdist.sum_over_ranks(st["rays_closest"]); shadow = pdist.sum_over_ranks(st["rays_shadow"])
    ntiles = pdist.sum_over_ranks(st["tiles"])
    if rank == 0:
        np.savez(out_path, film=t.numpy(), closest=closest, shadow=shadow, ntiles=ntiles, mine=len(tiles))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_sum_to_the_one_process_film(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.npz")
    port = 29500 + (os.getpid() % 2000)
    # fresh children (spawn): each makes its own HIP context on GPU 0
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    from phosphorus_mk2_amd import xpu
    full, st = xpu.render(_scene(), spp=SPP, pps=1, depth=9, seed=SEED, native_sink=True)
    assert bits_equal(got["film"], full) and full[..., :3].max() > 0.05
    assert got["closest"] == st["rays_closest"] and got["shadow"] == st["rays_shadow"]
    assert got["ntiles"] == st["tiles"] == 20 and got["mine"] == 10  # 5 x 4 tiles, half of them per rank


@pytest.mark.timeout(600)
def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """bench.py's N > 1 code — torch.distributed.run launch, tile shard by rank, zero-initialised device films, ONE reduce to rank 0,
    max-over-ranks time, rays summed over ranks, one JSON line from rank 0 — rehearsed with two ranks on this box's one GPU
    (PHX_BENCH_REHEARSAL=1: both ranks on GPU 0, the reduce on gloo).  The traced rays and the film must be those of the N = 1 run."""
    import json
    import subprocess
    common = ["--triangles", "3000", "--width", "160", "--height", "96", "--spp", "9", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--one-sink"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    env = dict(os.environ, PHX_BENCH_REHEARSAL="1")
    port = 29700 + (os.getpid() % 200)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + [x if x not in ("1", "0") else {"1": "3", "0": "1"}[x] for x in common],  # 3 steps after 1 warm-up: two frames in flight
                         capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["config"]["film_collective"].startswith("reduce") and "rehearsal" in d2 and d2["config"]["frames_in_flight"] == 2
    assert d2["config"]["rays_per_step"] == d1["config"]["rays_per_step"] and d2["config"]["camera_samples_per_step"] == 160 * 96 * 9
    assert d2["config"]["film_mean"] == d1["config"]["film_mean"] and d2["config"]["film_finite"]
    assert d2["roofline"]["frac"] is None and d2["cpu_baseline"] is None  # reported at N = 1 only
