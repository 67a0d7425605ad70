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
    """`python bench.py --gpus 2` itself — the launch decision, the torch.distributed.run child, tile shard by rank, zero-initialised
    device films, ONE reduce to rank 0 inside the bracket, max-over-ranks time, rays summed over ranks, one compact JSON line relayed
    from rank 0 — rehearsed with two ranks on this box's one GPU (PHX_BENCH_REHEARSAL=1: both ranks on GPU 0, the reduce on gloo).
    The traced rays and the film must be those of the N = 1 run, and `value` must be the same experiment at both N: one frame in
    flight, with the two-frames-in-flight throughput beside it."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    common = ["--triangles", "3000", "--width", "160", "--height", "96", "--spp", "9", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    recs = {}
    for n in (1, 2):
        full = os.path.join(ROOT, "gpurun_out", f"bench_rehearsal_{n}.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--full-json", full] + common, capture_output=True, text=True, timeout=300,
                           env=dict(env, PHX_BENCH_REHEARSAL="1") if n > 1 else env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096  # rank 0 alone prints, last, compactly
        recs[n] = (json.loads(lines[0]), json.load(open(full)))
    (d1, f1), (d2, f2) = recs[1], recs[2]
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["scaling"] == "strong" and "rehearsal" in d2 and "rehearsal" not in d1
    assert d2["config"]["film_collective"].startswith("reduce") and d1["config"]["film_collective"] == "none"
    # ONE experiment across N: `value` is one frame in flight at both N, the pipelined rate is its own field at both N
    assert d1["config"]["frames_in_flight"] == d2["config"]["frames_in_flight"] == 1 and d1["value_definition"] == d2["value_definition"]
    assert d1["value_two_frames_in_flight"] > 0 and d2["value_two_frames_in_flight"] > 0 and d1["value_host_film"] > 0 and d2["value_host_film"] is None
    assert d1["config"]["hbm_bytes_per_rank"] > 0 and d2["config"]["hbm_bytes_per_rank"] > 0
    assert d2["config"]["hbm_bytes_per_rank_two_frames_in_flight"] > d2["config"]["hbm_bytes_per_rank"]
    assert d2["config"]["rays_per_step"] == d1["config"]["rays_per_step"] and f2["config"]["camera_samples_per_step"] == 160 * 96 * 9
    assert d2["config"]["film_mean"] == d1["config"]["film_mean"] and d2["config"]["film_finite"]
    assert d2["roofline"]["frac"] is None and d2["cpu_baseline"] is None  # reported at N = 1 only
    assert d1["roofline"]["kernel"] == "k_trace" and d1["roofline"]["avg_launch_ms"] > 0 and d1["roofline"]["stream_GBps"] > 0


@pytest.mark.timeout(300)
def test_bench_launched_with_a_mismatched_world_fails():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.timeout(600)
def test_bench_rccl_path_at_world_one():
    """`bench.py --force-dist`: the N > 1 code with the REAL backend (nccl = RCCL) at world size 1 — process group on the device,
    zero-initialised device film, the synchronous film reduce inside the bracket, the asynchronous one in the two-frames-in-flight
    pass, max / sum over ranks on the device — is what one GPU can exercise of the RCCL path.  Same rays and film as the plain run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "PHX_BENCH_REHEARSAL")}
    env["MASTER_ADDR"] = "127.0.0.1"; env["MASTER_PORT"] = str(29900 + (os.getpid() % 90))
    common = ["--triangles", "3000", "--width", "160", "--height", "96", "--spp", "9", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--full-json", os.path.join(ROOT, "gpurun_out", "bench_force_dist.json")]
    recs = []
    for extra in ([], ["--force-dist"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        recs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    plain, forced = recs
    assert forced["n_gpus"] == 1 and forced["config"]["film_collective"].startswith("reduce") and plain["config"]["film_collective"] == "none"
    assert forced["config"]["rays_per_step"] == plain["config"]["rays_per_step"] and forced["config"]["film_mean"] == plain["config"]["film_mean"]
    assert forced["value"] > 0 and forced["value_two_frames_in_flight"] > 0 and forced["value_host_film"] is None
