"""The N>1 path on CPU: two processes (gloo), tiles interleaved over the ranks, every rank renders its
shard into a zero film (with the oracle standing in for the device on this GPU-less box), ONE reduce(sum)
to rank 0 — which must equal a single-process render bit for bit (disjoint tiles: x + 0)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    from oracle import oracle as orc
    from phosphorus_mk2_amd import dist as pdist
    from phosphorus_mk2_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    pdist.init_process_group("gloo", rank, world)
    sc = scenes.cornell(96, 80)
    O = orc.Oracle(sc, spp=2)
    tiles = pdist.shard_tiles(96, 80, 32, rank, world)
    film, st = O.render(rng=orc.RNG_COUNTER, seed=4, threads=1, tiles=tiles)
    t = torch.from_numpy(film)
    pdist.reduce_film(t, dst=0)
    rays = pdist.sum_over_ranks(st["rays_closest"] + st["rays_shadow"])
    slow = pdist.max_over_ranks(float(rank + 1))
    if rank == 0:
        np.savez(out_path, film=t.numpy(), rays=rays, slow=slow, ntiles=len(tiles))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_shard_and_film_reduce(tmp_path, orc):
    import torch.multiprocessing as mp
    from phosphorus_mk2_amd import scenes
    out = str(tmp_path / "r0.npz")
    port = 29000 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    full, st = orc.Oracle(scenes.cornell(96, 80), spp=2).render(rng=orc.RNG_COUNTER, seed=4, threads=2)
    assert np.array_equal(got["film"].view(np.uint32), full.view(np.uint32))
    assert got["rays"] == st["rays_closest"] + st["rays_shadow"]
    assert got["slow"] == 2.0 and got["ntiles"] == 5  # 9 tiles: ranks get 5 + 4
