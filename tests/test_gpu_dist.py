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
