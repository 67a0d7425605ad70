"""Multi-GPU plumbing: one process per GPU, tiles interleaved over the ranks, ONE film collective.

The reference has a single parallelism strategy — data parallelism over image tiles through a shared
atomic tile cursor (src/jobs/tiles.hpp:40-47, src/xpu/cpu.cpp:223-238).  Across GPUs the cursor becomes
a static interleave (tile (tx, ty) -> rank (tx + s*ty) % world, diagonals over the film: deterministic, and with the counter-based sampler the
image does not depend on the GPU count), every rank accumulates into its own zero-initialised film and
the films are summed onto rank 0 with one reduce — `backend="nccl"` is RCCL over xGMI on ROCm; the
same code runs on `gloo` for the CPU tests.  Disjoint tiles make the sum exact (x + 0).
"""
import os


def shard_tiles(width, height, tile_size, rank, world):
    """The tiles of job::tiles_t::make (src/jobs/tiles.hpp:49-89) that belong to `rank`."""
    out = []
    s = tile_shift(world)
    for ty, y in enumerate(range(0, height, tile_size)):
        for tx, x in enumerate(range(0, width, tile_size)):
            if (tx + s * ty) % world == rank:
                out.append((x, y, min(tile_size, width - x), min(tile_size, height - y)))
    return out


def tile_shift(world):
    """owner of tile (tx, ty) = (tx + s*ty) mod world with s the smallest odd number >= 3 coprime to world: diagonals over the
    whole film instead of the vertical stripes "tile id mod world" gives when the row length is a multiple of world."""
    import math
    s = 3
    while math.gcd(s, world) != 1:
        s += 2
    return s


def init_process_group(backend, rank, world, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def reduce_film(film, dst=0, async_op=False):
    """The single collective of a frame: sum the per-rank films onto rank `dst` (in place).  async_op: returns the work handle instead
    of the film — wait() on it before the film is read or cleared (bench.py overlaps the reduce with the next frame's rendering)."""
    import torch.distributed as dist
    work = dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)
    return work if async_op else film


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
