"""The C-ABI library loads on a box without a GPU, exports every symbol include/phx_xpu.h declares, its
struct layouts match the ctypes mirror, and — with no device — every compute entry point fails loudly
(no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from phosphorus_mk2_amd import xpu
    if not os.path.exists(xpu.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return xpu.load_library()


def test_exports_every_declared_symbol(lib):
    from phosphorus_mk2_amd import abi
    hdr = open(os.path.join(ROOT, "include", "phx_xpu.h")).read()
    declared = set(re.findall(r"\b(phx_[a-z_]+)\s*\(", hdr)) - {"phx_next_tile_fn", "phx_add_tile_fn"}
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    raw = C.CDLL(os.path.join(ROOT, "phosphorus_mk2_amd", "libphx_hip.so"))
    for s in declared:
        assert hasattr(raw, s), s


def test_struct_layouts(lib):
    from phosphorus_mk2_amd import abi
    for i, t in enumerate([abi.Options, abi.Lobe, abi.Material, abi.FaceSet, abi.Mesh, abi.Camera, abi.Scene, abi.Tile, abi.Frame, abi.Stats]):
        assert C.sizeof(t) == lib.phx_abi_sizeof(i), t.__name__


def test_header_is_plain_c():
    """the boundary is a C ABI: the header must compile as C99 with no C++ / torch types"""
    import subprocess
    src = '#include "phx_xpu.h"\nint main(void){ phx_options o; (void)o; return (int)sizeof(phx_scene) == 0; }\n'
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                   input=src.encode(), check=True)


def test_tile_queue_matches_reference_layout(lib):
    """job::tiles_t::make (src/jobs/tiles.hpp:49-89): row-major 32x32 tiles with edge remainders; rank/world shard."""
    from phosphorus_mk2_amd import dist, xpu
    q = xpu.Tiles.make(1280, 720, 32)
    assert len(q) == 40 * 23
    tiles = []
    while True:
        t = q.next()
        if t is None:
            break
        tiles.append(t)
    assert tiles[0] == (0, 0, 32, 32) and tiles[39] == (1248, 0, 32, 32) and tiles[-1] == (1248, 704, 32, 16)
    assert tiles == dist.shard_tiles(1280, 720, 32, 0, 1)
    assert q.next() is None
    q.reset()
    assert q.next() == (0, 0, 32, 32)
    for world in (2, 3, 8):
        got = []
        for r in range(world):
            qr = xpu.Tiles.make(100, 70, 32, r, world)
            mine = []
            while (t := qr.next()) is not None:
                mine.append(t)
            assert mine == dist.shard_tiles(100, 70, 32, r, world)
            got += mine
        assert sorted(got) == sorted(dist.shard_tiles(100, 70, 32, 0, 1))  # a partition of the film
    with pytest.raises(xpu.DeviceError):
        xpu.Tiles.make(64, 64, 32, rank=2, world=2)


def test_tile_shard_is_balanced_and_not_striped(lib):
    """the multi-GPU shard: every rank gets its fair share of tiles and touches every tile column and row band of the
    baseline film (plain 'tile id mod world' gives vertical stripes at 40 tiles per row)"""
    from phosphorus_mk2_amd import dist
    for world in (2, 3, 4, 6, 8):
        counts = []
        for r in range(world):
            mine = dist.shard_tiles(1280, 720, 32, r, world)
            counts.append(len(mine))
            assert len({x for x, _, _, _ in mine}) == 40, (world, r)
            assert len({y for _, y, _, _ in mine}) == 23, (world, r)
        assert sum(counts) == 40 * 23 and max(counts) - min(counts) <= 23, (world, counts)


def test_no_silent_cpu_fallback(lib):
    """Without a visible MI355X the device cannot be made; nothing renders on the host instead."""
    from phosphorus_mk2_amd import xpu
    n = C.c_int(-1)
    o = xpu.Options().pack()
    rc = lib.phx_discover(C.byref(o), C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible: covered by the -m gpu tests")
    assert n.value == 0 and rc != 0 and lib.phx_last_error()
    with pytest.raises(xpu.DeviceError):
        xpu.HipDevice.make(xpu.Options())
    o.host_only = 1
    assert lib.phx_discover(C.byref(o), C.byref(n)) == 0 and n.value == 0  # --no-gpu is not an error
    assert lib.phx_dev_preprocess(None, None) != 0 and lib.phx_dev_join(None) != 0


def test_render_makes_one_device_on_a_multi_gpu_box(monkeypatch):
    """On a box with several GPUs render() (and every single-device script) makes exactly ONE device, with device_ordinal -1 =
    the caller's current GPU; discover() is what makes one per GPU — or only the one asked for."""
    from phosphorus_mk2_amd import scenes, xpu

    class FakeLib:
        def phx_discover(self, opts, n):
            n._obj.value = 4
            return 0

        def phx_last_error(self):
            return b""

    made = []

    class Stop(Exception):
        pass

    def fake_make(options):
        made.append(options.device_ordinal)
        raise Stop()

    monkeypatch.setattr(xpu, "load_library", lambda: FakeLib())
    monkeypatch.setattr(xpu.HipDevice, "make", staticmethod(fake_make))
    with pytest.raises(Stop):
        xpu.render(scenes.cornell(32, 32), spp=1)
    assert made == [-1]
    made.clear()

    def record_make(options):
        made.append(options.device_ordinal)
        return object()

    monkeypatch.setattr(xpu.HipDevice, "make", staticmethod(record_make))
    assert len(xpu.HipDevice.discover(xpu.Options())) == 4 and made == [0, 1, 2, 3]
    made.clear()
    assert len(xpu.HipDevice.discover(xpu.Options(device_ordinal=2))) == 1 and made == [2]


def test_reference_patch_applies():
    """integration/reference.patch (the maintainer's change list: xpu_t::discover, four mesh_t accessors, the CMake source list,
    parsed_options_t::host_only) still applies to the reference tree — a dry run, nothing is written"""
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src")) or not shutil.which("patch"):
        pytest.skip("no reference tree on this box")
    r = subprocess.run(["patch", "-p1", "--dry-run", "--batch", "-d", ref, "-i", os.path.join(ROOT, "integration", "reference.patch")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("checking file") == 5
