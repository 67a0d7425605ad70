"""Scene ingestion (phosphorus_mk2_amd/sceneio.py): the reference's YAML scene schema + OBJ geometry -> SceneDesc.
Fixture: tests/golden/room/{scene.yaml,room.obj} (hand-written, 9 triangles)."""
import os

import numpy as np

from conftest import ROOT

ROOM = os.path.join(ROOT, "tests", "golden", "room", "scene.yaml")


def test_yaml_obj_scene_loads_and_renders(tmp_path, orc):
    from phosphorus_mk2_amd import abi, sceneio
    sc = sceneio.load_scene(ROOM)
    assert sc.camera.width == 64 and abs(sc.camera.fov - 2 * np.arctan2(18.0, 35.0)) < 1e-6
    assert np.allclose(sc.camera.to_world[3, :3], (0, 0, 3)) and np.allclose(sc.camera.to_world[2, :3], (0, 0, 1))
    assert sc.environment_material == 3 and sc.materials[2].is_emitter and len(sc.materials) == 4
    m = sc.meshes[0]
    assert len(m.faces) == 4 + 2 + 3 and [s[0] for s in m.sets] == [0, 1, 2] and [len(s[1]) for s in m.sets] == [4, 3, 2]
    assert m.smooth.tolist() == [0, 0, 0, 0, 0, 0, 1, 1, 1] and not (m.flags & abi.MESH_NORMALS_PER_VERTEX)
    assert np.allclose(m.normals[3 * 6], (0, 1, 0)) and np.allclose(m.normals[3 * 6 + 1], (0.7, 0.3, 0.3))
    film, st = orc.Oracle(sc, spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=2)
    assert np.isfinite(film).all() and film[..., :3].max() > 0.1
    sceneio.save_pfm(str(tmp_path / "out.pfm"), film)
    assert open(tmp_path / "out.pfm", "rb").read(16).startswith(b"PF\n64 48\n")


def test_dof_block_turns_the_thin_lens_on_with_the_blender_convention(tmp_path, orc):
    """plugins/blender/import.hpp:573-579: aperture_radius = lens[mm] * 1e-3 / (2 * max(fstop, 1e-5)), focal_distance = the focus distance"""
    import shutil
    import yaml
    from phosphorus_mk2_amd import sceneio
    shutil.copy(os.path.join(os.path.dirname(ROOM), "room.obj"), tmp_path / "room.obj")
    cfg = yaml.safe_load(open(ROOM))
    cfg["camera"]["dof"] = {"fstop": 1.4, "focus-distance": 3.0}
    (tmp_path / "scene.yaml").write_text(yaml.safe_dump(cfg, sort_keys=False))
    sc = sceneio.load_scene(str(tmp_path / "scene.yaml"))
    focal = float(cfg["camera"].get("focal-length", 35.0))
    assert abs(sc.camera.aperture_radius - focal * 1e-3 / 2.8) < 1e-9 and sc.camera.focal_distance == 3.0
    assert sceneio.load_scene(ROOM).camera.aperture_radius == 0.0
    a, _ = orc.Oracle(sc, spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=2)
    b, _ = orc.Oracle(sceneio.load_scene(ROOM), spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=2)
    assert np.isfinite(a).all() and (a != b).any() and abs(a[..., :3].mean() / b[..., :3].mean() - 1) < 0.1


def test_bottom_up_film_sink_flips_tiles_like_the_blender_sink():
    """plugins/blender/sink.cpp:34-69: tile (x, y) lands at row height-h-y, rows reversed, primary alpha = 1"""
    import ctypes as C
    from phosphorus_mk2_amd import xpu
    W, H = 8, 6
    top, bottom = xpu.Film(W, H, 4), xpu.BottomUpFilm(W, H, 4)
    rng = np.random.default_rng(0)
    for (x, y, w, h) in [(0, 0, 4, 4), (4, 0, 4, 4), (0, 4, 4, 2), (4, 4, 4, 2)]:
        tile = rng.random((h, w, 4)).astype(np.float32)
        ptr = tile.ctypes.data_as(C.POINTER(C.c_float))
        top.add_tile(x, y, w, h, ptr, 4, 4 * w)
        bottom.add_tile(x, y, w, h, ptr, 4, 4 * w)
    assert np.array_equal(bottom.data[..., :3], top.data[::-1, :, :3])
    assert (bottom.data[..., 3] == 1.0).all() and bottom.tiles == 4


def test_exr_writer_round_trips_through_a_minimal_reader(tmp_path):
    """uncompressed scanline OpenEXR: magic, sorted FLOAT channel list, one offset per scanline, channel-planar rows"""
    import struct
    from phosphorus_mk2_amd import sceneio
    rng = np.random.default_rng(3)
    film = rng.random((5, 7, 4)).astype(np.float32)
    p = str(tmp_path / "film.exr")
    sceneio.save_exr(p, film)
    b = open(p, "rb").read()
    assert struct.unpack_from("<ii", b, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert struct.unpack("<iiii", attrs["dataWindow"][1]) == (0, 0, 6, 4)
    ch, q, names = attrs["channels"][1], 0, []
    while ch[q] != 0:
        e = ch.index(b"\0", q); names.append(ch[q:e].decode()); q = e + 1
        assert struct.unpack_from("<iB3xii", ch, q) == (2, 0, 1, 1); q += 16
    assert names == ["A", "B", "G", "R"]
    offsets = struct.unpack_from("<5Q", b, pos)
    back = np.zeros_like(film)
    for y, off in enumerate(offsets):
        yy, size = struct.unpack_from("<ii", b, off)
        assert yy == y and size == 4 * 7 * 4
        row = np.frombuffer(b, "<f4", 28, off + 8).reshape(4, 7)
        back[y, :, 3], back[y, :, 2], back[y, :, 1], back[y, :, 0] = row[0], row[1], row[2], row[3]
    assert np.array_equal(back, film) and offsets[-1] + 8 + 112 == len(b)
