"""Scene ingestion: the reference's YAML scene file (src/codecs/scene.cpp:41-76) with the
same top-level keys — `materials` (shader graphs, baked by closures.py), `data` (geometry files), `camera`,
`world.environment` — feeding the same `scene -> preprocess` step as the synthetic scenes.

Differences, by necessity: geometry comes from Wavefront OBJ instead of Alembic (`.abc` needs the Alembic library,
absent here); the camera block is completed into a look-at matrix (the reference's YAML camera decoder reads
position/at/up and then drops them, src/codecs/scene/entities.hpp:18-33) with the Alembic importer's convention
`fov = 2*atan2(sensor_width/2, focal_length)` (src/codecs/scene/alembic.hpp:69); an optional `dof: {fstop, focus-distance}` block
turns the thin lens on with the Blender importer's convention (plugins/blender/import.hpp:573-579).

OBJ subset: v, vn, f (polygons are fan-triangulated; v//vn and v/vt/vn index forms), usemtl NAME (one face set per
material, material ids = order of the YAML `materials` map, src/scene.cpp:84-90), `s off|0` / `s 1` (flat / smooth).
"""
import math
import os

import numpy as np
import yaml

from . import abi, closures
from .scenes import CameraDesc, MeshDesc, SceneDesc


def load_obj(path, material_ids, default_material=0):
    verts, norms, faces, fnorm, smooth, fmat = [], [], [], [], [], []
    cur_mat, cur_smooth = default_material, False
    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v":
            verts.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            norms.append([float(x) for x in t[1:4]])
        elif t[0] == "usemtl":
            if t[1] not in material_ids:
                raise ValueError(f"{path}: usemtl {t[1]!r} is not in the scene's materials")
            cur_mat = material_ids[t[1]]
        elif t[0] == "s":
            cur_smooth = t[1] not in ("off", "0")
        elif t[0] == "f":
            idx = []
            for tok in t[1:]:
                p = tok.split("/")
                vi = int(p[0]); vi = vi - 1 if vi > 0 else len(verts) + vi
                ni = None
                if len(p) == 3 and p[2]:
                    ni = int(p[2]); ni = ni - 1 if ni > 0 else len(norms) + ni
                idx.append((vi, ni))
            for k in range(1, len(idx) - 1):
                tri = (idx[0], idx[k], idx[k + 1])
                faces.append([v for v, _ in tri])
                fnorm.append([n for _, n in tri])
                smooth.append(1 if (cur_smooth and all(n is not None for _, n in tri)) else 0)
                fmat.append(cur_mat)
    if not faces:
        raise ValueError(f"{path}: no faces")
    faces = np.array(faces, np.uint32)
    verts = np.array(verts, np.float32)
    fmat = np.array(fmat)
    # normals per face corner (mesh_t without NormalsPerVertex, src/mesh.cpp:188-192): index 3*face + corner
    nrm = np.zeros((len(faces) * 3, 3), np.float32)
    na = np.array(norms, np.float32) if norms else np.zeros((0, 3), np.float32)
    for f, tri in enumerate(fnorm):
        for c, n in enumerate(tri):
            if n is not None:
                nrm[3 * f + c] = na[n]
    sets = [(int(m), np.nonzero(fmat == m)[0].astype(np.uint32)) for m in sorted(set(fmat.tolist()))]
    return MeshDesc(vertices=verts, faces=faces, normals=nrm, smooth=np.array(smooth, np.uint8), sets=sets, flags=abi.MESH_UV_PER_VERTEX)


def look_at(position, at, up):
    """camera-to-world in Imath's row-vector convention (v' = v * M): the camera looks down -z, +y is up (camera.hpp:86-88)"""
    p, a, u = (np.asarray(x, np.float64) for x in (position, at, up))
    z = p - a; z /= np.linalg.norm(z)
    x = np.cross(u, z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3], m[3, :3] = x, y, z, p
    return m.astype(np.float32)


def load_scene(path, width=1280, height=720):
    cfg = yaml.safe_load(open(path))
    base = os.path.dirname(os.path.abspath(path))
    baked = closures.bake_materials(cfg["materials"])
    names = list(baked)
    ids = {n: i for i, n in enumerate(names)}
    meshes = [load_obj(os.path.join(base, d["path"]), ids, ids.get(d.get("material", names[0]), 0)) for d in cfg.get("data", [])]
    cam = cfg.get("camera", {}) or {}
    focal, sensor = float(cam.get("focal-length", 35.0)), float(cam.get("sensor-width", 32.0))
    fov = 2.0 * math.atan2(sensor / 2.0, focal)
    to_world = look_at(cam.get("position", (0, 0, 0)), cam.get("at", (0, 0, -1)), cam.get("up", (0, 1, 0)))
    film = cam.get("film", {}) or {}
    camera = CameraDesc(int(film.get("width", width)), int(film.get("height", height)), fov, to_world)
    dof = cam.get("dof") or {}
    if dof:  # the Blender importer's depth of field (plugins/blender/import.hpp:573-579): focal length in mm, radius = lens / (2 f-stop) in m
        fstop = max(float(dof.get("fstop", 2.8)), 1e-5)
        camera.aperture_radius = (focal * 1e-3) / (2.0 * fstop)
        camera.focal_distance = float(dof.get("focus-distance", 1.0))
    env = -1
    world = cfg.get("world") or {}
    if "environment" in world:
        env = ids[world["environment"]]  # import_world_data, scene.cpp:30-36
    return SceneDesc(meshes, [baked[n] for n in names], camera, environment_material=env, name=os.path.basename(path))


def save_pfm(path, rgb):
    """film sink to disk (the reference writes EXR through OpenImageIO, src/film/file.cpp:43-46; PFM needs no library)"""
    img = np.ascontiguousarray(rgb[::-1, :, :3], np.float32)
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def save_exr(path, film):
    """film::file_t::finalize (src/film/file.cpp:27-46) writes the 4-component float image as OpenEXR through OpenImageIO;
    this writes the same image as an uncompressed scanline OpenEXR 2 file (FLOAT channels A, B, G, R; a film without
    alpha gets A = 1) with nothing but numpy.  `film`: H x W x (3|4+) float32, row 0 = top row."""
    import struct
    img = np.asarray(film, np.float32)
    h, w = img.shape[:2]
    planes = {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2],
              "A": img[..., 3] if img.shape[2] >= 4 else np.ones((h, w), np.float32)}
    names = sorted(planes)  # the channel list is sorted by name and so is the data of a scanline

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in names) + b"\0"  # 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0") +
              attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
              attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    row_bytes = 4 * w * len(names)
    first = len(header) + 8 * h
    offsets = first + (8 + row_bytes) * np.arange(h, dtype=np.uint64)
    rows = np.stack([np.ascontiguousarray(planes[n], "<f4") for n in names], axis=1)  # h x channels x w
    with open(path, "wb") as f:
        f.write(header)
        f.write(offsets.astype("<u8").tobytes())
        for y in range(h):
            f.write(struct.pack("<ii", y, row_bytes))
            f.write(rows[y].tobytes())
