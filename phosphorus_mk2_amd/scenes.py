"""Scene descriptions handed to a device: the host-side stand-in for what the reference builds through
`mesh_t::builder_t` (src/mesh.hpp:46-67) + `scene_t::add` (src/scene.cpp:64-90), and the synthetic
inputs of SURVEY §8(d): the Cornell box (C1) and Soup(N, seed) triangle soups (C2/C4).

A `SceneDesc` owns numpy arrays; `pack()` builds the ctypes `phx_scene` that points into them.
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import abi


@dataclass
class LobeDesc:
    type: int
    weight: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    alpha: float = 0.0
    eta: float = 0.0
    xalpha: float = 0.0
    yalpha: float = 0.0
    refract: int = 0
    r: float = 0.0
    fac_mode: int = 0          # abi.FAC_*: per-hit Fresnel mix factor on this closure's weight (glass)
    fac_ior: float = 0.0
    pre_weight: Tuple[float, float, float] = (1.0, 1.0, 1.0)


@dataclass
class MaterialDesc:
    lobes: List[LobeDesc] = field(default_factory=list)
    emission: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    is_emitter: bool = False


@dataclass
class MeshDesc:
    vertices: np.ndarray  # (n,3) f32
    faces: np.ndarray     # (k,3) u32
    sets: List[Tuple[int, np.ndarray]]  # (material, face indices u32)
    normals: np.ndarray = None  # (m,3) f32
    smooth: np.ndarray = None   # (k,) u8
    flags: int = abi.MESH_UV_PER_VERTEX | abi.MESH_NORMALS_PER_VERTEX

    def __post_init__(self):
        self.vertices = np.ascontiguousarray(self.vertices, dtype=np.float32).reshape(-1, 3)
        self.faces = np.ascontiguousarray(self.faces, dtype=np.uint32).reshape(-1, 3)
        if self.normals is None:
            self.normals = np.zeros((0, 3), np.float32)
        self.normals = np.ascontiguousarray(self.normals, dtype=np.float32).reshape(-1, 3)
        if self.smooth is None:
            self.smooth = np.zeros(len(self.faces), np.uint8)
        self.smooth = np.ascontiguousarray(self.smooth, dtype=np.uint8)
        self.sets = [(int(m), np.ascontiguousarray(f, dtype=np.uint32)) for m, f in self.sets]


@dataclass
class CameraDesc:
    width: int
    height: int
    fov: float = 1.9
    to_world: np.ndarray = None  # Imath M44f x[i][j], row-vector convention
    focal_distance: float = 1.0
    aperture_radius: float = 0.0

    def __post_init__(self):
        if self.to_world is None:
            self.to_world = np.eye(4, dtype=np.float32)
        self.to_world = np.ascontiguousarray(self.to_world, dtype=np.float32).reshape(4, 4)


@dataclass
class SceneDesc:
    meshes: List[MeshDesc]
    materials: List[MaterialDesc]
    camera: CameraDesc
    environment_material: int = -1
    name: str = "scene"

    @property
    def num_triangles(self):
        return sum(sum(len(f) for _, f in m.sets) for m in self.meshes)

    def pack(self):
        """-> (abi.Scene, keepalive).  The ctypes struct borrows the numpy buffers."""
        keep = []
        mats = (abi.Material * len(self.materials))()
        for i, m in enumerate(self.materials):
            mats[i].num_lobes = len(m.lobes)
            mats[i].is_emitter = 1 if m.is_emitter else 0
            mats[i].emission[:] = [np.float32(x) for x in m.emission]
            for j, l in enumerate(m.lobes):
                d = mats[i].lobes[j]
                d.type = l.type
                d.weight[:] = [np.float32(x) for x in l.weight]
                d.alpha, d.eta, d.xalpha, d.yalpha, d.refract, d.r = l.alpha, l.eta, l.xalpha, l.yalpha, l.refract, l.r
                d.fac_mode, d.fac_ior = l.fac_mode, l.fac_ior
                d.pre_weight[:] = [np.float32(x) for x in l.pre_weight]
        meshes = (abi.Mesh * len(self.meshes))()
        for i, m in enumerate(self.meshes):
            sets = (abi.FaceSet * len(m.sets))()
            for k, (mat, faces) in enumerate(m.sets):
                sets[k].material = mat
                sets[k].num_faces = len(faces)
                sets[k].faces = faces.ctypes.data_as(abi.u32p)
            keep.append(sets)
            meshes[i].vertices = m.vertices.ctypes.data_as(abi.f32p)
            meshes[i].num_vertices = len(m.vertices)
            meshes[i].normals = m.normals.ctypes.data_as(abi.f32p)
            meshes[i].num_normals = len(m.normals)
            meshes[i].faces = m.faces.ctypes.data_as(abi.u32p)
            meshes[i].num_faces = len(m.faces)
            meshes[i].smooth = m.smooth.ctypes.data_as(abi.u8p)
            meshes[i].flags = m.flags
            meshes[i].num_sets = len(m.sets)
            meshes[i].sets = sets
        s = abi.Scene()
        s.num_meshes = len(self.meshes)
        s.meshes = meshes
        s.num_materials = len(self.materials)
        s.materials = mats
        s.environment_material = self.environment_material
        s.camera.to_world[:] = [float(x) for x in self.camera.to_world.reshape(-1)]
        s.camera.fov = self.camera.fov
        s.camera.focal_distance = self.camera.focal_distance
        s.camera.aperture_radius = self.camera.aperture_radius
        s.camera.film_width = self.camera.width
        s.camera.film_height = self.camera.height
        keep += [mats, meshes, self]
        return s, keep


def diffuse(r, g, b):
    return MaterialDesc(lobes=[LobeDesc(abi.LOBE_DIFFUSE, (r, g, b))])


def emitter(r, g, b):
    # diffuse_emitter_node.osl:18 -> only an emission() closure: 0 lobes, e = weight
    return MaterialDesc(lobes=[], emission=(r, g, b), is_emitter=True)


def _quad(a, b, c, d, material):
    """One quad = its own mesh, 4 vertices, two flat faces (a,b,c),(a,c,d), one face set."""
    v = np.array([a, b, c, d], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    return MeshDesc(vertices=v, faces=f, sets=[(material, np.array([0, 1], np.uint32))])


LE = (17.0 / math.pi, 12.0 / math.pi, 4.0 / math.pi)


def cornell(width=256, height=256, fov=1.9):
    """C1: 5 diffuse quads + 1 emissive quad (12 triangles), camera at the origin looking down -z.

    Geometric normals ((b-a) x (c-a), never flipped: src/mesh.cpp:203-215) face into the box so that
    next-event estimation is not masked (src/kernels/cpu/spt.hpp:138-141).
    """
    W, R, G, L = 0, 1, 2, 3
    mats = [diffuse(0.73, 0.73, 0.73), diffuse(0.65, 0.05, 0.05), diffuse(0.12, 0.45, 0.15), emitter(*LE)]
    x0, x1, y0, y1, zf, zb = -1.0, 1.0, -1.0, 1.0, -1.5, -3.5
    meshes = [
        _quad((x0, y0, zf), (x1, y0, zf), (x1, y0, zb), (x0, y0, zb), W),   # floor, n = +y
        _quad((x0, y1, zf), (x0, y1, zb), (x1, y1, zb), (x1, y1, zf), W),   # ceiling, n = -y
        _quad((x0, y0, zb), (x1, y0, zb), (x1, y1, zb), (x0, y1, zb), W),   # back, n = +z
        _quad((x0, y0, zf), (x0, y0, zb), (x0, y1, zb), (x0, y1, zf), R),   # left, n = +x
        _quad((x1, y0, zf), (x1, y1, zf), (x1, y1, zb), (x1, y0, zb), G),   # right, n = -x
        _quad((-0.25, 0.99, -2.25), (-0.25, 0.99, -2.75), (0.25, 0.99, -2.75), (0.25, 0.99, -2.25), L),  # lamp, n = -y
    ]
    return SceneDesc(meshes, mats, CameraDesc(width, height, fov), name="cornell")


def _u01(idx, seed):
    """Counter-based uniform [0,1) with 24 bits: splitmix64 finaliser of (seed, idx).  numpy uint64 wraps."""
    with np.errstate(over="ignore"):
        x = (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def soup(n, seed=1234, width=1280, height=720, fov=1.9, light=True, materials=None):
    """Soup(N, seed), SURVEY §8(d): N triangles, centre uniform in [-0.98,0.98]^3 shifted to z=-2.5,
    vertices = centre + e*U(-1,1)^3 with e = 2*N^(-1/3); 12 draws per triangle in the order centre xyz,
    a xyz, b xyz, c xyz; flat faces, unshared vertices; one grey diffuse material (0.73) unless a list
    of materials is given (then triangle i uses material i % len).  Lit by a 4x4 emissive quad at
    y=+1.5 facing down, L_e=(17,12,4)/pi, outside the cloud.
    """
    n = int(n)
    e = np.float32(2.0 * n ** (-1.0 / 3.0))
    verts = np.empty((n, 3, 3), np.float32)
    step = 1 << 16  # triangles per block: the 64-bit hash temporaries of a block stay in cache (10 M triangles: 30 s -> 4 s)
    for t0 in range(0, n, step):
        t1 = min(n, t0 + step)
        idx = np.arange(t0 * 12, t1 * 12, dtype=np.uint64)
        u = (_u01(idx, seed) * np.float32(2.0) - np.float32(1.0)).reshape(t1 - t0, 12)
        centre = u[:, 0:3] * np.float32(0.98)
        centre[:, 2] -= np.float32(2.5)
        for k in range(3):
            verts[t0:t1, k, :] = centre + e * u[:, 3 + 3 * k:6 + 3 * k]
    faces = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
    mats = list(materials) if materials else [diffuse(0.73, 0.73, 0.73)]
    nm = len(mats)
    if nm == 1:
        sets = [(0, np.arange(n, dtype=np.uint32))]
    else:
        sets = [(m, np.arange(m, n, nm, dtype=np.uint32)) for m in range(nm)]
    meshes = [MeshDesc(vertices=verts.reshape(-1, 3), faces=faces, sets=sets)]
    if light:
        mats.append(emitter(*LE))
        meshes.append(_quad((-2.0, 1.5, -0.5), (-2.0, 1.5, -4.5), (2.0, 1.5, -4.5), (2.0, 1.5, -0.5), len(mats) - 1))
    return SceneDesc(meshes, mats, CameraDesc(width, height, fov), name=f"soup{n}")


def stress(seed=7, width=64, height=64):
    """Geometry chosen to break builders and box tests, not to look like anything: a small soup + zero-area triangles
    (collinear, repeated vertices) + ten coincident copies of each of 20 triangles (exact distance ties: the lowest
    primitive index must win) + a chain whose sizes fall from 1 to 2^-20 ("teapot in a stadium") + triangles with
    coordinates up to 1e4 + axis-aligned flat triangles (boxes of zero extent).  One grey material, the soup's light."""
    rng = np.random.default_rng(seed)
    def rnd(n, scale, centre=(0.0, 0.0, -2.5)):
        c = rng.uniform(-0.9, 0.9, (n, 1, 3)) + np.array(centre)
        return (c + rng.uniform(-scale, scale, (n, 3, 3))).astype(np.float32)
    parts = [rnd(600, 0.12)]
    deg = rnd(200, 0.2)
    deg[:100, 2] = deg[:100, 1]                                             # repeated vertex
    deg[100:, 2] = (0.25 * deg[100:, 0] + 0.75 * deg[100:, 1]).astype(np.float32)  # (nearly) collinear
    parts.append(deg)
    parts.append(np.repeat(rnd(20, 0.3), 10, axis=0))                          # coincident copies, interleaved below
    chain = np.zeros((63, 3, 3), np.float32)
    for k in range(63):
        sz = np.float32(2.0 ** -(k % 21))
        base = np.array([-0.9 + 0.028 * k, -0.5, -2.0 - 0.01 * k], np.float32)
        chain[k] = base + sz * np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.25]], np.float32)
    parts.append(chain)
    parts.append((rnd(50, 1.0) * np.float32(1.0e4)).astype(np.float32))        # far away and huge
    flat = rnd(120, 0.25)
    flat[:40, :, 0] = flat[:40, :1, 0]; flat[40:80, :, 1] = flat[40:80, :1, 1]; flat[80:, :, 2] = flat[80:, :1, 2]
    parts.append(flat)
    verts = np.concatenate(parts).astype(np.float32)
    order = rng.permutation(len(verts))                                        # shuffle primitive ids
    verts = verts[order]
    n = len(verts)
    faces = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
    mats = [diffuse(0.73, 0.73, 0.73), emitter(*LE)]
    meshes = [MeshDesc(vertices=verts.reshape(-1, 3), faces=faces, sets=[(0, np.arange(n, dtype=np.uint32))]),
              _quad((-2.0, 1.5, -0.5), (-2.0, 1.5, -4.5), (2.0, 1.5, -4.5), (2.0, 1.5, -0.5), 1)]
    return SceneDesc(meshes, mats, CameraDesc(width, height, 1.9), name="stress")


def smooth_blobs(width=96, height=64, per_vertex=True):
    """Smooth-shaded geometry for the interpolated-normal path of mesh_t::shading_parameters
    (src/mesh.cpp:187-199): two subdivided octahedra ("blobs") with per-vertex normals — or per-face-corner
    normals when `per_vertex` is False (set_normals_per_vertex_per_face, src/mesh.hpp:64) — inside a
    flat-shaded floor/back wall, lit by TWO emissive quads (light pick by index, src/sampling.cpp:165) of
    which one is a 4-triangle strip (uniform triangle pick, src/light.cpp:55)."""
    def octa(centre, radius, levels):
        v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
        f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
        v = [np.array(x, np.float64) for x in v]
        for _ in range(levels):
            nf, cache = [], {}
            def mid(a, b):
                k = (min(a, b), max(a, b))
                if k not in cache:
                    m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
                return cache[k]
            for a, b, c in f:
                ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
                nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
            f = nf
        n = np.array(v, np.float32)
        return (n * np.float32(radius) + np.array(centre, np.float32)).astype(np.float32), n, np.array(f, np.uint32)
    mats = [diffuse(0.73, 0.73, 0.73), diffuse(0.2, 0.5, 0.7),
            MaterialDesc([LobeDesc(abi.LOBE_DIFFUSE, (0.5, 0.4, 0.3)), LobeDesc(abi.LOBE_MICROFACET, (0.3, 0.3, 0.3), xalpha=0.09, yalpha=0.09)]),
            emitter(*LE), emitter(4.0, 4.0, 6.0)]
    meshes = [
        _quad((-2, -1, -1), (2, -1, -1), (2, -1, -4), (-2, -1, -4), 0),
        _quad((-2, -1, -4), (2, -1, -4), (2, 2, -4), (-2, 2, -4), 0),
    ]
    for centre, radius, mat in (((-0.6, -0.4, -2.6), 0.6, 1), ((0.7, -0.5, -2.2), 0.5, 2)):
        v, n, f = octa(centre, radius, 2)
        if per_vertex:
            m = MeshDesc(vertices=v, faces=f, normals=n, smooth=np.ones(len(f), np.uint8), sets=[(mat, np.arange(len(f), dtype=np.uint32))])
        else:  # one normal per face corner, indexed by 3*face + corner (mesh.cpp:188-192); every other face flat
            smooth = (np.arange(len(f)) % 2 == 0).astype(np.uint8)
            m = MeshDesc(vertices=v, faces=f, normals=n[f.reshape(-1)], smooth=smooth, sets=[(mat, np.arange(len(f), dtype=np.uint32))],
                         flags=abi.MESH_UV_PER_VERTEX)
        meshes.append(m)
    meshes.append(_quad((-0.5, 1.9, -2.0), (-0.5, 1.9, -3.0), (0.5, 1.9, -3.0), (0.5, 1.9, -2.0), 3))
    strip_v = np.array([(-1.9, 0.0, -1.2), (-1.9, 1.0, -1.2), (-1.9, 0.0, -2.0), (-1.9, 1.0, -2.0), (-1.9, 0.0, -2.8), (-1.9, 1.0, -2.8)], np.float32)
    strip_f = np.array([(0, 2, 1), (1, 2, 3), (2, 4, 3), (3, 4, 5)], np.uint32)  # normals +x
    meshes.append(MeshDesc(vertices=strip_v, faces=strip_f, sets=[(4, np.arange(4, dtype=np.uint32))]))
    return SceneDesc(meshes, mats, CameraDesc(width, height, 1.6), name="smooth_blobs")


def closure_zoo():
    """One material per lobe type of src/bsdf.hpp:14-24 plus mixes, mirroring the constant-input
    mappings of the BSDF-node shaders: used by the BSDF known-answer tests and
    the multi-material stand-in scenes."""
    D, ON, RF, RR, MF, SH, TR = (abi.LOBE_DIFFUSE, abi.LOBE_OREN_NAYAR, abi.LOBE_REFLECTION, abi.LOBE_REFRACTION,
                                 abi.LOBE_MICROFACET, abi.LOBE_SHEEN, abi.LOBE_TRANSPARENT)
    return [
        diffuse(0.73, 0.73, 0.73),                                                          # 0 diffuse_bsdf_node, roughness 0
        MaterialDesc([LobeDesc(ON, (0.6, 0.5, 0.4), alpha=0.5)]),                           # 1 oren_nayar(N, roughness)
        MaterialDesc([LobeDesc(RF, (0.9, 0.9, 0.9), eta=0.0)]),                             # 2 glossy sharp -> reflection(N,0)
        MaterialDesc([LobeDesc(RR, (0.95, 0.95, 0.95), eta=1.45)]),                         # 3 refraction sharp
        MaterialDesc([LobeDesc(MF, (0.8, 0.7, 0.3), xalpha=0.09, yalpha=0.09)]),            # 4 glossy r=0.3 -> microfacet(r^2)
        MaterialDesc([LobeDesc(MF, (0.9, 0.9, 0.9), eta=1.33, xalpha=0.2, yalpha=0.2, refract=1)]),  # 5 rough refraction
        MaterialDesc([LobeDesc(SH, (0.5, 0.2, 0.6), r=0.4)]),                               # 6 sheen
        MaterialDesc([LobeDesc(TR, (0.8, 0.9, 0.8))]),                                      # 7 transparent
        MaterialDesc([LobeDesc(D, (0.4, 0.3, 0.2)), LobeDesc(MF, (0.3, 0.3, 0.3), xalpha=0.04, yalpha=0.04)]),  # 8 mix diffuse+glossy
        MaterialDesc([LobeDesc(D, (0.3, 0.3, 0.5)), LobeDesc(SH, (0.3, 0.3, 0.3), r=0.4), LobeDesc(ON, (0.2, 0.2, 0.2), alpha=0.3)]),  # 9 three lobes
        MaterialDesc([LobeDesc(RR, (0.6, 0.6, 0.6), eta=1.5), LobeDesc(RF, (0.3, 0.3, 0.3))]),  # 10 glass-like constant mix
        MaterialDesc([LobeDesc(MF, (0.7, 0.7, 0.7), xalpha=0.25, yalpha=0.0625)]),          # 11 anisotropic glossy
    ]


def glass(ior=1.45, roughness=0.0, Cs_refraction=(1.0, 1.0, 1.0), Cs_reflection=(1.0, 1.0, 1.0)):
    """Blender's glass node as the reference's exporter builds it (plugins/blender/blender/shader.hpp:306-335):
    mix_closure_node(A = refraction_bsdf_node(IoR, roughness), B = glossy_bsdf_node(roughness), fac = fresnel_dielectric_node(IoR)),
    i.e. what closures.bake_material returns for that node group: two closures whose weights depend on the hit."""
    from . import closures as cl
    tree = cl.mix_closure_node(A=cl.refraction_bsdf_node(Cs=Cs_refraction, IoR=ior, roughness=roughness),
                               B=cl.glossy_bsdf_node(Cs=Cs_reflection, roughness=roughness), fac=cl.fresnel_dielectric_node(IoR=ior))
    return cl.flatten(tree)


def glass_blobs(width=96, height=64):
    """smooth_blobs() with one blob of sharp glass (IoR 1.45) and one of frosted glass (IoR 1.33, roughness 0.2)"""
    s = smooth_blobs(width, height)
    s.materials[1] = glass(1.45, 0.0, (0.95, 0.98, 0.95), (1.0, 1.0, 1.0))
    s.materials[2] = glass(1.33, 0.2, (0.9, 0.9, 1.0), (0.9, 0.9, 0.9))
    s.name = "glass_blobs"
    return s


def multi_material_soup(n, seed=1234, width=1280, height=720):
    """Declared stand-in for the BMW configs (no scene data ships with the reference):
    Soup(N) whose triangles cycle through closure_zoo() + 4 more diffuse tints = 16 closure recipes."""
    mats = closure_zoo() + [diffuse(0.7, 0.2, 0.2), diffuse(0.2, 0.7, 0.2), diffuse(0.2, 0.2, 0.7), diffuse(0.5, 0.5, 0.1)]
    s = soup(n, seed, width, height, materials=mats)
    s.name = f"zoo_soup{n}"
    return s


def showroom(n, seed=5, width=1280, height=720, materials=None, closed=False):
    """A non-uniform, mesh-like scene (connected surfaces, three orders of magnitude of triangle sizes): a closed grey room with a
    ceiling light holding 24 lat/long-tessellated spheres of radius 0.04..0.55 whose triangle counts are NOT proportional to their
    area (the smallest sphere gets as many triangles as the largest: "teapot in a stadium"), about n triangles in total.  Used to
    check both tree builders — and PHX_BVH_AUTO's choice between them — on something that is not the uniform soup of SURVEY §8(d).
    With `materials` the spheres cycle through that list (material 0 stays the room's).  The room is open towards the camera (which stands
    in front of it, at the origin); `closed` pulls floor, ceiling and side walls past the camera and adds the wall behind it: no path
    leaves the room, every path ends by roulette, the depth limit or absorption."""
    rng = np.random.default_rng(seed)
    mats = [diffuse(0.73, 0.73, 0.73)] + (list(materials) if materials else [diffuse(0.6, 0.3, 0.2), diffuse(0.2, 0.5, 0.7)])
    nm = len(mats)
    mats.append(emitter(*LE))
    X0, X1, Y0, Y1, Z0, Z1 = -2.0, 2.0, -1.0, 1.6, -4.6, (0.4 if closed else -0.4)
    meshes = [
        _quad((X0, Y0, Z1), (X1, Y0, Z1), (X1, Y0, Z0), (X0, Y0, Z0), 0),   # floor
        _quad((X0, Y0, Z0), (X1, Y0, Z0), (X1, Y1, Z0), (X0, Y1, Z0), 0),   # back
        _quad((X0, Y0, Z1), (X0, Y0, Z0), (X0, Y1, Z0), (X0, Y1, Z1), 0),   # left
        _quad((X1, Y0, Z0), (X1, Y0, Z1), (X1, Y1, Z1), (X1, Y1, Z0), 0),   # right
        _quad((X0, Y1, Z0), (X1, Y1, Z0), (X1, Y1, Z1), (X0, Y1, Z1), 0),   # ceiling
        _quad((-0.8, Y1 - 0.01, -1.7), (-0.8, Y1 - 0.01, -3.3), (0.8, Y1 - 0.01, -3.3), (0.8, Y1 - 0.01, -1.7), nm),
    ]
    if closed:
        meshes.append(_quad((X1, Y0, Z1), (X0, Y0, Z1), (X0, Y1, Z1), (X1, Y1, Z1), 0))   # the wall behind the camera, facing the room
    nspheres = 24
    per = max(8, int(n) // nspheres)
    for k in range(nspheres):
        radius = float(0.04 * (0.55 / 0.04) ** rng.random())
        centre = np.array([rng.uniform(X0 + radius, X1 - radius), rng.uniform(Y0 + radius, Y1 - 0.3 - radius), rng.uniform(Z0 + radius, -1.2 - radius)])
        rings = max(2, int(round(np.sqrt(per / 4.0))))          # 2*rings*segs triangles with segs = 2*rings
        segs = 2 * rings
        th = np.linspace(0.0, np.pi, rings + 1)[:, None]
        ph = np.linspace(0.0, 2.0 * np.pi, segs + 1)[None, :-1]
        p = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], -1)  # (rings+1, segs, 3)
        v = (centre + radius * p.reshape(-1, 3)).astype(np.float32)
        i = np.arange(rings)[:, None]; j = np.arange(segs)[None, :]
        a = (i * segs + j).ravel(); b = (i * segs + (j + 1) % segs).ravel()
        c = ((i + 1) * segs + j).ravel(); d = ((i + 1) * segs + (j + 1) % segs).ravel()
        f = np.concatenate([np.stack([a, b, d], 1), np.stack([a, d, c], 1)]).astype(np.uint32)  # outward-facing; the pole rows hold slivers of zero area
        meshes.append(MeshDesc(vertices=v, faces=f, sets=[(1 + k % (nm - 1), np.arange(len(f), dtype=np.uint32))]))
    return SceneDesc(meshes, mats, CameraDesc(width, height, 1.9), name=f"showroom{n}{'_closed' if closed else ''}")


def showroom_materials(per_hit_glass=True):
    """the 16 closure recipes of the BMW stand-in (closure_zoo() + four diffuse tints) plus Blender's glass node twice — sharp
    (IoR 1.45) and frosted (IoR 1.33, roughness 0.2) — whose closure weights depend on the hit (k_shade_g<PERHIT>): 18 materials.
    per_hit_glass=False (A/B experiments only): the two glass materials as constant refraction + reflection mixes."""
    z = closure_zoo()
    g = [glass(1.45, 0.0, (0.95, 0.98, 0.95), (1.0, 1.0, 1.0)), glass(1.33, 0.2, (0.9, 0.9, 1.0), (0.9, 0.9, 0.9))] if per_hit_glass else [z[10], z[5]]
    return z + [diffuse(0.7, 0.2, 0.2), diffuse(0.2, 0.7, 0.2), diffuse(0.2, 0.2, 0.7), diffuse(0.5, 0.5, 0.1)] + g


def bmw_showroom(n=500_000, width=1920, height=1080, per_hit_glass=True):
    """The mesh-geometry stand-in for BASELINE configs 3 / 5: the CLOSED showroom — connected surfaces, triangle sizes
    over three decades, no path escapes — with showroom_materials() on its 24 spheres.  multi_material_soup() stays the open, uniform one."""
    s = showroom(n, width=width, height=height, materials=showroom_materials(per_hit_glass), closed=True)
    s.name = f"bmw_showroom{n}" + ("" if per_hit_glass else "_constant_glass")
    return s
