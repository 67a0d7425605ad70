"""ctypes mirror of include/phx_xpu.h (the C ABI of the gfx950 device).

Field order and types must match the header exactly; tests/test_abi.py checks struct sizes against
`phx_abi_sizeof` exported by the library and that every declared symbol is exported.
"""
import ctypes as C

PHX_OK = 0
PHX_HIT, PHX_MASKED, PHX_SHADOW, PHX_SPECULAR = 1, 2, 4, 8
LOBE_EMISSIVE, LOBE_DIFFUSE, LOBE_OREN_NAYAR, LOBE_REFLECTION = 0, 1, 2, 4
LOBE_REFRACTION, LOBE_MICROFACET, LOBE_SHEEN, LOBE_BACKGROUND, LOBE_TRANSPARENT = 8, 16, 32, 64, 128
BSDF_DIFFUSE, BSDF_GLOSSY, BSDF_SPECULAR, BSDF_REFLECT, BSDF_TRANSMIT = 1, 2, 4, 8, 16
MAX_LOBES = 8
FAC_NONE, FAC_MIX_B, FAC_MIX_A = 0, 1, 2
BVH_AUTO, BVH_DEVICE_LBVH, BVH_HOST_SAH = 0, 1, 2
MESH_UV_PER_VERTEX, MESH_NORMALS_PER_VERTEX = 1, 2

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class Options(C.Structure):
    _fields_ = [
        ("samples_per_pixel", C.c_uint32), ("paths_per_sample", C.c_uint32), ("path_depth", C.c_uint32),
        ("single_threaded", C.c_uint32), ("host_only", C.c_uint32), ("render_normals", C.c_uint32),
        ("verbose", C.c_uint32), ("device_ordinal", C.c_int32), ("samples_in_flight", C.c_uint32),
        ("tiles_per_batch", C.c_uint32), ("bvh_builder", C.c_uint32), ("reserved", C.c_uint32 * 5),
    ]


class Lobe(C.Structure):
    _fields_ = [
        ("type", C.c_uint32), ("weight", C.c_float * 3), ("alpha", C.c_float), ("eta", C.c_float),
        ("xalpha", C.c_float), ("yalpha", C.c_float), ("refract", C.c_uint32), ("r", C.c_float),
        ("fac_mode", C.c_uint32), ("fac_ior", C.c_float), ("pre_weight", C.c_float * 3), ("pad", C.c_uint32),
    ]


class Material(C.Structure):
    _fields_ = [
        ("num_lobes", C.c_uint32), ("is_emitter", C.c_uint32), ("emission", C.c_float * 3),
        ("pad", C.c_uint32 * 3), ("lobes", Lobe * MAX_LOBES),
    ]


class FaceSet(C.Structure):
    _fields_ = [("material", C.c_uint32), ("num_faces", C.c_uint32), ("faces", u32p)]


class Mesh(C.Structure):
    _fields_ = [
        ("vertices", f32p), ("num_vertices", C.c_uint32), ("normals", f32p), ("num_normals", C.c_uint32),
        ("faces", u32p), ("num_faces", C.c_uint32), ("smooth", u8p), ("flags", C.c_uint32),
        ("num_sets", C.c_uint32), ("sets", C.POINTER(FaceSet)),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("to_world", C.c_float * 16), ("fov", C.c_float), ("focal_distance", C.c_float),
        ("aperture_radius", C.c_float), ("film_width", C.c_uint32), ("film_height", C.c_uint32),
    ]


class Scene(C.Structure):
    _fields_ = [
        ("num_meshes", C.c_uint32), ("meshes", C.POINTER(Mesh)), ("num_materials", C.c_uint32),
        ("materials", C.POINTER(Material)), ("environment_material", C.c_int32), ("camera", Camera),
    ]


class Tile(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]


NEXT_TILE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Tile))
ADD_TILE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, C.c_uint32, C.c_uint32)


class Frame(C.Structure):
    _fields_ = [
        ("tiles_user", C.c_void_p), ("next_tile", C.c_void_p), ("film_user", C.c_void_p), ("add_tile", C.c_void_p),
        ("sampler_seed", C.c_uint64), ("primary_components", C.c_uint32), ("normals_channel", C.c_uint32),
        ("device_film", C.c_void_p), ("host_film", C.c_void_p), ("reserved", C.c_uint32 * 2),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("camera_samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("rays_masked", C.c_uint64), ("tiles", C.c_uint64), ("trace_launches", C.c_uint64),
        ("trace_ms", C.c_double), ("closest_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
        ("frame_ms", C.c_double), ("bvh_nodes", C.c_uint64), ("bvh_bytes", C.c_uint64), ("triangles", C.c_uint64),
        ("preprocess_ms", C.c_double), ("bvh_build_ms", C.c_double),
        ("trace_block", C.c_uint64), ("trace_ntop", C.c_uint64), ("trace_levels", C.c_uint64), ("trace_waves_per_cu", C.c_uint64),
        ("bvh_depth", C.c_uint64), ("paths_in_flight", C.c_uint64),
        ("node_visits_lds", C.c_uint64 * 2), ("node_visits_mem", C.c_uint64 * 2), ("tri_tests", C.c_uint64 * 2),
        ("instrumented", C.c_uint64), ("wave_iters", C.c_uint64), ("node_block_execs", C.c_uint64), ("tri_block_execs", C.c_uint64),
        ("refills", C.c_uint64), ("idle_lane_iters", C.c_uint64), ("tri_pending_lane_iters", C.c_uint64), ("stack_pushes", C.c_uint64 * 8), ("bvh_cost_model", C.c_double), ("trace_lds_levels", C.c_uint64), ("bvh_built_on_device", C.c_uint64),
        ("shade_kernel_ms", C.c_double), ("shade_launches", C.c_uint64), ("shade_general", C.c_uint64), ("primary_ms", C.c_double), ("primary_launches", C.c_uint64),
        ("primary_packets", C.c_uint64), ("primary_fallbacks", C.c_uint64), ("primary_node_tests", C.c_uint64), ("primary_tri_tests", C.c_uint64), ("primary_tri_lanes_hit", C.c_uint64),
        ("device_bytes", C.c_uint64), ("tri_pairs_pending", C.c_uint64), ("tri_pairs_hist", C.c_uint64 * 8), ("trace_stack_packed", C.c_uint64),
    ]


# every entry point include/phx_xpu.h declares
EXPORTS = [
    "phx_discover", "phx_dev_make", "phx_dev_preprocess", "phx_dev_start", "phx_dev_join", "phx_dev_destroy",
    "phx_last_error", "phx_dev_get_stats", "phx_tiles_make", "phx_tiles_next", "phx_tiles_count", "phx_tiles_reset",
    "phx_tiles_free", "phx_dev_trace", "phx_dev_bsdf_f", "phx_dev_bsdf_sample", "phx_dev_copy_bvh",
]


def declare(lib):
    """Attach argtypes/restypes of the C ABI to a loaded libphx_hip.so."""
    vp = C.c_void_p
    lib.phx_discover.argtypes = [C.POINTER(Options), C.POINTER(C.c_int)]; lib.phx_discover.restype = C.c_int
    lib.phx_dev_make.argtypes = [C.POINTER(Options)]; lib.phx_dev_make.restype = vp
    lib.phx_dev_preprocess.argtypes = [vp, C.POINTER(Scene)]; lib.phx_dev_preprocess.restype = C.c_int
    lib.phx_dev_start.argtypes = [vp, C.POINTER(Frame)]; lib.phx_dev_start.restype = C.c_int
    lib.phx_dev_join.argtypes = [vp]; lib.phx_dev_join.restype = C.c_int
    lib.phx_dev_destroy.argtypes = [vp]; lib.phx_dev_destroy.restype = None
    lib.phx_last_error.argtypes = []; lib.phx_last_error.restype = C.c_char_p
    lib.phx_dev_get_stats.argtypes = [vp, C.POINTER(Stats)]; lib.phx_dev_get_stats.restype = C.c_int
    lib.phx_tiles_make.argtypes = [C.c_uint32] * 5; lib.phx_tiles_make.restype = vp
    lib.phx_tiles_next.argtypes = [vp, C.POINTER(Tile)]; lib.phx_tiles_next.restype = C.c_int
    lib.phx_tiles_count.argtypes = [vp]; lib.phx_tiles_count.restype = C.c_uint32
    lib.phx_tiles_reset.argtypes = [vp]; lib.phx_tiles_reset.restype = None
    lib.phx_tiles_free.argtypes = [vp]; lib.phx_tiles_free.restype = None
    lib.phx_dev_trace.argtypes = [vp, C.c_uint32, f32p, f32p, f32p, C.c_int, f32p, f32p, f32p, u32p, u8p]
    lib.phx_dev_trace.restype = C.c_int
    lib.phx_dev_bsdf_f.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, f32p, f32p, f32p]; lib.phx_dev_bsdf_f.restype = C.c_int
    lib.phx_dev_bsdf_sample.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, f32p, f32p, f32p, f32p, f32p, u32p]
    lib.phx_dev_bsdf_sample.restype = C.c_int
    lib.phx_dev_copy_bvh.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), f32p]; lib.phx_dev_copy_bvh.restype = C.c_int
    return lib
