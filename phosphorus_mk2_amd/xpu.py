"""Host-side mirror of the reference's device interface over the C ABI (include/phx_xpu.h).

    reference (C++)                                   here
    -----------------------------------------------   -------------------------------------------
    xpu_t::discover(options)      src/xpu.cpp:7-9     HipDevice.discover(options)
    T::make(options)              src/xpu/cpu.hpp:35  HipDevice.make(options)
    xpu_t::preprocess(scene)      src/xpu.hpp:20      HipDevice.preprocess(scene)
    xpu_t::start(scene, frame)    src/xpu.hpp:26      HipDevice.start(scene, frame)   (non-blocking)
    xpu_t::join()                 src/xpu.hpp:32      HipDevice.join()
    frame_state_t{sampler,tiles,film} state.hpp:18    FrameState(sampler_seed, tiles, film)
    job::tiles_t::make/next       jobs/tiles.hpp      Tiles.make(...) / Tiles.next()
    film_t<>::add_tile            film.hpp:12-15      Film.add_tile(...)

The compute path is libphx_hip.so (hand-written HIP for gfx950).  There is no CPU fallback: if the
library is missing or no MI355X is visible, construction raises.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHX_LIB selects another build of the same library (A/B experiments: scripts/ab.sh builds variants with extra -D flags)
LIB_PATH = os.environ.get("PHX_LIB") or os.path.join(_HERE, "libphx_hip.so")
_lib = None


class DeviceError(RuntimeError):
    pass


def load_library():
    """Load the in-tree HIP extension.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DeviceError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback path")
        _lib = abi.declare(C.CDLL(LIB_PATH))
        _lib.phx_abi_sizeof.argtypes = [C.c_int]
        _lib.phx_abi_sizeof.restype = C.c_uint32
    return _lib


def _check(lib, rc, what):
    if rc != abi.PHX_OK:
        raise DeviceError(f"{what} failed ({rc}): {lib.phx_last_error().decode()}")


class Options:
    """parsed_options_t (src/options.hpp:6-43) plus the device knobs."""

    def __init__(self, samples_per_pixel=16, paths_per_sample=16, path_depth=9, single_threaded=False, host_only=False,
                 render_normals=False, verbose=False, device_ordinal=-1, samples_in_flight=0, tiles_per_batch=0, bvh_builder="auto"):
        self.samples_per_pixel = samples_per_pixel
        self.paths_per_sample = paths_per_sample
        self.path_depth = path_depth
        self.single_threaded = single_threaded
        self.host_only = host_only
        self.render_normals = render_normals
        self.verbose = verbose
        self.device_ordinal = device_ordinal
        self.samples_in_flight = samples_in_flight
        self.tiles_per_batch = tiles_per_batch
        self.bvh_builder = bvh_builder  # "host": binned SAH on the host cores; "device": LBVH built on the GPU; "auto": by scene size

    def pack(self):
        o = abi.Options()
        o.samples_per_pixel, o.paths_per_sample, o.path_depth = self.samples_per_pixel, self.paths_per_sample, self.path_depth
        o.single_threaded, o.host_only = int(self.single_threaded), int(self.host_only)
        o.render_normals, o.verbose = int(self.render_normals), int(self.verbose)
        o.device_ordinal, o.samples_in_flight, o.tiles_per_batch = self.device_ordinal, self.samples_in_flight, self.tiles_per_batch
        o.bvh_builder = {"auto": abi.BVH_AUTO, "host": abi.BVH_HOST_SAH, "device": abi.BVH_DEVICE_LBVH}[self.bvh_builder]
        return o


class Tiles:
    """job::tiles_t: precomputed tile_size x tile_size tiles behind an atomic cursor; `rank`/`world`
    keep only the tiles of one GPU (tile (tx, ty) -> rank (tx + s*ty) % world, see dist.tile_shift) for the multi-GPU shard."""

    def __init__(self, handle, lib, width, height):
        self._h, self._lib, self.width, self.height = handle, lib, width, height

    @staticmethod
    def make(width, height, tile_size=32, rank=0, world=1):
        lib = load_library()
        h = lib.phx_tiles_make(width, height, tile_size, rank, world)
        if not h:
            raise DeviceError(lib.phx_last_error().decode())
        return Tiles(h, lib, width, height)

    def next(self):
        t = abi.Tile()
        if self._lib.phx_tiles_next(self._h, C.byref(t)):
            return (t.x, t.y, t.w, t.h)
        return None

    def reset(self):
        self._lib.phx_tiles_reset(self._h)

    def __len__(self):
        return self._lib.phx_tiles_count(self._h)

    def __del__(self):
        try:
            if self._h:
                self._lib.phx_tiles_free(self._h)
                self._h = None
        except Exception:
            pass


class CallbackTiles:
    """A tile queue implemented in Python (any object with next() -> (x,y,w,h) | None): exercises the
    phx_next_tile_fn callback path exactly like a foreign host's own job::tiles_t would."""

    def __init__(self, tiles):
        self._tiles = list(tiles)
        self._i = 0
        self._lock = threading.Lock()

    def next(self):
        with self._lock:
            if self._i < len(self._tiles):
                self._i += 1
                return self._tiles[self._i - 1]
        return None


class Film:
    """An in-memory film sink: film_t<>::add_tile copies a tile's render buffer into a full frame
    (what film::file_t / the Blender sink do, src/film/file.cpp:27-41)."""

    def __init__(self, width, height, primary_components=4, normals=False):
        self.width, self.height = width, height
        self.primary_components, self.normals = primary_components, normals
        self.xstride = primary_components + (3 if normals else 0)
        self.data = np.zeros((height, width, self.xstride), np.float32)
        self.tiles = 0
        self._lock = threading.Lock()

    def add_tile(self, x, y, w, h, buffer, xstride, ystride):
        src = np.ctypeslib.as_array(buffer, shape=(h * ystride,)).reshape(h, w, xstride)
        with self._lock:
            self.data[y:y + h, x:x + w, :] = src
            self.tiles += 1

    @property
    def primary(self):
        return self.data[..., :3]


class BottomUpFilm(Film):
    """The Blender-style sink (plugins/blender/sink.cpp:34-69): the host's (0,0) is the BOTTOM-left pixel, so a tile lands at
    row `height - h - y` with its rows reversed, and the alpha of channel "primary" is set to one (sink.cpp:61-63).
    Needs the add_tile callback path (FrameState.native_sink = False): the flip happens per tile, as in the reference."""

    def add_tile(self, x, y, w, h, buffer, xstride, ystride):
        src = np.ctypeslib.as_array(buffer, shape=(h * ystride,)).reshape(h, w, xstride)
        inv_y = self.height - h - y
        with self._lock:
            self.data[inv_y:inv_y + h, x:x + w, :] = src[::-1]
            if self.primary_components == 4:
                self.data[inv_y:inv_y + h, x:x + w, 3] = 1.0
            self.tiles += 1


class FrameState:
    """frame_state_t (src/state.hpp:18-31).  `sampler_seed` replaces the shared sampler_t."""

    def __init__(self, sampler_seed, tiles, film, device_film_ptr=None, native_sink=False):
        self.sampler_seed, self.tiles, self.film, self.device_film_ptr = sampler_seed, tiles, film, device_film_ptr
        self.native_sink = native_sink  # fill film.data from C (phx_frame.host_film) instead of one Python add_tile call per tile


class HipDevice:
    """The gfx950 device behind the xpu_t interface."""

    def __init__(self, handle, lib, options):
        self._h, self._lib, self.options = handle, lib, options
        self._keep = None
        self._scene = None

    @staticmethod
    def discover(options):
        """xpu_t::discover (src/xpu.cpp:7-9): one device object per usable gfx950 GPU — all of them are meant to drain the frame's
        one tile queue into its one film (src/core.cpp:103-115, see render_on).  An explicit `options.device_ordinal >= 0` restricts
        the list to that GPU (what one rank of a one-process-per-GPU job wants)."""
        lib = load_library()
        o = options.pack()
        n = C.c_int(0)
        rc = lib.phx_discover(C.byref(o), C.byref(n))
        if options.host_only:
            return []
        _check(lib, rc, "phx_discover")
        if options.device_ordinal >= 0 or n.value == 1:
            return [HipDevice.make(options)]
        devs = []
        for i in range(n.value):
            oi = Options(**{**options.__dict__, "device_ordinal": i})
            devs.append(HipDevice.make(oi))
        return devs

    @staticmethod
    def make(options):
        lib = load_library()
        o = options.pack()
        h = lib.phx_dev_make(C.byref(o))
        if not h:
            raise DeviceError("phx_dev_make: " + lib.phx_last_error().decode())
        return HipDevice(h, lib, options)

    def preprocess(self, scene_desc):
        s, keep = scene_desc.pack()
        _check(self._lib, self._lib.phx_dev_preprocess(self._h, C.byref(s)), "phx_dev_preprocess")
        self._scene = scene_desc

    def start(self, scene_desc, frame):
        assert scene_desc is self._scene, "start() must get the scene that was preprocessed"
        f = abi.Frame()
        tiles = frame.tiles
        if isinstance(tiles, Tiles):  # native queue: no Python in the loop
            f.tiles_user = tiles._h
            f.next_tile = C.cast(self._lib.phx_tiles_next, C.c_void_p)
            cb_next = None
        else:
            def _next(user, out):
                t = tiles.next()
                if t is None:
                    return 0
                out[0].x, out[0].y, out[0].w, out[0].h = t
                return 1
            cb_next = abi.NEXT_TILE_FN(_next)
            f.next_tile = C.cast(cb_next, C.c_void_p)
        cb_add = None
        if frame.film is not None:
            film = frame.film

            def _add(user, x, y, w, h, buf, xs, ys):
                film.add_tile(x, y, w, h, buf, xs, ys)
            if frame.native_sink:
                f.host_film = film.data.ctypes.data
            else:
                cb_add = abi.ADD_TILE_FN(_add)
                f.add_tile = C.cast(cb_add, C.c_void_p)
            f.primary_components = film.primary_components
            f.normals_channel = 1 if film.normals else 0
        else:
            f.primary_components = 4
        if frame.device_film_ptr:
            f.device_film = frame.device_film_ptr
        f.sampler_seed = frame.sampler_seed
        self._keep = (f, cb_next, cb_add, frame)
        _check(self._lib, self._lib.phx_dev_start(self._h, C.byref(f)), "phx_dev_start")

    def join(self):
        rc = self._lib.phx_dev_join(self._h)
        self._keep = None
        _check(self._lib, rc, "phx_dev_join")

    def stats(self):
        st = abi.Stats()
        _check(self._lib, self._lib.phx_dev_get_stats(self._h, C.byref(st)), "phx_dev_get_stats")
        out = {}
        for k, _ in st._fields_:
            if k == "reserved":
                continue
            v = getattr(st, k)
            out[k] = list(v) if hasattr(v, "__len__") else v
        return out

    # ---- stage-level hooks (parity tests) -------------------------------------------------------
    def trace(self, o, d, tmax, shadow=False):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(tmax)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32); hit = np.zeros(n, np.uint8)
        fp = lambda a: a.ctypes.data_as(abi.f32p)
        _check(self._lib, self._lib.phx_dev_trace(self._h, n, fp(o), fp(d), fp(tmax), 1 if shadow else 0, fp(t), fp(u), fp(v),
                                                  prim.ctypes.data_as(abi.u32p), hit.ctypes.data_as(abi.u8p)), "phx_dev_trace")
        return {"t": t, "u": u, "v": v, "prim": prim, "hit": hit.astype(bool)}

    def bsdf_f(self, material, n, wi, wo):
        n = np.ascontiguousarray(n, np.float32); wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32)
        out = np.zeros_like(wi)
        fp = lambda a: a.ctypes.data_as(abi.f32p)
        _check(self._lib, self._lib.phx_dev_bsdf_f(self._h, material, len(wi), fp(n), fp(wi), fp(wo), fp(out)), "phx_dev_bsdf_f")
        return out

    def bsdf_sample(self, material, n, wi, u2):
        n = np.ascontiguousarray(n, np.float32); wi = np.ascontiguousarray(wi, np.float32); u2 = np.ascontiguousarray(u2, np.float32)
        k = len(wi)
        wo = np.zeros((k, 3), np.float32); f = np.zeros((k, 3), np.float32); pdf = np.zeros(k, np.float32); fl = np.zeros(k, np.uint32)
        fp = lambda a: a.ctypes.data_as(abi.f32p)
        _check(self._lib, self._lib.phx_dev_bsdf_sample(self._h, material, k, fp(n), fp(wi), fp(u2), fp(wo), fp(f), fp(pdf),
                                                        fl.ctypes.data_as(abi.u32p)), "phx_dev_bsdf_sample")
        return wo, f, pdf, fl

    def bvh_pool(self):
        """the acceleration structure as the kernels read it: (uint32 array [elements, 16], grid lo[3], grid cell[3]) — phx_dev_copy_bvh"""
        n = C.c_uint64(0); grid = np.zeros(6, np.float32)
        _check(self._lib, self._lib.phx_dev_copy_bvh(self._h, None, 0, C.byref(n), grid.ctypes.data_as(abi.f32p)), "phx_dev_copy_bvh")
        pool = np.zeros(n.value // 4, np.uint32)
        _check(self._lib, self._lib.phx_dev_copy_bvh(self._h, pool.ctypes.data_as(C.c_void_p), n.value, C.byref(n), grid.ctypes.data_as(abi.f32p)), "phx_dev_copy_bvh")
        return pool.reshape(-1, 16), grid[:3].copy(), grid[3:].copy()

    def close(self):
        """~xpu_t: joins a frame that is still running (its callbacks may fire until then: they are kept alive up to here)"""
        if self._h:
            self._lib.phx_dev_destroy(self._h)
            self._h = None
        self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_on(devices, scene_desc, seed=1, normals=False, tile_size=32, native_sink=True, rank=0, world=1):
    """The reference's frame on SEVERAL devices in one process (src/core.cpp:103-115): every device is started on the SAME tile
    queue and the SAME film and joined in turn; no collective.  `devices` must have been preprocessed with `scene_desc`.
    Returns (film array, [stats per device])."""
    W, H = scene_desc.camera.width, scene_desc.camera.height
    tiles = Tiles.make(W, H, tile_size, rank, world)
    film = Film(W, H, 4, normals)
    for d in devices:
        d.start(scene_desc, FrameState(seed, tiles, film, native_sink=native_sink))
    for d in devices:
        d.join()
    return film.data, [d.stats() for d in devices]


def render(scene_desc, spp=16, pps=1, depth=9, seed=1, normals=False, tile_size=32, rank=0, world=1, callback_tiles=False,
           samples_in_flight=0, tiles_per_batch=0, native_sink=False, bvh_builder="auto"):
    """Convenience: the call sequence of session_t::render (plugins/blender/session.cpp:73-94):
    make -> preprocess -> tiles_t::make -> start -> join on ONE device.  Returns (film array HxWxC, stats)."""
    opts = Options(samples_per_pixel=spp, paths_per_sample=pps, path_depth=depth, samples_in_flight=samples_in_flight,
                   tiles_per_batch=tiles_per_batch, bvh_builder=bvh_builder)
    # ONE device, on the caller's current GPU (device_ordinal -1): discover() is for hosts that drive every GPU (render_on) and
    # would build a context, a stream and the kernel attributes on each of them only to use the first
    dev = HipDevice.make(opts)
    try:
        dev.preprocess(scene_desc)
        W, H = scene_desc.camera.width, scene_desc.camera.height
        tiles = Tiles.make(W, H, tile_size, rank, world)
        if callback_tiles:
            lst = []
            while True:
                t = tiles.next()
                if t is None:
                    break
                lst.append(t)
            tiles = CallbackTiles(lst)
        film = Film(W, H, 4, normals)
        dev.start(scene_desc, FrameState(seed, tiles, film, native_sink=native_sink))
        dev.join()
        return film.data, dev.stats()
    finally:
        dev.close()
