"""Stability soak on one GPU: many frames on one device, alternating scenes and tile sets; every frame must reproduce the
first frame of its kind bit for bit and device memory must not grow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from phosphorus_mk2_amd import scenes, xpu

lib = xpu.load_library()
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scs = [scenes.soup(20000, width=320, height=192), scenes.cornell(256, 256), scenes.multi_material_soup(5000, width=192, height=128)]
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=8, paths_per_sample=1))
ref, t0, f0 = {}, time.time(), None
for k in range(frames):
    s = scs[k % 3]
    if k % 7 == 0 or k < 3:
        dev.preprocess(s); cur = s
    else:
        s = cur
    W, H = s.camera.width, s.camera.height
    world = 1 + (k % 2)
    film = xpu.Film(W, H, 4)
    dev.start(s, xpu.FrameState(5, xpu.Tiles.make(W, H, 32, 0, world), film, native_sink=(k % 3 == 0)))
    dev.join()
    key = (id(s), world)
    if key not in ref: ref[key] = film.data.copy()
    assert np.array_equal(ref[key].view(np.uint32), film.data.view(np.uint32)), f"frame {k} differs"
    if k == 30: f0 = free_bytes()
    if k % 50 == 0: print(f"frame {k}: free {free_bytes() / 2**30:.2f} GiB, {time.time() - t0:.1f} s", flush=True)
f1 = free_bytes()
print(f"{frames} frames ok; free memory after frame 30: {f0 / 2**30:.3f} GiB, at the end: {f1 / 2**30:.3f} GiB")
assert f0 - f1 < 64 << 20, "device memory grew"
dev.close()
