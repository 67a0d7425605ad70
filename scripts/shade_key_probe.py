"""Run ON the GPU box: what the material gather of k_shade_g's sort phase costs — a one-material scene rendered with the product library and with
a probe build whose sort key is a constant (`make variant NAME=keyprobe EXTRA=-DPHX_SHADE_KEY_PROBE=1`, PHX_LIB=...).  Round 4: 22.87 vs 22.45 ms:
the gather costs 1.8 % of the kernel (it warms the cache for the shading rounds' own read of the same record)."""
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
from phosphorus_mk2_amd import scenes, xpu
mats = [scenes.closure_zoo()[4]]  # one non-Lambert recipe for every triangle: k_shade_g, the sort key is the same for every hit
sc = scenes.soup(500000, width=1920, height=1080, materials=mats)
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=256, paths_per_sample=1, path_depth=9)); dev.preprocess(sc)
film = xpu.Film(1920, 1080, 4); tiles = xpu.Tiles.make(1920, 1080, 32)
for i in range(3):
    tiles.reset(); dev.start(sc, xpu.FrameState(1, tiles, film, native_sink=True)); dev.join()
st = dev.stats()
print(os.path.basename(os.environ.get("PHX_LIB", "libphx_hip.so")), "shade %.2f ms  k_trace %.1f  general %d  film %s" % (st["shade_kernel_ms"], st["closest_ms"], st["shade_general"], hashlib.sha1(film.data.tobytes()).hexdigest()[:12]))
