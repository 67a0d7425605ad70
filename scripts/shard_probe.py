"""Strong-scaling rehearsal on ONE GPU: render rank 0's share of the BASELINE frame for world sizes 1, 2, 4, 8 and print where
the frame time goes (kernel time by HIP events vs host wall time of start..join)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from phosphorus_mk2_amd import scenes, xpu

xpu.load_library()
W, H, SPP = 1280, 720, 256
scene = scenes.soup(100000, seed=1234, width=W, height=H)
dev = xpu.HipDevice.make(xpu.Options(samples_per_pixel=SPP, paths_per_sample=1, path_depth=9))
dev.preprocess(scene)
film = xpu.Film(W, H, 4, False)
WORLDS = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 2, 4, 8]
for world in WORLDS:
    best = None
    for rep in range(3):
        tiles = xpu.Tiles.make(W, H, 32, 0, world)
        film.data[:] = 0
        fs = xpu.FrameState(1, tiles, film, native_sink=True)
        t0 = time.perf_counter()
        dev.start(scene, fs); dev.join()
        wall = (time.perf_counter() - t0) * 1e3
        st = dev.stats()
        if best is None or wall < best[0]:
            best = (wall, st)
    wall, st = best
    rays = st["rays_closest"] + st["rays_shadow"]
    print(f"world {world}: wall {wall:7.2f} ms  frame {st['frame_ms']:7.2f}  trace {st['trace_ms']:7.2f}  shade+gen+film {st['shade_ms']:6.2f}  "
          f"other {st['frame_ms'] - st['trace_ms'] - st['shade_ms']:5.2f}  launches {st['trace_launches']}  Mrays/s x world {rays / wall / 1e3 * world:8.1f}", flush=True)
dev.close()
