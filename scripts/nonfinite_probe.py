#!/usr/bin/env python3
"""The non-finite pixels of a many-sample frame: where they are, whether the CPU oracle has the SAME ones, and what made them.
(bench.py's config-5 record ships film_finite = false — 9 of 8.3 M pixels — with an explanation that did not hold.)

Run ON the GPU box:
    python scripts/nonfinite_probe.py [--scene zoo|bmwroom] [--width 3840 --height 2160 --spp 4096] > gpurun_out/nonfinite_probe.json

1. the whole frame on the device; the pixels with a non-finite component,
2. the 32x32 tiles that hold them, rendered by the oracle (device tie rule): the non-finite MASKS must be equal and every finite pixel
   bit-equal — the samples are products of the restated arithmetic, not device artefacts,
3. for each such pixel the oracle bisects the sample range down to the ONE sample that is non-finite, and renders it with the
   diagnostic hook (oracle.set_debug_nonfinite): which term — f, the pdf, the throughput — went non-finite, with its operands.
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from phosphorus_mk2_amd import scenes, xpu  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (the checker, never the thing measured)


@contextlib.contextmanager
def captured_stderr():
    """the oracle's diagnostic lines go to the C stderr: redirect fd 2 into a file for the duration"""
    sys.stderr.flush()
    saved = os.dup(2)
    tmp = tempfile.TemporaryFile(mode="w+b")
    os.dup2(tmp.fileno(), 2)
    box = {}
    try:
        yield box
    finally:
        os.dup2(saved, 2); os.close(saved)
        tmp.seek(0); box["text"] = tmp.read().decode(errors="replace"); tmp.close()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scene", default="zoo", choices=["zoo", "bmwroom"])
    p.add_argument("--triangles", type=int, default=500000)
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--spp", type=int, default=4096)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--max-tiles", type=int, default=12)
    p.add_argument("--threads", type=int, default=16)
    a = p.parse_args()
    W, H, spp = a.width, a.height, a.spp
    sc = scenes.multi_material_soup(a.triangles, seed=1234, width=W, height=H) if a.scene == "zoo" else scenes.bmw_showroom(a.triangles, width=W, height=H)
    t0 = time.time()
    film, st = xpu.render(sc, spp=spp, pps=1, depth=9, seed=a.seed, native_sink=True)
    t_dev = time.time() - t0
    bad = np.argwhere(~np.isfinite(film[..., :3]).all(-1))  # (y, x)
    out = {"scene": sc.name, "film": [W, H], "spp": spp, "seed": a.seed, "device_frame_s": t_dev, "rays": st["rays_closest"] + st["rays_shadow"],
           "nonfinite_pixels_xy": [[int(x), int(y)] for y, x in bad], "pixels": W * H}
    tiles = sorted({(int(x) // 32 * 32, int(y) // 32 * 32) for y, x in bad})[:a.max_tiles]
    tiles = [(x, y, min(32, W - x), min(32, H - y)) for x, y in tiles]
    out["tiles_checked"] = tiles
    if tiles:
        O = orc.Oracle(sc, spp=spp, pps=1, depth=9)
        orc.set_tie_rule(1)
        try:
            t0 = time.time()
            ref, ost = O.render(rng=orc.RNG_COUNTER, seed=a.seed, threads=a.threads, tiles=tiles)
            out["oracle_tiles_s"] = time.time() - t0
            masks_equal, finite_equal = True, True
            for (x, y, w, h) in tiles:
                d, r = film[y:y + h, x:x + w, :3], ref[y:y + h, x:x + w, :3]
                fd, fr = np.isfinite(d).all(-1), np.isfinite(r).all(-1)
                masks_equal &= bool(np.array_equal(fd, fr))
                both = fd & fr
                finite_equal &= bool(np.array_equal(d[both].view(np.uint32), r[both].view(np.uint32)))
            out["nonfinite_masks_equal"], out["finite_pixels_bit_equal"] = masks_equal, finite_equal
            # which sample, and what made it: bisect [0, spp) on the pixel's own 8x1 strip, then one sample with the diagnostic hook on
            causes = []
            for y, x in bad[:a.max_tiles]:
                x, y = int(x), int(y)
                strip = [(x // 8 * 8, y, min(8, W - x // 8 * 8), 1)]
                lo, hi = 0, spp
                while hi - lo > 1:
                    mid = (lo + hi) // 2
                    f, _ = O.render(rng=orc.RNG_COUNTER, seed=a.seed, threads=1, tiles=strip, sample_begin=lo, sample_end=mid)
                    if not np.isfinite(f[y, x, :3]).all():
                        hi = mid
                    else:
                        lo = mid
                orc.set_debug_nonfinite(1)
                try:
                    with captured_stderr() as cap:
                        f, _ = O.render(rng=orc.RNG_COUNTER, seed=a.seed, threads=1, tiles=strip, sample_begin=lo, sample_end=lo + 1)
                finally:
                    orc.set_debug_nonfinite(0)
                lines = [l for l in cap["text"].splitlines() if l.startswith("orc nonfinite")]
                causes.append({"pixel_xy": [x, y], "sample": lo, "sample_alone_is_nonfinite": bool(not np.isfinite(f[y, x, :3]).all()),
                               "device_value": [float(v) for v in film[y, x, :3]], "oracle_diagnostic": lines[:4]})
            out["causes"] = causes
        finally:
            orc.set_tie_rule(0)
            O.close()
    print(json.dumps(out))
    return 0 if (not tiles or (out["nonfinite_masks_equal"] and out["finite_pixels_bit_equal"])) else 1


if __name__ == "__main__":
    sys.exit(main())
