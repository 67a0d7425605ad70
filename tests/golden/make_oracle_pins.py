"""Regression pins written from the oracle itself (NOT reference-derived): a 32x32 Cornell film in counter
mode and the BSDF known-answer table of every closure recipe in scenes.closure_zoo()."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from phosphorus_mk2_amd import scenes  # noqa: E402

film, _ = orc.Oracle(scenes.cornell(32, 32), spp=4).render(rng=orc.RNG_COUNTER, seed=1, threads=2)
np.save(os.path.join(ROOT, "tests", "golden", "oracle_cornell_32x32_spp4.npy"), film[..., :3])

sc = scenes.multi_material_soup(64, width=32, height=32)
O = orc.Oracle(sc, spp=1)
rng = np.random.default_rng(77)
k = 48
unit = lambda n: (lambda v: (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))(rng.normal(size=(n, 3)))
n, wi, wo = unit(k), unit(k), unit(k)
u2 = rng.random((k, 2)).astype(np.float32)
out = {"n": n, "wi": wi, "wo": wo, "u2": u2}
for m in range(12):
    out[f"f_{m}"] = O.bsdf_f(m, n, wi, wo)
    s_wo, s_f, s_pdf, s_fl = O.bsdf_sample(m, n, wi, u2)
    out[f"s_wo_{m}"], out[f"s_f_{m}"], out[f"s_pdf_{m}"], out[f"s_fl_{m}"] = s_wo, s_f, s_pdf, s_fl
# the input tuple of the reference's scratch program src/test.cpp:8-14 (rough refraction, eta 1.1,
# xalpha = yalpha = 0.001 before precompute): wi = (0.564088, 0.197126, 0.801694), uv = (0.319097, 0.997709), n = +y
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_bsdf_kat.npz"), **out)
print("wrote oracle pins")
