"""Run ON the GPU box after `make -C phosphorus_mk2_amd/csrc variant NAME=shtime EXTRA=-DPHX_SHADE_TIMING=1`: how k_shade_g's wave cycles split between
the material sort of a window and its eight shading rounds (s_memtime per wave, summed).  Round 4: sort 16 %, shading rounds 84 %."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["PHX_LIB"] = os.path.join(os.getcwd(), "phosphorus_mk2_amd", "libphx_hip_shtime.so")
from phosphorus_mk2_amd import scenes, xpu
sc = scenes.multi_material_soup(500000, width=1920, height=1080)
film, st = xpu.render(sc, spp=256, seed=1, native_sink=True)
tot = st["wave_iters"] + st["node_block_execs"]
print("windows x waves", st["refills"], "sort cycles %.3g shade cycles %.3g -> sort share %.3f" % (st["wave_iters"], st["node_block_execs"], st["wave_iters"] / tot), "shade_kernel_ms", st["shade_kernel_ms"])
