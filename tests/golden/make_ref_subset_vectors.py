"""Generates tests/golden/ref_subset_vectors.npz from oracle/_ref/libphx_ref_subset.so — object code
compiled from the reference's own dependency-free headers (src/math/fresnel.hpp, src/math/trigonometry.hpp,
src/math/simd/float8.hpp, src/math/simd/int8.hpp, src/utils/compiler.hpp, src/options.hpp, src/math/config.hpp) where they lie under /root/reference.  These are the only
vectors that pin the oracle to outputs of the reference itself; run here (the reference never travels):

    make -C oracle ref && python tests/golden/make_ref_subset_vectors.py
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libphx_ref_subset.so"))
f32p = C.POINTER(C.c_float)
fp = lambda a: a.ctypes.data_as(f32p)
rng = np.random.default_rng(20261003)

n = 2048
cosi = rng.uniform(-1, 1, n).astype(np.float32)
eta = rng.choice(np.array([0.0, 0.5, 1.0, 1.1, 1.33, 1.5, 2.4, 0.75], np.float32), n)
cosi[:6] = [0.0, 1.0, -1.0, 1e-8, -1e-8, 0.5]
fres = np.zeros(n, np.float32)
lib.ref_fresnel_dielectric(n, fp(cosi), fp(eta), fp(fres))
deg = rng.uniform(-720, 720, 256).astype(np.float32)
rad = np.zeros(256, np.float32)
lib.ref_radians(256, fp(deg), fp(rad))

# simd::select(m, l, r) and compares on one 8-wide vector each
l = rng.normal(size=(64, 8)).astype(np.float32); r = rng.normal(size=(64, 8)).astype(np.float32)
l[0, :4] = [0.0, -0.0, np.nan, 1.0]; r[0, :4] = [-0.0, 0.0, 1.0, np.nan]
mask = (rng.integers(0, 2, (64, 8)).astype(np.uint32) * np.uint32(0xffffffff)).view(np.float32)
sel = np.zeros((64, 8), np.float32); cmp = np.zeros((4, 64, 8), np.float32); mm = np.zeros((2, 64, 8), np.float32)
for i in range(64):
    lib.ref_select8(fp(mask[i]), fp(l[i]), fp(r[i]), fp(sel[i]))
    for op in range(4):
        lib.ref_cmp8(op, fp(l[i]), fp(r[i]), fp(cmp[op, i]))
    for k in range(2):
        lib.ref_minmax8(k, fp(l[i]), fp(r[i]), fp(mm[k, i]))
lib.ref_bscf.restype = C.c_uint64
lib.ref_bscf.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
bs_in = rng.integers(1, 2 ** 62, 64).astype(np.uint64); bs_idx = np.zeros(64, np.uint64); bs_rest = np.zeros(64, np.uint64)
for i in range(64):
    rest = C.c_uint64()
    bs_idx[i] = lib.ref_bscf(int(bs_in[i]), C.byref(rest)); bs_rest[i] = rest.value
# simd::int32_t<8> (src/math/simd/int8.hpp): integer ops done with float instructions on the bits (SURVEY A-20)
i32p = C.POINTER(C.c_int32)
ip = lambda a: a.ctypes.data_as(i32p)
il = rng.integers(0, 16, (32, 8)).astype(np.int32); ir = rng.integers(0, 16, (32, 8)).astype(np.int32)   # the flags domain
il[8:16] = rng.integers(-2 ** 31, 2 ** 31, (8, 8)); ir[8:16] = rng.integers(-2 ** 31, 2 ** 31, (8, 8))       # any bit pattern
il[16:24] = rng.integers(0, 3000000, (8, 8)); ir[16:24] = rng.integers(0, 3000000, (8, 8))                # face / mesh ids
il[24, :6] = [0, -2 ** 31, 0x7fc00000, 0x7fc00000, 1, 0x00800000]; ir[24, :6] = [-2 ** 31, 0, 0x7fc00000, 0, 1, 0x00800000]
il[25:] = il[25:] | (rng.integers(0, 2, (7, 8)).astype(np.int32) << 16); ir[25:] = il[25:]               # equal operands
iop = np.zeros((8, 32, 8), np.int32)
for op in range(8):
    for i in range(32):
        lib.ref_int8_op(op, ip(il[i]), ip(ir[i]), ip(iop[op, i]))
flag_words = np.arange(16, dtype=np.int32).reshape(2, 8)
flag_out = np.zeros((4, 2, 8), np.int32)
for k, bit in enumerate((1, 2, 4, 8)):   # HIT, MASKED, SHADOW, SPECULAR (src/state.hpp:33-36)
    for i in range(2):
        lib.ref_int8_flag_test(ip(flag_words[i]), bit, ip(flag_out[k, i]))
cvt_in = rng.uniform(-1000, 1000, (16, 8)).astype(np.float32); cvt_in[0] = [0.5, 1.5, 2.5, -0.5, -1.5, 0.49999997, 1e6, -0.0]
cvt_out = np.zeros((16, 8), np.int32)
isel = np.zeros((32, 8), np.int32)
for i in range(16):
    lib.ref_int8_from_float(fp(cvt_in[i]), ip(cvt_out[i]))
for i in range(32):
    lib.ref_int8_select(fp(mask[i]), ip(il[i]), ip(ir[i]), ip(isel[i]))
# parsed_options_t defaults (src/options.hpp:6-43) and config::STREAM_SIZE (src/math/config.hpp:6)
opt = np.zeros(7, np.uint32); name = C.create_string_buffer(64)
lib.ref_options_defaults.restype = C.c_uint32
nlen = lib.ref_options_defaults(opt.ctypes.data_as(C.POINTER(C.c_uint32)), name, 64)
lib.ref_stream_size.restype = C.c_uint32
stream_size = np.uint32(lib.ref_stream_size())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_subset_vectors.npz"), cosi=cosi, eta=eta, fresnel=fres, deg=deg, rad=rad,
                    l=l, r=r, mask=mask, select=sel, cmp=cmp.view(np.uint32), minmax=mm, bscf_in=bs_in, bscf_idx=bs_idx, bscf_rest=bs_rest,
                    int_l=il, int_r=ir, int_op=iop, flag_words=flag_words, flag_test=flag_out, cvt_in=cvt_in, cvt_out=cvt_out, int_select=isel,
                    options_defaults=opt, options_output=np.frombuffer(name.value[:nlen], np.uint8), stream_size=stream_size)
print("wrote ref_subset_vectors.npz")
