"""The oracle's pinned transcendental functions (oracle/omath.h) against the host libm, the counter
sampler's range/uniformity, and the restated sampling maps against their closed forms."""
import numpy as np

from conftest import bits_equal


def fp(a):
    from phosphorus_mk2_amd import abi
    return a.ctypes.data_as(abi.f32p)


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai); bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


def test_sincos_within_one_ulp_of_libm(orc):
    lib = orc.load()
    x = np.concatenate([np.linspace(0, 2 * np.pi, 200001), np.random.default_rng(1).uniform(-50, 50, 100000)]).astype(np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    lib.orc_sincos(len(x), fp(x), fp(s), fp(c))
    rs = np.sin(x.astype(np.float64)).astype(np.float32); rc = np.cos(x.astype(np.float64)).astype(np.float32)
    assert ulp_diff(s, rs).max() <= 1 and ulp_diff(c, rc).max() <= 1
    assert (s == rs).mean() > 0.999 and (c == rc).mean() > 0.999  # correctly rounded almost everywhere


def test_exp_log_pow_within_one_ulp(orc):
    lib = orc.load()
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(1e-6, 1.0, 50000), rng.uniform(1.0, 80.0, 50000)]).astype(np.float32)
    y = rng.uniform(0.1, 12.0, len(x)).astype(np.float32)
    e = np.zeros_like(x); l = np.zeros_like(x); p = np.zeros_like(x)
    lib.orc_exp_log_pow(len(x), fp(x), fp(y), fp(e), fp(l), fp(p))
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    assert ulp_diff(e, np.exp(x64).astype(np.float32))[np.isfinite(e)].max() <= 1
    assert ulp_diff(l, np.log(x64).astype(np.float32)).max() <= 1
    ref = np.power(x64, y64)
    ok = np.isfinite(ref) & (ref < 3e38) & (ref > 1e-37)
    assert ulp_diff(p[ok], ref[ok].astype(np.float32)).max() <= 1
    # the values sheen.hpp feeds in: pow(sin_theta in [0,1], 1/r) and pow(x, c ~ 0.17..0.2)
    st = rng.uniform(0, 1, 20000).astype(np.float32); ex = np.full_like(st, 2.5)
    lib.orc_exp_log_pow(len(st), fp(st), fp(ex), fp(e[:len(st)].copy()), fp(l[:len(st)].copy()), fp(p[:len(st)]))
    assert ulp_diff(p[:len(st)], np.power(st.astype(np.float64), 2.5).astype(np.float32)).max() <= 1


def test_counter_sampler(orc):
    lib = orc.load()
    out = np.zeros(80, np.float32)
    vals = []
    for pixel in range(2000):
        lib.orc_counter_rng(12345, pixel, 3, 80, fp(out))
        vals.append(out.copy())
    v = np.array(vals)
    assert v.min() >= 0.0 and v.max() < 1.0
    assert abs(v.mean() - 0.5) < 0.005 and abs(v.var() - 1 / 12) < 0.002
    # distinct (pixel, sample, dimension) -> decorrelated streams
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 0.08 and abs(np.corrcoef(v[:-1, 4], v[1:, 4])[0, 1]) < 0.08
    a = np.zeros(8, np.float32); b = np.zeros(8, np.float32)
    lib.orc_counter_rng(1, 5, 0, 8, fp(a)); lib.orc_counter_rng(1, 5, 0, 8, fp(b))
    assert bits_equal(a, b)
    lib.orc_counter_rng(2, 5, 0, 8, fp(b))
    assert not bits_equal(a, b)


def test_cosine_weighted_and_onb(orc):
    lib = orc.load()
    rng = np.random.default_rng(4)
    u = rng.random((20000, 2)).astype(np.float32)
    out = np.zeros((len(u), 3), np.float32); pdf = np.zeros(len(u), np.float32)
    lib.orc_cosine_weighted(len(u), fp(u), fp(out), fp(pdf))
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=2e-6)  # y is up (math/sampling.hpp:23-36)
    assert np.allclose(pdf, out[:, 1] / np.pi, rtol=1e-6)
    assert abs(out[:, 1].mean() - 2 / 3) < 0.01  # E[cos] under a cosine-weighted density
    n = rng.normal(size=(5000, 3)); n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    n[0] = [0.57735026, 0.57735026, 0.57735026]  # n.x == n.y == n.z: the second branch of orthogonal_base_t
    abc = np.zeros((len(n), 9), np.float32)
    lib.orc_onb(len(n), fp(n), fp(abc))
    a, b, c = abc[:, 0:3], abc[:, 3:6], abc[:, 6:9]
    assert bits_equal(b, n)
    for p, q in ((a, b), (a, c), (b, c)):
        assert np.abs((p * q).sum(1)).max() < 1e-5
    assert np.allclose(np.linalg.norm(a, axis=1), 1, atol=1e-6) and np.allclose(np.linalg.norm(c, axis=1), 1, atol=1e-6)


def test_light_sampling_is_on_the_emitter(orc):
    from phosphorus_mk2_amd import scenes
    O = orc.Oracle(scenes.cornell(32, 32), spp=1)
    rng = np.random.default_rng(9)
    pick = rng.random(5000).astype(np.float32); u2 = rng.random((5000, 2)).astype(np.float32)
    p, uv, pdf, mesh, face = O.light_sample(pick, u2)
    assert np.allclose(p[:, 1], 0.99) and (np.abs(p[:, 0]) <= 0.25 + 1e-6).all() and (p[:, 2] <= -2.25 + 1e-6).all() and (p[:, 2] >= -2.75 - 1e-6).all()
    assert np.allclose(pdf, 1.0 / 0.25)  # 1 / (total area * nlights), lamp = 0.5 x 0.5
    assert set(np.unique(mesh & 0xffff)) == {5} and set(np.unique(mesh >> 16)) == {3}
    assert set(np.unique(face)) == {0, 3}  # uniform pick by index between the two triangles (light.cpp:55)
    assert abs((face == 0).mean() - 0.5) < 0.03
