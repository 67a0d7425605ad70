"""Closure models of the oracle (oracle/obsdf.h): closed-form known answers per lobe, the invariants the
reference's own debugging aids state (sampled pdf == pdf() within 1e-3, microfacet.hpp:270-274; finite
radiance, cpu.cpp:181-189), and the committed known-answer table."""
import os

import numpy as np
import pytest

from conftest import ROOT, bits_equal
from phosphorus_mk2_amd import abi

PI = np.float32(np.pi)


@pytest.fixture(scope="module")
def zoo(orc):
    from phosphorus_mk2_amd import scenes
    sc = scenes.multi_material_soup(64, width=32, height=32)
    return orc.Oracle(sc, spp=1), sc


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_known_answer_table(zoo):
    O, _ = zoo
    K = np.load(os.path.join(ROOT, "tests", "golden", "oracle_bsdf_kat.npz"))
    for m in range(12):
        assert bits_equal(O.bsdf_f(m, K["n"], K["wi"], K["wo"]), K[f"f_{m}"])
        wo, f, pdf, fl = O.bsdf_sample(m, K["n"], K["wi"], K["u2"])
        assert bits_equal(wo, K[f"s_wo_{m}"]) and bits_equal(f, K[f"s_f_{m}"]) and bits_equal(pdf, K[f"s_pdf_{m}"])
        assert np.array_equal(fl, K[f"s_fl_{m}"])


def test_lambert_closed_form(zoo):
    O, _ = zoo
    rng = np.random.default_rng(1)
    n = unit(rng, 2000); wi = unit(rng, 2000); wo = unit(rng, 2000)
    f = O.bsdf_f(0, n, wi, wo)
    cos_i = (n * wi).sum(1); cos_o = (n * wo).sum(1)
    expect = np.where((cos_i * cos_o > 0)[:, None], (np.float32(0.73) / PI * cos_i)[:, None], 0.0)  # f * weight * (n.wi), bsdf.cpp:122-127
    assert np.allclose(f, expect, rtol=2e-6, atol=1e-7)
    # sampling: cosine-weighted around n, flags REFLECT|DIFFUSE, pdf = cos/pi, f = weight/pi
    u2 = rng.random((2000, 2)).astype(np.float32)
    swo, sf, pdf, fl = O.bsdf_sample(0, n, wi, u2)
    assert (fl == (abi.BSDF_REFLECT | abi.BSDF_DIFFUSE)).all()
    assert np.allclose((swo * n).sum(1), pdf * PI, atol=3e-6)
    assert np.allclose(sf, np.float32(0.73) / PI, rtol=1e-6)
    assert np.allclose(np.linalg.norm(swo, axis=1), 1, atol=3e-6)


def test_specular_lobes(zoo):
    O, _ = zoo
    rng = np.random.default_rng(2)
    n = unit(rng, 1000); wi = unit(rng, 1000); u2 = rng.random((1000, 2)).astype(np.float32)
    wo, f, pdf, fl = O.bsdf_sample(2, n, wi, u2)  # reflection(N, 0)
    assert (fl == (abi.BSDF_REFLECT | abi.BSDF_SPECULAR)).all() and np.allclose(pdf, 1)
    assert np.allclose(wo, -wi + 2 * (n * wi).sum(1, keepdims=True) * n, atol=2e-6)
    assert np.allclose(f, 0.9)
    assert (O.bsdf_f(2, n, wi, unit(rng, 1000)) == 0).all()  # delta lobes evaluate to 0 (bsdf.cpp:92-98)
    wo, f, pdf, fl = O.bsdf_sample(3, n, wi, u2)  # refraction(N, 1.45): Snell, or black on TIR
    ok = pdf > 0
    live = ok & (f.sum(1) > 0)
    cos_i = (n * wi).sum(1); eta = np.where(cos_i > 0, 1 / 1.45, 1.45)
    sin_t = np.sqrt(np.maximum(0, 1 - (n[live] * wo[live]).sum(1) ** 2)); sin_i = np.sqrt(np.maximum(0, 1 - cos_i[live] ** 2))
    assert np.allclose(sin_t, eta[live] * sin_i, atol=1e-5)
    assert ((n[live] * wo[live]).sum(1) * cos_i[live] < 0).all()  # transmitted to the other side
    tir = 1 - eta ** 2 * (1 - cos_i ** 2) < 0
    assert (f[tir] == 0).all()
    wo, f, pdf, fl = O.bsdf_sample(7, n, wi, u2)  # transparent: straight through, TRANSMIT only (not SPECULAR)
    assert bits_equal(wo, -wi) and (fl == abi.BSDF_TRANSMIT).all() and np.allclose(f, (0.8, 0.9, 0.8))


def test_microfacet_sampling_invariants(zoo):
    """Cook-Torrance GGX reflect sampling (microfacet.hpp:237-277): the half vector bisects wi/wo, the
    sample stays in the upper hemisphere, f and pdf are finite and non-negative; rejected samples
    (lo below the horizon, li.wh < 0) terminate with pdf 0."""
    O, _ = zoo
    rng = np.random.default_rng(3)
    k = 4000
    n = np.tile(np.array([[0, 1, 0]], np.float32), (k, 1))
    wi = unit(rng, k); wi[:, 1] = np.abs(wi[:, 1]) * 0.8 + 0.2; wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    u2 = rng.random((k, 2)).astype(np.float32)
    wo, f, pdf, fl = O.bsdf_sample(4, n, wi, u2)
    ok = pdf > 0
    assert ok.mean() > 0.75 and np.isfinite(f).all() and (f >= 0).all()
    assert (fl[ok] == abi.BSDF_REFLECT).all()  # microfacet reflect carries REFLECT only (bsdf.hpp:70-72)
    assert ((wo[ok] * n[ok]).sum(1) > 0).all()
    # mirror check: wh bisects wi and wo
    wh = wi[ok] + wo[ok]; wh /= np.linalg.norm(wh, axis=1, keepdims=True)
    assert np.allclose((wi[ok] * wh).sum(1), (wo[ok] * wh).sum(1), atol=1e-5)


def test_mix_rules(zoo):
    """bsdf_t::sample: lobe = floor(u.x * lobes); other lobes whose flags are a subset add f*w and pdf; the
    pdf is AVERAGED (bsdf.cpp:226-245)."""
    O, sc = zoo
    rng = np.random.default_rng(4)
    k = 2000
    n = unit(rng, k); wi = n * 0.8 + unit(rng, k) * 0.2; wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    u2 = rng.random((k, 2)).astype(np.float32)
    wo, f, pdf, fl = O.bsdf_sample(8, n, wi, u2)  # diffuse + glossy microfacet
    first = u2[:, 0] * 2 < 1
    assert (fl[first & (pdf > 0)] == (abi.BSDF_REFLECT | abi.BSDF_DIFFUSE)).all()
    assert (fl[~first & (pdf > 0)] == abi.BSDF_REFLECT).all()
    # chosen diffuse: glossy (flags REFLECT) is a subset of REFLECT|DIFFUSE -> matched; chosen glossy: diffuse is not
    ok = first & (pdf > 0)
    f_d = np.float32(0.4) / PI
    assert (f[ok, 0] >= f_d - 1e-6).all()
    ok2 = ~first & (pdf > 0)
    single = O.bsdf_sample(4, n, wi, np.stack([np.minimum(u2[:, 0] * 2 - 1, 0.9999999), u2[:, 1]], 1).astype(np.float32))
    assert ok2.sum() > 100 and np.isfinite(f[ok2]).all()
    # an emitter-only material (0 lobes) terminates the path (SURVEY A-10)
    wo, f, pdf, fl = O.bsdf_sample(len(sc.materials) - 1, n, wi, u2)
    assert (pdf == 0).all() and (f == 0).all()


def test_all_closures_finite(zoo):
    O, sc = zoo
    rng = np.random.default_rng(5)
    k = 3000
    n = unit(rng, k); wi = unit(rng, k); wo = unit(rng, k); u2 = rng.random((k, 2)).astype(np.float32)
    for m in range(len(sc.materials) - 1):
        f = O.bsdf_f(m, n, wi, wo)
        swo, sf, pdf, fl = O.bsdf_sample(m, n, wi, u2)
        lit = (n * wi).sum(1) >= 0  # the integrator only evaluates f() for unmasked light directions (spt.hpp:138)
        assert np.isfinite(f[lit]).all(), m
        live = (pdf > 0) & (sf.sum(1) > 0)  # black f (e.g. total internal reflection) also terminates
        assert np.isfinite(swo[live]).all() and np.isfinite(sf[live]).all() and np.isfinite(pdf).all(), m
        assert np.allclose(np.linalg.norm(swo[live], axis=1), 1, atol=1e-4), m
