"""The host-only legs of bench.py (no GPU): the CPU baseline object, the committed-measurement lookups and the roofline
arithmetic keep the shape the bench contract asks for."""
import argparse
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_cpu_baseline_object_has_the_contract_keys():
    import bench
    from phosphorus_mk2_amd import scenes
    a = argparse.Namespace(spp=16, depth=5, width=128, height=96, cpu_spp=1, seed=1, cpu_seconds=0.3)
    base, visits = bench.cpu_baseline(scenes.soup(2000, width=128, height=96), a, seconds=0.3, thread_counts=[1, 2])
    for k in ("value", "unit", "cores", "kind", "sample", "scaling", "host"):
        assert k in base
    assert base["kind"] == "port" and base["unit"] == "Mrays/s" and base["value"] > 0 and base["cores"] in (1, 2)
    assert [r["threads"] for r in base["scaling"]] == [1, 2] and all(r["seconds"] >= 0.3 and r["Mrays_per_s"] > 0 for r in base["scaling"])
    assert base["scaling"][0]["efficiency_vs_1_thread"] == 1.0
    assert base["host"]["hardware_threads"] >= base["host"]["physical_cores"] >= 1
    (vn, vl), (vns, vls) = visits["closest"], visits["shadow"]
    assert vn > 0 and vl > 0 and vns > 0 and vls > 0  # node / leaf-packet visits per ray price the reference-layout bytes


def test_committed_measurements_come_from_the_newest_profile_of_this_workload():
    import bench
    a = argparse.Namespace(triangles=100000, width=1280, height=720, depth=9)
    traffic, src = bench.committed_traffic(a)
    assert traffic and traffic > 1e9 and src.startswith("profiles/r") and os.path.exists(os.path.join(ROOT, src))
    a.triangles = 12345
    assert bench.committed_traffic(a) == (None, None)  # another workload: no PMC figure is claimed
    peak, psrc = bench.committed_valu_peak()
    assert peak["node_tests_per_s"] > 1e10 and peak["tri_tests_per_s"] > peak["node_tests_per_s"] and os.path.exists(os.path.join(ROOT, psrc))


def test_roofline_fractions_are_fractions():
    """the roofline object from synthetic inputs: every ceiling carries achieved / peak / frac with frac = achieved / peak <= 1 for
    physically possible inputs, and `bound` names the highest one"""
    import bench
    acc = {"closest": 441_000_000, "shadow": 123_000_000, "closest_ms": 73.0, "shade_ms": 16.0, "launches": 10, "frame_ms": 90.0}
    work = {k: {"rays": r, "node_visits_lds_per_ray": 5.3, "node_visits_mem_per_ray": 9.0, "tri_tests_per_ray": 5.5} for k, r in (("closest", 441_000_000), ("shadow", 123_000_000))}
    work["wave"] = {"lanes_per_node_block": 52.0, "lanes_per_tri_block": 22.0}
    a = argparse.Namespace(triangles=100000, width=1280, height=720, depth=9)
    pmc, src = bench.committed_pmc(a)
    roof = bench.roofline(acc, 1, work, pmc, src, {"closest": (14.4, 7.2), "shadow": (13.1, 6.5)}, a)
    assert roof["bound"] in roof["ceilings"] and roof["frac"] == max(c["frac"] for c in roof["ceilings"].values())
    for name, c in roof["ceilings"].items():
        assert 0 < c["frac"] <= 1.0, name
        assert abs(c["frac"] - c["achieved"] / c["peak"]) < 1e-9, name
    assert roof["traffic"] > 1e9 and "GBps" in roof["algorithmic_ref_layout"] and roof["device_layout"]["bytes_per_ray"]["closest"] > 48


def test_newest_committed_bench_record_keeps_the_contract():
    """the newest profiles/r*_bench_full.json (written by `python bench.py` on an MI355X) carries every key of the bench contract,
    fractions that are fractions, and the two secondary records"""
    import glob, json
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_full.json")))
    assert recs
    d = json.loads(open(recs[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["vs_baseline"] is None
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["traffic"] > 1e9
    assert all(0 < c["frac"] <= 1 for c in rf["ceilings"].values()) and rf["bound"] in rf["ceilings"]
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert len(d["secondary"]) == 2 and all(s["film_finite"] and 0 < s["roofline"]["frac"] <= 1 for s in d["secondary"])
