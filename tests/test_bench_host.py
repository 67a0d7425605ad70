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
