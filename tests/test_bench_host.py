"""The host-only legs of bench.py (no GPU): the CPU baseline object and the committed-traffic lookup keep the shape the
bench contract asks for."""
import argparse
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_cpu_baseline_object_has_the_contract_keys():
    import bench
    from phosphorus_mk2_amd import scenes
    a = argparse.Namespace(spp=16, depth=5, width=128, height=96, cpu_tiles=6, cpu_spp=1, seed=1)
    base, visits = bench.cpu_baseline(scenes.soup(2000, width=128, height=96), a)
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread"):
        assert k in base
    assert base["kind"] == "port" and base["unit"] == "Mrays/s" and base["value"] > 0 and base["cores"] >= 1
    assert base["single_thread"]["value"] > 0
    (vn, vl), (vns, vls) = visits["closest"], visits["shadow"]
    assert vn > 0 and vl > 0 and vns > 0 and vls > 0  # node / leaf-packet visits per ray price the algorithmic bytes


def test_committed_traffic_comes_from_the_newest_profile_of_this_workload():
    import bench
    a = argparse.Namespace(triangles=100000, width=1280, height=720, depth=9)
    traffic, src = bench.committed_traffic(a)
    assert traffic and traffic > 1e9 and src.startswith("profiles/r") and os.path.exists(os.path.join(ROOT, src))
    a.triangles = 12345
    assert bench.committed_traffic(a) == (None, None)  # another workload: no PMC figure is claimed
