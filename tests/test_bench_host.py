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
    cap, src = bench.load_capture(bench.workload_tag("soup", 100000, 1280, 720, 9))
    assert cap and src.startswith("profiles/r") and os.path.exists(os.path.join(ROOT, src))
    assert cap["kernels"]["k_trace"]["hbm_bytes_per_launch_corrected"] > 1e9
    assert bench.workload_tag("soup", 12345, 1280, 720, 9) is None and bench.load_capture(None) == (None, None)  # another workload: no PMC figure is claimed
    peak, psrc, refused = bench.committed_valu_peak()
    assert refused is None, refused  # the newest committed peak was measured on the node / triangle test of THIS tree
    assert peak["node_tests_per_s"] > 1e10 and peak["tri_tests_per_s"] > peak["node_tests_per_s"] and os.path.exists(os.path.join(ROOT, psrc))
    assert peak["src_hash"] == bench.priced_source_hash() and 1.5e9 < peak["clock_hz"] < 2.6e9


def test_a_peak_measured_on_other_sources_is_refused(monkeypatch):
    """the VALU peak behind roofline.frac carries the hash of the bvh8.h functions it timed (scripts/src_hash.py); when the tree's
    node test or triangle test has changed since, bench.py reports frac = None with the reason instead of a fraction of a stale peak"""
    import bench
    from scripts import src_hash
    h = src_hash.priced_source_hash()
    assert len(h) == 16 and h == bench.priced_source_hash()
    # the hash follows the text of the priced functions and nothing else
    src = open(src_hash.BVH8).read()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for edit, same in ((src.replace("struct Hit {", "struct Hit  {"), True),
                           (src.replace("const float pad_far = 1.00000095367431640625f;", "const float pad_far = 1.0000019073486328125f;"), False),
                           (src.replace("const bool umask = us >= 0.0f;", "const bool umask = us > 0.0f;"), False)):
            assert edit != src
            f = os.path.join(d, "bvh8.h"); open(f, "w").write(edit)
            assert (src_hash.priced_source_hash(f) == h) == same
    work = {k: {"rays": r, "node_visits_lds_per_ray": 5.3, "node_visits_mem_per_ray": 9.0, "tri_tests_per_ray": 5.5} for k, r in (("closest", 441_000_000), ("shadow", 123_000_000))}
    work["wave"] = {"lanes_per_node_block": 52.0, "lanes_per_tri_block": 22.0}
    acc = bench.new_acc()
    acc.update({"closest": 441_000_000, "shadow": 123_000_000, "camera": 235_929_600, "closest_ms": 73.0, "shade_ms": 16.0, "launches": 10, "frame_ms": 90.0})
    ok = bench.roofline(acc, 1, work, "100k", None)
    assert ok["frac"] is not None and "frac_unavailable" not in ok and ok["work"]["peak_src_hash"] == h
    monkeypatch.setattr(bench, "priced_source_hash", lambda: "0123456789abcdef")
    stale = bench.roofline(acc, 1, work, "100k", None)
    assert stale["frac"] is None and stale["achieved"] is None and "stale peak" in stale["frac_unavailable"] and "0123456789abcdef" in stale["frac_unavailable"]
    assert stale["stream_GBps"] == ok["stream_GBps"] and stale["diagnostics"] == ok["diagnostics"]  # everything measured stays
    line = bench.compact_roofline(stale)
    assert line["frac"] is None and "stale peak" in line["frac_unavailable"]


def test_roofline_fraction_is_work_based_and_diagnostics_use_their_own_capture():
    """the k_trace roofline from synthetic inputs: `frac` is the work-based VALU fraction (minimum ALU time of the counted work over
    THIS run's kernel time); every diagnostic is a counter of one committed capture over the kernel time OF THAT CAPTURE, so it
    does not move when this run's time does"""
    import bench
    work = {k: {"rays": r, "node_visits_lds_per_ray": 5.3, "node_visits_mem_per_ray": 9.0, "tri_tests_per_ray": 5.5} for k, r in (("closest", 441_000_000), ("shadow", 123_000_000))}
    work["wave"] = {"lanes_per_node_block": 52.0, "lanes_per_tri_block": 22.0}
    roofs = []
    for ms in (73.0, 146.0):
        acc = bench.new_acc()
        acc.update({"closest": 441_000_000, "shadow": 123_000_000, "camera": 235_929_600, "closest_ms": ms, "shade_ms": 16.0, "launches": 10, "frame_ms": 90.0})
        roofs.append(bench.roofline(acc, 1, work, "100k", {"closest": (14.4, 7.2), "shadow": (13.1, 6.5)}))
    a, b = roofs
    assert a["bound"] == "valu" and 0 < a["frac"] <= 1 and abs(a["frac"] - a["achieved"] / a["peak"]) < 1e-9
    assert abs(a["frac"] - a["work"]["min_alu_ms_per_frame"] / 73.0) < 1e-9 and abs(b["frac"] - a["frac"] / 2) < 1e-9
    D = a["diagnostics"]
    assert D["source"].startswith("profiles/r") and D["kernel_ms_in_capture"] > 0
    for name in ("valu_issue", "vector_l1", "l2", "hbm"):
        assert 0 < D[name]["frac"] <= 1.0 and abs(D[name]["frac"] - D[name]["achieved"] / D[name]["peak"]) < 1e-9, name
        assert D[name] == b["diagnostics"][name], name
    cap = __import__("json").load(open(os.path.join(ROOT, D["source"])))
    k = cap["kernels"]["k_trace"]
    t = D["kernel_ms_in_capture"] * 1e-3
    assert abs(D["valu_issue"]["achieved"] * 1e9 - k["counters"]["SQ_INSTS_VALU"] / t) < 1e-6 * k["counters"]["SQ_INSTS_VALU"] / t
    assert a["traffic"] == k["hbm_bytes_per_launch_corrected"] and "GBps" in a["algorithmic_ref_layout"] and a["device_layout"]["bytes_per_ray"]["closest"] > 48


def test_camera_rays_are_not_k_traces_work():
    """k_trace_primary walks the camera rays: the k_trace roofline counts neither their number nor their time, and the record of the
    primary kernel carries its own HIP-event time and the packet statistics of the instrumented build"""
    import bench
    work = {k: {"rays": r, "node_visits_lds_per_ray": 6.4, "node_visits_mem_per_ray": 13.5, "tri_tests_per_ray": 7.0} for k, r in (("closest", 205_000_000), ("shadow", 123_000_000))}
    work["wave"] = {"lanes_per_node_block": 50.0, "lanes_per_tri_block": 18.0}
    work["primary"] = {"rays": 236_000_000, "packets": 921_600, "fallback_packets": 2000, "node_tests_per_packet": 9.5, "tri_tests_per_packet": 3.7, "lanes_improved_per_tri_test": 12.0}
    acc = bench.new_acc()
    acc.update({"closest": 441_000_000, "shadow": 123_000_000, "camera": 236_000_000, "primary_rays": 236_000_000, "primary_ms": 3.7, "primary_launches": 1,
                "closest_ms": 47.0, "shade_ms": 12.0, "shade_kernel_ms": 11.0, "launches": 9, "frame_ms": 63.5})
    r = bench.roofline(acc, 1, work, None, None)
    assert abs(r["kernel_rays_per_s"] - (205e6 + 123e6) / 0.047) < 1.0 and abs(r["rays_per_launch"] - 328e6 / 9) < 1.0
    assert abs(r["frac"] - r["work"]["min_alu_ms_per_frame"] / 47.0) < 1e-9
    p = bench.primary_record(acc, 1, work)
    assert p["kernel"] == "k_trace_primary" and abs(p["rays_per_s"] - 236e6 / 3.7e-3) < 1.0 and p["packets"]["node_tests_per_packet"] == 9.5
    assert bench.kernel_ms(acc, 1) == {"primary": 3.7, "trace": 47.0, "shade": 11.0, "begin_pass_film": 1.0}
    assert bench.primary_record(bench.new_acc(), 1, None) is None


def test_shade_roofline_counts_algorithmic_bytes_per_entry():
    import bench
    acc = bench.new_acc()
    acc.update({"closest": 600_000_000, "shadow": 400_000_000, "camera": 500_000_000, "shade_kernel_ms": 40.0, "shade_launches": 18})
    r = bench.shade_roofline(acc, 1, {"shade_general": 1}, None)
    # 500 M camera entries (hit + triangle record in, radiance out: no ray, no path state) + 100 M later entries, each a survivor of a shade launch
    alg = 500e6 * bench.SHADE_BYTES_IN_CAMERA + 100e6 * bench.SHADE_BYTES_IN + 100e6 * bench.SHADE_BYTES_SURVIVOR + 400e6 * bench.SHADE_BYTES_NEE
    assert bench.SHADE_BYTES_IN == 80 and bench.SHADE_BYTES_IN_CAMERA == 48
    assert r["kernel"] == "k_shade_g" and r["bound"] == "hbm" and abs(r["achieved"] - alg / 0.040 / 1e9) < 1e-6 and 0 < r["frac"] <= 1


def test_newest_committed_bench_record_keeps_the_contract():
    """the newest profiles/r*_bench_full.json (written by `python bench.py` on an MI355X) carries every key of the bench contract,
    fractions that are fractions, and the secondary records"""
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
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert all(0 < s["roofline"]["frac"] <= 1 for s in d["secondary"])
    # Lambert soups: every pixel finite, strictly.  The general-closure scenes may hold a few NaN pixels at thousands of samples per pixel — the
    # reference's sheen lobe has no guard for a direction within rounding of the normal (src/bsdf/sheen.hpp:51-64; the oracle has the same pixels:
    # profiles/r06_nonfinite_probe_*.json) — and from round 6 on the record says WHERE they are
    assert all(s["film_finite"] or (s["roofline"]["kernel"] == "k_shade_g" and s.get("film_finite_fraction", 1.0) > 0.9999) for s in d["secondary"])
    assert all(s["film_finite"] for s in d["secondary"] if s["roofline"]["kernel"] == "k_trace")
    if os.path.basename(recs[-1]) >= "r06":
        for s in d["secondary"]:
            assert (len(s["nonfinite_pixels_xy"]) == 0) == s["film_finite"], s["workload"]
    if os.path.basename(recs[-1]) >= "r04":  # round 4 on: one protocol at every N, the stdout line is a compact digest of this record
        assert d["config"]["frames_in_flight"] == 1 and "one frame in flight" in d["value_definition"].lower()
        assert 0 < d["value_host_film"] < 1.05 * d["value"] and 0.9 * d["value"] < d["value_two_frames_in_flight"] < 1.25 * d["value"]
        assert d["config"]["hbm_bytes_per_rank"] > d["config"]["bvh_bytes"] and d["config"]["hbm_bytes_per_rank_two_frames_in_flight"] > d["config"]["hbm_bytes_per_rank"]
        assert rf["bound"] == "valu" and abs(rf["frac"] - rf["work"]["min_alu_ms_per_frame"] / rf["work"]["k_trace_ms_per_frame"]) < 1e-9
        assert 0 < rf["stream_GBps"] < 8000 and all(0 < rf["diagnostics"][k]["frac"] <= 1 for k in ("valu_issue", "vector_l1", "l2", "hbm"))
        if os.path.basename(recs[-1]) >= "r06":  # + BASELINE configs 3 / 5 on mesh geometry in a closed room; the peak is priced at k_trace's own clock and asm-checked
            kinds = [s["roofline"]["kernel"] for s in d["secondary"]]
            assert kinds[:3] == ["k_trace", "k_trace", "k_shade_g"] and set(kinds[3:]) == {"k_shade_g"} and len(kinds) in (5, 6)
            room = [s for s in d["secondary"] if "bmw_showroom" in s["workload"]]
            assert room and all(s["rays_per_camera_sample"] > 6 for s in room) and all("roofline_k_trace" in s for s in room)
            w = rf["work"]
            assert w["peak_asm_check"]["node_test_cvt_ubyte_micro"] == w["peak_asm_check"]["node_test_cvt_ubyte_k_trace"] == 48
            assert abs(w["peak_asm_check"]["node_test_valu_micro"] / w["peak_asm_check"]["node_test_valu_k_trace"] - 1) <= 0.05
            assert 0.85 < w["peak_clock_scale"] < 1.0 and abs(w["peak_node_tests_per_s"] - w["peak_node_tests_per_s_at_micro_clock"] * w["peak_clock_scale"]) < 1e-3 * w["peak_node_tests_per_s"]
        else:
            assert len(d["secondary"]) == 4 and [s["roofline"]["kernel"] for s in d["secondary"]] == ["k_trace", "k_trace", "k_shade_g", "k_shade_g"]
    elif os.path.basename(recs[-1]) >= "r03":  # round 3: work-based fraction, both film sinks, the config 3 / 5 stand-ins
        assert rf["bound"] == "valu" and abs(rf["frac"] - rf["work"]["min_alu_ms_per_frame"] / rf["work"]["k_trace_ms_per_frame"]) < 1e-9
        assert all(0 < rf["diagnostics"][k]["frac"] <= 1 for k in ("valu_issue", "vector_l1", "l2", "hbm"))
        assert d["value_hbm_film"] == d["value"] and 0 < d["value_host_film"] < 1.05 * d["value"]
        assert len(d["secondary"]) == 4 and [s["roofline"]["kernel"] for s in d["secondary"]] == ["k_trace", "k_trace", "k_shade_g", "k_shade_g"]
        if "value_two_frames_in_flight" in d:  # r03_zzg on: the N = 1 rate with two frames in flight, what every rank of an N > 1 run does
            assert d["config"]["frames_in_flight"] == 1 and 0.9 * d["value"] < d["value_two_frames_in_flight"] < 1.2 * d["value"]
        if "primary" in d:  # r03_zq on: the camera rays have their own kernel, k_trace's roofline is about the other rays
            pr = d["primary"]
            assert pr["kernel"] == "k_trace_primary" and pr["rays_per_step"] == d["config"]["camera_samples_per_step"] and pr["ms_per_step"] > 0
            assert abs(d["config"]["kernel_ms_per_step"]["primary"] - pr["ms_per_step"]) < 1e-9 and pr["packets"]["fallback_packets"] < 0.01 * pr["packets"]["packets"]
            assert abs(rf["kernel_rays_per_s"] - (d["config"]["rays_per_step"] - pr["rays_per_step"]) / (rf["work"]["k_trace_ms_per_frame"] * 1e-3)) < 1e-6 * rf["kernel_rays_per_s"]
    else:
        assert all(0 < c["frac"] <= 1 for c in rf["ceilings"].values()) and rf["bound"] in rf["ceilings"] and len(d["secondary"]) == 2


def test_the_stdout_line_is_a_compact_digest_that_a_4_KB_reader_can_parse():
    """round 3's line grew to 21.9 KB and the driver recorded `parsed: null`: the line bench.py prints is now made from the full record
    by compact_line() and must stay under LINE_BUDGET whatever the full record holds — checked on the newest committed full record"""
    import glob, json
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_full.json")))
    full = json.loads(open(recs[-1]).read().strip().splitlines()[-1])
    full.setdefault("value_definition", bench.VALUE_DEFINITION)
    line = bench.compact_line(full, os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    assert len(line) < bench.LINE_BUDGET == 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert abs(d["value"] - full["value"]) < 1e-5 * full["value"] and "workload" in d["config"] and "model" not in d["config"]
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["kernel"] == "k_trace" and d["roofline"]["traffic"] > 1e9
    assert 0 < d["roofline"]["hbm_frac"] < 1 and d["roofline"]["hbm_GBps"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert len(d["secondary"]) == len(full["secondary"]) >= 4 and all(set(s) == {"workload", "value", "ms_per_step", "roofline"} and 0 < s["roofline"]["frac"] <= 1 for s in d["secondary"])
    assert len({s["workload"] for s in d["secondary"]}) == len(d["secondary"])  # the shortened names still tell the workloads apart
    assert all(("frac_by_counters" in s["roofline"]) == (s["roofline"]["bound"] == "hbm") for s in d["secondary"])  # HBM-bound kernels: by counters beside the algorithmic fraction
    assert d["roofline_shade"]["kernel"] == "k_shade" and 0 < d["roofline_shade"]["frac_by_counters"] < d["roofline_shade"]["frac"] <= 1
    assert "hardware threads" in d["cpu_baseline"]["cores_of"]
    # a record ten times as rich still fits: the optional parts go first, the contract keys stay
    fat = dict(full, secondary=full["secondary"] * 12)
    line = bench.compact_line(fat, None)
    assert len(line) < bench.LINE_BUDGET and json.loads(line)["roofline"]["frac"] == d["roofline"]["frac"]


def test_gpus_flag_decides_the_launch_before_anything_touches_the_gpu():
    """--gpus N (src/core.cpp:103-115: one device per GPU; here one process per GPU): alone it spawns torch.distributed.run with N
    ranks; under a launcher WORLD_SIZE must agree; fewer visible devices than ranks is an error, never a silent share of GPU 0"""
    import bench
    D = bench.launch_decision
    assert D(1, {}, 1) == ("run", 1) and D(1, {}, 0) == ("run", 1) and D(1, {}, None) == ("run", 1)  # N = 1 on a box without a GPU fails later, loudly, in phx_dev_make
    assert D(8, {}, 8) == ("spawn", 8) and D(2, {}, 8) == ("spawn", 2)
    assert D(8, {}, 1)[0] == "error" and "only 1 GPU" in D(8, {}, 1)[1]
    assert D(8, {"WORLD_SIZE": "8"}, 8) == ("run", 8) and D(4, {"WORLD_SIZE": "4"}, 8) == ("run", 4)
    assert D(8, {"WORLD_SIZE": "1"}, 8)[0] == "error" and D(1, {"WORLD_SIZE": "8"}, 8)[0] == "error" and D(8, {"WORLD_SIZE": "8"}, 4)[0] == "error"
    assert D(2, {}, 1, rehearsal=True) == ("spawn", 2) and D(2, {"WORLD_SIZE": "2"}, 1, rehearsal=True) == ("run", 2)
    assert D(0, {}, 8)[0] == "error"


def test_bench_refuses_more_ranks_than_gpus():
    """no GPU here: `bench.py --gpus 2` must exit non-zero with a message instead of rendering the frame on whatever device 0 is"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PHX_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="4"))
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr
