"""CPU: the arithmetic of bench.py's roofline record (pure functions, no GPU): algorithmic bytes per organisation and precision, the
fractions, and that `bound` follows the committed PMC summaries (profiles/traffic_*.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _counters(**kw):
    c = dict(samples=1000, segments=2190, rays_enqueued=400, radiance_rmw=10, radiance_stores=1000, passes=1, launches=3, kernel_ms=6.2,
             bounce_kernel_ms=5.6, bounce_launches=1)
    c.update(kw)
    return c


def test_algorithmic_bytes_per_organisation_and_precision():
    c = _counters()
    # k_path: 10 values per packet, + a 4-byte reference word in Float32; written once, read once; radiance 3 values per store, 6 per RMW
    assert bench.algorithmic_bytes(c, 8, "wavefront") == 2 * 80 * 400 + (3 * 1000 + 6 * 10) * 8
    assert bench.algorithmic_bytes(c, 4, "wavefront") == 2 * 44 * 400 + (3 * 1000 + 6 * 10) * 4
    assert bench.algorithmic_bytes(c, 4, "bounce") == 2 * 40 * 400 + (3 * 1000 + 6 * 10) * 4
    # mesh scenes: an entry of a wave's mesh list is 12 values, written once and read once
    m = _counters(rays_parked=70)
    assert bench.algorithmic_bytes(m, 8, "wavefront") - bench.algorithmic_bytes(c, 8, "wavefront") == 2 * 96 * 70
    assert bench.algorithmic_bytes(m, 4, "wavefront") - bench.algorithmic_bytes(c, 4, "wavefront") == 2 * 48 * 70


def test_roofline_record_fields_and_bound(tmp_path, monkeypatch):
    c = _counters(samples=132710400, segments=290665553, rays_enqueued=44000000, radiance_stores=132710400, radiance_rmw=300000, bounce_kernel_ms=5.6)
    nbytes = bench.algorithmic_bytes(c, 8, "wavefront")
    # a PMC summary taken on THESE kernel sources is attached ...
    prof = tmp_path / "profiles"
    prof.mkdir()
    fresh = {"kernel": "k_path", "source_hash": bench.kernel_source_hash(), "hbm_bytes_per_launch": 13.3e9,
             "valu": {"issue_slots": 0.43, "valubusy_rocprof": 0.89, "lane_utilisation": 0.72, "simd_cycles_per_valu_inst": 4.5, "wave_cycle_shares": None}}
    json.dump(fresh, open(prof / "traffic_s1_f64.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: fresh["source_hash"])
    r = bench.roofline_record(c, "f64", "wavefront", "s1", True)
    assert r["kernel"] == "k_path" and r["launches"] == 1 and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - nbytes / 5.6e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert r["avg_launch_ms"] * r["launches"] <= c["kernel_ms"]
    assert r["traffic"] == round(13.3e9) and r["traffic_stale"] is False and r["valu"]["issue_slots"] == 0.43 and r["valu"]["issue_slots"] <= 1.0
    assert r["bound"] == "valu"
    # ... one taken on other sources is not: no traffic, no VALU figures, no bound claimed, and the line says so
    stale = dict(fresh, source_hash="0123456789abcdef")
    json.dump(stale, open(prof / "traffic_s1_f64.json", "w"))
    r1 = bench.roofline_record(c, "f64", "wavefront", "s1", True)
    assert r1["traffic"] is None and r1["valu"] is None and r1["traffic_stale"] is True and r1["bound"] is None and "stale" in r1["traffic_source"]
    # not the shape the summaries were taken on: nothing is attached
    r2 = bench.roofline_record(c, "f64", "wavefront", "s1", False)
    assert r2["traffic"] is None and r2["valu"] is None and r2["bound"] == "hbm" and r2["traffic_stale"] is False


def test_committed_pmc_summaries_are_sane():
    """The committed summaries carry measured figures only (a modelled ceiling the kernel exceeds is not a ceiling): the share of VALU
    issue slots is <= 1 by construction, no priced model is left, and every summary names the kernel sources it was taken on."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")):
        tj = json.load(open(f))
        v = tj.get("valu") or {}
        if "issue_slots" in v:
            assert 0.0 < v["issue_slots"] <= 1.0 and 0.0 < v["lane_utilisation"] <= 1.0, f
            assert "priced_model" not in v and "issue_frac" not in v, f
            assert "source_hash" in tj, f


def test_configs_name_the_baseline_configurations():
    assert bench.CONFIGS["c3"]["spp"] == 64 and bench.CONFIGS["c3"]["depth"] == 8 and bench.CONFIGS["c3"]["scaling"] == "weak" and bench.CONFIGS["c3"]["scene"] == "s1"
    assert bench.CONFIGS["c4"]["spp"] == 256 and bench.CONFIGS["c4"]["scaling"] == "strong" and bench.CONFIGS["c4"]["scene"] == "s3"
    assert bench.CONFIGS["c5"]["scene"] == "s4" and bench.CONFIGS["c5"]["depth"] == 12


def test_metric_string_follows_the_workload():
    assert bench.metric_string(1920, 1080, 64, 8) == "Msamples/sec at 1920x1080 spp=64 depth=8; fraction of HBM roofline"      # BASELINE.json's string
    assert "depth=12" in bench.metric_string(1920, 1080, 64, 12) and "spp=256" in bench.metric_string(1920, 1080, 256, 8)


def test_bench_line_keys_are_declared():
    """The keys VERDICT r2 item 2 asks the driver-run line to carry are produced by main() (static check: no GPU here)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"configs": other_configs', '"end_to_end": end_to_end', '"roofline": roof', '"cpu_baseline": cpu', '"kernel_source_hash": src_hash',
                'for cname in ("c4", "c5")', 'for pr in ("f64", "f32")', '"traffic_stale"'):
        assert key in src, key


def test_c4_tile_arithmetic_on_8_gpus():
    """bench.py --config c4 --gpus 8: interleaved rows give every rank 135 rows of 1080, and spp stays 256 in total."""
    sys.path.insert(0, os.path.join(ROOT, "julia-spira_amd"))
    from spira_hip import distributed as D
    rows = [D.tile_params(1080, 8, r)["rows"] for r in range(8)]
    assert rows == [135] * 8
    assert bench.CONFIGS["c4"]["scaling"] == "strong"      # spp_total = spp (not spp * world): bench.py main()


def test_the_library_carries_the_hash_of_the_sources_it_was_built_from():
    """spira_build_id() (csrc/Makefile: sha256 of the kernel sources + the Makefile) is what bench.py compares with the hash of the files on disk and with the
    `source_hash` of a committed PMC summary: figures measured on other sources — or a library older than its sources — never reach a bench line."""
    sys.path.insert(0, os.path.join(ROOT, "julia-spira_amd"))
    from spira_hip import _binding as B
    assert B.build_id() == bench.kernel_source_hash() and len(B.build_id()) == 16
