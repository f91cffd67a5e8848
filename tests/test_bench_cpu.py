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


def test_roofline_record_fields_and_bound():
    c = _counters(samples=132710400, segments=290665553, rays_enqueued=44000000, radiance_stores=132710400, radiance_rmw=300000, bounce_kernel_ms=5.6)
    r = bench.roofline_record(c, "f64", "wavefront", "s1", True)
    nbytes = bench.algorithmic_bytes(c, 8, "wavefront")
    assert r["kernel"] == "k_path" and r["launches"] == 1 and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - nbytes / 5.6e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert r["avg_launch_ms"] * r["launches"] <= c["kernel_ms"]
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_s1_f64.json")))
    assert tj["kernel"] == "k_path" and r["traffic"] == round(tj["hbm_bytes_per_launch"]) and r["valu"]["issue_frac"] == tj["valu"]["issue_frac"]
    assert r["bound"] == ("valu" if tj["valu"]["issue_frac"] > r["frac"] else "hbm")
    # not the headline shape: no PMC figures are attached
    r2 = bench.roofline_record(c, "f64", "wavefront", "s1", False)
    assert r2["traffic"] is None and r2["valu"] is None and r2["bound"] == "hbm"


def test_configs_name_the_baseline_configurations():
    assert bench.CONFIGS["c3"]["spp"] == 64 and bench.CONFIGS["c3"]["depth"] == 8 and bench.CONFIGS["c3"]["scaling"] == "weak"
    assert bench.CONFIGS["c4"]["spp"] == 256 and bench.CONFIGS["c4"]["scaling"] == "strong"
    assert bench.CONFIGS["c5"]["scene"] == "s4" and bench.CONFIGS["c5"]["depth"] == 12
