"""GPU (MI355X): bench.py itself at a reduced size — the line the driver keeps carries every key the contract and VERDICT r2 item 2 name, the
figures are consistent with each other, and PMC figures are attached only when they were measured on the kernel sources of the run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"] + list(args),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys_and_is_consistent():
    d = _bench()                                         # the default workload: 1080p, spp 64, depth 8, Float64 (a few seconds on the device)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "end_to_end", "configs", "kernel_source_hash"):
        assert k in d, k
    assert d["metric"] == "Msamples/sec at 1920x1080 spp=64 depth=8; fraction of HBM roofline" and d["unit"] == "Msamples/s" and d["dtype"] == "f64"
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    # the kernel fits inside the step it was timed in (its events bracket the LAST timed step, ms_per_step is the mean of all: 5 % for the step-to-step spread)
    assert r["avg_launch_ms"] * r["launches"] <= d["ms_per_step"] * 1.05
    assert abs(d["value"] - d["config"]["samples_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    if r["traffic"] is None:                              # PMC figures of other kernel sources never reach the line
        assert r["traffic_stale"] in (True, False) and r["valu"] is None
    else:
        assert r["traffic_stale"] is False and r["traffic_source"].startswith("profiles/traffic_") and 0 < r["valu"]["issue_slots"] <= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["end_to_end"]["ms"] > d["ms_per_step"] * 0.5 and d["end_to_end"]["d2h_bytes"] == 3 * 1920 * 1080 * 8
    t8 = d["tiles_of_8_gpus"]      # every rank's tile of `--gpus 8` rendered here in turn: 135 rows each, about what the frame costs
    assert t8["rows_per_rank"] == [135] and t8["spp"] == 512 and 0.7 < t8["render_bound_of_weak_scaling_efficiency"] < 1.15
    for cname, scene, spp, depth in (("c4", "scene s3", 256, 8), ("c5", "scene s4", 64, 12)):
        e = d["configs"][cname]
        assert scene in e["workload"] and "spp=%d depth=%d" % (spp, depth) in e["metric"]
        for pr in ("f64", "f32"):
            assert e[pr]["value"] > 0 and e[pr]["roofline"]["launches"] == spp // 64 and e[pr]["roofline"]["frac"] > 0


def test_config_c5_is_labelled_depth_12():
    d = _bench("--config", "c5", "--prec", "f32", "--no-extras", "--no-cpu-baseline", "--no-alt-precision")
    assert "depth=12" in d["metric"] and d["config"]["max_depth"] == 12 and d["config"]["scene"] == "s4" and d["dtype"] == "f32"
    assert d["roofline"]["segments_per_sample"] > 1.0
    assert d["roofline"]["parked_per_sample"] > 0          # the mesh lists are part of the algorithmic bytes


def test_two_ranks_launched_the_driver_way_on_one_gpu():
    """The N > 1 path of bench.py exactly as the driver starts it (torch.distributed.run, one process per rank) — on the one-GPU box both
    ranks share GPU 0 and the exchange runs over gloo (--rehearse-on-one-gpu: not a measurement).  Weak scaling: spp 64 per rank, so the
    frame is spp 128; every rank renders its interleaved stripes; ONE line, from rank 0, with the per-rank render / gather times."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-on-one-gpu"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["spp"] == 128 and "spp=128" in d["metric"]
    assert d["config"]["samples_per_step"] == 1920 * 1080 * 128 and d["cpu_baseline"] is None and d["configs"] is None
    p = d["per_rank"]
    assert p["rows_per_rank"] == {"max": 540, "min": 540}                 # 1080 rows dealt row by row to 2 ranks
    assert 0 < p["render_ms"]["min"] <= p["render_ms"]["max"] <= d["ms_per_step"] * 1.05 and p["gather_ms"]["max"] > 0
    assert abs(d["value"] - d["config"]["samples_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
