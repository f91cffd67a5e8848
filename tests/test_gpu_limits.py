"""GPU (MI355X): odd shapes and limits of the hot path (sizes chosen so the oracle still finishes in seconds where used)."""
import numpy as np
import pytest

from spira_hip import scenes
from test_gpu_parity import _args, _close, _counts

pytestmark = pytest.mark.gpu


def test_tall_thin_and_many_slots(gpu, oracle):
    s = scenes.scene_s2()
    ns, nm, nt = _counts(s)
    for (W, H, spp, depth) in [(17, 301, 3, 4), (2, 64, 40, 3), (301, 2, 5, 6)]:
        hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=8), "f32")
        ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=8), "f32")
        assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg, (W, H)


def test_max_depth_255(gpu, oracle):
    s = scenes.scene_s3()          # closed box: paths really run deep
    ns, nm, nt = _counts(s)
    hdr, _ = gpu.render(*_args(s), gpu.make_params(48, 27, 2, 255, ns, nm, nt, seed=3), "f64")
    ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(48, 27, 2, 255, ns, nm, nt, seed=3), "f64")
    assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg and oseg > 48 * 27 * 2 * 200


def test_4k_frame_runs_and_matches_tiles(gpu):
    """3840x2160 (8.3 M pixels): several slots per pass; the top half rendered as a slab equals the full frame's top half."""
    s = scenes.scene_s1()
    ns, nm, _ = _counts(s)
    full, _ = gpu.render(*_args(s), gpu.make_params(3840, 2160, 3, 4, ns, nm, seed=6), "f32")
    top, _ = gpu.render(*_args(s), gpu.make_params(3840, 2160, 3, 4, ns, nm, seed=6, row0=0, rows=1080), "f32")
    assert np.isfinite(full).all() and np.array_equal(top, full[:, :1080])
    c = gpu.counters()
    assert c["samples"] == 3840 * 1080 * 3


def test_small_batches_many_passes(gpu):
    s = scenes.scene_s1()
    ns, nm, _ = _counts(s)
    ref, _ = gpu.render(*_args(s), gpu.make_params(64, 36, 33, 5, ns, nm, seed=2), "f32")
    got, _ = gpu.render(*_args(s), gpu.make_params(64, 36, 33, 5, ns, nm, seed=2, batch_rays=1), "f32")   # 33 passes of one slot
    assert np.array_equal(ref, got) and gpu.counters()["passes"] == 33
