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


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_carried_rng_key_depth_boundary(gpu, oracle, prec):
    """Paths of a sphere scene carry their half-made RNG key through the hit queue while the bounce fits the key's low byte (max_depth <= 128:
    k_path, kCarry); deeper renders derive it from the path index at every scatter.  Both sides of the boundary, bit for bit, on a scene that
    compacts (S1: open) and one that never leaves the registers (S3: closed)."""
    for mk in (scenes.scene_s1, scenes.scene_s3):
        s = mk()
        ns, nm, nt = _counts(s)
        for depth in (128, 129):
            hdr, _ = gpu.render(*_args(s), gpu.make_params(40, 24, 2, depth, ns, nm, nt, seed=21), prec)
            ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(40, 24, 2, depth, ns, nm, nt, seed=21), prec)
            assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg, (mk.__name__, depth)


def test_camera_ray_sphere_constants(gpu, oracle, monkeypatch):
    """Camera rays share their origin, so k_path keeps what a sphere test does not need the direction for in LDS, one packet per sphere (closest_hit_local,
    CAM) — where the launch's LDS block has the room.  With the packets, without them (SPIRA_CAM_CONSTS=0), on a scene that just leaves the room (1024 spheres +
    1180 materials in Float64: 163 488 of 163 840 bytes with them) and on one that does not (1250 materials: 167 968): the same bits, and the oracle's."""
    from test_gpu_parity import random_scene
    rng = np.random.default_rng(31)
    for s, prec, (W, H, spp, depth), fits in [(scenes.scene_s1(), "f64", (64, 36, 3, 5), True), (random_scene(rng, 300, 20), "f32", (48, 27, 2, 4), True),
                                              (scenes.scene_s4(level=2), "f64", (48, 27, 2, 5), True), (random_scene(rng, 1024, 0, n_mats=1180), "f64", (24, 14, 1, 3), True),
                                              (random_scene(rng, 1024, 0, n_mats=1250), "f64", (24, 14, 1, 3), False)]:
        ns, nm, nt = _counts(s)
        out = {}
        for on in ("1", "0"):
            monkeypatch.setenv("SPIRA_CAM_CONSTS", on)
            out[on], _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=17), prec)
        ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=17), prec)
        assert np.array_equal(out["1"], out["0"]), (ns, prec, fits)
        assert _close(out["1"], ohdr)[0] == 0 and gpu.counters()["segments"] == oseg, (ns, prec)


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


def test_config4_full_size_shard_equals_frame_rows(gpu):
    """BASELINE configs[3] (1920x1080, spp 256, depth 8, 8 GPUs): what rank 3 of 8 renders (rows 3, 11, 19, ...) is,
    bit for bit, those rows of the whole frame rendered on one device; the shard's sample count is 1/8 of the frame's."""
    from spira_hip import distributed as D
    s = scenes.scene_s3()
    ns, nm, nt = _counts(s)
    W, H, spp, depth = 1920, 1080, 256, 8
    full, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=scenes.seed_for(4)), "f32")
    tile = D.tile_params(H, 8, 3)
    part, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=scenes.seed_for(4), **tile), "f32")
    c = gpu.counters()
    rows = D.rows_of_rank(H, 8, 3)
    assert part.shape == (3, len(rows), W) and np.array_equal(part, full[:, rows])
    # closed box: paths run to max_depth, except the ~1e-6 that slip through a wall seam inside t_min
    assert c["samples"] == len(rows) * W * spp and c["samples"] * depth * (1 - 1e-4) < c["segments"] <= c["samples"] * depth


def test_config5_full_size_mesh_properties(gpu):
    """BASELINE configs[4] shape (1920x1080, depth 12, ~82 k triangles through the BVH) at spp 4: wavefront == megakernel bit for bit,
    a row slab equals the frame's rows, every pixel finite and non-negative."""
    s = scenes.scene_s4()
    ns, nm, nt = _counts(s)
    W, H, spp, depth = 1920, 1080, 4, 12
    wf, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=scenes.seed_for(5)), "f32")
    seg = gpu.counters()["segments"]
    mg, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=scenes.seed_for(5), flags=gpu.KERNEL_MEGA), "f32")
    assert np.array_equal(wf, mg) and gpu.counters()["segments"] == seg
    slab, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=scenes.seed_for(5), row0=400, rows=96), "f32")
    assert np.array_equal(slab, wf[:, 400:496])
    assert np.isfinite(wf).all() and wf.min() >= 0 and nt > 80000 and W * H * spp < seg <= W * H * spp * depth


def test_mesh_scene_beyond_128_segments_and_tiny_meshes(gpu, oracle):
    """max_depth > 128: packets cannot carry their own stage any more, so a mesh scene walks the tree in place (no parking, no second launch);
    and the smallest meshes that go through the BVH (33 triangles: a root node with leaf children only).  Both against the oracle's linear scan."""
    s = scenes.scene_s4(level=2)                # 320 triangles
    ns, nm, nt = _counts(s)
    for depth in (129, 200):
        hdr, _ = gpu.render(*_args(s), gpu.make_params(64, 36, 2, depth, ns, nm, nt, seed=6), "f64")
        ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(64, 36, 2, depth, ns, nm, nt, seed=6), "f64")
        assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg, depth
    rng = np.random.default_rng(12)
    from test_gpu_parity import random_scene
    for n_tri in (33, 34, 40, 41, 64, 65):
        t = random_scene(rng, 2, n_tri)
        ns, nm, nt = _counts(t)
        hdr, _ = gpu.render(*_args(t), gpu.make_params(80, 45, 3, 5, ns, nm, nt, seed=9), "f32")
        ohdr, _, oseg = oracle.render(*_args(t), oracle.make_params(80, 45, 3, 5, ns, nm, nt, seed=9), "f32")
        assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg, n_tri
