"""GPU (MI355X): the BASELINE.json configurations at their FULL sizes.

Where the oracle cannot render a whole 1080p frame in seconds it renders a row slab / an interleaved stripe of it at the
full spp (tiling is bit-invariant: the RNG is keyed by the global pixel), and the same rows of the HIP full-frame render
must match it to the north-star tolerance, with identical segment counts.  Float64 is the bench dtype, Float32 beside it.

  configs[2]  1920x1080 spp 64  depth 8   S1 (headline) and S3 (closed box)
  configs[3]  1920x1080 spp 256 depth 8   S3, the 8-GPU config: rank 5's interleaved stripes
  configs[4]  1920x1080 spp 64  depth 12  S4 (81 920-triangle mesh through the BVH; stand-in for the bunny)
"""
import numpy as np
import pytest

from spira_hip import distributed as D
from spira_hip import scenes
from test_gpu_parity import _args, _close, _counts

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


def _slab_vs_oracle(gpu, oracle, s, spp, depth, seed, prec, tile, full):
    """`tile` = tiling fields; the HIP tile render must equal the full frame's rows bit for bit, the oracle's tile render
    must match it to tolerance with the same number of segments."""
    ns, nm, nt = _counts(s)
    part, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=seed, **tile), prec)
    seg = gpu.counters()["segments"]
    if tile.get("stripe_count", 0) > 1:
        rows = D.rows_of_rank(H, tile["stripe_count"], tile["stripe_rank"], tile["stripe_h"])[:part.shape[1]]
    else:
        rows = list(range(tile["row0"], tile["row0"] + tile["rows"]))
    assert np.array_equal(part, full[:, rows])
    ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=seed, **tile), prec)
    nbad, worst = _close(part, ohdr)
    assert nbad == 0, (prec, tile, nbad, worst)
    assert seg == oseg, (seg, oseg)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("scene_name", ["s1", "s3"])
def test_config3_full_size_rows_match_oracle(gpu, oracle, scene_name, prec):
    """BASELINE configs[2] = the bench workload: 1080p, spp 64, depth 8, in the bench dtype (f64) and f32."""
    s = {"s1": scenes.scene_s1, "s3": scenes.scene_s3}[scene_name]()
    ns, nm, nt = _counts(s)
    seed = scenes.seed_for(3)
    full, _ = gpu.render(*_args(s), gpu.make_params(W, H, 64, 8, ns, nm, nt, seed=seed), prec)
    c = gpu.counters()
    # (one pass for the whole frame on the device; a host-output frame of this size is rendered as 4 (f64 HDR, 49.8 MB) / 2 (f32) row slabs of one pass each)
    assert c["samples"] == W * H * 64 and c["passes"] == (4 if prec == "f64" else 2) and np.isfinite(full).all() and full.min() >= 0
    # 8 rows through the spheres (rows 600..607 of the top-based image), and an INTERLEAVED tile: rank 77 of 135 with 2-row
    # stripes = rows {154,155, 424,425, 694,695, 964,965} (sky, spheres, ground)
    _slab_vs_oracle(gpu, oracle, s, 64, 8, seed, prec, dict(row0=600, rows=8), full)
    _slab_vs_oracle(gpu, oracle, s, 64, 8, seed, prec, D.tile_params(H, 135, 77, 2), full)
    if scene_name == "s1":
        _slab_vs_oracle(gpu, oracle, s, 64, 8, seed, prec, dict(row0=0, rows=4), full)        # sky rows: paths of one segment


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_config4_full_size_stripe_matches_oracle(gpu, oracle, prec):
    """BASELINE configs[3]: 1080p, spp 256, depth 8, sharded 8 ways — rank 5's tile on the GPU equals those rows of the frame
    rendered whole, and a 4-row slab of it matches the oracle at the full spp 256."""
    s = scenes.scene_s3()
    ns, nm, nt = _counts(s)
    seed = scenes.seed_for(4)
    full, _ = gpu.render(*_args(s), gpu.make_params(W, H, 256, 8, ns, nm, nt, seed=seed), prec)
    tile = D.tile_params(H, 8, 5)
    part, _ = gpu.render(*_args(s), gpu.make_params(W, H, 256, 8, ns, nm, nt, seed=seed, **tile), prec)
    assert np.array_equal(part, full[:, D.rows_of_rank(H, 8, 5)])
    _slab_vs_oracle(gpu, oracle, s, 256, 8, seed, prec, dict(row0=540, rows=4), full)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_config5_full_size_spp64(gpu, prec):
    """BASELINE configs[4] at its real size: 81 920 triangles, 1080p, spp 64, depth 12.  Wavefront == megakernel bit for bit,
    the 8-way stripe tiling reassembles to the frame, every pixel finite and non-negative, segment bounds."""
    s = scenes.scene_s4()
    ns, nm, nt = _counts(s)
    seed, spp, depth = scenes.seed_for(5), 64, 12
    assert nt == 81920
    wf, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=seed), prec)
    c = gpu.counters()
    assert c["samples"] == W * H * spp and W * H * spp < c["segments"] <= W * H * spp * depth
    mg, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=seed, flags=gpu.KERNEL_MEGA), prec)
    assert np.array_equal(wf, mg) and gpu.counters()["segments"] == c["segments"]
    del mg
    assert np.isfinite(wf).all() and wf.min() >= 0
    tiles, seg = [], 0
    for r in range(8):
        t, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=seed, **D.tile_params(H, 8, r)), prec)
        seg += gpu.counters()["segments"]
        tiles.append(t)
    assert np.array_equal(D.assemble(tiles, H, 8), wf) and seg == c["segments"]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_config5_full_mesh_geometry_vs_linear_scan(gpu, oracle, prec):
    """The 81 920-triangle tree against the oracle's linear scan over all 81 920 triangles (like the reference's
    closest-hit loop, examples/julia-raytracer.jl:242-258): per-segment object, distance and direction, bit for bit."""
    s = scenes.scene_s4()
    ns, nm, nt = _counts(s)
    rng = np.random.default_rng(23)
    n, spp, depth, seed = 2400, 4, 12, scenes.seed_for(5)
    # the mesh projects to i in [881, 1040], j in [470, 608] of the 1080p frame; aim 3/4 of the paths at it
    ij_mesh = np.stack([rng.integers(875, 1046, n * 3 // 4), rng.integers(465, 613, n * 3 // 4)], axis=1)
    ij_any = np.stack([rng.integers(1, W + 1, n - len(ij_mesh)), rng.integers(1, H + 1, n - len(ij_mesh))], axis=1)
    ijs = np.concatenate([np.concatenate([ij_mesh, ij_any]), rng.integers(0, spp, (n, 1))], axis=1).astype(np.uint32)
    prims, ts, dirs, rad = gpu.trace_paths(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=seed), ijs, prec)
    po = oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=seed)
    tri_hits = 0
    for k in range(n):
        cnt, oprims, ots, odirs, orad = oracle.trace_path(*_args(s), po, int(ijs[k, 0]), int(ijs[k, 1]), int(ijs[k, 2]), prec)
        oprims = np.where(np.arange(depth) < cnt, oprims, -2)
        assert np.array_equal(prims[k], oprims), (k, prims[k], oprims)
        assert np.array_equal(ts[k][:cnt].view(np.uint8), ots[:cnt].view(np.uint8)), (k, ts[k], ots)
        assert np.array_equal(dirs[k][:cnt].view(np.uint8), odirs[:cnt].view(np.uint8)), k
        tri_hits += int((oprims >= ns).sum())
    assert tri_hits > 800, tri_hits


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_config5_slab_matches_oracle_image(gpu, oracle, prec):
    """An 8-row slab through the 81 920-triangle mesh at 1080p, depth 12 (spp 2: the oracle tests every triangle for every
    segment, ~5e9 triangle tests), GPU BVH vs the oracle's linear scan: image to tolerance, identical segment count."""
    s = scenes.scene_s4()
    ns, nm, nt = _counts(s)
    seed = scenes.seed_for(5)
    tile = dict(row0=536, rows=8)
    part, _ = gpu.render(*_args(s), gpu.make_params(W, H, 2, 12, ns, nm, nt, seed=seed, **tile), prec)
    seg = gpu.counters()["segments"]
    ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(W, H, 2, 12, ns, nm, nt, seed=seed, **tile), prec)
    assert _close(part, ohdr)[0] == 0 and seg == oseg
