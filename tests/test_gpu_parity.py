"""GPU (MI355X): parity of the HIP path, called through the C ABI, against the CPU oracle.

Bars (DESIGN.md "Parity"):
  * path GEOMETRY (which object each segment hits, hit distance, ray directions): bit-exact,
    Float32 kernel vs Float32 oracle and Float64 kernel vs Float64 oracle;
  * images: relative 1e-5 (north_star tolerance) — the only difference allowed is the rounding of
    the radiance sum (kernel: L += beta*e iteratively; oracle: the reference's recursion);
  * integer quantities (segment counts) and tiling / batching / kernel-organisation invariances:
    bit-exact.
Nothing here reads /root/reference.
"""
import json
import os

import numpy as np
import pytest

from spira_hip import distributed as D
from spira_hip import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL, ATOL = 1e-5, 1e-6      # |gpu - oracle| <= ATOL + RTOL*|oracle| per pixel and channel


def _args(s):
    return s["spheres5"], s["materials8"], s["triangles10"], s["camera12"]


def _counts(s):
    return len(s["spheres5"]), len(s["materials8"]), 0 if s["triangles10"] is None else len(s["triangles10"])


def _close(a, b, rtol=RTOL, atol=ATOL):
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    bad = np.abs(a64 - b64) > atol + rtol * np.abs(b64)
    return int(bad.sum()), float(np.max(np.abs(a64 - b64) / (np.abs(b64) + atol / rtol)))


def random_scene(rng, n_spheres, n_tris, n_mats=6):
    mats = np.zeros((n_mats, 8))
    mats[:, 0:3] = rng.uniform(0.1, 0.95, (n_mats, 3))
    mats[:, 6] = np.where(rng.random(n_mats) < 0.4, rng.uniform(0.2, 1.0, n_mats), 0.0)
    mats[:, 7] = np.where(rng.random(n_mats) < 0.3, 0.0, rng.uniform(0.05, 1.0, n_mats))
    mats[0, 3:6] = [3, 2.5, 2]
    sph = np.zeros((n_spheres, 5))
    sph[:, 0:3] = rng.uniform(-3, 3, (n_spheres, 3)) + [0, 0, -4]
    sph[:, 3] = rng.uniform(0.1, 0.7, n_spheres)
    sph[:, 4] = rng.integers(1, n_mats + 1, n_spheres)
    tri = None
    if n_tris:
        tri = np.zeros((n_tris, 10))
        base = rng.uniform(-3, 3, (n_tris, 3)) + [0, 0, -5]
        tri[:, 0:3] = base
        tri[:, 3:6] = base + rng.uniform(-1, 1, (n_tris, 3))
        tri[:, 6:9] = base + rng.uniform(-1, 1, (n_tris, 3))
        tri[:, 9] = rng.integers(1, n_mats + 1, n_tris)
    f32 = lambda a: None if a is None else a.astype(np.float32).astype(np.float64)   # exactly representable in both
    return dict(spheres5=f32(sph), materials8=f32(mats), triangles10=f32(tri), camera12=scenes.scene_s1()["camera12"])


# ---------------------------------------------------------------------------------- geometry, bit-exact
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("scene_name", ["s1", "s2", "s3", "random"])
def test_path_geometry_bit_exact(gpu, oracle, prec, scene_name):
    rng = np.random.default_rng(7)
    s = {"s1": scenes.scene_s1, "s2": scenes.scene_s2, "s3": scenes.scene_s3}.get(scene_name, lambda: random_scene(rng, 40, 25))()
    ns, nm, nt = _counts(s)
    W, H, SPP, DEPTH = 320, 180, 8, 8
    n = int(os.environ.get("SPIRA_GEOM_PATHS", "4000"))     # longer one-off campaigns: SPIRA_GEOM_PATHS=200000
    ijs = np.stack([rng.integers(1, W + 1, n), rng.integers(1, H + 1, n), rng.integers(0, SPP, n)], axis=1).astype(np.uint32)
    pg = gpu.make_params(W, H, SPP, DEPTH, ns, nm, nt, seed=42)
    prims, ts, dirs, rad = gpu.trace_paths(*_args(s), pg, ijs, prec)
    po = oracle.make_params(W, H, SPP, DEPTH, ns, nm, nt, seed=42)
    n_seg = 0
    for k in range(n):
        cnt, oprims, ots, odirs, orad = oracle.trace_path(*_args(s), po, int(ijs[k, 0]), int(ijs[k, 1]), int(ijs[k, 2]), prec)
        oprims = np.where(np.arange(DEPTH) < cnt, oprims, -2)
        assert np.array_equal(prims[k], oprims), (k, prims[k], oprims)
        assert np.array_equal(ts[k][:cnt].view(np.uint8), ots[:cnt].view(np.uint8)), (k, ts[k], ots)
        assert np.array_equal(dirs[k][:cnt].view(np.uint8), odirs[:cnt].view(np.uint8)), k
        assert np.allclose(rad[k], orad, rtol=RTOL, atol=ATOL), (k, rad[k], orad)
        n_seg += cnt
    assert n_seg > n   # paths did bounce


# ---------------------------------------------------------------------------------- images vs oracle
CONFIGS = [
    # name, scene, W, H, spp, depth
    ("c1_s2", scenes.scene_s2, 320, 180, 4, 4),        # BASELINE configs[0]: the reference's own CPU case
    ("c2_s1", scenes.scene_s1, 640, 360, 16, 4),       # BASELINE configs[1]
    ("s3_closed", scenes.scene_s3, 256, 144, 8, 8),    # S1 in a closed box: paths run full depth
    ("ragged", scenes.scene_s2, 333, 77, 3, 5),        # sizes that are not multiples of 64 / 256
    ("tiny", scenes.scene_s2, 2, 2, 5, 3),             # smallest legal image
    ("depth1", scenes.scene_s1, 128, 72, 2, 1),
]


@pytest.mark.parametrize("kernel", ["wavefront", "mega"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_image_matches_oracle(gpu, oracle, cfg, prec, kernel):
    name, mk, W, H, spp, depth = cfg
    s = mk()
    ns, nm, nt = _counts(s)
    kflag = gpu.KERNEL_MEGA if kernel == "mega" else gpu.KERNEL_WAVEFRONT
    seed = scenes.seed_for(3)
    hdr, img = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=kflag, seed=seed), prec, want_img=True)
    ohdr, oimg, oseg = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=seed), prec, want_img=True)
    nbad, worst = _close(hdr, ohdr)
    assert nbad == 0, "%s/%s/%s: %d pixel-channels off, worst rel %.3g" % (name, prec, kernel, nbad, worst)
    nbad, worst = _close(img, oimg)
    assert nbad == 0, "display image: %d off, worst %.3g" % (nbad, worst)
    c = gpu.counters()
    assert c["samples"] == W * H * spp and c["segments"] == oseg      # integer work: exact


@pytest.mark.parametrize("fixture", ["c1_s2_320x180_spp4_d4.json", "c2_s1_160x90_spp16_d4.json"])
def test_golden_fixtures(gpu, fixture):
    g = json.load(open(os.path.join(GOLD, fixture)))
    s = scenes.scene_s2() if "s2" in fixture else scenes.scene_s1()
    ns, nm, nt = _counts(s)
    W, H, spp, depth = (320, 180, 4, 4) if "s2" in fixture else (160, 90, 16, 4)
    for prec in ("f32", "f64"):
        hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=g["seed"]), prec)
        assert gpu.counters()["segments"] == g[prec]["segments"]
        got = np.array([hdr[:, y, x] for y, x in g["pixels"]], dtype=np.float64)
        assert np.allclose(got, np.array(g[prec]["values"]), rtol=RTOL, atol=ATOL)
        assert np.allclose(hdr.astype(np.float64).mean(axis=(1, 2)), g[prec]["mean"], rtol=1e-5)


def test_many_primitives_vs_oracle(gpu, oracle):
    rng = np.random.default_rng(3)
    s = random_scene(rng, 600, 300, n_mats=9)      # LDS-resident linear scan near its size limit
    ns, nm, nt = _counts(s)
    W, H, spp, depth = 96, 54, 2, 4
    hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=5), "f32")
    ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=5), "f32")
    assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg


def test_triangles_only_and_no_spheres(gpu, oracle):
    s = scenes.scene_s2()
    s["spheres5"] = np.zeros((0, 5))
    hdr, _ = gpu.render(*_args(s), gpu.make_params(64, 36, 4, 3, 0, 6, 1, seed=2), "f32")
    ohdr, _, _ = oracle.render(*_args(s), oracle.make_params(64, 36, 4, 3, 0, 6, 1, seed=2), "f32")
    assert _close(hdr, ohdr)[0] == 0


def test_dense_continuation_threshold_does_not_change_results(gpu):
    """k_path keeps a sub-chunk's hits in registers when at least SPIRA_DENSE_PCT % of its scattered rays hit again (default 90), else
    compacts them into the queue.  With 50 % the queues hold hits of different segments side by side (packets carry their stage), with 0
    every hit goes through the queue, with 100 only fully dense sub-chunks stay in registers: same pixels, same segment counts."""
    import os
    rng = np.random.default_rng(21)
    cases = [(scenes.scene_s1(), 161, 91, 6, 8), (scenes.scene_s3(), 96, 54, 4, 9), (scenes.scene_s4(level=3), 120, 68, 3, 12), (random_scene(rng, 30, 20), 97, 55, 5, 7)]
    old = os.environ.get("SPIRA_DENSE_PCT")
    try:
        for s, W, H, spp, depth in cases:
            ns, nm, nt = _counts(s)
            ref, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.KERNEL_MEGA, seed=3), "f32")
            seg = gpu.counters()["segments"]
            enq = {}
            for pct in ("0", "50", "90", "100"):
                os.environ["SPIRA_DENSE_PCT"] = pct
                got, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.KERNEL_WAVEFRONT, seed=3, batch_rays=30000), "f32")
                c = gpu.counters()
                assert np.array_equal(ref, got) and c["segments"] == seg, pct
                enq[pct] = c["rays_enqueued"]
            assert enq["0"] >= enq["100"] >= enq["90"] >= enq["50"] and enq["0"] > enq["50"]     # a lower threshold keeps more hits in registers
    finally:
        if old is None:
            os.environ.pop("SPIRA_DENSE_PCT", None)
        else:
            os.environ["SPIRA_DENSE_PCT"] = old


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_empty_scene_is_the_sky(gpu, oracle, prec):
    """No sphere, no triangle: every camera ray misses; all three organisations, against the oracle."""
    s = scenes.scene_s1()
    sph, mats, cam = np.zeros((0, 5)), s["materials8"][:1], s["camera12"]
    ohdr, _, oseg = oracle.render(sph, mats, None, cam, oracle.make_params(70, 40, 3, 5, 0, 1, 0, seed=4), prec)
    assert oseg == 70 * 40 * 3
    for k in (gpu.KERNEL_DEFAULT, gpu.KERNEL_WAVEFRONT, gpu.KERNEL_BOUNCE, gpu.KERNEL_MEGA):
        hdr, _ = gpu.render(sph, mats, None, cam, gpu.make_params(70, 40, 3, 5, 0, 1, 0, flags=k, seed=4), prec)
        assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg and gpu.counters()["rays_enqueued"] == 0, k


def test_max_depth_zero_is_black(gpu):
    s = scenes.scene_s1()
    hdr, img = gpu.render(*_args(s), gpu.make_params(32, 18, 2, 0, 5, 5, seed=1), "f32", want_img=True)
    assert not hdr.any() and not img.any()


# ---------------------------------------------------------------------------------- invariances, bit-exact
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_mega_equals_wavefront_bitwise(gpu, prec):
    s = scenes.scene_s2()
    a, _ = gpu.render(*_args(s), gpu.make_params(200, 120, 6, 6, 5, 6, 1, flags=gpu.KERNEL_WAVEFRONT, seed=8), prec)
    seg = gpu.counters()["segments"]
    b, _ = gpu.render(*_args(s), gpu.make_params(200, 120, 6, 6, 5, 6, 1, flags=gpu.KERNEL_MEGA, seed=8), prec)
    assert np.array_equal(a, b) and gpu.counters()["segments"] == seg
    c, _ = gpu.render(*_args(s), gpu.make_params(200, 120, 6, 6, 5, 6, 1, flags=gpu.KERNEL_BOUNCE, seed=8), prec)     # round-1 organisation
    assert np.array_equal(a, c) and gpu.counters()["segments"] == seg


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("scene_name", ["s1", "s3", "s4"])
def test_three_kernel_organisations_agree_bitwise(gpu, prec, scene_name):
    """persistent hit-queue kernel (default) == per-bounce ray queues (round 1) == megakernel, pixel for pixel and segment for
    segment, over max_depth 1..9, odd sizes, several passes (batch_rays) and R = 1 / 2 rays per lane."""
    s = {"s1": scenes.scene_s1, "s3": scenes.scene_s3, "s4": lambda: scenes.scene_s4(level=3)}[scene_name]()
    ns, nm, nt = _counts(s)
    for (W, H, spp, depth, batch) in [(97, 61, 5, 1, 0), (97, 61, 5, 2, 0), (160, 90, 7, 3, 20000), (131, 77, 9, 9, 0), (64, 36, 33, 6, 5000)]:
        ref, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.KERNEL_MEGA, seed=17), prec)
        seg = gpu.counters()["segments"]
        for k in (gpu.KERNEL_WAVEFRONT, gpu.KERNEL_BOUNCE):
            got, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=k, seed=17, batch_rays=batch), prec)
            assert np.array_equal(ref, got) and gpu.counters()["segments"] == seg, (W, H, spp, depth, batch, k)


def test_batch_size_does_not_change_results(gpu):
    s = scenes.scene_s1()
    ref, _ = gpu.render(*_args(s), gpu.make_params(160, 90, 7, 5, 5, 5, seed=4), "f32")
    for batch in (160 * 90, 160 * 90 * 2, 160 * 90 * 3 + 17, 1 << 24):   # 1, 2, 3 slots (ragged last pass), all-in-one
        got, _ = gpu.render(*_args(s), gpu.make_params(160, 90, 7, 5, 5, 5, seed=4, batch_rays=batch), "f32")
        assert np.array_equal(got, ref), batch


def test_tiling_reproduces_full_image_bitwise(gpu):
    s = scenes.scene_s2()
    W, H = 192, 108
    full, fimg = gpu.render(*_args(s), gpu.make_params(W, H, 4, 5, 5, 6, 1, seed=6), "f32", want_img=True)
    slab, _ = gpu.render(*_args(s), gpu.make_params(W, H, 4, 5, 5, 6, 1, seed=6, row0=40, rows=13), "f32")
    assert np.array_equal(slab, full[:, 40:53])
    for world, sh in [(8, 8), (3, 5), (2, 64)]:
        tiles = []
        for r in range(world):
            t, _ = gpu.render(*_args(s), gpu.make_params(W, H, 4, 5, 5, 6, 1, seed=6, **D.tile_params(H, world, r, sh)), "f32")
            tiles.append(t)
        mr = D.max_rows(H, world, sh)
        padded = [np.concatenate([t, np.zeros((3, mr - t.shape[1], W), np.float32)], axis=1) for t in tiles]
        assert np.array_equal(D.assemble(padded, H, world, sh), full)
    up, _ = gpu.render(*_args(s), gpu.make_params(W, H, 4, 5, 5, 6, 1, flags=gpu.ROWS_BOTTOM_UP, seed=6), "f32")
    assert np.array_equal(up, full[:, ::-1])


def test_repeatable_and_seed_sensitive(gpu):
    s = scenes.scene_s1()
    a, _ = gpu.render(*_args(s), gpu.make_params(128, 72, 4, 6, 5, 5, seed=1), "f32")
    b, _ = gpu.render(*_args(s), gpu.make_params(128, 72, 4, 6, 5, 5, seed=1), "f32")
    c, _ = gpu.render(*_args(s), gpu.make_params(128, 72, 4, 6, 5, 5, seed=2), "f32")
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_device_pointer_entry_and_gather(gpu):
    """spira_render_device_f32 into torch tensors on torch's stream + the gather path at world size 1."""
    import torch
    s = scenes.scene_s1()
    W, H = 160, 90
    p = gpu.make_params(W, H, 4, 4, 5, 5, seed=12)
    ref, _ = gpu.render(*_args(s), p, "f32")
    out = torch.empty((3, H, W), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream()
    gpu.render_device(*_args(s), p, out.data_ptr(), 0, st.cuda_stream, "f32")
    st.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


# ---------------------------------------------------------------------------------- full-size properties
FULL = dict(W=1920, H=1080, spp=64, depth=8)     # BASELINE.json metric configuration


def test_full_size_furnace_closed_form(gpu):
    """Size-independent property at the headline size: inside a closed scene whose surfaces all have
    diffuse albedo rho and emission e, every path contributes e * sum_{k<depth} (0.5*rho)^k whatever it
    hits (semantics A attenuates by 0.5*diffuse per bounce, examples/julia-raytracer.jl:360).  With
    rho = 0.5, e = 0.75 every operation is exact in binary floating point, so pixels must EQUAL the closed
    form.  The only exception the reference's semantics allow: a path born within t_min = 0.001 of a wall
    junction passes through that wall (`t < t_min` rejects, :179) and sees the sky — a few paths in 10^7."""
    s = scenes.scene_s3()
    rho, e = 0.5, 0.75
    s["materials8"] = np.tile(np.array([[rho, rho, rho, e, e, e, 0.0, 1.0]]), (len(s["materials8"]), 1))
    ns, nm, nt = _counts(s)
    W, H, spp, depth = FULL["W"], FULL["H"], FULL["spp"], FULL["depth"]
    hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=scenes.seed_for(3)), "f32")
    want = np.float32(e * sum((0.5 * rho) ** k for k in range(depth)))
    off = (hdr != want)       # a pixel is off if ANY of its 64 paths leaked: ~64 x 5e-6
    assert off.mean() < 2e-3 and np.abs(hdr - want).max() < 0.02, (off.mean(), hdr.min(), hdr.max(), want)
    c = gpu.counters()
    n = W * H * spp
    assert c["samples"] == n and n * depth * (1 - 1e-5) <= c["segments"] <= n * depth


def test_full_size_linearity_and_tiling_checksum(gpu):
    """1080p (spp reduced to bound test time).  Closed scene (no sky term): emission x2 => image x2 EXACTLY
    (scaling by a power of two commutes with every rounding).  Open scene: the 8-way stripe tiling of the
    multi-GPU path reproduces the untiled image checksum."""
    import zlib
    W, H, depth = FULL["W"], FULL["H"], FULL["depth"]
    s3 = scenes.scene_s3()
    n3, m3, t3 = _counts(s3)
    c1, _ = gpu.render(*_args(s3), gpu.make_params(W, H, 2, depth, n3, m3, t3, seed=78), "f32")
    s3b = dict(s3)
    s3b["materials8"] = s3["materials8"].copy()
    s3b["materials8"][:, 3:6] *= 2.0
    c2, _ = gpu.render(*_args(s3b), gpu.make_params(W, H, 2, depth, n3, m3, t3, seed=78), "f32")
    crack = (c2 != 2.0 * c1)          # paths that leaked to the sky through a t_min crack (see furnace test)
    assert crack.mean() < 1e-4 and c1.max() > 0 and np.isfinite(c1).all()
    s = scenes.scene_s1()
    ns, nm, _ = _counts(s)
    a, _ = gpu.render(*_args(s), gpu.make_params(W, H, 8, depth, ns, nm, seed=77), "f32")
    tiles = [gpu.render(*_args(s), gpu.make_params(W, H, 8, depth, ns, nm, seed=77, **D.tile_params(H, 8, r, 8)), "f32")[0]
             for r in range(8)]
    assert zlib.crc32(D.assemble(tiles, H, 8, 8).tobytes()) == zlib.crc32(a.tobytes())


def test_f32_kernel_statistically_matches_f64_oracle(gpu, oracle):
    """Float32 kernels vs the Float64 restatement (the reference's own precision): same RNG stream, so the
    images agree closely except where a rounding flips a hit/miss; check the mean image and the fraction."""
    s = scenes.scene_s2()
    W, H, spp, depth = 160, 90, 64, 8
    hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, 5, 6, 1, seed=21), "f32")
    ohdr, _, _ = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, 5, 6, 1, seed=21), "f64")
    rel = np.abs(hdr.astype(np.float64) - ohdr) / (np.abs(ohdr) + 1e-3)
    assert np.median(rel) < 1e-6 and (rel > 1e-3).mean() < 2e-3
    assert abs(hdr.astype(np.float64).mean() / ohdr.mean() - 1) < 1e-4
