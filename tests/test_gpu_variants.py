"""GPU (MI355X): the two secondary integrator variants of the reference against their CPU restatements.
SPIRA_SEM_CPU   = trace_ray of render_with_cpu  (src/spira-metal-optimized.jl:1346-1450)
SPIRA_SEM_METAL = path_trace                     (src/spira_path_trace_kernel.metal:140-269)
Geometry bit-exact (incl. the shared polynomial sin/cos and the per-pixel LCG streams), images 1e-5."""
import numpy as np
import pytest

from spira_hip import scenes
from test_gpu_parity import _close

pytestmark = pytest.mark.gpu
SEMS = {"cpu": 0x1, "metal": 0x2}


def _scene():
    s = scenes.scene_s1()
    return s["spheres5"], s["materials8"], s["camera12"]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("sem", ["cpu", "metal"])
def test_variant_geometry_bit_exact(gpu, oracle, sem, prec):
    sp, ma, cam = _scene()
    rng = np.random.default_rng(31)
    W, H, SPP, DEPTH, n = 200, 120, 6, 8, 3000
    ijs = np.stack([rng.integers(1, W + 1, n), rng.integers(1, H + 1, n), rng.integers(0, SPP, n)], axis=1).astype(np.uint32)
    pg = gpu.make_params(W, H, SPP, DEPTH, 5, 5, 0, flags=SEMS[sem], seed=77)
    prims, ts, dirs, rad = gpu.trace_paths(sp, ma, None, cam, pg, ijs, prec)
    po = oracle.make_params(W, H, SPP, DEPTH, 5, 5, 0, flags=SEMS[sem], seed=77)
    deep = 0
    for k in range(n):
        cnt, oprims, ots, odirs, orad = oracle.trace_path_variant(sp, ma, cam, po, int(ijs[k, 0]), int(ijs[k, 1]), int(ijs[k, 2]), prec)
        oprims = np.where(np.arange(DEPTH) < cnt, oprims, -2)
        assert np.array_equal(prims[k], oprims), (k, prims[k], oprims)
        assert np.array_equal(ts[k][:cnt].view(np.uint8), ots[:cnt].view(np.uint8)), (k, ts[k], ots)
        assert np.array_equal(dirs[k][:cnt].view(np.uint8), odirs[:cnt].view(np.uint8)), k
        assert np.allclose(rad[k], orad, rtol=1e-5, atol=1e-6), (k, rad[k], orad)
        deep += cnt >= 5
    assert deep > 20       # long paths exist (METAL: they passed Russian roulette)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("sem", ["cpu", "metal"])
def test_variant_image_matches_oracle(gpu, oracle, sem, prec):
    sp, ma, cam = _scene()
    post = gpu.POST_CLAMP_GAMMA if sem == "cpu" else gpu.POST_NONE       # render_with_cpu: clamp + sqrt (:1441-1442)
    W, H, spp, depth = 240, 135, 8, 8
    hdr, img = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, spp, depth, 5, 5, 0, flags=SEMS[sem] | post, seed=5), prec, want_img=True)
    ohdr, oimg, oseg = oracle.render_variant(sp, ma, cam, oracle.make_params(W, H, spp, depth, 5, 5, 0, flags=SEMS[sem] | post, seed=5), prec, want_img=True)
    assert _close(hdr, ohdr)[0] == 0 and _close(img, oimg)[0] == 0
    c = gpu.counters()
    assert c["samples"] == W * H * spp and c["segments"] == oseg


@pytest.mark.parametrize("sem", ["cpu", "metal"])
def test_variant_tiling_and_determinism(gpu, sem):
    from spira_hip import distributed as D
    sp, ma, cam = _scene()
    W, H = 160, 90
    full, _ = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 5, 6, 5, 5, 0, flags=SEMS[sem], seed=9), "f32")
    again, _ = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 5, 6, 5, 5, 0, flags=SEMS[sem], seed=9), "f32")
    assert np.array_equal(full, again)
    tiles = [gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 5, 6, 5, 5, 0, flags=SEMS[sem], seed=9, **D.tile_params(H, 3, r, 4)), "f32")[0]
             for r in range(3)]
    mr = D.max_rows(H, 3, 4)
    padded = [np.concatenate([t, np.zeros((3, mr - t.shape[1], W), np.float32)], axis=1) for t in tiles]
    assert np.array_equal(D.assemble(padded, H, 3, 4), full)


def test_variants_reject_triangles(gpu):
    s = scenes.scene_s2()
    with pytest.raises(gpu.SpiraError) as e:
        gpu.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"], gpu.make_params(32, 18, 1, 2, 5, 6, 1, flags=0x2))
    assert "error -5" in str(e.value)


def test_variants_agree_in_the_mean(gpu):
    """The three estimators differ (lobe choice, attenuation, RR) but see the same scene: on the sky rows they must
    agree closely, and METAL's Russian roulette must not bias the image vs. a deeper cut-off."""
    sp, ma, cam = _scene()
    W, H = 160, 90
    imgs = {k: gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 64, 8, 5, 5, 0, flags=v, seed=3), "f32")[0] for k, v in [("a", 0), ("cpu", 1), ("metal", 2)]}
    for k in ("cpu", "metal"):
        # (not identical: METAL maps pixels with (x + xi)/W, the others with (i - 1 + xi)/(W - 1))
        assert np.allclose(imgs[k][:, :10].mean(axis=(1, 2)), imgs["a"][:, :10].mean(axis=(1, 2)), rtol=1e-2)
    deep = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 64, 16, 5, 5, 0, flags=2, seed=4), "f32")[0]
    assert abs(deep.mean() / imgs["metal"].mean() - 1) < 0.03


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_metal_wavefront_equals_one_lane_per_pixel_bitwise(gpu, oracle, prec):
    """N2 of VERDICT r1: the .metal estimator (Russian roulette, per-pixel LCG carried from sample to sample) in the wavefront
    organisation (k_path_metal: hit queues, ballot/popcount compaction, LCG state in the packet) against the one-lane-per-pixel
    kernel (k_variant_metal): identical images, segment counts and final LCG states; and against the oracle."""
    rng = np.random.default_rng(11)
    from test_gpu_parity import random_scene
    for s in (scenes.scene_s1(), random_scene(rng, 60, 0, n_mats=8)):
        sp, ma, cam = s["spheres5"], s["materials8"], s["camera12"]
        ns, nm = len(sp), len(ma)
        npdt = np.float32 if prec == "f32" else np.float64
        for (W, H, spp, depth) in [(131, 77, 7, 8), (64, 36, 3, 1), (200, 113, 4, 12)]:
            res = {}
            for name, k in (("wave", gpu.KERNEL_WAVEFRONT), ("lane", gpu.KERNEL_MEGA)):
                sums = np.zeros((3, H, W), dtype=npdt)
                states = np.zeros(H * W, dtype=np.uint32)
                gpu.accumulate(sp, ma, None, cam, gpu.make_params(W, H, spp, depth, ns, nm, 0, flags=gpu.SEM_METAL | k, seed=19), 0, sums, states, prec)
                res[name] = (sums, states, gpu.counters()["segments"])
            assert np.array_equal(res["wave"][0], res["lane"][0]) and np.array_equal(res["wave"][1], res["lane"][1])
            assert res["wave"][2] == res["lane"][2]
            ohdr, _, oseg = oracle.render_variant(sp, ma, cam, oracle.make_params(W, H, spp, depth, ns, nm, 0, flags=gpu.SEM_METAL | gpu.POST_NONE, seed=19), prec)
            assert _close(res["wave"][0] / npdt(spp), ohdr)[0] == 0 and res["wave"][2] == oseg


def test_metal_wavefront_progressive_and_full_size(gpu):
    """k calls of n samples == one call (states carried), in wavefront form; and the 1080p spp 64 depth 8 frame runs and equals
    the one-lane-per-pixel kernel on a checksum of row slabs."""
    s = scenes.scene_s1()
    sp, ma, cam = s["spheres5"], s["materials8"], s["camera12"]
    W, H = 160, 90
    one = np.zeros((3, H, W), np.float32); st1 = np.zeros(H * W, np.uint32)
    gpu.accumulate(sp, ma, None, cam, gpu.make_params(W, H, 12, 8, 5, 5, 0, flags=gpu.SEM_METAL, seed=2), 0, one, st1, "f32")
    acc = np.zeros((3, H, W), np.float32); st = np.zeros(H * W, np.uint32)
    s0 = 0
    for n in (5, 1, 6):
        gpu.accumulate(sp, ma, None, cam, gpu.make_params(W, H, n, 8, 5, 5, 0, flags=gpu.SEM_METAL, seed=2), s0, acc, st, "f32")
        s0 += n
    assert np.array_equal(acc, one) and np.array_equal(st, st1)
    W, H = 1920, 1080
    a, _ = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 64, 8, 5, 5, 0, flags=gpu.SEM_METAL | gpu.POST_NONE, seed=7), "f32")
    seg = gpu.counters()["segments"]
    b, _ = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, 64, 8, 5, 5, 0, flags=gpu.SEM_METAL | gpu.KERNEL_MEGA | gpu.POST_NONE, seed=7), "f32")
    assert np.array_equal(a, b) and gpu.counters()["segments"] == seg and np.isfinite(a).all()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_hybrid_estimator_equals_the_oracle_bitwise(gpu, oracle, prec):
    """SPIRA_SEM_HYBRID = render_hybrid_gpu as written (src/spira-metal-optimized.jl:1228-1343): per-pixel xorshift32 jitter, image-wide lock step,
    last bounce shaded, tone map per sample.  Same statements in the same order in oracle and kernel (k_hybrid), no reduction whose order could differ:
    the images are the same BITS, and so are the intersection counts; both row orders; a scene nobody hits stays black (the image-wide break)."""
    sp, ma, cam = _scene()
    for (W, H, spp, depth, extra) in ((97, 55, 5, 4, 0), (64, 36, 3, 7, 0x1000), (33, 20, 2, 1, 0)):
        pg = gpu.make_params(W, H, spp, depth, 5, 5, 0, flags=0x3 | extra, seed=21)
        hdr, img = gpu.render(sp, ma, None, cam, pg, prec, want_hdr=True, want_img=True)
        want, seg = oracle.render_hybrid(sp, ma, cam, oracle.make_params(W, H, spp, depth, 5, 5, 0, flags=0x3 | extra, seed=21), prec)
        assert np.array_equal(hdr, want) and np.array_equal(img, want), (W, H, float(np.abs(hdr - want).max()))
        assert gpu.counters()["segments"] == seg == W * H * spp * depth
    empty, _ = gpu.render(None, ma, None, cam, gpu.make_params(32, 18, 2, 3, 0, 5, 0, flags=0x3, seed=1), prec)
    assert float(np.abs(empty).max()) == 0.0 and gpu.counters()["segments"] == 32 * 18 * 2


def test_hybrid_estimator_limits_and_mirror(gpu, oracle):
    sp, ma, cam = _scene()
    with pytest.raises(gpu.SpiraError):          # whole images only: the lock step needs every pixel
        gpu.render(sp, ma, None, cam, gpu.make_params(32, 18, 2, 3, 5, 5, 0, flags=0x3, seed=1, rows=8), "f32")
    with pytest.raises(gpu.SpiraError):          # spheres only, like its source
        tri = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0, 1]], dtype=np.float64)
        gpu.render(sp, ma, tri, cam, gpu.make_params(32, 18, 2, 3, 5, 5, 1, flags=0x3, seed=1), "f32")
    with pytest.raises(gpu.SpiraError):          # no accumulate entry
        gpu.accumulate(sp, ma, None, cam, gpu.make_params(32, 18, 2, 3, 5, 5, 0, flags=0x3, seed=1), 0, np.zeros((3, 18, 32), np.float32), prec="f32")
    # the package surface: render_hybrid_gpu(...; semantics = "hybrid") of the Python mirror (SPIRA.jl: semantics = :hybrid)
    from spira_hip import spira as S
    scene = S.create_scene()
    camera = S.Camera(S.Point3(0, 1, 3), S.Point3(0, 0, 0), S.Vec3(0, 1, 0), 40.0, 16 / 9)
    img = S.render_hybrid_gpu(48, 27, scene, camera, samples_per_pixel=3, max_depth=4, seed=6, semantics="hybrid")
    sd, md = S.prepare_scene_data(scene)
    want, _ = oracle.render_hybrid(sd.reshape(-1, 5), md.reshape(-1, 8), camera.flat(), oracle.make_params(48, 27, 3, 4, 5, 5, 0, flags=0x3, seed=6), "f32")
    assert img.shape == (27, 48, 3) and np.array_equal(np.moveaxis(img, -1, 0), want)


def test_hybrid_estimator_fuzz(gpu, oracle):
    """Random sphere scenes (metals, rough metals, diffuse, emitters), image sizes, spp, depths, precisions, row orders: SPIRA_SEM_HYBRID == its oracle, bit for bit."""
    from test_gpu_parity import random_scene
    rng = np.random.default_rng(20261006)
    for it in range(24):
        s = random_scene(rng, int(rng.integers(1, 12)), 0)
        ns, nm = len(s["spheres5"]), len(s["materials8"])
        W, H, spp, depth = int(rng.integers(2, 80)), int(rng.integers(2, 50)), int(rng.integers(1, 6)), int(rng.integers(0, 9))
        prec = "f32" if rng.random() < 0.5 else "f64"
        fl = 0x3 | (0x1000 if rng.random() < 0.3 else 0) | int(rng.choice([0x00, 0x10, 0x30]))          # (the organisation flags are ignored)
        seed = int(rng.integers(0, 2 ** 40))
        hdr, _ = gpu.render(s["spheres5"], s["materials8"], None, s["camera12"], gpu.make_params(W, H, spp, depth, ns, nm, 0, flags=fl, seed=seed), prec)
        want, seg = oracle.render_hybrid(s["spheres5"], s["materials8"], s["camera12"], oracle.make_params(W, H, spp, depth, ns, nm, 0, flags=fl, seed=seed), prec)
        assert np.array_equal(hdr, want), (it, W, H, spp, depth, prec, float(np.abs(hdr - want).max()))
        assert gpu.counters()["segments"] == seg, it
