"""CPU: the oracle's restatement of SPIRA_SEM_HYBRID = render_hybrid_gpu as written (src/spira-metal-optimized.jl:1228-1343) against
answers that follow from the cited statements alone.  PARITY UNPINNED like the rest of the oracle (the reference cannot run here and
seeds nothing): what is checked is the restatement's own arithmetic and control flow."""
import numpy as np
import pytest

SEM_HYBRID = 0x3


def _aces_sqrt32(x):
    x = np.float32(x)
    a, b, c, d, e = (np.float32(v) for v in (2.51, 0.03, 2.43, 0.59, 0.14))
    r = (x * (a * x + b)) / (x * (c * x + d) + e)                        # :1133-1135, Float32 operation by operation
    return np.sqrt(np.clip(r, np.float32(0), np.float32(1)))              # :1136-1143


def _cam(binding):
    return binding.camera_lookat([0, 0, 3], [0, 0, 0], [0, 1, 0], 40.0, np.float32(16 / 9), prec="f32").astype(np.float64)


def test_empty_scene_ends_every_sample_at_depth_one(oracle, binding):
    """`if sum(hit_results[:, 1]) == 0 break` (:1303): no ray of the image hits anything -> the sample ends before depth 1 is shaded and adds
    nothing (depth < max_depth) — the reference's own picture of an empty scene is black, not sky."""
    mats = np.zeros((1, 8))
    img, seg = oracle.render_hybrid(None, mats, _cam(binding), oracle.make_params(16, 9, 3, 4, 0, 1, 0, flags=SEM_HYBRID, seed=1), "f32")
    assert float(np.abs(img).max()) == 0.0 and seg == 16 * 9 * 3            # one intersection pass per sample, then the break


@pytest.mark.parametrize("depth,metallic", [(1, 0.0), (3, 1.0), (6, 1.0)])
def test_sphere_around_the_camera_gives_the_last_bounce_colour(oracle, binding, depth, metallic):
    """A sphere around the camera: every ray hits its inside.  Its normals point outward (:782-790, never flipped), so a DIFFUSE bounce (:966-979)
    leaves the sphere, nothing is hit at depth 2 and the image-wide break (:1303) ends every sample unshaded: with max_depth = 1 the picture is K6's
    albedo * 0.5 + emission (:1093-1095), with max_depth > 1 it is black.  A MIRROR (:908-910 with d . n > 0) reflects inward for ever: every pixel of
    every sample is albedo * 0.5^max_depth + emission through K7 (:1128-1144)."""
    spheres = np.array([[0, 0, 3, 50.0, 1]], dtype=np.float64)               # the camera sits at its centre
    mats = np.array([[0.8, 0.4, 0.2, 0.5, 0.25, 0.0, metallic, 0.0 if metallic else 1.0]])
    img, seg = oracle.render_hybrid(spheres, mats, _cam(binding), oracle.make_params(12, 7, 5, depth, 1, 1, 0, flags=SEM_HYBRID, seed=9), "f32")
    contrib = np.float32(0.5) ** depth
    want = [_aces_sqrt32(np.float32(al) * contrib + np.float32(em)) for al, em in ((0.8, 0.5), (0.4, 0.25), (0.2, 0.0))]
    for ch in range(3):
        # five identical Float32 terms summed and divided by 5: exact up to the rounding of the sum
        assert np.allclose(img[ch], want[ch], rtol=3e-7, atol=0), (ch, img[ch].ravel()[:3], want[ch])
    assert seg == 12 * 7 * 5 * depth
    if not metallic:         # the same diffuse sphere at max_depth 3: black, after two intersection passes per sample
        img3, seg3 = oracle.render_hybrid(spheres, mats, _cam(binding), oracle.make_params(12, 7, 5, 3, 1, 1, 0, flags=SEM_HYBRID, seed=9), "f32")
        assert float(np.abs(img3).max()) == 0.0 and seg3 == 12 * 7 * 5 * 2


def test_thread_count_precision_and_row_order(oracle, binding):
    from spira_hip import scenes
    s = scenes.scene_s1()
    args = (s["spheres5"], s["materials8"], s["camera12"])
    p = oracle.make_params(40, 24, 3, 4, 5, 5, 0, flags=SEM_HYBRID, seed=5)
    a, sa = oracle.render_hybrid(*args, p, "f32", n_threads=1)
    b, sb = oracle.render_hybrid(*args, p, "f32", n_threads=4)
    assert np.array_equal(a, b) and sa == sb == 40 * 24 * 3 * 4
    assert 0.0 <= float(a.min()) and float(a.max()) <= 1.0                     # a mean of tone-mapped samples
    d, _ = oracle.render_hybrid(*args, p, "f64")
    assert float(np.abs(d - a).max()) < 2e-2 and float(np.median(np.abs(d - a))) < 1e-6      # same streams; a few hit / miss flips at Float32 silhouettes
    up, _ = oracle.render_hybrid(*args, oracle.make_params(40, 24, 3, 4, 5, 5, 0, flags=SEM_HYBRID | 0x1000, seed=5), "f32")
    assert np.array_equal(up, a[:, ::-1])                                      # SPIRA_ROWS_BOTTOM_UP = the device-buffer order of :1177-1188
    with pytest.raises(RuntimeError):                                          # whole images only
        oracle.render_hybrid(*args, oracle.make_params(40, 24, 3, 4, 5, 5, 0, flags=SEM_HYBRID, seed=5, rows=8), "f32")


def test_xorshift_stream_feeds_the_jitter(oracle):
    """K3 (:632-638): state = xorshift(rng + sample), then two more steps give the jitter; known answers of the stream."""
    lib = oracle.lib()
    assert [lib.oracle_xorshift32(1), lib.oracle_xorshift32(270369), lib.oracle_xorshift32(67634689)] == [270369, 67634689, 2647435461]
    assert lib.oracle_xorshift_uniform(0xFFFFFFFF) == 1.0                      # Float32(s / typemax(UInt32)) CAN return 1.0 (:421-425)
