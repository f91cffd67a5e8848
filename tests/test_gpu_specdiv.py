"""GPU (MI355X): k_path's speculative division (csrc/spira_device.h, SpecDiv).

The default organisation divides through a shared refined reciprocal and a slimmed square root that are bit-identical to the
compiler's IEEE expansions while operand exponents are moderate; waves that saw anything else are rendered again by the exact
instantiation.  Here: (1) the arithmetic contract itself, 2^28 random operand groups per precision on the device
(tests/native/div_exact.hip, built with hipcc at test time); (2) images with speculation off / on / "every wave rendered again"
are the same bits, and so are the counters; (3) scenes that DO leave the window — scaled by 1e-30, or holding one sphere of
astronomic size far away — are rendered again exactly where needed and still match the oracle."""
import os
import subprocess

import numpy as np
import pytest

from spira_hip import scenes
from test_gpu_parity import _args, _close, _counts

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_quotients_and_square_roots_are_the_ieee_ones(tmp_path):
    exe = str(tmp_path / "div_exact")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-o", exe, os.path.join(ROOT, "tests", "native", "div_exact.hip")], check=True, timeout=600)
    out = subprocess.run([exe, "4096", "256", "20261004"], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    for line in out.stdout.strip().splitlines():
        assert " 0 mismatching quotients" in line and "roots: 0 mismatching" in line and "pixel quotients: 0 mismatching" in line, line


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_modes_are_bit_identical(gpu, prec):
    for s, (W, H, spp, depth) in ((scenes.scene_s1(), (480, 270, 16, 8)), (scenes.scene_s3(), (320, 180, 8, 8)), (scenes.scene_s2(), (256, 144, 8, 6))):
        ns, nm, nt = _counts(s)
        p = gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.POST_NONE, seed=77)
        got = {}
        for mode in (0, 1, 2):
            with _Env(SPIRA_SPEC_DIV=mode):
                hdr, _ = gpu.render(*_args(s), p, prec)
                c = gpu.counters()
            got[mode] = (hdr, c)
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][0], got[2][0])
        for k in ("segments", "rays_enqueued", "radiance_rmw", "radiance_stores", "rays_parked"):
            assert got[0][1][k] == got[1][1][k] == got[2][1][k], k
        assert got[0][1]["redone_waves"] == 0
        assert got[1][1]["redone_waves"] == 0          # an ordinary scene: nothing leaves the window (zeros are handled where they occur)
        assert got[2][1]["redone_waves"] > 0


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_metal_estimator_modes_are_bit_identical(gpu, prec):
    """The .metal estimator's one-lane-per-pixel kernel speculates the same way (fresh renders only): sums, LCG states and segment
    counts are the same with speculation off / on / every wave rendered again; a progressive continuation never speculates."""
    s = scenes.scene_s1()
    ns, nm, nt = _counts(s)
    W, H, spp, depth = 320, 180, 8, 8
    p = gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.SEM_METAL | gpu.POST_NONE, seed=41)
    got = {}
    for mode in (0, 1, 2):
        with _Env(SPIRA_SPEC_DIV=mode):
            hdr, _ = gpu.render(*_args(s), p, prec)
            got[mode] = (hdr, gpu.counters())
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][0], got[2][0])
    assert got[0][1]["segments"] == got[1][1]["segments"] == got[2][1]["segments"]
    assert got[0][1]["redone_waves"] == 0 and got[1][1]["redone_waves"] == 0 and got[2][1]["redone_waves"] == (W * H + 63) // 64
    # the wavefront form of the estimator (k_path_metal): same sums in all three modes, and the same as the one-lane kernel's
    pw = gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.SEM_METAL | gpu.KERNEL_WAVEFRONT | gpu.POST_NONE, seed=41)
    for mode in (0, 1, 2):
        with _Env(SPIRA_SPEC_DIV=mode):
            hdr, _ = gpu.render(*_args(s), pw, prec)
            c = gpu.counters()
        assert np.array_equal(hdr, got[0][0]) and c["segments"] == got[0][1]["segments"], mode
        assert (c["redone_waves"] > 0) == (mode == 2), (mode, c["redone_waves"])
    # two progressive calls of 4 samples (the second continues sums and LCG states in place) == one call of 8
    npdt = np.float32 if prec == "f32" else np.float64
    sums = np.zeros((3, H, W), dtype=npdt)
    states = np.zeros(W * H, dtype=np.uint32)
    p4 = gpu.make_params(W, H, 4, depth, ns, nm, nt, flags=gpu.SEM_METAL, seed=41)
    gpu.accumulate(*_args(s), p4, 0, sums, states, prec)
    gpu.accumulate(*_args(s), p4, 4, sums, states, prec)
    assert gpu.counters()["redone_waves"] == 0
    assert np.array_equal(sums / npdt(8), got[0][0])


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_extension_instantiations_modes_are_bit_identical(gpu, prec):
    """The dielectric / spectral instantiations of k_path speculate too (their own divisions — 1/ior, Schlick's r0, Snell's square root —
    stay the compiler's)."""
    from test_gpu_ext import glass_scene
    s = glass_scene()
    ns, nm, nt = _counts(s)
    for ext in (gpu.EXT_DIELECTRIC, gpu.EXT_DIELECTRIC | gpu.EXT_SPECTRAL):
        p = gpu.make_params(240, 136, 6, 8, ns, nm, nt, flags=ext | gpu.POST_NONE, seed=17)
        got = {}
        for mode in (0, 1, 2):
            with _Env(SPIRA_SPEC_DIV=mode):
                hdr, _ = gpu.render(*_args(s), p, prec)
                got[mode] = (hdr, gpu.counters())
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][0], got[2][0])
        assert got[0][1]["segments"] == got[1][1]["segments"] == got[2][1]["segments"]
        assert got[1][1]["redone_waves"] == 0 and got[2][1]["redone_waves"] > 0


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_cpu_estimator_modes_are_bit_identical(gpu, prec):
    """SPIRA_SEM_CPU (render_with_cpu's estimator, one lane per path): the same three modes."""
    s = scenes.scene_s1()
    ns, nm, nt = _counts(s)
    W, H, spp, depth = 320, 180, 8, 8
    p = gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.SEM_CPU | gpu.POST_NONE, seed=43)
    got = {}
    for mode in (0, 1, 2):
        with _Env(SPIRA_SPEC_DIV=mode):
            hdr, _ = gpu.render(*_args(s), p, prec)
            got[mode] = (hdr, gpu.counters())
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][0], got[2][0])
    assert got[0][1]["segments"] == got[1][1]["segments"] == got[2][1]["segments"]
    assert got[0][1]["redone_waves"] == 0 and got[1][1]["redone_waves"] == 0 and got[2][1]["redone_waves"] > 0


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_scene_outside_the_window_is_rendered_again_and_matches_the_oracle(gpu, oracle, prec):
    """S1 scaled by 1e-30 (1e-200 in Float64): every square underflows the window.  The predictor would switch speculation off
    (SPIRA_SPEC_DIV=1 renders it with redone_waves == 0); forced on (=3), every wave reports itself and is rendered again."""
    s = scenes.scene_s1()
    k = 1e-30 if prec == "f32" else 1e-200
    sp = s["spheres5"].astype(np.float64).copy(); sp[:, :4] *= k
    cam = s["camera12"].astype(np.float64) * k
    ns, nm, nt = _counts(s)
    W, H, spp, depth = 160, 90, 4, 6
    p = gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=gpu.POST_NONE, seed=5)
    with _Env(SPIRA_SPEC_DIV=0):
        ref, _ = gpu.render(sp, s["materials8"], None, cam, p, prec)
        seg = gpu.counters()["segments"]
    with _Env(SPIRA_SPEC_DIV=1):
        a, _ = gpu.render(sp, s["materials8"], None, cam, p, prec)
        assert gpu.counters()["redone_waves"] == 0
    with _Env(SPIRA_SPEC_DIV=3):
        b, _ = gpu.render(sp, s["materials8"], None, cam, p, prec)
        c = gpu.counters()
    assert c["redone_waves"] > 0 and c["segments"] == seg
    assert np.array_equal(ref, a) and np.array_equal(ref, b)
    ohdr, _, oseg = oracle.render(sp, s["materials8"], None, cam, oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=5), prec)
    # t_min = 0.001 is not scaled with the scene: the tiny world is one where every hit is closer than t_min — the sky everywhere
    assert _close(b, ohdr)[0] == 0 and oseg == seg


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_partial_redo(gpu, oracle, prec):
    """Only SOME waves leave the window: a sphere far outside the ordinary range (radius 3e7 at a distance of 1e9 in Float32, 3e118 at
    1e120 in Float64) straight ahead of the camera.  b*b of its test is beyond the window, but the fold only happens for rays whose
    line meets it (:120) — the camera rays of a disc of ~27 pixels radius, and the odd scattered ray.  Forced on (the predictor would
    decline this scene), the waves that own those rays are rendered again, the others are not; the image is the exact one."""
    s = scenes.scene_s1()
    ns, nm, nt = _counts(s)
    cam = s["camera12"].astype(np.float64)
    view = cam[3:6] + 0.5 * cam[6:9] + 0.5 * cam[9:12] - cam[0:3]
    view /= np.linalg.norm(view)
    D, R = (1e9, 3e7) if prec == "f32" else (1e120, 3e118)
    far = cam[0:3] + view * D
    sp = np.vstack([s["spheres5"].astype(np.float64), [[far[0], far[1], far[2], R, 1.0]]])
    W, H, spp, depth = 320, 180, 8, 6
    p = gpu.make_params(W, H, spp, depth, ns + 1, nm, nt, flags=gpu.POST_NONE, seed=9)
    with _Env(SPIRA_SPEC_DIV=0):
        ref, _ = gpu.render(sp, s["materials8"], None, s["camera12"], p, prec)
    with _Env(SPIRA_SPEC_DIV=1):
        dflt, _ = gpu.render(sp, s["materials8"], None, s["camera12"], p, prec)
        assert gpu.counters()["redone_waves"] == 0                    # declined by the predictor
    with _Env(SPIRA_SPEC_DIV=3):
        got, _ = gpu.render(sp, s["materials8"], None, s["camera12"], p, prec)
        c = gpu.counters()
    waves = (W * H * spp + 127) // 128
    print("far sphere, %s: %d of %d waves rendered again" % (prec, c["redone_waves"], waves))
    assert 0 < c["redone_waves"] < 0.9 * waves
    assert np.array_equal(ref, got) and np.array_equal(ref, dflt)
    ohdr, _, oseg = oracle.render(sp, s["materials8"], None, s["camera12"], oracle.make_params(W, H, spp, depth, ns + 1, nm, nt, seed=9), prec)
    assert _close(got, ohdr)[0] == 0 and oseg == c["segments"]
