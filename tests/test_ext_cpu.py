"""CPU: the oracle's restatement of the two extensions (SPIRA_EXT_DIELECTRIC, SPIRA_EXT_SPECTRAL) against analytic known
answers.  PARITY UNPINNED by construction: the reference only names these features (README.md:10,
src/spira_path_trace_kernel.metal:225), it has no code for them — the semantics are the build's own (include/spira_hip.h)."""
import os
import re

import numpy as np

from spira_hip import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT_DIELECTRIC, EXT_SPECTRAL, POST_NONE = 0x20000, 0x40000, 0x300


def _table():
    txt = open(os.path.join(ROOT, "include", "spira_spd.h")).read()
    rows = re.findall(r"\{([^{}]+)\},", txt)
    return np.array([[float(v) for v in r.split(",")] for r in rows])


def _mean_of_product(p, q):
    a0, a1, b0, b1 = p[:-1], p[1:], q[:-1], q[1:]
    return float(np.sum(a0 * b0 / 3 + a1 * b1 / 3 + (a0 * b1 + a1 * b0) / 6) / 35.0)


def test_spd_table_properties():
    t = _table()
    assert t.shape == (6, 36)
    assert np.allclose(t[0] + t[1] + t[2], 1.0, atol=1e-15) and (t[:3] >= 0).all()        # the uplift basis is a partition of unity
    m = np.array([[_mean_of_product(t[3 + c], t[k]) for k in range(3)] for c in range(3)])
    assert np.allclose(m, np.eye(3), atol=1e-12)                                          # E_lambda[w_c b_k] = delta_ck


def _slab_scene(ior, tint=(1.0, 1.0, 1.0)):
    """One big triangle (a horizontal slab surface at y = 0, normal +y) of a dielectric material, camera above looking down at 45 deg."""
    mats = np.array([[tint[0], tint[1], tint[2], 0, 0, 0, 0.0, -ior]], dtype=np.float64)
    tri = np.array([[-50, 0, 50, 50, 0, 50, 0, 0, -50, 1]], dtype=np.float64)           # cross(e1, e2) points to +y
    from spira_hip import _binding as B
    cam = B.camera_lookat([0.0, 2.0, 2.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 20.0, 1.0, 1.0, prec="f64")
    return dict(spheres5=np.zeros((0, 5)), materials8=mats, triangles10=tri, camera12=cam)


def test_dielectric_obeys_snell_and_schlick(oracle):
    ior = 1.5
    s = _slab_scene(ior)
    W = H = 33
    p = oracle.make_params(W, H, 4000, 2, 0, 1, 1, flags=EXT_DIELECTRIC | POST_NONE, seed=3)
    n_refl = n_refr = 0
    for smp in range(4000):
        cnt, prims, ts, dirs, rad = oracle.trace_path(np.zeros((1, 5)), s["materials8"], s["triangles10"], s["camera12"],
                                                      oracle.make_params(W, H, 4000, 2, 0, 1, 1, flags=EXT_DIELECTRIC | POST_NONE, seed=3), 17, 17, smp, "f64")
        assert cnt == 2 and prims[0] == 0
        d0, d1 = dirs[0], dirs[1]
        ci = -d0[1]
        if d1[1] > 0:                                        # reflected: mirror direction
            assert np.allclose(d1, [d0[0], -d0[1], d0[2]], atol=1e-12)
            n_refl += 1
        else:                                                # refracted: Snell, in the plane of incidence
            si, st = np.sqrt(1 - ci * ci), np.sqrt(d1[0] ** 2 + d1[2] ** 2)
            assert abs(st - si / ior) < 1e-9 and abs(np.linalg.norm(d1) - 1) < 1e-12
            n_refr += 1
    r0 = ((1 - ior) / (1 + ior)) ** 2
    ci = 1 / np.sqrt(2)                                      # the centre pixel looks down at about 45 degrees
    schlick = r0 + (1 - r0) * (1 - ci) ** 5
    assert abs(n_refl / 4000 - schlick) < 0.02, (n_refl, schlick)
    assert p.flags & EXT_DIELECTRIC


def test_dielectric_total_internal_reflection(oracle):
    """From inside the medium (camera below the surface, looking up at a grazing angle) beyond the critical angle: always reflected."""
    ior = 1.5
    s = _slab_scene(ior)
    from spira_hip import _binding as B
    cam = B.camera_lookat([0.0, -1.0, 4.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 5.0, 1.0, 1.0, prec="f64")   # incidence ~76 deg > asin(1/1.5) = 41.8
    for smp in range(200):
        cnt, prims, ts, dirs, rad = oracle.trace_path(np.zeros((1, 5)), s["materials8"], s["triangles10"], cam,
                                                      oracle.make_params(9, 9, 200, 2, 0, 1, 1, flags=EXT_DIELECTRIC, seed=1), 5, 5, smp, "f64")
        assert cnt == 2 and dirs[1][1] < 0                   # came from below going up, leaves going down


def test_spectral_single_interaction_reproduces_rgb_in_expectation(oracle):
    """max_depth 1: a pixel is the emission of what it sees, or the sky: ONE uplifted triple, so the expectation over the wavelength is
    the RGB value itself (E[w_c b_k] = delta_ck).  4096 wavelength samples per pixel: the mean must be close to the RGB render."""
    s = scenes.scene_s2()
    a = (s["spheres5"], s["materials8"], s["triangles10"], s["camera12"])
    rgb, _, _ = oracle.render(*a, oracle.make_params(24, 14, 1, 1, 5, 6, 1, flags=POST_NONE, seed=2), "f64")
    spec, _, _ = oracle.render(*a, oracle.make_params(24, 14, 4096, 1, 5, 6, 1, flags=EXT_SPECTRAL | POST_NONE, seed=2), "f64")
    rgb4096, _, _ = oracle.render(*a, oracle.make_params(24, 14, 4096, 1, 5, 6, 1, flags=POST_NONE, seed=2), "f64")
    assert not np.allclose(spec, rgb4096, rtol=1e-6)         # it IS a different estimator
    lit = rgb4096 > 0.05
    assert np.abs(spec[lit] / rgb4096[lit] - 1).max() < 0.4                               # single pixels: Monte-Carlo noise of the wavelength draw
    assert np.abs(spec.mean(axis=(1, 2)) / rgb4096.mean(axis=(1, 2)) - 1).max() < 0.02    # channel means over the frame (1.4 M draws)


def test_extensions_off_change_nothing(oracle):
    s = scenes.scene_s2()
    a = (s["spheres5"], s["materials8"], s["triangles10"], s["camera12"])
    x, _, sx = oracle.render(*a, oracle.make_params(40, 22, 3, 5, 5, 6, 1, flags=POST_NONE, seed=9), "f32")
    y, _, sy = oracle.render(*a, oracle.make_params(40, 22, 3, 5, 5, 6, 1, flags=POST_NONE | EXT_DIELECTRIC, seed=9), "f32")   # no negative roughness in S2
    assert np.array_equal(x, y) and sx == sy


def test_extension_golden_fixture(oracle):
    """Regression pin of the extensions' restatement (tests/golden/ext_s2glass_96x54_spp4_d8.json, generated by this oracle — NOT reference
    output: none exists)."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "ext_s2glass_96x54_spp4_d8.json")))
    s = scenes.scene_s2()
    m = np.array(g["materials8"])
    for name in ("dielectric", "spectral", "both"):
        for prec in ("f64", "f32"):
            hdr, _, seg = oracle.render(s["spheres5"], m, s["triangles10"], s["camera12"],
                                        oracle.make_params(96, 54, 4, 8, 5, 6, 1, flags=g[name]["flags"] | POST_NONE, seed=g["seed"]), prec)
            got = np.array([[hdr[c, y, x] for c in range(3)] for y, x in g["pixels"]], dtype=np.float64)
            assert seg == g[name][prec]["segments"] and np.array_equal(got, np.array(g[name][prec]["values"])), (name, prec)
