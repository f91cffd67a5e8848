"""GPU (MI355X): seeded fuzz of the C ABI against the oracle — random image sizes, spp, depth, tilings, batch sizes,
kernel organisations, precisions, estimators, the two extensions and scenes (spheres, LDS triangles, BVH meshes).  Every case: image within
the north-star tolerance of the oracle and identical segment counts."""
import os

import numpy as np
import pytest

from spira_hip import distributed as D
from spira_hip import scenes
from test_gpu_parity import _args, _close, _counts, random_scene

pytestmark = pytest.mark.gpu


def _case(rng):
    kinds = os.environ.get("SPIRA_FUZZ_KINDS", "s1,s2,s3,soup_lds,soup_bvh,blob").split(",")      # (one-off campaigns may narrow the scene kinds, e.g. to the BVH ones)
    kind = rng.choice(kinds)
    if kind == "s1":
        s = scenes.scene_s1()
    elif kind == "s2":
        s = scenes.scene_s2()
    elif kind == "s3":
        s = scenes.scene_s3()
    elif kind == "soup_lds":
        s = random_scene(rng, int(rng.integers(0, 12)), int(rng.integers(1, 33)))
    elif kind == "soup_bvh":
        s = random_scene(rng, int(rng.integers(0, 6)), int(rng.integers(33, 400)))
    else:
        s = scenes.scene_s4(level=int(rng.integers(1, 4)))
    W, H = int(rng.integers(2, 97)), int(rng.integers(2, 61))
    spp, depth = int(rng.integers(1, 9)), int(rng.integers(1, 10))
    return kind, s, W, H, spp, depth


def test_fuzz_against_oracle(gpu, oracle):
    # SPIRA_FUZZ_SEED / SPIRA_FUZZ_ITERS: longer one-off campaigns (the defaults are what the suite runs)
    rng = np.random.default_rng(int(os.environ.get("SPIRA_FUZZ_SEED", "20261004")))
    for it in range(int(os.environ.get("SPIRA_FUZZ_ITERS", "60"))):
        kind, s, W, H, spp, depth = _case(rng)
        ns, nm, nt = _counts(s)
        prec = "f32" if rng.random() < 0.5 else "f64"
        sem = 0
        if nt == 0 and rng.random() < 0.3:
            sem = int(rng.choice([1, 2]))
        kflag = (gpu.KERNEL_DEFAULT, gpu.KERNEL_WAVEFRONT, gpu.KERNEL_BOUNCE, gpu.KERNEL_MEGA)[int(rng.integers(0, 4))]
        ext = 0
        if sem == 0 and kflag != gpu.KERNEL_BOUNCE and rng.random() < 0.3:        # the extensions: some materials become glass, and / or spectral transport
            ext = int(rng.choice([gpu.EXT_DIELECTRIC, gpu.EXT_SPECTRAL, gpu.EXT_DIELECTRIC | gpu.EXT_SPECTRAL]))
            if ext & gpu.EXT_DIELECTRIC:
                m = s["materials8"].copy()
                glass = rng.random(len(m)) < 0.4
                m[glass, 7] = -rng.uniform(1.05, 2.4, int(glass.sum())).astype(np.float32).astype(np.float64)
                s = dict(s, materials8=m)
        seed = int(rng.integers(0, 2 ** 40))
        batch = int(rng.choice([0, 1, W * H * 2 + 3, 1 << 20]))
        tile = {}
        if rng.random() < 0.4:
            world = int(rng.integers(2, 5))
            tile = D.tile_params(H, world, int(rng.integers(0, world)), int(rng.integers(1, 9)))
            if tile["rows"] == 0:
                tile = {}
        flags = sem | kflag | ext | gpu.POST_NONE
        if os.environ.get("SPIRA_FUZZ_FRESH"):    # free the device's cached workspaces first: every case must size what it uses itself
            gpu.lib().spira_shutdown()
        if os.environ.get("SPIRA_FUZZ_LOG"):      # one line per case BEFORE it runs (flushed): which case a crash of the process belongs to
            with open(os.environ["SPIRA_FUZZ_LOG"], "a") as fh:
                fh.write("%d %s ns=%d nm=%d nt=%d %dx%d spp=%d depth=%d %s flags=%#x seed=%d batch=%d tile=%s\n" % (it, kind, ns, nm, nt, W, H, spp, depth, prec, flags, seed, batch, tile))
        hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, depth, ns, nm, nt, flags=flags, seed=seed, batch_rays=batch, **tile), prec)
        po = oracle.make_params(W, H, spp, depth, ns, nm, nt, flags=flags, seed=seed, **tile)
        if sem == 0:
            ohdr, _, oseg = oracle.render(*_args(s), po, prec)
        else:
            ohdr, _, oseg = oracle.render_variant(s["spheres5"], s["materials8"], s["camera12"], po, prec)
        nbad, worst = _close(hdr, ohdr)
        assert nbad == 0, (it, kind, W, H, spp, depth, prec, sem, kflag, batch, tile, nbad, worst)
        assert gpu.counters()["segments"] == oseg, (it, kind, prec, sem)
