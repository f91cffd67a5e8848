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


def test_fuzz_api_sequences(gpu, oracle):
    """Seeded random SEQUENCES of C-ABI calls on one process: resident scene handles (single and multi-device form) created, used through
    every entry that takes one (host outputs, device outputs on a torch stream, the RCCL entry with one device) and destroyed in any order,
    host-array renders and progressive accumulation in between, spira_shutdown() at random moments, image sizes jumping up and down
    — so that every call meets workspaces, communicators and cached scenes left by any other.  Every image against the oracle.
    SPIRA_FUZZ_SEQ_SEED / SPIRA_FUZZ_SEQ_OPS: longer one-off campaigns."""
    import torch
    rng = np.random.default_rng(int(os.environ.get("SPIRA_FUZZ_SEQ_SEED", "20261005")))
    n_ops = int(os.environ.get("SPIRA_FUZZ_SEQ_OPS", "80"))
    os.environ.pop("SPIRA_MULTI_REHEARSE", None)
    handles = []                      # (Scene, scene dict)
    memo = {}

    def shape():
        big = rng.random() < 0.15
        W, H = (int(rng.integers(150, 400)), int(rng.integers(100, 260))) if big else (int(rng.integers(2, 97)), int(rng.integers(2, 61)))
        return W, H, int(rng.integers(1, 6)), int(rng.integers(1, 9))

    def expect(key, s, W, H, spp, depth, seed, prec):
        k = (key, W, H, spp, depth, seed, prec)
        if k not in memo:
            ns, nm, nt = _counts(s)
            memo[k] = oracle.render(*_args(s), oracle.make_params(W, H, spp, depth, ns, nm, nt, seed=seed), prec)
        return memo[k]

    def check(what, hdr, exp, seg=None):
        nbad, worst = _close(hdr, exp[0])
        assert nbad == 0, (what, nbad, worst)
        assert (gpu.counters()["segments"] if seg is None else seg) == exp[2], what

    scene_id = 0
    try:
        for it in range(n_ops):
            op = rng.choice(["create", "create", "handle", "handle", "handle", "handle", "host", "host", "accumulate", "destroy", "shutdown"])
            if os.environ.get("SPIRA_FUZZ_LOG"):
                with open(os.environ["SPIRA_FUZZ_LOG"], "a") as fh:
                    fh.write("seq %d %s (%d handles)\n" % (it, op, len(handles)))
            if op == "create" or (op in ("handle", "destroy") and not handles):
                if len(handles) >= 4:
                    continue
                kind, s, *_ = _case(rng)
                prec = "f32" if rng.random() < 0.5 else "f64"
                multi = rng.random() < 0.4
                scene_id += 1
                handles.append((gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec, n_devices=1 if multi else 0), s, scene_id, multi))
            elif op == "handle":
                h, s, sid, multi = handles[int(rng.integers(0, len(handles)))]
                W, H, spp, depth = shape()
                seed = int(rng.integers(0, 2 ** 40))
                p = h.params(W, H, spp, depth, seed=seed, flags=gpu.POST_NONE)
                exp = expect(sid, s, W, H, spp, depth, seed, h.prec)
                how = rng.choice(["host", "device", "multi"] if multi else ["host", "device"])
                if how == "host":
                    check((it, "handle.render", h.prec), h.render(s["camera12"], p)[0], exp)
                elif how == "multi":
                    check((it, "handle.render_multi", h.prec), h.render_multi(s["camera12"], p, 1)[0], exp)
                else:
                    out = torch.empty((3, H, W), dtype=torch.float32 if h.prec == "f32" else torch.float64, device="cuda")
                    st = torch.cuda.Stream() if rng.random() < 0.5 else torch.cuda.current_stream()
                    h.render_device(s["camera12"], p, out.data_ptr(), 0, st.cuda_stream)
                    st.synchronize()
                    check((it, "handle.render_device", h.prec), out.cpu().numpy(), exp)
            elif op == "host":
                kind, s, W, H, spp, depth = _case(rng)
                ns, nm, nt = _counts(s)
                prec = "f32" if rng.random() < 0.5 else "f64"
                seed = int(rng.integers(0, 2 ** 40))
                scene_id += 1
                exp = expect(scene_id, s, W, H, spp, depth, seed, prec)
                p = gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=seed, flags=gpu.POST_NONE)
                if rng.random() < 0.3:
                    check((it, "render_multi", prec), gpu.render_multi(*_args(s), p, 1, prec)[0], exp)
                else:
                    check((it, "render", prec), gpu.render(*_args(s), p, prec)[0], exp)
            elif op == "accumulate":
                kind, s, W, H, spp, depth = _case(rng)
                ns, nm, nt = _counts(s)
                prec = "f32" if rng.random() < 0.5 else "f64"
                npdt = np.float32 if prec == "f32" else np.float64
                seed = int(rng.integers(0, 2 ** 40))
                total = spp + int(rng.integers(1, 5))
                scene_id += 1
                exp = expect(scene_id, s, W, H, total, depth, seed, prec)
                sums = np.zeros((3, H, W), dtype=npdt)
                seg = 0
                for s0, n in ((0, spp), (spp, total - spp)):
                    gpu.accumulate(*_args(s), gpu.make_params(W, H, n, depth, ns, nm, nt, seed=seed), s0, sums, None, prec)
                    seg += gpu.counters()["segments"]
                check((it, "accumulate", prec), sums / npdt(total), exp, seg)
            elif op == "destroy":
                handles.pop(int(rng.integers(0, len(handles))))[0].destroy()
            else:
                # frees the cached workspaces and communicators of every device; resident scene handles stay valid
                gpu.lib().spira_shutdown()
    finally:
        for h, *_ in handles:
            h.destroy()
