"""GPU (MI355X): host-runtime contracts of the C ABI — stream ordering of the shared workspaces, scene handles,
large LDS scenes, progressive accumulation against the ORACLE (not only against the HIP one-shot)."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from spira_hip import scenes
from test_gpu_parity import _args, _close, _counts, random_scene

pytestmark = pytest.mark.gpu


def test_two_streams_without_host_sync(gpu):
    """every call on a device shares one set of workspaces.  A device-pointer render on a side stream
    immediately followed — no synchronisation — by renders on other streams (the library's own, another side stream, with
    a DIFFERENT scene and size) must all produce the images they produce when run alone."""
    import torch
    s1, s3 = scenes.scene_s1(), scenes.scene_s3()
    jobs = [(s1, 640, 360, 16, 6, "f32"), (s3, 320, 180, 8, 8, "f64"), (s1, 256, 144, 32, 5, "f64"), (s3, 480, 270, 4, 8, "f32")]
    want = []
    for (s, W, H, spp, d, prec) in jobs:
        ns, nm, nt = _counts(s)
        hdr, _ = gpu.render(*_args(s), gpu.make_params(W, H, spp, d, ns, nm, nt, flags=gpu.POST_NONE, seed=5), prec)
        want.append(hdr)
    for rep in range(3):
        streams = [torch.cuda.Stream() for _ in jobs]
        outs = [torch.empty((3, H, W), dtype=torch.float32 if prec == "f32" else torch.float64, device="cuda:0") for (_, W, H, _, _, prec) in jobs]
        for (s, W, H, spp, d, prec), st, out in zip(jobs, streams, outs):
            ns, nm, nt = _counts(s)
            gpu.render_device(*_args(s), gpu.make_params(W, H, spp, d, ns, nm, nt, flags=gpu.POST_NONE, seed=5), out.data_ptr(), 0, st.cuda_stream, prec)
        # a host-pointer render on the library's own stream, still without any synchronisation of the side streams
        ns, nm, nt = _counts(s3)
        host, _ = gpu.render(*_args(s3), gpu.make_params(200, 100, 8, 8, ns, nm, nt, flags=gpu.POST_NONE, seed=5), "f32")
        for st in streams:
            st.synchronize()
        for out, w in zip(outs, want):
            assert np.array_equal(out.cpu().numpy(), w)
        if rep == 0:
            host0 = host
        assert np.array_equal(host, host0)


def test_scene_handle_equals_arrays_and_is_reusable(gpu):
    s = scenes.scene_s4(level=4)      # 5 120 triangles: BVH path
    ns, nm, nt = _counts(s)
    for prec in ("f32", "f64"):
        ref, _ = gpu.render(*_args(s), gpu.make_params(160, 90, 4, 6, ns, nm, nt, seed=9), prec)
        with gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec) as sc:
            assert sc.counts == (ns, nm, nt)
            for _ in range(3):
                hdr, _ = sc.render(s["camera12"], sc.params(160, 90, 4, 6, seed=9))
                assert np.array_equal(hdr, ref)
            # interleave a host-array render of ANOTHER scene: the handle's buffers are its own
            s1 = scenes.scene_s1()
            gpu.render(*_args(s1), gpu.make_params(64, 36, 2, 4, 5, 5, 0, seed=1), prec)
            hdr, _ = sc.render(s["camera12"], sc.params(160, 90, 4, 6, seed=9))
            assert np.array_equal(hdr, ref)
            other = "f64" if prec == "f32" else "f32"
            cam = np.ascontiguousarray(s["camera12"], dtype=np.float64 if other == "f64" else np.float32)
            fn = gpu.lib().spira_render_scene_f64 if other == "f64" else gpu.lib().spira_render_scene_f32
            p = sc.params(16, 9, 1, 1)
            out = np.zeros((3, 9, 16), dtype=cam.dtype)
            assert fn(sc._h, cam.ctypes.data_as(C.c_void_p), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == -1   # wrong precision
    with pytest.raises(gpu.SpiraError):
        bad = s["spheres5"].copy(); bad[0, 4] = 99
        gpu.Scene(bad, s["materials8"], None, "f32")


def test_scene_handle_device_output_on_a_side_stream(gpu):
    """spira_render_scene_device_*: a handle, device outputs, the caller's stream (what bench.py's timed loop does)."""
    import torch
    s = scenes.scene_s3()
    ns, nm, nt = _counts(s)
    for prec, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        p = gpu.make_params(192, 108, 6, 8, ns, nm, nt, flags=gpu.POST_ACES_GAMMA, seed=8)
        hdr, img = gpu.render(*_args(s), p, prec, want_img=True)
        st = torch.cuda.Stream()
        d_hdr = torch.empty((3, 108, 192), dtype=tdt, device="cuda:0")
        d_img = torch.empty((3, 108, 192), dtype=tdt, device="cuda:0")
        with gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec) as sc:
            for _ in range(3):
                sc.render_device(s["camera12"], p, d_hdr.data_ptr(), d_img.data_ptr(), st.cuda_stream)
            st.synchronize()
        assert np.array_equal(d_hdr.cpu().numpy(), hdr) and np.array_equal(d_img.cpu().numpy(), img)


def test_scene_handle_removes_per_frame_host_work(gpu):
    """Config-5 mesh (81 920 triangles): per-call host time of the array entry point (re-validates every material index and
    hashes 3.3 / 6.5 MB of triangles to find the cached tree) against a handle.  Reported, and the handle must not be slower."""
    s = scenes.scene_s4()
    ns, nm, nt = _counts(s)
    p = gpu.make_params(256, 144, 1, 2, ns, nm, nt, seed=3)
    for prec in ("f32", "f64"):
        gpu.render(*_args(s), p, prec)       # builds + caches the tree
        t0 = time.perf_counter()
        for _ in range(5):
            ref, _ = gpu.render(*_args(s), p, prec)
        t_arr = (time.perf_counter() - t0) / 5
        with gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec) as sc:
            sc.render(s["camera12"], p)
            t0 = time.perf_counter()
            for _ in range(5):
                hdr, _ = sc.render(s["camera12"], p)
            t_h = (time.perf_counter() - t0) / 5
            assert np.array_equal(hdr, ref)
        print("config-5 scene, %s: %.2f ms per call with host arrays, %.2f ms with a scene handle" % (prec, t_arr * 1e3, t_h * 1e3))
        assert t_h < t_arr


@pytest.mark.parametrize("kernel", ["wavefront", "mega"])
def test_large_lds_scene_1024_spheres(gpu, oracle, kernel):
    """SPIRA_MAX_LDS_SPHERES spheres and 400 materials in Float64: 32 KB + 4 KB + 25 KB of scene (+ 16 KB of work lists) —
    a dynamic-LDS launch above 64 KB — against the oracle."""
    rng = np.random.default_rng(3)
    s = random_scene(rng, 1024, 8, n_mats=400)
    ns, nm, nt = _counts(s)
    flags = gpu.KERNEL_MEGA if kernel == "mega" else gpu.KERNEL_WAVEFRONT
    for prec in ("f64", "f32"):
        hdr, _ = gpu.render(*_args(s), gpu.make_params(96, 54, 2, 5, ns, nm, nt, flags=flags, seed=21), prec)
        ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(96, 54, 2, 5, ns, nm, nt, seed=21), prec)
        assert _close(hdr, ohdr)[0] == 0 and gpu.counters()["segments"] == oseg, prec


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_progressive_sums_match_oracle(gpu, oracle, prec):
    """k calls of n samples against the ORACLE's one render of k*n samples (r1 only compared with the HIP one-shot)."""
    s = scenes.scene_s2()
    ns, nm, nt = _counts(s)
    W, H, depth, total = 120, 68, 6, 20
    npdt = np.float32 if prec == "f32" else np.float64
    ohdr, _, oseg = oracle.render(*_args(s), oracle.make_params(W, H, total, depth, ns, nm, nt, seed=31), prec)
    for chunks in ([20], [7, 7, 6], [1, 19]):
        sums = np.zeros((3, H, W), dtype=npdt)
        s0, seg = 0, 0
        for n in chunks:
            gpu.accumulate(*_args(s), gpu.make_params(W, H, n, depth, ns, nm, nt, seed=31), s0, sums, None, prec)
            seg += gpu.counters()["segments"]
            s0 += n
        assert _close(sums / npdt(total), ohdr)[0] == 0 and seg == oseg, chunks


def test_progressive_metal_needs_rng_states(gpu):
    """SPIRA_SEM_METAL with sample0 > 0 and no rng_states would replay the first call's samples: SPIRA_E_INVALID."""
    s = scenes.scene_s1()
    sums = np.zeros((3, 36, 64), dtype=np.float32)
    gpu.accumulate(s["spheres5"], s["materials8"], None, s["camera12"], gpu.make_params(64, 36, 2, 4, 5, 5, 0, flags=gpu.SEM_METAL, seed=1), 0, sums, None, "f32")
    with pytest.raises(gpu.SpiraError, match="rng_states"):
        gpu.accumulate(s["spheres5"], s["materials8"], None, s["camera12"], gpu.make_params(64, 36, 2, 4, 5, 5, 0, flags=gpu.SEM_METAL, seed=1), 2, sums, None, "f32")


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_host_outputs_through_the_staged_copy_equal_device_outputs(gpu, prec):
    """Frames of 4 MB and more reach a host-pointer caller through pinned staging in chunks, moved on by several host threads (copy_out): both outputs, a
    size that is no multiple of the chunk, any thread count / chunk size — bit for bit what the device-output entry leaves on the device; a small frame
    takes the plain copy."""
    import torch
    s = scenes.scene_s2()
    ns, nm, nt = _counts(s)
    tdt = torch.float64 if prec == "f64" else torch.float32
    for W, H in ((1500, 1101), (160, 90)):
        p = gpu.make_params(W, H, 2, 4, ns, nm, nt, seed=9)
        d_hdr = torch.empty((3, H, W), dtype=tdt, device="cuda")
        d_img = torch.empty((3, H, W), dtype=tdt, device="cuda")
        gpu.render_device(*_args(s), p, d_hdr.data_ptr(), d_img.data_ptr(), torch.cuda.current_stream().cuda_stream, prec)
        torch.cuda.synchronize()
        for env in ({}, {"SPIRA_STAGE_THREADS": "1"}, {"SPIRA_STAGE_THREADS": "7", "SPIRA_STAGE_CHUNK_MB": "3"}, {"SPIRA_STAGE_CHUNK_MB": "1000"}):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                hdr, img = gpu.render(*_args(s), p, prec, want_img=True)
                only_img = gpu.render(*_args(s), p, prec, want_hdr=False, want_img=True)[1]
            finally:
                for k, v in old.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            assert np.array_equal(hdr, d_hdr.cpu().numpy()) and np.array_equal(img, d_img.cpu().numpy()) and np.array_equal(only_img, img), (W, H, env)


def test_concurrent_host_threads_share_one_device_context(gpu):
    """Four host threads render at once on the same device — both precisions (two translation units of the library), host-array and scene-handle entries,
    different image sizes — and one of them keeps provoking validation errors: the context's mutex serialises the workspaces, the error string and the
    counters stay per thread / per call, and every image equals the one the same call gives alone."""
    import threading
    jobs = []
    for i, (builder, prec, W, H, spp, depth) in enumerate(((scenes.scene_s1, "f32", 1500, 1101, 1, 4), (scenes.scene_s2, "f64", 1100, 700, 1, 4),      # (two frames big enough for the staged host copy)
                                                            (lambda: scenes.scene_s4(level=3), "f32", 128, 72, 2, 7), (scenes.scene_s3, "f64", 64, 36, 2, 8))):
        s = builder()
        p = gpu.make_params(W, H, spp, depth, *_counts(s), seed=40 + i)
        ref, _ = gpu.render(*_args(s), p, prec)
        jobs.append((s, prec, p, ref))
    errors = []

    def worker(k):
        s, prec, p, ref = jobs[k]
        try:
            h = gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec) if k % 2 else None
            for it in range(8):
                hdr = (h.render(s["camera12"], p) if h else gpu.render(*_args(s), p, prec))[0]
                if not np.array_equal(hdr, ref):
                    errors.append((k, it, "image differs"))
                if k == 0:                               # an error on this thread must leave the others' calls and messages alone
                    try:
                        gpu.render(*_args(s), gpu.make_params(1, 10, 1, 1, *_counts(s)), prec)
                        errors.append((k, it, "no error raised"))
                    except gpu.SpiraError as e:
                        if "width and height" not in str(e):
                            errors.append((k, it, str(e)))
            if h:
                h.destroy()
        except Exception as e:                           # noqa: BLE001 (reported below)
            errors.append((k, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_plain_flags_library():
    """The Makefile drops its LLVM-internal flag when hipcc rejects it (csrc/Makefile, MAIN_LLVM_OK); `make plain` builds that fallback library on
    purpose — the three translation units with plain flags only.  It must load, carry the same ABI and pass the smoke checks (HIP path == oracle,
    spheres and the mesh path, both precisions, the three organisations): speed is what the special flags buy, never results."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "julia-spira_amd", "csrc")
    # built by __graft_entry__.build() and travelled with the snapshot; never compiled here (three translation units on a cold GPU box take minutes)
    assert subprocess.run(["make", "-q", "-C", csrc, "plain"]).returncode == 0, "libspira_hip_plain.so is missing or older than its sources: run __graft_entry__.build()"
    lib = os.path.join(csrc, "libspira_hip_plain.so")
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; from spira_hip import _binding as B; "
            "assert B.LIB_PATH.endswith('libspira_hip_plain.so') and B.build_id() == 'plain-flags', (B.LIB_PATH, B.build_id()); g.smoke()" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SPIRA_HIP_LIB=lib), timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_host_output_frames_render_as_row_slabs(gpu, monkeypatch):
    """A large frame for a host-pointer caller is rendered as 4 row slabs so that a slab's copy to the host runs beside the next slab's kernels
    (spira_hip.hip, render_host_slabs).  The slabs are, bit for bit, the rows of the frame rendered whole (SPIRA_HOST_SLABS=0), for every organisation,
    row order, output selection and a sub-tile; and the call's counters add up over its slabs."""
    from spira_hip import _binding as B
    cases = [
        (scenes.scene_s1, "f64", 1024, 512, 32, 6, dict(flags=B.KERNEL_WAVEFRONT), True, True),
        (scenes.scene_s1, "f32", 1536, 768, 16, 5, dict(flags=B.KERNEL_WAVEFRONT | B.ROWS_BOTTOM_UP | B.POST_ACES_GAMMA), False, True),     # display image only
        (scenes.scene_s3, "f32", 1536, 770, 16, 4, dict(flags=B.KERNEL_MEGA), True, True),                                                # rows % 4 != 0
        (scenes.scene_s2, "f64", 1024, 600, 32, 4, dict(flags=B.KERNEL_BOUNCE, row0=37, rows=523), True, False),                          # a sub-tile, HDR only
        (lambda: scenes.scene_s4(level=3), "f32", 1536, 768, 16, 6, dict(flags=B.KERNEL_WAVEFRONT), True, True),                          # mesh: two launches per pass
        (scenes.scene_s1, "f64", 1024, 512, 32, 4, dict(flags=B.KERNEL_WAVEFRONT | B.SEM_METAL), True, True),
    ]
    for mk, prec, W, H, spp, depth, kw, want_hdr, want_img in cases:
        s = mk()
        ns, nm, nt = _counts(s)
        p = gpu.make_params(W, H, spp, depth, ns, nm, nt, seed=12, **kw)
        out = {}
        for slabs in ("0", "4"):
            monkeypatch.setenv("SPIRA_HOST_SLABS", slabs)
            hdr, img = gpu.render(*_args(s), p, prec, want_hdr=want_hdr, want_img=want_img)
            out[slabs] = (hdr, img, gpu.counters())
        (h0, i0, c0), (h4, i4, c4) = out["0"], out["4"]
        if want_hdr:
            assert np.array_equal(h0, h4), (mk, prec)
        if want_img:
            assert np.array_equal(i0, i4), (mk, prec)
        for k in ("samples", "segments", "rays_parked", "radiance_stores", "radiance_rmw"):      # (what is queued depends on who shares a wave: not an invariant)
            assert c0[k] == c4[k], (k, c0[k], c4[k])
        # (mesh scenes are never split: four small mesh passes have four fat-wave tails, more than the copy hides)
        n_slabs = 1 if nt > 32 else 4
        assert c4["passes"] == n_slabs * c0["passes"] and (c4["launches"] > c0["launches"]) == (n_slabs > 1) and c4["kernel_ms"] > 0
