"""GPU (MI355X): progressive accumulation (spira_accumulate_*): k calls of n samples leave, bit for bit, the sums of
one call of k*n samples — for the counter-RNG estimators (sample index offset) and for SPIRA_SEM_METAL (LCG states
carried in rng_states, src/spira_path_trace_kernel.metal:155,:268)."""
import numpy as np
import pytest

from spira_hip import scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("sem", [0x0, 0x1, 0x2], ids=["A", "cpu", "metal"])
def test_progressive_equals_one_shot(gpu, sem, prec):
    s = scenes.scene_s1()
    sp, ma, cam = s["spheres5"], s["materials8"], s["camera12"]
    W, H, depth, total = 160, 90, 6, 24
    npdt = np.float32 if prec == "f32" else np.float64
    one, _ = gpu.render(sp, ma, None, cam, gpu.make_params(W, H, total, depth, 5, 5, 0, flags=sem | gpu.POST_NONE, seed=13), prec)
    for chunks in ([24], [8, 8, 8], [1, 5, 18], [23, 1]):
        sums = np.zeros((3, H, W), dtype=npdt)
        rng = np.zeros(H * W, dtype=np.uint32) if sem == 0x2 else None
        s0 = 0
        for n in chunks:
            gpu.accumulate(sp, ma, None, cam, gpu.make_params(W, H, n, depth, 5, 5, 0, flags=sem, seed=13), s0, sums, rng, prec)
            s0 += n
        assert np.array_equal(sums / npdt(total), one), (sem, chunks)      # division is correctly rounded on both sides


def test_progressive_tiles_and_mesh(gpu):
    from spira_hip import distributed as D
    s = scenes.scene_s4(level=3)
    ns, nm, nt = len(s["spheres5"]), len(s["materials8"]), len(s["triangles10"])
    W, H = 96, 54
    one, _ = gpu.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"],
                        gpu.make_params(W, H, 6, 5, ns, nm, nt, flags=gpu.POST_NONE, seed=2), "f32")
    tiles = []
    for r in range(2):
        tp = D.tile_params(H, 2, r, 4)
        sums = np.zeros((3, tp["rows"], W), dtype=np.float32)
        for s0 in (0, 3):
            gpu.accumulate(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"],
                           gpu.make_params(W, H, 3, 5, ns, nm, nt, seed=2, **tp), s0, sums, None, "f32")
        tiles.append(sums / np.float32(6))
    assert np.array_equal(D.assemble(tiles, H, 2, 4), one)


def test_device_pointer_accumulate(gpu):
    import ctypes as C
    import torch
    s = scenes.scene_s1()
    W, H = 128, 72
    one, _ = gpu.render(s["spheres5"], s["materials8"], None, s["camera12"], gpu.make_params(W, H, 8, 4, 5, 5, 0, flags=gpu.POST_NONE, seed=4), "f32")
    sums = torch.zeros((3, H, W), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream()
    sp = np.ascontiguousarray(s["spheres5"], dtype=np.float32)
    ma = np.ascontiguousarray(s["materials8"], dtype=np.float32)
    cam = np.ascontiguousarray(s["camera12"], dtype=np.float32)
    for s0 in (0, 4):
        p = gpu.make_params(W, H, 4, 4, 5, 5, 0, seed=4)
        rc = gpu.lib().spira_accumulate_device_f32(sp.ctypes.data_as(C.c_void_p), ma.ctypes.data_as(C.c_void_p), None,
                                                   cam.ctypes.data_as(C.c_void_p), C.byref(p), C.c_uint32(s0),
                                                   C.c_void_p(sums.data_ptr()), None, C.c_void_p(st.cuda_stream or None))
        assert rc == 0, gpu.lib().spira_last_error()
    st.synchronize()
    assert np.array_equal((sums / 8).cpu().numpy(), one)
