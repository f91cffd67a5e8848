"""GPU (MI355X): spira_render_multi_* — the multi-device entry of the C ABI (one host thread + stream per device inside the
library, one RCCL gather to device 0).  The GPU box has ONE device: n_devices = 1 runs the real RCCL path (dlopen, communicator,
grouped ncclSend/ncclRecv to itself); SPIRA_MULTI_REHEARSE=1 runs the tiling, re-pitching and reassembly for any n with the
ranks taking turns on device 0.  N > 1 on separate GPUs is run by the driver's scaling bench only."""
import os

import numpy as np
import pytest

from spira_hip import scenes
from test_gpu_parity import _args, _counts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_multi_one_device_through_rccl_equals_render(gpu, prec):
    s = scenes.scene_s2()
    ns, nm, nt = _counts(s)
    p = gpu.make_params(200, 117, 5, 6, ns, nm, nt, seed=12)
    hdr, img = gpu.render(*_args(s), p, prec, want_img=True)
    os.environ.pop("SPIRA_MULTI_REHEARSE", None)
    for _ in range(2):       # second call reuses the communicator
        mh, mi = gpu.render_multi(*_args(s), p, 1, prec, want_img=True)
        assert np.array_equal(mh, hdr) and np.array_equal(mi, img)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_multi_rehearsal_any_n_is_bit_identical(gpu, n):
    """Ragged heights (117 and 200 rows over 2, 3, 8 devices) so that ranks own different numbers of rows."""
    s = scenes.scene_s4(level=3)
    ns, nm, nt = _counts(s)
    os.environ["SPIRA_MULTI_REHEARSE"] = "1"
    try:
        for prec, (W, H) in (("f32", (200, 117)), ("f64", (64, 200))):
            p = gpu.make_params(W, H, 3, 5, ns, nm, nt, seed=5)
            hdr, img = gpu.render(*_args(s), p, prec, want_img=True)
            mh, mi = gpu.render_multi(*_args(s), p, n, prec, want_img=True)
            assert np.array_equal(mh, hdr) and np.array_equal(mi, img), (n, prec)
    finally:
        os.environ.pop("SPIRA_MULTI_REHEARSE", None)


def test_multi_argument_errors(gpu):
    s = scenes.scene_s1()
    os.environ.pop("SPIRA_MULTI_REHEARSE", None)
    with pytest.raises(gpu.SpiraError):
        gpu.render_multi(*_args(s), gpu.make_params(64, 36, 1, 2, 5, 5, 0), gpu.device_count() + 1)
    with pytest.raises(gpu.SpiraError):
        gpu.render_multi(*_args(s), gpu.make_params(64, 36, 1, 2, 5, 5, 0, row0=0, rows=8), 1)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_multi_scene_handle_one_device_through_rccl(gpu, prec):
    """spira_scene_create_multi_* + spira_render_multi_scene_*: validated / built once, resident per device; the real RCCL path with one device."""
    s = scenes.scene_s4(level=3)                    # 1 280 triangles: through the BVH
    ns, nm, nt = _counts(s)
    p = gpu.make_params(160, 90, 4, 6, ns, nm, nt, seed=3)
    hdr, img = gpu.render(*_args(s), p, prec, want_img=True)
    os.environ.pop("SPIRA_MULTI_REHEARSE", None)
    with gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec, n_devices=1) as h:
        for _ in range(2):
            mh, mi = h.render_multi(s["camera12"], p, want_img=True)
            assert np.array_equal(mh, hdr) and np.array_equal(mi, img)
        sh, _ = h.render(s["camera12"], p)          # the handle is device 0's: the single-device entry points take it too
        assert np.array_equal(sh, hdr)
        with pytest.raises(gpu.SpiraError):         # resident on one device only
            h.render_multi(s["camera12"], p, n_devices=2)
    with pytest.raises(gpu.SpiraError):
        gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec, n_devices=gpu.device_count() + 1)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_multi_scene_handle_rehearsal_any_n(gpu, n):
    s = scenes.scene_s4(level=3)
    ns, nm, nt = _counts(s)
    os.environ["SPIRA_MULTI_REHEARSE"] = "1"
    try:
        for prec, (W, H) in (("f32", (200, 117)), ("f64", (64, 200))):
            p = gpu.make_params(W, H, 3, 5, ns, nm, nt, seed=5)
            hdr, img = gpu.render(*_args(s), p, prec, want_img=True)
            with gpu.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec, n_devices=n) as h:
                mh, mi = h.render_multi(s["camera12"], p, want_img=True)
            assert np.array_equal(mh, hdr) and np.array_equal(mi, img), (n, prec)
    finally:
        os.environ.pop("SPIRA_MULTI_REHEARSE", None)
