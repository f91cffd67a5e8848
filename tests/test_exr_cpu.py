"""EXR output step (examples/julia-raytracer.jl:424-463 save_exr): write, read back bit-exactly, check the header."""
import struct

import numpy as np
import pytest

from spira_hip import exr


def test_exr_roundtrip_bit_exact(tmp_path):
    rng = np.random.default_rng(3)
    hdr = (rng.random((7, 13, 3)) * 40.0).astype(np.float32)
    hdr[0, 0] = [0.0, 1e-30, 6.5e4]
    p = tmp_path / "a.exr"
    exr.save_exr(str(p), hdr)
    back = exr.load_exr(str(p))
    assert back.dtype == np.float32 and back.shape == hdr.shape
    assert np.array_equal(back.view(np.uint32), hdr.view(np.uint32))


def test_exr_header_layout(tmp_path):
    p = tmp_path / "b.exr"
    exr.save_exr(str(p), np.zeros((2, 3, 3), np.float64))
    blob = p.read_bytes()
    assert struct.unpack_from("<i", blob, 0)[0] == 20000630 and blob[4] == 2
    for key in (b"channels\0chlist\0", b"compression\0compression\0", b"dataWindow\0box2i\0", b"lineOrder\0lineOrder\0"):
        assert key in blob
    # 2 scanlines x (8-byte line header + 3 channels x 3 px x 4 B) after header + 2 offsets
    assert len(blob) == blob.index(b"screenWindowWidth") + len("screenWindowWidth\0float\0") + 4 + 4 + 1 + 2 * 8 + 2 * (8 + 36)


def test_exr_rejects_bad_shape(tmp_path):
    with pytest.raises(ValueError):
        exr.save_exr(str(tmp_path / "c.exr"), np.zeros((4, 4)))
