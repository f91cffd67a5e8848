"""Minimal PNG writer (stdlib zlib) — the step after the path: `save(output_path, img)`
(src/spira-metal-optimized.jl:1486).  No Images/FileIO-like dependency."""
import struct
import zlib

import numpy as np


def save_png(path, img):
    """img: (H, W, 3) floats in [0, 1], row 0 = top."""
    a = np.asarray(img, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) image")
    u8 = (np.clip(np.nan_to_num(a), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    h, w, _ = u8.shape
    raw = b"".join(b"\x00" + u8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
