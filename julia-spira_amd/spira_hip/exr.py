"""Minimal OpenEXR writer/reader (scanline, uncompressed, 32-bit float R/G/B) — the HDR output step after the path:
`save_exr(hdr_data, filename)` of examples/julia-raytracer.jl:424-463 ("32-bit EXR").  No OpenEXR/FileIO dependency."""
import struct

import numpy as np

_MAGIC = 20000630


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


def save_exr(path, hdr):
    """hdr: (H, W, 3) linear radiance, row 0 = image top.  Channels are stored as FLOAT (32-bit), no compression."""
    a = np.ascontiguousarray(hdr, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) image")
    h, w, _ = a.shape
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in ("B", "G", "R")) + b"\0"   # 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", _MAGIC, 2)
              + _attr("channels", "chlist", chlist)
              + _attr("compression", "compression", b"\0")
              + _attr("dataWindow", "box2i", box)
              + _attr("displayWindow", "box2i", box)
              + _attr("lineOrder", "lineOrder", b"\0")                    # increasing y: scanline 0 first (image top)
              + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
              + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
              + b"\0")
    line_bytes = 3 * w * 4
    table_pos = len(header)
    data_pos = table_pos + 8 * h
    offsets = struct.pack("<%dQ" % h, *[data_pos + y * (8 + line_bytes) for y in range(h)])
    with open(path, "wb") as f:
        f.write(header)
        f.write(offsets)
        for y in range(h):
            f.write(struct.pack("<ii", y, line_bytes))
            f.write(a[y, :, 2].tobytes())     # channels in alphabetical order: B, G, R
            f.write(a[y, :, 1].tobytes())
            f.write(a[y, :, 0].tobytes())


def load_exr(path):
    """Reads back what save_exr writes (uncompressed FLOAT B/G/R scanlines) -> (H, W, 3) float32."""
    blob = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", blob, 0)
    if magic != _MAGIC or (version & 0xFF) != 2:
        raise ValueError("not an OpenEXR 2 file")
    pos, attrs = 8, {}
    while blob[pos] != 0:
        e = blob.index(b"\0", pos)
        name = blob[pos:e].decode()
        e2 = blob.index(b"\0", e + 1)
        typ = blob[e + 1:e2].decode()
        size, = struct.unpack_from("<i", blob, e2 + 1)
        attrs[name] = (typ, blob[e2 + 5:e2 + 5 + size])
        pos = e2 + 5 + size
    pos += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("only uncompressed files")
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offsets = struct.unpack_from("<%dQ" % h, blob, pos)
    out = np.empty((h, w, 3), dtype=np.float32)
    for off in offsets:
        y, nbytes = struct.unpack_from("<ii", blob, off)
        line = np.frombuffer(blob, dtype="<f4", count=3 * w, offset=off + 8).reshape(3, w)
        out[y - y0, :, 2], out[y - y0, :, 1], out[y - y0, :, 0] = line[0], line[1], line[2]
    return out
