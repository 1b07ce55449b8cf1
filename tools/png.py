"""Minimal PNG writer (zlib + struct) for eyeballing renders; no third-party imaging library needed."""
import struct
import zlib

import numpy as np


def save_png(path, rgb_linear, exposure=1.0):
    """rgb_linear: HxWx3 float linear sRGB; applies exposure, the sRGB OETF and 8-bit quantisation."""
    x = np.clip(np.asarray(rgb_linear, dtype=np.float64) * exposure, 0.0, 1.0)
    x = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)
    img = (x * 255.0 + 0.5).astype(np.uint8)
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
