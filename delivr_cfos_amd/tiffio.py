"""Baseline TIFF planes (little-endian, one uncompressed strip, 8/16-bit grayscale) - enough for the z-plane files the
pipeline exchanges.  The reference writes its planes with tifffile + LZW (blob_highlighter.py:131-133); LZW is a
codec, not part of the accelerated path: the planes written here decode to the same pixels in any reader."""
from __future__ import annotations

import struct

import numpy as np


def write_tiff_plane(path: str, plane: np.ndarray) -> None:
    plane = np.ascontiguousarray(plane)
    if plane.ndim != 2 or plane.dtype not in (np.uint8, np.uint16):
        raise TypeError("write_tiff_plane: 2-D uint8 / uint16 arrays only")
    h, w = plane.shape
    bits = 8 * plane.dtype.itemsize
    data = plane.astype(plane.dtype.newbyteorder("<"), copy=False).tobytes()
    tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8),
            (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, len(data)), (339, 3, 1, 1)]
    ifd_off = 8 + len(data) + (len(data) & 1)
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, ifd_off))
        fh.write(data)
        if len(data) & 1:
            fh.write(b"\0")
        fh.write(struct.pack("<H", len(tags)))
        for tag, typ, cnt, val in tags:
            fh.write(struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val)))
        fh.write(struct.pack("<I", 0))
