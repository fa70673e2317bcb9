"""TIFF planes for the files the pipeline exchanges: written through the native writer in libdelivr_hip.so
(csrc/tiffio.hip; host code, no GPU needed) as classic little-endian TIFF with LZW strips - what the reference's
``tifffile.imwrite(..., compression='lzw')`` calls produce (blob_highlighter.py:131-133, :160) - or uncompressed."""
from __future__ import annotations

import ctypes as C

import numpy as np


def write_tiff_plane(path: str, plane: np.ndarray, compression: str = "lzw") -> None:
    from . import _lib

    plane = np.ascontiguousarray(plane)
    if plane.ndim != 2 or plane.dtype not in (np.uint8, np.uint16):
        raise TypeError("write_tiff_plane: 2-D uint8 / uint16 arrays only")
    if plane.dtype.byteorder == ">":
        plane = plane.astype(plane.dtype.newbyteorder("<"))
    comp = {"lzw": 5, None: 1, "none": 1}[compression]
    lib = _lib.load()
    h, w = plane.shape
    rc = lib.dlv_tiff_write_plane(path.encode(), plane.ctypes.data_as(C.c_void_p), h, w, 8 * plane.dtype.itemsize, comp)
    if rc != 0:
        raise OSError(lib.dlv_tiff_last_error().decode())
