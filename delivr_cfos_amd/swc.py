"""Cell table -> SWC files exactly as the atlas-alignment step consumes them.

Mirror of ``rewrite_swc`` / ``split_parameters`` in the reference's ``automate_mBrainaligner.py``
(:75-197, :199-213): the CSV written by ``count_blobs`` is re-read, ``Coords`` "[z, y, x]" is split, values
are rounded to 3 decimals, and rows ``<row> 1 <x> <y> <z> <Size> -1`` are written under the header
``##n type x y z radius parent`` - in one file, or in ceil(n / (cpu_count - 1))-row chunks named
``<csv>chunk_<first row, 7 digits>.swc``.  File names lose blanks and brackets (the external mBrainAligner
binaries trip over them).  Pure host formatting; pinned by tests/golden/ref_swc.npz (the reference's own
function run under stubs).
"""
from __future__ import annotations

import math
import os
import re
from typing import List, Optional, Sequence


def split_parameters(file_path: str) -> List[int]:
    """"(Z, Y, X)_brain.csv..." -> [Z, Y, X]  (reference :199-213)."""
    filename = os.path.split(file_path)[1]
    params = re.findall(r"\(([^)]+)", filename)
    return [int(v) for v in str(params[0]).replace(" ", "").split(",")]


def _fmt(v: float) -> str:
    # pandas' to_csv writes floats with repr(); round(…, 3) as DataFrame.round does (numpy rounding)
    import numpy as np

    return repr(float(np.round(np.float64(v), 3)))


def _parse_csv(csv_path: str):
    rows = []
    with open(csv_path) as fh:
        header = fh.readline()
        if header.strip() != ",Blob,Coords,Size":
            raise ValueError(f"{csv_path}: not a count_blobs cell table")
        for line in fh:
            m = re.match(r'^\d+,(\d+),"\[([^\]]*)\]",(\d+)\s*$', line)
            if not m:
                raise ValueError(f"{csv_path}: cannot parse line {line!r}")
            z, y, x = (float(t) for t in m.group(2).replace(",", " ").split())
            rows.append((z, y, x, int(m.group(3))))
    return rows


def rewrite_swc(csv_path: str, output_dir: str, XYZ: bool = False, parallel_processing: bool = False,
                n_chunks: Optional[int] = None) -> List[str]:
    """Same parameters and return value (list of written .swc paths) as the reference; ``n_chunks``
    overrides os.cpu_count() - 1 for reproducible chunking."""
    rows = _parse_csv(csv_path)
    lines = []
    for i, (a, b, c, size) in enumerate(rows):
        z, y, x = (a, b, c) if not XYZ else (c, b, a)
        lines.append(f"{i} 1 {_fmt(x)} {_fmt(y)} {_fmt(z)} {size} -1\n")

    def clean(name: str) -> str:
        return name.replace(" ", "").replace("(", "").replace(")", "")

    base = os.path.split(csv_path)[1]
    written = []
    if parallel_processing:
        n_chunks = n_chunks if n_chunks is not None else (os.cpu_count() or 2) - 1
        chunk_length = int(round(math.ceil(len(lines) / max(n_chunks, 1))))
        for first in range(0, len(lines), max(chunk_length, 1)):
            name = clean(os.path.join(output_dir, base + "chunk_" + str(first).zfill(7) + ".swc"))
            with open(name, "w") as fh:
                fh.write("##n type x y z radius parent\n")
                fh.writelines(lines[first:first + chunk_length])
            written.append(name)
        return written
    name = clean(os.path.join(output_dir, base + ".swc"))
    with open(name, "w") as fh:
        fh.write("##n type x y z radius parent\n")
        fh.writelines(lines)
    return [name]


def sampling_factors(csv_or_swc_path: str, downsampled_shape_zyx: Sequence[int], XYZ: bool = False):
    """(ds_factor_x, ds_factor_y, ds_factor_z) = original / downsampled per axis, original read from the file
    name (reference compute_sampling_factors :261-284; the TIFF read there only supplies the shape)."""
    p = split_parameters(csv_or_swc_path)
    oz, oy, ox = (p[0], p[1], p[2]) if not XYZ else (p[2], p[1], p[0])
    dz, dy, dx = (int(v) for v in downsampled_shape_zyx)
    return ox / dx, oy / dy, oz / dz
