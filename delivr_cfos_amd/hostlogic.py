"""Host-side integer logic of the path that the reference performs in Python (no device work)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def padded_shape(stack_shape: Sequence[int], crop: Sequence[int]) -> Tuple[int, ...]:
    """ceil(dim / crop) * crop per axis (inference/inference.py:229-231;
    downsample/downsample_and_mask.py:390-393)."""
    return tuple(int(np.ceil(int(n) / int(c)) * int(c)) for n, c in zip(stack_shape, crop))


def arrayterator_zblock(shape_zyx: Sequence[int], buf_size: int = 1000**3) -> int:
    """Planes per block of np.lib.Arrayterator(volume[(Z,Y,X)], buf_size) - the granularity at which
    the reference erodes the re-mask (inference/inference.py:53,77-84).  Returns Z when the whole
    volume fits one block."""
    Z, Y, X = (int(v) for v in shape_zyx)
    count = int(buf_size)
    if count <= X or count // X <= Y:
        raise NotImplementedError("a single z-plane exceeds the Arrayterator buffer (Y*X > buf_size)")
    count = (count // X) // Y
    return Z if count >= Z else max(count, 1)


def pass_schedule(tta: bool) -> List[Tuple[Optional[int], int]]:
    """(flip_dim, repeat) per DISTINCT pass.  The reference runs 1 plain pass, then 4 x {noise,
    noise + flip Z (dim 2), noise + flip Y (dim 3)} (inference/inference.py:261-279); its noise is
    N(0, <=1e-3) on raw uint16-scale intensities (sliding_window_inferer.py:212-215), i.e. nil, so the
    13 passes collapse to plain x5, flipZ x4, flipY x4."""
    if not tta:
        return [(None, 1)]
    return [(None, 5), (2, 4), (3, 4)]


def cells_csv_text(stats: dict, n: int) -> str:
    """The text pandas writes for the reference's cell table (count_blobs.py:98-114): header
    ``,Blob,Coords,Size``; one row per label 1..N-1 (the reference's ``range(1, N)`` drops the last
    label); index column always 0; Coords = python repr of [z, y, x] floats, quoted by the CSV
    writer because it contains commas."""
    cent = np.asarray(stats["centroids"], dtype=np.float64)[1:max(int(n), 1)].tolist()  # (python floats: repr as pandas writes them)
    counts = np.asarray(stats["voxel_counts"])[1:max(int(n), 1)].tolist()
    lines = [",Blob,Coords,Size"]
    lines.extend(f'0,{i},"{c!r}",{k}' for i, (c, k) in enumerate(zip(cent, counts), 1))
    return "\n".join(lines) + "\n"


def cells_csv_bytes(stats: dict, n: int) -> bytes:
    """cells_csv_text through the library's host-side writer (dlv_cells_csv: C, ~0.1 s for 540 k cells instead of 0.8 s of Python
    string formatting - the same text, compared case by case in tests/test_host_cpu.py)"""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    counts = np.ascontiguousarray(stats["voxel_counts"], dtype=np.uint32)
    cent = np.ascontiguousarray(stats["centroids"], dtype=np.float64)
    n = int(n)
    if len(counts) < n or cent.shape[0] < n or cent.shape[1:] != (3,):
        raise ValueError("statistics shorter than the label count")
    buf = C.create_string_buffer(64 + 128 * max(n, 1))
    ln = C.c_size_t()
    rc = lib.dlv_cells_csv(counts.ctypes.data_as(C.c_void_p), cent.ctypes.data_as(C.c_void_p), n, buf, len(buf), C.byref(ln))
    if rc != 0:
        raise _lib.DelivrHipError(rc, "dlv_cells_csv failed")
    return buf.raw[: ln.value]


def csv_name(shape_zyx: Sequence[int], brain: str) -> str:
    """f"{bin_img.shape}_{brain}.csv" (count_blobs.py:113): "(Z, Y, X)_<brain>.csv"."""
    return f"{tuple(int(v) for v in shape_zyx)}_{brain.replace('.nii.gz', '')}.csv"


def scale_cell_coords(coords_zyx, original_shape: Sequence[int], downsampled_shape: Sequence[int], direction: str = "down"):
    """Cell-coordinate scaling of automate_mBrainaligner.py:261-284: factor = original/downsampled
    per axis; "down" divides (original -> atlas space), "up" multiplies."""
    f = np.asarray(original_shape, dtype=np.float64) / np.asarray(downsampled_shape, dtype=np.float64)
    c = np.asarray(coords_zyx, dtype=np.float64)
    return c / f if direction == "down" else c * f


def downsample_ratios(steps: dict) -> Tuple[int, int, int]:
    """(z, y, x) integer block-mean factors from config.json's mask_detection.downsample_steps
    (downsample/downsample_and_mask.py:161-163): round(downsample_um / original_um)."""
    return (round(steps["downsample_um_z"] / steps["original_um_z"]),
            round(steps["downsample_um_y"] / steps["original_um_y"]),
            round(steps["downsample_um_x"] / steps["original_um_x"]))


def padded_boxes(bounding_boxes: np.ndarray, cc_ids, shape_zyx, times: int = 1) -> np.ndarray:
    """Half-open slices the reference paints for the listed cells (blob_highlighter.py:18-23, :112-113): the inclusive
    cc3d box with every upper end moved up by one, `times` times, each time only while it is still below the axis
    length (pad_bb mutates the statistics in place, so the region-id loop - which runs after the RGB loop - sees boxes
    that were already padded once: times=2).  Returns (n,6) int32 [z0,z1,y0,y1,x0,x1]."""
    bb = np.asarray(bounding_boxes)[np.asarray(cc_ids, dtype=np.int64)].astype(np.int64)
    dims = np.asarray(shape_zyx, dtype=np.int64)
    out = np.empty((len(bb), 6), dtype=np.int64)
    out[:, 0::2] = bb[:, 0::2]
    hi = bb[:, 1::2].copy()
    for _ in range(times):
        hi = np.where(hi < dims[None, :], hi + 1, hi)
    out[:, 1::2] = hi
    return out.astype(np.int32)


def max_window_multiplicity(starts, roi) -> int:
    """Largest number of windows covering one voxel, for the (n,3) window starts of the reference's enumeration
    (inference/sliding_window_inferer.py:143-145) and the window size: the peak of the count map after one pass."""
    import numpy as np

    starts = np.asarray(starts)
    mult = 1
    for k in range(3):
        st = np.unique(starts[:, k])
        ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([st + int(roi[k]), -np.ones_like(st)], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]  # closings (-1) before openings (+1) at equal coordinates
        mult *= int(np.cumsum(ev[:, 1]).max())
    return mult


def affine_apply(matrix34, coords_zyx):
    """Maps (n,3) cell coordinates (z,y,x) found in a volume produced by HipEngine.affine_warp_u16 back into the index
    space of its input: c_in = M . (c_out, 1) - the coordinate leg of BASELINE config 5 (the reference moves cell
    coordinates, never volumes: automate_mBrainaligner.py:261-284 scales them, the registration binaries warp them)."""
    import numpy as np

    m = np.asarray(matrix34, dtype=np.float64).reshape(3, 4)
    c = np.asarray(coords_zyx, dtype=np.float64).reshape(-1, 3)
    return c @ m[:, :3].T + m[:, 3]
