"""Mirror of the numeric part of the reference's ``cells_to_atlas.py`` (region assignment step): cell coordinates
from mBrainAligner-atlas space to CCF3 voxels, region id per cell, cell-density heat map.

  mbrainaligner_atlas_to_ccf  (:114-151)  flip x -> 264-x, y -> 160-y, swap x<->y, x2, id+1, round, drop out-of-bounds
  region_ids                  (:204-212)  LabelImage[z,y,x], +1 where != 0 (row index into the ontology table)
  create_heatmap              (:174-200)  counts per voxel -> gaussian_filter(float32, sigma=2.25)   [HIP: csrc/paint.hip]

Tables are dicts of equally long numpy arrays (the reference uses pandas DataFrames; a DataFrame's columns can be
passed as ``{c: df[c].to_numpy() for c in df}``).  The ontology join, the per-region tables and the Excel/CSV
summaries (:154-172, :214-240, :330-) are table work outside the accelerated path.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

Table = Dict[str, np.ndarray]


def mbrainaligner_atlas_to_ccf(cells: Table, label_shape) -> Table:
    """cells: columns connected_component_id, x, y, z, Size (any numeric dtype).  label_shape: LabelImage.shape (z,y,x).
    Returns integer columns of the cells that fall inside the label volume, in their original order."""
    x = 264 - np.asarray(cells["x"], dtype=np.float64)
    y = 160 - np.asarray(cells["y"], dtype=np.float64)
    z = np.asarray(cells["z"], dtype=np.float64)
    out = {"connected_component_id": np.asarray(cells["connected_component_id"]) + 1,
           "y": x * 2, "x": y * 2, "z": z * 2}        # the rename x<->y (:123), then x2 (:131)
    for k, v in cells.items():
        if k not in out:
            out[k] = np.asarray(v)
    out = {k: np.round(np.asarray(v, dtype=np.float64)).astype(np.int64) for k, v in out.items()}   # DataFrame.round().astype(int)
    Z, Y, X = (int(v) for v in label_shape)
    keep = (out["x"] < X) & (out["y"] < Y) & (out["z"] < Z) & (out["x"] >= 0) & (out["y"] >= 0) & (out["z"] >= 0)
    print("discarded out of bounds cells: ", int((~keep).sum()))
    return {k: v[keep] for k, v in out.items()}


def region_ids(cells: Table, label_image: np.ndarray) -> np.ndarray:
    """Row of the ontology table per cell (cells_to_atlas :204-212)."""
    rid = np.asarray(label_image)[cells["z"], cells["y"], cells["x"]].astype(np.int64)
    rid[rid != 0] += 1
    return rid


def create_heatmap(cells: Table, label_shape, engine=None, sigma: float = 2.25) -> np.ndarray:
    """float32 (z,y,x) blurred cell-density map, bit-identical to the reference's scipy call; computed on the device."""
    from .engine import HipEngine

    own = engine is None
    eng = engine or HipEngine(0)
    try:
        xyz = np.stack([np.asarray(cells["x"]), np.asarray(cells["y"]), np.asarray(cells["z"])], axis=1).astype(np.int32)
        heat = eng.heatmap(xyz, label_shape, sigma)
        return heat.cpu().numpy()
    finally:
        if own:
            eng.close()
