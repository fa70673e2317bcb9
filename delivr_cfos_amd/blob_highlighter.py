"""Mirror of the reference's ``blob_highlighter.py`` (blob_highlighter :38-169): colour every detected cell by its atlas
region, on the device.  Same call, same inputs, same plane files:

in : <input_prediction_location>/<dir containing brain>/binary_segmentations/binaries.npy  ('|u1', (Z,Y,X), 128-byte header)  (:62)
     <input_csv_location>/cells_<brain>*.csv  columns connected_component_id, acronym, red, green, blue, graph_order          (:66-71)
     <postprocessing.output_location>/<brain>-stats.pickle when present, else CCL-26 + statistics on the device              (:78-92)
out: <output_location>/<brain>_rgb_tiffs/<brain>rgb_C0{0,1,2}_z####.tif     uint8   (region_id_rgb, :94-133)
     <output_location>/<brain>/<brain>_region_id_tiffs/region_id_####.tif   uint16  (region_id_grayvalues, :137-161)

The per-cell loop (bounding box x colour, later boxes overwrite earlier ones) is one scatter + one gather on the GPU
(csrc/paint.hip); the box arithmetic, including pad_bb's in-place mutation of the statistics between the two loops, is
hostlogic.padded_boxes.  Planes are written as LZW-compressed TIFF like the reference's (native writer, tiffio.py).
"""
from __future__ import annotations

import csv
import datetime
import os
import pickle

import numpy as np

from .hostlogic import padded_boxes
from .tiffio import write_tiff_plane


def load_cached_stats(settings, brain):
    """reference :25-36"""
    from .count_blobs import _find_cached

    return _find_cached(settings["postprocessing"]["output_location"], ".pickle", brain)


def read_cell_table(path: str) -> dict:
    """pd.read_csv(path, index_col=0) + the 'bgr' filter (:70-71) without pandas: columns as numpy arrays."""
    with open(path, newline="") as fh:
        rows = list(csv.reader(fh))
    head = rows[0]
    col = {name: i for i, name in enumerate(head)}
    for need in ("connected_component_id", "acronym", "red", "green", "blue", "graph_order"):
        if need not in col:
            raise KeyError(f"{path}: column {need!r} missing")
    body = [r for r in rows[1:] if r and r[col["acronym"]] != "bgr"]

    def ints(name):
        return np.array([int(float(r[col[name]])) for r in body], dtype=np.int64)

    return {"connected_component_id": ints("connected_component_id"), "red": ints("red"), "green": ints("green"),
            "blue": ints("blue"), "graph_order": ints("graph_order")}


def blob_highlighter(settings, brain_item, stack_shape, engine=None):
    """Same positional parameters as the reference (:38).  ``engine``: a HipEngine to reuse."""
    import torch

    from .engine import HipEngine

    brain = brain_item[0]
    viz = settings["visualization"]
    if viz.get("no_atlas_depthmap"):
        # "only map the blobs over their distance from the sample's outside" (:163-165; no atlas, hence no cell table -
        # the reference skips loading it at :67-69).  Its depth_map_blobs cannot run (IndexError at blob_depthmap.py:139);
        # the mirror implements what the function states after that line (blob_depthmap.py in this package)
        from .blob_depthmap import depth_map_blobs

        depth_map_blobs(settings, brain, stack_shape, engine=engine)
        print(f"{datetime.datetime.now()} : Cleanup")
        return
    path_binary = viz["input_prediction_location"]
    path_cell_csv = viz["input_csv_location"]
    path_out = viz["output_location"]
    path_out_rgb = os.path.join(path_out, brain + "_rgb_tiffs")
    os.makedirs(path_out_rgb, exist_ok=True)
    path_brain_binary = path_binary + [x for x in os.listdir(path_binary) if brain in x][0] + "/binary_segmentations/binaries.npy"
    path_brain_cell_csv = path_cell_csv + [x for x in os.listdir(path_cell_csv) if "cells_" + brain in x and ".csv" in x][0]
    print(path_brain_cell_csv)
    print(f"{datetime.datetime.now()} : Loading csv")
    cells = read_cell_table(path_brain_cell_csv)
    ids = cells["connected_component_id"]
    if len(np.unique(ids)) != len(ids):
        raise ValueError("connected_component_id values must be unique (the reference's broadcast fails on duplicates, :156)")
    print(f"{datetime.datetime.now()} : Loading brain")
    shape = tuple(int(v) for v in stack_shape[2:])
    bin_img = np.memmap(path_brain_binary, dtype=np.uint8, mode="r", shape=shape, offset=128)
    own = engine is None
    eng = engine or HipEngine(0)
    try:
        bin_dev = eng.to_device(np.ascontiguousarray(bin_img))
        cached = load_cached_stats(settings, brain)
        if not cached:
            labels, n = eng.ccl26(bin_dev)
            stats = eng.cc_stats(labels, n)
            del labels
        else:
            print(f"Found stats at {cached}")
            with open(cached, "rb") as fh:
                stats = pickle.load(fh)
        bboxes = np.asarray(stats["bounding_boxes"])
        if len(ids) and int(ids.max()) >= len(bboxes):
            raise IndexError("connected_component_id beyond the statistics table")
        pads = 0
        if viz.get("region_id_rgb"):
            print(f"{datetime.datetime.now()} : coloring blobs")
            pads += 1
            boxes = padded_boxes(bboxes, ids, shape, pads)
            chans = eng.paint_boxes(bin_dev, boxes, [cells[c].astype(np.uint8) for c in ("red", "green", "blue")])
            print(f"{datetime.datetime.now()} : Generating RGB tiffs")
            for c, img in enumerate(chans):
                host = img.cpu().numpy()
                for z in range(shape[0]):
                    write_tiff_plane(os.path.join(path_out_rgb, brain + f"rgb_C0{c}_z" + str(z).zfill(4) + ".tif"), host[z])
            del chans
        print(f"{datetime.datetime.now()} : Generating region_id gray-value tiffs")
        if viz.get("region_id_grayvalues"):
            path_out_region_id = os.path.join(path_out, brain, brain + "_region_id_tiffs")
            os.makedirs(path_out_region_id, exist_ok=True)
            pads += 1  # pad_bb already moved these boxes once if the RGB loop ran (in-place mutation of the statistics)
            boxes = padded_boxes(bboxes, ids, shape, pads)
            (rid,) = eng.paint_boxes(bin_dev, boxes, [cells["graph_order"].astype(np.uint16)])
            host = rid.cpu().numpy()
            for z in range(shape[0]):
                write_tiff_plane(os.path.join(path_out_region_id, "region_id_" + str(z).zfill(4) + ".tif"), host[z])
    finally:
        if own:
            eng.close()
    print(f"{datetime.datetime.now()} : Cleanup")
