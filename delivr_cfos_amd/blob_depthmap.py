"""Mirror of the reference's ``blob_depthmap.py`` (depth_map_blobs :114-213): every detected cell painted with its depth
below the brain surface, on the device.

in : <visualization.input_prediction_location>/<dir containing brain>/binary_segmentations/binaries.npy ('|u1', (Z,Y,X))  (:133-139)
     <postprocessing.output_location>/<brain>*.pickle when present, else CCL-26 + statistics on the device             (:141-156)
     <mask_detection.output_location>/<brain>/downsampled_masked_stack.tif + the um-per-voxel settings                 (:158-170)
out: <visualization.output_location>/<brain>/<brain>_depthmap_tiffs/depthmap_####.tif   uint16, LZW                   (:200-204)

The reference function cannot run: it indexes its 3-D memmap with four indices (``bin_img[0,:,:,:]``, :139) and raises
IndexError before anything is computed.  What is mirrored is everything the function states after that line, literally:
depth = Euclidean distance transform of the zero-padded down-sampled masked stack (scipy's, here dlv_edt_u16_dev) read at
the cell's centroid scaled to the down-sampled grid; ``for cc_id in range(N)`` walks the STATISTICS rows - row 0 (the
background, whose box is the whole volume) first, row N never - and IMG[box] = bin_img[box] * depth, later boxes winning
(dlv_paint_owner_dev / dlv_paint_apply_dev, the kernels of blob_highlighter).  The parity oracle is
oracle.delivr_oracle.depth_map_blobs (scipy EDT + the sequential loop); there is no reference output to pin against.
"""
from __future__ import annotations

import datetime
import os
import pickle

import numpy as np

from .blob_highlighter import load_cached_stats
from .hostlogic import padded_boxes
from .tiffio import write_tiff_plane


def depth_values(stats: dict, n: int, distances: np.ndarray, down_um_zyx, orig_um_zyx) -> np.ndarray:
    """:172-178 + :190-191: centroids scaled to the down-sampled grid, astype(int), looked up in the distance map
    (numpy's indexing rules: negative indices wrap, indices beyond the map raise IndexError - as in the reference)."""
    coords = np.asarray(stats["centroids"], dtype=np.float64).copy()
    for k in range(3):
        coords[:, k] = coords[:, k] / (float(down_um_zyx[k]) / float(orig_um_zyx[k]))
    coords = coords.astype(int)
    return np.asarray([distances[coords[c, 0], coords[c, 1], coords[c, 2]] for c in range(n)], dtype=np.uint16)


def depth_map_volume(engine, bin_dev, stats: dict, n: int, masked_stack: np.ndarray, down_um_zyx, orig_um_zyx):
    """The numeric part of depth_map_blobs: returns the uint16 depth-coded image (HBM tensor shaped like bin_dev)."""
    shape = tuple(int(v) for v in bin_dev.shape)
    stack = np.ascontiguousarray(masked_stack)
    if stack.dtype != np.uint16:
        stack = (stack != 0).astype(np.uint16)  # only zero / non-zero matters to the transform
    distances = engine.edt_u16(engine.to_device(stack), down_um_zyx).cpu().numpy()
    depths = depth_values(stats, n, distances, down_um_zyx, orig_um_zyx)
    boxes = padded_boxes(np.asarray(stats["bounding_boxes"]), np.arange(n), shape, 1)
    (img,) = engine.paint_boxes(bin_dev, boxes, [depths])
    return img


def read_masked_stack(path: str) -> np.ndarray:
    """The down-sampled masked stack (:159-167 reads it with tifffile.imread): a multi-page TIFF written by
    skimage/tifffile - read page by page through Pillow (libtiff) - or the same array as ``<name>.npy`` next to it."""
    npy = os.path.splitext(path)[0] + ".npy"
    if os.path.isfile(npy):
        return np.load(npy)
    try:
        from PIL import Image, ImageSequence
    except ImportError as exc:  # pragma: no cover - Pillow is part of the image this runs in
        raise ImportError(f"{path}: reading a multi-page TIFF needs Pillow (or save the stack as {npy})") from exc
    with Image.open(path) as im:
        return np.stack([np.array(page) for page in ImageSequence.Iterator(im)])


def depth_map_blobs(settings, brain, stack_shape, engine=None):
    """Same positional parameters as the reference (:114)."""
    from .engine import HipEngine

    viz = settings["visualization"]
    path_out_depthmap = os.path.join(viz["output_location"], brain, brain + "_depthmap_tiffs")
    os.makedirs(path_out_depthmap, exist_ok=True)
    path_binary = viz["input_prediction_location"]
    path_brain_binary = path_binary + [x for x in os.listdir(path_binary) if brain in x][0] + "/binary_segmentations/binaries.npy"
    print(f"{datetime.datetime.now()} : Loading brain")
    shape = tuple(int(v) for v in stack_shape[2:])
    bin_img = np.memmap(path_brain_binary, dtype=np.uint8, mode="r", shape=shape, offset=128)
    steps = settings["mask_detection"]["downsample_steps"]
    orig = (steps["original_um_z"], steps["original_um_y"], steps["original_um_x"])
    down = (steps["downsample_um_z"], steps["downsample_um_y"], steps["downsample_um_x"])
    own = engine is None
    eng = engine or HipEngine(0)
    try:
        bin_dev = eng.to_device(np.ascontiguousarray(bin_img))
        print(f"{datetime.datetime.now()} : calculating connected-component analysis")
        cached = load_cached_stats(settings, brain)
        if not cached:
            labels, n = eng.ccl26(bin_dev)
            stats = eng.cc_stats(labels, n)
            del labels
        else:
            print(f"Found stats at {cached}")
            with open(cached, "rb") as fh:
                stats = pickle.load(fh)
            n = len(stats["voxel_counts"]) - 1
        print(f"{datetime.datetime.now()} : calculating euclidean distance transform")
        stack_path = os.path.join(settings["mask_detection"]["output_location"], brain, "downsampled_masked_stack.tif")
        masked_stack = read_masked_stack(stack_path)
        print(f"{datetime.datetime.now()} : generating depth-coded blob map")
        img = depth_map_volume(eng, bin_dev, stats, n, masked_stack, down, orig).cpu().numpy()
        print(f"{datetime.datetime.now()} : exporting depth-coded tiffs")
        for z in range(shape[0]):
            write_tiff_plane(os.path.join(path_out_depthmap, "depthmap_" + str(z).zfill(4) + ".tif"), img[z])
    finally:
        if own:
            eng.close()
    print(f"{datetime.datetime.now()} : Cleanup")
