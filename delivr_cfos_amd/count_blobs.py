"""Mirror of the reference's ``count_blobs.py`` (count_blobs :36-118, caches :10-34): 26-connected
components + statistics on the device instead of cc3d, same files out.

in : <path_in>/<brain>/binary_segmentations/binaries.npy ('|u1', (Z,Y,X), 128-byte header)   (:45-46)
out: <post_out>/<brain>-<N>-cc3d.npy (labels), <post_out>/<brain>-stats.pickle
     (dict voxel_counts / bounding_boxes / centroids), <post_out>(Z, Y, X)_<brain>.csv          (:65,86-88,113-114)
"""
from __future__ import annotations

import datetime
import os
import pickle

import numpy as np

from .hostlogic import cells_csv_text, csv_name


def load_cached_brain(settings, brain):
    """reference :10-21"""
    path_in = settings["postprocessing"]["output_location"]
    result = False
    for item in [x for x in os.listdir(path_in) if ".npy" in x]:
        if brain in item:
            result = os.path.join(path_in, item)
    return result


def load_cached_stats(settings, brain):
    """reference :23-34"""
    path_in = settings["postprocessing"]["output_location"]
    result = False
    for item in [x for x in os.listdir(path_in) if ".pickle" in x]:
        if brain in item:
            result = os.path.join(path_in, item)
    return result


def _label_dtype(n: int):
    # cc3d picks the smallest unsigned type that holds the label count [3P-recall]
    return np.uint16 if n < 2**16 else np.uint32


def count_blobs(settings, path_in, brain_i, brain, stack_shape, min_size=-1, max_size=-1, engine=None):
    """Same positional parameters as the reference.  ``engine``: a HipEngine to reuse (one is
    created on device 0 otherwise)."""
    from .engine import HipEngine

    path_out = settings["postprocessing"]["output_location"]
    if not os.path.exists(path_out):
        os.mkdir(path_out)
    len_b = len(os.listdir(path_in))
    start = datetime.datetime.now()
    print(f"{start} Now postprocessing inference for {brain} - {brain_i}/{len_b}")
    brain_path = os.path.join(path_in, brain, "binary_segmentations", "binaries.npy")
    shape = tuple(int(v) for v in stack_shape[2:])
    bin_img = np.memmap(brain_path, dtype=np.uint8, mode="r", shape=shape, offset=128)
    own = engine is None
    eng = engine or HipEngine(0)
    labels_dev = None
    try:
        cached = load_cached_brain(settings, brain)
        if not cached:
            print("No cached brain found, performing connected components on the GPU...")
            mask_dev = eng.to_device(np.ascontiguousarray(bin_img))
            labels_dev, N = eng.ccl26(mask_dev)
            labels = labels_dev.cpu().numpy().view(np.uint32).astype(_label_dtype(N), copy=False)
            np.save(os.path.join(path_out, f"{brain}-{N}-cc3d.npy"), labels)
        else:
            N = int(cached.split("/")[-1].split("-")[1])
            print(f"Cached brain found at {cached} with {N} components, loading...")
            labels = np.load(cached)
        mid = datetime.datetime.now()
        print(f"{mid} labelling+writing/loading took {mid - start} : {N}")
        cached_stats = load_cached_stats(settings, brain)
        if not cached_stats:
            if labels_dev is None:
                import torch

                labels_dev = torch.from_numpy(labels.astype(np.uint32).view(np.int32)).to(eng.device)
            stats = eng.cc_stats(labels_dev, N)
            with open(os.path.join(path_out, f"{brain}-stats.pickle"), "wb") as fh:
                pickle.dump(stats, fh, protocol=pickle.HIGHEST_PROTOCOL)
        else:
            print(f"Found stats at {cached_stats}")
            with open(cached_stats, "rb") as fh:
                stats = pickle.load(fh)
    finally:
        if own:
            eng.close()
    # note: size filtering happens later in the reference too (count_blobs.py:105)
    with open(path_out + csv_name(bin_img.shape, brain), "w") as fh:
        fh.write(cells_csv_text(stats, N))
    end = datetime.datetime.now()
    print(f"{end} {brain} {brain_i} / {len_b} Done; Took {end - start}")
    return N
