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


def _even_slabs(Z: int, world: int):
    cuts = [(Z * r) // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _count_blobs_sharded(eng, bin_img, dist, path_out, brain):
    """One process per GPU: every rank labels a Z-slab of the mask, seams are merged (parallel.ccl_sharded) and every
    rank writes ITS label slab straight into the output .npy (rank 0 creates the file once N - and with it the label
    dtype - is known); only the merged statistics travel to rank 0.  No rank ever holds the whole label volume (17 GB for
    1024x2048x2048).  Returns (N, stats | None)."""
    from .parallel import ccl_sharded

    rank, world = dist.get_rank(), dist.get_world_size()
    Z, Y, X = bin_img.shape
    slabs = _even_slabs(Z, world)
    lo, hi = slabs[rank]
    slab = eng.to_device(np.ascontiguousarray(bin_img[lo:hi])) if hi > lo else None
    labels, N, stats = ccl_sharded(eng, slab, slabs, rank, dist, (Z, Y, X))
    out_path = os.path.join(path_out, f"{brain}-{N}-cc3d.npy")
    err = [None]
    if rank == 0:
        try:
            np.lib.format.open_memmap(out_path, mode="w+", dtype=_label_dtype(N), shape=(Z, Y, X)).flush()
        except Exception as exc:  # every rank must learn about it: they all wait in the broadcast below
            err[0] = repr(exc)
    dist.broadcast_object_list(err, src=0)
    if err[0] is not None:
        raise RuntimeError(f"count_blobs: rank 0 could not create {out_path}: {err[0]}")
    if hi > lo:
        mm = np.load(out_path, mmap_mode="r+")
        mm[lo:hi] = labels.cpu().numpy().view(np.uint32).astype(_label_dtype(N), copy=False)
        mm.flush()
        del mm
    dist.barrier()
    return N, stats


def count_blobs(settings, path_in, brain_i, brain, stack_shape, min_size=-1, max_size=-1, engine=None):
    """Same positional parameters as the reference.  ``engine``: a HipEngine to reuse (one is
    created on device 0 otherwise).  Under torch.distributed (one process per GPU) the labelling is sharded over the
    ranks along z; rank 0 writes the statistics and the CSV, every rank writes its slab of the labels and returns N.
    Rank 0 alone looks for a cached labelling and tells the others which branch to take, so the ranks cannot disagree
    about the collectives that follow (different cache views on a shared file system); a failure on rank 0 reaches the
    other ranks as an error instead of a hang."""
    from .engine import HipEngine

    try:
        import torch.distributed as dist
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except ImportError:  # pragma: no cover
        dist, sharded = None, False
    rank = dist.get_rank() if sharded else 0

    path_out = settings["postprocessing"]["output_location"]
    os.makedirs(path_out, exist_ok=True)  # (every rank may get here first)
    len_b = len(os.listdir(path_in))
    start = datetime.datetime.now()
    print(f"{start} Now postprocessing inference for {brain} - {brain_i}/{len_b}")
    brain_path = os.path.join(path_in, brain, "binary_segmentations", "binaries.npy")
    shape = tuple(int(v) for v in stack_shape[2:])
    bin_img = np.memmap(brain_path, dtype=np.uint8, mode="r", shape=shape, offset=128)
    own = engine is None
    eng = engine or HipEngine(int(os.environ.get("LOCAL_RANK", 0)) if sharded else 0)
    if sharded:
        branch = [None]
        if rank == 0:
            try:
                branch[0] = ("cached", bool(load_cached_brain(settings, brain)))
            except Exception as exc:
                branch[0] = ("error", repr(exc))
        dist.broadcast_object_list(branch, src=0)
        if branch[0][0] == "error":
            if own:
                eng.close()
            raise RuntimeError(f"count_blobs: rank 0 failed while looking for a cached labelling: {branch[0][1]}")
        if not branch[0][1]:
            try:
                N, stats = _count_blobs_sharded(eng, bin_img, dist, path_out, brain)
            finally:
                if own:
                    eng.close()
            if rank == 0:
                with open(os.path.join(path_out, f"{brain}-stats.pickle"), "wb") as fh:
                    pickle.dump(stats, fh, protocol=pickle.HIGHEST_PROTOCOL)
                with open(path_out + csv_name(bin_img.shape, brain), "w") as fh:
                    fh.write(cells_csv_text(stats, N))
                end = datetime.datetime.now()
                print(f"{end} {brain} {brain_i} / {len_b} Done ({dist.get_world_size()} ranks); Took {end - start}")
            dist.barrier()
            return N
        if rank != 0:
            # a cached labelling exists: rank 0 alone re-uses it (and writes the statistics / CSV), the others wait for
            # N - or for the error rank 0 ran into
            if own:
                eng.close()
            box = [None]
            dist.broadcast_object_list(box, src=0)
            if isinstance(box[0], tuple):
                raise RuntimeError(f"count_blobs: rank 0 failed on the cached labelling: {box[0][1]}")
            return box[0]
    result = [("error", "rank 0 did not finish")]
    try:
        N, stats = _count_blobs_single(settings, brain, bin_img, eng, own, path_out, start)
        with open(path_out + csv_name(bin_img.shape, brain), "w") as fh:
            fh.write(cells_csv_text(stats, N))
        result = [N]
    except Exception as exc:
        result = [("error", repr(exc))]
        raise
    finally:
        if sharded:  # always: N, or the error the waiting ranks re-raise
            dist.broadcast_object_list(result, src=0)
    end = datetime.datetime.now()
    print(f"{end} {brain} {brain_i} / {len_b} Done; Took {end - start}")
    return N


def _count_blobs_single(settings, brain, bin_img, eng, own, path_out, start):
    """The one-device path (also rank 0 of a sharded run that found a cached labelling): returns (N, stats)."""
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
    return N, stats
