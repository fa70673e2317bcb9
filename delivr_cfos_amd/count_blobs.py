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

from . import hostio
from .hostlogic import cells_csv_bytes, csv_name


def _find_cached(path: str, suffix: str, brain: str):
    """The LAST directory entry (os.listdir order, as the reference iterates) whose name holds `suffix` and `brain`; False
    when there is none - what the reference's three cache look-ups return (count_blobs.py:10-34, blob_highlighter.py)."""
    hits = [x for x in os.listdir(path) if suffix in x and brain in x]
    return os.path.join(path, hits[-1]) if hits else False


def load_cached_brain(settings, brain):
    """reference :10-21"""
    return _find_cached(settings["postprocessing"]["output_location"], ".npy", brain)


def load_cached_stats(settings, brain):
    """reference :23-34"""
    return _find_cached(settings["postprocessing"]["output_location"], ".pickle", brain)


def _label_dtype(n: int):
    # cc3d picks the smallest unsigned type that holds the label count [3P-recall]
    return np.uint16 if n < 2**16 else np.uint32


def _labels_in_file_dtype(labels_dev, n: int):
    """the uint32 labels (held in an int32 tensor) in the width of the file's dtype, converted in HBM: a volume with fewer
    than 2^16 components crosses PCIe and reaches the file as 2 bytes per voxel"""
    if _label_dtype(n) == np.uint16:
        import torch

        return labels_dev.to(torch.int16)  # (labels < 2^16: the low half is the value)
    return labels_dev


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
    slab = hostio.upload(eng, bin_img[lo:hi], what="h2d_mask") if hi > lo else None
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
    # every rank writes ITS slab into the one file: path_out must be a directory all ranks share.  The outcome of every
    # write is exchanged - a rank that cannot see or write the file (node-local path, ENOSPC) must not leave the others
    # waiting in a barrier
    mine = None
    try:
        if hi > lo:
            mm = np.load(out_path, mmap_mode="r")
            off = int(mm.offset) + lo * Y * X * mm.dtype.itemsize
            del mm
            hostio.download(eng, _labels_in_file_dtype(labels, N), out_path, offset=off, what="d2h_labels", sparse=True)  # (rank 0 created the file just now)
    except Exception as exc:
        mine = f"rank {rank}: {exc!r}"
    _raise_if_any_failed(dist, mine, f"count_blobs: writing the label slabs into {out_path} (path_out must be shared by all ranks)")
    return N, stats


def _raise_if_any_failed(dist, mine, what: str):
    """all_gather of every rank's error text (None = fine): all ranks raise together or none does"""
    outcomes = [None] * dist.get_world_size()
    dist.all_gather_object(outcomes, mine)
    bad = [o for o in outcomes if o]
    if bad:
        raise RuntimeError(f"{what} failed: " + "; ".join(bad))


def count_blobs(settings, path_in, brain_i, brain, stack_shape, min_size=-1, max_size=-1, engine=None, defer_write=False):  # noqa: C901
    """Same positional parameters as the reference.  ``engine``: a HipEngine to use (default: the process-wide engine of the
    device, engine.shared_engine - the one run_inference used, with its workspaces).  ``defer_write``: return when the statistics
    and the CSV are written - the label volume keeps streaming into its file on a background worker (hostio.wait_deferred() joins;
    the file appears under its name only when complete), so that the next brain's labelling does not wait for 17 GB of writes.  Under torch.distributed (one process per GPU) the labelling is sharded over the
    ranks along z; rank 0 writes the statistics and the CSV, every rank writes its slab of the labels and returns N.
    Rank 0 alone looks for a cached labelling and tells the others which branch to take, so the ranks cannot disagree
    about the collectives that follow (different cache views on a shared file system); a failure on rank 0 reaches the
    other ranks as an error instead of a hang."""
    from .engine import shared_engine

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
    own = False  # (the shared engine outlives the call)
    eng = engine or shared_engine(int(os.environ.get("LOCAL_RANK", 0)) if sharded else 0)
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
            mine = None
            if rank == 0:
                try:
                    with open(os.path.join(path_out, f"{brain}-stats.pickle"), "wb") as fh:
                        pickle.dump(stats, fh, protocol=pickle.HIGHEST_PROTOCOL)
                    with open(path_out + csv_name(bin_img.shape, brain), "wb") as fh:
                        fh.write(cells_csv_bytes(stats, N))
                    end = datetime.datetime.now()
                    print(f"{end} {brain} {brain_i} / {len_b} Done ({dist.get_world_size()} ranks); Took {end - start}")
                except Exception as exc:
                    mine = f"rank 0: {exc!r}"
            _raise_if_any_failed(dist, mine, "count_blobs: writing the statistics / CSV")
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
        import time

        N, stats, labels_written = _count_blobs_single(settings, brain, bin_img, eng, own, path_out, start)
        t_csv = time.perf_counter()
        with open(path_out + csv_name(bin_img.shape, brain), "wb") as fh:
            fh.write(cells_csv_bytes(stats, N))  # (the text pandas writes for the reference, formatted by the library: dlv_cells_csv)
        count_blobs.last_timings["csv_s"] = time.perf_counter() - t_csv
        t_join = time.perf_counter()
        if defer_write:
            hostio.submit_deferred(eng, labels_written)
        else:
            labels_written()  # the label volume has been streaming into its file since the labelling finished
        count_blobs.last_timings["wait_for_labels_s"] = time.perf_counter() - t_join
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
    """The one-device path (also rank 0 of a sharded run that found a cached labelling): returns (N, stats, wait) - wait()
    returns when the label file is complete (it is written by a side thread while the statistics, the pickle and the CSV are
    made: 17 GB at the 4-7 GB/s one file takes from the kernel) and re-raises what that thread ran into."""
    import time

    labels_dev = None
    wait = lambda: None  # noqa: E731
    tm = count_blobs.last_timings = {}
    t_prev = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        tm[name + "_s"] = now - t_prev[0]
        t_prev[0] = now
    try:
        cached = load_cached_brain(settings, brain)
        from .streaming import ccl_bytes_per_voxel, ccl_streamed, hbm_budget_bytes

        budget = hbm_budget_bytes(eng, settings)
        need = int(bin_img.size) * ccl_bytes_per_voxel()
        if not cached and need > budget:
            # the mask + its uint32 labels do not fit this GPU: Z-slabs through the device, seams merged on the host
            # (streaming.py) - the reference's counterpart is cc3d writing into an out_file memmap (:59-64)
            plane = int(bin_img.shape[1]) * int(bin_img.shape[2]) * ccl_bytes_per_voxel()
            n_slabs = -(-need // max(budget, 1))
            if plane > budget or n_slabs > bin_img.shape[0]:
                raise MemoryError(f"delivr_cfos_amd (DLV_ENOMEM): one mask plane with its labels and scratch needs {plane / 2**20:.1f} MiB, "
                                  f"the HBM budget is {budget / 2**20:.1f} MiB; raise settings['mi355x']['hbm_budget_gb']")
            print(f"No cached brain found; mask + labels of {need / 2**30:.1f} GiB exceed the HBM budget of {budget / 2**30:.1f} GiB: "
                  f"connected components on {n_slabs} Z-slabs...")
            made = {}

            def create_output(n):
                # written under a name load_cached_brain does NOT match (no '.npy' suffix) and renamed when pass 2 has finished:
                # a run killed while it renumbers the slabs must not leave a complete-looking label file behind as a cache
                made["path"] = os.path.join(path_out, f"{brain}-{n}-cc3d.npy")
                made["tmp"] = os.path.join(path_out, f".{brain}-{n}-cc3d.partial")
                return np.lib.format.open_memmap(made["tmp"], mode="w+", dtype=_label_dtype(n), shape=tuple(bin_img.shape))

            N, stats = ccl_streamed(eng, bin_img, int(n_slabs), os.path.join(path_out, f".{brain}-provisional-u32.tmp"), create_output)
            if made:
                os.replace(made["tmp"], made["path"])
            with open(os.path.join(path_out, f"{brain}-stats.pickle"), "wb") as fh:
                pickle.dump(stats, fh, protocol=pickle.HIGHEST_PROTOCOL)
            return N, stats, wait
        if not cached:
            print("No cached brain found, performing connected components on the GPU...")
            # binaries.npy -> HBM and the label volume -> its .npy both stream through pinned staging with parallel
            # readers / writers (hostio.py): 4.3 GB in and 17 GB out for a 1024 x 2048 x 2048 brain, around 20 ms of kernels
            mask_dev = hostio.upload(eng, bin_img, what="h2d_mask")
            mark("upload")
            labels_dev, N = eng.ccl26(mask_dev)
            del mask_dev
            mark("ccl26")
            final = os.path.join(path_out, f"{brain}-{N}-cc3d.npy")
            # (written as <name>.partial and renamed: never a partly written file under the cache's name) - by a side thread that
            # touches torch's copy stream only, never the context, while this thread goes on to the statistics and the CSV
            file_labels = _labels_in_file_dtype(labels_dev, N)
            eng.sync()
            from concurrent.futures import ThreadPoolExecutor

            writer = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dlv-labels")
            fut = writer.submit(hostio.save_npy, eng, file_labels, final, _label_dtype(N), "d2h_labels", True, True)

            def wait(fut=fut, writer=writer):
                try:
                    fut.result()
                finally:
                    writer.shutdown(wait=True)

            labels = None
            mark("start_label_write")
        else:
            N = int(cached.split("/")[-1].split("-")[1])
            print(f"Cached brain found at {cached} with {N} components, loading...")
            labels = np.load(cached, mmap_mode="r")
        mid = datetime.datetime.now()
        print(f"{mid} labelling+writing/loading took {mid - start} : {N}")
        cached_stats = load_cached_stats(settings, brain)
        if not cached_stats:
            if labels_dev is None and int(labels.size) * 4 > budget:
                # cached labels that do not fit the HBM budget: statistics slab by slab (raw sums add up; streaming.py)
                from .streaming import stats_streamed

                stats = stats_streamed(eng, labels, N, budget)
            else:
                if labels_dev is None:
                    import torch

                    if labels.dtype == np.uint32:
                        labels_dev = hostio.upload(eng, labels, what="h2d_labels")
                    elif labels.dtype == np.uint16:  # widened in HBM, not on the host
                        labels_dev = hostio.upload(eng, labels, what="h2d_labels").view(torch.int16).to(torch.int32) & 0xFFFF
                    else:
                        labels_dev = torch.from_numpy(np.ascontiguousarray(labels).astype(np.uint32).view(np.int32)).to(eng.device)
                stats = eng.cc_stats(labels_dev, N)
            with open(os.path.join(path_out, f"{brain}-stats.pickle"), "wb") as fh:
                pickle.dump(stats, fh, protocol=pickle.HIGHEST_PROTOCOL)
            mark("stats")
        else:
            print(f"Found stats at {cached_stats}")
            with open(cached_stats, "rb") as fh:
                stats = pickle.load(fh)
    except BaseException:
        try:
            wait()  # (do not leave the writer thread behind an error of this one)
        except Exception:
            pass
        raise
    finally:
        if own:
            eng.close()
    # note: size filtering happens later in the reference too (count_blobs.py:105)
    return N, stats, wait
