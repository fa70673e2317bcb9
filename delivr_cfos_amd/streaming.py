"""Volumes larger than the HBM of one GPU: Z-slabs of the shard plan, one after another on ONE device.

The reference has no size limit - it streams everything through memmaps (fp16 accumulators on disk,
inference/inference.py:240-247; Arrayterator blocks, :285-299 / :53; cc3d's out_file, count_blobs.py:59-64).  The
resident path of this build holds volume + fp32 accumulator (+ count map, + labels) in HBM, which ends near 45 Gvoxel
on a 288 GB part.  Beyond that - or below `settings["mi355x"]["hbm_budget_gb"]` - the SAME plan the multi-GPU path uses
(`dlv_shard_plan_make` / `ShardPlan.slab`: contiguous window ranges = Z-slabs with their seam and erosion margins) is
executed sequentially:

  inference   for every slab: host memmap -> HBM, all passes of its window range into its own accumulator slab; the
              planes it computed but a neighbour owns are kept (in HBM) until the owner has run; a slab is finalized
              (threshold + eroded re-mask on the reference's z-block grid) as soon as every contribution to the planes
              it owns has arrived - i.e. one slab later - and its mask planes go straight into binaries.npy.  The
              additions happen in the order of the sharded run (owner first, then the neighbours in slab order), so the
              result IS the N-rank result bit for bit, and equals the resident pass up to fp32 association of the seam
              sums (mask identical wherever |mean logit| is not at rounding level; tests/test_gpu_streaming.py).
  labelling   pass 1: CCL-26 per mask slab (local labels -> a scratch file next to the output), raw statistics, the
              label pairs that touch across each seam; host union-find + raster renumbering (parallel.merge_components,
              the code of the multi-GPU path); pass 2: every slab is renumbered with its table and written into the
              output .npy.  Bit-identical to the single-volume labelling.

When not even the thinnest plan fits, MemoryError states the sizes that would (the C ABI's DLV_ENOMEM in words).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np

# bytes of 16-bit activations + partial sums one window voxel needs in a forward of the 16-bit path (4 level-0 tensors of
# 32 channels, the deeper levels, statistics): measured ~330
ACT_BYTES_PER_PATCH_VOXEL = 340


def hbm_budget_bytes(engine, settings: Optional[dict]) -> int:
    """settings["mi355x"]["hbm_budget_gb"] when given, else 92 % of what this process can use right now: what the device reports
    free plus what torch's caching allocator holds without using it (the previous brain's volume, sums and mask: the next brain's
    tensors come out of those blocks - counting them as "used" sent the second brain of a process down the slab-streamed path
    when another process held most of the device)."""
    gb = (settings or {}).get("mi355x", {}).get("hbm_budget_gb") if settings else None
    if gb:
        return int(float(gb) * (1 << 30))
    torch = engine.torch
    free, _total = torch.cuda.mem_get_info(engine.device)
    reusable = max(int(torch.cuda.memory_reserved(engine.device)) - int(torch.cuda.memory_allocated(engine.device)), 0)
    return int((free + reusable) * 0.92)


def forward_workspace_bytes(roi: Sequence[int], precision: str, lanes: int = 3) -> int:
    tile = int(roi[0]) * int(roi[1]) * int(roi[2])
    if precision == "fp32":
        return 6 * 100 * tile * 4  # (parity mode: fp32 NCDHW tiles, small batches)
    batch_vox = max(min(1 << 25, 64 * tile), tile)
    return lanes * batch_vox * ACT_BYTES_PER_PATCH_VOXEL + (1 << 30)


def inference_bytes_per_voxel(need_count: bool, gaussian: bool, want_prob: bool) -> int:
    """HBM per padded voxel of a slab: uint16 volume + fp32 sums (+ count map) + what finalize adds (uint8 mask, distance
    maps, optional fp32 sigmoid)."""
    return 2 + 4 + ((4 if gaussian else 1) if need_count else 0) + 1 + 2 + (4 if want_prob else 0)


def plan_slabs(engine, params, Z: int, plane_voxels: int, bytes_per_voxel: int, fixed_bytes: int, budget: int, zblock: int,
               erode_iters: int = 30, resident_slabs: int = 2):
    """The thinnest-possible search: smallest number of slabs whose thickest slab, `resident_slabs` times (a slab waits for
    its successor before it is finalized), fits the budget beside the forward's workspace.  -> (plan, n_slabs)."""
    from .parallel import plan_from_params

    starts = engine.window_starts(params)
    n_rows = len(np.unique(starts[:, 0]))
    best = None
    for n in range(1, n_rows + 1):
        plan = plan_from_params(params, n, None)
        thick = 0
        for r in range(n):
            lo, hi = plan.slab(r, Z, erode_iters, zblock)
            thick = max(thick, hi - lo)
        need = fixed_bytes + min(resident_slabs, n) * thick * plane_voxels * bytes_per_voxel
        best = (need, thick, n)
        if need <= budget:
            return plan, n
    need, thick, n = best
    raise MemoryError(
        f"delivr_cfos_amd (DLV_ENOMEM): the HBM budget of {budget / 2**30:.1f} GiB cannot hold the thinnest slab plan: {n} slabs "
        f"of up to {thick} planes x {plane_voxels} voxels x {bytes_per_voxel} B, {min(resident_slabs, n)} resident, plus "
        f"{fixed_bytes / 2**30:.1f} GiB of forward workspace = {need / 2**30:.1f} GiB; raise settings['mi355x']['hbm_budget_gb'], "
        "use smaller windows (blob_detection.window_dimensions) or more GPUs (torch.distributed.run)")


def run_inference_streamed(engine, dataset_host, pad, stack_zyx, crop_size, overlap, tta: bool, precision: str, threshold: float,
                           need_count: bool, gaussian: bool, plan, out_mask, out_prob=None, verbose: bool = True):
    """dataset_host: (Zp, Yp, Xp) uint16 host array (the memmap of masked_nifti.npy); out_mask: (Z, Y, X) uint8 host array
    (the memmap of binaries.npy), out_prob: fp32 (Z, Y, X) or None.  Runs every slab of `plan`; raises DelivrHipError
    (DLV_ERANGE included: the caller rescales the offending conv block and repeats the run, range_guard.py)."""
    from .hostlogic import arrayterator_zblock, pass_schedule
    from .parallel import finalize_owned

    torch = engine.torch
    Z, Y, X = (int(v) for v in stack_zyx)
    nb = arrayterator_zblock((Z, Y, X))
    n = plan.world
    cm_dtype = torch.float32 if gaussian else torch.uint8
    seam = {}     # (src, dst) -> (lo, hi, acc planes, cnt planes | None): computed by src, owned by dst
    pending = {}  # slab -> its resident tensors, waiting for the neighbours' contributions

    def last_source(q):
        srcs = [s for s, _lo, _hi in plan.recvs(q)]
        return max(srcs) if srcs else q

    def finalize(q):
        slo, vol, acc, cnt = pending.pop(q)
        for src, lo, hi in sorted(plan.recvs(q), key=lambda t: t[0]):  # increasing source order, as the sharded exchange
            _lo, _hi, a, c = seam.pop((src, q))
            acc[lo - slo:hi - slo] += a
            if cnt is not None:
                cnt[lo - slo:hi - slo] += c
        mask, prob, (olo, ohi) = finalize_owned(engine, plan, q, acc, cnt, vol, (Z, Y, X), threshold, 30,
                                                want_prob=out_prob is not None, z0=slo)
        engine.sync()
        if mask is not None:
            out_mask[olo:ohi] = mask.cpu().numpy()
            if out_prob is not None:
                out_prob[olo:ohi] = prob.cpu().numpy()
        del vol, acc, cnt

    for r in range(n):
        wb, we = plan.win_ranges[r]
        slo, shi = plan.slab(r, Z, 30, nb)
        if shi <= slo:
            continue
        if verbose:
            print(f"  slab {r + 1}/{n}: planes [{slo}, {shi}), windows [{wb}, {we})")
        vol = engine.upload_volume(dataset_host, slo, shi)
        acc = torch.zeros((shi - slo,) + tuple(pad[1:]), dtype=torch.float32, device=engine.device)
        cnt = torch.zeros((shi - slo,) + tuple(pad[1:]), dtype=cm_dtype, device=engine.device) if need_count else None
        if we > wb:
            for flip_dim, repeat in pass_schedule(bool(tta)):
                if gaussian:
                    engine.sw_infer(engine.make_sw_params(pad, crop_size, overlap, flip_dim, 0, precision, win_range=(wb, we),
                                                          slab=(slo, shi - slo), repeat=repeat, blend="gaussian", wsum=cnt), vol, acc)
                else:
                    engine.sw_infer(engine.make_sw_params(pad, crop_size, overlap, flip_dim, 0, precision, win_range=(wb, we),
                                                          slab=(slo, shi - slo), repeat=repeat), vol, acc, cnt)
            engine.sync()
        for dst, lo, hi in plan.sends(r):
            seam[(r, dst)] = (lo, hi, acc[lo - slo:hi - slo].clone(), None if cnt is None else cnt[lo - slo:hi - slo].clone())
        pending[r] = (slo, vol, acc, cnt)
        for q in sorted(pending):
            if last_source(q) <= r:
                finalize(q)
        del vol, acc, cnt
    for q in sorted(pending):
        finalize(q)
    assert not seam, "every seam contribution has an owner"


def ccl_bytes_per_voxel() -> int:
    """uint8 mask + uint32 labels (the union-find runs in the label volume itself) + the bit masks, chunk list and renumbering
    scratch of dlv_ccl26_dev (0.5 B) + the statistics accumulators and slack"""
    return 1 + 4 + 2


def even_slabs(Z: int, n: int):
    cuts = [(Z * r) // n for r in range(n + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(n)]


def ccl_streamed(engine, bin_img, n_slabs: int, scratch_path: str, create_output):
    """bin_img: (Z, Y, X) uint8 host array.  create_output(N) -> writable host array (Z, Y, X) of the final label dtype.
    Returns (N, stats) - labels and statistics identical to the single-volume labelling (count_blobs.py:57-88)."""
    from .parallel import merge_components, merge_stats

    torch = engine.torch
    Z, Y, X = (int(v) for v in bin_img.shape)
    slabs = even_slabs(Z, n_slabs)
    prov = np.lib.format.open_memmap(scratch_path, mode="w+", dtype=np.uint32, shape=(Z, Y, X))
    counts, raws, seams = [], [], []
    prev_last, prev_k = None, None
    try:
        for k, (lo, hi) in enumerate(slabs):
            if hi <= lo:
                counts.append(0)
                raws.append(None)
                continue
            mask = engine.to_device(np.ascontiguousarray(bin_img[lo:hi]))
            labels, n_local = engine.ccl26(mask)
            del mask
            if prev_last is not None:
                pairs = engine.seam_pairs(prev_last, labels[0])
                if len(pairs):
                    seams.append((prev_k, k, pairs))
            counts.append(int(n_local))
            raws.append(engine.cc_stats_raw(labels, n_local))
            prov[lo:hi] = labels.cpu().numpy().view(np.uint32)
            prev_last, prev_k = labels[-1].clone(), k
            del labels
        luts, n_total = merge_components(counts, seams)
        stats = merge_stats(luts, raws, [s[0] for s in slabs], (Z, Y, X), n_total)
        out = create_output(n_total)
        for k, (lo, hi) in enumerate(slabs):
            if hi <= lo:
                continue
            lab = torch.from_numpy(np.ascontiguousarray(prov[lo:hi]).view(np.int32)).to(engine.device)
            engine.relabel(lab, luts[k])
            out[lo:hi] = lab.cpu().numpy().view(np.uint32).astype(out.dtype, copy=False)
            del lab
        if hasattr(out, "flush"):
            out.flush()
    finally:
        del prov
        try:
            os.remove(scratch_path)
        except OSError:
            pass
    return n_total, stats


def stats_streamed(engine, labels, n_total: int, budget_bytes: int) -> dict:
    """cc3d.statistics layout (count_blobs.py:85) of a FINAL label volume (host array / memmap, (Z, Y, X), any unsigned dtype)
    that does not fit the HBM budget: raw accumulators slab by slab (dlv_cc_stats_raw_dev), merged exactly like the sharded
    run's (parallel.merge_stats with identity renumbering) - counts, boxes and integer coordinate sums add up across slabs."""
    from .parallel import merge_stats

    import itertools

    torch = engine.torch
    Z, Y, X = (int(v) for v in labels.shape)
    # the device-side accumulators of dlv_cc_stats_raw_dev (52 B per label) come out of the same budget as the label slab
    acc_bytes = (int(n_total) + 1) * 52
    planes = max(1, int(max(budget_bytes - acc_bytes, 0) // max(1, Y * X * 4 * 2)))  # uint32 slab + headroom
    ident = np.arange(n_total + 1, dtype=np.uint32)
    starts = list(range(0, Z, planes))

    def raw_of_slabs():
        # one slab's raw accumulators at a time: merge_stats folds each into its running sums as zip() hands it over (host
        # memory O(N), not O(slabs x N) - a cached labelling with millions of components over hundreds of slabs)
        for lo in starts:
            hi = min(Z, lo + planes)
            lab = torch.from_numpy(np.ascontiguousarray(labels[lo:hi]).astype(np.uint32).view(np.int32)).to(engine.device)
            raw = engine.cc_stats_raw(lab, n_total)
            del lab
            yield raw

    return merge_stats(itertools.repeat(ident), raw_of_slabs(), starts, (Z, Y, X), n_total)
