"""Sharding of one sliding-window pass over the ranks of a node (one process per GPU).

Replaces torch.nn.DataParallel (inference/inference.py:217-219: per-forward scatter, parameter
re-broadcast and gather through GPU 0) with:
  * one broadcast of the packed weight blob at start-up (RCCL over xGMI),
  * a static partition of the reference's window list into contiguous ranges (Z-slowest order, so a
    rank's windows form a Z-slab of tile rows),
  * one seam exchange per pass: a rank that computed contributions to planes owned by another rank
    sends exactly those planes (point-to-point, one message per seam), the owner adds them in rank
    order (deterministic),
  * each rank finalizes the planes it owns; mask slabs are gathered to rank 0 on request.
No all-reduce is needed on this path.

Connected components over the sharded mask: every rank labels the planes it owns, the ranks
exchange ONE boundary plane of provisional labels per seam, the (label, label) pairs that touch across a seam are
united in a small host-side union-find over (rank, local label) nodes, and every rank renumbers its slab with a
lookup table.  Because local labels are numbered in raster order and slabs are ordered along z, "component with
the smallest (rank, local label) member first" IS the raster order of the components' first voxels - the merged
labels equal the single-volume labelling bit for bit, for any number of shards (tests/test_host_cpu.py with gloo,
tests/test_gpu_pipeline.py on the device).

The plan is pure integer arithmetic (testable without a GPU); the exchange works on any
torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class ShardPlan:
    world: int
    n_windows: int
    win_ranges: List[Tuple[int, int]]      # [begin, end) of the reference's window enumeration per rank
    z_computed: List[Tuple[int, int]]      # planes [lo, hi) a rank's windows touch (empty: (0, 0))
    z_owned: List[Tuple[int, int]]         # planes [lo, hi) a rank finalizes (partition of [0, Zp))

    def slab(self, rank: int, Z: int, erode_iters: int = 30, zblock: int = 0) -> Tuple[int, int]:
        """Planes [lo, hi) `rank` must hold: its windows' planes and the planes it owns, extended by `erode_iters`
        planes inside their z-blocks (what the eroded re-mask of the owned planes looks at) - dlv_shard_slab."""
        lo, hi = self.z_computed[rank]
        olo, ohi = self.z_owned[rank]
        ohi = min(ohi, Z)
        if ohi > olo:
            nb = zblock if zblock > 0 else Z
            elo = max(olo - erode_iters, (olo // nb) * nb)
            ehi = min(ohi + erode_iters, min(((ohi - 1) // nb + 1) * nb, Z))
            lo, hi = (elo, ehi) if hi <= lo else (min(lo, elo), max(hi, ehi))
        return lo, max(hi, lo)

    def sends(self, rank: int) -> List[Tuple[int, int, int]]:
        """(dst, lo, hi): plane ranges `rank` computed that another rank owns."""
        out = []
        clo, chi = self.z_computed[rank]
        for dst in range(self.world):
            if dst == rank:
                continue
            olo, ohi = self.z_owned[dst]
            lo, hi = max(clo, olo), min(chi, ohi)
            if lo < hi:
                out.append((dst, lo, hi))
        return out

    def recvs(self, rank: int) -> List[Tuple[int, int, int]]:
        """(src, lo, hi): plane ranges owned by `rank` that another rank contributed to."""
        out = []
        olo, ohi = self.z_owned[rank]
        for src in range(self.world):
            if src == rank:
                continue
            clo, chi = self.z_computed[src]
            lo, hi = max(clo, olo), min(chi, ohi)
            if lo < hi:
                out.append((src, lo, hi))
        return out


def make_plan(starts: np.ndarray, roi_z: int, Zp: int, world: int, weights: Optional[np.ndarray] = None) -> ShardPlan:
    """starts: (n_windows, 3) window origins in the reference's order (Z slowest).  Windows are split into
    `world` contiguous ranges of near-equal WORK.  Without `weights` every window counts the same and cuts
    snap to Z tile-row boundaries when there are enough rows (every seam is then one half-tile slab).  With
    `weights` (e.g. 1 for a window that runs the network, ~0.02 for a background-skipped one) the cuts fall
    where the cumulative weight is balanced, anywhere in the list: ranks that share a tile row exchange a
    thicker seam, which is still far cheaper than idling (a brain fills the central slabs, not the outer ones)."""
    n = int(len(starts))
    zs = starts[:, 0]
    cuts = [0]
    if weights is None:
        # boundaries of tile rows (indices where z changes)
        row_edges = [0] + [int(i) for i in np.nonzero(np.diff(zs))[0] + 1] + [n]
        for r in range(1, world):
            target = n * r / world
            if len(row_edges) - 1 >= world:
                c = min(row_edges, key=lambda e: abs(e - target))
            else:
                c = int(round(target))
            cuts.append(max(c, cuts[-1]))
    else:
        w = np.asarray(weights, dtype=np.float64)
        cum = np.concatenate([[0.0], np.cumsum(w)])
        total = cum[-1]
        for r in range(1, world):
            c = int(np.searchsorted(cum, total * r / world, side="left")) if total > 0 else int(round(n * r / world))
            cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    win_ranges = [(cuts[r], cuts[r + 1]) for r in range(world)]
    z_computed = []
    for b, e in win_ranges:
        if e > b:
            z_computed.append((int(zs[b:e].min()), int(zs[b:e].max()) + int(roi_z)))
        else:
            z_computed.append((0, 0))
    # ownership: split [0, Zp) at the midpoints of the seams between consecutive non-empty ranks
    owned = []
    lo = 0
    live = [r for r in range(world) if win_ranges[r][1] > win_ranges[r][0]]
    for r in range(world):
        if r not in live:
            owned.append((lo, lo))
            continue
        nxt = [s for s in live if s > r]
        if nxt:
            s = nxt[0]
            hi = (z_computed[s][0] + z_computed[r][1]) // 2
            hi = min(max(hi, lo), Zp)
        else:
            hi = Zp
        owned.append((lo, hi))
        lo = hi
    return ShardPlan(world, n, win_ranges, z_computed, owned)


def plan_from_params(params, world: int, weights: Optional[np.ndarray] = None) -> ShardPlan:
    """The same plan from the C ABI (dlv_shard_plan_make: what a host without Python uses); `params`: _lib.SwParams."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    out = _lib.ShardPlanC()
    wp = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float32)
        wp = w.ctypes.data_as(C.POINTER(C.c_float))
    rc = lib.dlv_shard_plan_make(C.byref(params), int(world), wp, C.byref(out))
    if rc != 0:
        raise _lib.DelivrHipError(rc, "dlv_shard_plan_make: bad geometry or world size")
    return ShardPlan(int(world), int(out.n_windows), [(int(out.win_begin[r]), int(out.win_end[r])) for r in range(world)],
                     [(int(out.z_comp_lo[r]), int(out.z_comp_hi[r])) for r in range(world)],
                     [(int(out.z_own_lo[r]), int(out.z_own_hi[r])) for r in range(world)])


def balanced_plan(engine, params, fetch_planes, world: int, rank: int, dist, Z: int, erode_iters: int = 30, zblock: int = 0):
    """Slab-resident sharding of one volume: no rank ever holds (or reads) the whole volume.
      1. unweighted plan -> every rank fetches the planes of ITS window range (`fetch_planes(lo, hi)` -> uint16 tensor
         (hi-lo, Yp, Xp) in HBM: a slice of the host memmap, or a generated chunk) and computes the maxima of its windows,
      2. one small all_gather of the maxima -> plan balanced by the windows that actually run the network,
      3. every rank completes its final slab (planes it already holds are re-used).
    Returns (plan, slab_lo, slab_hi, vol_slab)."""
    import torch

    from . import _lib

    plan0 = plan_from_params(params, world, None)
    wb, we = plan0.win_ranges[rank]
    lo0, hi0 = plan0.z_computed[rank]
    vol0 = fetch_planes(lo0, hi0) if hi0 > lo0 else None
    if we > wb:
        q = _lib.SwParams.from_buffer_copy(params)
        q.win_begin, q.win_end, q.z0, q.nz = wb, we, lo0, hi0 - lo0
        mine = engine.window_max(q, vol0)
    else:
        mine = np.zeros(0, dtype=np.int32)
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    wmax = np.concatenate([np.asarray(x, dtype=np.int32) for x in parts])
    plan = plan_from_params(params, world, np.where(wmax > params.skip_threshold, 1.0, 0.02).astype(np.float32))
    lo, hi = plan.slab(rank, Z, erode_iters, zblock)
    if hi <= lo:
        return plan, lo, lo, torch.empty((0, int(params.Yp), int(params.Xp)), dtype=torch.uint16, device=engine.device)
    vol = torch.empty((hi - lo, int(params.Yp), int(params.Xp)), dtype=torch.uint16, device=engine.device)
    have_lo, have_hi = (max(lo, lo0), min(hi, hi0)) if vol0 is not None else (lo, lo)
    if have_hi > have_lo:
        vol[have_lo - lo: have_hi - lo] = vol0[have_lo - lo0: have_hi - lo0]
        if lo < have_lo:
            vol[: have_lo - lo] = fetch_planes(lo, have_lo)
        if have_hi < hi:
            vol[have_hi - lo:] = fetch_planes(have_hi, hi)
    else:
        vol[:] = fetch_planes(lo, hi)
    return plan, lo, hi, vol


def _needs_host_staging(t, dist) -> bool:
    """gloo moves CPU tensors only for point-to-point; RCCL ("nccl") moves HBM directly over xGMI."""
    try:
        return bool(t.is_cuda) and dist.get_backend() == "gloo"
    except Exception:
        return False


def exchange_seams(acc, plan: ShardPlan, rank: int, dist, group=None, z0: int = 0) -> None:
    """acc: fp32 (or uint8 count) tensor holding planes [z0, z0 + acc.shape[0]) of the padded volume - the rank's slab
    (z0 = 0 with the full extent also works).  After the call the planes owned by `rank` hold the complete sum.
    Contributions are added in increasing source-rank order so that the result does not depend on arrival order."""
    import torch

    stage = _needs_host_staging(acc, dist)
    sends = plan.sends(rank)
    recvs = plan.recvs(rank)
    ops, bufs = [], []
    for dst, lo, hi in sends:
        t = acc[lo - z0:hi - z0].contiguous()
        ops.append(dist.P2POp(dist.isend, t.cpu() if stage else t, dst, group=group))
    for src, lo, hi in recvs:
        buf = torch.empty_like(acc[lo - z0:hi - z0], device="cpu") if stage else torch.empty_like(acc[lo - z0:hi - z0])
        bufs.append((src, lo, hi, buf))
        ops.append(dist.P2POp(dist.irecv, buf, src, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for src, lo, hi, buf in sorted(bufs, key=lambda t: t[0]):
        acc[lo - z0:hi - z0] += buf.to(acc.device) if stage else buf


def p2p_selftest(dist, device, rank: int, world: int, nbytes: int = 64 << 20, group=None) -> None:
    """One ring exchange (rank -> rank+1, at world size 1: to itself) of `nbytes` through the same batch_isend_irecv the seam
    exchange uses, compared byte for byte.  Raises RuntimeError naming the backend when the transport delivers anything else."""
    import torch

    n = int(nbytes) // 8
    base = torch.arange(n, dtype=torch.int64, device=device)
    send = base * 2654435761 + rank
    recv = torch.zeros_like(send)
    stage = _needs_host_staging(send, dist)
    sbuf, rbuf = (send.cpu(), recv.cpu()) if stage else (send, recv)
    if world == 1 and dist.get_backend() != "nccl":
        rbuf.copy_(sbuf)  # (gloo has no connection of a rank to itself; RCCL runs the grouped send / recv to self)
    else:
        ops = [dist.P2POp(dist.isend, sbuf, (rank + 1) % world, group=group), dist.P2POp(dist.irecv, rbuf, (rank - 1) % world, group=group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if not stage and send.is_cuda:
        torch.cuda.synchronize(device)
    want = base * 2654435761 + (rank - 1) % world
    if not torch.equal(rbuf.to(device), want):
        raise RuntimeError(f"p2p self-test failed on backend {dist.get_backend()}: rank {rank} did not receive rank {(rank - 1) % world}'s buffer intact")


def broadcast_weights(engine, dist, rank: int, features: Optional[Sequence[int]] = None, src: int = 0, group=None) -> None:
    """Rank `src` has called engine.load_state_dict(); the others receive the packed blob
    (fp32 + MFMA-packed bf16 parameters, ~35 MB) with ONE broadcast."""
    import torch

    from ._lib import ABI_VERSION

    # features + the per-block shifts the packs were made with + the sender's ABI version (the blob's layout is part of it: a
    # rank on another build of the library must refuse the blob, not misread it)
    meta = torch.zeros(6 + 18 + 1, dtype=torch.int64, device=engine.device)
    if rank == src:
        meta[:] = torch.tensor(list(engine.features) + list(engine.conv_shifts()) + [int(engine.lib.dlv_abi_version())], dtype=torch.int64)
    dist.broadcast(meta, src, group=group)
    if int(meta[24].item()) != int(engine.lib.dlv_abi_version()) or int(meta[24].item()) != ABI_VERSION:
        raise RuntimeError(f"broadcast_weights: rank {src} packed the weights with ABI version {int(meta[24].item())}, rank {rank} runs "
                           f"libdelivr_hip ABI version {int(engine.lib.dlv_abi_version())} - the blob layouts may differ")
    if rank != src:
        engine.alloc_weight_blob([int(v) for v in meta[:6].tolist()])
    blob = engine.weight_blob()
    engine.sync()
    dist.broadcast(blob, src, group=group)
    torch.cuda.synchronize(engine.device)
    if rank != src:
        engine.note_conv_shifts([int(v) for v in meta[6:24].tolist()])


def gather_slabs(slab, plan: ShardPlan, rank: int, dist, out=None, dst: int = 0, group=None):
    """Mask slabs (planes z_owned[r]) -> one (Z, Y, X) tensor on rank `dst` (point-to-point)."""
    import torch

    stage = _needs_host_staging(slab if slab is not None else out, dist)
    ops, landed = [], []
    if rank == dst:
        for r in range(plan.world):
            lo, hi = plan.z_owned[r]
            hi = min(hi, out.shape[0])
            if hi <= lo:
                continue
            if r == dst:
                out[lo:hi] = slab[: hi - lo]
            elif stage:
                buf = torch.empty_like(out[lo:hi], device="cpu")
                landed.append((lo, hi, buf))
                ops.append(dist.P2POp(dist.irecv, buf, r, group=group))
            else:
                ops.append(dist.P2POp(dist.irecv, out[lo:hi], r, group=group))
    else:
        if slab is not None and slab.numel():
            t = slab.contiguous()
            ops.append(dist.P2POp(dist.isend, t.cpu() if stage else t, dst, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for lo, hi, buf in landed:
        out[lo:hi] = buf.to(out.device)
    return out


def finalize_owned(engine, plan: ShardPlan, rank: int, acc, cnt, vol, stack_shape, threshold=0.5, erode_iters=30,
                   want_prob=False, z0: int = 0):
    """Threshold + eroded re-mask of the planes `rank` owns.  acc / cnt / vol hold planes [z0, z0 + n) of the padded
    volume (the rank's slab, ShardPlan.slab; z0 = 0 with full-extent buffers).  The erosion is evaluated on the
    reference's Arrayterator z-block grid (inference/inference.py:53) at ABSOLUTE plane indices
    (dlv_finalize_slab_dev), on the owned planes plus `erode_iters` planes of margin inside their blocks, so the result
    equals the single-volume one.  Returns (slab uint8 (n_owned, Y, X), prob or None, (lo, hi))."""
    from .hostlogic import arrayterator_zblock

    Z, Y, X = (int(v) for v in stack_shape)
    nb = arrayterator_zblock((Z, Y, X))
    lo, hi = plan.z_owned[rank]
    hi = min(hi, Z)
    if hi <= lo:
        return None, None, (lo, lo)
    elo = max(lo - erode_iters, (lo // nb) * nb, z0)
    ehi = min(hi + erode_iters, min(((hi - 1) // nb + 1) * nb, Z), z0 + int(acc.shape[0]))
    if elo > max(lo - erode_iters, (lo // nb) * nb) or ehi < min(hi + erode_iters, min(((hi - 1) // nb + 1) * nb, Z)):
        raise ValueError(f"rank {rank}: the slab [{z0}, {z0 + int(acc.shape[0])}) lacks the erosion margin of its owned planes [{lo}, {hi})")
    res = engine.finalize_slab(acc[elo - z0:ehi - z0], None if cnt is None else cnt[elo - z0:ehi - z0], vol[elo - z0:ehi - z0], elo,
                               (Y, X), threshold, erode_iters, nb, want_prob=want_prob)
    mask, prob = res if want_prob else (res, None)
    return mask[lo - elo: hi - elo], (None if prob is None else prob[lo - elo: hi - elo]), (lo, hi)


# ---------------------------------------------------------------------------------------------------------------
# connected components over Z-slabs
# ---------------------------------------------------------------------------------------------------------------
def merge_components(counts: Sequence[int], seam_pairs: Sequence[Tuple[int, int, np.ndarray]]):
    """counts[r] = number of local components of slab r (slabs in z order; 0 for an empty slab).
    seam_pairs: (r_upper, r_lower, pairs (k,2)) - local labels of r_upper / r_lower that touch across their seam.
    Returns (luts, N): luts[r] uint32 (counts[r]+1,) maps a local label to the global one (luts[r][0] = 0);
    global labels 1..N are numbered by the smallest (slab, local label) member of each component."""
    counts = [int(c) for c in counts]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)   # node id = offs[r] + local label - 1
    total = int(offs[-1])
    roots = np.arange(total, dtype=np.int64)
    # only the components that touch a seam take part in the union-find (a few thousand of up to millions)
    edges = []
    for ra, rb, pairs in seam_pairs:
        pr = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        if len(pr):
            edges.append(np.stack([offs[ra] + pr[:, 0] - 1, offs[rb] + pr[:, 1] - 1], axis=1))
    if edges:
        e = np.unique(np.concatenate(edges), axis=0)
        nodes, inv = np.unique(e.ravel(), return_inverse=True)       # nodes ascending: compact index order = id order
        inv = inv.reshape(-1, 2)
        parent = list(range(len(nodes)))

        def find(i):
            root = i
            while parent[root] != root:
                root = parent[root]
            while parent[i] != root:
                parent[i], i = root, parent[i]
            return root

        for a, b in inv.tolist():
            a, b = find(a), find(b)
            if a != b:
                if a < b:
                    parent[b] = a
                else:
                    parent[a] = b
        roots[nodes] = nodes[np.array([find(i) for i in range(len(nodes))], dtype=np.int64)]
    # the root is the minimum node id of its component (unions always keep the smaller id)
    is_root = roots == np.arange(total)
    rank_of_root = np.cumsum(is_root)            # 1-based raster rank of each root
    glob = rank_of_root[roots].astype(np.uint32)
    luts = []
    for r, c in enumerate(counts):
        lut = np.zeros(c + 1, dtype=np.uint32)
        lut[1:] = glob[offs[r]: offs[r] + c]
        luts.append(lut)
    return luts, int(is_root.sum())


def merge_stats(luts, raws, z_offsets, shape, n_total: int) -> dict:
    """Per-slab raw statistics (engine.cc_stats_raw; None for an empty slab) -> the cc3d.statistics layout of the
    whole volume (count_blobs.py:85): voxel_counts uint32, bounding_boxes uint16 (N+1,6) inclusive, centroids float64."""
    Z, Y, X = (int(v) for v in shape)
    rows = n_total + 1
    counts = np.zeros(rows, dtype=np.uint64)
    sums = np.zeros((rows, 3), dtype=np.uint64)
    bbmin = np.full((rows, 3), np.iinfo(np.uint32).max, dtype=np.uint64)
    bbmax = np.zeros((rows, 3), dtype=np.uint64)
    for lut, raw, z0 in zip(luts, raws, z_offsets):
        if raw is None:
            continue
        c = raw["counts"].astype(np.uint64)
        present = c > 0
        present[0] = False
        g = lut.astype(np.int64)
        idx = g[present]
        np.add.at(counts, idx, c[present])
        s = raw["sums"].astype(np.uint64).copy()
        s[:, 0] += c * np.uint64(z0)                      # slab-local z -> volume z
        np.add.at(sums, idx, s[present])
        lo = raw["bbmin"].astype(np.uint64).copy()
        hi = raw["bbmax"].astype(np.uint64).copy()
        lo[:, 0] += np.uint64(z0)
        hi[:, 0] += np.uint64(z0)
        np.minimum.at(bbmin, idx, lo[present])
        np.maximum.at(bbmax, idx, hi[present])
        # background box of this slab (row 0 of the raw arrays is valid when the slab has any background voxel)
        if raw["bbmin"][0, 0] != np.iinfo(np.uint32).max:
            bbmin[0] = np.minimum(bbmin[0], lo[0])
            bbmax[0] = np.maximum(bbmax[0], hi[0])
    nvox = Z * Y * X
    fg = int(counts[1:].sum())
    bgc = nvox - fg
    counts[0] = bgc
    dims = (Z, Y, X)
    fs = sums[1:].sum(axis=0)
    cent = np.full((rows, 3), np.nan, dtype=np.float64)
    nz = counts[1:] > 0
    cent[1:][nz] = sums[1:][nz].astype(np.float64) / counts[1:][nz].astype(np.float64)[:, None]
    bbox = np.zeros((rows, 6), dtype=np.uint16)
    bbox[1:, 0::2] = bbmin[1:].astype(np.uint16)
    bbox[1:, 1::2] = bbmax[1:].astype(np.uint16)
    for k in range(3):
        allk = (nvox // dims[k]) * (dims[k] * (dims[k] - 1) // 2)
        if bgc:
            cent[0, k] = float(allk - int(fs[k])) / float(bgc)
            bbox[0, 2 * k], bbox[0, 2 * k + 1] = int(bbmin[0, k]), int(bbmax[0, k])
    return {"voxel_counts": counts.astype(np.uint32), "bounding_boxes": bbox, "centroids": cent}


def slab_ranges(plan: ShardPlan, Z: int) -> List[Tuple[int, int]]:
    """Planes [lo, hi) of the UNPADDED volume each rank owns (the padded tail belongs to nobody)."""
    return [(min(lo, Z), min(hi, Z)) for lo, hi in plan.z_owned]


def ccl_sharded(engine, mask_slab, slabs: Sequence[Tuple[int, int]], rank: int, dist, shape, want_stats: bool = True,
                group=None):
    """26-connectivity labelling of a mask whose Z-slabs live on different ranks.
    mask_slab: uint8 (hi-lo, Y, X) in HBM, the planes slabs[rank] of the volume (None / empty for a rank without planes).
    Returns (labels_slab int32 tensor (uint32 payload) or None, N, stats or None): labels are the GLOBAL ones, identical
    to a single-volume cc3d.connected_components; stats (rank 0 only, when want_stats) in cc3d.statistics layout."""
    import torch

    world = len(slabs)
    live = [r for r in range(world) if slabs[r][1] > slabs[r][0]]
    mine = rank in live
    labels, n_local = (engine.ccl26(mask_slab) if mine else (None, 0))
    # one boundary plane per seam: the lower slab sends its first plane of provisional labels to the upper one
    pairs = np.zeros((0, 2), dtype=np.uint32)
    if mine and len(live) > 1:
        k = live.index(rank)
        upper = live[k - 1] if k > 0 else None
        lower = live[k + 1] if k + 1 < len(live) else None
        stage = _needs_host_staging(labels, dist)
        ops, recv_buf = [], None
        if upper is not None:
            t = labels[0].contiguous()
            ops.append(dist.P2POp(dist.isend, t.cpu() if stage else t, upper, group=group))
        if lower is not None:
            recv_buf = torch.empty_like(labels[0], device="cpu") if stage else torch.empty_like(labels[0])
            ops.append(dist.P2POp(dist.irecv, recv_buf, lower, group=group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if lower is not None:
            pairs = engine.seam_pairs(labels[-1], recv_buf.to(labels.device) if stage else recv_buf)
    # tiny all-gather: component counts and seam pairs of every rank; every rank runs the same merge
    gathered = [None] * world
    dist.all_gather_object(gathered, (int(n_local), pairs), group=group)
    counts = [g[0] for g in gathered]
    seams = []
    for k, r in enumerate(live[:-1]):
        if len(gathered[r][1]):
            seams.append((r, live[k + 1], gathered[r][1]))
    luts, n_total = merge_components(counts, seams)
    raw = None
    if mine:
        if want_stats:
            raw = engine.cc_stats_raw(labels, n_local)   # on the local labels: the table maps the rows afterwards
        engine.relabel(labels, luts[rank])
    stats = None
    if want_stats:
        raws = [None] * world
        dist.gather_object(raw, raws if rank == 0 else None, dst=0, group=group)
        if rank == 0:
            stats = merge_stats(luts, raws, [s[0] for s in slabs], shape, n_total)
    return labels, n_total, stats
