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


def _needs_host_staging(t, dist) -> bool:
    """gloo moves CPU tensors only for point-to-point; RCCL ("nccl") moves HBM directly over xGMI."""
    try:
        return bool(t.is_cuda) and dist.get_backend() == "gloo"
    except Exception:
        return False


def exchange_seams(acc, plan: ShardPlan, rank: int, dist, group=None) -> None:
    """acc: (Zp, Yp, Xp) fp32 tensor (full padded extent on every rank; only the computed planes are
    non-zero).  After the call the planes owned by `rank` hold the complete sum.  Contributions are
    added in increasing source-rank order so that the result does not depend on arrival order."""
    import torch

    stage = _needs_host_staging(acc, dist)
    sends = plan.sends(rank)
    recvs = plan.recvs(rank)
    ops, bufs = [], []
    for dst, lo, hi in sends:
        t = acc[lo:hi].contiguous()
        ops.append(dist.P2POp(dist.isend, t.cpu() if stage else t, dst, group=group))
    for src, lo, hi in recvs:
        buf = torch.empty_like(acc[lo:hi], device="cpu") if stage else torch.empty_like(acc[lo:hi])
        bufs.append((src, lo, hi, buf))
        ops.append(dist.P2POp(dist.irecv, buf, src, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for src, lo, hi, buf in sorted(bufs, key=lambda t: t[0]):
        acc[lo:hi] += buf.to(acc.device) if stage else buf


def broadcast_weights(engine, dist, rank: int, features: Optional[Sequence[int]] = None, src: int = 0, group=None) -> None:
    """Rank `src` has called engine.load_state_dict(); the others receive the packed blob
    (fp32 + MFMA-packed bf16 parameters, ~35 MB) with ONE broadcast."""
    import torch

    meta = torch.zeros(6, dtype=torch.int64, device=engine.device)
    if rank == src:
        meta[:] = torch.tensor(engine.features, dtype=torch.int64)
    dist.broadcast(meta, src, group=group)
    if rank != src:
        engine.alloc_weight_blob([int(v) for v in meta.tolist()])
    blob = engine.weight_blob()
    engine.sync()
    dist.broadcast(blob, src, group=group)
    torch.cuda.synchronize(engine.device)


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
                   want_prob=False):
    """Threshold + eroded re-mask of the planes `rank` owns.  The erosion is evaluated on the reference's
    Arrayterator z-block grid (inference/inference.py:53), so the sub-volume handed to the kernel starts and
    ends on block boundaries; every rank holds the whole raw volume, only the accumulator is sharded.
    Returns (slab uint8 (n_owned, Y, X), prob or None, (lo, hi))."""
    from .hostlogic import arrayterator_zblock

    Z, Y, X = (int(v) for v in stack_shape)
    nb = arrayterator_zblock((Z, Y, X))
    lo, hi = plan.z_owned[rank]
    hi = min(hi, Z)
    if hi <= lo:
        return None, None, (lo, lo)
    blo, bhi = (lo // nb) * nb, min(-(-hi // nb) * nb, Z)
    res = engine.finalize(acc[blo:bhi], None if cnt is None else cnt[blo:bhi], vol[blo:bhi], (bhi - blo, Y, X),
                          threshold, erode_iters, nb, want_prob=want_prob)
    mask, prob = res if want_prob else (res, None)
    return mask[lo - blo: hi - blo], (None if prob is None else prob[lo - blo: hi - blo]), (lo, hi)
