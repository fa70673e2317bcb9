"""Seeded synthetic light-sheet volumes of the shapes BASELINE.json names (no datasets are
reachable offline).  Statistics follow SURVEY.md section 8(d) / the reference's training patches
(training_data/cFos/raw: uint16-range tissue, median ~1.6-3.3k; gt: 10-40-voxel blobs at
~4e-4 per voxel): background 0 outside a centred ellipsoid "brain" (semi-axes 0.45*dim), tissue =
clip(N(2500, 600^2), 200, 20000) (strictly > 0 so the eroded re-mask of
inference/inference.py:77-84 keeps it), cells = small ellipsoidal blobs with peak +3000...+30000.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

CELL_DENSITY = 4e-4


def _blob_kernel_np() -> np.ndarray:
    r = np.arange(-2, 3, dtype=np.float32)
    zz, yy, xx = np.meshgrid(r, r, r, indexing="ij")
    return np.exp(-(zz**2 + yy**2 + xx**2) / (2 * 0.9**2)).astype(np.float32)


def synth_volume_np(shape: Sequence[int], seed: int = 1, dense: bool = False) -> np.ndarray:
    """CPU generator for tests and small configs.  Returns uint16 (Z,Y,X)."""
    from scipy.ndimage import convolve

    rng = np.random.default_rng(seed)
    Z, Y, X = shape
    tissue = np.clip(rng.normal(2500.0, 600.0, size=shape), 200, 20000).astype(np.float32)
    imp = (rng.random(shape) < CELL_DENSITY).astype(np.float32)
    imp *= rng.uniform(3000.0, 30000.0, size=shape).astype(np.float32)
    cells = convolve(imp, _blob_kernel_np(), mode="constant")
    vol = np.clip(tissue + cells, 0, 65535)
    if not dense:
        z = (np.arange(Z) - (Z - 1) / 2) / (0.45 * Z)
        y = (np.arange(Y) - (Y - 1) / 2) / (0.45 * Y)
        x = (np.arange(X) - (X - 1) / 2) / (0.45 * X)
        inside = (z[:, None, None] ** 2 + y[None, :, None] ** 2 + x[None, None, :] ** 2) <= 1.0
        vol = vol * inside
    return vol.astype(np.uint16)


def synth_volume_torch(shape: Sequence[int], seed: int, device, dense: bool = False, chunk: int = 64):
    """Device-side generator for the bench volumes (up to 1024x2048x2048): built z-chunk by
    z-chunk so that the fp32 temporaries stay ~1 GB.  Returns a uint16 torch tensor (Z,Y,X) on
    ``device`` (torch is the memory container here, not the product)."""
    return synth_planes_torch(shape, seed, device, 0, int(shape[0]), dense=dense, chunk=chunk)


def synth_planes_torch(shape: Sequence[int], seed: int, device, z_lo: int, z_hi: int, dense: bool = False, chunk: int = 64):
    """Planes [z_lo, z_hi) of the same volume.  Every z-chunk has its own generator seed, so a rank of a sharded run
    generates exactly its slab (cells are blurred inside a chunk: a blob never crosses a chunk boundary)."""
    import torch
    Z, Y, X = shape
    out = torch.empty((z_hi - z_lo, Y, X), dtype=torch.uint16, device=device)
    k1 = torch.tensor(np.exp(-(np.arange(-2, 3) ** 2) / (2 * 0.9**2)), dtype=torch.float32, device=device)
    yy = ((torch.arange(Y, device=device) - (Y - 1) / 2) / (0.45 * Y)) ** 2
    xx = ((torch.arange(X, device=device) - (X - 1) / 2) / (0.45 * X)) ** 2
    for z0 in range((z_lo // chunk) * chunk, z_hi, chunk):
        z1 = min(z0 + chunk, Z)
        n = z1 - z0
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + z0 // chunk)
        t = torch.randn((n, Y, X), generator=g, device=device).mul_(600.0).add_(2500.0).clamp_(200.0, 20000.0)
        imp = (torch.rand((n, Y, X), generator=g, device=device) < CELL_DENSITY).float()
        imp.mul_(torch.rand((n, Y, X), generator=g, device=device).mul_(27000.0).add_(3000.0))
        c = imp
        for ax in range(3):  # separable 5-tap blur as shifted adds (no library convolution)
            acc = torch.zeros_like(c)
            for k in range(-2, 3):
                wgt = float(k1[k + 2])
                n_ax = c.shape[ax]
                if abs(k) >= n_ax:
                    continue
                dst = [slice(None)] * 3
                src = [slice(None)] * 3
                dst[ax] = slice(max(0, -k), n_ax - max(0, k))
                src[ax] = slice(max(0, k), n_ax - max(0, -k))
                acc[tuple(dst)].add_(c[tuple(src)], alpha=wgt)
            c = acc
        t.add_(c).clamp_(0.0, 65535.0)
        if not dense:
            zz = ((torch.arange(z0, z1, device=device) - (Z - 1) / 2) / (0.45 * Z)) ** 2
            inside = (zz[:, None, None] + yy[None, :, None] + xx[None, None, :]) <= 1.0
            t.mul_(inside)
        a, b = max(z0, z_lo), min(z1, z_hi)
        out[a - z_lo:b - z_lo] = t[a - z0:b - z0].to(torch.int32).to(torch.uint16)
        del t, imp, c
    return out
