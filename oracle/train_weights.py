"""Trained-like weights for the oracle BasicUNet - TEST INFRASTRUCTURE ONLY (see oracle/delivr_oracle.py's header).

The reference's checkpoint ``models/inference_weights.tar`` (inference/inference.py:199-200,222) is not part of the
snapshot (/root/reference/.MISSING_LARGE_BLOBS:1), so every tolerance used to be shown on seeded RANDOM weights whose
logits straddle zero and whose mask is one giant component.  This script makes a checkpoint whose logits are bimodal
and whose mask is thousands of small blobs - what count_blobs.py:57-114 really sees:

    python -m oracle.train_weights            # build container only (reads /root/reference/training_data)

* network: the oracle BasicUNet carrying ``delivr_cfos_amd.weights.random_state_dict(seed=0)`` (numpy generator: the same
  values on every machine) - regenerated from the seed wherever the fixture is loaded.  Only the two top levels are
  trained (conv_0, down_1, upcat_2, upcat_1, final_conv: 275 137 of the 5 749 377 parameters); the deep levels keep their
  seeded random values, so the fixture is 0.5 MB instead of 11 MB.
* data: random 64^3 crops of the reference's own training patches (training_data/cFos/{raw,gt}: 41 pairs of 100^3,
  float64 raw / uint32 gt NIfTI) and of ``delivr_cfos_amd.synth`` volumes with the generator's own cell map as ground
  truth (the benchmark and the parity crops are synth volumes), one of each per step; flips as augmentation.
* loss: BCE-with-logits (positive weight) + soft Dice; Adam, fixed seeds, 8 torch threads.
* output: ``tests/golden/trained_like_weights.npz`` - the trained tensors as fp16 (both the oracle and the HIP path load
  exactly these fp16 values widened to fp32) + the training log.

``delivr_cfos_amd.weights.trained_like_state_dict()`` merges the fixture into the seeded state dict (host logic, no oracle
import); ``build_trained_like()`` here is the oracle network carrying it (tests, bench.py's cpu_baseline leg).
"""
from __future__ import annotations

import glob
import gzip
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "trained_like_weights.npz")
TRAINED_PREFIXES = ("conv_0.", "down_1.", "upcat_2.", "upcat_1.", "final_conv.")
GT_CELL_LEVEL = 1500.0  # a synth voxel belongs to a cell where the blob component adds more than this to the tissue


def build_trained_like(fixture: str = FIXTURE):
    """The oracle network carrying the fixture's weights (seeded random deep levels + trained top levels)."""
    from delivr_cfos_amd.weights import trained_like_state_dict
    from oracle import delivr_oracle as orc

    net = orc.build_unet(seed=None)
    net.load_state_dict(trained_like_state_dict(fixture, module_prefix=False))
    net.eval()
    return net


def read_nii_gz(path: str) -> np.ndarray:
    """Minimal NIfTI-1 reader (nibabel is not installed): little-endian, 352-byte header, 3-D."""
    raw = gzip.open(path, "rb").read()
    dim = struct.unpack("<8h", raw[40:56])
    dt = struct.unpack("<h", raw[70:72])[0]
    off = int(struct.unpack("<f", raw[108:112])[0])
    np_dt = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}[dt]
    n = dim[1] * dim[2] * dim[3]
    a = np.frombuffer(raw, dtype=np_dt, count=n, offset=off)
    return a.reshape(dim[3], dim[2], dim[1])  # NIfTI is x-fastest: (z,y,x) in C order


def synth_with_cells(shape, seed: int):
    """delivr_cfos_amd.synth.synth_volume_np(dense=True) restated with its cell component exposed: (uint16 volume, gt)."""
    from scipy.ndimage import convolve

    from delivr_cfos_amd.synth import CELL_DENSITY, _blob_kernel_np

    rng = np.random.default_rng(seed)
    tissue = np.clip(rng.normal(2500.0, 600.0, size=shape), 200, 20000).astype(np.float32)
    imp = (rng.random(shape) < CELL_DENSITY).astype(np.float32)
    imp *= rng.uniform(3000.0, 30000.0, size=shape).astype(np.float32)
    cells = convolve(imp, _blob_kernel_np(), mode="constant")
    vol = np.clip(tissue + cells, 0, 65535).astype(np.uint16)
    return vol, (cells > GT_CELL_LEVEL).astype(np.uint8)


def main(steps: int = 360, patch: int = 64, seed: int = 7) -> None:
    import torch
    import torch.nn.functional as F

    from oracle import delivr_oracle as orc

    torch.manual_seed(seed)
    torch.set_num_threads(8)
    rng = np.random.default_rng(seed)
    from delivr_cfos_amd.weights import random_state_dict

    net = orc.build_unet(seed=None)
    net.load_state_dict(random_state_dict(seed=0, module_prefix=False))
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    trained = []
    for k, p in net.named_parameters():
        p.requires_grad_(k.startswith(TRAINED_PREFIXES))
        if p.requires_grad:
            trained.append(k)
    n_tr = sum(p.numel() for p in net.parameters() if p.requires_grad)
    print(f"training {n_tr} of {sum(p.numel() for p in net.parameters())} parameters", flush=True)

    pairs = []
    gt_dir = "/root/reference/training_data/cFos/gt"
    for g in sorted(glob.glob(os.path.join(gt_dir, "*.nii.gz"))):
        r = os.path.join("/root/reference/training_data/cFos/raw", os.path.basename(g))
        if os.path.isfile(r):
            pairs.append((read_nii_gz(r).astype(np.float32), (read_nii_gz(g) > 0).astype(np.float32)))
    print(f"{len(pairs)} reference training pairs; raw median {np.median([np.median(p[0]) for p in pairs]):.0f}, "
          f"gt fraction {np.mean([p[1].mean() for p in pairs]):.2e}", flush=True)
    synth = [synth_with_cells((96, 96, 96), 1000 + i) for i in range(12)]
    print(f"synth gt fraction {np.mean([s[1].mean() for s in synth]):.2e}", flush=True)

    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=steps, pct_start=0.15)
    log = []
    t0 = time.time()
    for step in range(steps):
        xs, ys = [], []
        for src in (pairs[rng.integers(len(pairs))], synth[rng.integers(len(synth))]):
            raw, gt = src
            o = [int(rng.integers(0, n - patch + 1)) for n in raw.shape]
            sl = tuple(slice(a, a + patch) for a in o)
            x, y = raw[sl].astype(np.float32), gt[sl].astype(np.float32)
            for ax in range(3):
                if rng.random() < 0.5:
                    x, y = np.flip(x, ax), np.flip(y, ax)
            xs.append(np.ascontiguousarray(x))
            ys.append(np.ascontiguousarray(y))
        x = torch.from_numpy(np.stack(xs))[:, None]
        y = torch.from_numpy(np.stack(ys))[:, None]
        logit = net(x)
        bce = F.binary_cross_entropy_with_logits(logit, y, pos_weight=torch.tensor(20.0))
        p = torch.sigmoid(logit)
        dice = 1.0 - (2.0 * (p * y).sum() + 1.0) / (p.sum() + y.sum() + 1.0)
        loss = bce + dice
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        if step % 10 == 0 or step == steps - 1:
            with torch.no_grad():
                m = logit > 0
                inter = float((m & (y > 0)).sum())
                f1 = 2 * inter / max(float(m.sum() + (y > 0).sum()), 1.0)
            log.append((step, float(loss.detach()), float(bce.detach()), float(dice.detach()), f1))
            print(f"step {step:4d} loss {float(loss.detach()):.4f} bce {float(bce.detach()):.4f} dice {float(dice.detach()):.4f} F1 {f1:.3f} "
                  f"logit [{float(logit.detach().min()):.1f}, {float(logit.detach().max()):.1f}]  {time.time() - t0:.0f} s", flush=True)

    net.eval()
    sd = net.state_dict()
    out = {"w:" + k: sd[k].numpy().astype(np.float16) for k in trained}
    out["log"] = np.asarray(log, dtype=np.float64)
    out["meta"] = np.asarray([steps, patch, seed, n_tr], dtype=np.int64)
    np.savez_compressed(FIXTURE, **out)
    print(f"wrote {FIXTURE}: {os.path.getsize(FIXTURE)} bytes", flush=True)
    # what the fixture does on a held-out synth crop (fp16-rounded weights, as loaded everywhere)
    net2 = build_trained_like(FIXTURE)
    vol, gt = synth_with_cells((96, 96, 96), 4242)
    lg = orc.unet_forward(net2, vol[:64, :64, :64].astype(np.float32)[None, None])[0, 0]
    g = gt[:64, :64, :64] > 0
    m = lg > 0
    print(f"held-out synth 64^3: F1 {2 * (m & g).sum() / max(m.sum() + g.sum(), 1):.3f}, mask fraction {m.mean():.2e}, "
          f"|logit| < 0.5 on {float((np.abs(lg) < 0.5).mean()):.2e} of the voxels, logit range [{lg.min():.1f}, {lg.max():.1f}]")


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main(*(int(a) for a in sys.argv[1:]))
