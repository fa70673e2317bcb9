"""Logit-margin histogram per storage format (hard part iv): how far from the decision boundary are
the voxels whose mask bit a 16-bit format flips?  256^3 volume (synthetic brain, seed 1), 27 windows of 128^3, seeded
random weights; reference = the fp32 HIP path (<= 2e-4 from the torch-fp32 oracle).  Per format: histogram of the
reference's |mean logit| over all tissue voxels, the same histogram over the voxels whose sign differs, the largest
margin at which a flip occurs, and mask IoU.   python profiles/logit_margin.py > profiles/r02_logit_margin_hist.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

shape, roi = (256, 256, 256), (128, 128, 128)
eng = HipEngine(0)
eng.load_state_dict({"state_dict": random_state_dict(0)})
vol = synth_volume_torch(shape, 1, eng.device)
res = {}
for prec in ("fp32", "fp16", "bf16"):
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    eng.sw_infer(eng.make_sw_params(shape, roi, 0.5, None, 0, prec), vol, acc, cnt)
    mask = eng.finalize(acc, cnt, vol, shape, 0.5, 30, 0)
    eng.sync()
    res[prec] = (acc / cnt.float(), mask)
ref_mean, ref_mask = res["fp32"]
live = ref_mean > -100  # not a skipped (background) window
edges = [0.0, 1e-4, 2e-4, 5e-4, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2, 0.1, 0.2, 0.5, 1.0, 1e9]
e = torch.tensor(edges, device="cuda")
margin = ref_mean.abs()[live]
out = {"volume": list(shape), "roi": list(roi), "weights": "seeded random (no trained checkpoint in the snapshot)",
       "reference": "fp32 HIP path", "bin_edges_abs_mean_logit": edges,
       "logit_mean": ref_mean[live].mean().item(), "logit_std": ref_mean[live].std().item(),
       "voxels": int(live.sum().item()), "hist_all_voxels": torch.histc(torch.bucketize(margin, e[1:-1]).float(), bins=len(edges) - 1, min=0, max=len(edges) - 1).long().tolist(),
       "formats": {}}
for prec in ("fp16", "bf16"):
    mean, mask = res[prec]
    flip = ((mean >= 0) != (ref_mean >= 0))[live]
    fm = margin[flip]
    inter = (mask.bool() & ref_mask.bool()).sum().item()
    union = (mask.bool() | ref_mask.bool()).sum().item()
    out["formats"][prec] = {
        "flipped_voxels": int(flip.sum().item()),
        "flipped_fraction": flip.float().mean().item(),
        "hist_flipped_voxels": torch.histc(torch.bucketize(fm, e[1:-1]).float(), bins=len(edges) - 1, min=0, max=len(edges) - 1).long().tolist(),
        "largest_margin_of_a_flip": fm.max().item() if fm.numel() else 0.0,
        "rel_rms_of_mean_logit": ((mean - ref_mean)[live].pow(2).mean().sqrt() / ref_mean[live].std()).item(),
        "max_abs_diff_of_mean_logit": (mean - ref_mean)[live].abs().max().item(),
        "mask_iou_vs_fp32_path": inter / max(union, 1),
    }
print(json.dumps(out, indent=1))
