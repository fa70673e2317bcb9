"""Mask agreement of the 16-bit path against the fp32 HIP path (which matches the torch-fp32 oracle to 2e-4) on a
256^3 synthetic volume, 128^3 windows, seeded random weights: IoU of the final masks and sign agreement of the
blended logits, plus the same figures restricted to voxels with |mean logit| above a margin."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.synth import synth_volume_torch
from delivr_cfos_amd.weights import random_state_dict

shape, roi = (256, 256, 256), (128, 128, 128)
eng = HipEngine(0)
eng.load_state_dict({"state_dict": random_state_dict(0)})
vol = synth_volume_torch(shape, 1, eng.device)
res = {}
for prec in sys.argv[1:] or ["fp32", "bf16", "fp16"]:
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    eng.sw_infer(eng.make_sw_params(shape, roi, 0.5, None, 0, prec), vol, acc, cnt)
    mask = eng.finalize(acc, cnt, vol, shape, 0.5, 30, 0)
    eng.sync()
    res[prec] = (acc / cnt.float(), mask)
ref_mean, ref_mask = res["fp32"]
out = {}
for prec, (mean, mask) in res.items():
    if prec == "fp32":
        continue
    live = ref_mean > -100
    inter = (mask.bool() & ref_mask.bool()).sum().item()
    union = (mask.bool() | ref_mask.bool()).sum().item()
    agree = ((mean >= 0) == (ref_mean >= 0))[live].float().mean().item()
    rel = ((mean - ref_mean)[live].pow(2).mean().sqrt() / ref_mean[live].std()).item()
    big = live & (ref_mean.abs() > 0.05)
    out[prec] = {"mask_iou": inter / max(union, 1), "sign_agreement": agree, "rel_rms": rel,
                 "sign_agreement_margin_0.05": ((mean >= 0) == (ref_mean >= 0))[big].float().mean().item(),
                 "logit_std": ref_mean[live].std().item(), "fg_fraction": ref_mask.float().mean().item()}
print(json.dumps(out))
