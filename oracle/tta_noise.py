"""How much does the reference's test-time-augmentation NOISE move the mask?  (missing #4)

The reference adds RandGaussianNoise(prob=1.0, mean=0.0, std=0.001) to every window of its 12 augmented passes
(inference/sliding_window_inferer.py:211-215; MONAI 1.2.0: per call a standard deviation drawn uniformly from [0, std], then
x + N(mean, that sigma) [3P-recall]) - on raw uint16 intensities of 10^2..10^4.  The build declares it nil and runs the 13
passes as 3 distinct ones weighted 5:4:4.  This script measures the claim with the oracle: the 13-pass
schedule in the reference's arithmetic (fp16 logits summed in fp16, uint8 count, fp16 divide) once without and once WITH the
noise (fresh noise per pass and window, seeded), on the 256^3 crop of the parity tests; reports the flipped voxels.
TEST INFRASTRUCTURE ONLY.   usage: python -m oracle.tta_noise [--out profiles/...json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import delivr_oracle as orc  # noqa: E402
from oracle.parity import LogitCache, flip_report, reference_arithmetic  # noqa: E402

ROI, CROP = (128, 128, 128), (256, 256, 256)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--std", type=float, default=1e-3)
    ap.add_argument("--cache", default="/tmp/wino_gate_cache")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    from delivr_cfos_amd.synth import synth_volume_np

    net = orc.build_unet(seed=0)
    orc.randomize_affine(net, seed=1)
    vol = synth_volume_np(CROP, seed=21)
    t0 = time.time()
    # noise-free: the logits of the three distinct passes (cached by oracle.winograd_gate when it ran)
    clean = LogitCache(lambda x: orc.unet_forward(net, x))
    p = os.path.join(a.cache, "oracle32_random.npz")
    if os.path.isfile(p):
        z = np.load(p)
        for k in z.files:
            f, i = k.split("_")
            clean.store[(None if f == "n" else int(f), int(i))] = z[k]
    ref = reference_arithmetic(orc, vol, ROI, clean, tta=True)
    print(f"[{time.time() - t0:.0f} s] noise-free 13 passes: {int(ref['mask'].sum())} foreground voxels", flush=True)
    # with the noise: every (pass, window) gets its own draw, so nothing is cached
    rng = np.random.default_rng(2024)
    acc16 = np.zeros(CROP, dtype=np.float16)
    cnt = np.zeros(CROP, dtype=np.uint8)
    sigmas = []
    for pi, flip in enumerate(orc.pass_schedule(True)):
        def predict(x, _first=(pi == 0)):
            if _first:  # pass 0 runs without tta (inference.py:262)
                return orc.unet_forward(net, x)
            sigma = rng.uniform(0.0, a.std)
            sigmas.append(sigma)
            return orc.unet_forward(net, x + rng.normal(0.0, sigma, size=x.shape).astype(np.float32))

        orc.sliding_window_pass(vol, ROI, predict, acc16, cnt, 0.5, flip, 1, threshold=0, fp16=True)
        print(f"[{time.time() - t0:.0f} s] noisy pass {pi + 1}/13 done", flush=True)
    mask = orc.finalize(acc16, cnt, vol, CROP, 0.5, 30)
    rep = flip_report(mask, ref["mask"], ref["mean"])
    with np.errstate(divide="ignore", invalid="ignore"):
        mean16 = (acc16 / cnt).astype(np.float32)
    rep["max_abs_mean_logit_shift"] = float(np.nanmax(np.abs(mean16 - ref["mean"])))
    rep["noise_std_upper"] = a.std
    rep["sigma_draws"] = len(sigmas)
    print(json.dumps(rep))
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
