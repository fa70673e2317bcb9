"""Folded UpCat conv on window shapes that exercise the persistent kernel's tile walk (tiles per window not a multiple of 8,
fewer tiles than workgroups, non-cubic windows) and the one-tile kernel (coarse width not a multiple of 16): folded vs the
one-tile kernel vs the unfolded path, one to three windows per launch.  usage: python profiles/tools/upconv_fuzz.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_np  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

sd = {"state_dict": random_state_dict(3)}
engs = {}
for tag, env in (("folded", {}), ("simple", {"DLV_UPCONV_SIMPLE": "1"}), ("unfolded", {"DLV_NO_UPCONV": "1"})):
    for k in ("DLV_UPCONV_SIMPLE", "DLV_NO_UPCONV"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = HipEngine(0)
    e.load_state_dict(sd)
    engs[tag] = e
for k in ("DLV_UPCONV_SIMPLE", "DLV_NO_UPCONV"):
    os.environ.pop(k, None)
rois = [(48, 48, 96), (32, 64, 160), (80, 48, 64), (64, 32, 32), (32, 32, 96), (48, 80, 48), (112, 64, 96), (64, 64, 224), (32, 48, 80)]
worst = 0.0
for roi in rois:
    for nwin in (1, 3, 5):
        shape = (roi[0], roi[1], roi[2] * nwin)
        vol = synth_volume_np(shape, seed=sum(roi) + nwin, dense=True)
        outs, ran = {}, {}
        for tag, e in engs.items():
            acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
            e.prof_reset()
            e.prof_enable(True)
            e.sw_infer(e.make_sw_params(shape, roi, 0.0, None, 0, "fp16"), e.to_device(vol), acc)
            e.sync()
            e.prof_enable(False)
            ran[tag] = sorted(k for k, v in e.prof_report().items() if v["launches"] and "upconv" in k)
            outs[tag] = acc.cpu().numpy()
        std = float(outs["unfolded"].std())
        r1 = float(np.sqrt(np.mean((outs["folded"] - outs["unfolded"]) ** 2)) / std)
        r2 = float(np.sqrt(np.mean((outs["folded"] - outs["simple"]) ** 2)) / std)
        ok = np.isfinite(outs["folded"]).all() and r1 < 2e-3 and r2 < 5e-4
        worst = max(worst, r1)
        print(f"roi {roi} x {nwin} windows: {ran['folded']} vs unfolded {r1:.2e}  vs one-tile kernel {r2:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            sys.exit(1)
print("all ok, worst rel rms vs unfolded", worst)
