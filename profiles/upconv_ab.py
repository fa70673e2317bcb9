"""A/B of the folded UpCat conv (upconv.hip: transposed conv folded into the first conv of upcat_1) against the unfolded path
(DLV_NO_UPCONV=1: transposed-conv kernel + 64-channel conv) in one process: per-kernel HIP-event times on a dense volume of
128^3 windows (one lane), difference of the blended logits, and both against the fp32 VALU path on a small volume.
usage: python profiles/upconv_ab.py [rounds, default 3] [Z,Y,X, default 256,256,512] [precision fp16]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "256,256,512").split(","))
prec = sys.argv[3] if len(sys.argv) > 3 else "fp16"
sd = {"state_dict": random_state_dict(0)}
engs = {}
for tag in ("folded", "unfolded"):
    if tag == "unfolded":
        os.environ["DLV_NO_UPCONV"] = "1"
    else:
        os.environ.pop("DLV_NO_UPCONV", None)
    e = HipEngine(0)
    e.load_state_dict(sd)
    e.set_lanes(1)
    engs[tag] = e
os.environ.pop("DLV_NO_UPCONV", None)
# accuracy against the fp32 VALU path of the same library on a small volume (windows of 64^3 and of 128^3)
for small, roi in (((64, 128, 128), (64, 64, 64)), ((128, 128, 256), (128, 128, 128))):
    vol = synth_volume_torch(small, 3, engs["folded"].device, dense=True)
    ref = torch.zeros(small, dtype=torch.float32, device="cuda")
    e = engs["folded"]
    e.sw_infer(e.make_sw_params(small, roi, 0.5, None, 0, "fp32"), vol, ref)
    e.sync()
    r = ref.cpu().numpy()
    for tag, e in engs.items():
        acc = torch.zeros(small, dtype=torch.float32, device="cuda")
        e.sw_infer(e.make_sw_params(small, roi, 0.5, None, 0, prec), vol, acc)
        e.sync()
        a = acc.cpu().numpy()
        print(f"{small} roi {roi[0]} [{tag:8s}] vs fp32: rel rms {np.sqrt(np.mean((a - r) ** 2)) / r.std():.3e}  sign agreement {((a >= 0) == (r >= 0)).mean():.5f}", flush=True)
vol = synth_volume_torch(shape, 1, engs["folded"].device, dense=True)
roi = (128, 128, 128)
res, accs = {}, {}
for rnd in range(rounds + 1):
    for tag, e in engs.items():
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        e.prof_reset()
        e.prof_enable(True)
        e.sync()
        w0 = time.perf_counter()
        e.sw_infer(e.make_sw_params(shape, roi, 0.0, None, 0, prec), vol, acc)
        e.sync()
        wall = time.perf_counter() - w0
        e.prof_enable(False)
        rep = e.prof_report()
        if rnd == 0:
            accs[tag] = acc.cpu().numpy()
            continue
        r = res.setdefault(tag, {"wall_ms": [], "kernels": {}})
        r["wall_ms"].append(wall * 1e3)
        for k, v in rep.items():
            if v["launches"]:
                r["kernels"].setdefault(k, []).append(v["total_ms"] * 1e3 / v["launches"])
base = accs["unfolded"]
for tag in ("unfolded", "folded"):
    r = res[tag]
    d = accs[tag] - base
    print(f"{tag}: wall {np.median(r['wall_ms']):.2f} ms per forward set  rel rms vs unfolded {np.sqrt(np.mean(d ** 2)) / base.std():.2e}  "
          f"sign agreement {((accs[tag] >= 0) == (base >= 0)).mean():.6f}")
    ks = {k: float(np.median(v)) for k, v in r["kernels"].items()}
    for k in sorted(ks, key=lambda k: -ks[k])[:9]:
        print(f"    {k:36s} {ks[k]:9.1f} us")
