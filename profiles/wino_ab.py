"""A/B of the conv algorithms (dlv_set_conv_algo: direct vs Winograd F(2,3) along x) inside one process, interleaved rounds:
(1) the Cin-32 conv blocks in isolation against a torch fp32 reference on the same fp16-rounded inputs, (2) per-kernel
HIP-event times on a dense volume of 128^3 windows (one lane) and the difference of the blended logits.
usage: python profiles/wino_ab.py [rounds, default 3] [Z,Y,X, default 256,256,512]"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "256,256,512").split(","))
sd = random_state_dict(0)
eng = HipEngine(0)
eng.load_state_dict({"state_dict": sd})
eng.set_lanes(1)
names = {1: "conv_0.conv_1", 3: "down_1.convs.conv_1", 17: "upcat_1.convs.conv_1"}
for li, (D, H, W) in ((1, (24, 40, 64)), (1, (64, 64, 64)), (3, (32, 48, 96)), (17, (128, 128, 128))):
    g = torch.Generator().manual_seed(li * 100 + D)
    x = torch.randn((2 if D < 128 else 1, 32, D, H, W), generator=g).half().float()
    k = "module." + names[li]
    with torch.no_grad():
        raw = F.conv3d(x, sd[k + ".conv.weight"].half().float(), sd[k + ".conv.bias"], padding=1)
        ref = F.mish(F.instance_norm(raw, weight=sd[k + ".adn.N.weight"], bias=sd[k + ".adn.N.bias"], eps=1e-5))
    for algo in ("direct", "winograd"):
        eng.set_conv_algo(algo)
        out = eng.debug_layer_bf16(0, li, x.cuda(), None, precision="fp16").cpu()
        err = (out - ref).abs()
        print(f"conv block {li} {D}x{H}x{W} [{algo}]: max abs err {float(err.max()):.3e}  mean {float(err.mean()):.3e}", flush=True)
vol = synth_volume_torch(shape, 1, eng.device, dense=True)
roi = (128, 128, 128)
res, accs = {}, {}
for rnd in range(rounds + 1):
    for v in ("direct", "winograd"):
        eng.set_conv_algo(v)
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        eng.prof_reset()
        eng.prof_enable(True)
        eng.sync()
        w0 = time.perf_counter()
        eng.sw_infer(eng.make_sw_params(shape, roi, 0.0, None, 0, "fp16"), vol, acc)
        eng.sync()
        wall = time.perf_counter() - w0
        eng.prof_enable(False)
        rep = eng.prof_report()
        if rnd == 0:
            accs[v] = acc.cpu().numpy()
            continue
        r = res.setdefault(v, {"wall_ms": [], "kernels": {}})
        r["wall_ms"].append(wall * 1e3)
        for k, e in rep.items():
            if e["launches"]:
                d = r["kernels"].setdefault(k, {"us": [], "tflops": []})
                us = e["total_ms"] * 1e3 / e["launches"]
                d["us"].append(us)
                d["tflops"].append(e["flops"] / e["launches"] / (us * 1e-6) / 1e12 if e["flops"] else 0.0)
base = accs["direct"]
out = {"shape": shape, "variants": {}}
for v in ("direct", "winograd"):
    r = res[v]
    ks = {k: {"us_med": float(np.median(d["us"])), "tflops_med": float(np.median(d["tflops"]))} for k, d in r["kernels"].items()}
    diff = accs[v] - base
    out["variants"][v] = {"wall_ms_med": float(np.median(r["wall_ms"])), "kernels": ks,
                          "rel_rms_vs_direct": float(np.sqrt(np.mean(diff ** 2)) / base.std()),
                          "sign_agreement_vs_direct": float(((accs[v] >= 0) == (base >= 0)).mean())}
    print(f"{v}: wall {out['variants'][v]['wall_ms_med']:.1f} ms  rel rms vs direct {out['variants'][v]['rel_rms_vs_direct']:.2e}  "
          f"sign agreement {out['variants'][v]['sign_agreement_vs_direct']:.6f}")
    for k in sorted(ks, key=lambda k: -ks[k]["us_med"])[:12]:
        print(f"    {k:34s} {ks[k]['us_med']:9.1f} us  {ks[k]['tflops_med']:7.1f} TFLOP/s (direct-conv FLOPs)")
print(json.dumps(out))
