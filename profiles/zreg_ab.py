"""A/B of the z-march conv builds inside one process (same device, interleaved rounds): per-kernel HIP-event times on a
dense volume of 128^3 windows (one lane), and the difference of the blended logits between the builds.
usage: python profiles/zreg_ab.py [variants, default "0,50"] [rounds, default 3] [Z,Y,X, default 256,256,512]"""
import json
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,50").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = tuple(int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "256,256,512").split(","))
prec = sys.argv[4] if len(sys.argv) > 4 else "fp16"
eng = HipEngine(0)
eng.load_state_dict({"state_dict": random_state_dict(0)})
eng.set_lanes(1)
vol = synth_volume_torch(shape, 1, eng.device, dense=True)
roi = (128, 128, 128)
res, accs = {}, {}
for rnd in range(rounds + 1):  # round 0 = warm-up
    for v in variants:
        eng.set_zm_variant(v)
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        eng.prof_reset()
        eng.prof_enable(True)
        t0 = torch.cuda.Event(enable_timing=True)
        eng.sync()
        import time
        w0 = time.perf_counter()
        try:
            eng.sw_infer(eng.make_sw_params(shape, roi, 0.0, None, 0, prec), vol, acc)
        except Exception as e:  # timing-only ablation builds compute garbage: the range guard (DLV_ERANGE) fires at the end of the pass
            if os.environ.get("DLV_ALLOW_WRONG_RESULTS") != "1":
                raise
            print("ignored:", str(e)[:80], file=sys.stderr)
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
base = accs[variants[0]]
out = {"shape": shape, "precision": prec, "variants": {}}
for v in variants:
    r = res[v]
    ks = {k: {"us_med": float(np.median(d["us"])), "us_min": float(np.min(d["us"])), "tflops_med": float(np.median(d["tflops"]))}
          for k, d in r["kernels"].items()}
    diff = accs[v] - base
    out["variants"][v] = {"wall_ms_med": float(np.median(r["wall_ms"])), "kernels": ks,
                          "rel_rms_vs_first": float(np.sqrt(np.mean(diff ** 2)) / base.std()),
                          "sign_agreement_vs_first": float(((accs[v] >= 0) == (base >= 0)).mean())}
    print(f"variant {v}: wall {out['variants'][v]['wall_ms_med']:.1f} ms  rel rms vs {variants[0]}: {out['variants'][v]['rel_rms_vs_first']:.2e}")
    for k in sorted(ks, key=lambda k: -ks[k]["us_med"]):
        if True:
            print(f"    {k:32s} {ks[k]['us_med']:9.1f} us  {ks[k]['tflops_med']:7.1f} TFLOP/s")
print(json.dumps(out))
