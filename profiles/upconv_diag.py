"""Where the time of the persistent upconv kernel goes: the same forward set with its stores and/or halo loads dropped
(DLV_UPCONV_DBG: a buffer resource of zero records - the instruction stream stays), and the simple kernel beside it.
Timing only: the results of the dbg engines are wrong by construction, and only the DIAGNOSTIC library reads the switch:
    make -C delivr_cfos_amd/csrc diag && DLV_LIB=libdelivr_hip_diag.so python profiles/upconv_diag.py
usage: python profiles/upconv_diag.py [Z,Y,X default 256,256,512]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,256,512").split(","))
sd = {"state_dict": random_state_dict(0)}
settings = [("persistent", {}), ("no stores", {"DLV_UPCONV_DBG": "1"}), ("no halo loads", {"DLV_UPCONV_DBG": "2"}),
            ("neither", {"DLV_UPCONV_DBG": "3"}), ("simple kernel", {"DLV_UPCONV_SIMPLE": "1"})]
engs = []
for tag, env in settings:
    for k in ("DLV_UPCONV_DBG", "DLV_UPCONV_SIMPLE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = HipEngine(0)
    e.load_state_dict(sd)
    e.set_lanes(1)
    engs.append((tag, e))
vol = synth_volume_torch(shape, 1, engs[0][1].device, dense=True)
for tag, e in engs:
    ts = []
    for rnd in range(4):
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        e.prof_reset()
        e.prof_enable(True)
        e.sw_infer(e.make_sw_params(shape, (128, 128, 128), 0.0, None, 0, "fp16"), vol, acc)
        e.sync()
        e.prof_enable(False)
        rep = e.prof_report()
        if rnd:
            ts.append({k: v["total_ms"] * 1e3 / v["launches"] for k, v in rep.items() if v["launches"] and ("upconv" in k or "_add" in k)})
    print(f"{tag:14s} " + "  ".join(f"{k} {np.median([t[k] for t in ts]):7.1f} us" for k in ts[0]), flush=True)
