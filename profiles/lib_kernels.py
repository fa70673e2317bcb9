"""Per-label HIP-event times of one forward set (16 dense windows of 128^3, one lane) for the library named by DLV_LIB:
the per-kernel view behind a lib_ab.sh comparison.  usage: DLV_LIB=libX.so python profiles/lib_kernels.py [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import random_state_dict  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shape = (256, 256, 512)
e = HipEngine(0)
e.load_state_dict({"state_dict": random_state_dict(0)})
e.set_lanes(1)
vol = synth_volume_torch(shape, 1, e.device, dense=True)
ts = {}
walls = []
for rnd in range(rounds + 1):
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    e.prof_reset()
    e.prof_enable(True)
    e.sw_infer(e.make_sw_params(shape, (128, 128, 128), 0.0, None, 0, "fp16"), vol, acc)
    e.sync()
    e.prof_enable(False)
    if rnd:
        tot = 0.0
        for k, v in e.prof_report().items():
            if v["launches"]:
                ts.setdefault(k, []).append((v["total_ms"] * 1e3, v["launches"]))
                tot += v["total_ms"] * 1e3
        walls.append(tot)
print(os.environ.get("DLV_LIB", "libdelivr_hip.so"), f"sum of kernels {np.median(walls):.0f} us")
for k in sorted(ts, key=lambda k: -np.median([t for t, _ in ts[k]])):
    print(f"   {k:36s} {np.median([t for t, _ in ts[k]]):9.1f} us total in {ts[k][0][1]:3d} launches")
