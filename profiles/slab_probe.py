"""What does ONE rank of an N-way sharded C3 pass cost on this GPU?  Runs the balanced plan's slab of rank r (its window range
on its Z-slab, no exchange) and compares the time per active window with the full single-GPU pass: the difference is what the
pipeline drain at the colour-class boundaries and the partially filled last batches cost when a rank has 1/N of the windows.
usage: python profiles/slab_probe.py [N=8] [ranks, default "0,3"]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.parallel import plan_from_params  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import trained_like_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,3").split(",")]
shape, roi = (1024, 2048, 2048), (128, 128, 128)
eng = HipEngine(0)
eng.load_state_dict({"state_dict": trained_like_state_dict(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trained_like_weights.npz"))})
vol = synth_volume_torch(shape, 2, eng.device)
p_all = eng.make_sw_params(shape, roi, 0.5, None, 0, "fp16")
wmax = eng.window_max(p_all, vol)
active = wmax > 0


def timed(params, v, acc, reps=2):
    eng.sw_infer(params, v, acc)
    eng.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st = eng.sw_infer(params, v, acc)
    eng.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, st


acc = torch.zeros(shape, dtype=torch.float32, device=eng.device)
t_full, st = timed(p_all, vol, acc, 1)
n_act = int(active.sum())
print(f"full pass: {t_full * 1e3:.0f} ms, {n_act} active windows, {t_full * 1e3 / n_act:.3f} ms per active window")
del acc
plan = plan_from_params(p_all, N, np.where(active, 1.0, 0.02).astype(np.float32))
for r in ranks:
    wb, we = plan.win_ranges[r]
    lo, hi = plan.z_computed[r]
    a = torch.zeros((hi - lo,) + shape[1:], dtype=torch.float32, device=eng.device)
    pr = eng.make_sw_params(shape, roi, 0.5, None, 0, "fp16", win_range=(wb, we), slab=(lo, hi - lo))
    t, st = timed(pr, vol[lo:hi], a)
    na = int(active[wb:we].sum())
    print(f"rank {r}/{N}: windows [{wb},{we}) {na} active, planes [{lo},{hi}): {t * 1e3:.0f} ms = {t * 1e3 / max(na, 1):.3f} ms per active window "
          f"({(t / max(na, 1)) / (t_full / n_act):.3f} x the full pass), launches {st['n_forward_launches']}")
    del a
