"""The HBM-class stages either side of the pass (BASELINE configs 4/5) on a C3-size synthetic volume, without the pass itself:
finalize (threshold + L1-30 eroded re-mask), CCL-26 + statistics on the synthetic cells, block mean (4,15,15), spline-2
zoom back to full size.  Per stage: ms, algorithmic bytes, fraction of 8 TB/s, and the per-kernel HIP-event split.
usage: python profiles/extras_perf.py [Z,Y,X default 1024,2048,2048] [reps default 3]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.hostlogic import arrayterator_zblock  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024,2048,2048").split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vox = shape[0] * shape[1] * shape[2]
eng = HipEngine(0)
vol = synth_volume_torch(shape, 2, eng.device)
acc = torch.empty(shape, dtype=torch.float32, device=eng.device)
for z in range(0, shape[0], 64):  # logit > 0 on the synthetic cells
    acc[z:z + 64] = (vol[z:z + 64].view(torch.int16).to(torch.float32) - 6500.0) * 1e-3
torch.cuda.synchronize()


def timed(fn, n=reps):
    fn()
    eng.sync()
    torch.cuda.synchronize()
    eng.prof_reset()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    out = None
    for _ in range(n):
        out = None
        out = fn()
    eng.sync()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    eng.prof_enable(False)
    ks = {k: round(1e3 * e["total_ms"] / max(e["launches"], 1), 1) for k, e in eng.prof_report().items() if e["launches"]}
    return ms, out, ks


def entry(ms, nbytes, ks):
    return {"ms": round(ms, 3), "GBps": round(nbytes / (ms * 1e6), 1), "hbm_frac": round(nbytes / (ms * 1e-3) / 8e12, 4), "kernels_us": ks}


res = {"shape": shape}
zb = arrayterator_zblock(shape)
ms, mask, ks = timed(lambda: eng.finalize(acc, None, vol, shape, 0.5, 30, zb))
res["finalize"] = entry(ms, vox * 7, ks)
mask = mask.contiguous()
ms, (labels, n), ks = timed(lambda: eng.ccl26(mask))
res["ccl26"] = entry(ms, vox * 5, ks)
res["ccl26"]["components"] = int(n)
ms, _, ks = timed(lambda: eng.cc_stats(labels, n), 1)
res["cc_stats"] = entry(ms, vox * 4, ks)
del labels
ms, ds, ks = timed(lambda: eng.block_mean_u16(vol, (4, 15, 15)))
res["block_mean"] = entry(ms, vox * 2 + ds.numel() * 2, ks)
small = (ds.to(torch.int32) > 0).to(torch.uint8)
ms, _, ks = timed(lambda: eng.zoom_spline2_u8(small, shape))
res["zoom"] = entry(ms, vox + small.numel(), ks)
for k, v in res.items():
    print(k, json.dumps(v) if isinstance(v, dict) else v)
