"""Where does bf16 lose the mask IoU?    TEST INFRASTRUCTURE ONLY (oracle/delivr_oracle.py's header).

The CPU emulation of the 16-bit HIP forward (oracle/winograd_gate.py: every tensor that crosses HBM or enters an MFMA rounded
to the 16-bit format, fp32 accumulation and statistics) with a format PER CONV BLOCK, pushed through the parity chain of
tests/test_gpu_production_shapes.py::test_mask_vs_reference_accumulate_arithmetic (256^3 crop, 27 windows of 128^3, margin-free
seeded random weights: the worst case, where HIP bf16 measures IoU 0.99807 and HIP fp16 0.99975 against the reference's
arithmetic).  Each variant keeps a set of blocks in fp16 and runs the rest in bf16 (or the other way round).

usage:  python -m oracle.bf16_budget [--out profiles/r05t_bf16_budget.json] [--variants all_bf16,...]
Reference network: MONAI BasicUNet, inference/inference.py:190-197 (fp32 in the reference).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import delivr_oracle as orc  # noqa: E402
from oracle.parity import LogitCache, flip_report, fp32_arithmetic, reference_arithmetic  # noqa: E402
from oracle.winograd_gate import CROP, ROI, inorm_stats, mish, rnd  # noqa: E402

LEVEL0 = {"conv_0.conv_0", "conv_0.conv_1", "upcat_1.up", "upcat_1.convs.conv_0", "upcat_1.convs.conv_1"}
LEVEL1 = {"down_1.convs.conv_0", "down_1.convs.conv_1", "upcat_2.up", "upcat_2.convs.conv_0", "upcat_2.convs.conv_1"}
ALL = LEVEL0 | LEVEL1 | {f"down_{l}.convs.conv_{k}" for l in (2, 3, 4) for k in (0, 1)} | {f"upcat_{l}.convs.conv_{k}" for l in (3, 4) for k in (0, 1)} | {
    "upcat_3.up", "upcat_4.up"}
VARIANTS = {  # name -> blocks kept in fp16 (everything else bf16)
    "all_bf16": set(),
    "all_fp16": set(ALL),
    "fp16_last_block": {"upcat_1.convs.conv_1"},
    "fp16_upcat_1": {"upcat_1.up", "upcat_1.convs.conv_0", "upcat_1.convs.conv_1"},
    "fp16_decoder_top2": {"upcat_1.up", "upcat_1.convs.conv_0", "upcat_1.convs.conv_1", "upcat_2.up", "upcat_2.convs.conv_0", "upcat_2.convs.conv_1"},
    "fp16_level0": set(LEVEL0),
    "fp16_levels01": LEVEL0 | LEVEL1,
    "bf16_level0_only": set(ALL) - LEVEL0,          # fp16 everywhere but level 0: is level 0 where the bits go?
    "bf16_last_block_only": set(ALL) - {"upcat_1.convs.conv_1"},
}


class EmuMixed:
    """Emu16 of winograd_gate.py with the format looked up per block (weights, raw store and activated store of that block)."""

    def __init__(self, net, fp16_blocks):
        self.sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        self.fp16 = set(fp16_blocks)

    def fmt(self, name):
        return "fp16" if name in self.fp16 else "bf16"

    def block(self, x16, name):
        f = self.fmt(name)
        w = self.sd[name + ".conv.weight"]
        gamma, beta = self.sd[name + ".adn.N.weight"], self.sd[name + ".adn.N.bias"]
        raw32 = F.conv3d(rnd(x16, f), rnd(w, f), None, 1, 1)  # (an input produced in the other format is re-rounded: a format change costs a pass)
        m, v = inorm_stats(raw32)
        raw16 = rnd(raw32, f)
        sc = gamma.view(1, -1, 1, 1, 1) / torch.sqrt(v + 1e-5)
        sh = beta.view(1, -1, 1, 1, 1) - m * sc
        return rnd(mish(raw16 * sc + sh), f)

    def stem(self, x):
        f = self.fmt("conv_0.conv_0")
        w = self.sd["conv_0.conv_0.conv.weight"]
        raw32 = F.conv3d(x, rnd(w, f), None, 1, 1)
        m, v = inorm_stats(raw32)
        sc = self.sd["conv_0.conv_0.adn.N.weight"].view(1, -1, 1, 1, 1) / torch.sqrt(v + 1e-5)
        sh = self.sd["conv_0.conv_0.adn.N.bias"].view(1, -1, 1, 1, 1) - m * sc
        return rnd(mish(raw32 * sc + sh), f)

    def up(self, x16, name):
        f = self.fmt(name + ".up")
        w, b = self.sd[name + ".upsample.deconv.weight"], self.sd[name + ".upsample.deconv.bias"]
        return rnd(F.conv_transpose3d(rnd(x16, f), rnd(w, f), b, 2), f)

    def forward(self, x: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            x = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
            xs = [self.block(self.stem(x), "conv_0.conv_1")]
            for lv in range(1, 5):
                t = F.max_pool3d(xs[-1], 2)
                xs.append(self.block(self.block(t, f"down_{lv}.convs.conv_0"), f"down_{lv}.convs.conv_1"))
            u = xs[4]
            for lv in range(4, 0, -1):
                cat = torch.cat([xs[lv - 1], self.up(u, f"upcat_{lv}")], dim=1)
                u = self.block(self.block(cat, f"upcat_{lv}.convs.conv_0"), f"upcat_{lv}.convs.conv_1")
            w, b = self.sd["final_conv.weight"], self.sd["final_conv.bias"]
            return F.conv3d(u, w, b).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--cache", default="/tmp/bf16_budget_cache")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from delivr_cfos_amd.synth import synth_volume_np

    net = orc.build_unet(seed=0)
    orc.randomize_affine(net, seed=1)
    vol = synth_volume_np(CROP, seed=21)
    os.makedirs(a.cache, exist_ok=True)

    def cached(tag, fwd):
        c = LogitCache(fwd)
        path = os.path.join(a.cache, f"{tag}.npz")
        if os.path.isfile(path):
            z = np.load(path)
            for k in z.files:
                f, i = k.split("_")
                c.store[(None if f == "n" else int(f), int(i))] = z[k]
        return c, path

    def save(c, path):
        np.savez(path, **{f"{'n' if f is None else f}_{i}": v for (f, i), v in c.store.items()})

    t0 = time.time()
    ref_c, ref_p = cached("oracle32", lambda x: orc.unet_forward(net, x))
    ref = reference_arithmetic(orc, vol, ROI, ref_c, False)
    save(ref_c, ref_p)
    rep = {"crop": CROP, "roi": ROI, "weights": "seeded random (margin-free)", "variants": {}}
    for name in a.variants.split(","):
        emu = EmuMixed(net, VARIANTS[name])
        c, p = cached(f"emu_{name}", emu.forward)
        r = fp32_arithmetic(orc, vol, ROI, c, False)
        save(c, p)
        fr = flip_report(r["mask"], ref["mask"], ref["mean"])
        num = sum(float(((c.store[(None, i)] - ref_c.store[(None, i)]) ** 2).sum()) for i in range(27))
        den = sum(float((ref_c.store[(None, i)] ** 2).sum()) for i in range(27))
        fr["logit_rel_l2"] = float(np.sqrt(num / den))
        fr["fp16_blocks"] = sorted(VARIANTS[name])
        rep["variants"][name] = fr
        print(f"[{time.time() - t0:.0f} s] {name}: flipped {fr.get('flipped')} IoU {fr.get('iou'):.5f} logit rel l2 {fr['logit_rel_l2']:.2e}", flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
