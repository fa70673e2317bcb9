"""Gate A of the Winograd z-reg convolution: a CPU emulation of the 16-bit HIP forward with the
level-0/1 3x3x3 convolutions (Cin >= 32 -> 32: 91 % of the FLOPs; reference call site inference/inference.py:190-197,
inference/sliding_window_inferer.py:222) evaluated as Winograd F(2,3) - pushed through the parity chain of
tests/test_gpu_production_shapes.py::test_mask_vs_reference_accumulate_arithmetic on the same 256^3 crop, BEFORE any
kernel is written.  TEST INFRASTRUCTURE ONLY (see oracle/delivr_oracle.py's header); nothing in the product imports it.

Emulated arithmetic (what the HIP kernels do): every tensor that crosses HBM or enters an MFMA is
rounded to the 16-bit format, weights are rounded once, accumulation and InstanceNorm statistics are fp32.
    direct   the shipped kernels: 27-tap implicit GEMM, weights g rounded to fp16
    wino_x   F(2,3) along x only, direct in y and z:  U = G g (fp32) -> fp16,  V = B^T d in fp16 (one v_pk_add_f16 per
             element: a single rounding), M = sum U V in fp32 (MFMA), Y = A^T M in fp32.  18 instead of 27 MACs per voxel
    wino_xy  F(2x2,3x3) in the (y, x) plane, direct in z (the verdict's proposal): 12 MACs per voxel; V needs two rounded
             adds per element

usage:  python -m oracle.winograd_gate [--tta] [--modes direct,wino_x,wino_xy] [--fmt fp16|bf16] [--out profiles/...json]
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

ROI = (128, 128, 128)
CROP = (256, 256, 256)


def rnd(t: torch.Tensor, fmt: str) -> torch.Tensor:
    """Round to the 16-bit storage / operand format and come back to fp32."""
    return t.to(torch.float16 if fmt == "fp16" else torch.bfloat16).to(torch.float32)


def mish(x):
    return F.mish(x)


def inorm_stats(raw32: torch.Tensor):
    """Per-(n, c) mean and biased variance of the fp32 accumulators (the kernels sum the unrounded values)."""
    m = raw32.mean(dim=(2, 3, 4), keepdim=True)
    v = raw32.var(dim=(2, 3, 4), keepdim=True, unbiased=False)
    return m, v


def conv_direct(x16, w32, fmt):
    return F.conv3d(x16, rnd(w32, fmt), None, 1, 1)


def conv_wino_x(x16, w32, fmt):
    """F(2,3) along x (last axis).  x16: (B,C,D,H,W) already rounded; W even."""
    B, C, D, H, W = x16.shape
    g0, g1, g2 = w32[..., 0], w32[..., 1], w32[..., 2]
    U = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
    U = [rnd(u, fmt).unsqueeze(-1) for u in U]  # (Co,Ci,3,3,1)
    xp = F.pad(x16, (1, 1))
    d = [xp[..., j : j + W - 1 : 2] for j in range(4)]
    V = [rnd(d[0] - d[2], fmt), rnd(d[1] + d[2], fmt), rnd(d[2] - d[1], fmt), rnd(d[1] - d[3], fmt)]
    M = [F.conv3d(V[i].contiguous(), U[i], None, 1, (1, 1, 0)) for i in range(4)]
    y = torch.empty((B, w32.shape[0], D, H, W), dtype=torch.float32)
    y[..., 0::2] = M[0] + M[1] + M[2]
    y[..., 1::2] = M[1] - M[2] - M[3]
    return y


def conv_wino_xy(x16, w32, fmt):
    """F(2x2,3x3) in (y, x), direct in z.  V = B^T d B with a rounding after each of the two 1-D transforms (two packed
    adds per element), U = G g G^T in fp32 then rounded."""
    B, C, D, H, W = x16.shape

    def gt(a0, a1, a2):
        return [a0, (a0 + a1 + a2) * 0.5, (a0 - a1 + a2) * 0.5, a2]

    def bt(d0, d1, d2, d3, r):
        out = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
        return [rnd(o, fmt) for o in out] if r else out

    gx = gt(w32[..., 0], w32[..., 1], w32[..., 2])  # each (Co,Ci,3,3): [kz, ky]
    U = [[rnd(u, fmt) for u in gt(g[..., 0], g[..., 1], g[..., 2])] for g in gx]  # U[nu_x][nu_y]: (Co,Ci,3)
    xp = F.pad(x16, (1, 1, 1, 1))
    dx = bt(*[xp[..., j : j + W - 1 : 2] for j in range(4)], True)  # (B,C,D,H+2,W/2)
    y = torch.empty((B, w32.shape[0], D, H, W), dtype=torch.float32)
    Mx = []
    for i in range(4):
        dy = bt(*[dx[i][..., j : j + H - 1 : 2, :] for j in range(4)], True)  # (B,C,D,H/2,W/2)
        My = [F.conv3d(dy[k].contiguous(), U[i][k].unsqueeze(-1).unsqueeze(-1), None, 1, (1, 0, 0)) for k in range(4)]
        Mx.append((My[0] + My[1] + My[2], My[1] - My[2] - My[3]))  # output rows 2t, 2t+1
    for r in range(2):
        y[..., r::2, 0::2] = Mx[0][r] + Mx[1][r] + Mx[2][r]
        y[..., r::2, 1::2] = Mx[1][r] - Mx[2][r] - Mx[3][r]
    return y


CONVS = {"direct": conv_direct, "wino_x": conv_wino_x, "wino_xy": conv_wino_xy}


class Emu16:
    """The 16-bit forward of the HIP path, layer by layer (unet_bf16.hip / conv_zreg_kernel.h), on the CPU."""

    def __init__(self, net, mode: str, fmt: str = "fp16", wino_cin=(32, 64)):
        self.sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        self.mode, self.fmt, self.wino_cin = mode, fmt, wino_cin

    def block(self, x16, name, level):
        """Conv3d(k3) -> InstanceNorm -> Mish with the raw tensor stored in 16 bits between conv and normalisation.  The
        bias is dropped as in the kernels (InstanceNorm removes a per-channel constant exactly)."""
        w = self.sd[name + ".conv.weight"]
        gamma, beta = self.sd[name + ".adn.N.weight"], self.sd[name + ".adn.N.bias"]
        cout, cin = w.shape[0], w.shape[1]
        zreg = level <= 1 and cout == 32 and cin in self.wino_cin  # the seven register-resident-weights convs
        raw32 = (CONVS[self.mode] if zreg else conv_direct)(x16, w, self.fmt)
        m, v = inorm_stats(raw32)
        raw16 = rnd(raw32, self.fmt)
        sc = gamma.view(1, -1, 1, 1, 1) / torch.sqrt(v + 1e-5)
        sh = beta.view(1, -1, 1, 1, 1) - m * sc
        return rnd(mish(raw16 * sc + sh), self.fmt)

    def two(self, x16, name, level):
        return self.block(self.block(x16, name + ".conv_0", level), name + ".conv_1", level)

    def stem(self, x):
        """Conv3d(1->32) on exact uint16 intensities (hi/lo byte split: exact operands), fp32 statistics, activated output
        stored in 16 bits - no raw tensor."""
        w = self.sd["conv_0.conv_0.conv.weight"]
        raw32 = F.conv3d(x, rnd(w, self.fmt), None, 1, 1)
        m, v = inorm_stats(raw32)
        sc = self.sd["conv_0.conv_0.adn.N.weight"].view(1, -1, 1, 1, 1) / torch.sqrt(v + 1e-5)
        sh = self.sd["conv_0.conv_0.adn.N.bias"].view(1, -1, 1, 1, 1) - m * sc
        return rnd(mish(raw32 * sc + sh), self.fmt)

    def up(self, x16, name):
        w, b = self.sd[name + ".upsample.deconv.weight"], self.sd[name + ".upsample.deconv.bias"]
        return rnd(F.conv_transpose3d(x16, rnd(w, self.fmt), b, 2), self.fmt)

    def forward(self, x: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            x = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
            x0 = self.block(self.stem(x), "conv_0.conv_1", 0)
            xs = [x0]
            for lv in range(1, 5):
                xs.append(self.two(F.max_pool3d(xs[-1], 2), f"down_{lv}.convs", lv))
            u = xs[4]
            for lv in range(4, 0, -1):
                cat = torch.cat([xs[lv - 1], self.up(u, f"upcat_{lv}")], dim=1)
                u = self.two(cat, f"upcat_{lv}.convs", lv - 1)
            w, b = self.sd["final_conv.weight"], self.sd["final_conv.bias"]
            return F.conv3d(u, w, b).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tta", action="store_true")
    ap.add_argument("--modes", default="direct,wino_x,wino_xy")
    ap.add_argument("--fmt", default="fp16")
    ap.add_argument("--weights", default="random", choices=["random", "trained"])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--cache", default="/tmp/wino_gate_cache")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from delivr_cfos_amd.synth import synth_volume_np

    if a.weights == "random":
        net = orc.build_unet(seed=0)
        orc.randomize_affine(net, seed=1)
    else:
        from delivr_cfos_amd.weights import trained_like_state_dict

        net = orc.build_unet(seed=0)
        net.load_state_dict(trained_like_state_dict(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trained_like_weights.npz")))
    vol = synth_volume_np(CROP, seed=21)
    os.makedirs(a.cache, exist_ok=True)

    def cached(tag, fwd):
        """LogitCache whose store persists in a.cache (the oracle forwards are the slow part)."""
        c = LogitCache(fwd)
        path = os.path.join(a.cache, f"{tag}_{a.weights}.npz")
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
    ref = reference_arithmetic(orc, vol, ROI, ref_c, a.tta)
    f32 = fp32_arithmetic(orc, vol, ROI, ref_c, a.tta)
    save(ref_c, ref_p)
    rep = {"crop": CROP, "roi": ROI, "tta": a.tta, "fmt": a.fmt, "weights": a.weights,
           "oracle_fp32_accumulate": flip_report(f32["mask"], ref["mask"], ref["mean"])}
    print(f"[{time.time() - t0:.0f} s] oracle fp32-accumulate vs reference arithmetic: {json.dumps(rep['oracle_fp32_accumulate'])}", flush=True)
    for mode in a.modes.split(","):
        emu = Emu16(net, mode, a.fmt)
        c, p = cached(f"emu_{a.fmt}_{mode}", emu.forward)
        r = fp32_arithmetic(orc, vol, ROI, c, a.tta)
        save(c, p)
        fr = flip_report(r["mask"], ref["mask"], ref["mean"])
        # logit error of the emulated forward against the oracle's, over the 27 plain windows
        num = sum(float(((c.store[(None, i)] - ref_c.store[(None, i)]) ** 2).sum()) for i in range(27))
        den = sum(float((ref_c.store[(None, i)] ** 2).sum()) for i in range(27))
        fr["logit_rel_l2"] = float(np.sqrt(num / den))
        rep[mode] = fr
        print(f"[{time.time() - t0:.0f} s] emulated {a.fmt} [{mode}] vs reference arithmetic: {json.dumps(fr)}", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
