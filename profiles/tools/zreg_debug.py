"""debug: one conv block (layer li) through the default z-march build and through variant V; where do they differ?
usage: python profiles/tools/zreg_debug.py [li=1] [D,H,W=40,24,64] [variant=50] [B=1]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.weights import random_state_dict

li = int(sys.argv[1]) if len(sys.argv) > 1 else 1
D, H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "40,24,64").split(","))
var = int(sys.argv[3]) if len(sys.argv) > 3 else 50
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
eng = HipEngine(0)
eng.load_state_dict({"state_dict": random_state_dict(0)})
c1 = 32
c2 = 32 if li in (14, 16) else 0
g = torch.Generator().manual_seed(5)
x1 = torch.randn((B, c1, D, H, W), generator=g).half().float().cuda()
x2 = torch.randn((B, c2, D, H, W), generator=g).half().float().cuda() if c2 else None
import torch.nn.functional as F
from delivr_cfos_amd.engine import CONV_BLOCKS
sd = {k.replace("module.", ""): v for k, v in random_state_dict(0).items()}
wt = sd[CONV_BLOCKS[li] + ".conv.weight"].half().float()
xin = x1 if x2 is None else torch.cat([x1, x2], 1)
ref = F.conv3d(xin.cpu(), wt, None, padding=1).numpy()
eng.set_zm_variant(var)
out = eng.debug_layer_bf16(2, li, x1, x2, precision="fp16").cpu().numpy()
eng.set_zm_variant(0)
print("variant", var, "RAW conv: nan", int(np.isnan(out).sum()), "inf", int(np.isinf(out).sum()), "of", out.size)
d = np.abs(np.nan_to_num(out, nan=1e3, posinf=1e4, neginf=1e4) - ref)
print("max diff", d.max(), "mean", d.mean(), "ref std", ref.std())
np.set_printoptions(linewidth=200)
print("per z:", np.round(d.max(axis=(0, 1, 3, 4)), 2))
print("per y:", np.round(d.max(axis=(0, 1, 2, 4)), 2))
print("per x:", np.round(d.max(axis=(0, 1, 2, 3)), 2))
print("per c:", np.round(d.max(axis=(0, 2, 3, 4)), 2))
base = eng.debug_layer_bf16(0, li, x1, x2, precision="fp16").cpu().numpy()
eng.set_zm_variant(var)
outn = eng.debug_layer_bf16(0, li, x1, x2, precision="fp16").cpu().numpy()
eng.set_zm_variant(0)
dn = np.abs(outn - base)
print("NORMED vs variant 0: max", dn.max(), "mean", dn.mean(), " per sample max:", np.round(dn.reshape(B, -1).max(1), 3))
print("RAW per sample max:", np.round(d.reshape(B, -1).max(1), 3))
bias = sd[CONV_BLOCKS[li] + ".conv.bias"]
gam, bet = sd[CONV_BLOCKS[li] + ".adn.N.weight"], sd[CONV_BLOCKS[li] + ".adn.N.bias"]
rawt = torch.from_numpy(ref) + bias.view(1, -1, 1, 1, 1)
reft = F.mish(F.instance_norm(rawt, weight=gam, bias=bet, eps=1e-5)).numpy()
for nm, arr in (("variant 0", base), (f"variant {var}", outn)):
    e = np.abs(arr - reft)
    print(f"NORMED {nm} vs torch: max {e.max():.4f} mean {e.mean():.5f}")
# statistics of the raw outputs per channel: mean / var from the kernel's raw tensor vs torch
m_k, v_k = out.mean(axis=(2, 3, 4)), out.var(axis=(2, 3, 4))
m_t, v_t = ref.mean(axis=(2, 3, 4)), ref.var(axis=(2, 3, 4))
print("raw-tensor stats: max |dmean|", np.abs(m_k - m_t).max(), "max |dvar|/var", (np.abs(v_k - v_t) / v_t).max())
for v in (0, var):
    eng.set_zm_variant(v)
    raw = eng.debug_layer_bf16(2, li, x1, x2, precision="fp16").cpu().numpy().astype(np.float64)
    ssb = eng.debug_layer_bf16(3, li, x1, x2, precision="fp16").cpu().numpy().reshape(-1)[: B * 32 * 2].reshape(B, 32, 2)
    eng.set_zm_variant(0)
    mean, varr = raw.mean(axis=(2, 3, 4)), raw.var(axis=(2, 3, 4))
    sc = gam.numpy()[None] / np.sqrt(varr + 1e-5)
    shf = bet.numpy()[None] - mean * sc
    print(f"variant {v}: scale rel err max {np.abs(ssb[..., 0] / sc - 1).max():.2e}  shift abs err max {np.abs(ssb[..., 1] - shf).max():.2e}")
    if v == var:
        print("  scale ratio per channel (sample 0):", np.round(ssb[0, :, 0] / sc[0], 4))
# which rows are missing from the kernel's statistics?  implied variance of variant `var` vs candidate subsets
eng.set_zm_variant(var)
raw = eng.debug_layer_bf16(2, li, x1, x2, precision="fp16").cpu().numpy().astype(np.float64)
ssb = eng.debug_layer_bf16(3, li, x1, x2, precision="fp16").cpu().numpy().reshape(-1)[: B * 32 * 2].reshape(B, 32, 2)
eng.set_zm_variant(0)
var_k = (gam.numpy()[None] / ssb[..., 0]) ** 2 - 1e-5
N = D * H * W
def cand(mask):
    r = raw * mask[None, None, :, :, None]
    s1, s2 = r.sum(axis=(2, 3, 4)) / N, (r ** 2).sum(axis=(2, 3, 4)) / N
    return s2 - s1 ** 2
full = np.ones((D, H))
c = {"all": full.copy()}
m = full.copy(); m[[15, 31, D - 1], :] = 0; c["no planes 15,31,last"] = m
for tyt in (8, 16):
    rw = tyt // 2
    m = full.copy()
    for pz in (15, 31, D - 1):
        if pz < D:
            m[pz, rw - 1::rw] = 0
    c[f"no pending rows (TYT {tyt}) at planes 15,31,last"] = m
    m = full.copy(); m[:, rw - 1::rw] = 0; c[f"no pending rows at all (TYT {tyt})"] = m
m = full.copy(); m[D - 1, :] = 0; c["no last plane"] = m
m = full.copy(); m[0, :] = 0; c["no first plane"] = m
for k, mk in c.items():
    print(f"  implied var / candidate var [{k}]: mean {np.mean(var_k / cand(mk)):.5f}  spread {np.std(var_k / cand(mk)):.5f}")
