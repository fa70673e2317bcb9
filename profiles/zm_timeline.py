"""In-kernel timeline of the z-march conv (diagnostic build DLV_ZM_VARIANT=30, s_memtime stamps per step phase).
Usage (GPU box): python profiles/zm_timeline.py [layer_index=1] [batch=16] [edge=128]   (1: 32->32, 16: 64->32 concat)
Slots: 0 step start, 1 loads issued, 2 MFMA loop issued, 3 epilogue done, 4 past barrier A, 5 plane written, 6 past barrier B."""
import os, sys, json
VAR = os.environ.setdefault("DLV_ZM_VARIANT", "30")   # 30: v1 stamped, 41: v4 stamped
os.environ["DLV_LANES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.weights import random_state_dict

li = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
E = int(sys.argv[3]) if len(sys.argv) > 3 else 128
eng = HipEngine(0)
eng.load_state_dict(random_state_dict(seed=0))
g = torch.Generator(device="cpu").manual_seed(1)
x1 = torch.randn((B, 32, E, E, E), generator=g).cuda()
x2 = torch.randn((B, 32, E, E, E), generator=g).cuda() if li in (16,) else None
tiles = (E // 8) * (E // 32)
nrec = tiles * 8 * (E + 4) * 8 * 2
st = torch.zeros(nrec, dtype=torch.int64, device="cuda")
eng.debug_layer_bf16(0, li, x1, x2, precision="fp16")          # warm
torch.cuda.synchronize()
eng._check(eng.lib.dlv_debug_stamps(eng.ctx, st.data_ptr()))
eng.debug_layer_bf16(0, li, x1, x2, precision="fp16")
torch.cuda.synchronize()
eng._check(eng.lib.dlv_debug_stamps(eng.ctx, None))
if VAR == "41":
    nsrc = 2 if li == 16 else 1
    s = st.cpu().numpy()[: tiles * 8 * (E + 4) * nsrc * 8].reshape(tiles, 8, (E + 4) * nsrc, 8)
    steps = slice(8 * nsrc, (E - 8) * nsrc)
    t = s[:, :, steps, :6].astype(np.float64)
    names = ["mfma_loop(issue)+fetch+pieces", "flush", "write_plane", "pack", "barrier"]
else:
    s = st.cpu().numpy()[: tiles * 8 * (E + 4) * 8].reshape(tiles, 8, E + 4, 8)
    steps = slice(8, E - 8)                  # steady state
    t = s[:, :, steps, :7].astype(np.float64)
    names = ["issue_loads", "mfma_loop(issue)", "epilogue", "barrierA", "write_plane", "barrierB"]
rt = s[:, :, steps, 7].astype(np.float64)
d = np.diff(t, axis=-1)                  # (tiles, waves, steps, 6)
step_len = t[:, :, 1:, 0] - t[:, :, :-1, 0]
clk = (t[:, :, -1, 0] - t[:, :, 0, 0]) / (rt[:, :, -1] - rt[:, :, 0]) * 100.0   # MHz
res = {"layer": li, "B": B, "edge": E, "step_cycles_mean": float(step_len.mean()), "step_cycles_p10": float(np.percentile(step_len, 10)),
       "step_cycles_p90": float(np.percentile(step_len, 90)), "clock_MHz_median": float(np.median(clk)),
       "phase_cycles_mean": {n: float(d[..., i].mean()) for i, n in enumerate(names)},
       "phase_cycles_by_wave": {n: [float(d[:, w, :, i].mean()) for w in range(8)] for i, n in enumerate(names)}}
# skew of the waves of one workgroup when they reach barrier A
arr = t[..., -2 if VAR == "41" else 3]
res["barrierA_arrival_skew_mean"] = float((arr.max(axis=1) - arr.min(axis=1)).mean())
print(json.dumps(res, indent=1))
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"zm_timeline_v{VAR}_l{li}.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
