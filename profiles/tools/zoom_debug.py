"""Where do the zoom kernels differ from scipy?  python profiles/tools/zoom_debug.py  (DLV_RESAMPLE_SIMPLE=1: old kernel)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from delivr_cfos_amd.engine import HipEngine
from oracle import delivr_oracle as orc

eng = HipEngine(0)
for in_shape, out_shape in (((12, 20, 24), (45, 290, 355)), ((30, 40, 50), (13, 17, 21)), ((12, 20, 24), (48, 300, 352))):
    rng = np.random.default_rng(in_shape[0] * 100 + out_shape[2])
    m = np.zeros(in_shape, dtype=np.uint8)
    m[in_shape[0] // 4:, in_shape[1] // 3:, in_shape[2] // 3:] = 1
    m[: in_shape[0] // 3, : in_shape[1] // 2, : in_shape[2] // 4] = 255
    noise = rng.random(in_shape) < 0.03
    m[noise] = rng.integers(0, 256, size=int(noise.sum())).astype(np.uint8)
    out = eng.zoom_spline2_u8(eng.to_device(m), out_shape).cpu().numpy()
    ref = orc.zoom_spline2_u8(m, out_shape)
    d = np.argwhere(out != ref)
    print(in_shape, out_shape, "mismatches", len(d), "of", out.size)
    for z, y, x in d[:12]:
        print("   ", (z, y, x), "got", out[z, y, x], "want", ref[z, y, x])
    if len(d):
        print("    x of mismatches mod 16:", np.bincount(d[:, 2] % 16, minlength=16).tolist())
        print("    z values:", np.unique(d[:, 0])[:20].tolist(), " y values:", np.unique(d[:, 1])[:20].tolist())
