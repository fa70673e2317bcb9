"""CCL-26 + statistics on a realistic cell mask (synthetic volume thresholded above the tissue level) - run under
rocprofv3 --kernel-trace --stats for the per-kernel split:  rocprofv3 ... -- python3 profiles/ccl_prof.py [Z Y X]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.synth import synth_volume_torch

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
eng = HipEngine(0)
vol = synth_volume_torch(shape, 2, eng.device)
cells = (vol.view(torch.int16) > 6500).to(torch.uint8)
del vol
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    labels, n = eng.ccl26(cells)
    eng.sync(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = eng.cc_stats(labels, n)
    t2 = time.perf_counter()
    print(f"rep {rep}: ccl26 {1e3*(t1-t0):.1f} ms, stats {1e3*(t2-t1):.1f} ms, N={n}, fg={int(st['voxel_counts'][1:].sum())}")
