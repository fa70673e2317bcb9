#!/usr/bin/env python3
"""dlv_tiff_stack_to_device on a C5-like stack (Z LZW planes of 2048 x 2048 on tmpfs): decoded GB/s by the number of host threads
(explicit n_threads; 0 = the library's default), two rounds interleaved.   python profiles/tools/tiff_threads_probe.py [Z]"""
import json
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from delivr_cfos_amd.downsample import downsample_and_mask as dm  # noqa: E402
from delivr_cfos_amd.engine import shared_engine  # noqa: E402
from delivr_cfos_amd.synth import synth_planes_torch  # noqa: E402
from delivr_cfos_amd.tiffio import write_tiff_plane  # noqa: E402

Z = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (Z, 2048, 2048)
d = tempfile.mkdtemp(prefix="dlv_tiffthr_", dir="/dev/shm")
out = {"planes": Z, "raw_GB": Z * 2048 * 2048 * 2 / 1e9, "runs": []}
try:
    eng = shared_engine(0)
    with ThreadPoolExecutor(32) as ex:
        for lo in range(0, Z, 64):
            blk = synth_planes_torch((1024, 2048, 2048), 2, eng.device, 256 + lo, 256 + min(lo + 64, Z)).cpu().numpy()
            list(ex.map(lambda i: write_tiff_plane(os.path.join(d, f"Z{lo + i:04d}.tif"), blk[i]), range(blk.shape[0])))
    planes = sorted(os.path.join(d, f) for f in os.listdir(d))
    out["file_GB"] = sum(os.path.getsize(p) for p in planes) / 1e9
    dst = torch.empty(shape, dtype=torch.uint16, device=eng.device)
    for rep in range(2):
        for n in (0, 16, 32, 64, 96, 128):
            t0 = time.perf_counter()
            dm.load_stack_to_device(eng, planes, out=dst, n_threads=n)
            eng.sync()
            dt = time.perf_counter() - t0
            out["runs"].append({"n_threads": n, "s": round(dt, 3), "GBps_decoded": round(out["raw_GB"] / dt, 2)})
finally:
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps(out))
