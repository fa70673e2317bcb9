#!/usr/bin/env python3
"""Does binding hostio's reader / writer threads to the GPU's NUMA node steady the one-file write rate?  8 GB from HBM into a
fresh file on /dev/shm and back, four times per setting, settings interleaved (DLV_IO_NUMA=off | auto | 0 | 1)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from delivr_cfos_amd import hostio  # noqa: E402
from delivr_cfos_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0)
out = {"nodes": {}}
for n in sorted(os.listdir("/sys/devices/system/node")):
    if n.startswith("node") and n[4:].isdigit():
        out["nodes"][n] = open(f"/sys/devices/system/node/{n}/cpulist").read().strip()
os.environ["DLV_IO_NUMA"] = "auto"
cp = hostio.io_cpus()
out["gpu_node_cpus"] = None if cp is None else f"{min(cp)}-{max(cp)} ({len(cp)} cpus)"
t = torch.randint(1, 2**31 - 1, (2048, 1024, 1024), dtype=torch.int32, device="cuda")  # 8 GiB, no zero block
path = "/dev/shm/dlv_io_numa_probe.npy"
res = {"off": [], "auto": [], "0": [], "1": []}
for rep in range(4):
    for mode in ("off", "auto", "0", "1"):
        os.environ["DLV_IO_NUMA"] = mode
        hostio._pool = None  # the next transfer builds its pool under this setting
        if os.path.exists(path):
            os.remove(path)
        hostio.save_npy(eng, t, path, np.uint32, what="w")
        w = hostio.last_transfer["w"]["GBps"]
        mm = np.load(path, mmap_mode="r")
        back = hostio.upload(eng, mm, what="r")
        r = hostio.last_transfer["r"]["GBps"]
        ok = bool(torch.equal(back.view(torch.int32), t)) if rep == 0 else None
        del back, mm
        res[mode].append({"write_GBps": round(w, 2), "read_GBps": round(r, 2), "ok": ok})
os.remove(path)
out["runs"] = res
print(json.dumps(out))
