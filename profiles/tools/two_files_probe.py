#!/usr/bin/env python3
"""One file takes 2.5-7 GB/s from the kernel (inode lock).  Do TWO files written at once take twice that?  8 GB each from HBM into
fresh tmpfs files: one after the other, then both at once (two engines = two staging rings, two threads)."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from delivr_cfos_amd import hostio  # noqa: E402
from delivr_cfos_amd.engine import HipEngine  # noqa: E402

e1, e2 = HipEngine(0), HipEngine(0)
t = torch.randint(1, 2**31 - 1, (2048, 1024, 1024), dtype=torch.int32, device="cuda")  # 8 GiB, no zero block
paths = ["/dev/shm/dlv_two_a.npy", "/dev/shm/dlv_two_b.npy"]
out = {"one_after_the_other_GBps": [], "both_at_once_GBps": []}
for rep in range(3):
    for p in paths:
        if os.path.exists(p):
            os.remove(p)
    t0 = time.perf_counter()
    hostio.save_npy(e1, t, paths[0], np.uint32, what="a")
    hostio.save_npy(e1, t, paths[1], np.uint32, what="b")
    out["one_after_the_other_GBps"].append(round(2 * t.numel() * 4 / (time.perf_counter() - t0) / 1e9, 2))
    for p in paths:
        os.remove(p)
    ths = [threading.Thread(target=hostio.save_npy, args=(e, t, p, np.uint32, w)) for e, p, w in ((e1, paths[0], "a"), (e2, paths[1], "b"))]
    t0 = time.perf_counter()
    [x.start() for x in ths]
    [x.join() for x in ths]
    out["both_at_once_GBps"].append(round(2 * t.numel() * 4 / (time.perf_counter() - t0) / 1e9, 2))
for p in paths:
    os.remove(p)
print(json.dumps(out))
