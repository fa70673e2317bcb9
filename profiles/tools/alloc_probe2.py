#!/usr/bin/env python3
"""When does a large device allocation become slow?  torch.zeros(16 GiB) and a plain hipMalloc(16 GiB) + memset, timed (a) in a
fresh process, (b) with 8 GiB held, (c) after a libdelivr_hip context exists, (d) after the synthetic-volume generator ran
(many temporaries through torch's caching allocator), (e) after torch's cache was emptied (blocks handed back to the driver)."""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.cuda.init()
torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
GiB = 2**30
out = {}


def measure(tag):
    t0 = time.perf_counter()
    z = torch.zeros(16 * GiB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    a = time.perf_counter() - t0
    p = C.c_void_p()
    t0 = time.perf_counter()
    assert hip.hipMalloc(C.byref(p), 16 * GiB) == 0
    b = time.perf_counter() - t0
    t0 = time.perf_counter()
    hip.hipMemset(p, 0, 16 * GiB)
    hip.hipDeviceSynchronize()
    c = time.perf_counter() - t0
    t0 = time.perf_counter()
    hip.hipFree(p)
    d = time.perf_counter() - t0
    free, total = torch.cuda.mem_get_info()
    out[tag] = {"torch_zeros_16GiB_s": round(a, 4), "hipMalloc_16GiB_s": round(b, 4), "first_memset_s": round(c, 4), "hipFree_s": round(d, 4),
                "free_GiB": round(free / GiB, 1)}
    return z


z = measure("a_fresh")
del z
hold = torch.empty(8 * GiB, dtype=torch.uint8, device="cuda")
z = measure("b_8GiB_held_and_16GiB_cached")
del z
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402

eng = HipEngine(0)
z = measure("c_after_context")
del z
vol = synth_volume_torch((512, 2048, 2048), 2, eng.device)
torch.cuda.synchronize()
z = measure("d_after_synth_volume")
del z
torch.cuda.empty_cache()
z = measure("e_after_empty_cache")
del z
torch.cuda.empty_cache()
z = measure("f_again")
print(json.dumps(out))
