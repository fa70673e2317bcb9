#!/usr/bin/env python3
"""What committing device memory costs on this platform, where the cost falls (hipMalloc or first touch), and whether several
host threads share it or queue up.  torch's blocks map lazily (torch.empty of 16 GiB: 0.2 ms) and pay at the first touch; a plain
hipMalloc (what libdelivr_hip's workspaces use) pays when it is called."""
import ctypes as C
import json
import threading
import time

import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
out = {}
GiB = 2**30


def t_empty(gib):
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    t = torch.empty(int(gib * GiB), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    a = time.perf_counter() - t0
    t0 = time.perf_counter()
    t.zero_()
    torch.cuda.synchronize()
    b = time.perf_counter() - t0
    t0 = time.perf_counter()
    t.zero_()
    torch.cuda.synchronize()
    c = time.perf_counter() - t0
    return round(a, 4), round(b, 4), round(c, 4)


out["torch_16GiB_empty_firstzero_secondzero_s"] = t_empty(16)


def hmalloc(gib, box):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), int(gib * GiB))
    box.append((rc, p))


for nthr in (1, 2, 4):
    box = []
    ths = [threading.Thread(target=hmalloc, args=(16 / nthr, box)) for _ in range(nthr)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    out[f"hipMalloc_16GiB_in_{nthr}_threads_s"] = round(time.perf_counter() - t0, 4)
    # first touch of hipMalloc'ed memory
    if nthr == 1:
        rc, p = box[0]
        t0 = time.perf_counter()
        hip.hipMemset(p, 0, C.c_size_t(16 * GiB))
        hip.hipDeviceSynchronize()
        out["hipMalloc_16GiB_first_memset_s"] = round(time.perf_counter() - t0, 4)
        t0 = time.perf_counter()
        hip.hipMemset(p, 0, C.c_size_t(16 * GiB))
        hip.hipDeviceSynchronize()
        out["hipMalloc_16GiB_second_memset_s"] = round(time.perf_counter() - t0, 4)
    for rc, p in box:
        assert rc == 0
        hip.hipFree(p)


def zeros(gib, keep, stream):
    with torch.cuda.stream(stream):
        keep.append(torch.zeros(int(gib * GiB), dtype=torch.uint8, device="cuda"))
        stream.synchronize()


for nthr in (1, 2, 4):
    torch.cuda.empty_cache()
    keep = []
    ths = [threading.Thread(target=zeros, args=(16 / nthr, keep, torch.cuda.Stream())) for _ in range(nthr)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    torch.cuda.synchronize()
    out[f"torch_zeros_16GiB_in_{nthr}_threads_s"] = round(time.perf_counter() - t0, 4)
    del keep
print(json.dumps(out))
