#!/usr/bin/env python3
"""Where the wall clock of the two steps goes in a FRESH process (what `python -m delivr_cfos_amd` is): context creation, weight
load, first pass against second pass (workspace allocation, code-object load, clocks), first labelling against second, and
the tmpfs write strategies of hostio.download.  Prints one JSON object.  Usage: python profiles/tools/step_probe.py [c3|c2]"""
import json
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def tmpfs_write_probe(nbytes, out):
    """host array -> file on /dev/shm: pwrite from T threads vs memcpy into a shared mapping from T threads"""
    src = np.random.default_rng(0).integers(0, 255, 64 << 20, dtype=np.uint8)
    path = "/dev/shm/dlv_wprobe.bin"
    res = {}
    for mode in ("pwrite", "mmap"):
        for T in (8, 16, 32, 64, 128):
            if os.path.exists(path):
                os.remove(path)
            fd = os.open(path, os.O_RDWR | os.O_CREAT)
            os.ftruncate(fd, nbytes)
            pool = ThreadPoolExecutor(T)
            chunk = 64 << 20
            per = chunk // T
            t0 = time.perf_counter()
            if mode == "pwrite":
                mv = memoryview(src)
                for lo in range(0, nbytes, chunk):
                    fs = [pool.submit(os.pwrite, fd, mv[i * per:(i + 1) * per], lo + i * per) for i in range(T)]
                    [f.result() for f in fs]
            else:
                mm = mmap.mmap(fd, nbytes)
                dst = np.frombuffer(mm, dtype=np.uint8)
                for lo in range(0, nbytes, chunk):
                    fs = [pool.submit(np.copyto, dst[lo + i * per:lo + (i + 1) * per], src[i * per:(i + 1) * per]) for i in range(T)]
                    [f.result() for f in fs]
                del dst
                mm.close()
            res[f"{mode}_{T}"] = round(nbytes / (time.perf_counter() - t0) / 1e9, 2)
            pool.shutdown()
            os.close(fd)
            os.remove(path)
    out["tmpfs_write_GBps"] = res


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    out = {"workload": wl}
    tmpfs_write_probe(8 << 30, out)
    t0 = time.perf_counter()
    import torch

    out["import_torch_s"] = time.perf_counter() - t0
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_torch
    from delivr_cfos_amd.weights import trained_like_state_dict

    shape = {"c3": (1024, 2048, 2048), "c2": (512, 512, 512)}[wl]
    roi = (128, 128, 128)
    t0 = time.perf_counter()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    out["torch_cuda_init_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng = HipEngine(0)
    out["ctx_create_s"] = time.perf_counter() - t0
    sd = trained_like_state_dict(os.path.join(ROOT, "tests", "golden", "trained_like_weights.npz"))
    t0 = time.perf_counter()
    eng.load_state_dict({"state_dict": sd})
    eng.sync()
    out["load_state_dict_s"] = time.perf_counter() - t0
    vol = synth_volume_torch(shape, 2, eng.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    out["alloc_zero_acc_s"] = time.perf_counter() - t0
    p = eng.make_sw_params(shape, roi, 0.5, None, 0, "fp16")
    passes = []
    for _ in range(3):
        acc.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sw_infer(p, vol, acc)
        eng.sync()
        passes.append(time.perf_counter() - t0)
    out["pass_s"] = passes
    fin = []
    for _ in range(2):
        t0 = time.perf_counter()
        mask = eng.finalize(acc, None, vol, shape, 0.5, 30, 238)
        eng.sync()
        fin.append(time.perf_counter() - t0)
    out["finalize_s"] = fin
    del acc
    ccl = []
    for _ in range(3):
        t0 = time.perf_counter()
        lab = torch.empty(shape, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        del lab
        labels, n = eng.ccl26(mask)
        eng.sync()
        ccl.append({"torch_empty_s": t1 - t0, "ccl26_s": time.perf_counter() - t1, "n": n})
        del labels
    out["ccl"] = ccl
    # a second context in the same process (what count_blobs creates after run_inference in one CLI run)
    t0 = time.perf_counter()
    eng2 = HipEngine(0)
    out["ctx2_create_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    labels, n = eng2.ccl26(mask)
    eng2.sync()
    out["ctx2_first_ccl26_s"] = time.perf_counter() - t0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
