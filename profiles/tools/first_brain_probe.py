#!/usr/bin/env python3
"""The first brain of a FRESH process, step by step (what `python -m delivr_cfos_amd` pays once): run_inference + count_blobs on a
C3-size volume written to tmpfs, twice, in a process that has not touched the GPU before - with the breakdown of each call."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import torch  # noqa: E402

out = {"import_torch_s": round(time.perf_counter() - t0, 3)}
from delivr_cfos_amd import hostio  # noqa: E402
from delivr_cfos_amd.count_blobs import count_blobs  # noqa: E402
from delivr_cfos_amd.engine import HipEngine  # noqa: E402
from delivr_cfos_amd.inference.inference import run_inference  # noqa: E402
from delivr_cfos_amd.synth import synth_volume_torch  # noqa: E402
from delivr_cfos_amd.weights import trained_like_state_dict  # noqa: E402

shape = (1024, 2048, 2048)
d = tempfile.mkdtemp(prefix="dlv_first_", dir="/dev/shm")
try:
    # the input file, made by a helper engine that is closed (and its memory released) before the steps run
    gen = HipEngine(0)
    vol = synth_volume_torch(shape, 2, gen.device)
    nifti = os.path.join(d, "masked_nifti.npy")
    hostio.save_npy(gen, vol.reshape((1, 1) + shape), nifti, np.uint16)
    del vol
    gen.close()
    del gen
    torch.cuda.empty_cache()
    time.sleep(3.0)
    sd = trained_like_state_dict(os.path.join(ROOT, "tests", "golden", "trained_like_weights.npz"))
    settings = {"postprocessing": {"output_location": os.path.join(d, "post") + "/"}}
    for which in ("first", "second"):
        shutil.rmtree(os.path.join(d, "blob"), ignore_errors=True)
        shutil.rmtree(os.path.join(d, "post"), ignore_errors=True)
        t0 = time.perf_counter()
        run_inference([nifti], os.path.join(d, "blob"), (1, 1) + shape, comment="brain", crop_size=(128, 128, 128), state_dict={"state_dict": sd})
        s2 = time.perf_counter() - t0
        t2 = dict(run_inference.last_timings)
        t0 = time.perf_counter()
        n = count_blobs(settings, os.path.join(d, "blob"), 0, "brain", (1, 1) + shape)
        s3 = time.perf_counter() - t0
        out[which] = {"step2_s": round(s2, 3), "step3_s": round(s3, 3), "step2": {k: round(v, 3) for k, v in t2.items()},
                      "step3": {k: round(v, 3) for k, v in count_blobs.last_timings.items()}, "components": int(n)}
finally:
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps(out))
