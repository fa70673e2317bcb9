#!/usr/bin/env python3
"""Step 1 of the pipeline, file -> file, at C5 size: 1024 LZW TIFF planes of 2048 x 2048 (the raw stack, on tmpfs) ->
downsampled_stack.npy + masked_niftis/masked_nifti.npy (`downsample_mask`, reference downsample/downsample_and_mask.py:139-427),
with the simple threshold and with an ilastik mask on the down-sampled grid (synthetic planes where the reference reads
ilastik's output), twice each in one process (first / next brain), with the wall clock of the pieces.
    python profiles/tools/step1_probe.py [Z]            (Z: planes, default 1024)"""
import json
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from delivr_cfos_amd import hostio  # noqa: E402
from delivr_cfos_amd.downsample import downsample_and_mask as dm  # noqa: E402
from delivr_cfos_amd.engine import shared_engine  # noqa: E402
from delivr_cfos_amd.hostlogic import downsample_ratios  # noqa: E402
from delivr_cfos_amd.synth import synth_planes_torch  # noqa: E402
from delivr_cfos_amd.tiffio import write_tiff_plane  # noqa: E402

Z = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (Z, 2048, 2048)
steps = {"original_um_x": 1.62, "original_um_y": 1.62, "original_um_z": 6.0, "downsample_um_x": 25.0, "downsample_um_y": 25.0, "downsample_um_z": 25.0}
out = {"shape": list(shape), "ratios": list(downsample_ratios(steps))}
d = tempfile.mkdtemp(prefix="dlv_step1_", dir=os.environ.get("DLV_BENCH_TMP") or "/dev/shm")
try:
    eng = shared_engine(0)
    raw_dir = os.path.join(d, "raw", "brain")
    os.makedirs(raw_dir)
    # the raw stack as the microscope leaves it: one LZW TIFF per z-plane (written by the native writer, 32 threads)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(32) as ex:
        for lo in range(0, Z, 64):
            blk = synth_planes_torch(shape, 2, eng.device, lo, min(lo + 64, Z)).cpu().numpy()
            list(ex.map(lambda i: write_tiff_plane(os.path.join(raw_dir, f"Z{lo + i:04d}.tif"), blk[i]), range(blk.shape[0])))
    out["write_raw_tiffs_s"] = round(time.perf_counter() - t0, 2)
    out["raw_tiff_GB"] = round(sum(os.path.getsize(os.path.join(raw_dir, f)) for f in os.listdir(raw_dir)) / 1e9, 2)
    settings = {"raw_location": os.path.join(d, "raw"),
                "mask_detection": {"output_location": os.path.join(d, "01_mask"), "downsample_steps": steps, "mask_with_Ilastik": False,
                                   "simple_threshold_value": 250},
                "blob_detection": {"window_dimensions": {"window_dim_0": 96, "window_dim_1": 96, "window_dim_2": 64}}}
    torch.cuda.empty_cache()

    def pieces(use_ilastik):
        """the same calls downsample_mask makes, timed one by one"""
        t = {}
        planes = sorted(os.path.join(raw_dir, f) for f in os.listdir(raw_dir))
        t0 = time.perf_counter()
        raw = dm.load_stack_to_device(eng, planes)
        eng.sync()
        t["tiff_decode_to_hbm_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        ds = dm.downsample_volume(eng, raw, downsample_ratios(steps))
        ds_host = ds.cpu().numpy()
        t["block_mean_and_d2h_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if use_ilastik:
            m = dm.load_ilastik_mask(os.path.join(d, "01_mask", "brain"))
            mask_us = dm.upsample_mask(eng, eng.to_device(m), tuple(int(v) for v in raw.shape))
            padded = dm.mask_and_pad(eng, raw, mask_us, (96, 96, 64))
        else:
            padded = dm.mask_and_pad(eng, raw, None, (96, 96, 64), 250)
        eng.sync()
        t["mask_and_pad_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        p = os.path.join(d, "scratch.npy")
        dm.write_masked_nifti_npy(p, padded, engine=eng)
        t["write_masked_nifti_s"] = time.perf_counter() - t0
        t["write"] = dict(hostio.last_transfer.get("d2h_volume", {})) if hasattr(hostio, "last_transfer") else {}
        os.remove(p)
        return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()}

    for tag, ilastik in (("threshold", False), ("ilastik", True)):
        settings["mask_detection"]["mask_with_Ilastik"] = ilastik
        res_dir = os.path.join(d, "01_mask", "brain")
        if ilastik:
            # ilastik's probability planes on the down-sampled grid (0..255; >= 125 = ventricle / outside): here the complement of a
            # smooth ellipsoid, so that the mask keeps the brain
            ds = np.load(os.path.join(res_dir, "downsampled_stack.npy"))
            zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, n) for n in ds.shape], indexing="ij")
            inside = (zz ** 2 + yy ** 2 + xx ** 2) < 0.9
            os.makedirs(os.path.join(res_dir, "ventricles_zplanes"), exist_ok=True)
            for i in range(ds.shape[0]):
                write_tiff_plane(os.path.join(res_dir, "ventricles_zplanes", f"p{i:04d}.tif"), np.where(inside[i], 255, 0).astype(np.uint8))
            out["ilastik_grid"] = list(ds.shape)
        walls = []
        for rep in range(2):
            shutil.rmtree(os.path.join(res_dir, "masked_niftis"), ignore_errors=True)
            t0 = time.perf_counter()
            dm.downsample_mask(settings, "brain", engine=eng)
            walls.append(round(time.perf_counter() - t0, 3))
        nifti = os.path.join(res_dir, "masked_niftis", "masked_nifti.npy")
        hdr = np.load(nifti, mmap_mode="r")
        out[tag] = {"step1_wall_s": walls, "masked_nifti_shape": list(hdr.shape), "masked_nifti_GB_on_disk": round(os.stat(nifti).st_blocks * 512 / 1e9, 2),
                    "nonzero_fraction_plane_512": float((np.asarray(hdr[0, 0, min(512, Z - 1)]) != 0).mean()), "pieces": pieces(ilastik)}
        del hdr
finally:
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps(out))
