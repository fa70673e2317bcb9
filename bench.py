#!/usr/bin/env python3
"""bench.py - voxels/s of sliding-window 3D-U-Net inference on synthetic light-sheet volumes.

    python bench.py --gpus 1 --steps 2 --warmup 1            # one MI355X
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W   # N ranks, RCCL over xGMI
    python bench.py --gpus N ...                              # the same: spawns that launch line as a child process

One "step" = ONE sliding-window pass (no TTA) over the whole volume, volume resident in HBM as
uint16: tiler -> per-window background skip -> U-Net forward (16-bit MFMA operands, fp32 accumulate) -> fp32 overlap blend ->
(N>1: seam exchange) -> threshold + L1-30 eroded re-mask -> uint8 mask (N>1: it stays on the ranks as Z-slabs, the planes
each rank owns; only the voxel count and checksum travel to rank 0).
That is BASELINE.json's metric ("voxels/sec sliding-window 3D U-Net inference, 2048x2048x1024 vol @
1/2/4/8 GPU"); the volume and its window list are the same at every N, so scaling is "strong".

Rank 0 prints ONE JSON line.  `value` comes from EXACTLY --steps passes with no per-kernel instrumentation; the same
passes are then repeated with every launch bracketed by HIP events on the engine's stream (dlv_prof_*) for `kernels`
and `roofline`, once more with the background skip disabled for `value_dense`, and `cpu_baseline` times the CPU oracle
(oracle/, a port of the reference's algorithm) on BASELINE.md section 3's bounded samples on the host cores of this
box and checks that the HIP path gives the same result on that sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "voxels/sec sliding-window 3D U-Net inference, 2048x2048x1024 vol @1/2/4/8 GPU"
WORKLOADS = {
    # name: (stack (Z, Y, X), window (roi), seed, TTA).  The volume in HBM is the stack zero-padded to multiples of the window
    # (what step 1 writes into masked_nifti.npy, downsample/downsample_and_mask.py:390-417); `value` counts the STACK's voxels.
    "c3": ((1024, 2048, 2048), (128, 128, 128), 2, False),  # BASELINE configs[2]: 2048x2048x1024 synthetic brain
    "c2": ((512, 512, 512), (128, 128, 128), 1, False),     # BASELINE configs[1]: 512^3 volume
    "tiny": ((128, 256, 256), (64, 64, 64), 5, False),      # plumbing check
    "tiny_default": ((100, 180, 150), (96, 96, 64), 5, True),  # plumbing check of `default`: padding (192, 192, 192) + the TTA schedule
    # what the reference SHIPS (config.json:24-28,63): windows 96 x 96 x 64 and test-time augmentation - the 13 passes of
    # inference/inference.py:261-279 = 3 distinct passes weighted 5:4:4; one step = all of them
    "default": ((1024, 2048, 2048), (96, 96, 64), 2, True),
    # run_inference's own defaults (inference/inference.py:113-129): crop_size (64, 64, 32), tta False
    "legacy": ((1024, 2048, 2048), (64, 64, 32), 2, False),
}
# the benchmark's checkpoint is test data of this repository (tests/golden/; the product package does not look there)
TRAINED_LIKE_FIXTURE = os.environ.get("DLV_TRAINED_LIKE_FIXTURE") or os.path.join(ROOT, "tests", "golden", "trained_like_weights.npz")
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
FLOP_PER_PATCH_VOXEL = 285104.0


def _cpu_info():
    """model name, sockets, physical cores, logical CPUs of this host (from /proc/cpuinfo)"""
    model, phys, cores = None, set(), set()
    try:
        pid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model is None:
                    model = v
                elif k == "physical id":
                    pid = v
                    phys.add(v)
                elif k == "core id":
                    cores.add((pid, v))
    except OSError:
        pass
    # the container's CPU bandwidth limit (cgroup v2 cpu.max "quota period", v1 cfs_quota_us / cfs_period_us): a box that shows 256
    # logical CPUs may grant this process 16 cores' worth of time - threads beyond that only get throttled (the thread probe of
    # the CPU baseline and the TIFF decode pool both stop scaling there)
    quota = None
    for qf, pf in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            with open(qf) as f:
                txt = f.read().split()
            if pf is None:
                q, per = txt[0], txt[1]
            else:
                q = txt[0]
                with open(pf) as f:
                    per = f.read().split()[0]
            if q not in ("max", "-1"):
                quota = round(int(q) / int(per), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"model": model, "sockets": len(phys) or None, "physical_cores": len(cores) or None, "logical_cpus": os.cpu_count(),
            "cgroup_cpu_quota_cores": quota}


def cpu_baseline(eng, sd, vol, shape, roi, n_active_full, vox_full, threads, crop_edge, precision, weights_name, acc_full, zblock):
    """BASELINE.md section 3's CPU legs on the host cores of this box, with the oracle (the CPU port of the reference's
    algorithm; tests and this leg are its only callers):
      (i)  one 64^3 patch forward, fp32 (BASELINE config 1): 1 warm-up, median of 3;
      (ii) a `crop_edge`^3 crop of the benchmark volume (centre: tissue) with the benchmark's windows / 50 % overlap
           through tiler -> U-Net -> blend (every window timed: the MEDIAN window time is extrapolated) -> finalize
           (threshold + 30x erosion, median of 3) -> 26-connected labels;
      (iii) finalize + erosion once more (median of 3) on a crop that holds the brain SURFACE: scipy's erosion costs a
           sweep per voxel plus work per voxel it removes, and the centre crop removes none.  The full-volume projection is
           a * voxels + b * shell voxels with (a, b) from the two crops and the volume's shell (tissue within L1 distance 30
           of background inside its z-block) counted exactly on the GPU.
    The thread count is the fastest of a short probe AT THE BENCHMARK'S WINDOW SIZE (candidates up to os.cpu_count()).
    The HIP path then runs the SAME crop and the two results must agree (logit sums, mask, component count): a
    baseline that computes something else than the thing measured is not a baseline."""
    import torch

    from oracle import delivr_oracle as orc

    net = orc.build_unet(seed=None)
    net.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})
    net.eval()
    rng = np.random.default_rng(0)
    probe = rng.integers(0, 4000, size=(1, 1, 64, 64, 64)).astype(np.float32)
    probe_roi = rng.integers(0, 4000, size=(1, 1) + tuple(int(r) for r in roi)).astype(np.float32)
    # pick the thread count that serves the CPU best (oversubscribing a many-core host slows oneDNN down): one forward of
    # ONE BENCHMARK WINDOW per candidate (a 64^3 probe under-states what many cores do with a 128^3 window)
    best_t, best_dt, probe_log = threads, float("inf"), {}
    for cand in sorted({c for c in (8, 16, 32, 64, 128, threads) if c <= threads}):
        torch.set_num_threads(cand)
        orc.unet_forward(net, probe)  # warm-up of the primitives at this thread count
        t0 = time.perf_counter()
        orc.unet_forward(net, probe_roi)
        dt = time.perf_counter() - t0
        probe_log[str(cand)] = round(dt, 3)
        if dt < best_dt:
            best_t, best_dt = cand, dt
    threads = best_t
    torch.set_num_threads(threads)
    # (i) config 1
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.unet_forward(net, probe)
        reps.append(time.perf_counter() - t0)
    fwd64 = sorted(reps)[1]
    # (ii) the crop
    Z, Y, X = shape
    ce = [min(crop_edge, n) for n in (Z, Y, X)]
    z0, y0, x0 = (Z - ce[0]) // 2, (Y - ce[1]) // 2, (X - ce[2]) // 2
    crop_dev = vol[z0:z0 + ce[0], y0:y0 + ce[1], x0:x0 + ce[2]].contiguous()
    crop = crop_dev.cpu().numpy()
    if crop.dtype != np.uint16:
        crop = crop.view(np.uint16)
    from oracle.parity import LogitCache, flip_report, match_cells, reference_arithmetic

    win_s = []

    def timed_forward(x):
        t = time.perf_counter()
        y = orc.unet_forward(net, x)
        win_s.append(time.perf_counter() - t)
        return y

    cache = LogitCache(timed_forward)
    acc = np.zeros(crop.shape, dtype=np.float32)
    t0 = time.perf_counter()
    info = orc.sliding_window_pass(crop, roi, cache.predictor(None), acc, None, 0.5, None, 1, fp16=False)
    t1 = time.perf_counter()
    # first use of scipy.ndimage in this process: on a fresh box the import alone pages in for tens of seconds (the 60 s
    # "finalize" of BENCH_r03) - warm it up on a toy volume before anything is timed
    orc.finalize(np.zeros((8, 8, 8), np.float32), None, np.ones((8, 8, 8), np.uint16), (8, 8, 8), 0.5, 2)

    def timed_finalize(a, raw):
        ts, m = [], None
        for _ in range(3):
            t = time.perf_counter()
            m = orc.finalize(a, None, raw, raw.shape, 0.5, 30)
            ts.append(time.perf_counter() - t)
        return sorted(ts)[1], m, [round(v, 3) for v in ts]

    fin_centre_s, mask, fin_centre_reps = timed_finalize(acc, crop)
    t2 = time.perf_counter()
    lab_cpu, ncomp = orc.ccl26(mask)
    t3 = time.perf_counter()
    # (iii) the surface crop: where the x axis leaves the ellipsoid (background, 30-voxel shell, tissue)
    scrop_dev = vol[z0:z0 + ce[0], y0:y0 + ce[1], 0:ce[2]].contiguous()
    scrop = scrop_dev.cpu().numpy()
    if scrop.dtype != np.uint16:
        scrop = scrop.view(np.uint16)
    fin_surf_s, smask, fin_surf_reps = timed_finalize(np.zeros(scrop.shape, np.float32), scrop)  # (logit 0: the mask is the kept tissue)
    shell_centre = int((crop > 0).sum()) - int(orc.finalize(np.zeros(crop.shape, np.float32), None, crop, crop.shape, 0.5, 30).sum())
    shell_surf = int((scrop > 0).sum()) - int(smask.sum())
    a_vox = fin_centre_s / crop.size if shell_centre == 0 else None
    keep_full = eng.finalize(acc_full, None, vol, shape, 0.0, 30, zblock)  # threshold 0: every voxel passes, the mask is the kept tissue
    tissue_full = 0
    for zc in range(0, Z, 64):
        tissue_full += int((vol[zc:zc + 64].view(torch.int16) != 0).sum())
    shell_full = tissue_full - int(keep_full.sum(dtype=torch.int64))
    del keep_full
    if a_vox is not None and shell_surf > 0:
        b_shell = max(fin_surf_s - a_vox * scrop.size, 0.0) / shell_surf
        fin_full_s = a_vox * vox_full + b_shell * shell_full
        fin_model = "a*voxels + b*shell voxels (a from the all-tissue centre crop, b from the surface crop)"
    else:  # (a volume without an all-tissue centre / a dense volume: plain per-voxel extrapolation of the slower crop)
        b_shell = None
        fin_full_s = max(fin_centre_s / crop.size, fin_surf_s / scrop.size) * vox_full
        fin_model = "per-voxel time of the slower crop"
    n_done = info["n_windows"] - info["n_skipped"]
    per_window = float(np.median(win_s)) if win_s else (t1 - t0) / max(n_done, 1)
    projected = per_window * n_active_full + fin_full_s  # the timed region of `value`: volume -> eroded mask
    # the same windows in the REFERENCE's arithmetic (fp16 logits summed in fp16, uint8 count, fp16 divide): replayed from
    # the cached logits, untimed
    ref = reference_arithmetic(orc, crop, roi, cache, tta=False)
    lab_ref, ncomp_ref = orc.ccl26(ref["mask"])
    # the HIP path on the same crop
    g_acc = torch.zeros(crop.shape, dtype=torch.float32, device=eng.device)
    g_cnt = torch.zeros(crop.shape, dtype=torch.uint8, device=eng.device)
    eng.sw_infer(eng.make_sw_params(crop.shape, roi, 0.5, None, 0, precision), crop_dev, g_acc, g_cnt)
    g_mask = eng.finalize(g_acc, g_cnt, crop_dev, crop.shape, 0.5, 30, 0)
    g_lab, g_ncomp = eng.ccl26(g_mask.contiguous())
    g_stats = eng.cc_stats(g_lab, g_ncomp)
    ga, gm = g_acc.cpu().numpy(), g_mask.cpu().numpy()
    rel = float(np.linalg.norm(ga - acc) / max(np.linalg.norm(acc), 1e-30))
    inter, union = int(np.logical_and(gm, mask).sum()), int(np.logical_or(gm, mask).sum())
    iou = inter / union if union else 1.0
    sign = float(((ga > 0) == (acc > 0)).mean())
    vs_ref = flip_report(gm, ref["mask"], ref["mean"])
    cells = match_cells(g_lab.cpu().numpy().view(np.uint32), g_ncomp, g_stats, lab_ref, ncomp_ref, orc.cc_stats(lab_ref, ncomp_ref))
    # tolerances of tests/test_gpu_production_shapes.py / tests/test_gpu_trained_like.py for the format measured
    tol_rel, tol_iou = {"fp32": (1e-3, 0.9995), "fp16": (1e-2, 0.999), "bf16": (5e-2, 0.99)}[precision]
    ok = bool(rel <= tol_rel and iou >= tol_iou and vs_ref["iou"] >= tol_iou)
    # the other 16-bit format on the same crop (reported, not part of `ok`)
    other = None
    alt = {"fp16": "bf16", "bf16": "fp16"}.get(precision)
    if alt:
        a_acc = torch.zeros(crop.shape, dtype=torch.float32, device=eng.device)
        a_cnt = torch.zeros(crop.shape, dtype=torch.uint8, device=eng.device)
        eng.sw_infer(eng.make_sw_params(crop.shape, roi, 0.5, None, 0, alt), crop_dev, a_acc, a_cnt)
        a_mask = eng.finalize(a_acc, a_cnt, crop_dev, crop.shape, 0.5, 30, 0)
        a_lab, a_n = eng.ccl26(a_mask.contiguous())
        a_rep = flip_report(a_mask.cpu().numpy(), ref["mask"], ref["mean"])
        a_cells = match_cells(a_lab.cpu().numpy().view(np.uint32), a_n, eng.cc_stats(a_lab, a_n), lab_ref, ncomp_ref,
                              orc.cc_stats(lab_ref, ncomp_ref))
        other = {"precision": alt, "mask_iou_vs_reference_arithmetic": a_rep["iou"], "flipped_vs_reference_arithmetic": a_rep["flipped"],
                 "components_hip": int(a_n), "cells": a_cells}
        del a_acc, a_cnt, a_mask, a_lab
    return {
        "value": vox_full / projected if projected > 0 else None,
        "unit": "voxels/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_done} windows of {roi[0]}x{roi[1]}x{roi[2]} (centre crop {crop.shape} of the benchmark volume) through the "
                  f"oracle's tiler+U-Net(fp32, torch CPU)+blend in {t1 - t0:.1f} s (median window {per_window:.2f} s), finalize+30x "
                  f"erosion {fin_centre_s:.2f} s on the centre crop and {fin_surf_s:.2f} s on a surface crop (medians of 3); "
                  f"extrapolated by window count to the {n_active_full} non-background windows and by {fin_model} to the "
                  f"finalize of the benchmark volume ({fin_full_s:.0f} s)",
        "cpu": _cpu_info(),
        "thread_probe_s_per_window": probe_log,
        "finalize": {"centre_crop_s": fin_centre_s, "centre_reps_s": fin_centre_reps, "surface_crop_s": fin_surf_s,
                     "surface_reps_s": fin_surf_reps, "shell_voxels_surface_crop": shell_surf, "shell_voxels_volume": int(shell_full),
                     "tissue_voxels_volume": int(tissue_full), "s_per_voxel": a_vox, "s_per_shell_voxel": b_shell,
                     "projected_volume_s": fin_full_s, "model": fin_model},
        "window_s": {"median": per_window, "min": float(min(win_s)) if win_s else None, "max": float(max(win_s)) if win_s else None},
        "seconds_per_window": per_window,
        "forward_64cube_fp32_s": fwd64,
        "forward_64cube_voxels_per_s": 64.0**3 / fwd64,
        "crop": {"shape": list(crop.shape), "windows": info["n_windows"], "pass_s": t1 - t0, "finalize_erosion_s": fin_centre_s,
                 "labels_s": t3 - t2, "voxels_per_s_end_to_end": crop.size / ((t1 - t0) + fin_centre_s + (t3 - t2)), "components": int(ncomp)},
        "agreement": {"ok": ok, "precision": precision, "weights": weights_name, "logit_sum_rel_l2": rel, "sign_agreement": sign,
                      "mask_iou": iou, "components_hip": int(g_ncomp), "components_cpu": int(ncomp),
                      # against the oracle run in the reference's own arithmetic (fp16 accumulate, uint8 count, fp16 divide)
                      "mask_iou_vs_reference_arithmetic": vs_ref["iou"], "flipped_vs_reference_arithmetic": vs_ref["flipped"],
                      "flip_margin_hist": {"edges": vs_ref["hist_edges"], "counts": vs_ref["hist"]},
                      "components_reference_arithmetic": int(ncomp_ref), "cells": cells, "other_format": other,
                      "tol_rel_l2": tol_rel, "tol_iou": tol_iou},
    }


def step_walls_pair(spec, sd):
    """run_inference + count_blobs on the file spec["nifti"], twice (first brain of the process, next brain): wall clock,
    breakdown and transfer rates of each call"""
    import contextlib
    import io
    import shutil

    from delivr_cfos_amd import hostio
    from delivr_cfos_amd.count_blobs import count_blobs
    from delivr_cfos_amd.inference.inference import run_inference

    Z, Y, X = spec["stack"]
    log = io.StringIO()
    settings = {"postprocessing": {"output_location": spec["post_dir"]}}
    runs = []
    for which in ("first_brain", "next_brain"):
        shutil.rmtree(spec["out_dir"], ignore_errors=True)
        shutil.rmtree(spec["post_dir"], ignore_errors=True)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(log):  # (the step's own progress lines: bench.py prints ONE line)
            run_inference([spec["nifti"]], spec["out_dir"], (1, 1, Z, Y, X), comment="brain", tta=bool(spec["tta"]),
                          crop_size=tuple(spec["roi"]), state_dict={"state_dict": sd}, precision=spec["precision"])
        step2 = time.perf_counter() - t0
        t2 = dict(getattr(run_inference, "last_timings", {}))
        xfer = {k: dict(v) for k, v in hostio.last_transfer.items() if k in ("h2d_volume", "d2h_mask")}
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(log):
            n_comp = count_blobs(settings, spec["out_dir"], 0, "brain", (1, 1, Z, Y, X))
        step3 = time.perf_counter() - t0
        t3 = dict(getattr(count_blobs, "last_timings", {}))
        xfer.update({k: dict(v) for k, v in hostio.last_transfer.items() if k in ("h2d_mask", "d2h_labels")})
        runs.append({"which": which, "step2_wall_s": step2, "step3_wall_s": step3, "step2_breakdown": t2, "step3_breakdown": t3,
                     "transfers": xfer, "components": int(n_comp)})
    # ... and three brains the way python -m delivr_cfos_amd runs a batch: while one brain's passes run, the next volume is read into
    # HBM and the previous binaries.npy streams out (run_inference prefetch / defer_write); the label files stream out behind the
    # next brain's labelling (count_blobs defer_write).  Per-brain wall = total / 3, every file complete (wait_deferred) inside it.
    nb = int(spec.get("pipelined_brains", 3))
    if nb > 0:
        names = [f"pipe{k}" for k in range(nb)]
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(log):
            for k, name in enumerate(names):
                run_inference([spec["nifti"]], spec["out_dir"], (1, 1, Z, Y, X), comment=name, tta=bool(spec["tta"]), crop_size=tuple(spec["roi"]),
                              state_dict={"state_dict": sd}, precision=spec["precision"], prefetch=spec["nifti"] if k + 1 < nb else None,
                              defer_write=True)
            hostio.wait_deferred()
        p2 = (time.perf_counter() - t0) / nb
        t0 = time.perf_counter()
        comps = []
        with contextlib.redirect_stdout(log):
            for k, name in enumerate(names):
                comps.append(int(count_blobs(settings, spec["out_dir"], k, name, (1, 1, Z, Y, X), defer_write=True)))
            hostio.wait_deferred()
        p3 = (time.perf_counter() - t0) / nb
        same = all(c == runs[-1]["components"] for c in comps) and all(
            os.path.isfile(os.path.join(spec["post_dir"], f"{name}-{c}-cc3d.npy")) for name, c in zip(names, comps))
        for name in names:  # (the sequential brain's files stay for the caller's check)
            shutil.rmtree(os.path.join(spec["out_dir"], name), ignore_errors=True)
        for f in os.listdir(spec["post_dir"]):
            if f.startswith("pipe") or "_pipe" in f:
                os.remove(os.path.join(spec["post_dir"], f))
        runs.append({"which": "pipelined", "brains": nb, "step2_per_brain_s": p2, "step3_per_brain_s": p3, "files_ok": bool(same),
                     "components": comps[-1]})
    return runs


def step_walls_child(spec_json):
    """`python bench.py --step-walls-child <spec>`: the two steps on two brains in a process that has done nothing else"""
    spec = json.loads(spec_json)
    from delivr_cfos_amd.weights import random_state_dict, trained_like_state_dict

    sd = trained_like_state_dict(TRAINED_LIKE_FIXTURE) if spec["weights"] == "trained-like" else random_state_dict(seed=0)
    runs = step_walls_pair(spec, sd)
    print(json.dumps({"step_walls_child": runs}))


def step_walls(eng, vol, stack, shape, roi, tta, precision, sd, mask_voxels, weights_name):
    """File -> file wall-clock of the two steps the reference's caller sees (__main__.py:133-140,166): `run_inference`
    (masked_nifti.npy -> binaries.npy) and `count_blobs` (binaries.npy -> <brain>-<N>-cc3d.npy + statistics + CSV) on THIS
    volume, files on tmpfs (/dev/shm; $DLV_BENCH_TMP overrides), so that what is timed is the step's own host work - reads,
    PCIe, the passes, writes - and not a disk.  Returns the `step_walls` block of the bench line (H2D / D2H rates included)."""
    import shutil
    import tempfile

    from delivr_cfos_amd import hostio
    from delivr_cfos_amd.count_blobs import count_blobs
    from delivr_cfos_amd.inference.inference import run_inference

    Z, Y, X = stack
    need = int(vol.numel()) * 2 + 4 * Z * Y * X * (1 + 4) + (1 << 30)  # (input + the outputs of the sequential brain and of three pipelined ones)
    base = os.environ.get("DLV_BENCH_TMP") or "/dev/shm"
    try:
        free = shutil.disk_usage(base).free
        avail = None
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
    except OSError as exc:
        return {"skipped": f"{base}: {exc}"}
    if free < 1.25 * need or (avail is not None and avail < 2.5 * need):
        return {"skipped": f"{base} has {free / 2**30:.0f} GiB free, the host {0 if avail is None else avail / 2**30:.0f} GiB available: "
                           f"{need / 2**30:.0f} GiB of files needed"}
    d = tempfile.mkdtemp(prefix="dlv_bench_", dir=base)
    try:
        nifti_dir = os.path.join(d, "01_mask", "brain", "masked_niftis")
        out_dir = os.path.join(d, "02_blob")
        post_dir = os.path.join(d, "03_post") + "/"
        os.makedirs(nifti_dir)
        nifti = os.path.join(nifti_dir, "masked_nifti.npy")
        t0 = time.perf_counter()
        hostio.save_npy(eng, vol.reshape((1, 1) + tuple(shape)), nifti, np.uint16, what="d2h_volume")
        write_volume_s = time.perf_counter() - t0
        if os.path.getsize(nifti) != 128 + int(vol.numel()) * 2:
            return {"skipped": "masked_nifti.npy: numpy's header is not the 128 bytes the reference assumes (inference.py:234)"}
        # Two brains in a row (python -m delivr_cfos_amd loops over brains: the shared engine and torch's allocator keep what they
        # allocated), in a FRESH child process - what the CLI is: this process has run a benchmark, holds ~100 GB in caches, and
        # its late allocations are slow (profiles/r06r_alloc_probe2.json), which is not what a user's first brain meets.  The child is
        # started before it touches the GPU (bench.py --step-walls-child); $DLV_BENCH_WALLS_INPROC=1 runs the pair in this process.
        import torch

        torch.cuda.empty_cache()  # (this process's caches - tens of GB after the extras - would push the child towards the slab-streamed path)
        spec = {"nifti": nifti, "out_dir": out_dir, "post_dir": post_dir, "stack": [Z, Y, X], "roi": list(roi), "tta": bool(tta),
                "precision": precision, "weights": weights_name}
        if os.environ.get("DLV_BENCH_WALLS_INPROC") == "1":
            runs = step_walls_pair(spec, sd)
            where = "this process (after the benchmark)"
        else:
            import subprocess

            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--step-walls-child", json.dumps(spec)], capture_output=True,
                               text=True, timeout=1800)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"step_walls_child"')]
            if r.returncode != 0 or not lines:
                return {"skipped": f"child process failed (rc {r.returncode}): {r.stderr[-600:]}"}
            runs = json.loads(lines[-1])["step_walls_child"]
            where = "a fresh child process (python bench.py --step-walls-child)"
        pipe = runs.pop() if runs and runs[-1].get("which") == "pipelined" else None
        n_comp = runs[-1]["components"]
        # the files the next step / the reference's consumers read
        binaries = np.load(os.path.join(out_dir, "brain", "binary_segmentations", "binaries.npy"), mmap_mode="r")
        lab_path = os.path.join(post_dir, f"brain-{n_comp}-cc3d.npy")
        labels = np.load(lab_path, mmap_mode="r")
        fg = int(np.count_nonzero(binaries))
        n_lab = 0
        for zc in range(0, Z, 64):  # (the labels number the components 1..N in raster order: the largest label is N)
            n_lab = max(n_lab, int(labels[zc:zc + 64].max()))
        files_ok = bool(binaries.shape == (Z, Y, X) and binaries.dtype == np.uint8 and fg == int(mask_voxels) and labels.shape == (Z, Y, X)
                        and n_lab == int(n_comp) and runs[0]["components"] == runs[1]["components"]
                        and os.path.isfile(os.path.join(post_dir, "brain-stats.pickle"))
                        and os.path.isfile(post_dir + f"({Z}, {Y}, {X})_brain.csv") and (pipe is None or pipe["files_ok"]))
        first, nxt = runs
        tr = nxt["transfers"]
        return {"step2_wall_s": first["step2_wall_s"], "step3_wall_s": first["step3_wall_s"],
                "step2_wall_next_brain_s": nxt["step2_wall_s"], "step3_wall_next_brain_s": nxt["step3_wall_s"],
                "first_brain": first, "next_brain": nxt,
                # a batch of brains as the CLI runs it (next volume read and previous files written behind the passes / the labelling)
                "pipelined": pipe, "step2_wall_pipelined_per_brain_s": pipe["step2_per_brain_s"] if pipe else None,
                "step3_wall_pipelined_per_brain_s": pipe["step3_per_brain_s"] if pipe else None,
                "h2d_volume": tr.get("h2d_volume"), "d2h_mask": tr.get("d2h_mask"), "h2d_mask": tr.get("h2d_mask"),
                "d2h_labels": tr.get("d2h_labels"), "write_input_volume_s": write_volume_s, "components": int(n_comp),
                "label_dtype": str(labels.dtype), "mask_voxels_in_file": fg, "files_ok": files_ok, "files_on": base,
                "io_threads": {"read": hostio.io_threads(), "write": hostio.WRITE_THREADS},
                "process": where,
                "what": "run_inference(masked_nifti.npy -> binaries.npy) and count_blobs(binaries.npy -> labels .npy, stats pickle, CSV) "
                        "called as python -m delivr_cfos_amd calls them, files on tmpfs; wall clock of each call - for the first brain of "
                        "a process (fresh context: workspaces allocated) and for the next one (context, workspaces and pinned staging kept)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--dense", action="store_true", help="no background: every window runs the network")
    ap.add_argument("--precision", default="fp16", choices=["bf16", "fp16", "fp32"],
                    help="fp16 (default): IEEE-half MFMA operands and storage, mask IoU 0.9998 vs the oracle; bf16: bf16 at levels 1-4 of the "
                         "U-Net, fp16 at level 0 (DLV_PREC_BF16), IoU 0.9994, the same speed; fp32: the VALU parity path")
    ap.add_argument("--sw-batch", type=int, default=0)
    ap.add_argument("--weights", default="trained", choices=["trained", "random"],
                    help="trained (default): the trained-like checkpoint (tests/golden/trained_like_weights.npz: top levels trained on the "
                         "reference's patches + synth volumes, deep levels seeded random; bimodal logits, a mask of small blobs); "
                         "random: seeded random weights (margin-free logits, one giant component)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=256, help="edge of the centre crop the CPU baseline runs (256: 27 windows of 128^3)")
    ap.add_argument("--no-dense", action="store_true", help="skip the extra pass with the background skip disabled (`value_dense`)")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stages either side of the pass (finalize, CCL-26 + statistics, "
                    "resamplers, one Gaussian-blend pass: BASELINE configs 4/5), which run by default at N=1")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra single-lane step that times the kernels alone")
    ap.add_argument("--diag", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B runs: a kernel-selection switch of include/delivr_hip_diag.h (dlv_diag_set) for the benchmark's engine, e.g. "
                         "--diag fuse_levels=1; recorded in config.diag")
    ap.add_argument("--step-walls-child", default=None, help=argparse.SUPPRESS)  # (internal: the fresh process of `step_walls`)
    ap.add_argument("--no-step-walls", action="store_true", help="skip the file -> file wall-clock of run_inference and count_blobs "
                    "on this volume (N = 1; ~30 GiB of files on /dev/shm for c3)")
    args = ap.parse_args()

    if args.step_walls_child:
        step_walls_child(args.step_walls_child)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the contract's launch line as a CHILD process - before this
        # process has imported torch or touched the GPU (a process that has initialised HIP must never exec or fork workers) -
        # and relay its output and exit code
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (world == 1 and args.gpus == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    # (device_count() does not initialise the GPU on this image)
    ndev = torch.cuda.device_count()
    if world > 1 and os.environ.get("DLV_BENCH_SAME_DEVICE") != "1" and ndev < world:
        raise SystemExit(f"bench.py --gpus {world}: needs {world} devices, this box has {ndev} (one rank per GPU; "
                         "DLV_BENCH_SAME_DEVICE=1 runs the ranks on cuda:0 over gloo as a functional check)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # functional check of the N>1 path on a single GPU: DLV_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # uses gloo (RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank + RCCL
    same_device = os.environ.get("DLV_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # DLV_BENCH_FORCE_DIST=1 (with `torch.distributed.run --nproc-per-node 1`): take the N > 1 code path - nccl (= RCCL)
    # process group, weight broadcast, balanced plan with its all_gather_object, p2p self-test, all_reduce, barrier - at world
    # size 1, so that every RCCL call of the multi-GPU path has run once on a one-GPU box (tests/test_gpu_rccl.py)
    dist_mode = world > 1 or os.environ.get("DLV_BENCH_FORCE_DIST") == "1"
    if dist_mode:
        import torch.distributed as dist

        if same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.hostlogic import arrayterator_zblock, padded_shape, pass_schedule
    from delivr_cfos_amd.parallel import balanced_plan, broadcast_weights, exchange_seams, finalize_owned, p2p_selftest, plan_from_params
    from delivr_cfos_amd.synth import synth_planes_torch, synth_volume_torch
    from delivr_cfos_amd.weights import random_state_dict, trained_like_state_dict

    stack, roi, seed, tta = WORKLOADS[args.workload]
    Z, Y, X = stack
    shape = padded_shape(stack, roi)  # the volume the windows tile (Zp, Yp, Xp)
    Zp, Yp, Xp = shape
    schedule = pass_schedule(tta)     # [(flip_dim, repeat)]: the distinct passes of one inference

    def planes(lo, hi):
        """planes [lo, hi) of the padded volume: the synthetic brain fills the stack, the padding is background"""
        if (Zp, Yp, Xp) == (Z, Y, X):
            return synth_planes_torch(stack, seed, eng.device, lo, hi, dense=args.dense)
        out = torch.zeros((hi - lo, Yp, Xp), dtype=torch.uint16, device=eng.device)
        if min(hi, Z) > lo:
            out[: min(hi, Z) - lo, :Y, :X] = synth_planes_torch(stack, seed, eng.device, lo, min(hi, Z), dense=args.dense)
        return out

    eng = HipEngine(local_rank)
    for kv in args.diag:
        name, _, val = kv.partition("=")
        eng.diag_set(name, int(val or 1))
    if args.weights == "trained" and not os.path.isfile(TRAINED_LIKE_FIXTURE):
        # (a different checkpoint means another mask, skip fraction and CCL workload under the same metric name)
        raise SystemExit(f"bench.py: the trained-like checkpoint {TRAINED_LIKE_FIXTURE} is missing; pass --weights random to "
                         "benchmark on seeded random weights instead")
    weights_name = "trained-like" if args.weights == "trained" else "seeded random"
    sd = trained_like_state_dict(TRAINED_LIKE_FIXTURE) if weights_name == "trained-like" else random_state_dict(seed=0)
    if rank == 0:
        eng.load_state_dict({"state_dict": sd})
    if dist_mode:
        broadcast_weights(eng, dist, rank)
        # one ring exchange of a seam-sized buffer through batch_isend_irecv, compared byte for byte: a broken p2p transport
        # fails HERE with a clear message, not as a wrong mask after the pass
        p2p_selftest(dist, eng.device, rank, world, 64 << 20)

    params_all = eng.make_sw_params(shape, roi, 0.5, None, 0, args.precision, sw_batch=args.sw_batch)
    nb = arrayterator_zblock((Z, Y, X))
    if not dist_mode:
        vol = planes(0, Zp)
        torch.cuda.synchronize()
        plan = plan_from_params(params_all, 1, None)
        slo, shi = 0, Zp
    else:
        # slab-resident: every rank generates (a real run: reads) and holds only the planes of ITS Z-slab; the shards are
        # balanced by the windows that actually run the network (a brain fills the central slabs, not the outer ones)
        t_plan = time.perf_counter()
        plan, slo, shi, vol = balanced_plan(
            eng, params_all, planes, world, rank, dist, Z, 30, nb)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t_plan  # (unweighted slab -> window maxima -> all_gather -> weighted plan -> the slab completed)
    wb, we = plan.win_ranges[rank]

    def pass_params(precision, skip_threshold=0):
        """one dlv_sw_params per distinct pass of the schedule (None: this rank has no windows - it still takes part in the exchange)"""
        if we <= wb:
            return None
        return [eng.make_sw_params(shape, roi, 0.5, flip, skip_threshold, precision, sw_batch=args.sw_batch, win_range=(wb, we),
                                   slab=(slo, shi - slo), repeat=rep_n) for flip, rep_n in schedule]

    params = pass_params(args.precision)
    acc = torch.zeros((shi - slo, Yp, Xp), dtype=torch.float32, device=eng.device)

    stats_last = {}
    cur = {"params": params}

    phase_s = {}  # (filled by step(phases=True): one extra, untimed step with a device synchronisation after every phase)

    def step(phases=False):
        def lap(name, t0):
            if phases:
                eng.sync()
                torch.cuda.synchronize()
                phase_s[name] = time.perf_counter() - t0
            return time.perf_counter()

        t = lap("", 0.0)
        acc.zero_()
        if cur["params"] is not None:
            for q in cur["params"]:
                stats_last.update(eng.sw_infer(q, vol, acc))
        t = lap("windows_ms", t)
        if dist_mode:
            eng.sync()
            exchange_seams(acc, plan, rank, dist, z0=slo)
            t = lap("seam_ms", t)
        slab, _, _ = finalize_owned(eng, plan, rank, acc, None, vol, (Z, Y, X), 0.5, 30, z0=slo)
        t = lap("finalize_ms", t)
        if slab is None:
            slab = torch.empty((0, Y, X), dtype=torch.uint8, device=eng.device)
        # N > 1: the mask stays on the ranks as Z-slabs (the planes each rank owns) - what the sharded count_blobs /
        # parallel.ccl_sharded consume; nothing is gathered to rank 0
        return slab

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist_mode:
            dist.barrier()

    def timed_steps(k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            out = step()
        fence()
        dt = time.perf_counter() - t0
        if dist_mode:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if same_device else eng.device)  # (gloo reduces host tensors)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(args.warmup):
        step()
    # the timed region: EXACTLY `steps` passes, no per-kernel event bracketing inside it
    elapsed, slab = timed_steps(args.steps)
    stats_timed = dict(stats_last)
    # the same passes again with the HIP-event bracketing of every launch switched on (dlv_prof_*): the per-kernel view
    prof, elapsed_prof = {}, None
    if not args.no_prof:
        eng.prof_reset()
        eng.prof_enable(True)
        elapsed_prof, _ = timed_steps(args.steps)
        prof = eng.prof_report()
        eng.prof_enable(False)
    # every window through the network (background skip disabled): the number the MFMA ceiling is quoted against
    elapsed_dense, stats_dense = None, {}
    if not args.no_dense and not args.dense:
        cur["params"] = pass_params(args.precision, -1)  # skip_threshold -1: no window's maximum is <= -1
        elapsed_dense, _ = timed_steps(1)
        stats_dense = dict(stats_last)
        cur["params"] = params
    # N > 1: where a rank's step goes - its windows, the seam exchange, the finalize of the planes it owns - from one extra step with
    # a synchronisation after every phase (the first 8-GPU record can be held against DESIGN section 6's model line by line)
    per_rank_phases = None
    if dist_mode:
        fence()
        step(phases=True)
        fence()
        mine = {k: round(1e3 * v, 3) for k, v in phase_s.items() if k}
        mine["window_max_gather_ms"] = round(1e3 * plan_s, 3)
        mine["slab_planes"] = [int(slo), int(shi)]
        per_rank_phases = [None] * world
        dist.all_gather_object(per_rank_phases, mine)
    # the other 16-bit format, one pass (BASELINE's configs name bf16; fp16 is the default for its closer masks)
    elapsed_alt, alt_prec = None, {"fp16": "bf16", "bf16": "fp16"}.get(args.precision)
    if alt_prec and not args.no_dense and not args.dense:
        cur["params"] = pass_params(alt_prec)
        step()  # warm-up of the other format's kernels
        elapsed_alt, _ = timed_steps(1)
        cur["params"] = params
    if not args.no_dense and not args.dense:
        slab = step()  # leave acc / slab holding the benchmark's own pass (the extras and the CPU leg read them)
        fence()
    stats_last.clear()
    stats_last.update(stats_timed)
    # the mask of the benchmark's own pass: foreground voxels and a position-weighted checksum (identical for every N up to
    # the voxels whose mean logit is within fp32 rounding of 0: the seam sums associate differently), summed over the ranks
    own_lo = plan.z_owned[rank][0] if dist_mode else 0
    CK_MOD = (1 << 55) - 55  # per-rank residues of up to 16 ranks add up without leaving int64
    n_fg, ck = 0, 0
    for zc in range(0, int(slab.shape[0]), 32):  # plane blocks: torch's arange / int64 temporaries stay small
        m64 = slab[zc:zc + 32].reshape(-1).to(torch.int64)
        idx = torch.arange(m64.numel(), dtype=torch.int64, device=eng.device) + (int(own_lo) + zc) * Y * X
        n_fg += int(m64.sum())
        ck = (ck + int((m64 * (((idx % 2147483629) * 48271) % 2147483629)).sum())) % CK_MOD
        del m64, idx
    mask_sig = torch.tensor([n_fg, ck], dtype=torch.int64, device=eng.device)
    if dist_mode:
        st = torch.tensor([stats_last.get("n_windows", 0), stats_last.get("n_skipped", 0)], dtype=torch.int64,
                          device=eng.device)
        if dist.get_backend() == "gloo":
            st, mask_sig = st.cpu(), mask_sig.cpu()
        per_rank = [None] * world
        dist.all_gather_object(per_rank, [int(st[0]), int(st[1])])  # (windows, of which skipped) of every rank's range
        dist.all_reduce(st)
        dist.all_reduce(mask_sig)
        n_windows, n_skipped = int(st[0]), int(st[1])
    else:
        n_windows, n_skipped = stats_last.get("n_windows", 0), stats_last.get("n_skipped", 0)
        per_rank = None
    mask_voxels, mask_checksum = int(mask_sig[0]), int(mask_sig[1]) % CK_MOD

    if rank != 0:
        if dist_mode:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    vox = float(Z) * Y * X
    value = vox / (elapsed / args.steps)
    n_active = n_windows - n_skipped
    n_passes = len(schedule)  # distinct passes per step (1, or 3 under TTA: the 13 of the reference weighted 5:4:4)
    tile_vox = float(roi[0] * roi[1] * roi[2])

    # ---- roofline of the dominant kernel (rank 0's HIP-event timings) -----------------------------------
    def roofline_of(prof, lanes, steps_covered):
        if not prof:
            return None, {}
        kernels = {}
        for name, e in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
            kernels[name] = {"launches": e["launches"], "total_ms": round(e["total_ms"], 3),
                             "avg_us": round(1e3 * e["total_ms"] / max(e["launches"], 1), 2)}
        name, e = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        avg_s = 1e-3 * e["total_ms"] / max(e["launches"], 1)
        if e["flops"] > 0 and name.startswith(("conv3_", "deconv2_mfma")):
            ach = e["flops"] / max(e["launches"], 1) / avg_s / 1e12
            r = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach / PEAK_BF16_TFLOPS, "traffic": None, "avg_launch_us": 1e6 * avg_s, "launches": e["launches"]}
        else:
            ach = e["bytes"] / max(e["launches"], 1) / avg_s / 1e9
            r = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                 "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_launch_us": 1e6 * avg_s, "launches": e["launches"]}
        if name.endswith("_act") and name[:-4] in prof and prof[name[:-4]]["launches"]:
            # the dominant kernel is the instantiation that also normalises + activates its input while it stages it (default since
            # the end of round 6 for upcat_1.conv_1: the pass is 1.4-1.7 % faster): its FLOPs are those of the convolution
            # alone, the InstanceNorm + Mish of 1.33 x its input ride on top - the same kernel without that work, for comparison:
            tw = prof[name[:-4]]
            tw_s = 1e-3 * tw["total_ms"] / tw["launches"]
            tw_ach = tw["flops"] / tw["launches"] / tw_s / 1e12
            r["note"] = ("this instantiation also applies the producer layer's InstanceNorm + Mish to its input while staging it "
                         "(replaces a normalisation pass of ~0.66 ms per launch at 128^3 windows); `achieved` counts the convolution's FLOPs only")
            r["plain_twin"] = {"kernel": name[:-4], "avg_launch_us": 1e6 * tw_s, "launches": tw["launches"], "achieved": tw_ach,
                               "frac": tw_ach / PEAK_BF16_TFLOPS}
        # HBM traffic per launch of that kernel from the committed PMC passes of THIS build and format
        # (profiles/run_pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as the gfx950
        # guide prescribes).  The passes ran the 512^3 workload (25k launches of C3 under PMC exceed the time limit): same
        # kernels, same batch of 16 windows per launch; a launch of another batch is scaled by algorithmic bytes.
        # ... and only of THIS library: the JSON records the sha256 of the .so that ran under the counters and of its sources
        # (profiles/make_traffic.py); when neither matches what is loaded now, `traffic` stays null and says why.
        try:
            from delivr_cfos_amd._lib import build_fingerprint

            cands = [(f"traffic_r06_{args.workload}.json", False), ("traffic_r06_c2.json", True), ("traffic_r05_c2.json", True)]
            tfile, scaled = next(((os.path.join(ROOT, "profiles", n), sc) for n, sc in cands if os.path.isfile(os.path.join(ROOT, "profiles", n))),
                                 (None, True))
            if tfile is None:
                r["traffic_note"] = "no PMC traffic file of this round under profiles/"
            elif args.sw_batch != 0:
                r["traffic_note"] = "non-default --sw-batch: the PMC passes ran the default batch"
            else:
                tj = json.load(open(tfile))
                fp_now, fp_then = build_fingerprint(), tj.get("fingerprint") or {}
                same_lib = fp_then.get("lib_sha256") is not None and fp_then.get("lib_sha256") == fp_now["lib_sha256"]
                same_src = fp_then.get("src_sha256") is not None and fp_then.get("src_sha256") == fp_now["src_sha256"]
                if not (same_lib or same_src):
                    r["traffic_note"] = (f"{os.path.relpath(tfile, ROOT)} was measured on another build (library sha256 "
                                         f"{str(fp_then.get('lib_sha256'))[:12]}, sources {str(fp_then.get('src_sha256'))[:12]}; loaded: "
                                         f"{str(fp_now['lib_sha256'])[:12]}, {str(fp_now['src_sha256'])[:12]}): regenerate with profiles/run_pmc_traffic.sh")
                elif name not in tj["kernels"] or tj.get("precision") != args.precision:
                    r["traffic_note"] = f"{os.path.relpath(tfile, ROOT)} holds no {args.precision} figure for {name}"
                else:
                    k = tj["kernels"][name]
                    r["algorithmic_bytes"] = e["bytes"] / max(e["launches"], 1)
                    t = k["traffic_bytes"]
                    if k.get("algorithmic_bytes"):  # the PMC run's launches held a different number of windows on average
                        t *= r["algorithmic_bytes"] / k["algorithmic_bytes"]
                    r["traffic"] = t
                    r["traffic_over_algorithmic"] = t / r["algorithmic_bytes"] if r["algorithmic_bytes"] else None
                    r["traffic_source"] = os.path.relpath(tfile, ROOT) + (
                        " (PMC run of the c2 workload, scaled by algorithmic bytes per launch)" if scaled else "") + (
                        f"; same library sha256 {fp_now['lib_sha256'][:12]}" if same_lib else f"; same sources sha256 {fp_now['src_sha256'][:12]} (library rebuilt)") + (
                        f"; git {tj.get('git_head')}" if tj.get("git_head") else "")
        except Exception as ex:
            r["traffic_note"] = f"traffic lookup failed: {ex}"
        r["lanes"] = lanes
        net_ms = sum(v["total_ms"] for v in prof.values())
        r["forward_tflops"] = (FLOP_PER_PATCH_VOXEL * tile_vox * n_active * n_passes * steps_covered / (1e-3 * net_ms) / 1e12
                               if net_ms > 0 and world == 1 and lanes == 1 else None)
        return r, kernels

    lanes_used = int(os.environ.get("DLV_LANES", "3"))
    roofline, kernels = roofline_of(prof, lanes_used, args.steps)
    # the timed region runs two overlapping lanes, which stretches every kernel's event-to-event time; one extra,
    # untimed step on a single lane gives the dominant kernel's own efficiency
    roofline_isolated = None
    if prof and lanes_used > 1 and not args.no_isolated and world == 1:
        eng.set_lanes(1)
        # (one unmeasured one-lane pass first: the chip has just spent a second on the checksum's light kernels, and a pass that
        # starts on a rested chip runs above the sustained clock for its first seconds - 1276 us per launch of the hot kernel
        # against the 1340 us rocprofv3 sees over whole one-lane passes of the same box, profiles/README.md r06final)
        step()
        fence()
        eng.prof_reset()
        eng.prof_enable(True)
        step()
        fence()
        prof1 = eng.prof_report()
        eng.prof_enable(False)
        eng.set_lanes(lanes_used)
        roofline_isolated, _ = roofline_of(prof1, 1, 1)

    # ---- the stages either side of the pass (BASELINE configs 4 and 5), timed separately, N=1 ----------------------
    # every entry: ms per call on this volume, the ALGORITHMIC HBM bytes of the stage (inputs read once + outputs written
    # once, stated per voxel), and hbm_frac = bytes / time / 8 TB/s
    extras = None
    if not args.no_extras and world == 1:
        def timed(fn, reps=2):
            fn()
            eng.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = None
            for _ in range(reps):
                out = None  # hand the previous result back to the caching allocator: a fresh 17 GB hipMalloc per call is not CCL time
                out = fn()
            eng.sync()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / reps, out

        def entry(ms, nbytes, what):
            return {"ms": ms, "algorithmic_bytes": nbytes, "bytes_per_voxel": what, "GBps": nbytes / (ms * 1e6),
                    "hbm_frac": nbytes / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9)}

        extras = {"voxels": vox}
        zb = arrayterator_zblock(stack)
        ms, mask = timed(lambda: eng.finalize(acc, None, vol, stack, 0.5, 30, zb), reps=2)
        extras["finalize"] = entry(ms, vox * 7, "4 (fp32 sums) + 2 (uint16 raw) read, 1 (mask) written")
        mask = mask.contiguous()
        ms, (labels, ncomp) = timed(lambda: eng.ccl26(mask))
        extras["ccl26"] = entry(ms, vox * 5, "1 (mask) read, 4 (uint32 labels) written")
        extras["ccl26"]["components"] = ncomp
        extras["ccl26"]["mask"] = f"the pass's own mask ({weights_name} weights)"
        ms, _ = timed(lambda: eng.cc_stats(labels, ncomp), reps=1)
        extras["cc_stats"] = entry(ms, vox * 4, "4 (labels) read")
        del labels
        # the synthetic cells themselves (blobs of 10-40 voxels, ~4e-4 per tissue voxel): a second, denser cell mask
        volv = vol if tuple(shape) == tuple(stack) else vol[:Z, :Y, :X].contiguous()  # (the stack without its padding)
        cells = (volv.view(torch.int16) > 6500).to(torch.uint8) if volv.dtype == torch.uint16 else (volv > 6500).to(torch.uint8)
        ms, (labels, ncells) = timed(lambda: eng.ccl26(cells))
        extras["ccl26_cells"] = entry(ms, vox * 5, "1 (mask) read, 4 (uint32 labels) written")
        extras["ccl26_cells"]["components"] = ncells
        ms, _ = timed(lambda: eng.cc_stats(labels, ncells), reps=1)
        extras["cc_stats_cells"] = entry(ms, vox * 4, "4 (labels) read")
        del labels, cells
        ms, ds = timed(lambda: eng.block_mean_u16(volv, (4, 15, 15)))
        extras["block_mean_4x15x15"] = entry(ms, vox * 2 + ds.numel() * 2, "2 (uint16) read, 2/900 written")
        small = (ds.to(torch.int32) > 0).to(torch.uint8)
        ms, _ = timed(lambda: eng.zoom_spline2_u8(small, stack), reps=1)
        extras["zoom_spline2_to_full"] = entry(ms, vox * 1 + small.numel(), "1 (uint8 mask) written, 1/900 read")
        del small, ds
        # one pass with MONAI's Gaussian importance map instead of constant weights (option)
        if params is not None:
            wsum = torch.zeros_like(acc)
            gp = eng.make_sw_params(shape, roi, 0.5, None, 0, args.precision, sw_batch=args.sw_batch, blend="gaussian", wsum=wsum)  # (one plain pass)
            acc_g = torch.zeros_like(acc)
            fence()
            t0 = time.perf_counter()
            eng.sw_infer(gp, vol, acc_g)
            fence()
            extras["gaussian_blend_pass_ms"] = 1e3 * (time.perf_counter() - t0)
            del wsum, acc_g
        # the configuration the reference SHIPS on the same brain (config.json:24-28,63: windows 96 x 96 x 64, test-time
        # augmentation = 13 passes = 3 distinct ones weighted 5:4:4; `--workload default` is the full bench of it): one warm-up
        # pass (the shapes' workspaces), then one timed inference of the volume = 3 passes + finalize
        if args.workload == "c3" and not args.dense and params is not None:
            d_roi = (96, 96, 64)
            d_pad = padded_shape(stack, d_roi)
            vol_d = torch.zeros(d_pad, dtype=torch.uint16, device=eng.device)
            vol_d[:Z, :Y, :X] = vol
            acc_d = torch.zeros(d_pad, dtype=torch.float32, device=eng.device)
            qs = [eng.make_sw_params(d_pad, d_roi, 0.5, flip, 0, args.precision, repeat=rep_n) for flip, rep_n in pass_schedule(True)]
            eng.sw_infer(qs[0], vol_d, acc_d)
            fence()
            acc_d.zero_()
            fence()
            t0 = time.perf_counter()
            for q in qs:
                st_d = eng.sw_infer(q, vol_d, acc_d)
            mask_d = eng.finalize(acc_d, None, vol_d, stack, 0.5, 30, zb)
            fence()
            dt_d = time.perf_counter() - t0
            act_d = st_d["n_windows"] - st_d["n_skipped"]
            pv_d = float(d_roi[0] * d_roi[1] * d_roi[2]) * act_d * len(qs) / dt_d
            pv_here = tile_vox * n_active * n_passes / (elapsed / args.steps)
            extras["default_config"] = {
                "what": "the reference's shipped configuration on this brain: windows 96x96x64, TTA (13 passes = 3 distinct passes weighted "
                        f"5:4:4), volume zero-padded to {d_pad[0]}x{d_pad[1]}x{d_pad[2]}; one inference = 3 passes + finalize",
                "ms_per_inference": 1e3 * dt_d, "voxels_per_s": vox / dt_d, "windows": st_d["n_windows"], "windows_skipped": st_d["n_skipped"],
                "patch_voxels_per_s": pv_d, "patch_voxels_per_s_over_this_line": pv_d / pv_here,
                "mask_voxels": int(mask_d.sum(dtype=torch.int64))}
            del vol_d, acc_d, mask_d

    walls = None
    if not args.no_step_walls and world == 1 and not dist_mode:
        try:
            walls = step_walls(eng, vol, stack, shape, roi, tta, args.precision, sd, mask_voxels, weights_name)
        except Exception as exc:  # reported, never fatal for the metric
            walls = {"skipped": f"failed: {exc!r}"}

    cpu = None
    if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N=1 only (bench contract)
        try:
            threads = os.cpu_count() or 1
            cpu = cpu_baseline(eng, sd, vol, stack, roi, n_active * n_passes, vox, threads, args.cpu_crop, args.precision, weights_name, acc, nb)
        except Exception as exc:  # the baseline is reported, never fatal
            cpu = {"value": None, "unit": "voxels/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {exc}"}

    out = {
        "metric": METRIC,
        "value": value,
        "unit": "voxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        # the same volume with the background skip disabled: every one of the `windows` runs the network
        "value_dense": (vox / elapsed_dense) if elapsed_dense else None,
        "windows_run_dense": (stats_dense.get("n_windows", 0) - stats_dense.get("n_skipped", 0)) if (elapsed_dense and world == 1) else None,
        "ms_per_step_dense": 1e3 * elapsed_dense if elapsed_dense else None,
        # one pass in the other 16-bit format (same volume, same skip list)
        "other_format": {"dtype": {"bf16": "bf16", "fp16": "f16"}[alt_prec], "ms_per_step": 1e3 * elapsed_alt,
                         "value": vox / elapsed_alt} if elapsed_alt else None,
        # the same `steps` passes with every launch bracketed by HIP events (what `kernels` / `roofline_timed_region` saw)
        "ms_per_step_profiled": 1e3 * elapsed_prof / args.steps if elapsed_prof else None,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.precision],
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {Z}x{Y}x{X} (Z,Y,X) uint16 synthetic brain"
                        + (f" zero-padded to {Zp}x{Yp}x{Xp}" if shape != tuple(stack) else "")
                        + f", windows {roi[0]}x{roi[1]}x{roi[2]}, overlap 0.5, "
                        + ("TTA: the reference's 13 passes as 3 distinct passes (plain, flip Z, flip Y) weighted 5:4:4" if tta else "1 pass (no TTA)")
                        + f", {weights_name} BasicUNet(32,32,64,128,256,32) weights, {args.precision} operands / fp32 accumulate"
                        + (", dense (no background)" if args.dense else ", ellipsoid brain (background skipped)"),
            "volume_zyx": [Z, Y, X], "padded_zyx": list(shape), "roi": list(roi), "overlap": 0.5, "tta": bool(tta),
            "passes_per_step": n_passes, "pass_weights": [r for _, r in schedule],
            "windows": n_windows, "windows_skipped": n_skipped, "per_rank_windows": per_rank,
            # per rank: ms of its windows / the seam exchange / the finalize of its planes in one synchronised step, and of the
            # balanced plan (slab generation + window maxima + their all_gather) at start-up
            "per_rank_phases": per_rank_phases,
            "skipped_fraction": (n_skipped / n_windows) if n_windows else None,
            "patch_voxels_per_s": tile_vox * n_active * n_passes / (elapsed / args.steps),
            "timed_region": "uint16 volume in HBM -> uint8 eroded mask in HBM" + (" (Z-slabs resident on their ranks)" if world > 1 else ""),
            "lanes": int(os.environ.get("DLV_LANES", "3")), "diag": args.diag or None,
            "mask_voxels": mask_voxels, "mask_checksum": mask_checksum,
            "parallelism": f"windows sharded in {world} contiguous Z-slabs, seam exchange p2p" if world > 1 else "1 GPU",
            "dist_backend": (dist.get_backend() + (" (forced at world size 1: DLV_BENCH_FORCE_DIST)" if world == 1 else "")) if dist_mode else None,
        },
        # `roofline` describes the dominant kernel itself: measured with HIP events in one extra step on a single lane
        # (DLV_LANES=1 reproduces it over the timed region; profiles/*_1lane_kernel_stats.csv is rocprofv3's view).
        # In the timed region the lanes (default 3) overlap, which stretches every kernel's event-to-event time: that view is kept
        # as `roofline_timed_region`.
        "roofline": roofline_isolated if roofline_isolated is not None else roofline,
        "roofline_timed_region": roofline if roofline_isolated is not None else None,
        "cpu_baseline": cpu,
        "kernels": kernels,
        "extras": extras,
        # the step boundary: wall clock file -> file of the two steps, with the H2D / D2H transfers timed separately
        "step_walls": walls,
    }
    print(json.dumps(out))
    if dist_mode:
        dist.barrier()
        dist.destroy_process_group()
    if cpu and cpu.get("agreement") and not cpu["agreement"]["ok"]:
        raise SystemExit(f"bench.py: the HIP path and the CPU oracle disagree on the baseline crop: {cpu['agreement']}")


if __name__ == "__main__":
    main()
