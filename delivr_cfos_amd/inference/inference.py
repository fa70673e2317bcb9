"""Mirror of the reference's ``inference/inference.py`` (run_inference :113-332, create_nifti_seg
:31-95) on top of libdelivr_hip: same call signature, same files in and out.

in : <niftis[0]>  NPY v1, 128-byte header, '<u2', C-order (1,1,Zp,Yp,Xp)          (:234)
out: <output_folder>/<comment>/binary_segmentations/binaries.npy  '|u1' (Z,Y,X)   (:312)
     optional .../binary_segmentations/network_output.npy '<f4' (Z,Y,X)           (:315-318)
"""
from __future__ import annotations

import datetime
import os
from typing import Optional, Sequence

import numpy as np

from ..hostio import save_npy
from ..hostlogic import arrayterator_zblock, padded_shape, pass_schedule
from .._lib import DLV_ERANGE, DelivrHipError
from ..range_guard import next_shifts, run_with_range_recovery
from ..model import HipBasicUNet
from .sliding_window_inferer import SlidingWindowInferer


# volumes read ahead for the NEXT call of run_inference (pipelined runs over several brains): key -> (thread, box)
_prefetched = {}


def _prefetch_key(path):
    st = os.stat(path)
    return (os.path.abspath(path), int(st.st_size), int(st.st_mtime_ns))


def create_nifti_seg(threshold, model_output, output_file, network_output_file, dataset, original_stack_shape,
                     count_map=None, engine=None, erode_iters: int = 30, mask_out=None, defer_write: bool = False):
    """sigmoid >= threshold, eroded re-mask, crop to the original stack, write binaries.npy
    (reference :31-95).  ``model_output``: (1,1,Zp,Yp,Xp) or (Zp,Yp,Xp) fp32 tensor in HBM holding the
    blended logits (sum; pass ``count_map`` to divide, or the mean already); ``dataset``: the uint16
    volume in HBM; ``mask_out``: a uint8 (Z,Y,X) tensor in HBM to hold the mask (allocated ahead of the passes);
    ``defer_write``: return as soon as the mask exists in HBM - the file(s) stream out on the engine's background worker
    (hostio.wait_deferred() joins; a file appears under its name only when complete)."""
    Z, Y, X = (int(v) for v in original_stack_shape[-3:])
    if count_map is None and float(threshold) != 0.5:
        # sigmoid(sum of the window logits) >= t equals the reference's sigmoid(sum / count) >= t (:295) only at t = 0.5
        raise ValueError(f"threshold {threshold} needs the count map (or mean logits with threshold 0.5 semantics): "
                         "without it only the sign of the blended sum is known")
    acc = model_output[0, 0] if model_output.dim() == 5 else model_output
    raw = dataset[0, 0] if dataset.dim() == 5 else dataset
    cnt = None if count_map is None else (count_map[0, 0] if count_map.dim() == 5 else count_map)
    zb = arrayterator_zblock((Z, Y, X))
    res = engine.finalize(acc, cnt, raw, (Z, Y, X), float(threshold), erode_iters, 0 if zb >= Z else zb,
                          want_prob=network_output_file is not None, out=mask_out)
    mask, prob = res if network_output_file is not None else (res, None)
    engine.sync()
    # the reference creates binaries.npy with open_memmap (:312) and fills it block by block; here numpy writes the same
    # header and the payload streams out of HBM through pinned staging with parallel writers (hostio.py)
    if defer_write:
        from ..hostio import submit_deferred

        submit_deferred(engine, save_npy, engine, mask, output_file, np.uint8, "d2h_mask", True, True)  # (partial=True, synced=True)
        if network_output_file is not None:
            submit_deferred(engine, save_npy, engine, prob, network_output_file, np.float32, "d2h_prob", True, True)
        return mask
    save_npy(engine, mask, output_file, np.uint8, what="d2h_mask")
    if network_output_file is not None:
        save_npy(engine, prob, network_output_file, np.float32, what="d2h_prob")
    return mask


def run_inference(
    niftis,
    output_folder,
    stack_shape,
    comment: str = "none",
    model_weights: str = "weights/inference_weights.tar",
    tta: bool = False,
    threshold: float = 0.5,
    cuda_devices: str = "0,1",
    crop_size: Sequence[int] = (64, 64, 32),
    workers: int = 0,
    sw_batch_size: int = 100,
    overlap: float = 0.5,
    verbosity: bool = True,
    load_all_ram: bool = False,
    settings: Optional[dict] = None,
    precision: Optional[str] = None,
    state_dict=None,
    prefetch: Optional[str] = None,
    defer_write: bool = False,
):
    """Same parameters as the reference (:113-129) plus ``precision`` ("fp16" default / "bf16" / "bf16_all" / "fp32",
    also settings["mi355x"]["precision"]) and ``state_dict`` (use instead of reading
    ``model_weights``).  Returns "<abs output_folder>/<comment>".

    Pipelining over several brains (python -m delivr_cfos_amd does it; single device, resident volumes): ``prefetch`` = the
    masked_nifti.npy of the NEXT brain - it is read into HBM by a side thread while this brain's passes run, and the next call
    finds it there; ``defer_write`` = return when the mask exists in HBM, binaries.npy streams out in the background
    (hostio.wait_deferred() before anything reads the file).  Per brain the step then costs its passes."""
    import time

    import torch

    marks = [("start", time.perf_counter())]  # wall-clock marks of this call -> run_inference.last_timings (bench.py: step2_wall_s)

    def mark(name):
        marks.append((name, time.perf_counter()))

    print(f"{datetime.datetime.now()} : Setting up inference parameters ")
    if settings is not None:
        wd = settings["blob_detection"]["window_dimensions"]
        crop_size = (wd["window_dim_0"], wd["window_dim_1"], wd["window_dim_2"])
        if precision is None:
            precision = settings.get("mi355x", {}).get("precision")
    precision = precision or "fp16"
    crop_size = tuple(int(c) for c in crop_size)
    print("using crop size:  ", crop_size)
    if any(c < 16 for c in crop_size) or (crop_size[0] // 16) * (crop_size[1] // 16) * (crop_size[2] // 16) < 2:
        # the reference takes any window (inference/inference.py:162-168 -> SlidingWindowInferer), and so does the HIP U-Net: a
        # level with an odd size is pooled (last plane dropped) and its up-sampled partner replicate-padded like MONAI's UpCat.
        # Where torch raises, this raises: below 16 the fourth pooling has nothing left ("Output size is too small"), and a level 4
        # of ONE voxel has no InstanceNorm statistics (ValueError "Expected more than 1 spatial element", e.g. 16 x 16 x 16).
        raise ValueError(f"window_dimensions {crop_size}: every dimension must be at least 16 and level 4 of the U-Net (each dimension "
                         "// 16) must hold more than one voxel (torch's InstanceNorm3d raises on a single spatial element; the reference "
                         "builds its windows from window_dim_0..2, inference/inference.py:162-168)")
    if not torch.cuda.is_available():
        raise RuntimeError("run_inference needs an MI355X: the HIP path has no CPU fallback")
    # one process per GPU (torch.distributed.run): the window list is sharded over the ranks, see parallel.py.
    # Single process: the first entry of cuda_devices (the reference's DataParallel spans all of them).
    import torch.distributed as dist

    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if sharded else 0
    world = dist.get_world_size() if sharded else 1
    if sharded:
        device_index = int(os.environ.get("LOCAL_RANK", rank))
    else:
        device_index = int(str(cuda_devices).split(",")[0]) if str(cuda_devices).strip() else 0
    if device_index >= torch.cuda.device_count():
        device_index = 0

    # ~~<< M O D E L >>~~  (reference :190-222)
    # the process-wide engine of the device: its context, workspaces and pinned staging survive between brains and steps
    model = HipBasicUNet(device=device_index, precision=precision, shared=True)
    eng = model.engine
    if rank == 0:
        if state_dict is None:
            checkpoint = torch.load(os.path.abspath(model_weights), map_location="cpu", weights_only=False)
        else:
            checkpoint = state_dict
        model.load_state_dict(checkpoint)
    if sharded:
        from ..parallel import broadcast_weights

        broadcast_weights(eng, dist, rank)  # ONE broadcast instead of DataParallel's per-forward replicate
    model.eval()
    mark("model")
    # settings["mi355x"]["blend"] = "gaussian" makes the mode argument take effect (option; the reference's own call
    # passes mode="gaussian" too, but its inferer blends with constant weights: SURVEY D2)
    gaussian = bool(settings and settings.get("mi355x", {}).get("blend", "constant") == "gaussian")
    inferer = SlidingWindowInferer(roi_size=crop_size, sw_batch_size=sw_batch_size, sw_device=eng.device,
                                   device=eng.device, overlap=overlap, mode="gaussian", padding_mode="replicate",
                                   honour_mode=gaussian)

    # ### DATA PREP ###  (reference :225-251)
    print(f"{datetime.datetime.now()} : Loading Data")
    stack_shape = tuple(int(v) for v in stack_shape)
    pad = (1, 1) + padded_shape(stack_shape[2:], crop_size)
    dataset_host = np.memmap(niftis[0], dtype=np.uint16, mode="r", shape=pad, offset=128)
    if rank == 0:
        os.makedirs(os.path.join(output_folder, comment), exist_ok=True)
    save_activated = bool(settings and settings.get("FLAGS", {}).get("SAVE_ACTIVATED_OUTPUT"))
    # (one process per GPU: the buffers below are allocated per slab further down)
    # The count map is needed whenever the MEAN logit matters: for network_output.npy and for any threshold other than
    # 0.5 (reference: mean = sum / count before the sigmoid, inference.py:295); at 0.5 the sign of the sum decides.
    need_count = save_activated or float(threshold) != 0.5
    cm_dtype = torch.float32 if gaussian else torch.uint8
    # Does the volume fit this GPU?  The reference streams through memmaps and has no size limit (:240-247, :285-299); the
    # resident path needs volume + fp32 sums (+ count map) + the finalize maps in HBM.  Beyond the budget
    # (settings["mi355x"]["hbm_budget_gb"], default: what the device reports free) the slabs of the shard plan run one
    # after another on this device (streaming.py); MemoryError when not even that fits.
    stream_plan = None
    if not sharded:
        from ..streaming import forward_workspace_bytes, hbm_budget_bytes, inference_bytes_per_voxel, plan_slabs

        budget = hbm_budget_bytes(eng, settings)
        bpv = inference_bytes_per_voxel(need_count, gaussian, save_activated)
        # (the pass's workspaces: already in the shared engine when it has served a brain with these windows and this format - then
        # the device's free memory, which the default budget is derived from, no longer contains them.  An explicit
        # hbm_budget_gb is a TOTAL: the workspaces always count against it)
        ws_key = (tuple(crop_size), precision)
        explicit_budget = bool((settings or {}).get("mi355x", {}).get("hbm_budget_gb"))
        held = not explicit_budget and ws_key in eng.__dict__.setdefault("_ws_reserved", set())
        fixed = 0 if held else forward_workspace_bytes(crop_size, precision)
        pad_vox = int(pad[2]) * int(pad[3]) * int(pad[4])
        forced = int((settings or {}).get("mi355x", {}).get("stream_slabs", 0) or 0)  # explicit: exactly this many slabs
        if forced > 0:
            from ..parallel import plan_from_params

            p_all = eng.make_sw_params(pad[2:], crop_size, overlap, None, 0, precision)
            stream_plan, n_slabs = plan_from_params(p_all, forced, None), forced
            print(f"settings['mi355x']['stream_slabs'] = {forced}: streaming {n_slabs} Z-slabs through the device")
        elif pad_vox * bpv + fixed > budget:
            p_all = eng.make_sw_params(pad[2:], crop_size, overlap, None, 0, precision)
            stream_plan, n_slabs = plan_slabs(eng, p_all, int(stack_shape[2]), int(pad[3]) * int(pad[4]), bpv, fixed, budget,
                                              arrayterator_zblock(tuple(stack_shape[2:])))
            print(f"volume of {pad_vox * bpv / 2**30:.1f} GiB (+ {fixed / 2**30:.1f} GiB workspace) exceeds the HBM budget of "
                  f"{budget / 2**30:.1f} GiB: streaming {n_slabs} Z-slabs through the device")
    resident = not sharded and stream_plan is None
    output_image = count_map = mask_buf = None
    prealloc_t = {}
    if resident:
        # This step needs ~65 GB of device memory for a 1024 x 2048 x 2048 brain - sums, the pass's activation workspaces, the
        # finalize maps.  Memory nobody has used since the driver's last clear costs nothing (16 GiB: 0.3 ms); memory that was used
        # before - by this process or an earlier one - is cleared when it is handed out again, at 28-140 ms per GB
        # (profiles/r06r_alloc_probe2.json, r06y_*: 1.8 s for this step's buffers).  A second thread allocates them while
        # this one reads the volume (preads -> pinned staging -> HBM, hostio.py), and nothing is handed back between brains: the
        # shared engine keeps its workspaces, torch's caching allocator its blocks.
        import threading

        side, side_err = {}, []

        def preallocate():
            try:
                with torch.cuda.device(eng.device):
                    t_a = time.perf_counter()
                    side["acc"] = torch.zeros(pad[2:], dtype=torch.float32, device=eng.device)
                    if need_count:
                        side["cnt"] = torch.zeros(pad[2:], dtype=cm_dtype, device=eng.device)
                    side["mask"] = torch.empty(tuple(stack_shape[2:]), dtype=torch.uint8, device=eng.device)
                    torch.cuda.current_stream(eng.device).synchronize()
                    t_b = time.perf_counter()
                    eng.reserve(eng.make_sw_params(pad[2:], crop_size, overlap, None, 0, precision), stack_shape[2:])
                    eng.__dict__.setdefault("_ws_reserved", set()).add((tuple(crop_size), precision))
                    torch.cuda.current_stream(eng.device).synchronize()
                    side["t"] = {"prealloc_torch_s": t_b - t_a, "prealloc_reserve_s": time.perf_counter() - t_b}
            except Exception as exc:  # re-raised on the main thread
                side_err.append(exc)

        th = threading.Thread(target=preallocate, name="dlv-prealloc")
        th.start()
        try:
            ahead = _prefetched.pop(_prefetch_key(niftis[0]), None)
            dataset = None
            if ahead is not None:  # the previous call read this volume while its passes ran
                ahead[0].join()
                if "err" not in ahead[1] and int(ahead[1]["vol"].numel()) == int(pad[2]) * int(pad[3]) * int(pad[4]):
                    dataset = ahead[1]["vol"].reshape(tuple(pad[2:]))
            if dataset is None:
                dataset = eng.upload_volume(dataset_host[0, 0])  # parallel preads -> pinned staging ring -> HBM (hostio.py)
        finally:
            th.join()
        if side_err:
            raise side_err[0]
        _prefetched.clear()  # (a volume read ahead for a call that never came is dropped)
        if prefetch is not None and os.path.isfile(prefetch):
            nxt_shape = (os.path.getsize(prefetch) - 128) // 2
            box = {}

            def read_ahead(path=prefetch, n=nxt_shape):
                try:
                    host = np.memmap(path, dtype=np.uint16, mode="r", shape=(n,), offset=128)
                    box["vol"] = eng.upload_volume(host.reshape(1, 1, n))[0, 0].reshape(-1)  # (flat: the next call knows the shape)
                except Exception as exc:  # the next call simply reads the file itself
                    box["err"] = exc

            ra = threading.Thread(target=read_ahead, name="dlv-prefetch")
            _prefetched[_prefetch_key(prefetch)] = (ra, box)
            ra.start()
        output_image, count_map, mask_buf = side["acc"], side.get("cnt"), side["mask"]
        prealloc_t = side.get("t", {})
        mark("upload+alloc")
    if need_count and cm_dtype == torch.uint8:
        # uint8 like the reference's LOAD_ALL_RAM map (:241): refuse geometries whose multiplicity cannot be held
        from ..hostlogic import max_window_multiplicity

        p_chk = eng.make_sw_params(pad[2:], crop_size, overlap, None, 0, precision)
        mult = max_window_multiplicity(eng.window_starts(p_chk), [int(p_chk.roi[k]) for k in range(3)]) * (13 if tta else 1)
        if mult > 255:
            raise NotImplementedError(f"up to {mult} (window, pass) contributions per voxel do not fit the uint8 count map "
                                      "(overlap too large for this threshold / SAVE_ACTIVATED_OUTPUT setting)")
    print("output_image shape", tuple(pad[2:]))

    testing_session_path = os.path.abspath(output_folder + "/" + comment)
    binaries_path = testing_session_path + "/binary_segmentations/"
    output_file = os.path.join(binaries_path, "binaries.npy")
    network_output_file = os.path.join(binaries_path, "network_output.npy") if save_activated else None

    # inference passes (reference :261-279)
    print(f"{datetime.datetime.now()} : Starting inference")
    if stream_plan is not None:
        from ..streaming import run_inference_streamed

        os.makedirs(binaries_path, exist_ok=True)
        Z, Y, X = stack_shape[2:]
        out_mask = np.lib.format.open_memmap(output_file, mode="w+", dtype=np.uint8, shape=(Z, Y, X))
        out_prob = np.lib.format.open_memmap(network_output_file, mode="w+", dtype=np.float32, shape=(Z, Y, X)) if save_activated else None
        args = (eng, dataset_host[0, 0], tuple(pad[2:]), (Z, Y, X), crop_size, overlap, bool(tta))
        kw = dict(threshold=threshold, need_count=need_count, gaussian=gaussian, plan=stream_plan, out_mask=out_mask, out_prob=out_prob,
                  verbose=bool(verbosity))
        # range guard (DLV_ERANGE): the failed run's slabs are released before the repeat starts (the retry runs outside the
        # except block, whose traceback would keep the streamed slabs of the failed run alive in HBM)
        def reset_streamed():
            import gc

            gc.collect()
            torch.cuda.empty_cache()

        precision = run_with_range_recovery(eng, precision, lambda prec: run_inference_streamed(*args, precision=prec, **kw), reset_streamed)
        print(f"{datetime.datetime.now()} : Creating binarized blob output")
        out_mask.flush()
        if out_prob is not None:
            out_prob.flush()
        del out_mask, out_prob
    elif not sharded:
        def run_passes(prec):
            model.precision = prec
            for flip_dim, repeat in pass_schedule(bool(tta)):
                kw = dict(output_image=output_image, count_map=count_map, repeat=repeat)
                if flip_dim is not None:
                    kw.update(tta=True, flip_dim=flip_dim)
                inferer(dataset, model, **kw)
            eng.sync()

        def reset_sums():
            output_image.zero_()
            if count_map is not None:
                count_map.zero_()

        # range guard (DLV_ERANGE): this checkpoint drives a raw activation beyond fp16's 65504 - the library reports it instead
        # of painting a garbage mask; the block is rescaled (range_guard.py: fp16 keeps its 11 bits), bf16 is the last resort
        precision = run_with_range_recovery(eng, model.precision, run_passes, reset_sums)
        mark("passes")
        # block-wise averaging + binarisation (reference :282-329) happen in one fused finalize pass
        print(f"{datetime.datetime.now()} : Creating binarized blob output")
        os.makedirs(binaries_path, exist_ok=True)
        if save_activated:
            os.makedirs(testing_session_path + "/network_outputs/", exist_ok=True)
        create_nifti_seg(threshold=threshold, model_output=output_image, output_file=output_file,
                         network_output_file=network_output_file, dataset=dataset, original_stack_shape=stack_shape,
                         count_map=count_map, engine=eng, mask_out=mask_buf, defer_write=defer_write)
    else:
        from ..parallel import balanced_plan, exchange_seams, finalize_owned, gather_slabs

        # Slab-resident: every rank reads, uploads and accumulates only the planes of ITS Z-slab (its windows' planes and
        # the erosion margin of the planes it owns); the plan is balanced by the windows that run the network.
        Z, Y, X = stack_shape[2:]
        p_all = eng.make_sw_params(pad[2:], crop_size, overlap, None, 0, precision)
        plan, slo, shi, dataset = balanced_plan(
            eng, p_all, lambda lo, hi: eng.upload_volume(dataset_host[0, 0], lo, hi), world, rank, dist, Z, 30,
            arrayterator_zblock((Z, Y, X)))
        output_image = torch.zeros((shi - slo,) + tuple(pad[3:]), dtype=torch.float32, device=eng.device)
        count_map = torch.zeros((shi - slo,) + tuple(pad[3:]), dtype=cm_dtype, device=eng.device) if need_count else None
        def run_passes(prec):
            """-> 0, or 1 when this rank's passes left the 16-bit format's range (every rank must learn of it)"""
            try:
                for flip_dim, repeat in pass_schedule(bool(tta)):
                    wb, we = plan.win_ranges[rank]
                    if we > wb:
                        if gaussian:
                            eng.sw_infer(eng.make_sw_params(pad[2:], crop_size, overlap, flip_dim, 0, prec, win_range=(wb, we),
                                                            slab=(slo, shi - slo), repeat=repeat, blend="gaussian", wsum=count_map),
                                         dataset, output_image)
                        else:
                            eng.sw_infer(eng.make_sw_params(pad[2:], crop_size, overlap, flip_dim, 0, prec, win_range=(wb, we),
                                                            slab=(slo, shi - slo), repeat=repeat), dataset, output_image, count_map)
                eng.sync()
            except DelivrHipError as e:
                if e.code != DLV_ERANGE or prec not in ("fp16", "bf16"):
                    raise
                print(f"WARNING (rank {rank}): {e}")
                return 1
            return 0

        # Every rank must end in the same format with the same block shifts (one mask): after each attempt the ranks exchange
        # what their guards saw - the named layer (max) and the per-block peaks (max) - and derive the SAME remedy from it
        gdev = "cpu" if dist.get_backend() == "gloo" else eng.device
        for attempt in range(6):
            bad = run_passes(precision)
            layer, peaks = eng.range_report() if bad else (-1, [0.0] * 18)
            rep = torch.tensor([float(bad), float(layer)] + list(peaks), dtype=torch.float32, device=gdev)
            dist.all_reduce(rep, op=dist.ReduceOp.MAX)
            if not rep[0].item():
                break
            shifts = next_shifts(int(rep[1].item()), [float(v) for v in rep[2:].tolist()], eng.conv_shifts()) if attempt < 4 else None
            if shifts is None or precision not in ("fp16", "bf16"):
                if precision not in ("fp16", "bf16"):
                    raise RuntimeError("unreachable: bf16_all range errors are raised")
                if rank == 0:
                    print("WARNING: repeating the inference passes with bf16 operands at every level on every rank")
                precision = "bf16_all"
            else:
                for p, k in sorted(shifts.items()):
                    if rank == 0:
                        print(f"WARNING: conv block {p}: storing its raw output scaled by 2^-{k} on every rank and repeating the passes in {precision}")
                    eng.set_conv_shift(p, k)
            output_image.zero_()
            if count_map is not None:
                count_map.zero_()
        else:  # (not reachable: the bf16 attempt either succeeds or raises in run_passes)
            raise RuntimeError("range guard: the passes did not come to an end in six attempts")
        exchange_seams(output_image, plan, rank, dist, z0=slo)
        if count_map is not None:
            exchange_seams(count_map, plan, rank, dist, z0=slo)
        slab, prob, _ = finalize_owned(eng, plan, rank, output_image, count_map, dataset, (Z, Y, X), threshold, 30,
                                       want_prob=save_activated, z0=slo)
        eng.sync()
        if slab is None:
            slab = torch.empty((0, Y, X), dtype=torch.uint8, device=eng.device)
        full = torch.empty((Z, Y, X), dtype=torch.uint8, device=eng.device) if rank == 0 else None
        gather_slabs(slab, plan, rank, dist, out=full)
        if save_activated:
            if prob is None:
                prob = torch.empty((0, Y, X), dtype=torch.float32, device=eng.device)
            pfull = torch.empty((Z, Y, X), dtype=torch.float32, device=eng.device) if rank == 0 else None
            gather_slabs(prob, plan, rank, dist, out=pfull)
        if rank == 0:
            print(f"{datetime.datetime.now()} : Creating binarized blob output")
            os.makedirs(binaries_path, exist_ok=True)
            save_npy(eng, full, output_file, np.uint8, what="d2h_mask")
            if save_activated:
                os.makedirs(testing_session_path + "/network_outputs/", exist_ok=True)
                save_npy(eng, pfull, network_output_file, np.float32, what="d2h_prob")
        dist.barrier()
    mark("finalize+write")
    run_inference.last_timings = {"total_s": marks[-1][1] - marks[0][1],
                                  **{f"{b[0]}_s": b[1] - a[1] for a, b in zip(marks, marks[1:])}, **prealloc_t}
    print(f"{datetime.datetime.now()} : Blob Detection finished")
    return testing_session_path
