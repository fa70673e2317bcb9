"""Mirror of the resampler + padded-npy writer of the reference's
``downsample/downsample_and_mask.py`` on top of libdelivr_hip.

Accelerated (device) pieces:
    downsample_volume  <- downsample_zplanes' skimage block mean              (:32-47, ratios :161-163)
    upsample_mask      <- scipy.ndimage.zoom(order=2, prefilter=False) -> u8  (:285-299)
    mask_and_pad       <- img *= mask_us[i]; masked_nii[0,0,i,:Y,:X] = img    (:383-417)
    write_masked_nifti_npy: the (1,1,Zp,Yp,Xp) uint16 .npy with a 128-byte header the inference step memmaps
    downsample_mask    <- the step itself, both branches: ilastik's mask (read where the reference reads it, binarised at
                          125, :268-269) or the simple threshold
Out of scope: the ilastik / TeraConverter subprocesses themselves.  Raw planes are read by the native
TIFF reader (csrc/tiffio.hip).
"""
from __future__ import annotations

import glob
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from ..hostlogic import downsample_ratios, padded_shape

def read_tiff_plane(path: str) -> np.ndarray:
    """One z-plane as uint16 (8-bit planes are widened): classic TIFF or BigTIFF, strips or tiles, uncompressed / LZW /
    deflate (with or without the horizontal predictor), decoded by the native reader in libdelivr_hip.so
    (csrc/tiffio.hip; host code, no GPU needed).  Anything else (JPEG, multi-channel, float samples) raises
    NotImplementedError."""
    import ctypes as C

    from .. import _lib

    lib = _lib.load()
    h, w, bits = C.c_int(), C.c_int(), C.c_int()
    rc = lib.dlv_tiff_plane_size(path.encode(), C.byref(h), C.byref(w), C.byref(bits))
    if rc != 0:
        msg = lib.dlv_tiff_last_error().decode()
        raise (FileNotFoundError if rc == -1 else NotImplementedError)(msg)
    out = np.empty((h.value, w.value), dtype=np.uint16)
    if lib.dlv_tiff_read_plane_u16(path.encode(), out.ctypes.data_as(C.c_void_p), h.value, w.value) != 0:
        raise NotImplementedError(lib.dlv_tiff_last_error().decode())
    return out


def load_stack_to_device(engine, planes: Sequence[str], out=None, n_threads: int = 0):
    """All planes of a raw stack -> uint16 (Z,Y,X) tensor in HBM (reference: one cv2.imread per plane on one core,
    :396-404).  Planes are decoded by a pool of host threads into pinned buffers and copied while the next chunk
    decodes (dlv_tiff_stack_to_device).  ``out``: an existing uint16 tensor whose leading (Z,Y,X) corner is filled
    (e.g. the zero-initialised padded network input); its strides are honoured."""
    import ctypes as C

    torch = engine.torch
    lib = engine.lib
    h, w, bits = C.c_int(), C.c_int(), C.c_int()
    if lib.dlv_tiff_plane_size(planes[0].encode(), C.byref(h), C.byref(w), C.byref(bits)) != 0:
        raise NotImplementedError(lib.dlv_tiff_last_error().decode())
    Z, Y, X = len(planes), h.value, w.value
    if out is None:
        out = torch.empty((Z, Y, X), dtype=torch.uint16, device=engine.device)
    if out.dtype != torch.uint16 or out.dim() != 3 or out.shape[0] < Z or out.shape[1] < Y or out.shape[2] < X or out.stride(2) != 1:
        raise ValueError("out: uint16 (>=Z, >=Y, >=X) tensor with unit x stride")
    arr = (C.c_char_p * Z)(*[p.encode() for p in planes])
    engine._enter()
    engine._check(lib.dlv_tiff_stack_to_device(engine.ctx, arr, Z, Y, X, C.c_void_p(out.data_ptr()), int(out.stride(0)),
                                               int(out.stride(1)), int(n_threads)))
    engine._leave()
    return out[:Z, :Y, :X]


def get_real_size(raw_folder: str) -> Tuple[int, int, int]:
    """(z, y, x) of a folder of .tif z-planes (reference :25-30: counts *.tif, reads the first plane's
    shape) - here from the TIFF header alone, no decode.  A folder holding ``stack.npy`` is accepted too."""
    names = sorted(i for i in os.listdir(raw_folder) if ".tif" in i)
    if names:
        import ctypes as C

        from .. import _lib

        lib = _lib.load()
        h, w, bits = C.c_int(), C.c_int(), C.c_int()
        rc = lib.dlv_tiff_plane_size(os.path.join(raw_folder, names[0]).encode(), C.byref(h), C.byref(w), C.byref(bits))
        if rc != 0:
            raise (FileNotFoundError if rc == -1 else NotImplementedError)(lib.dlv_tiff_last_error().decode())
        return (len(names), h.value, w.value)
    npy = os.path.join(raw_folder, "stack.npy")
    if os.path.isfile(npy):
        return tuple(int(v) for v in np.load(npy, mmap_mode="r").shape[-3:])
    raise FileNotFoundError(f"{raw_folder}: no .tif planes and no stack.npy")


# ---- device resamplers ---------------------------------------------------------------------------------
def downsample_volume(engine, vol_dev, factors_zyx: Sequence[int], drop_last_chunk: bool = True):
    """uint16 (Z,Y,X) in HBM -> block-mean down-sampled uint16.  ``drop_last_chunk`` reproduces the
    reference's zip(z_series, z_series[1:]) (:166,187), which never processes the final z-chunk."""
    fz = int(factors_zyx[0])
    Z = int(vol_dev.shape[0])
    if drop_last_chunk:
        nchunks = len(range(0, Z, fz)) - 1
        if nchunks <= 0:
            raise ValueError("fewer than two z-chunks: the reference would produce an empty stack")
        vol_dev = vol_dev[: nchunks * fz].contiguous()
    return engine.block_mean_u16(vol_dev, factors_zyx)


def upsample_mask(engine, mask_dev, out_shape_zyx: Sequence[int]):
    """uint8 mask (downsampled grid) -> uint8 mask at the raw stack's shape: spline-2 zoom, bit-exact with
    scipy.ndimage.zoom(mask, out/in, output=uint8, order=2, prefilter=False) (reference :299)."""
    return engine.zoom_spline2_u8(mask_dev, out_shape_zyx)


def mask_and_pad(engine, raw_dev, mask_dev, crop_size: Sequence[int], threshold: Optional[int] = None):
    """raw (Z,Y,X) uint16 * mask (uint8) -> zero-padded (Zp,Yp,Xp) uint16 with Zp.. = ceil(dim/crop)*crop
    (reference :390-417).  mask_dev None + threshold: the simple-threshold branch (img[img < thr] = 0)."""
    pad = padded_shape(tuple(int(v) for v in raw_dev.shape), crop_size)
    return engine.mask_pad_u16(raw_dev, mask_dev, pad, threshold or 0)


def write_masked_nifti_npy(path: str, padded_dev, engine=None) -> None:
    """(Zp,Yp,Xp) uint16 tensor -> <path> as NPY v1 (1,1,Zp,Yp,Xp) '<u2' with the 128-byte header the
    inference step skips with offset=128 (inference/inference.py:234).  With an engine the payload streams out of HBM through
    pinned staging (hostio.py; the masked-out background stays a hole of the file), else through one host copy."""
    shape = (1, 1) + tuple(int(v) for v in padded_dev.shape)
    if engine is not None and padded_dev.is_contiguous():
        from .. import hostio

        if hostio.create_npy(path, np.uint16, shape) != 128:
            raise RuntimeError("npy header is not the 128 bytes the pipeline expects")
        hostio.download(engine, padded_dev, path, offset=128, what="d2h_volume", sparse=True)
        return
    out = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint16, shape=shape)
    if out.offset != 128:
        raise RuntimeError(f"npy header is {out.offset} bytes, the pipeline expects 128")
    out[0, 0] = padded_dev.cpu().numpy()
    out.flush()


def find_ilastik_mask_planes(results_folder: str):
    """Where the reference picks up ilastik's output (ilastik_ventricles, :85-93): the probability planes
    ``<results>/ventricles_zplanes/*.tif`` (sorted), which it also concatenates into ``<downsampled_name>_mask.tif``.
    -> sorted plane paths, or [] when ilastik has not run."""
    return sorted(glob.glob(os.path.join(results_folder, "ventricles_zplanes", "*.tif")))


def load_ilastik_mask(results_folder: str) -> np.ndarray:
    """ilastik's ventricle / outside-the-brain probabilities (0..255) on the down-sampled grid -> the binary uint8 mask the
    reference makes of them: ``mask[mask < 125] = 0; mask[mask >= 125] = 1`` (:268-269)."""
    planes = find_ilastik_mask_planes(results_folder)
    if not planes:
        multi = sorted(glob.glob(os.path.join(results_folder, "*_mask.tif")))
        raise FileNotFoundError(
            f"mask_with_Ilastik=true but no ilastik output under {os.path.join(results_folder, 'ventricles_zplanes')}: this package does "
            "not shell out to the ilastik binary (reference :71-93) - run it on the down-sampled stack first (its headless "
            "project writes one probability plane per z there), or set mask_with_Ilastik=false"
            + (f" ({multi[0]} exists, but multi-page TIFFs are not read: keep the per-plane files)" if multi else ""))
    prob = np.stack([read_tiff_plane(p) for p in planes])
    return (prob >= 125).astype(np.uint8)


def downsample_mask(settings: dict, brain: str, engine=None):
    """Step 1 of the pipeline for one brain (reference :139-427):
    raw planes -> block-mean stack (saved as downsampled_stack.npy) -> mask of the raw planes -> zero-padded
    masked_niftis/masked_nifti.npy.  The mask is either (``mask_with_Ilastik: true``, the reference's default) ilastik's
    output on the down-sampled grid, binarised at 125 (:268-269) and brought to the raw stack's shape with the spline-2 zoom
    (:285-299), or (false) the simple threshold on the raw intensities (:404-408).  ilastik itself is an external binary:
    its output is CONSUMED where the reference reads it, FileNotFoundError when it is absent."""
    from ..engine import HipEngine

    md = settings["mask_detection"]
    raw_location = os.path.join(settings["raw_location"], brain)
    planes = sorted(glob.glob(raw_location + "/*.tif"))
    if not planes:
        raise FileNotFoundError(f"no .tif planes under {raw_location}")
    results = os.path.join(md["output_location"], brain)
    use_ilastik = bool(md.get("mask_with_Ilastik", True))
    mask_ds = load_ilastik_mask(results) if use_ilastik else None  # (before any device work: fails early when absent)
    own = engine is None
    eng = engine or HipEngine(0)
    try:
        raw_dev = load_stack_to_device(eng, planes)   # parallel decode -> pinned staging -> HBM
        ratios = downsample_ratios(md["downsample_steps"])
        os.makedirs(os.path.join(results, "masked_niftis"), exist_ok=True)
        ds = downsample_volume(eng, raw_dev, ratios)
        ds_host = ds.cpu().numpy()
        np.save(os.path.join(results, "downsampled_stack.npy"), ds_host)
        wd = settings["blob_detection"]["window_dimensions"]
        crop = (wd["window_dim_0"], wd["window_dim_1"], wd["window_dim_2"])
        if use_ilastik:
            raw_shape = tuple(int(v) for v in raw_dev.shape)
            print(f"Before upsampling: {mask_ds.shape}\nRaw shape {raw_shape}")
            mask_us = upsample_mask(eng, eng.to_device(mask_ds), raw_shape)           # zoom(order=2, prefilter=False) -> uint8
            padded = mask_and_pad(eng, raw_dev, mask_us, crop)                        # img *= mask_us[i]; zero padding
            if mask_ds.shape == ds_host.shape:                                        # (reference :333: mask * stack)
                np.save(os.path.join(results, "downsampled_masked_stack.npy"), mask_ds.astype(ds_host.dtype) * ds_host)
        else:
            padded = mask_and_pad(eng, raw_dev, None, crop, int(md["simple_threshold_value"]))
            np.save(os.path.join(results, "downsampled_masked_stack.npy"),
                    (ds_host > int(md["simple_threshold_value"])).astype(ds_host.dtype) * ds_host)  # (:316, :333)
        eng.sync()
        write_masked_nifti_npy(os.path.join(results, "masked_niftis", "masked_nifti.npy"), padded, engine=eng)
    finally:
        if own:
            eng.close()
    return results
