"""ctypes binding of libdelivr_hip.so (C ABI: include/delivr_hip.h).

The product path has no CPU fallback: if the shared library is missing this module raises at
import of the symbol table, and every op fails loudly when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdelivr_hip.so")
# development: A/B of two builds of the library on one device (profiles/): DLV_LIB names a file under lib/ - a bare file
# name of the form libdelivr_hip*.so only (no directories), and never one of the timing-only ablation builds
# (libdelivr_hip_abl*.so compute WRONG results by construction) unless DLV_ALLOW_WRONG_RESULTS=1 says the caller knows
# (profiles/tools/zreg_abl.sh)
if os.environ.get("DLV_LIB"):
    _name = os.environ["DLV_LIB"]
    if os.path.basename(_name) != _name or not (_name.startswith("libdelivr_hip") and _name.endswith(".so")):
        raise ImportError(f"DLV_LIB={_name!r}: expected the bare name of a libdelivr_hip*.so under {os.path.dirname(LIB_PATH)}")
    if "_abl" in _name and os.environ.get("DLV_ALLOW_WRONG_RESULTS") != "1":
        raise ImportError(f"DLV_LIB={_name!r} is a timing-only ablation build (wrong results); set DLV_ALLOW_WRONG_RESULTS=1 "
                          "to time it (profiles/tools/zreg_abl.sh does)")
    LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), _name)



def build_fingerprint(lib_path: str | None = None) -> dict:
    """What identifies the kernels a measurement ran on: sha256 of the shared library that is (or would be) loaded and of the
    sources it is built from (csrc/*.hip, *.h, Makefile, include/*.h - a rebuild of the same sources may differ in bytes).
    profiles/make_traffic.py stores it next to the PMC traffic figures, bench.py compares it with the loaded library."""
    import glob
    import hashlib

    lib_path = lib_path or LIB_PATH
    out = {"lib": os.path.basename(lib_path), "lib_sha256": None, "src_sha256": None}
    if os.path.isfile(lib_path):
        h = hashlib.sha256()
        with open(lib_path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        out["lib_sha256"] = h.hexdigest()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")]
                   + glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")))
    h = hashlib.sha256()
    for fn in files:
        if os.path.isfile(fn):
            h.update(os.path.basename(fn).encode() + b"\0")
            with open(fn, "rb") as f:
                h.update(f.read())
    out["src_sha256"] = h.hexdigest()
    return out


DLV_OK, DLV_EINVAL, DLV_EHIP, DLV_ENOMEM, DLV_ESTATE, DLV_EUNSUP, DLV_ERANGE = 0, -1, -2, -3, -4, -5, -6
ABI_VERSION = 2  # include/delivr_hip.h: DLV_ABI_VERSION
PREC_F32, PREC_BF16, PREC_F16, PREC_BF16_ALL = 0, 1, 2, 3  # (DLV_PREC_*: BF16 = bf16 below fp16 level 0, BF16_ALL = bf16 everywhere)
N_CONV, N_DECONV = 18, 4
PROF_MAX = 64


class DelivrHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdelivr_hip error {code}: {msg}")
        self.code = code


class UnetWeights(C.Structure):
    _fields_ = [
        ("features", C.c_int * 6),
        ("conv_w", C.c_void_p * N_CONV),
        ("conv_b", C.c_void_p * N_CONV),
        ("norm_g", C.c_void_p * N_CONV),
        ("norm_b", C.c_void_p * N_CONV),
        ("deconv_w", C.c_void_p * N_DECONV),
        ("deconv_b", C.c_void_p * N_DECONV),
        ("final_w", C.c_void_p),
        ("final_b", C.c_void_p),
    ]


class SwParams(C.Structure):
    _fields_ = [
        ("Zp", C.c_int), ("Yp", C.c_int), ("Xp", C.c_int),
        ("roi", C.c_int * 3),
        ("overlap", C.c_float),
        ("flip_dim", C.c_int),
        ("skip_threshold", C.c_int),
        ("precision", C.c_int),
        ("sw_batch", C.c_int),
        ("win_begin", C.c_int64),
        ("win_end", C.c_int64),
        ("z0", C.c_int), ("nz", C.c_int),
        ("repeat", C.c_int),
        ("blend_mode", C.c_int),
        ("sigma_scale", C.c_float),
        ("wsum_dev", C.c_void_p),
    ]


class SwStats(C.Structure):
    _fields_ = [("n_windows", C.c_int64), ("n_skipped", C.c_int64), ("n_forward_launches", C.c_int64)]


MAX_RANKS = 16


class ShardPlanC(C.Structure):
    _fields_ = [("world", C.c_int), ("n_windows", C.c_int64),
                ("win_begin", C.c_int64 * MAX_RANKS), ("win_end", C.c_int64 * MAX_RANKS),
                ("z_comp_lo", C.c_int * MAX_RANKS), ("z_comp_hi", C.c_int * MAX_RANKS),
                ("z_own_lo", C.c_int * MAX_RANKS), ("z_own_hi", C.c_int * MAX_RANKS)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


# name -> (restype, argtypes); every symbol include/delivr_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "dlv_abi_version": (C.c_int, []),
    "dlv_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "dlv_ctx_destroy": (C.c_int, [_P]),
    "dlv_last_error": (C.c_char_p, [_P]),
    "dlv_sync": (C.c_int, [_P]),
    "dlv_stream": (_P, [_P]),
    "dlv_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "dlv_free": (C.c_int, [_P, _P]),
    "dlv_memset_dev": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "dlv_copy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "dlv_copy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "dlv_unet_load": (C.c_int, [_P, C.POINTER(UnetWeights)]),
    "dlv_unet_blob_size": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "dlv_unet_blob_dev": (C.c_int, [_P, C.POINTER(_P)]),
    "dlv_unet_alloc_blob": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "dlv_unet_forward_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dlv_sw_num_windows": (C.c_int, [C.POINTER(SwParams), C.POINTER(C.c_int64)]),
    "dlv_sw_window_starts": (C.c_int, [C.POINTER(SwParams), C.POINTER(C.c_int64), C.c_int64]),
    "dlv_sw_window_max_dev": (C.c_int, [_P, C.POINTER(SwParams), _P, C.POINTER(C.c_int32), C.c_int64]),
    "dlv_sw_infer_dev": (C.c_int, [_P, C.POINTER(SwParams), _P, _P, _P, C.POINTER(SwStats)]),
    "dlv_shard_plan_make": (C.c_int, [C.POINTER(SwParams), C.c_int, C.POINTER(C.c_float), C.POINTER(ShardPlanC)]),
    "dlv_shard_slab": (C.c_int, [C.POINTER(ShardPlanC), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dlv_comm_init_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_P)]),
    "dlv_comm_destroy": (C.c_int, [_P]),
    "dlv_comm_size": (C.c_int, [_P]),
    "dlv_comm_uses_rccl": (C.c_int, [_P]),
    "dlv_comm_selftest": (C.c_int, [_P, C.c_size_t]),
    "dlv_comm_ctx": (_P, [_P, C.c_int]),
    "dlv_comm_last_error": (C.c_char_p, [_P]),
    "dlv_bcast_weights": (C.c_int, [_P, C.c_int]),
    "dlv_sw_infer_sharded": (C.c_int, [_P, C.POINTER(SwParams), C.POINTER(ShardPlanC), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(SwStats)]),
    "dlv_finalize_slab_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                        C.c_int, _P, _P]),
    "dlv_finalize_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.c_int, _P, _P]),
    "dlv_ccl26_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_uint64)]),
    "dlv_cc_stats_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint64, _P, _P, _P]),
    "dlv_block_mean_u16_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "dlv_zoom_spline2_u8_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "dlv_mask_pad_u16_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "dlv_trilinear_u16_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "dlv_affine_warp_u16_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _P, C.c_int, C.c_int, C.c_int]),
    "dlv_seam_pairs_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "dlv_relabel_u32_dev": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64]),
    "dlv_cc_stats_raw_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint64, _P, _P, _P, _P]),
    "dlv_paint_owner_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_uint64, _P]),
    "dlv_paint_apply_dev": (C.c_int, [_P, _P, _P, C.c_uint64, _P, C.c_int, _P]),
    "dlv_edt_u16_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
    "dlv_heatmap_counts_dev": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "dlv_gauss_blur_f32_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "dlv_tiff_last_error": (C.c_char_p, []),
    "dlv_tiff_plane_size": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dlv_tiff_read_plane_u16": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "dlv_tiff_write_plane": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dlv_tiff_stack_to_device": (C.c_int, [_P, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, _P, C.c_longlong,
                                            C.c_longlong, C.c_int]),
    "dlv_diag_set": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "dlv_cells_csv": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dlv_reserve_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "dlv_debug_stamps": (C.c_int, [_P, _P]),
    "dlv_debug_set_zm_variant": (C.c_int, [_P, C.c_int]),
    "dlv_debug_set_format": (C.c_int, [_P, C.c_int]),
    "dlv_debug_layer_bf16": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                       C.c_int]),
    "dlv_unet_set_conv_shift": (C.c_int, [_P, C.c_int, C.c_int]),
    "dlv_unet_get_conv_shift": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "dlv_unet_note_conv_shifts": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "dlv_range_report": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "dlv_range_next_shifts": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dlv_range_recover": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "dlv_comm_range_recover": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "dlv_set_lanes": (C.c_int, [_P, C.c_int]),
    "dlv_prof_enable": (C.c_int, [_P, C.c_int]),
    "dlv_prof_reset": (C.c_int, [_P]),
    "dlv_prof_report": (C.c_int, [_P, C.POINTER(ProfEntry), C.c_int, C.POINTER(C.c_int)]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the shared library and binds every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C delivr_cfos_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.dlv_abi_version() != ABI_VERSION:
        raise ImportError(f"libdelivr_hip.so reports ABI version {lib.dlv_abi_version()}, this package binds version {ABI_VERSION} "
                          "(include/delivr_hip.h: DLV_ABI_VERSION): rebuild with make -C delivr_cfos_amd/csrc")
    _lib = lib
    return lib
