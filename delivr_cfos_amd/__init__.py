"""delivr_cfos_amd - MI355X-native (gfx950) implementation of DELiVR's tiled 3D-U-Net cFos inference
path behind the reference's step-level API.  Compute lives in libdelivr_hip.so (HIP, C ABI in
include/delivr_hip.h); this package is the thin Python host side.  No CPU fallback."""
from .hostlogic import (arrayterator_zblock, cells_csv_text, csv_name, downsample_ratios, padded_shape,  # noqa: F401
                        pass_schedule, scale_cell_coords)

__all__ = ["HipEngine", "HipBasicUNet", "run_inference", "SlidingWindowInferer", "count_blobs"]


def __getattr__(name):
    # heavy imports (torch, the shared library) only when the device API is touched
    if name == "HipEngine":
        from .engine import HipEngine

        return HipEngine
    if name == "HipBasicUNet":
        from .model import HipBasicUNet

        return HipBasicUNet
    if name == "run_inference":
        from .inference.inference import run_inference

        return run_inference
    if name == "SlidingWindowInferer":
        from .inference.sliding_window_inferer import SlidingWindowInferer

        return SlidingWindowInferer
    if name == "count_blobs":
        from .count_blobs import count_blobs

        return count_blobs
    raise AttributeError(name)
