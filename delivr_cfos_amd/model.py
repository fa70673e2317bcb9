"""HipBasicUNet - stands where the reference builds ``BasicUNet(...)`` + ``torch.nn.DataParallel``
(inference/inference.py:190-222): a callable network whose forward runs on libdelivr_hip."""
from __future__ import annotations

from typing import Optional

from .engine import HipEngine, shared_engine


class HipBasicUNet:
    """BasicUNet(spatial_dims=3, in_channels=1, out_channels=1, features=(32,32,64,128,256,32),
    act="mish", norm=instance) with its parameters resident in HBM.  ``precision``: "fp16" (MFMA on IEEE-half
    operands, default: mask IoU >= 0.999 vs the fp32 path), "bf16" (bf16 at levels 1-4, fp16 at full resolution), "bf16_all" (bf16 everywhere, 8 significant bits) or "fp32"
    (VALU parity mode)."""

    def __init__(self, device: int = 0, precision: str = "fp16", engine: Optional[HipEngine] = None, shared: bool = False):
        """engine: run on this engine; shared=True: on the process-wide engine of `device` (engine.shared_engine: context and
        workspaces survive between brains), else a fresh engine of its own"""
        self.engine = engine if engine is not None else (shared_engine(device) if shared else HipEngine(device))
        self.precision = precision

    # torch.nn.Module look-alikes used by the reference's call sequence (inference.py:217-222,262)
    def load_state_dict(self, state_dict, strict: bool = True):
        self.engine.load_state_dict(state_dict)
        return self

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def __call__(self, x):
        """(B,1,d,h,w) float32 tensor in HBM -> logits (sliding_window_inferer.py:222)."""
        return self.engine.unet_forward(x.contiguous(), self.precision)
