"""Import the reference's own Python for the hot path, unchanged, under stub modules.

TEST INFRASTRUCTURE ONLY, and usable ONLY in the build container (it reads /root/reference,
which does not exist on the GPU box).  It is used by ``oracle/make_goldens.py`` to produce
the committed fixtures under ``tests/golden/`` and by ``tests/test_oracle_golden.py``
(skipped automatically when /root/reference is absent).

The stubs restate the six MONAI 1.2.0 helpers the reference imports (monai==1.2.0 is pinned at
requirements.txt:21 but is not installed and not installable offline) [3P-recall]:
``dense_patch_slices``, ``get_valid_patch_size``, ``compute_importance_map`` (constant mode is
all the reference ever asks for: inference/sliding_window_inferer.py:148), ``fall_back_tuple``,
``look_up_option``, the ``BlendMode``/``PytorchPadMode`` enums, ``Inferer`` and
``RandGaussianNoise`` (stubbed to the identity: noise is <=1e-3 on raw-intensity scale).
``torch.Tensor.cuda`` is patched to the identity (inference/sliding_window_inferer.py:208
hard-codes ``.cuda()``).
"""
from __future__ import annotations

import enum
import importlib.util
import math
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("DELIVR_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "inference", "sliding_window_inferer.py"))


def _install_stubs() -> None:
    import torch

    if "monai" in sys.modules and getattr(sys.modules["monai"], "_delivr_stub", False):
        return

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    monai = mod("monai")
    monai._delivr_stub = True
    monai_data = mod("monai.data")
    monai_data_utils = mod("monai.data.utils")
    monai_utils = mod("monai.utils")
    monai_inferers = mod("monai.inferers")
    monai_inferers_inferer = mod("monai.inferers.inferer")
    monai_transforms = mod("monai.transforms")
    monai_networks = mod("monai.networks")
    monai_networks_nets = mod("monai.networks.nets")
    monai.data, monai.utils, monai.inferers, monai.transforms, monai.networks = (
        monai_data, monai_utils, monai_inferers, monai_transforms, monai_networks)
    monai_data.utils = monai_data_utils
    monai_inferers.inferer = monai_inferers_inferer
    monai_networks.nets = monai_networks_nets

    class BlendMode(enum.Enum):
        CONSTANT = "constant"
        GAUSSIAN = "gaussian"

    class PytorchPadMode(enum.Enum):
        CONSTANT = "constant"
        REFLECT = "reflect"
        REPLICATE = "replicate"
        CIRCULAR = "circular"

    def fall_back_tuple(user_provided, default, func=lambda x: x and x > 0):
        ndim = len(default)
        user = tuple(user_provided) if hasattr(user_provided, "__len__") else (user_provided,) * ndim
        return tuple(user_c if func(user_c) else default_c for default_c, user_c in zip(default, user))

    def look_up_option(opt, supported, default="no_default"):
        if isinstance(opt, enum.Enum):
            return opt
        return supported(opt)

    def get_valid_patch_size(image_size, patch_size):
        ndim = len(image_size)
        patch = tuple(patch_size) if hasattr(patch_size, "__len__") else (patch_size,) * ndim
        return tuple(min(ms, ps or ms) for ms, ps in zip(image_size, patch))

    def dense_patch_slices(image_size, patch_size, scan_interval):
        num_spatial_dims = len(image_size)
        patch_size = get_valid_patch_size(image_size, patch_size)
        scan_num = []
        for i in range(num_spatial_dims):
            if scan_interval[i] == 0:
                scan_num.append(1)
            else:
                num = int(math.ceil(float(image_size[i]) / scan_interval[i]))
                scan_dim = next((d for d in range(num) if d * scan_interval[i] + patch_size[i] >= image_size[i]), None)
                scan_num.append(scan_dim + 1 if scan_dim is not None else 1)
        starts = []
        for dim in range(num_spatial_dims):
            dim_starts = []
            for idx in range(scan_num[dim]):
                start_idx = idx * scan_interval[dim]
                start_idx -= max(start_idx + patch_size[dim] - image_size[dim], 0)
                dim_starts.append(start_idx)
            starts.append(dim_starts)
        out = np.asarray([x.flatten() for x in np.meshgrid(*starts, indexing="ij")]).T
        return [tuple(slice(s, s + patch_size[d]) for d, s in enumerate(x)) for x in out]

    def compute_importance_map(patch_size, mode=BlendMode.CONSTANT, sigma_scale=0.125, device="cpu"):
        mode = look_up_option(mode, BlendMode)
        if mode == BlendMode.CONSTANT:
            return torch.ones(patch_size, device=device).float()
        raise NotImplementedError("only the constant map is ever requested by the reference")

    class Inferer:
        def __init__(self):
            pass

    class RandGaussianNoise:
        def __init__(self, prob=0.1, mean=0.0, std=0.1):
            pass

        def __call__(self, x):
            return x

    class BasicUNet:  # only the name is needed to import inference.py
        def __init__(self, *a, **k):
            raise RuntimeError("MONAI is not installed; use oracle.delivr_oracle.build_unet")

    monai_utils.BlendMode = BlendMode
    monai_utils.PytorchPadMode = PytorchPadMode
    monai_utils.fall_back_tuple = fall_back_tuple
    monai_utils.look_up_option = look_up_option
    monai_data_utils.compute_importance_map = compute_importance_map
    monai_data_utils.dense_patch_slices = dense_patch_slices
    monai_data_utils.get_valid_patch_size = get_valid_patch_size
    monai_inferers_inferer.Inferer = Inferer
    monai_transforms.RandGaussianNoise = RandGaussianNoise
    monai_networks_nets.BasicUNet = BasicUNet

    # other third-party imports of inference/inference.py that are absent here
    if "nibabel" not in sys.modules:
        mod("nibabel")
    if "path" not in sys.modules:
        p = mod("path")

        class Path(str):
            def __add__(self, other):
                return Path(str.__add__(self, other))

        p.Path = Path
    if "skimage" not in sys.modules:
        sk = mod("skimage")
        sku = mod("skimage.util")
        sk.util = sku
        sku.view_as_windows = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))

    # inference/sliding_window_inferer.py:208 calls .cuda() unconditionally
    torch.Tensor.cuda = lambda self, *a, **k: self


_loaded = {}


def load_reference_inference():
    """Returns the reference's ``inference`` package modules:
    (sliding_window_inferer module, inference module)."""
    if "mods" in _loaded:
        return _loaded["mods"]
    if not available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    _install_stubs()
    pkg_dir = os.path.join(REFERENCE_ROOT, "inference")
    pkg = types.ModuleType("delivr_ref_inference")
    pkg.__path__ = [pkg_dir]
    sys.modules["delivr_ref_inference"] = pkg

    def load(name):
        spec = importlib.util.spec_from_file_location(
            f"delivr_ref_inference.{name}", os.path.join(pkg_dir, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        return m

    swi = load("sliding_window_inferer")
    inf = load("inference")
    _loaded["mods"] = (swi, inf)
    return swi, inf
