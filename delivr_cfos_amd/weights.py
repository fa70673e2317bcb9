"""Seeded random parameters with MONAI BasicUNet names/shapes (inference/inference.py:190-197).

The trained checkpoint (models/inference_weights.tar) is absent from the reference snapshot, so
benchmarks and smoke tests use random-initialised weights of the same architecture: Conv/ConvTranspose
weights and biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch's default bound), InstanceNorm affine
gamma ~ 1 + 0.25 N(0,1), beta ~ 0.2 N(0,1) so that every parameter path is exercised.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

FEATURES = (32, 32, 64, 128, 256, 32)


def random_state_dict(seed: int = 0, features: Sequence[int] = FEATURES, module_prefix: bool = True) -> Dict[str, "object"]:
    import torch

    f = tuple(int(v) for v in features)
    rng = np.random.default_rng(seed)
    sd = {}

    def put(name, arr):
        sd[("module." if module_prefix else "") + name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))

    def conv(prefix, cin, cout):
        b = 1.0 / np.sqrt(cin * 27)
        put(f"{prefix}.conv.weight", rng.uniform(-b, b, size=(cout, cin, 3, 3, 3)))
        put(f"{prefix}.conv.bias", rng.uniform(-b, b, size=(cout,)))
        put(f"{prefix}.adn.N.weight", 1.0 + 0.25 * rng.standard_normal(cout))
        put(f"{prefix}.adn.N.bias", 0.2 * rng.standard_normal(cout))

    def two(prefix, cin, cout):
        conv(f"{prefix}.conv_0", cin, cout)
        conv(f"{prefix}.conv_1", cout, cout)

    two("conv_0", 1, f[0])
    two("down_1.convs", f[0], f[1])
    two("down_2.convs", f[1], f[2])
    two("down_3.convs", f[2], f[3])
    two("down_4.convs", f[3], f[4])
    for k, cin, cat, cout, halves in ((4, f[4], f[3], f[3], True), (3, f[3], f[2], f[2], True),
                                      (2, f[2], f[1], f[1], True), (1, f[1], f[0], f[5], False)):
        up = cin // 2 if halves else cin
        b = 1.0 / np.sqrt(up * 8)
        put(f"upcat_{k}.upsample.deconv.weight", rng.uniform(-b, b, size=(cin, up, 2, 2, 2)))
        put(f"upcat_{k}.upsample.deconv.bias", rng.uniform(-b, b, size=(up,)))
        two(f"upcat_{k}.convs", cat + up, cout)
    b = 1.0 / np.sqrt(f[5])
    put("final_conv.weight", rng.uniform(-b, b, size=(1, f[5], 1, 1, 1)))
    put("final_conv.bias", rng.uniform(-b, b, size=(1,)))
    return sd


def _default_trained_like_fixture() -> Optional[str]:
    """The trained-like checkpoint is TEST DATA (the repository keeps it under tests/golden/, regenerable with
    oracle/train_weights.py), not part of the product: the package does not look into tests/.  Callers name the file
    (bench.py, the tests and the profiling tools pass tests/golden/trained_like_weights.npz); $DLV_TRAINED_LIKE_FIXTURE or a copy
    under delivr_cfos_amd/data/ serve an installed package."""
    here = os.path.dirname(os.path.abspath(__file__))
    for c in (os.environ.get("DLV_TRAINED_LIKE_FIXTURE"), os.path.join(here, "data", "trained_like_weights.npz")):
        if c and os.path.isfile(c):
            return c
    return None


def trained_like_state_dict(fixture: Optional[str] = None, module_prefix: bool = True) -> Dict[str, "object"]:
    """``random_state_dict(seed=0)`` with the tensors of the trained-like fixture put in place: the two top levels of the
    network trained for a few hundred steps on the reference's training patches and on synth volumes
    (oracle/train_weights.py made the file; it holds fp16 values), the deep levels seeded random.  The logits of this
    network are bimodal and its mask is thousands of small blobs - the stand-in for the reference's absent checkpoint
    (models/inference_weights.tar, inference/inference.py:199-200) wherever agreement of masks and cell tables is measured."""
    import torch

    sd = random_state_dict(seed=0, module_prefix=module_prefix)
    fixture = fixture or _default_trained_like_fixture()
    if fixture is None or not os.path.isfile(fixture):
        raise FileNotFoundError("trained_like_state_dict: name the fixture file (the repository's copy is tests/golden/trained_like_weights.npz; "
                                "regenerable with oracle/train_weights.py) or set DLV_TRAINED_LIKE_FIXTURE - nothing falls back to other weights")
    z = np.load(fixture)
    pre = "module." if module_prefix else ""
    for k in z.files:
        if k.startswith("w:"):
            name = pre + k[2:]
            if name not in sd or tuple(sd[name].shape) != tuple(z[k].shape):
                raise KeyError(f"trained-like fixture: unexpected tensor {k[2:]} {z[k].shape}")
            sd[name] = torch.from_numpy(z[k].astype(np.float32))
    return sd
