"""Range guard of the fp16 default: a checkpoint whose raw activations exceed 65504 must produce
an error (DLV_ERANGE) naming the layer - and run_inference must repeat its passes in bf16 - instead of a silent all-zero /
garbage mask.  The loader this protects: inference/inference.py:199-200,222 (any checkpoint the user points at).

The test checkpoint scales the weights and bias of ONE conv block by 1e6: InstanceNorm removes the factor exactly, so the
network's fp32 output is unchanged (up to the eps term), but the raw tensor between that conv and its normalisation is
1e6 x larger and leaves fp16's range."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_padded_npy(path, vol, crop):
    from delivr_cfos_amd.hostlogic import padded_shape

    pad = padded_shape(vol.shape, crop)
    out = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint16, shape=(1, 1) + pad)
    assert out.offset == 128
    out[0, 0, : vol.shape[0], : vol.shape[1], : vol.shape[2]] = vol
    out.flush()
    return pad


def _scaled(sd, key, factor):
    out = {k: v.clone() for k, v in sd.items()}
    out[f"module.{key}.conv.weight"] *= factor
    out[f"module.{key}.conv.bias"] *= factor
    return out


@pytest.mark.parametrize("key,layer_next", [("conv_0.conv_1", 2), ("upcat_2.convs.conv_0", 15), ("upcat_1.convs.conv_1", 18)])
def test_fp16_overflow_is_reported_not_painted(key, layer_next):
    import torch
    from delivr_cfos_amd._lib import DLV_ERANGE, DelivrHipError
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    shape, roi = (64, 64, 96), (64, 64, 64)
    vol = synth_volume_np(shape, seed=12, dense=True)
    sd = random_state_dict(4)
    eng = HipEngine(0)
    v = eng.to_device(vol)

    def run(state, prec):
        eng.load_state_dict({"state_dict": state})
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        eng.sw_infer(eng.make_sw_params(shape, roi, 0.5, None, 0, prec), v, acc)
        eng.sync()
        return acc.cpu().numpy()

    base16, base_bf = run(sd, "fp16"), run(sd, "bf16")
    big = _scaled(sd, key, 1.0e6)
    with pytest.raises(DelivrHipError) as ei:
        run(big, "fp16")
    assert ei.value.code == DLV_ERANGE and "fp16 range exceeded" in str(ei.value) and "bf16" in str(ei.value), str(ei.value)
    if layer_next < 18:
        assert f"conv block {layer_next} " in str(ei.value), str(ei.value)  # detected by the NEXT block's statistics
    else:
        assert "non-finite logits" in str(ei.value), str(ei.value)
    print(str(ei.value))
    # bf16 holds the range: the same checkpoint runs, and - InstanceNorm being scale invariant - gives the unscaled result
    got = run(big, "bf16")
    assert np.isfinite(got).all()
    rel = float(np.sqrt(np.mean((got - base_bf) ** 2)) / base_bf.std())
    print(f"bf16 on the scaled checkpoint vs bf16 on the original: rel rms {rel:.2e}")
    assert rel < 3e-2, rel
    # and the plain forward API reports it too
    x = torch.from_numpy(vol[None, None, :64, :64, :64].astype(np.float32)).cuda()
    eng.load_state_dict({"state_dict": big})
    with pytest.raises(DelivrHipError) as ei2:
        eng.unet_forward(x, "fp16")
    assert ei2.value.code == DLV_ERANGE
    eng.load_state_dict({"state_dict": sd})  # the context stays usable after the error
    again = run(sd, "fp16")
    assert np.array_equal(again, base16)
    eng.close()


def test_run_inference_repeats_the_passes_in_bf16(tmp_path, capsys):
    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    crop = (32, 32, 32)
    vol = synth_volume_np((40, 64, 64), seed=9, dense=True)
    vol[:, :, :6] = 0
    nifti = os.path.join(str(tmp_path), "masked_nifti.npy")
    _write_padded_npy(nifti, vol, crop)
    big = _scaled(random_state_dict(6), "down_1.convs.conv_0", 1.0e6)
    masks = {}
    for tag, prec in (("guarded", "fp16"), ("bf16", "bf16")):
        out = run_inference([nifti], str(tmp_path / tag), (1, 1) + vol.shape, comment="b", tta=True, crop_size=crop,
                            state_dict={"state_dict": big}, precision=prec)
        masks[tag] = np.load(os.path.join(out, "binary_segmentations", "binaries.npy"))
        txt = capsys.readouterr().out
        assert ("repeating the inference passes with bf16" in txt) == (tag == "guarded"), txt
    assert masks["guarded"].any()
    assert np.array_equal(masks["guarded"], masks["bf16"])  # the retry IS the bf16 run, from zeroed accumulators
