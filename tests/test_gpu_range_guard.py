"""Range guard of the fp16 default: a checkpoint whose raw activations exceed 65504 must produce an error (DLV_ERANGE) naming
the layer - and run_inference must repeat its passes with that block rescaled (still fp16: range_guard.py), bf16 only as the
last resort - instead of a silent all-zero / garbage mask.  The loader this protects: inference/inference.py:199-200,222 (any checkpoint the user points at).

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

    base16, base_bf = run(sd, "fp16"), run(sd, "bf16_all")
    big = _scaled(sd, key, 1.0e6)
    with pytest.raises(DelivrHipError) as ei:
        run(big, "fp16")
    assert ei.value.code == DLV_ERANGE and "fp16 range exceeded" in str(ei.value) and "dlv_unet_set_conv_shift" in str(ei.value), str(ei.value)
    if layer_next < 18:
        assert f"conv block {layer_next} " in str(ei.value), str(ei.value)  # detected by the NEXT block's statistics
    else:
        assert "non-finite logits" in str(ei.value), str(ei.value)
    print(str(ei.value))
    # bf16 holds the range: the same checkpoint runs, and - InstanceNorm being scale invariant - gives the unscaled result
    got = run(big, "bf16_all")
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


def test_conv_shift_is_invisible_behind_instancenorm_and_cures_the_overflow():
    """dlv_unet_set_conv_shift: a block's 16-bit weights times 2^-k (exact), eps times 4^-k: (i) on an ordinary checkpoint the
    result moves by rounding only, for the stem, an encoder block, the folded upcat_1.conv_0 and the last block; (ii) on the
    checkpoint scaled by 1e6 the shift the library's own report suggests makes the fp16 pass finite and equal to the unscaled
    one to fp16 rounding."""
    import torch
    from delivr_cfos_amd._lib import DLV_ERANGE, DelivrHipError
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.range_guard import next_shifts
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    shape, roi = (64, 64, 128), (64, 64, 64)
    vol = synth_volume_np(shape, seed=14, dense=True)
    sd = random_state_dict(4)
    eng = HipEngine(0)
    v = eng.to_device(vol)

    def run(prec="fp16"):
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        eng.sw_infer(eng.make_sw_params(shape, roi, 0.5, None, 0, prec), v, acc)
        eng.sync()
        return acc.cpu().numpy()

    eng.load_state_dict({"state_dict": sd})
    base = run()
    for layer in (0, 3, 9, 16, 17):
        eng.set_conv_shift(layer, 5)
        assert eng.conv_shifts()[layer] == 5
        got = run()
        rel = float(np.sqrt(np.mean((got - base) ** 2)) / base.std())
        print(f"shift 5 on block {layer}: rel rms vs unshifted {rel:.2e}")
        assert rel < 2e-3, (layer, rel)
        eng.set_conv_shift(layer, 0)
    assert np.array_equal(run(), base)  # shifts back to 0: the same packs, the same bits
    for key, layer in (("down_2.convs.conv_1", 5), ("upcat_1.convs.conv_0", 16)):
        eng.load_state_dict({"state_dict": _scaled(sd, key, 1.0e6)})
        assert eng.conv_shifts() == [0] * 18  # a new checkpoint starts unshifted
        with pytest.raises(DelivrHipError) as ei:
            run()
        assert ei.value.code == DLV_ERANGE
        for attempt in range(4):
            named, peaks = eng.range_report()
            plan = next_shifts(named, peaks, eng.conv_shifts())
            if layer == 5:  # the fp32 statistics of the block itself stayed finite and say how large it is: one step
                assert peaks[layer] > 1e6 and plan == {layer: plan[layer]} and 10 <= plan[layer] <= 30, (named, peaks, plan)
            else:           # the folded up half of block 16 (P, 16-bit) overflows before the block's statistics exist: 6 bits blind,
                            # then the block's own statistics - finite once P is - give the rest
                assert named in (16, 17) and list(plan) == [16], (named, plan)
            for p, k in plan.items():
                eng.set_conv_shift(p, k)
            try:
                got = run()
                break
            except DelivrHipError as e:
                assert e.code == DLV_ERANGE
        assert np.isfinite(got).all()
        plan = {i: k for i, k in enumerate(eng.conv_shifts()) if k}
        assert list(plan) == [layer] and 8 <= plan[layer] <= 30, plan
        rel = float(np.sqrt(np.mean((got - base) ** 2)) / base.std())
        print(f"{key} x 1e6 with shifts {plan}: rel rms vs the unscaled fp16 pass {rel:.2e}")
        assert rel < 2e-3, rel
    eng.close()


@pytest.mark.parametrize("streamed", [False, True])
def test_run_inference_recovers_in_fp16_within_the_north_star_tolerance(tmp_path, capsys, streamed):
    """run_inference on a checkpoint that overflows fp16: the passes are repeated in fp16 with the offending block rescaled
    (range_guard.py) - NOT in bf16 - and the mask meets north_star's IoU >= 0.999 against the oracle run of the same checkpoint
    in the reference's arithmetic (13 passes under TTA).  streamed: the slab-streamed path (the failed run's slabs must be gone
    before the repeat allocates its own)."""
    from oracle.parity import LogitCache, flip_report, reference_arithmetic
    from oracle import delivr_oracle as orc
    from delivr_cfos_amd.hostlogic import padded_shape
    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    crop = (32, 32, 32)
    vol = synth_volume_np((40, 64, 64), seed=9, dense=True)
    vol[:, :, :6] = 0
    nifti = os.path.join(str(tmp_path), "masked_nifti.npy")
    pad = _write_padded_npy(nifti, vol, crop)
    big = _scaled(random_state_dict(6), "down_1.convs.conv_0", 1.0e6)
    settings = {"blob_detection": {"window_dimensions": {"window_dim_0": 32, "window_dim_1": 32, "window_dim_2": 32}},
                "mi355x": {"precision": "fp16", **({"stream_slabs": 2} if streamed else {})}}
    out = run_inference([nifti], str(tmp_path / "guarded"), (1, 1) + vol.shape, comment="b", tta=True, crop_size=crop,
                        state_dict={"state_dict": big}, settings=settings)
    mask = np.load(os.path.join(out, "binary_segmentations", "binaries.npy"))
    txt = capsys.readouterr().out
    assert "fp16 range exceeded" in txt and "storing its raw output scaled by 2^-" in txt, txt
    assert "repeating the inference passes with bf16" not in txt, txt
    net = orc.build_unet(seed=None)
    net.load_state_dict({k.replace("module.", ""): v for k, v in big.items()})
    padded = np.zeros(pad, dtype=np.uint16)
    padded[: vol.shape[0], : vol.shape[1], : vol.shape[2]] = vol
    ref = reference_arithmetic(orc, padded, crop, LogitCache(lambda x: orc.unet_forward(net, x)), True, stack_shape=vol.shape)
    rep = flip_report(mask, ref["mask"], ref["mean"][: vol.shape[0], : vol.shape[1], : vol.shape[2]])
    print(f"fp16 recovery (streamed={streamed}) vs the reference arithmetic: {rep}")
    assert mask.any() and rep["iou"] >= 0.999, rep


def test_sharded_run_inference_recovers_in_fp16_on_every_rank(tmp_path):
    """The range guard under torch.distributed (two ranks on cuda:0, gloo): the rank whose windows overflow reports it, every rank
    learns the layer and the peaks through one all-reduce, applies the SAME block shift and repeats its passes in fp16; the
    gathered mask equals the single-process recovery's."""
    import socket
    import subprocess
    import sys

    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.weights import random_state_dict

    crop = (32, 32, 32)
    from delivr_cfos_amd.synth import synth_volume_np

    vol = synth_volume_np((72, 64, 64), seed=9, dense=True)
    vol[:, :, :6] = 0
    nifti = os.path.join(str(tmp_path), "masked_nifti.npy")
    _write_padded_npy(nifti, vol, crop)
    big = _scaled(random_state_dict(6), "down_1.convs.conv_0", 1.0e6)
    settings = {"blob_detection": {"window_dimensions": {"window_dim_0": 32, "window_dim_1": 32, "window_dim_2": 32}},
                "mi355x": {"precision": "fp16"}}
    single = run_inference([nifti], str(tmp_path / "one"), (1, 1) + vol.shape, comment="b", tta=False, crop_size=crop,
                           state_dict={"state_dict": big}, settings=settings)
    m1 = np.load(os.path.join(single, "binary_segmentations", "binaries.npy"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "run_inference_ranks.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), worker, nifti, str(tmp_path / "two"), *[str(v) for v in vol.shape]],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "storing its raw output scaled by 2^-" in r.stdout and "on every rank" in r.stdout, r.stdout
    assert "with bf16 operands" not in r.stdout, r.stdout
    m2 = np.load(os.path.join(str(tmp_path / "two"), "b", "binary_segmentations", "binaries.npy"))
    assert m1.any() and m2.shape == m1.shape
    # (the seam sums associate differently in fp32: voxels whose mean logit is within rounding of 0 may flip)
    assert int((m1 != m2).sum()) <= 4, int((m1 != m2).sum())


def test_block_shifts_travel_with_the_broadcast_blob():
    """dlv_bcast_weights sends packs made with rank 0's block shifts: the receiving context must normalise with the matching
    eps (4^-shift), i.e. know the shifts - same pass, same bits on both ranks (C-ABI communicator, both ranks on device 0)."""
    import torch
    from delivr_cfos_amd.engine import HipComm
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    shape, roi = (64, 64, 64), (64, 64, 64)
    vol = synth_volume_np(shape, seed=15, dense=True)
    comm = HipComm([0, 0])
    e0, e1 = comm.engines
    e0.load_state_dict({"state_dict": _scaled(random_state_dict(4), "down_2.convs.conv_1", 1.0e6)})
    e0.set_conv_shift(5, 13)
    e0.set_conv_shift(2, 3)  # (an ordinary block too: with the default eps its normalisation would be visibly off)
    comm.bcast_weights(0)
    assert e1.conv_shifts() == e0.conv_shifts() and e1.conv_shifts()[5] == 13 and e1.conv_shifts()[2] == 3
    outs = []
    for e in (e0, e1):
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        e.sw_infer(e.make_sw_params(shape, roi, 0.5, None, 0, "fp16"), e.to_device(vol), acc)
        e.sync()
        outs.append(acc.cpu().numpy())
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])
    comm.close()
