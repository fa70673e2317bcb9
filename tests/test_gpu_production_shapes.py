"""The 16-bit production path against the ORACLE (torch-fp32 CPU restatement of MONAI's BasicUNet, oracle/delivr_oracle.py)
at the tile shapes production uses: one 64^3 patch (BASELINE config 1), one 96x96x64 window (the reference's default,
config.json:24-28), 128^3 windows (BASELINE configs 2-5), a batch-16 launch of the fused sliding-window path, and the
mask after blend + finalize on a 256^3 crop with 27 windows.  Batch-16 launches, z-segment splitting of the z-march conv,
the LDS-weights variant of the 8^3 level and the pipeline lanes only occur at these sizes.

Tolerances (north_star: "mask IoU >= 0.999 vs reference"):
    fp32 VALU path   max |logit - oracle| <= 5e-4 (logit std ~0.4), sign agreement >= 0.9995
    fp16 MFMA path   relative RMS <= 1e-2, sign agreement >= 0.999, mask IoU vs the ORACLE's mask >= 0.999
    bf16 MFMA path   "bf16" = DLV_PREC_BF16: bf16 at levels 1-4 of the U-Net, fp16 at level 0 (full resolution, where bf16's 8
                     significant bits were lost: oracle/bf16_budget.py) - relative RMS <= 5e-2, sign agreement >= 0.99, mask
                     IoU >= 0.999 (north_star), also on the margin-free logits of the seeded random weights;
                     "bf16_all" = bf16 at every level: mask IoU REPORTED and asserted >= 0.995 only - it does NOT meet the
                     north_star tolerance on margin-free logits (strict expected failure below).
The oracle needs ~1.7 s per 128^3 window on the GPU box's host cores, so the module computes the 27 windows of the crop once
(and once more per flip of the TTA schedule for the 13-pass comparison).

The last two tests close the parity chain to the REFERENCE's own accumulate arithmetic (fp16 logits summed in fp16 in raster
window order, uint8 count, fp16 divide - oracle/parity.py): reference == oracle(fp16 accumulate) is pinned bit for bit by
tests/golden/ref_blend.npz; here HIP(fp32 accumulate, 3 weighted passes) is compared with oracle(fp16 accumulate, 13 passes).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROI = (128, 128, 128)
CROP = (256, 256, 256)


@pytest.fixture(scope="module")
def net():
    import torch
    from oracle import delivr_oracle as orc

    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    n = orc.build_unet(seed=0)
    orc.randomize_affine(n, seed=1)
    return n


@pytest.fixture(scope="module")
def eng(net):
    from delivr_cfos_amd.engine import HipEngine

    e = HipEngine(0)
    e.load_state_dict({"state_dict": net.state_dict()})
    yield e
    e.close()


def _stats(out, ref):
    rel = float(np.sqrt(np.mean((out - ref) ** 2)) / ref.std())
    agree = float(((out >= 0) == (ref >= 0)).mean())
    return rel, agree, float(np.abs(out - ref).max())


TOL = {"fp32": (1e-3, 0.9995), "fp16": (1e-2, 0.999), "bf16": (5e-2, 0.99), "bf16_all": (5e-2, 0.99)}


def _check(tag, prec, out, ref):
    rel, agree, mx = _stats(out, ref)
    print(f"{tag} [{prec}]: rel rms {rel:.2e}  sign agreement {agree:.5f}  max abs {mx:.2e}  (ref std {ref.std():.3f})")
    assert np.isfinite(out).all()
    assert rel < TOL[prec][0], (tag, prec, rel)
    assert agree >= TOL[prec][1], (tag, prec, agree)
    if prec == "fp32":
        assert mx < 5e-4, (tag, mx)


@pytest.fixture(scope="module")
def crop(net):
    """256^3 synthetic brain crop, its 27 windows of 128^3 at 50 % overlap through the oracle (per-window logits kept),
    the oracle's blended sum and its mask after create_nifti_seg."""
    from oracle.parity import LogitCache
    from delivr_cfos_amd.synth import synth_volume_np
    from oracle import delivr_oracle as orc

    vol = synth_volume_np(CROP, seed=21)
    wins = orc.window_list(CROP, ROI, 0.5)
    assert len(wins) == 27
    cache = LogitCache(lambda x: orc.unet_forward(net, x))
    acc = np.zeros(CROP, dtype=np.float32)
    cnt = np.zeros(CROP, dtype=np.uint8)
    info = orc.sliding_window_pass(vol, ROI, cache.predictor(None), acc, cnt, 0.5, None, 1, fp16=False)
    logits = {i: cache.store[(None, i)][0, 0] for i in range(27)}
    assert info["n_skipped"] == 0 and len(logits) == 27
    mask = orc.finalize(acc, cnt, vol, CROP, 0.5, 30)
    return {"vol": vol, "wins": np.asarray(wins), "logits": logits, "acc": acc, "cnt": cnt, "mask": mask, "cache": cache}


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16", "bf16_all"])
def test_c1_single_64cube_patch_vs_oracle(eng, net, prec):
    """BASELINE config 1 / SURVEY 8(d) C1: x = randn(1,1,64,64,64, seed 0)*100 + 500."""
    import torch
    from oracle import delivr_oracle as orc

    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 1, 64, 64, 64), generator=g) * 100.0 + 500.0
    ref = orc.unet_forward(net, x.numpy())[0, 0]
    out = eng.unet_forward(x.cuda(), prec).cpu().numpy()[0, 0]
    _check("C1 64^3", prec, out, ref)
    # the committed samples of the same logits (made in the build container: tests/golden/orc_unet_c1.npz)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "orc_unet_c1.npz"))
    np.testing.assert_allclose(ref[::4, ::4, ::4], g["logits_s4"], atol=2e-5, rtol=0)
    if prec == "fp32":
        assert np.abs(out[::4, ::4, ::4] - g["logits_s4"]).max() < 5e-4


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_default_window_96_96_64_vs_oracle(eng, net, prec):
    """The reference's default window (config.json:24-28: 96,96,64) as one fused sliding-window launch."""
    import torch
    from delivr_cfos_amd.synth import synth_volume_np
    from oracle import delivr_oracle as orc

    roi = (96, 96, 64)
    vol = synth_volume_np(roi, seed=5, dense=True)
    ref = orc.unet_forward(net, vol.astype(np.float32)[None, None])[0, 0]
    acc = torch.zeros(roi, dtype=torch.float32, device="cuda")
    st = eng.sw_infer(eng.make_sw_params(roi, roi, 0.5, None, 0, prec), eng.to_device(vol), acc)
    eng.sync()
    assert st["n_windows"] == 1 and st["n_skipped"] == 0
    _check("96x96x64", prec, acc.cpu().numpy(), ref)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("roi,batch", [((80, 96, 112), 3), ((128, 64, 160), 2), ((96, 96, 64), 3), ((64, 64, 32), 5)])
def test_uneven_windows_through_the_deep_level_kernels_vs_oracle(eng, net, prec, roi, batch):
    """Windows whose deep levels do not fill the tiles of conv_deep.hip (4 x 8 x 16 / 8 x 8 x 8 voxels): (80,96,112) -> 20x24x28,
    10x12x14 (the 8-wide tile, partial in y and x), 5x6x7; (128,64,160) -> 32x16x40, 16x8x20, 8x4x10; the reference's shipped
    window (96,96,64) (config.json:24-28) -> 24x24x16, 12x12x8, 6x6x4; run_inference's own default (64,64,32)
    (inference/inference.py:119) -> level 1 of 32x32x16 (rows of 16: the 32-channel layers leave the z-reg conv), 16x16x8, 8x8x4,
    4x4x2 - several windows per launch (the persistent walk crosses item and window boundaries), each against the oracle; no
    layer of a 16-bit forward of these windows falls back to the generic conv3_mfma kernel."""
    import torch
    from delivr_cfos_amd.synth import synth_volume_np
    from oracle import delivr_oracle as orc

    shape = (roi[0], roi[1], roi[2] * batch)
    vol = synth_volume_np(shape, seed=7, dense=True)
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    eng.prof_reset()
    eng.prof_enable(True)
    st = eng.sw_infer(eng.make_sw_params(shape, roi, 0.0, None, 0, prec), eng.to_device(vol), acc)
    eng.sync()
    eng.prof_enable(False)
    ran = [k for k, e in eng.prof_report().items() if e["launches"]]
    assert st["n_windows"] == batch and st["n_skipped"] == 0
    assert any(k.startswith("conv3_deep_") for k in ran) and any(k.startswith("deconv2_deep_") for k in ran), ran
    assert not any(k.startswith("conv3_mfma_") for k in ran), ran
    out = acc.cpu().numpy()
    for b in range(batch):
        sl = slice(b * roi[2], (b + 1) * roi[2])
        ref = orc.unet_forward(net, vol[:, :, sl].astype(np.float32)[None, None])[0, 0]
        _check(f"{roi} window {b}", prec, out[:, :, sl], ref)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("roi,batch", [((72, 88, 104), 2), ((40, 64, 24), 3), ((120, 128, 128), 1), ((100, 100, 52), 2), ((67, 93, 45), 2),
                                       ((36, 20, 28), 3), ((97, 64, 129), 1)])
def test_windows_with_odd_levels_pool_and_pad_like_monai_vs_oracle(eng, net, prec, roi, batch):
    """MONAI's BasicUNet takes any window: MaxPool3d(2) drops the last plane of an odd level and UpCat replicate-pads the
    up-sampled tensor back to the skip tensor's size (basic_unet.py UpCat.forward, is_pad=True); the reference passes
    settings' window_dim_0..2 straight through (inference/inference.py:162-168).  Multiples of 16 keep every level even.
    (72,88,104): level 3 of 9x11x13, odd in every dimension (the transposed conv of 4x5x6 gives 8x10x12); (40,64,24): 5x8x3
    there, level 4 of 2x4x1; (120,128,128): a production-size window, odd in z only; (100,100,52): level 2 of 25x25x13;
    (67,93,45): odd at level 0 - the folded up-conv of upcat_1 does not apply, the stem / final conv / blend see odd rows - and
    again at levels 1-3 in some dimension; (36,20,28): small, levels 2-4 of 9x5x7, 4x2x3, 2x1x1; (97,64,129): full-size rows of an
    odd length in z and x.  Several windows per launch; every precision against the oracle (oracle/delivr_oracle.py UpCat pads
    with torch's F.pad(mode="replicate") as MONAI does)."""
    import torch
    from delivr_cfos_amd.synth import synth_volume_np
    from oracle import delivr_oracle as orc

    shape = (roi[0], roi[1], roi[2] * batch)
    vol = synth_volume_np(shape, seed=11, dense=True)
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    st = eng.sw_infer(eng.make_sw_params(shape, roi, 0.0, None, 0, prec), eng.to_device(vol), acc)
    eng.sync()
    assert st["n_windows"] == batch and st["n_skipped"] == 0
    out = acc.cpu().numpy()
    for b in range(batch):
        sl = slice(b * roi[2], (b + 1) * roi[2])
        ref = orc.unet_forward(net, vol[:, :, sl].astype(np.float32)[None, None])[0, 0]
        _check(f"{roi} window {b}", prec, out[:, :, sl], ref)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_random_window_shapes_vs_oracle(eng, net, prec):
    """A seeded sweep of window shapes nobody picked by hand: every dimension uniform in [16, 90], plus three that put odd rows
    on the kernels of levels 0 and 1 that take rows of 32 / 64 voxels (z-reg conv: W >= 32; the pooling pass by full lines:
    W % 64 == 0) and the smallest window there is (level 4 of 1 x 1 x 2).  Two windows per launch."""
    import torch
    from delivr_cfos_amd.synth import synth_volume_np
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(20260603)
    shapes = [tuple(int(v) for v in rng.integers(16, 91, size=3)) for _ in range(6)] + [(16, 16, 32), (18, 130, 35), (33, 21, 128), (50, 33, 70)]
    for roi in shapes:
        shape = (roi[0], roi[1], roi[2] * 2)
        vol = synth_volume_np(shape, seed=13, dense=True)
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        st = eng.sw_infer(eng.make_sw_params(shape, roi, 0.0, None, 0, prec), eng.to_device(vol), acc)
        eng.sync()
        assert st["n_windows"] == 2 and st["n_skipped"] == 0
        out = acc.cpu().numpy()
        for b in range(2):
            sl = slice(b * roi[2], (b + 1) * roi[2])
            ref = orc.unet_forward(net, vol[:, :, sl].astype(np.float32)[None, None])[0, 0]
            _check(f"{roi} window {b}", prec, out[:, :, sl], ref)


@pytest.mark.parametrize("roi", [(12, 64, 64), (16, 16, 16), (31, 16, 20)])
def test_windows_torch_refuses_are_refused(eng, net, roi):
    """four 2 x poolings need 16 voxels (torch: "Output size is too small"), and a level 4 of one voxel has no InstanceNorm
    statistics (torch: ValueError "Expected more than 1 spatial element") - the oracle raises, the library returns DLV_EUNSUP"""
    import torch
    from delivr_cfos_amd._lib import DelivrHipError
    from oracle import delivr_oracle as orc

    with pytest.raises((ValueError, RuntimeError)):
        orc.unet_forward(net, np.zeros((1, 1) + roi, dtype=np.float32))
    with pytest.raises(DelivrHipError, match="at least 16"):
        eng.sw_infer(eng.make_sw_params(roi, roi, 0.0, None, 0, "fp16"), torch.zeros(roi, dtype=torch.int16, device="cuda").view(torch.uint16),
                     torch.zeros(roi, dtype=torch.float32, device="cuda"))
    with pytest.raises(DelivrHipError, match="at least 16"):
        eng.unet_forward(torch.zeros((1, 1) + roi, dtype=torch.float32, device="cuda"), "fp32")


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_window_128cube_forward_vs_oracle(eng, crop, prec):
    """dlv_unet_forward_dev on the centre window of the crop (fp32 patch input) vs the oracle's logits of that window."""
    import torch

    i = 13
    z, y, x = crop["wins"][i]
    patch = crop["vol"][z : z + 128, y : y + 128, x : x + 128].astype(np.float32)
    out = eng.unet_forward(torch.from_numpy(patch)[None, None].cuda(), prec).cpu().numpy()[0, 0]
    _check("128^3 window", prec, out, crop["logits"][i])


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_batch16_fused_pass_vs_oracle(eng, crop, prec):
    """16 windows of 128^3 in ONE forward launch of the fused path (stem reads the uint16 volume, final layer blends):
    16 of the crop's windows are laid out side by side in a (256,256,512) volume and run with overlap 0 (one colour
    class, sw_batch 16), so that the accumulator holds exactly the per-window logits the oracle already computed."""
    import torch

    ids = [0, 2, 4, 6, 8, 10, 12, 13, 14, 16, 18, 20, 22, 24, 25, 26]
    big = np.zeros((256, 256, 512), dtype=np.uint16)
    ref = np.zeros(big.shape, dtype=np.float32)
    k = 0
    for bz in range(2):
        for by in range(2):
            for bx in range(4):
                z, y, x = crop["wins"][ids[k]]
                sl = np.s_[bz * 128 : bz * 128 + 128, by * 128 : by * 128 + 128, bx * 128 : bx * 128 + 128]
                big[sl] = crop["vol"][z : z + 128, y : y + 128, x : x + 128]
                ref[sl] = crop["logits"][ids[k]]
                k += 1
    acc = torch.zeros(big.shape, dtype=torch.float32, device="cuda")
    st = eng.sw_infer(eng.make_sw_params(big.shape, ROI, 0.0, None, 0, prec, sw_batch=16), eng.to_device(big), acc)
    eng.sync()
    assert st["n_windows"] == 16 and st["n_skipped"] == 0
    if prec != "fp32":
        assert st["n_forward_launches"] == 1  # one batch of 16
    out = acc.cpu().numpy()
    _check("batch 16 x 128^3", prec, out, ref)
    # every window individually (a wrong window would hide in the global RMS)
    for bz in range(2):
        for by in range(2):
            for bx in range(4):
                sl = np.s_[bz * 128 : bz * 128 + 128, by * 128 : by * 128 + 128, bx * 128 : bx * 128 + 128]
                rel, agree, _ = _stats(out[sl], ref[sl])
                assert rel < TOL[prec][0] * 1.5 and agree >= TOL[prec][1] - 0.002, (prec, bz, by, bx, rel, agree)


def _iou(a, b):
    a = a.astype(bool)
    b = b.astype(bool)
    return float((a & b).sum()) / max(float((a | b).sum()), 1.0)


def test_mask_iou_vs_oracle_256cube(eng, crop):
    """north_star tolerance, against the ORACLE (not the build's own fp32 path): blend of 27 windows + count map +
    create_nifti_seg (sigmoid >= 0.5, L1-30 eroded re-mask) -> mask IoU.  Also at threshold 0.3, which needs the mean
    (sum / count), not just the sign of the sum (reference inference.py:295)."""
    import torch
    from oracle import delivr_oracle as orc

    v = eng.to_device(crop["vol"])
    ious = {}
    for prec in ("fp32", "fp16", "bf16", "bf16_all"):
        acc = torch.zeros(CROP, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(CROP, dtype=torch.uint8, device="cuda")
        st = eng.sw_infer(eng.make_sw_params(CROP, ROI, 0.5, None, 0, prec), v, acc, cnt)
        eng.sync()
        assert st["n_windows"] == 27 and st["n_skipped"] == 0
        assert np.array_equal(cnt.cpu().numpy(), crop["cnt"])
        _check("blended sum 256^3", prec, acc.cpu().numpy(), crop["acc"])
        mask = eng.finalize(acc, cnt, v, CROP, 0.5, 30, 0).cpu().numpy()
        ious[prec] = _iou(mask, crop["mask"])
        fg = float(crop["mask"].mean())
        print(f"mask IoU vs ORACLE mask [{prec}]: {ious[prec]:.5f} (foreground fraction {fg:.3f})")
        if prec == "fp16":
            m03 = eng.finalize(acc, cnt, v, CROP, 0.3, 30, 0).cpu().numpy()
            ref03 = orc.finalize(crop["acc"], crop["cnt"], crop["vol"], CROP, 0.3, 30)
            assert _iou(m03, ref03) >= 0.999
            assert abs(float(ref03.mean()) - fg) > 1e-3  # the other threshold really selects another mask
    assert ious["fp32"] >= 0.9995, ious
    assert ious["fp16"] >= 0.999, ious
    assert ious["bf16"] >= 0.999, ious      # bf16 below an fp16 level 0: BASELINE configs[1]'s format meets north_star
    assert ious["bf16_all"] >= 0.995, ious  # reported; bf16 at every level does not meet 0.999 (module docstring)


def _hip_mask(eng, v, prec, tta, threshold=0.5):
    """The product's pass(es) as run_inference issues them: distinct passes weighted by `repeat` (hostlogic.pass_schedule),
    fp32 sums + uint8 count, finalize."""
    import torch
    from delivr_cfos_amd.hostlogic import pass_schedule

    acc = torch.zeros(CROP, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(CROP, dtype=torch.uint8, device="cuda")
    for flip, rep in pass_schedule(tta):
        eng.sw_infer(eng.make_sw_params(CROP, ROI, 0.5, flip, 0, prec, repeat=rep), v, acc, cnt)
    eng.sync()
    return acc, cnt, eng.finalize(acc, cnt, v, CROP, threshold, 30, 0).cpu().numpy()


REF_ARITH_TOL = {"fp32": 0.9995, "fp16": 0.999, "bf16": 0.999, "bf16_all": 0.995}


@pytest.mark.parametrize("tta", [False, True], ids=["1pass", "13pass_tta"])
def test_mask_vs_reference_accumulate_arithmetic(eng, crop, tta):
    """The middle link of the parity chain: the HIP mask (fp32 accumulate; TTA as 3 passes weighted 5:4:4) against the
    oracle run in the REFERENCE's arithmetic - fp16 logits += in fp16 in raster window order over all 1 / 13 passes, uint8
    count, fp16 divide, create_nifti_seg.  Reports IoU, the number of flipped voxels and the histogram of the reference's
    |mean logit| at the flipped voxels, per format; asserts the north_star tolerance for fp32 and the fp16 default."""
    import json
    from oracle.parity import flip_report, fp32_arithmetic, reference_arithmetic
    from oracle import delivr_oracle as orc

    ref = reference_arithmetic(orc, crop["vol"], ROI, crop["cache"], tta)
    f32 = fp32_arithmetic(orc, crop["vol"], ROI, crop["cache"], tta)
    assert int(ref["cnt"].max()) == (8 * 13 if tta else 8)
    # what deliberate difference D5 costs by itself: the oracle's own logits, fp32 sums vs the reference's fp16 sums
    d5 = flip_report(f32["mask"], ref["mask"], ref["mean"])
    print(f"oracle fp32-accumulate vs reference arithmetic ({'13 passes' if tta else '1 pass'}): {json.dumps(d5)}")
    assert d5["iou"] >= 0.9999, d5
    v = eng.to_device(crop["vol"])
    for prec in ("fp32", "fp16", "bf16", "bf16_all"):
        acc, cnt, mask = _hip_mask(eng, v, prec, tta)
        assert np.array_equal(cnt.cpu().numpy(), ref["cnt"])  # 5+4+4 weighted counts == 13 passes of uint8 += 1
        rep = flip_report(mask, ref["mask"], ref["mean"])
        print(f"HIP [{prec}] vs reference arithmetic ({'13 passes' if tta else '1 pass'}): {json.dumps(rep)}")
        assert rep["iou"] >= REF_ARITH_TOL[prec], (prec, rep)
        if prec == "fp32":  # only voxels whose mean logit is at rounding level may differ
            assert rep["max_abs_mean_at_flip"] < 2e-3, rep



@pytest.mark.parametrize("tta", [False, True], ids=["1pass", "13pass_tta"])
def test_bf16_meets_the_north_star_tolerance_on_margin_free_logits(eng, crop, tta):
    """BASELINE.json's configs name bf16; north_star asks for mask IoU >= 0.999 vs the reference path.  Rounds 1-5 carried this
    test as a strict expected failure (bf16 at every level: 0.998 / 0.9988).  The format "bf16" now keeps fp16 at level 0 -
    the five full-resolution blocks where oracle/bf16_budget.py located the loss - and is held to the tolerance here, on the
    worst case (margin-free logits of seeded random weights), for 1 pass and for the 13-pass schedule."""
    from oracle.parity import flip_report, reference_arithmetic
    from oracle import delivr_oracle as orc

    ref = reference_arithmetic(orc, crop["vol"], ROI, crop["cache"], tta)
    _acc, _cnt, mask = _hip_mask(eng, eng.to_device(crop["vol"]), "bf16", tta)
    rep = flip_report(mask, ref["mask"], ref["mean"])
    print(f"bf16 vs reference arithmetic ({'13 passes' if tta else '1 pass'}): IoU {rep['iou']:.5f}, {rep['flipped']} flipped voxels")
    assert rep["iou"] >= 0.999, rep


@pytest.mark.xfail(strict=True, reason="bf16 at EVERY level (8 significant bits at full resolution) does NOT meet the north_star tolerance "
                                       "(mask IoU >= 0.999 vs the reference arithmetic) on the margin-free logits of seeded random weights: "
                                       "measured 0.998 / 0.9988.  It is the range guard's last resort, not a format to choose")
@pytest.mark.parametrize("tta", [False, True], ids=["1pass", "13pass_tta"])
def test_bf16_at_every_level_misses_the_tolerance_on_margin_free_logits(eng, crop, tta):
    from oracle.parity import flip_report, reference_arithmetic
    from oracle import delivr_oracle as orc

    ref = reference_arithmetic(orc, crop["vol"], ROI, crop["cache"], tta)
    _acc, _cnt, mask = _hip_mask(eng, eng.to_device(crop["vol"]), "bf16_all", tta)
    rep = flip_report(mask, ref["mask"], ref["mean"])
    print(f"bf16_all vs reference arithmetic ({'13 passes' if tta else '1 pass'}): IoU {rep['iou']:.5f}, {rep['flipped']} flipped voxels")
    assert rep["iou"] >= 0.999, rep


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_pass_is_bitwise_repeatable_on_three_lanes(prec):
    """The same pass six times on the default three lanes (batches of several 128^3 windows; the persistent upconv kernel's
    LDS-DMA staging, counted vmcnt and tile walk, the non-temporal streams, the lane joins): every repetition must give the
    same bits - a race in any of them would show up as a difference between two runs."""
    import numpy as np
    import torch

    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_torch
    from delivr_cfos_amd.weights import random_state_dict

    eng = HipEngine(0)
    try:
        eng.load_state_dict({"state_dict": random_state_dict(5)})
        shape = (256, 256, 384)
        vol = synth_volume_torch(shape, 4, eng.device, dense=True)
        ref = None
        for rep in range(6):
            acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
            eng.sw_infer(eng.make_sw_params(shape, (128, 128, 128), 0.5, None, 0, prec, sw_batch=5), vol, acc)
            eng.sync()
            if ref is None:
                ref = acc.clone()
                assert torch.isfinite(ref).all() and float(ref.std()) > 0
            else:
                assert torch.equal(acc, ref), f"repetition {rep} differs in {int((acc != ref).sum())} voxels"
    finally:
        eng.close()
