"""BASELINE-size checks through size-independent properties (the oracle cannot run at these sizes
in seconds): 512^3 volume, 128^3 windows, 50 % overlap (BASELINE configs[1])."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPE, ROI = (512, 512, 512), (128, 128, 128)


@pytest.fixture(scope="module")
def setup():
    import torch
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_torch
    from delivr_cfos_amd.weights import random_state_dict

    eng = HipEngine(0)
    eng.load_state_dict({"state_dict": random_state_dict(0)})
    vol = synth_volume_torch(SHAPE, 1, eng.device)
    vol[:, :, :130] = 0  # make sure some windows are pure background
    torch.cuda.synchronize()
    yield eng, vol
    eng.close()


def test_full_pass_properties(setup):
    """count map == analytic coverage (7x7x7 windows: 1/2/4/8 per voxel, separable); shards compose;
    skipped windows contribute exactly -1000; the blended field is finite."""
    import torch

    eng, vol = setup
    p = eng.make_sw_params(SHAPE, ROI, 0.5, None, 0, "bf16")
    n = eng.num_windows(p)
    assert n == 343
    acc = torch.zeros(SHAPE, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(SHAPE, dtype=torch.uint8, device="cuda")
    st = eng.sw_infer(p, vol, acc, cnt)
    eng.sync()
    assert st["n_windows"] == 343 and 0 < st["n_skipped"] < 343
    cov1 = np.ones(512, dtype=np.int64)
    cov1[64:448] = 2
    cov = torch.from_numpy(cov1).cuda()
    expect = (cov[:, None, None] * cov[None, :, None] * cov[None, None, :]).to(torch.uint8)
    assert torch.equal(cnt, expect)
    assert torch.isfinite(acc).all()
    # windows made only of background voxels: every voxel covered ONLY by skipped windows holds -1000 * coverage
    bg_only = acc[:, :, :64]  # x < 64 is covered only by windows starting at x = 0, which lie inside x < 130
    assert torch.equal(bg_only, -1000.0 * expect[:, :, :64].float())
    # two shards (cut between z tile rows) compose to the same sums bit for bit
    parts = torch.zeros_like(acc)
    for rng in ((0, 147), (147, 343)):
        eng.sw_infer(eng.make_sw_params(SHAPE, ROI, 0.5, None, 0, "bf16", win_range=rng), vol, parts)
    eng.sync()
    assert torch.equal(parts, acc)
    # determinism: a second run is bitwise identical
    again = torch.zeros_like(acc)
    eng.sw_infer(p, vol, again)
    eng.sync()
    assert torch.equal(again, acc)


def test_finalize_and_ccl_properties(setup):
    import torch

    eng, vol = setup
    g = torch.Generator(device="cuda").manual_seed(3)
    acc = torch.randn(SHAPE, generator=g, device="cuda") - 2.0  # ~2 % foreground
    mask = eng.finalize(acc, None, vol, SHAPE, 0.5, 30, 0)
    eng.sync()
    raw_fg = vol.to(torch.int32) > 0
    assert int(mask.sum()) > 0
    assert not bool((mask.bool() & ~raw_fg).any())            # mask is inside raw > 0
    assert not bool((mask.bool() & (acc < 0)).any())           # and only where logit >= 0
    assert int(mask[:, :, :160].sum()) == 0                    # background x<130 + L1 radius 30
    # erosion is idempotent w.r.t. a count map: acc*4 with cnt=4 gives the same mask
    cnt = torch.full(SHAPE, 4, dtype=torch.uint8, device="cuda")
    assert torch.equal(eng.finalize(acc * 4, cnt, vol, SHAPE, 0.5, 30, 0), mask)
    # z-blocked erosion keeps at least what the whole-volume erosion keeps (out-of-block counts as foreground)
    blocked = eng.finalize(acc, None, vol, SHAPE, 0.5, 30, 238)
    assert not bool((mask.bool() & ~blocked.bool()).any())

    labels, n = eng.ccl26(mask)
    st = eng.cc_stats(labels, n)
    lab = labels.view(torch.int32)
    assert n > 1000
    assert int(st["voxel_counts"][1:].sum()) == int(mask.sum())
    assert torch.equal(lab > 0, mask.bool())
    assert int(lab.max()) == n
    # idempotence: labelling the labelled foreground again gives identical labels
    labels2, n2 = eng.ccl26((lab > 0).to(torch.uint8))
    assert n2 == n and torch.equal(labels2, labels)
    # raster-order numbering: first voxel of label k precedes first voxel of label k+1
    flat = lab.flatten()
    idx = torch.nonzero(flat, as_tuple=False).flatten()
    first = torch.full((n + 1,), flat.numel(), dtype=torch.int64, device="cuda")
    first.scatter_reduce_(0, flat[idx].long(), idx, reduce="amin")
    assert bool((first[2:] > first[1:-1]).all())
    # 26-connectivity: no two different labels touch (checked along the 13 forward offsets on a sub-volume)
    sub = lab[:128, :256, :256]
    for dz in (0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue
                a = sub[: sub.shape[0] - dz, max(0, -dy): sub.shape[1] - max(0, dy), max(0, -dx): sub.shape[2] - max(0, dx)]
                b = sub[dz:, max(0, dy): sub.shape[1] - max(0, -dy), max(0, dx): sub.shape[2] - max(0, -dx)]
                both = (a > 0) & (b > 0)
                assert bool((a[both] == b[both]).all())
    # centroid inside the bounding box
    bb, ce = st["bounding_boxes"][1:], st["centroids"][1:]
    for k in range(3):
        assert np.all(ce[:, k] >= bb[:, 2 * k]) and np.all(ce[:, k] <= bb[:, 2 * k + 1])


def test_resamplers_fullsize_properties(setup):
    import torch

    eng, vol = setup
    ds = eng.block_mean_u16(vol, (4, 15, 15))
    assert tuple(ds.shape) == (128, 35, 35)
    # exact integer check of a few blocks against torch
    v = vol[:8, :30, :30].to(torch.int64).reshape(2, 4, 2, 15, 2, 15).sum(dim=(1, 3, 5)) // 900
    assert torch.equal(ds[:2, :2, :2].to(torch.int64), v)
    m = (ds.to(torch.int32) > 0).to(torch.uint8)
    up = eng.zoom_spline2_u8(m, SHAPE)
    assert set(torch.unique(up).tolist()) <= {0, 1}
    # align-corners mapping reproduces the samples at the corners
    assert int(up[0, 0, 0]) == int(m[0, 0, 0]) and int(up[-1, -1, -1]) == int(m[-1, -1, -1])
    padded = eng.mask_pad_u16(vol, up, (512, 512, 640))
    assert torch.equal(padded[:, :, :512].to(torch.int32), vol.to(torch.int32) * up.to(torch.int32))
    assert int(padded[:, :, 512:].to(torch.int32).abs().sum()) == 0


def test_mask_iou_vs_fp32_path_256cube():
    """north_star tolerance: mask IoU >= 0.999 against the fp32 path (itself within 2e-4 of the torch-fp32 oracle,
    test_unet_fp32_matches_oracle_golden) - met by the default fp16 format; bf16 (8 significant bits) reaches 0.998 with the
    seeded RANDOM weights that stand in for the absent checkpoint (logit std 0.39, no margin around 0)."""
    import torch
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_torch
    from delivr_cfos_amd.weights import random_state_dict

    shape, roi = (256, 256, 256), (128, 128, 128)
    eng = HipEngine(0)
    eng.load_state_dict({"state_dict": random_state_dict(0)})
    vol = synth_volume_torch(shape, 1, eng.device)
    masks = {}
    for prec in ("fp32", "fp16", "bf16"):
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        eng.sw_infer(eng.make_sw_params(shape, roi, 0.5, None, 0, prec), vol, acc)
        masks[prec] = eng.finalize(acc, None, vol, shape, 0.5, 30, 0).bool()
    eng.sync()

    def iou(a, b):
        return float((a & b).sum()) / max(float((a | b).sum()), 1.0)

    i16, ib = iou(masks["fp16"], masks["fp32"]), iou(masks["bf16"], masks["fp32"])
    print(f"mask IoU vs fp32 path: fp16 {i16:.5f}  bf16 {ib:.5f}  (foreground {float(masks['fp32'].float().mean()):.3f})")
    assert i16 >= 0.999, i16
    assert ib >= 0.995, ib
    eng.close()


# ---------------------------------------------------------------------------------------------------
# the headline size: (1024, 2048, 2048) = 2^32 voxels, 14 415 windows of 128^3 (BASELINE configs 3/4)
# ---------------------------------------------------------------------------------------------------
C3 = (1024, 2048, 2048)


def test_c3_size_pass_properties():
    """One default-format pass over the 2^32-voxel volume: 14 415 windows, a third of them background-skipped; the
    uint8 count map equals the analytic coverage (1/2/4/8, separable: 15 x 31 x 31 windows); voxels covered only by
    skipped windows hold exactly -1000 x coverage; everything is finite.  Voxel indices exceed 2^31 here."""
    import torch
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_torch
    from delivr_cfos_amd.weights import random_state_dict

    eng = HipEngine(0)
    try:
        eng.load_state_dict({"state_dict": random_state_dict(0)})
        vol = synth_volume_torch(C3, 2, eng.device)
        p = eng.make_sw_params(C3, ROI, 0.5, None, 0, "fp16")
        assert eng.num_windows(p) == 15 * 31 * 31
        acc = torch.zeros(C3, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(C3, dtype=torch.uint8, device="cuda")
        st = eng.sw_infer(p, vol, acc, cnt)
        eng.sync()
        assert st["n_windows"] == 14415 and 0.2 < st["n_skipped"] / 14415 < 0.6
        covs = []
        for n in C3:
            c = np.ones(n, dtype=np.int64)
            c[64: n - 64] = 2
            covs.append(torch.from_numpy(c).cuda())
        for z0 in range(0, C3[0], 128):   # compare slab by slab (the full product would need another 4 GB)
            expect = (covs[0][z0:z0 + 128, None, None] * covs[1][None, :, None] * covs[2][None, None, :]).to(torch.uint8)
            assert torch.equal(cnt[z0:z0 + 128], expect)
        assert bool(torch.isfinite(acc[::7]).all())
        # the corner block is outside the ellipsoid brain: only skipped windows cover it
        corner = acc[:64, :64, :64]
        assert torch.equal(corner, torch.full_like(corner, -1000.0))
        far = acc[-64:, -64:, -64:]      # linear indices just below 2^32
        assert torch.equal(far, torch.full_like(far, -1000.0))
    finally:
        eng.close()


def test_c3_size_ccl_and_stats_at_the_index_limits():
    """CCL-26 + statistics on a 2^32-voxel mask with components placed where 32-bit index arithmetic would break: at
    linear index 0, across index 2^31, at the last voxel (2^32 - 1), a diagonal 26-connected chain, and a plane-spanning
    bar.  Labels are numbered in raster order of the first voxel; counts, boxes and centroids are exact."""
    import torch
    from delivr_cfos_amd.engine import HipEngine

    Z, Y, X = C3
    eng = HipEngine(0)
    try:
        m = torch.zeros(C3, dtype=torch.uint8, device="cuda")
        m[0, 0, 0:3] = 1                                   # 1: starts at linear index 0
        for k in range(5):
            m[10 + k, 20 + k, 30 + k] = 1                  # 2: diagonal chain (26-connectivity only)
        m[511:513, 2047, 2040:2048] = 1                    # 3: crosses linear index 2^31 (z 511 -> 512)
        m[700, 100:1900, 77] = 1                           # 4: a long bar
        m[1023, 2047, 2045:2048] = 1                       # 5: ends at the last voxel, index 2^32 - 1
        m[1023, 2040, 2047] = 1                            # 6: isolated voxel in the last plane ... comes before 5 in raster order
        labels, n = eng.ccl26(m)
        assert n == 6
        lab = labels.view(torch.int32)
        assert int(lab[0, 0, 1]) == 1 and int(lab[12, 22, 32]) == 2 and int(lab[512, 2047, 2047]) == 3
        assert int(lab[700, 1000, 77]) == 4
        assert int(lab[1023, 2040, 2047]) == 5 and int(lab[1023, 2047, 2047]) == 6
        assert int((lab != 0).sum()) == 3 + 5 + 16 + 1800 + 1 + 3
        st = eng.cc_stats(labels, n)
        np.testing.assert_array_equal(st["voxel_counts"][1:], [3, 5, 16, 1800, 1, 3])
        np.testing.assert_array_equal(st["bounding_boxes"][3], [511, 512, 2047, 2047, 2040, 2047])
        np.testing.assert_array_equal(st["bounding_boxes"][6], [1023, 1023, 2047, 2047, 2045, 2047])
        np.testing.assert_allclose(st["centroids"][3], [511.5, 2047.0, 2043.5], rtol=0, atol=0)
        np.testing.assert_allclose(st["centroids"][4], [700.0, 999.5, 77.0], rtol=0, atol=0)
        np.testing.assert_allclose(st["centroids"][6], [1023.0, 2047.0, 2046.0], rtol=0, atol=0)
        assert int(st["voxel_counts"][0]) == (Z * Y * X - 1828) % (1 << 32)   # the background count wraps in uint32 like cc3d's
    finally:
        eng.close()


def test_c3_size_finalize_blocks_equal_per_block_evaluation():
    """create_nifti_seg at the headline size: the reference erodes inside Arrayterator z-blocks of floor(1e9/(Y*X)) = 238
    planes (inference/inference.py:53); finalize(zblock=238) over 1024 planes must equal finalize(zblock=0) applied to
    each 238-plane block on its own (5 blocks, the last one short), and differ from a whole-volume erosion at the
    block seams."""
    import torch
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.hostlogic import arrayterator_zblock
    from delivr_cfos_amd.synth import synth_volume_torch

    Z, Y, X = C3
    nb = arrayterator_zblock(C3)
    assert nb == 238
    eng = HipEngine(0)
    try:
        raw = synth_volume_torch(C3, 2, eng.device)
        g = torch.Generator(device="cuda").manual_seed(3)
        acc = torch.empty(C3, dtype=torch.float32, device="cuda")
        for z0 in range(0, Z, 128):
            acc[z0:z0 + 128] = torch.randn((min(128, Z - z0), Y, X), generator=g, device="cuda") + 0.5
        whole = eng.finalize(acc, None, raw, C3, 0.5, 30, nb)
        for z0 in range(0, Z, nb):
            z1 = min(z0 + nb, Z)
            part = eng.finalize(acc[z0:z1].contiguous(), None, raw[z0:z1].contiguous(), (z1 - z0, Y, X), 0.5, 30, 0)
            assert torch.equal(whole[z0:z1], part), (z0, z1)
        noblock = eng.finalize(acc, None, raw, C3, 0.5, 30, 0)
        assert not torch.equal(noblock, whole)            # the block seams matter (border_value=1 inside each block)
        assert torch.equal(noblock[40:190], whole[40:190])  # but only within 30 planes of a seam
        assert 0.05 < float(whole.float().mean()) < 0.6
    finally:
        eng.close()


def test_c5_shape_resamplers_at_the_headline_size():
    """BASELINE config 5: raw (1024,2048,2048) -> block mean (4,15,15) with the reference's dropped last z-chunk ->
    (255,137,137) -> threshold mask -> spline-2 zoom back to 2^32 voxels -> raw * mask into the padded
    network input.  Exact integer spot checks, partial-block zero padding, align-corners corners, padding region zero."""
    import torch
    from delivr_cfos_amd.downsample.downsample_and_mask import downsample_volume, mask_and_pad, upsample_mask
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.hostlogic import padded_shape
    from delivr_cfos_amd.synth import synth_volume_torch

    Z, Y, X = C3
    eng = HipEngine(0)
    try:
        raw = synth_volume_torch(C3, 2, eng.device)
        ds = downsample_volume(eng, raw, (4, 15, 15))
        assert tuple(ds.shape) == (255, 137, 137)
        blk = raw[400:408, 900:930, 1200:1230].to(torch.int64).reshape(2, 4, 2, 15, 2, 15).sum(dim=(1, 3, 5)) // 900
        assert torch.equal(ds[100:102, 60:62, 80:82].to(torch.int64), blk)
        # last in-plane block is partial (2048 = 136*15 + 8): zero padded, still divided by the full block size
        edge = raw[0:4, 2040:2048, 2040:2048].to(torch.int64).sum() // 900
        assert int(ds[0, 136, 136]) == int(edge)
        small = (ds.to(torch.int32) > 300).to(torch.uint8)
        up = upsample_mask(eng, small, C3)
        assert tuple(up.shape) == C3 and set(torch.unique(up[::16, ::16, ::16]).tolist()) <= {0, 1}
        assert int(up[0, 0, 0]) == int(small[0, 0, 0]) and int(up[-1, -1, -1]) == int(small[-1, -1, -1])
        assert int(up[512, 1024, 1024]) == 1 and int(up[0, 0, 0]) == 0      # brain centre in, box corner out
        crop = (96, 96, 64)                                                  # the reference's default window
        padded = mask_and_pad(eng, raw, up, crop)
        Zp, Yp, Xp = padded_shape(C3, crop)
        assert tuple(padded.shape) == (Zp, Yp, Xp) and padded.dtype == torch.uint16
        sl = padded[500:520, :Y, :X].to(torch.int32)
        assert torch.equal(sl, raw[500:520].to(torch.int32) * up[500:520].to(torch.int32))
        assert int(padded[Z:].to(torch.int32).abs().sum()) == 0 and int(padded[:, Y:].to(torch.int32).abs().sum()) == 0
    finally:
        eng.close()
