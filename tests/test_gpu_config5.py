"""BASELINE config 5: "HIP trilinear downsample -> affine atlas-space warp -> inference -> cell-coord upsample".
The reference has no volume warp and no trilinear resampler (block mean + spline-2 zoom; mBrainAligner moves
cell coordinates), so these two kernels are north-star extensions held to an fp64 numpy restatement in oracle/
(bit-exact: the kernels compute in fp64 with contraction off, in the oracle's operation order); the coordinate leg is the
reference's own arithmetic (automate_mBrainaligner.py:261-284)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from delivr_cfos_amd.engine import HipEngine

    e = HipEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("ishape,oshape", [((9, 17, 23), (9, 17, 23)), ((9, 17, 23), (18, 34, 46)), ((16, 40, 33), (5, 13, 7)),
                                           ((3, 5, 4), (31, 29, 37)), ((1, 1, 1), (4, 3, 2)), ((64, 96, 80), (17, 7, 64))])
def test_trilinear_u16_bit_exact_vs_fp64_restatement(eng, ishape, oshape):
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(sum(ishape) + sum(oshape))
    v = rng.integers(0, 65536, size=ishape).astype(np.uint16)
    v.flat[0] = 65535
    out = eng.trilinear_u16(eng.to_device(v), oshape).cpu().numpy()
    ref = orc.trilinear_u16(v, oshape)
    assert out.dtype == np.uint16 and out.shape == tuple(oshape)
    assert np.array_equal(out, ref), int((out != ref).sum())
    if ishape == oshape:
        assert np.array_equal(out, v)  # identity


def _rand_affine(rng, shape):
    a = np.eye(3) + 0.25 * rng.standard_normal((3, 3))
    c = (np.asarray(shape, dtype=np.float64) - 1) / 2
    t = c - a @ c + rng.uniform(-3, 3, size=3)
    return np.concatenate([a, t[:, None]], axis=1)


def test_affine_warp_u16_bit_exact_vs_fp64_restatement(eng):
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(3)
    v = rng.integers(0, 65536, size=(20, 33, 41)).astype(np.uint16)
    vd = eng.to_device(v)
    ident = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float64)
    assert np.array_equal(eng.affine_warp_u16(vd, ident, v.shape).cpu().numpy(), v)
    shift = ident.copy()
    shift[:, 3] = (2, -3, 5)  # integer translation: a shifted copy, zeros where the source lies outside
    w = eng.affine_warp_u16(vd, shift, v.shape).cpu().numpy()
    assert np.array_equal(w, orc.affine_warp_u16(v, shift, v.shape))
    assert np.array_equal(w[:18, 3:, :36], v[2:, :30, 5:]) and w[18:].sum() == 0
    for k in range(6):  # rotations / shears / scalings, other output shapes
        m = _rand_affine(rng, v.shape)
        oshape = (20, 33, 41) if k % 2 == 0 else (13, 50, 29)
        out = eng.affine_warp_u16(vd, m, oshape).cpu().numpy()
        ref = orc.affine_warp_u16(v, m, oshape)
        assert np.array_equal(out, ref), (k, int((out != ref).sum()))
    from delivr_cfos_amd._lib import DelivrHipError

    with pytest.raises(DelivrHipError):
        eng.affine_warp_u16(vd, np.full((3, 4), np.nan), v.shape)


def test_config5_chain_downsample_warp_inference_cells_coordinates(tmp_path):
    """raw -> block-mean downsample (the reference's downsampler, downsample_and_mask.py:44) -> affine warp (a pure
    translation here, so that the expected cells are known exactly) -> run_inference -> count_blobs -> cell coordinates
    mapped back through the warp (hostlogic.affine_apply) and up-scaled with the reference's factor arithmetic
    (automate_mBrainaligner.py:280-282) - against the oracle running the same chain on the CPU."""
    import os
    import pickle

    import torch
    from delivr_cfos_amd.count_blobs import count_blobs
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.hostlogic import affine_apply, scale_cell_coords
    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict
    from oracle import delivr_oracle as orc

    raw = synth_volume_np((64, 128, 128), seed=12, dense=True)
    factors = (2, 2, 2)
    crop = (32, 32, 32)
    m = np.array([[1, 0, 0, 1], [0, 1, 0, -2], [0, 0, 1, 3]], dtype=np.float64)
    sd = random_state_dict(4)
    eng = HipEngine(0)
    ds = eng.block_mean_u16(eng.to_device(raw), factors)
    warped = eng.affine_warp_u16(ds, m, tuple(ds.shape))
    ds_ref = orc.block_mean_u16(raw, factors)
    warped_ref = orc.affine_warp_u16(ds_ref, m, ds_ref.shape)
    assert np.array_equal(ds.cpu().numpy(), ds_ref) and np.array_equal(warped.cpu().numpy(), warped_ref)
    shape = tuple(int(v) for v in warped.shape)  # (32, 64, 64): a multiple of the window already
    nifti = os.path.join(str(tmp_path), "masked_nifti.npy")
    arr = np.lib.format.open_memmap(nifti, mode="w+", dtype=np.uint16, shape=(1, 1) + shape)
    arr[0, 0] = warped_ref
    arr.flush()
    del arr
    assert os.path.getsize(nifti) == 128 + 2 * int(np.prod(shape))
    out = run_inference([nifti], str(tmp_path / "blob"), (1, 1) + shape, comment="b5", tta=False, crop_size=crop,
                        state_dict={"state_dict": sd}, precision="fp32")
    settings = {"postprocessing": {"output_location": str(tmp_path / "post") + "/"}, "FLAGS": {}}
    n = count_blobs(settings, str(tmp_path / "blob"), 0, "b5", (1, 1) + shape, engine=eng)
    stats = pickle.load(open(os.path.join(str(tmp_path / "post"), "b5-stats.pickle"), "rb"))
    binaries = np.load(os.path.join(out, "binary_segmentations", "binaries.npy"))
    eng.close()

    # the oracle's chain
    net = orc.build_unet(seed=None)
    net.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})
    acc = np.zeros(shape, dtype=np.float32)
    cnt = np.zeros(shape, dtype=np.uint8)
    orc.sliding_window_pass(warped_ref, crop, lambda x: orc.unet_forward(net, x), acc, cnt, 0.5, None, 1, fp16=False)
    mask_ref = orc.finalize(acc, cnt, warped_ref, shape, 0.5, 30)
    margin = np.abs(acc / np.maximum(cnt, 1)) < 1e-3
    assert np.array_equal(binaries[~margin], mask_ref[~margin])
    lab_ref, n_ref = orc.ccl26(binaries)  # labels of the mask the device produced (identical to mask_ref outside the margin)
    st_ref = orc.cc_stats(lab_ref, n_ref)
    assert n == n_ref
    np.testing.assert_array_equal(stats["centroids"][1:], st_ref["centroids"][1:])
    # coordinate leg: warped space -> down-sampled space -> original space
    cells_ds = affine_apply(m, stats["centroids"][1:])
    cells_raw = scale_cell_coords(cells_ds, raw.shape, ds_ref.shape, direction="up")
    f = np.asarray(raw.shape, dtype=np.float64) / np.asarray(ds_ref.shape, dtype=np.float64)  # automate_mBrainaligner.py:280-282
    ref_ds = st_ref["centroids"][1:] @ m[:, :3].T + m[:, 3]
    np.testing.assert_allclose(cells_raw, ref_ds * f, rtol=0, atol=1e-9)
    np.testing.assert_allclose(orc.scale_coords(cells_raw, raw.shape, ds_ref.shape), ref_ds, rtol=0, atol=1e-9)  # and back down
    if n_ref:
        assert cells_raw.shape == (n_ref, 3) and (cells_raw[:, 0] <= raw.shape[0] * 1.1).all()


@pytest.mark.parametrize("sampling", [(25.0, 25.0, 25.0), (6.0, 2.5, 2.5), (3.0, 1.0, 1.0)])
def test_edt_bit_exact_vs_scipy(sampling):
    """dlv_edt_u16_dev against the oracle (scipy's distance_transform_edt of the zero-padded stack, astype(uint16)):
    blobs of tissue with holes, a slab that touches the border, an all-foreground and an all-zero volume."""
    from delivr_cfos_amd.engine import HipEngine
    from oracle import delivr_oracle as orc

    eng = HipEngine(0)
    rng = np.random.default_rng(12)
    cases = []
    z, y, x = np.mgrid[0:37, 0:52, 0:45]
    ell = (((z - 18) / 15.0) ** 2 + ((y - 26) / 22.0) ** 2 + ((x - 22) / 19.0) ** 2 < 1.0)
    holes = rng.random(ell.shape) < 0.002
    cases.append((ell & ~holes).astype(np.uint16) * 900)
    slab = np.zeros((20, 30, 33), dtype=np.uint16)
    slab[:, 5:, :20] = 7
    cases.append(slab)
    cases.append(np.full((9, 11, 13), 3, dtype=np.uint16))
    cases.append(np.zeros((5, 6, 7), dtype=np.uint16))
    cases.append((rng.random((24, 40, 31)) > 0.01).astype(np.uint16))
    for vol in cases:
        got = eng.edt_u16(eng.to_device(vol), sampling).cpu().numpy()
        np.testing.assert_array_equal(got, orc.edt_depth_u16(vol, sampling))
    eng.close()


def test_depth_map_blobs_files_vs_oracle(tmp_path):
    """blob_depthmap.depth_map_blobs end to end (binaries.npy + down-sampled masked stack -> depthmap_####.tif planes)
    against the oracle's literal restatement of the reference loop (statistics row 0 = background painted first)."""
    from delivr_cfos_amd import blob_depthmap
    from delivr_cfos_amd.downsample.downsample_and_mask import read_tiff_plane
    from delivr_cfos_amd.engine import HipEngine
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(3)
    shape = (24, 40, 48)
    orig_um, down_um = (6.0, 2.0, 2.0), (12.0, 8.0, 8.0)
    ds_shape = tuple(int(n * o / d) for n, o, d in zip(shape, orig_um, down_um))  # (12, 10, 12)
    zz, yy, xx = np.mgrid[0:ds_shape[0], 0:ds_shape[1], 0:ds_shape[2]]
    stack = ((((zz - 5.5) / 5.0) ** 2 + ((yy - 4.5) / 4.2) ** 2 + ((xx - 5.5) / 5.2) ** 2) < 1.0).astype(np.uint16) * 1200
    bin_img = np.zeros(shape, dtype=np.uint8)
    for _ in range(30):
        c = [int(rng.integers(2, n - 3)) for n in shape]
        bin_img[c[0]:c[0] + 2, c[1]:c[1] + 3, c[2]:c[2] + 2] = 1
    brain = "brainA"
    pred = tmp_path / "pred"
    (pred / (brain + "_out") / "binary_segmentations").mkdir(parents=True)
    with open(pred / (brain + "_out") / "binary_segmentations" / "binaries.npy", "wb") as fh:
        fh.write(b"\x00" * 128)
        fh.write(bin_img.tobytes())
    (tmp_path / "mask" / brain).mkdir(parents=True)
    np.save(tmp_path / "mask" / brain / "downsampled_masked_stack.npy", stack)
    (tmp_path / "post").mkdir()
    settings = {
        "visualization": {"input_prediction_location": str(pred) + "/", "output_location": str(tmp_path / "viz"),
                          "cache_location": str(tmp_path / "cache")},
        "postprocessing": {"output_location": str(tmp_path / "post")},
        "mask_detection": {"output_location": str(tmp_path / "mask"),
                           "downsample_steps": {"original_um_x": orig_um[2], "original_um_y": orig_um[1], "original_um_z": orig_um[0],
                                                "downsample_um_x": down_um[2], "downsample_um_y": down_um[1], "downsample_um_z": down_um[0]}},
        "FLAGS": {"LOAD_ALL_RAM": True},
    }
    eng = HipEngine(0)
    blob_depthmap.depth_map_blobs(settings, brain, (1, 1) + shape, engine=eng)
    labels, n = orc.ccl26(bin_img)
    stats = orc.cc_stats(labels, n)
    want = orc.depth_map_blobs(bin_img, stats, n, stack, down_um, orig_um)
    out_dir = tmp_path / "viz" / brain / (brain + "_depthmap_tiffs")
    got = np.stack([read_tiff_plane(str(out_dir / ("depthmap_" + str(z).zfill(4) + ".tif"))) for z in range(shape[0])])
    np.testing.assert_array_equal(got, want)
    assert int((want > 0).sum()) > 0
    eng.close()
