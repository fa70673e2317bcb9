"""HIP path vs the CPU oracle / committed golden vectors, through the C ABI.  Needs an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    from delivr_cfos_amd.engine import HipEngine

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    e = HipEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def net():
    from oracle import delivr_oracle as orc

    n = orc.build_unet(seed=0)
    orc.randomize_affine(n, seed=1)
    return n


@pytest.fixture(scope="module")
def eng_w(eng, net):
    eng.load_state_dict({"state_dict": {"module." + k: v for k, v in net.state_dict().items()}})
    return eng


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


# ---------------------------------------------------------------------------------------------------
def test_unet_fp32_matches_oracle_golden(eng_w, golden_dir):
    """fp32 VALU path vs torch-fp32 logits; tolerance: 2e-4 absolute on logits of std ~0.36."""
    import torch

    g = _g(golden_dir, "orc_unet.npz")
    for xk, lk in (("x32", "logits32"), ("x_odd", "logits_odd")):
        x = torch.from_numpy(g[xk].astype(np.float32))[None, None].cuda()
        out = eng_w.unet_forward(x, "fp32").cpu().numpy()[0, 0]
        ref = g[lk]
        err = np.abs(out - ref).max()
        assert err < 2e-4, (xk, err)
        assert ((out >= 0) == (ref >= 0)).mean() > 0.9995


def test_unet_fp32_batch_is_per_sample(eng_w, golden_dir):
    """InstanceNorm statistics are per (sample, channel): batching must not change a sample."""
    import torch

    g = _g(golden_dir, "orc_unet.npz")
    x = torch.from_numpy(g["x32"].astype(np.float32))[None, None].cuda()
    xb = torch.cat([x, x.flip(2), x * 0.5], dim=0).contiguous()
    out = eng_w.unet_forward(xb, "fp32")
    single = eng_w.unet_forward(x, "fp32")
    assert torch.equal(out[0], single[0])


@pytest.mark.parametrize("flip", [None, 2, 3, 4])
def test_sw_pass_fp32_matches_oracle(eng_w, net, golden_dir, flip):
    """tiler + per-window skip + flip + forward + fp32 blend vs the oracle's pass (fp32 accumulate).
    Tolerance 1e-3 absolute on sums of up to 8 logits; count map bit-exact vs the REFERENCE golden."""
    import torch
    from oracle import delivr_oracle as orc

    g = _g(golden_dir, "ref_blend.npz")
    vol = g["volume"]
    roi = (32, 32, 16)
    acc_ref = np.zeros(vol.shape, dtype=np.float32)
    cnt_ref = np.zeros(vol.shape, dtype=np.uint8)
    info = orc.sliding_window_pass(vol, roi, lambda x: orc.unet_forward(net, x), acc_ref, cnt_ref, 0.5, flip,
                                   sw_batch_size=1, fp16=False)
    v = eng_w.to_device(vol)
    acc = torch.zeros(vol.shape, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(vol.shape, dtype=torch.uint8, device="cuda")
    p = eng_w.make_sw_params(vol.shape, roi, 0.5, flip, 0, "fp32", sw_batch=3)
    st = eng_w.sw_infer(p, v, acc, cnt)
    eng_w.sync()
    assert st["n_windows"] == info["n_windows"] and st["n_skipped"] == info["n_skipped"] > 0
    np.testing.assert_array_equal(cnt.cpu().numpy(), g["p1_b1_count"])
    np.testing.assert_array_equal(cnt.cpu().numpy(), cnt_ref)
    a = acc.cpu().numpy()
    assert np.abs(a - acc_ref).max() < 1e-3
    # skipped windows contribute exactly -1000 each
    assert (a <= -999).any()


@pytest.mark.parametrize("world", [2, 3])
def test_c_abi_sharded_pass_fp32_matches_the_oracle(net, golden_dir, world):
    """The DataParallel replacement against the ORACLE, not against the single-device HIP pass: dlv_comm_init_all /
    dlv_bcast_weights / dlv_sw_infer_sharded with every rank on device 0 and its own Z-slab of volume and accumulator (fp32
    network), every owned plane compared with the oracle's sliding-window pass (fp32 accumulate) - sums to 1e-3 absolute on up to
    8 logits, count map bit-exact vs the REFERENCE golden (ref_blend.npz: p1_b1).  inference/inference.py:217-219."""
    import torch
    from oracle import delivr_oracle as orc
    from delivr_cfos_amd.engine import HipComm

    g = _g(golden_dir, "ref_blend.npz")
    vol = g["volume"]
    roi = (32, 32, 16)
    acc_ref = np.zeros(vol.shape, dtype=np.float32)
    cnt_ref = np.zeros(vol.shape, dtype=np.uint8)
    info = orc.sliding_window_pass(vol, roi, lambda x: orc.unet_forward(net, x), acc_ref, cnt_ref, 0.5, None, sw_batch_size=1, fp16=False)
    comm = HipComm([0] * world)
    comm.engines[0].load_state_dict({"state_dict": {"module." + k: v for k, v in net.state_dict().items()}})
    comm.bcast_weights(0)
    p = comm.engines[0].make_sw_params(vol.shape, roi, 0.5, None, 0, "fp32")
    plan = comm.make_plan(p, None)
    v = comm.engines[0].to_device(vol)
    slabs, vols, accs, cnts = [], [], [], []
    for r in range(world):
        lo, hi = plan.slab(r, vol.shape[0], 0, 0)
        slabs.append((lo, hi - lo))
        vols.append(v[lo:hi].clone())
        accs.append(torch.zeros((hi - lo,) + vol.shape[1:], dtype=torch.float32, device="cuda"))
        cnts.append(torch.zeros((hi - lo,) + vol.shape[1:], dtype=torch.uint8, device="cuda"))
    stats = comm.sw_infer_sharded(p, plan, slabs, vols, accs, cnts)
    torch.cuda.synchronize()
    assert sum(s["n_windows"] for s in stats) == info["n_windows"] and sum(s["n_skipped"] for s in stats) == info["n_skipped"] > 0
    covered = 0
    for r in range(world):
        olo, ohi = plan.z_owned[r]
        if ohi <= olo:
            continue
        lo = slabs[r][0]
        a = accs[r][olo - lo:ohi - lo].cpu().numpy()
        assert np.abs(a - acc_ref[olo:ohi]).max() < 1e-3, r
        np.testing.assert_array_equal(cnts[r][olo - lo:ohi - lo].cpu().numpy(), g["p1_b1_count"][olo:ohi])
        covered += ohi - olo
    assert covered == vol.shape[0]
    comm.close()


def test_window_enumeration_matches_reference(eng, golden_dir):
    g = _g(golden_dir, "ref_tiler.npz")
    for i in range(int(g["n_cases"])):
        img, roi = tuple(int(v) for v in g[f"case{i}_image"]), tuple(int(v) for v in g[f"case{i}_roi"])
        p = eng.make_sw_params(img, roi, 0.5)
        np.testing.assert_array_equal(eng.window_starts(p), g[f"case{i}_starts"])


def test_sw_shards_compose(eng_w, golden_dir):
    """windows [0,k) + [k,n) in two calls == one call (the multi-GPU partition is a pure split)."""
    import torch

    vol = _g(golden_dir, "ref_blend.npz")["volume"]
    v = eng_w.to_device(vol)
    roi = (32, 32, 16)
    full = torch.zeros(vol.shape, dtype=torch.float32, device="cuda")
    eng_w.sw_infer(eng_w.make_sw_params(vol.shape, roi, 0.5, None, 0, "fp32"), v, full)
    n = eng_w.num_windows(eng_w.make_sw_params(vol.shape, roi, 0.5))
    parts = torch.zeros_like(full)
    for rng in ((0, n // 3), (n // 3, n)):
        eng_w.sw_infer(eng_w.make_sw_params(vol.shape, roi, 0.5, None, 0, "fp32", win_range=rng), v, parts)
    eng_w.sync()
    assert torch.allclose(full, parts, atol=1e-4)


@pytest.mark.parametrize("tag", ["oneblock", "blocks"])
def test_finalize_matches_reference_golden(eng, golden_dir, tag):
    """threshold + L1-30 eroded re-mask, bit-exact vs the reference's create_nifti_seg output."""
    import torch

    g = _g(golden_dir, "ref_finalize.npz")
    Z, Y, X = (int(v) for v in g["stack_shape"][2:])
    buf = int(g[f"{tag}_buf"])
    zblock = max((buf // X) // Y, 1)
    acc = torch.from_numpy(g["mean"].astype(np.float32)).cuda()
    raw = eng.to_device(g["raw"])
    out = eng.finalize(acc, None, raw, (Z, Y, X), 0.5, 30, zblock if zblock < Z else 0)
    np.testing.assert_array_equal(out.cpu().numpy(), g[f"{tag}_binaries"])
    # with an explicit count map (mean = acc*cnt / cnt)
    cnt = torch.full(acc.shape, 4, dtype=torch.uint8, device="cuda")
    out2 = eng.finalize(acc * 4, cnt, raw, (Z, Y, X), 0.5, 30, zblock if zblock < Z else 0)
    np.testing.assert_array_equal(out2.cpu().numpy(), g[f"{tag}_binaries"])


def test_erosion_random_vs_scipy(eng):
    import torch
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(4)
    # radius 56 is the last one the bit-mask x-distance kernel takes (cap 57), 57 the first on the log-step kernel;
    # the 300-wide rows with few zeros hold distances beyond both
    for shape, r, p0 in (((33, 47, 52), 3, 0.003), ((20, 64, 75), 30, 0.003), ((5, 9, 13), 1, 0.003), ((6, 10, 300), 56, 0.0008),
                         ((6, 10, 300), 57, 0.0008), ((3, 5, 2051), 30, 0.002)):
        raw = (rng.random(shape) > p0).astype(np.uint16) * 77
        acc = torch.ones(shape, dtype=torch.float32, device="cuda")
        out = eng.finalize(acc, None, eng.to_device(raw), shape, 0.5, r, 0).cpu().numpy()
        np.testing.assert_array_equal(out, orc.erode_l1((raw > 0).astype(np.uint8), r))


def test_ccl_goldens_bit_exact(eng, golden_dir):
    import torch

    g = _g(golden_dir, "orc_ccl.npz")
    cases = []
    for key in ("gt0", "gt1"):
        shape = tuple(int(v) for v in g[f"{key}_shape"])
        cases.append((np.unpackbits(g[f"{key}_maskbits"])[: int(np.prod(shape))].reshape(shape), key))
    cases.append((g["adv_mask"], "adv"))
    for mask, key in cases:
        lab, n = eng.ccl26(eng.to_device(mask.astype(np.uint8)))
        assert n == int(g[f"{key}_n"])
        np.testing.assert_array_equal(lab.cpu().numpy().view(np.uint32), g[f"{key}_labels"].astype(np.uint32))
        st = eng.cc_stats(lab, n)
        np.testing.assert_array_equal(st["voxel_counts"], g[f"{key}_counts"])
        np.testing.assert_array_equal(st["bounding_boxes"], g[f"{key}_bbox"])
        np.testing.assert_array_equal(st["centroids"][1:], g[f"{key}_centroids"][1:])
        np.testing.assert_allclose(st["centroids"][0], g[f"{key}_centroids"][0], rtol=1e-12)


def test_ccl_random_and_edge_cases(eng):
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(9)
    for shape, dens in (((40, 50, 61), 0.05), ((17, 33, 130), 0.3), ((8, 8, 8), 0.0), ((6, 7, 9), 1.0), ((1, 1, 5), 0.5)):
        mask = (rng.random(shape) < dens).astype(np.uint8)
        lab, n = eng.ccl26(eng.to_device(mask))
        ref, nref = orc.ccl26(mask)
        assert n == nref
        np.testing.assert_array_equal(lab.cpu().numpy().view(np.uint32), ref)
        if n:
            st, sr = eng.cc_stats(lab, n), orc.cc_stats(ref, nref)
            np.testing.assert_array_equal(st["voxel_counts"][1:], sr["voxel_counts"][1:])
            np.testing.assert_array_equal(st["bounding_boxes"][1:], sr["bounding_boxes"][1:])
            np.testing.assert_array_equal(st["centroids"][1:], sr["centroids"][1:])


def test_resamplers_vs_goldens(eng, golden_dir):
    g = _g(golden_dir, "scipy_resample.npz")
    out = eng.block_mean_u16(eng.to_device(g["bm_in"]), tuple(int(v) for v in g["bm_factors"]))
    np.testing.assert_array_equal(out.cpu().numpy(), g["bm_out"])
    z = eng.zoom_spline2_u8(eng.to_device(g["zoom_in"]), g["zoom_out"].shape)
    np.testing.assert_array_equal(z.cpu().numpy(), g["zoom_out"])  # bit-exact incl. ties


@pytest.mark.parametrize("shape,factors", [((10, 47, 256), (4, 15, 15)), ((9, 31, 4104), (4, 15, 15)), ((7, 20, 64), (2, 3, 8)),
                                           ((5, 9, 72), (1, 1, 9)), ((6, 33, 250), (4, 15, 15))])
def test_block_mean_strip_kernel_vs_oracle(eng, shape, factors):
    """The coalesced strip kernel (X % 8 == 0: 16-byte loads, LDS column sums; several x chunks for X > 2040) and the
    one-voxel-per-thread kernel behind it (X % 8 != 0) against the oracle's integer restatement of
    downscale_local_mean(...).astype(uint16) (downsample/downsample_and_mask.py:44), ragged block ends included."""
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(sum(shape))
    vol = rng.integers(0, 65536, size=shape).astype(np.uint16)
    vol[:, :, : shape[2] // 5] = 65535  # full-scale blocks: the uint32 column sums must hold fz*fy rows of them
    out = eng.block_mean_u16(eng.to_device(vol), factors).cpu().numpy()
    np.testing.assert_array_equal(out, orc.block_mean_u16(vol, factors))


@pytest.mark.parametrize("in_shape,out_shape", [((12, 20, 24), (48, 300, 352)), ((12, 20, 24), (45, 290, 355)), ((30, 40, 50), (13, 17, 21)),
                                                ((6, 9, 11), (24, 135, 176))])
def test_zoom_run_kernel_bit_exact_vs_scipy(eng, in_shape, out_shape):
    """The 16-outputs-per-thread zoom with the uniform-neighbourhood shortcut against scipy itself
    (scipy.ndimage.zoom(..., output=uint8, order=2, prefilter=False), downsample/downsample_and_mask.py:299): a blobby
    mask with constant regions of several values (0, 1, 3, 200, 255 - the shortcut returns the region's value) and noisy
    edges (full evaluation incl. 1/2 ties), rows that are / are not a multiple of 16 long, and a down-sampling zoom."""
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(in_shape[0] * 100 + out_shape[2])
    m = np.zeros(in_shape, dtype=np.uint8)
    m[in_shape[0] // 4:, in_shape[1] // 3:, in_shape[2] // 3:] = 1
    m[: in_shape[0] // 3, : in_shape[1] // 2, : in_shape[2] // 4] = 255
    m[in_shape[0] // 2:, : in_shape[1] // 4, in_shape[2] // 2:] = 200
    m[:2, -3:, -4:] = 3
    noise = rng.random(in_shape) < 0.03
    m[noise] = rng.integers(0, 256, size=int(noise.sum())).astype(np.uint8)
    out = eng.zoom_spline2_u8(eng.to_device(m), out_shape).cpu().numpy()
    np.testing.assert_array_equal(out, orc.zoom_spline2_u8(m, out_shape))


def test_mask_pad_writer(eng):
    rng = np.random.default_rng(2)
    raw = rng.integers(0, 65535, size=(5, 7, 9)).astype(np.uint16)
    mask = (rng.random((5, 7, 9)) < 0.5).astype(np.uint8)
    out = eng.mask_pad_u16(eng.to_device(raw), eng.to_device(mask), (16, 16, 16)).cpu().numpy()
    ref = np.zeros((16, 16, 16), dtype=np.uint16)
    ref[:5, :7, :9] = raw * mask
    np.testing.assert_array_equal(out, ref)


def test_errors_are_reported_not_crashed(eng):
    import torch
    from delivr_cfos_amd._lib import DelivrHipError
    from delivr_cfos_amd.engine import HipEngine

    fresh = HipEngine(0)
    x = torch.zeros((1, 1, 16, 16, 16), dtype=torch.float32, device="cuda")
    with pytest.raises(DelivrHipError):
        fresh.unet_forward(x, "fp32")  # no weights loaded
    fresh.close()


@pytest.mark.parametrize("shape,roi,overlap", [((48, 40, 40), (32, 32, 16), 0.25), ((40, 48, 56), (32, 32, 32), 0.5),
                                               ((32, 32, 48), (32, 32, 16), 0.6), ((50, 61, 70), (24, 40, 20), 0.5),
                                               ((40, 70, 45), (19, 35, 21), 0.3)])
def test_sw_pass_clamped_windows_and_other_overlaps(eng_w, net, shape, roi, overlap):
    """Volumes that are NOT a multiple of the scan interval: the last window of a dimension is clamped back
    (MONAI dense_patch_slices), so windows of equal parity may overlap - the colour classes must still be
    race-free and the sums equal the oracle's.  The last two: windows that are not multiples of 16 (odd levels inside the
    U-Net: MaxPool3d's dropped plane, UpCat's replicate padding), one of them odd itself with an odd scan interval."""
    import torch
    from delivr_cfos_amd.synth import synth_volume_np
    from oracle import delivr_oracle as orc

    vol = synth_volume_np(shape, seed=31, dense=True)
    vol[:, :, : shape[2] // 3] = 0
    ref = np.zeros(shape, dtype=np.float32)
    cref = np.zeros(shape, dtype=np.uint8)
    info = orc.sliding_window_pass(vol, roi, lambda x: orc.unet_forward(net, x), ref, cref, overlap, None, 1, fp16=False)
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    st = eng_w.sw_infer(eng_w.make_sw_params(shape, roi, overlap, None, 0, "fp32", sw_batch=4), eng_w.to_device(vol), acc, cnt)
    eng_w.sync()
    assert st["n_windows"] == info["n_windows"] and st["n_skipped"] == info["n_skipped"]
    np.testing.assert_array_equal(cnt.cpu().numpy(), cref)
    assert np.abs(acc.cpu().numpy() - ref).max() < 2e-3
    np.testing.assert_array_equal(eng_w.window_starts(eng_w.make_sw_params(shape, roi, overlap)), orc.window_list(shape, roi, overlap))


# ---------------------------------------------------------------------------------------------------
# blob painting (blob_highlighter.py:108-125, :150-158)
# ---------------------------------------------------------------------------------------------------
def test_paint_matches_reference_golden(eng, golden_dir):
    import torch
    from delivr_cfos_amd.hostlogic import padded_boxes

    g = np.load(os.path.join(golden_dir, "ref_paint.npz"))
    m = g["mask"]
    keep = g["acronym"] != "bgr"
    ids = g["cc_id"][keep]
    bin_dev = torch.from_numpy(m).cuda()
    r, gr, b = eng.paint_boxes(bin_dev, padded_boxes(g["bounding_boxes"], ids, m.shape, 1),
                                  [g[k][keep].astype(np.uint8) for k in ("red", "green", "blue")])
    for c, img in enumerate((r, gr, b)):
        np.testing.assert_array_equal(img.cpu().numpy(), g["rgb"][c])
    (rid,) = eng.paint_boxes(bin_dev, padded_boxes(g["bounding_boxes"], ids, m.shape, 2), [g["graph_order"][keep].astype(np.uint16)])
    np.testing.assert_array_equal(rid.cpu().numpy(), g["region_id"])


def test_paint_random_boxes_vs_sequential_loop(eng):
    """Overlapping boxes in random order, a whole-volume box (own launch, > 65536 voxels) in the middle of the list,
    empty boxes, values that wrap in uint8 (bin_img holds 1s and 2s): bit-exact against the sequential loop."""
    import torch

    rng = np.random.default_rng(5)
    Z, Y, X = 40, 64, 96
    m = (rng.random((Z, Y, X)) < 0.3).astype(np.uint8) * rng.integers(1, 3, (Z, Y, X)).astype(np.uint8)
    n = 400
    lo = np.stack([rng.integers(0, Z, n), rng.integers(0, Y, n), rng.integers(0, X, n)], 1)
    ext = np.stack([rng.integers(0, 9, n), rng.integers(0, 12, n), rng.integers(0, 14, n)], 1)
    hi = np.minimum(lo + ext, [Z, Y, X])
    boxes = np.empty((n, 6), np.int32)
    boxes[:, 0::2], boxes[:, 1::2] = lo, hi
    boxes[200] = [0, Z, 0, Y, 0, X]
    v8 = rng.integers(0, 256, n).astype(np.uint8)
    v16 = rng.integers(0, 65536, n).astype(np.uint16)
    want8 = np.zeros(m.shape, np.uint8)
    want16 = np.zeros(m.shape, np.uint16)
    for b, a8, a16 in zip(boxes, v8, v16):
        sl = (slice(b[0], b[1]), slice(b[2], b[3]), slice(b[4], b[5]))
        want8[sl] = (m[sl].astype(np.int64) * int(a8)).astype(np.uint8)
        want16[sl] = (m[sl].astype(np.uint16) * a16)
    o8, o16 = eng.paint_boxes(torch.from_numpy(m).cuda(), boxes, [v8, v16])
    np.testing.assert_array_equal(o8.cpu().numpy(), want8)
    np.testing.assert_array_equal(o16.cpu().numpy(), want16)
    with pytest.raises(RuntimeError):
        bad = boxes.copy()
        bad[3, 1] = Z + 1
        eng.paint_boxes(torch.from_numpy(m).cuda(), bad, [v8])


def test_heatmap_bit_exact_vs_reference_and_scipy(eng, golden_dir):
    """create_heatmap on the device == the reference's own output (tests/golden/ref_atlas.npz), bit for bit; and on a
    grid with lines shorter than the kernel radius (reflect wraps more than once) == the oracle (scipy)."""
    from delivr_cfos_amd.cells_to_atlas import create_heatmap
    from oracle import delivr_oracle as orc

    g = np.load(os.path.join(golden_dir, "ref_atlas.npz"))
    cells = {k[4:]: g[k] for k in g.files if k.startswith("ccf_")}
    got = create_heatmap(cells, g["label"].shape, engine=eng)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got.view(np.uint32), g["heatmap"].view(np.uint32))
    rng = np.random.default_rng(2)
    shape = (5, 23, 3)
    n = 400
    c2 = {"x": rng.integers(0, shape[2], n), "y": rng.integers(0, shape[1], n), "z": rng.integers(0, shape[0], n)}
    for sigma in (2.25, 0.8):
        got = create_heatmap(c2, shape, engine=eng, sigma=sigma)
        np.testing.assert_array_equal(got.view(np.uint32), orc.heatmap(c2, shape, sigma).view(np.uint32))


# ---------------------------------------------------------------------------------------------------
# Gaussian importance-weighted blend (option; the reference always blends with constant weights)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,flip", [("fp32", None), ("fp32", 3), ("fp16", None), ("fp16", 2)])
def test_sw_pass_gaussian_blend_vs_oracle(eng_w, net, golden_dir, precision, flip):
    """acc += w*logit, wsum += w with MONAI's Gaussian importance map (oracle: the map built with torch's own conv, the
    reference loop with that map): weighted sums within 1e-3 (fp32 path) / 2 % of the logit scale (fp16 path), weight
    sums within 1e-5 where they matter, skipped windows contribute -1000*w; a uint8 count map is refused in this mode."""
    import torch
    from oracle import delivr_oracle as orc

    vol = _g(golden_dir, "ref_blend.npz")["volume"]
    roi = (32, 32, 16)
    imp = orc.gaussian_importance_map(roi, 0.125)
    acc_ref = np.zeros(vol.shape, dtype=np.float32)
    ws_ref = np.zeros(vol.shape, dtype=np.float32)
    info = orc.sliding_window_pass(vol, roi, lambda x: orc.unet_forward(net, x), acc_ref, ws_ref, 0.5, flip, sw_batch_size=1,
                                   fp16=False, importance=imp)
    v = eng_w.to_device(vol)
    acc = torch.zeros(vol.shape, dtype=torch.float32, device="cuda")
    ws = torch.zeros(vol.shape, dtype=torch.float32, device="cuda")
    p = eng_w.make_sw_params(vol.shape, roi, 0.5, flip, 0, precision, sw_batch=3, blend="gaussian", sigma_scale=0.125, wsum=ws)
    st = eng_w.sw_infer(p, v, acc)
    eng_w.sync()
    assert st["n_skipped"] == info["n_skipped"] > 0
    # the far tails (1e-10) come from 0.5*(erf(a) - erf(b)) in float32: a few 1e-4 relative between libm and torch there
    np.testing.assert_allclose(ws.cpu().numpy(), ws_ref, rtol=1e-3, atol=1e-9)
    core = ws_ref > 0.05
    np.testing.assert_allclose(ws.cpu().numpy()[core], ws_ref[core], rtol=1e-5)
    a = acc.cpu().numpy()
    live = acc_ref > -100
    if precision == "fp32":
        assert np.abs(a - acc_ref)[live].max() < 1e-3
        np.testing.assert_allclose(a[~live], acc_ref[~live], rtol=1e-5)
    else:
        rel = float(np.sqrt(np.mean((a - acc_ref)[live] ** 2)) / acc_ref[live].std())
        assert rel < 2e-2, rel
    # the mean logit differs from the constant-weight blend (the weighting is not a no-op) ...
    acc_c = torch.zeros(vol.shape, dtype=torch.float32, device="cuda")
    cnt_c = torch.zeros(vol.shape, dtype=torch.uint8, device="cuda")
    eng_w.sw_infer(eng_w.make_sw_params(vol.shape, roi, 0.5, flip, 0, precision, sw_batch=3), v, acc_c, cnt_c)
    mean_c = (acc_c / cnt_c.clamp(min=1)).cpu().numpy()
    mean_g = (acc / ws.clamp(min=1e-30)).cpu().numpy()
    assert np.abs(mean_c - mean_g)[live].max() > 1e-3
    # ... and the uint8 count map cannot hold fractional weights
    with pytest.raises(RuntimeError):
        eng_w.sw_infer(p, v, acc, cnt_c)


def test_window_maxima_one_pass_and_general_path_vs_numpy(eng):
    """dlv_sw_window_max_dev against numpy maxima of the reference's window list: cell-maxima path (window starts and
    ends on the 8-voxel grid along x), general path (interval 18: x boundaries off that grid), a clamped last window,
    a shard of the window list on a Z-slab."""
    import torch
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(21)
    for shape, roi, ov in (((96, 80, 128), (32, 32, 64), 0.5), ((70, 72, 108), (36, 36, 36), 0.5), ((64, 64, 72), (32, 32, 32), 0.25),
                           ((40, 48, 64), (40, 48, 64), 0.5)):
        vol = np.zeros(shape, dtype=np.uint16)
        idx = tuple(rng.integers(0, n, 40) for n in shape)
        vol[idx] = rng.integers(1, 65535, 40)
        vol[: shape[0] // 3] = 0  # some windows stay empty
        v = eng.to_device(vol)
        p = eng.make_sw_params(shape, roi, ov, None, 0, "fp16")
        wins = orc.window_list(shape, roi, ov)
        d, h, w = (min(r, n) for r, n in zip(roi, shape))
        want = np.array([vol[z:z + d, y:y + h, x:x + w].max() for z, y, x in wins], dtype=np.int32)
        np.testing.assert_array_equal(eng.window_max(p, v), want)
        if len(wins) >= 8:  # a shard: windows [3, n-2) on the planes they touch
            lo, hi = 3, len(wins) - 2
            z0 = int(wins[lo:hi, 0].min())
            z1 = int(wins[lo:hi, 0].max()) + d
            q = eng.make_sw_params(shape, roi, ov, None, 0, "fp16", win_range=(lo, hi), slab=(z0, z1 - z0))
            np.testing.assert_array_equal(eng.window_max(q, v[z0:z1].contiguous()), want[lo:hi])


@pytest.mark.parametrize("in_shape,out_shape", [((12, 20, 24), (48, 300, 352)), ((9, 14, 30), (33, 207, 451)), ((255, 9, 137), (1024, 130, 2048))])
def test_zoom_kernels_agree_bitwise(eng, in_shape, out_shape):
    """The three zoom kernels - row-organised (default), run-per-thread (diag switch "resample_run16"), one voxel per thread
    ("resample_simple") - on the same blobby mask incl. the production zoom factors (4, 15, 15) along a 2048-voxel row."""
    rng = np.random.default_rng(in_shape[2])
    m = (rng.random(in_shape) < 0.5).astype(np.uint8)
    m[:, : in_shape[1] // 2] = 1
    m[: in_shape[0] // 3, :, : in_shape[2] // 2] = 0
    src = eng.to_device(m)
    rows = eng.zoom_spline2_u8(src, out_shape).cpu().numpy()
    try:
        eng.diag_set("resample_run16", 1)
        run16 = eng.zoom_spline2_u8(src, out_shape).cpu().numpy()
        eng.diag_set("resample_run16", 0)
        eng.diag_set("resample_simple", 1)
        simple = eng.zoom_spline2_u8(src, out_shape).cpu().numpy()
    finally:
        eng.diag_set("resample_run16", 0)
        eng.diag_set("resample_simple", 0)
    assert np.array_equal(rows, simple) and np.array_equal(run16, simple)
    assert 0 < int(rows.sum()) < rows.size


@pytest.mark.parametrize("X", [96, 100])
def test_finalize_and_ccl_kernel_switches_give_identical_results(X):
    """The A/B switches of round 4 select kernels, never results: fused x+y distance pass vs separate passes
    ("erode_xy_split"), one-sweep z decision vs forward + backward sweeps ("erode_z_two_sweeps"), zero fill + listed
    chunks vs whole-volume label stores ("ccl_simple") - erosion radii 2 / 7 / 30, with and without z-blocks, rows that
    are / are not multiples of 8.  The switches are per context (dlv_diag_set): one engine per setting."""
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_np

    shape = (70, 90, X)
    vol = synth_volume_np(shape, seed=5, dense=True)
    vol[:, :7] = 0
    vol[20:50, 30:60, 25:70] = 0
    acc = np.random.default_rng(1).normal(0.2, 1.0, size=shape).astype(np.float32)
    cells = (vol > 3200).astype(np.uint8)

    def run(switches):
        eng = HipEngine(0)
        try:
            for k in switches:
                eng.diag_set(k, 1)
            v, a = eng.to_device(vol), eng.to_device(acc)
            out = {}
            for er in (2, 7, 30):
                for nb in (0, 24):
                    out[f"m_{er}_{nb}"] = eng.finalize(a, None, v, shape, 0.5, er, nb).cpu().numpy()
            lab, n = eng.ccl26(eng.to_device(cells))
            out["labels"] = lab.cpu().numpy()
            out["n"] = np.int64(n)
            return out
        finally:
            eng.close()

    base = run(())
    assert base["m_30_0"].any() and int(base["n"]) > 10
    for sw in (("erode_xy_split",), ("erode_z_two_sweeps",), ("erode_xy_split", "erode_z_two_sweeps"), ("ccl_simple",)):
        other = run(sw)
        for k in base:
            assert np.array_equal(other[k], base[k]), (sw, k)
