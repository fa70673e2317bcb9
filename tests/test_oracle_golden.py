"""The CPU oracle against the committed golden vectors (tests/golden/, made by
oracle/make_goldens.py from the reference's own code / scipy).  No GPU needed."""
import os

import numpy as np
import pytest

from oracle import delivr_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_tiler_matches_reference(golden_dir):
    g = _load(golden_dir, "ref_tiler.npz")
    for i in range(int(g["n_cases"])):
        img, roi = tuple(g[f"case{i}_image"]), tuple(g[f"case{i}_roi"])
        assert orc.scan_interval(img, roi, 0.5) == tuple(g[f"case{i}_interval"])
        np.testing.assert_array_equal(orc.window_list(img, roi, 0.5), g[f"case{i}_starts"])


def _det(x):
    return (x - 2000.0) / 1000.0


@pytest.mark.parametrize("tag,bs,tta", [("p1_b1", 1, False), ("p1_b4", 4, False), ("p13_b1", 1, True)])
def test_blend_matches_reference(golden_dir, tag, bs, tta):
    g = _load(golden_dir, "ref_blend.npz")
    vol = g["volume"]
    acc = np.zeros(vol.shape, dtype=np.float16)
    cnt = np.zeros(vol.shape, dtype=np.uint8)
    for flip in orc.pass_schedule(tta):
        orc.sliding_window_pass(vol, (32, 32, 16), _det, acc, cnt, 0.5, flip, sw_batch_size=bs, fp16=True)
    np.testing.assert_array_equal(cnt, g[f"{tag}_count"])
    np.testing.assert_array_equal(acc.view(np.uint16), g[f"{tag}_sum"].view(np.uint16))  # bit-exact fp16


def test_per_batch_skip_differs_from_per_tile(golden_dir):
    """D7: the reference decides the background skip per sw-batch; the build adopts per tile."""
    g = _load(golden_dir, "ref_blend.npz")
    assert not np.array_equal(g["p1_b1_sum"], g["p1_b4_sum"])


@pytest.mark.parametrize("tag", ["oneblock", "blocks"])
def test_finalize_matches_reference(golden_dir, tag):
    g = _load(golden_dir, "ref_finalize.npz")
    Z, Y, X = (int(v) for v in g["stack_shape"][2:])
    out = orc.finalize(g["mean"], None, g["raw"], (Z, Y, X), 0.5, 30, buf_size=int(g[f"{tag}_buf"]))
    np.testing.assert_array_equal(out, g[f"{tag}_binaries"])
    # 3 zero columns -> 33 zero columns (L1-radius-30 erosion)
    assert out[:, :, :33].sum() == 0
    hdr = bytes(g[f"{tag}_header"])
    assert len(hdr) == 128 and hdr[:6] == b"\x93NUMPY" and hdr.endswith(b"\n")


def test_l1_distance_form_equals_scipy_erosion():
    rng = np.random.default_rng(0)
    m = (rng.random((20, 40, 50)) > 0.002).astype(np.uint8)
    for r in (1, 5, 30):
        np.testing.assert_array_equal(orc.l1_distance_keep(m, r), orc.erode_l1(m, r))


def test_ccl_goldens_and_invariants(golden_dir):
    g = _load(golden_dir, "orc_ccl.npz")
    for key in ("gt0", "gt1"):
        shape = tuple(g[f"{key}_shape"])
        mask = np.unpackbits(g[f"{key}_maskbits"])[: int(np.prod(shape))].reshape(shape)
        lab, n = orc.ccl26(mask)
        assert n == int(g[f"{key}_n"])
        np.testing.assert_array_equal(lab, g[f"{key}_labels"])
        # numbering rule: label k's first voxel (raster order) precedes label k+1's
        first = np.full(n + 1, lab.size, dtype=np.int64)
        idx = np.nonzero(lab.ravel())[0]
        np.minimum.at(first, lab.ravel()[idx], idx)
        assert np.all(np.diff(first[1:]) > 0)
        st = orc.cc_stats(lab, n)
        np.testing.assert_array_equal(st["voxel_counts"], g[f"{key}_counts"])
        np.testing.assert_array_equal(st["bounding_boxes"], g[f"{key}_bbox"])
        np.testing.assert_allclose(st["centroids"][1:], g[f"{key}_centroids"][1:], rtol=0, atol=0)
    lab, n = orc.ccl26(g["adv_mask"])
    assert n == int(g["adv_n"])
    np.testing.assert_array_equal(lab, g["adv_labels"])
    # diagonal chain is ONE component under 26-connectivity
    assert len(set(lab[k, k, k] for k in range(8))) == 1


def test_csv_format_matches_reference(golden_dir):
    g = _load(golden_dir, "ref_csv.npz")
    c = _load(golden_dir, "orc_ccl.npz")
    shape = tuple(c["gt0_shape"])
    mask = np.unpackbits(c["gt0_maskbits"])[: int(np.prod(shape))].reshape(shape)
    lab, n = orc.ccl26(mask)
    txt = orc.cells_csv_text(orc.cc_stats(lab, n), n)
    assert txt == str(g["csv_text"])
    assert str(g["csv_name"]) == "(100, 100, 100)_brainA.csv"
    assert txt.count("\n") == n  # header + rows 1..N-1 (last label dropped, count_blobs.py:104)


def test_unet_seed_reproduces_and_logits(golden_dir):
    g = _load(golden_dir, "orc_unet.npz")
    net = orc.build_unet(seed=0)
    orc.randomize_affine(net, seed=1)
    assert sum(p.numel() for p in net.parameters()) == orc.N_PARAMS
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["param_names"]]
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["param_sum"], rtol=1e-12, atol=1e-12)
    out = orc.unet_forward(net, g["x32"].astype(np.float32)[None, None])[0, 0]
    np.testing.assert_allclose(out, g["logits32"], rtol=1e-4, atol=1e-4)


def test_upcat_pads_an_odd_level_at_the_far_end_with_the_edge_value():
    """MONAI's UpCat (basic_unet.py, is_pad=True - the default the reference's BasicUNet(...) call leaves in place,
    inference/inference.py:190-197) replicate-pads the up-sampled tensor by ONE voxel at the FAR end of every dimension in which
    the skip tensor is odd; MaxPool3d(2) floors.  The oracle's UpCat against that statement written out with numpy (np.pad
    mode="edge" on the right only), and the whole network on a window with odd levels: finite, right shape, and changed by a
    change of the last plane of the input (the plane MaxPool3d drops at level 0 still feeds conv_0 and the skip connection)."""
    import torch
    from oracle import delivr_oracle as orc

    net = orc.build_unet(seed=3)
    up = net.upcat_4
    x = torch.randn(1, 256, 1, 2, 3)
    x_e = torch.randn(1, 128, 3, 5, 7)  # odd in every dimension: 2 * (1, 2, 3) = (2, 4, 6) is one short everywhere
    with torch.no_grad():
        got = up(x, x_e).numpy()
        x0 = up.upsample(x).numpy()
        assert x0.shape == (1, 128, 2, 4, 6)
        padded = np.pad(x0, ((0, 0), (0, 0), (0, 1), (0, 1), (0, 1)), mode="edge")
        want = up.convs(torch.cat([x_e, torch.from_numpy(padded)], dim=1)).numpy()
    np.testing.assert_array_equal(got, want)
    assert np.array_equal(padded[..., -1], padded[..., -2]) and np.array_equal(padded[:, :, -1], padded[:, :, -2])
    # even skip tensor: no padding at all
    x_e2 = torch.randn(1, 128, 2, 4, 6)
    with torch.no_grad():
        np.testing.assert_array_equal(up(x, x_e2).numpy(), up.convs(torch.cat([x_e2, up.upsample(x)], dim=1)).numpy())
    vol = np.random.default_rng(0).integers(200, 4000, size=(1, 1, 35, 21, 50)).astype(np.float32)
    out = orc.unet_forward(net, vol)
    assert out.shape == vol.shape and np.isfinite(out).all()
    vol2 = vol.copy()
    vol2[0, 0, -1] += 500.0
    assert np.abs(orc.unet_forward(net, vol2) - out).max() > 0


def test_resamplers(golden_dir):
    g = _load(golden_dir, "scipy_resample.npz")
    np.testing.assert_array_equal(orc.block_mean_u16(g["bm_in"], tuple(g["bm_factors"])), g["bm_out"])
    # integer form == float mean then truncate
    v = g["bm_in"].astype(np.int64)
    f = tuple(int(x) for x in g["bm_factors"])
    Zp, Yp, Xp = (-(-n // k) * k for n, k in zip(v.shape, f))
    pad = np.zeros((Zp, Yp, Xp), dtype=np.int64)
    pad[: v.shape[0], : v.shape[1], : v.shape[2]] = v
    s = pad.reshape(Zp // f[0], f[0], Yp // f[1], f[1], Xp // f[2], f[2]).sum(axis=(1, 3, 5))
    np.testing.assert_array_equal((s // (f[0] * f[1] * f[2])).astype(np.uint16), g["bm_out"])
    z = orc.zoom_spline2_f64(g["zoom_in"], g["zoom_out"].shape)
    mine = np.floor(z + 0.5).astype(np.uint8)
    ties = np.abs(z - 0.5) < 1e-9
    assert np.array_equal(mine[~ties], g["zoom_out"][~ties])
    np.testing.assert_array_equal(orc.zoom_spline2_u8(g["zoom_in"], g["zoom_out"].shape), g["zoom_out"])


def test_arrayterator_block_rule():
    for shape, buf in (((40, 70, 66), 12 * 70 * 66), ((10, 7, 5), 1000**3), ((9, 4, 5), 41), ((9, 4, 5), 20)):
        a = np.zeros(shape, dtype=np.uint8)
        blocks = [b.shape for b in np.lib.Arrayterator(a, buf)]
        axis, nb = orc.zblock_planes(shape, buf)
        if axis == "z":
            assert blocks[0] == (min(nb, shape[0]),) + shape[1:]


def test_paint_restatement_and_box_logic_match_reference_blob_highlighter(golden_dir):
    """The R/G/B and region-id images the reference's blob_highlighter wrote (captured under stubs by
    oracle/make_goldens.py::golden_paint) == the oracle's sequential loop == painting hostlogic.padded_boxes in order
    (the second loop sees boxes that pad_bb already moved once)."""
    from delivr_cfos_amd.hostlogic import padded_boxes

    g = np.load(os.path.join(golden_dir, "ref_paint.npz"))
    m = g["mask"]
    Z, Y, X = m.shape
    keep = g["acronym"] != "bgr"
    ids = g["cc_id"][keep]
    bb = None
    for c, key in enumerate(("red", "green", "blue")):
        bb = g["bounding_boxes"].copy()
        img = orc.paint_blobs(m, bb, ids, g[key][keep], (1, 1, Z, Y, X), np.uint8)
        np.testing.assert_array_equal(img, g["rgb"][c])
    rid = orc.paint_blobs(m, bb, ids, g["graph_order"][keep], (1, 1, Z, Y, X), np.uint16)   # bb: padded once already
    np.testing.assert_array_equal(rid, g["region_id"])

    def paint(boxes, vals, dt):
        out = np.zeros(m.shape, dt)
        for b, v in zip(boxes, vals):
            sl = (slice(b[0], b[1]), slice(b[2], b[3]), slice(b[4], b[5]))
            out[sl] = (m[sl].astype(np.int64) * int(v)).astype(dt)
        return out

    np.testing.assert_array_equal(paint(padded_boxes(g["bounding_boxes"], ids, (Z, Y, X), 1), g["red"][keep], np.uint8), g["rgb"][0])
    np.testing.assert_array_equal(paint(padded_boxes(g["bounding_boxes"], ids, (Z, Y, X), 2), g["graph_order"][keep], np.uint16),
                                  g["region_id"])


def test_atlas_transform_and_heatmap_match_reference(golden_dir):
    """mbrainaligner_atlas_to_ccf / create_heatmap of the reference (tests/golden/ref_atlas.npz) == the oracle == the
    host mirror's coordinate logic (the heat map itself is a device kernel: tests/test_gpu_parity.py)."""
    from delivr_cfos_amd.cells_to_atlas import mbrainaligner_atlas_to_ccf, region_ids

    g = np.load(os.path.join(golden_dir, "ref_atlas.npz"))
    raw = {k[4:]: g[k] for k in g.files if k.startswith("raw_")}
    want = {k[4:]: g[k] for k in g.files if k.startswith("ccf_")}
    label = g["label"]
    for fn in (orc.atlas_to_ccf, mbrainaligner_atlas_to_ccf):
        got = fn(raw, label.shape)
        assert set(got) == set(want)
        for k in want:
            np.testing.assert_array_equal(got[k], want[k])
    np.testing.assert_array_equal(region_ids(want, label), g["region_id"])
    np.testing.assert_array_equal(orc.heatmap(want, label.shape), g["heatmap"])


def test_unet_c1_64cube_reproduces_the_committed_samples(golden_dir):
    """BASELINE config 1 (one 64^3 patch, seeded input and weights): the oracle on this machine reproduces the samples
    committed from the build container (torch CPU kernels differ in summation order between thread counts: 2e-5)."""
    import torch

    g = _load(golden_dir, "orc_unet_c1.npz")
    net = orc.build_unet(seed=0)
    orc.randomize_affine(net, seed=1)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn((1, 1, 64, 64, 64), generator=gen) * 100.0 + 500.0
    assert abs(float(x.double().sum()) - float(g["x_checksum"])) < 1e-3
    lg = orc.unet_forward(net, x.numpy())[0, 0]
    np.testing.assert_allclose(lg[::4, ::4, ::4], g["logits_s4"], atol=2e-5, rtol=0)
    assert abs(float(lg.astype(np.float64).std()) - float(g["std"])) < 1e-5


def test_trilinear_and_affine_restatements_properties():
    """North-star extensions without a reference counterpart: self-consistency of the fp64 restatements the
    HIP kernels are held to - identity, integer translation with zero fill, constant volumes, 2x of a linear ramp."""
    rng = np.random.default_rng(0)
    v = rng.integers(0, 65535, size=(6, 7, 8)).astype(np.uint16)
    assert np.array_equal(orc.trilinear_u16(v, v.shape), v)
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    assert np.array_equal(orc.affine_warp_u16(v, ident, v.shape), v)
    w = orc.affine_warp_u16(v, [1, 0, 0, 1, 0, 1, 0, -2, 0, 0, 1, 0], v.shape)
    assert np.array_equal(w[:5, 2:, :], v[1:, :5, :]) and w[5].sum() == 0 and w[:, :2].sum() == 0
    c = np.full((4, 5, 6), 1234, dtype=np.uint16)
    assert (orc.trilinear_u16(c, (9, 11, 13)) == 1234).all()
    ramp = (np.arange(8, dtype=np.uint16) * 100)[None, None, :].repeat(2, 0).repeat(2, 1)
    up = orc.trilinear_u16(ramp, (2, 2, 16))[0, 0]
    assert list(up[1:15]) == [25 + 50 * i for i in range(14)] and up[0] == 0 and up[15] == 700  # clamp to edge
    # half-voxel shift: the average of two neighbours, rounded half up
    h = orc.affine_warp_u16(ramp, [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.5], ramp.shape)[0, 0]
    assert list(h[:7]) == [50 + 100 * i for i in range(7)] and h[7] == 350  # last: (700 + 0) / 2
