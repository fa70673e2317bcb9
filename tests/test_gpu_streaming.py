"""Volumes larger than the HBM budget: the slabs of the shard plan run one after another on the one device
(delivr_cfos_amd/streaming.py).  The reference streams through memmaps (inference/inference.py:240-247, :285-299;
count_blobs.py:59-64) and has no size limit; here the resident run is the yardstick:

  * run_inference with settings["mi355x"]["stream_slabs"] / a small "hbm_budget_gb" against the resident run: the same
    mask wherever the mean logit is not at fp32 rounding level of the threshold (the seam sums associate differently),
    identical network_output.npy up to 1e-6, and - on the trained-like checkpoint, whose logits keep a margin - the
    bit-identical mask, labels, statistics and CSV through count_blobs, itself streamed in 4 slabs;
  * a budget nothing fits: MemoryError that names the sizes, not an allocator trace.
"""
import os
import pickle

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


def _settings(root, crop, **mi):
    return {"blob_detection": {"window_dimensions": {"window_dim_0": crop[0], "window_dim_1": crop[1], "window_dim_2": crop[2]}},
            "postprocessing": {"output_location": str(root / "03") + "/"},
            "mi355x": {"precision": "fp16", **mi}, "FLAGS": {"SAVE_ACTIVATED_OUTPUT": True}}


def _volume():
    from delivr_cfos_amd.synth import synth_volume_np

    vol = synth_volume_np((150, 70, 90), seed=17, dense=True)
    vol[:, :9] = 0
    vol[60:95, 30:50, 40:70] = 0  # a hole: the erosion works across slab seams and z-blocks
    return vol


@pytest.mark.parametrize("tta", [False, True], ids=["1pass", "tta"])
def test_streamed_inference_equals_the_resident_run(tmp_path, capsys, tta):
    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.weights import random_state_dict

    crop = (32, 32, 32)
    vol = _volume()
    nifti = os.path.join(str(tmp_path), "masked_nifti.npy")
    _write_padded_npy(nifti, vol, crop)
    sd = {"state_dict": random_state_dict(7)}
    res = {}
    for tag, mi in (("resident", {}), ("streamed", {"stream_slabs": 4})):
        out = run_inference([nifti], str(tmp_path / tag), (1, 1) + vol.shape, comment="b", tta=tta, state_dict=sd,
                            settings=_settings(tmp_path / tag, crop, **mi))
        txt = capsys.readouterr().out
        assert ("streaming 4 Z-slabs" in txt) == (tag == "streamed"), txt
        res[tag] = (np.load(os.path.join(out, "binary_segmentations", "binaries.npy")),
                    np.load(os.path.join(out, "binary_segmentations", "network_output.npy")))
    (m0, p0), (m1, p1) = res["resident"], res["streamed"]
    assert m0.shape == vol.shape and m0.any()
    assert float(np.abs(p1 - p0).max()) < 1e-5
    sure = np.abs(p0 - 0.5) > 1e-5
    assert np.array_equal(m1[sure], m0[sure])
    assert int((m1 != m0).sum()) <= 3


def test_streamed_pipeline_is_bit_identical_on_the_trained_like_checkpoint(tmp_path, capsys):
    """budget-driven: hbm_budget_gb small enough that neither the volume nor the mask + labels fit"""
    from delivr_cfos_amd.count_blobs import count_blobs
    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.streaming import forward_workspace_bytes
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import trained_like_state_dict

    crop = (64, 64, 64)
    vol = synth_volume_np((256, 128, 128), seed=23)  # with the ellipsoid background: skipped windows, a brain surface
    sd = {"state_dict": trained_like_state_dict(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_like_weights.npz"))}
    fixed_gb = forward_workspace_bytes(crop, "fp16") / 2**30
    outs = {}
    for tag, mi in (("resident", {}), ("streamed", {"hbm_budget_gb": fixed_gb + 0.028})):  # 28.7 MiB beside the workspace: the resident run needs 36 MiB
        root = tmp_path / tag
        nifti_dir = root / "01" / "b" / "masked_niftis"
        os.makedirs(nifti_dir)
        _write_padded_npy(str(nifti_dir / "masked_nifti.npy"), vol, crop)
        st = _settings(root, crop, **mi)
        st["FLAGS"]["SAVE_ACTIVATED_OUTPUT"] = False
        run_inference([str(nifti_dir / "masked_nifti.npy")], str(root / "02") + "/", (1, 1) + vol.shape, comment="b", tta=False,
                      state_dict=sd, settings=st)
        txt = capsys.readouterr().out
        assert ("exceeds the HBM budget" in txt) == (tag == "streamed"), txt
        if tag == "streamed":
            st["mi355x"]["hbm_budget_gb"] = 0.012  # 12 MiB: mask + labels + scratch need 52 MiB
        N = count_blobs(st, str(root / "02"), 1, "b", (1, 1) + vol.shape)
        txt = capsys.readouterr().out
        assert ("Z-slabs" in txt) == (tag == "streamed"), txt
        post = st["postprocessing"]["output_location"]
        outs[tag] = {"mask": np.load(os.path.join(str(root / "02"), "b", "binary_segmentations", "binaries.npy")), "N": N,
                     "labels": np.load(os.path.join(post, f"b-{N}-cc3d.npy")),
                     "stats": pickle.load(open(os.path.join(post, "b-stats.pickle"), "rb")),
                     "csv": open(post + f"{vol.shape}_b.csv").read(),
                     "files": sorted(os.listdir(post))}
    a, b = outs["resident"], outs["streamed"]
    assert a["N"] > 20, a["N"]
    assert np.array_equal(a["mask"], b["mask"])
    assert a["N"] == b["N"] and a["labels"].dtype == b["labels"].dtype and np.array_equal(a["labels"], b["labels"])
    for k in ("voxel_counts", "bounding_boxes"):
        assert np.array_equal(a["stats"][k], b["stats"][k]), k
    assert np.array_equal(a["stats"]["centroids"], b["stats"]["centroids"], equal_nan=True)
    assert a["csv"] == b["csv"]
    assert a["files"] == b["files"]  # (the scratch file of the streamed labelling is gone, the label file carries its final name)
    # a second call finds the cached labelling; with the statistics removed they are recomputed from the CACHED labels - slab by
    # slab under the small budget (stats_streamed) - and equal the first run's
    st = _settings(tmp_path / "streamed", crop, hbm_budget_gb=0.012)
    out_dir = st["postprocessing"]["output_location"]
    os.remove(os.path.join(out_dir, "b-stats.pickle"))
    N2 = count_blobs(st, str(tmp_path / "streamed" / "02"), 1, "b", (1, 1) + vol.shape)
    txt = capsys.readouterr().out
    assert N2 == b["N"] and "Cached brain found" in txt, txt
    again = pickle.load(open(os.path.join(out_dir, "b-stats.pickle"), "rb"))
    for k in ("voxel_counts", "bounding_boxes"):
        assert np.array_equal(again[k], b["stats"][k]), k
    assert np.array_equal(again["centroids"], b["stats"]["centroids"], equal_nan=True)


def test_a_budget_nothing_fits_is_reported_with_sizes(tmp_path):
    from delivr_cfos_amd.count_blobs import count_blobs
    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.weights import random_state_dict

    crop = (32, 32, 32)
    vol = _volume()
    nifti = os.path.join(str(tmp_path), "masked_nifti.npy")
    _write_padded_npy(nifti, vol, crop)
    with pytest.raises(MemoryError) as ei:
        run_inference([nifti], str(tmp_path / "o"), (1, 1) + vol.shape, comment="b", tta=False,
                      state_dict={"state_dict": random_state_dict(7)}, settings=_settings(tmp_path, crop, hbm_budget_gb=0.5))
    assert "DLV_ENOMEM" in str(ei.value) and "GiB" in str(ei.value) and "hbm_budget_gb" in str(ei.value), str(ei.value)
    root = tmp_path / "02" / "b" / "binary_segmentations"
    os.makedirs(root)
    np.save(str(root / "binaries.npy"), (vol > 2500).astype(np.uint8))
    with pytest.raises(MemoryError) as ei:
        count_blobs(_settings(tmp_path, crop, hbm_budget_gb=1e-5), str(tmp_path / "02"), 1, "b", (1, 1) + vol.shape)
    assert "DLV_ENOMEM" in str(ei.value), str(ei.value)
