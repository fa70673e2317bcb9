"""Generate the committed fixtures under tests/golden/ (run ONLY in the build container).

    python -m oracle.make_goldens

TEST INFRASTRUCTURE.  Sources of truth:
  * ref_*  : produced by the reference's own code (inference/sliding_window_inferer.py,
             inference/inference.py:create_nifti_seg, count_blobs.py's CSV formatting) imported
             unchanged under oracle/ref_harness.py's stubs.
  * scipy_*: produced by scipy.ndimage (the reference's real dependency for zoom/erosion).
  * orc_*  : produced by oracle/delivr_oracle.py for the third-party pieces that cannot run here
             (MONAI U-Net -> torch.nn restatement, cc3d -> scipy.ndimage.label); these pin the
             HIP path to the oracle on the GPU box, they do not pin the oracle to the reference.
Fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import gzip
import io
import os
import struct
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import delivr_oracle as orc  # noqa: E402
from oracle import ref_harness  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def read_nifti_gz(path: str) -> np.ndarray:
    """Minimal NIfTI-1 reader (nibabel is absent): returns the array in file (x-fastest) order
    reshaped to (dim3, dim2, dim1)."""
    raw = gzip.open(path, "rb").read()
    dims = struct.unpack_from("<8h", raw, 40)
    dtype_code = struct.unpack_from("<h", raw, 70)[0]
    vox_offset = int(struct.unpack_from("<f", raw, 108)[0])
    dt = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
          768: np.uint32, 1024: np.int64, 1280: np.uint64}[dtype_code]
    n = dims[1] * dims[2] * dims[3]
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset)
    return arr.reshape(dims[3], dims[2], dims[1])


def det_predictor_torch(x):
    """Deterministic stand-in network for the host-logic goldens: (x - 2000) / 1000."""
    return (x - 2000.0) / 1000.0


def make_blend_volume(seed=3):
    rng = np.random.default_rng(seed)
    vol = rng.integers(500, 5000, size=(64, 64, 32)).astype(np.uint16)
    vol[:, :, 16:] = 0  # background half along X
    vol[40:, 40:, :] = 0  # plus a background corner
    return vol


def golden_tiler(swi):
    from monai.data.utils import dense_patch_slices  # the stub the reference itself calls

    cases = [((64, 64, 32), (32, 32, 16)), ((96, 96, 64), (96, 96, 64)), ((100, 70, 50), (32, 32, 16)),
             ((128, 128, 128), (64, 64, 64)), ((192, 96, 64), (96, 96, 64))]
    out = {}
    for i, (img, roi) in enumerate(cases):
        iv = swi._get_scan_interval(img, roi, 3, 0.5)
        sl = dense_patch_slices(img, roi, iv)
        starts = np.array([[s.start for s in w] for w in sl], dtype=np.int64)
        out[f"case{i}_image"] = np.array(img)
        out[f"case{i}_roi"] = np.array(roi)
        out[f"case{i}_interval"] = np.array(iv)
        out[f"case{i}_starts"] = starts
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(GOLD, "ref_tiler.npz"), **out)
    print("ref_tiler.npz", len(cases), "cases")


def golden_blend(swi):
    import torch

    vol = make_blend_volume()
    inp = vol[None, None]
    out = {"volume": vol}
    for tag, sched, bs in (("p1_b1", [None], 1), ("p1_b4", [None], 4), ("p13_b1", orc.pass_schedule(True), 1)):
        acc = torch.zeros(inp.shape, dtype=torch.float16)
        cnt = torch.zeros(inp.shape, dtype=torch.uint8)
        inferer = swi.SlidingWindowInferer(roi_size=(32, 32, 16), sw_batch_size=bs, overlap=0.5, mode="gaussian",
                                           padding_mode="replicate", sw_device=torch.device("cpu"),
                                           device=torch.device("cpu"))
        for k, flip in enumerate(sched):
            kw = dict(output_image=acc, count_map=cnt)
            if k > 0:
                kw["tta"] = True
            if flip is not None:
                kw["flip_dim"] = flip
            inferer(inp, det_predictor_torch, **kw)
        out[f"{tag}_sum"] = acc.numpy()[0, 0]
        out[f"{tag}_count"] = cnt.numpy()[0, 0]
    np.savez_compressed(os.path.join(GOLD, "ref_blend.npz"), **out)
    print("ref_blend.npz")


def golden_finalize(inf):
    """create_nifti_seg on random fp16 means + a raw volume with zero margins; second case forces
    >= 2 Arrayterator z-blocks by handing the (unchanged) reference function a numpy proxy whose
    Arrayterator ignores the hard-coded 1000**3 and uses a small buffer."""
    rng = np.random.default_rng(5)
    Z, Y, X = 40, 70, 66
    Zp, Yp, Xp = 48, 80, 80
    raw = np.zeros((1, 1, Zp, Yp, Xp), dtype=np.uint16)
    raw[0, 0, :Z, :Y, :X] = rng.integers(1, 4000, size=(Z, Y, X))
    raw[0, 0, :, :, :3] = 0  # 3 zero columns -> 33 zero columns after the L1-30 erosion
    raw[0, 0, 20:23, 30:50, 40:60] = 0  # an interior hole
    mean = (rng.standard_normal((1, 1, Zp, Yp, Xp)) * 2).astype(np.float16)
    out = {"raw": raw[0, 0], "mean": mean[0, 0], "stack_shape": np.array([1, 1, Z, Y, X])}
    real_np = inf.np
    for tag, buf in (("oneblock", None), ("blocks", 12 * Y * X)):
        if buf is not None:
            proxy = types.SimpleNamespace(**{k: getattr(real_np, k) for k in dir(real_np) if not k.startswith("__")})
            lib = types.SimpleNamespace(format=real_np.lib.format,
                                        Arrayterator=lambda a, _b, buf=buf: real_np.lib.Arrayterator(a, buf))
            proxy.lib = lib
            inf.np = proxy
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "binaries.npy")
            try:
                inf.create_nifti_seg(threshold=0.5, model_output=mean, output_file=f, network_output_file=None,
                                     dataset=raw, original_stack_shape=(1, 1, Z, Y, X))
            finally:
                inf.np = real_np
            with open(f, "rb") as fh:
                header = fh.read(128)
            out[f"{tag}_binaries"] = np.load(f)
            out[f"{tag}_header"] = np.frombuffer(header, dtype=np.uint8)
        out[f"{tag}_buf"] = np.array(buf if buf is not None else 1000**3)
    np.savez_compressed(os.path.join(GOLD, "ref_finalize.npz"), **out)
    print("ref_finalize.npz", {k: int(out[k].sum()) for k in ("oneblock_binaries", "blocks_binaries")})


def golden_ccl_and_csv():
    """CCL labels/stats from the oracle (scipy) on two of the reference's gt patches + adversarial
    shapes; CSV text produced by the reference's own count_blobs() with cc3d stubbed by the oracle
    (count_blobs.py:98-114 is the formatting code that is being pinned)."""
    import pickle

    td_root = os.path.join(ref_harness.REFERENCE_ROOT, "training_data", "cFos", "gt")
    out = {}
    names = ["patchvolume_1008_0.nii.gz", "patchvolume_1008_3.nii.gz"]
    for i, nm in enumerate(names):
        gt = (read_nifti_gz(os.path.join(td_root, nm)) > 0).astype(np.uint8)
        lab, n = orc.ccl26(gt)
        st = orc.cc_stats(lab, n)
        out[f"gt{i}_maskbits"] = np.packbits(gt.ravel())
        out[f"gt{i}_shape"] = np.array(gt.shape)
        out[f"gt{i}_n"] = np.array(n)
        out[f"gt{i}_labels"] = lab.astype(np.uint16)
        out[f"gt{i}_counts"] = st["voxel_counts"]
        out[f"gt{i}_bbox"] = st["bounding_boxes"]
        out[f"gt{i}_centroids"] = st["centroids"]
        print(nm, "components:", n)
    # adversarial: diagonal-only 26-links, single voxels, border-touching, a U shape that merges late
    adv = np.zeros((12, 16, 20), dtype=np.uint8)
    for k in range(8):
        adv[k, k, k] = 1  # pure 3-D diagonal chain
    adv[0, 15, 19] = 1
    adv[11, 0, 0] = 1
    adv[5, 2:14, 10] = 1
    adv[5, 2, 10:18] = 1
    adv[5, 13, 10:18] = 1
    adv[5, 3:13, 17] = 1  # ring
    adv[7, 6, 12] = 1
    adv[8, 7, 13] = 1  # 2-voxel diagonal
    adv[10:12, 10:12, 3:5] = 1
    lab, n = orc.ccl26(adv)
    st = orc.cc_stats(lab, n)
    out.update(adv_mask=adv, adv_n=np.array(n), adv_labels=lab.astype(np.uint16), adv_counts=st["voxel_counts"],
               adv_bbox=st["bounding_boxes"], adv_centroids=st["centroids"])
    np.savez_compressed(os.path.join(GOLD, "orc_ccl.npz"), **out)

    # ---- CSV through the reference's count_blobs() ----
    gt = np.unpackbits(out["gt0_maskbits"])[: 100**3].reshape(100, 100, 100)
    cc3d = types.ModuleType("cc3d")

    def connected_components(img, return_N=False, out_file=None):
        lab, n = orc.ccl26(np.asarray(img))
        return (lab, n) if return_N else lab

    cc3d.connected_components = connected_components
    cc3d.statistics = lambda labels, no_slice_conversion=False: orc.cc_stats(labels, int(labels.max()))
    sys.modules["cc3d"] = cc3d
    fh_stub = types.ModuleType("filehandling")
    fh_stub.read_nifti = fh_stub.write_nifti = None
    sys.modules["filehandling"] = fh_stub
    import importlib.util

    spec = importlib.util.spec_from_file_location("delivr_ref_count_blobs",
                                                  os.path.join(ref_harness.REFERENCE_ROOT, "count_blobs.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    with tempfile.TemporaryDirectory() as td:
        p_in = os.path.join(td, "in")
        p_out = os.path.join(td, "out") + "/"
        os.makedirs(os.path.join(p_in, "brainA", "binary_segmentations"))
        np.save(os.path.join(p_in, "brainA", "binary_segmentations", "binaries.npy"), gt.astype(np.uint8))
        settings = {"postprocessing": {"output_location": p_out}, "FLAGS": {"LOAD_ALL_RAM": True}}
        cb.count_blobs(settings, p_in, 0, "brainA", (1, 1, 100, 100, 100))
        files = sorted(os.listdir(p_out))
        csv_name = [f for f in files if f.endswith(".csv")][0]
        csv_text = open(os.path.join(p_out, csv_name)).read()
        with open(os.path.join(p_out, "brainA-stats.pickle"), "rb") as f:
            stats = pickle.load(f)
    np.savez_compressed(os.path.join(GOLD, "ref_csv.npz"), csv_name=np.array(csv_name),
                        csv_text=np.array(csv_text), files=np.array(files),
                        n=np.array(int(out["gt0_n"])), stat_keys=np.array(sorted(stats.keys())))
    print("ref_csv.npz", csv_name, len(csv_text), "bytes", files)


def golden_unet():
    """Seeded random-weight U-Net logits (fp32, torch CPU) for a 32^3 and a 48x32x16 patch,
    per-parameter checksums so the GPU box can verify that the seed reproduces the weights, and
    per-stage activations checksums."""
    import torch

    from delivr_cfos_amd.synth import synth_volume_np

    net = orc.build_unet(seed=0)
    orc.randomize_affine(net, seed=1)
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == orc.N_PARAMS, n_params
    vol = synth_volume_np((64, 64, 64), seed=7, dense=True)
    x32 = vol[16:48, 16:48, 16:48].astype(np.float32)[None, None]
    x_odd = vol[8:56, 16:48, 24:40].astype(np.float32)[None, None]  # 48x32x16
    out = {"x32": vol[16:48, 16:48, 16:48], "x_odd": vol[8:56, 16:48, 24:40],
           "logits32": orc.unet_forward(net, x32)[0, 0], "logits_odd": orc.unet_forward(net, x_odd)[0, 0]}
    names, sums, abss = [], [], []
    for k, v in net.state_dict().items():
        names.append(k)
        sums.append(float(v.double().sum()))
        abss.append(float(v.double().abs().sum()))
    out.update(param_names=np.array(names), param_sum=np.array(sums), param_abs=np.array(abss))
    # stage activations (mean, std) to localise a mismatch
    acts = {}
    hooks = []
    for nm, m in net.named_modules():
        if nm and nm.count(".") == 0:
            hooks.append(m.register_forward_hook(lambda mod, i, o, nm=nm: acts.__setitem__(nm, o)))
    with torch.no_grad():
        net(torch.as_tensor(x32))
    for h in hooks:
        h.remove()
    out["stage_names"] = np.array(list(acts.keys()))
    out["stage_mean"] = np.array([float(a.double().mean()) for a in acts.values()])
    out["stage_std"] = np.array([float(a.double().std()) for a in acts.values()])
    np.savez_compressed(os.path.join(GOLD, "orc_unet.npz"), **out)
    print("orc_unet.npz logits32 mean/std", out["logits32"].mean(), out["logits32"].std(),
          "frac>=0", (out["logits32"] >= 0).mean())


def golden_unet_c1():
    """BASELINE config 1 / SURVEY 8(d) C1: one 64^3 patch x = randn(1,1,64,64,64, seed 0)*100 + 500 through the
    seeded random-weight oracle U-Net (torch fp32 CPU).  The full logits are 1 MB; the fixture keeps every 4th voxel
    per axis (16^3 samples) plus the global mean/std, enough to pin the oracle across machines and to check the HIP
    path at this shape without re-running the oracle."""
    import torch

    net = orc.build_unet(seed=0)
    orc.randomize_affine(net, seed=1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 1, 64, 64, 64), generator=g) * 100.0 + 500.0
    lg = orc.unet_forward(net, x.numpy())[0, 0]
    np.savez_compressed(os.path.join(GOLD, "orc_unet_c1.npz"), logits_s4=lg[::4, ::4, ::4].copy(),
                        mean=np.array(float(lg.astype(np.float64).mean())), std=np.array(float(lg.astype(np.float64).std())),
                        frac_pos=np.array(float((lg >= 0).mean())), x_checksum=np.array(float(x.double().sum())))
    print("orc_unet_c1.npz mean/std", lg.mean(), lg.std(), "frac>=0", (lg >= 0).mean())


def golden_resample():
    rng = np.random.default_rng(11)
    vol = rng.integers(0, 65535, size=(9, 31, 47)).astype(np.uint16)
    bm = orc.block_mean_u16(vol, (4, 15, 15))
    mask = (rng.random((7, 9, 11)) < 0.5).astype(np.uint8)
    zo = orc.zoom_spline2_u8(mask, (26, 40, 37))
    np.savez_compressed(os.path.join(GOLD, "scipy_resample.npz"), bm_in=vol, bm_factors=np.array([4, 15, 15]),
                        bm_out=bm, zoom_in=mask, zoom_out=zo)
    print("scipy_resample.npz")


def main():
    os.makedirs(GOLD, exist_ok=True)
    swi, inf = ref_harness.load_reference_inference()
    golden_tiler(swi)
    golden_blend(swi)
    golden_finalize(inf)
    golden_ccl_and_csv()
    golden_unet()
    golden_unet_c1()
    golden_resample()
    golden_swc()




def golden_swc():
    """SWC files written by the reference's own rewrite_swc (automate_mBrainaligner.py:75-197), imported
    under stubs (tifffile, more_itertools absent), from the CSV golden."""
    import importlib.util

    g = np.load(os.path.join(GOLD, "ref_csv.npz"))
    for name in ("tifffile",):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "more_itertools" not in sys.modules:
        mi = types.ModuleType("more_itertools")

        def sliced(seq, n):
            for i in range(0, len(seq), n):
                yield seq[i:i + n]

        mi.sliced = sliced
        sys.modules["more_itertools"] = mi
    spec = importlib.util.spec_from_file_location("delivr_ref_mba", os.path.join(ref_harness.REFERENCE_ROOT, "automate_mBrainaligner.py"))
    mba = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mba)
    out = {}
    # the reference pins pandas 1.4.3 (requirements.txt), where Series.str.replace defaults to regex=True;
    # pandas 2.x defaults to regex=False, which makes the reference's re.escape()'d patterns literal.
    import pandas as pd
    from pandas.core.strings.accessor import StringMethods

    real_replace = StringMethods.replace

    def replace_pd14(self, pat, repl, n=-1, case=None, flags=0, regex=True):
        return real_replace(self, pat, repl, n=n, case=case, flags=flags, regex=regex)

    StringMethods.replace = replace_pd14
    with tempfile.TemporaryDirectory() as td:
        csv_path = os.path.join(td, str(g["csv_name"]))
        open(csv_path, "w").write(str(g["csv_text"]))
        single = mba.rewrite_swc(csv_path, td)
        out["single_name"] = np.array(os.path.basename(single[0]))
        out["single_text"] = np.array(open(single[0]).read())
        real_cpu = os.cpu_count
        os.cpu_count = lambda: 5  # 4 chunks, deterministic
        try:
            chunks = mba.rewrite_swc(csv_path, td, parallel_processing=True)
        finally:
            os.cpu_count = real_cpu
        out["chunk_names"] = np.array([os.path.basename(c) for c in chunks])
        out["chunk_texts"] = np.array([open(c).read() for c in chunks])
        out["params"] = np.array(mba.split_parameters(csv_path))
    StringMethods.replace = real_replace
    np.savez_compressed(os.path.join(GOLD, "ref_swc.npz"), **out)
    print("ref_swc.npz", out["single_name"], len(str(out["single_text"])), "bytes;", len(out["chunk_names"]), "chunks")


def golden_paint():
    """R/G/B and region-id images painted by the reference's own blob_highlighter (blob_highlighter.py:38-169),
    imported under stubs (cc3d, cv2, tifffile, skimage, filehandling, matplotlib absent; tifffile.imwrite captured),
    for a small volume with touching / nested bounding boxes, 'bgr' rows and non-listed blobs."""
    import importlib.util
    import pickle

    import pandas as pd

    captured = {}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    saved = {k: sys.modules.get(k) for k in ("cc3d", "cv2", "tifffile", "filehandling", "skimage", "skimage.morphology",
                                             "skimage.draw", "skimage.io", "inference", "inference.inference",
                                             "matplotlib", "matplotlib.pyplot", "blob_depthmap")}
    stub("cc3d")
    stub("cv2")
    stub("tifffile", imwrite=lambda path, arr, **kw: captured.__setitem__(os.path.basename(path), np.array(arr)))
    stub("filehandling", read_nifti=None, write_nifti=None)
    sk = stub("skimage")
    sk.morphology = stub("skimage.morphology", binary_dilation=None)
    sk.draw = stub("skimage.draw", ellipsoid=None)
    sk.io = stub("skimage.io")
    inf_pkg = stub("inference")
    inf_pkg.inference = stub("inference.inference", create_empty_memmap=None)
    mpl = stub("matplotlib")
    mpl.pyplot = stub("matplotlib.pyplot")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_harness.REFERENCE_ROOT, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    load("blob_depthmap")
    bh = load("blob_highlighter")

    rng = np.random.default_rng(21)
    Z, Y, X = 14, 20, 26
    m = np.zeros((Z, Y, X), dtype=np.uint8)
    for _ in range(16):
        z, y, x = rng.integers(1, Z - 3), rng.integers(1, Y - 4), rng.integers(1, X - 4)
        m[z:z + rng.integers(1, 3), y:y + rng.integers(1, 4), x:x + rng.integers(1, 4)] = 1
    m[5, 2:18, 3] = 1          # a long thin blob whose box swallows neighbours
    m[13, 19, 25] = 1          # a voxel in the far corner (pad_bb stops at the border)
    labels, n = orc.ccl26(m)
    stats = orc.cc_stats(labels, n)
    ids = list(range(1, n + 1))
    order = [int(v) for v in rng.permutation(ids)]
    order = order[:-2]                                 # two blobs are not listed (ids must be unique: the reference's
                                                       # second loop has no try/except around a duplicated id's broadcast error)
    acr = ["bgr" if i % 7 == 3 else "CTX" for i in range(len(order))]
    df = pd.DataFrame({"connected_component_id": order, "acronym": acr,
                       "red": rng.integers(0, 256, len(order)), "green": rng.integers(0, 256, len(order)),
                       "blue": rng.integers(0, 256, len(order)), "graph_order": rng.integers(1, 1300, len(order))})
    brain = "brainA"
    with tempfile.TemporaryDirectory() as td:
        d_bin = os.path.join(td, "bin") + "/"
        d_csv = os.path.join(td, "csv") + "/"
        d_out = os.path.join(td, "out")
        d_cache = os.path.join(td, "cache")
        d_post = os.path.join(td, "post")
        for d in (os.path.join(d_bin, brain, "binary_segmentations"), d_csv, d_out, d_cache, d_post):
            os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d_bin, brain, "binary_segmentations", "binaries.npy"), m)
        df.to_csv(os.path.join(d_csv, f"cells_{brain}.csv"))
        with open(os.path.join(d_post, f"{brain}-stats.pickle"), "wb") as fh:
            pickle.dump({k: v.copy() for k, v in stats.items()}, fh)
        settings = {"visualization": {"input_prediction_location": d_bin, "input_csv_location": d_csv, "output_location": d_out,
                                      "cache_location": d_cache, "no_atlas_depthmap": False, "region_id_rgb": True,
                                      "region_id_grayvalues": True},
                    "postprocessing": {"output_location": d_post}, "FLAGS": {"LOAD_ALL_RAM": True}}
        bh.blob_highlighter(settings, (brain, ""), (1, 1, Z, Y, X))
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    rgb = np.stack([np.stack([captured[f"{brain}rgb_C0{c}_z{str(z).zfill(4)}.tif"] for z in range(Z)]) for c in range(3)])
    rid = np.stack([captured[f"region_id_{str(z).zfill(4)}.tif"] for z in range(Z)])
    np.savez_compressed(os.path.join(GOLD, "ref_paint.npz"), mask=m, bounding_boxes=stats["bounding_boxes"],
                        cc_id=np.array(order), acronym=np.array(acr), red=df["red"].to_numpy(), green=df["green"].to_numpy(),
                        blue=df["blue"].to_numpy(), graph_order=df["graph_order"].to_numpy(), rgb=rgb, region_id=rid,
                        tiff_names=np.array(sorted(captured)))
    print("ref_paint.npz", rgb.shape, rid.shape, "painted voxels", int((rgb.sum(0) > 0).sum()), "of", int(m.sum()))


def golden_atlas():
    """Outputs of the reference's own mbrainaligner_atlas_to_ccf and create_heatmap (cells_to_atlas.py:114-151,
    :174-200), imported under stubs (nibabel, tifffile absent), for synthetic cells on a small label grid; the
    RegionID gather of cells_to_atlas (:204-212) restated next to them (it needs the ontology XML only for the join)."""
    import importlib.util

    import pandas as pd

    saved = {k: sys.modules.get(k) for k in ("nibabel", "tifffile")}
    for k in saved:
        if saved[k] is None:
            sys.modules[k] = types.ModuleType(k)
    spec = importlib.util.spec_from_file_location("delivr_ref_c2a", os.path.join(ref_harness.REFERENCE_ROOT, "cells_to_atlas.py"))
    c2a = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(c2a)
    rng = np.random.default_rng(9)
    label = rng.integers(0, 40, size=(36, 50, 44)).astype(np.uint16)      # (z, y, x) like the CCF3 annotation volume
    label[rng.random(label.shape) < 0.3] = 0
    n = 3000
    # atlas-space coordinates as mBrainAligner returns them (floats), some outside the grid after the transform
    cells = pd.DataFrame({"connected_component_id": np.arange(n), "x": rng.uniform(236, 266, n), "y": rng.uniform(136, 162, n),
                          "z": rng.uniform(-2, 20, n), "Size": rng.integers(1, 200, n)})
    cells.loc[:40, ["x", "y", "z"]] = np.round(cells.loc[:40, ["x", "y", "z"]]) + 0.25   # ties after *2: round-half-even
    raw = {c: cells[c].to_numpy().copy() for c in cells.columns}
    out = c2a.mbrainaligner_atlas_to_ccf(cells.copy(), label)
    region = label[out["z"].to_list(), out["y"].to_list(), out["x"].to_list()].astype(np.int64)
    region[region != 0] += 1
    # pandas 1.4.3 (requirements.txt): DataFrame.value_counts() gives an unnamed Series, so reset_index() names the
    # count column 0 - which create_heatmap indexes (:186).  pandas 2.x names it 'count'; restore the old name.
    real_vc = pd.DataFrame.value_counts

    def vc_pd14(self, *a, **k):
        s = real_vc(self, *a, **k)
        s.name = None
        return s

    pd.DataFrame.value_counts = vc_pd14
    try:
        heat = c2a.create_heatmap(out, label)
    finally:
        pd.DataFrame.value_counts = real_vc
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
    np.savez_compressed(os.path.join(GOLD, "ref_atlas.npz"), label=label, **{"raw_" + k: v for k, v in raw.items()},
                        **{"ccf_" + c: out[c].to_numpy() for c in out.columns}, region_id=region, heatmap=heat)
    print("ref_atlas.npz cells kept", len(out), "of", n, "heat", heat.dtype, heat.shape, float(heat.sum()))


def golden_tiff():
    """TIFF planes written by libtiff (through Pillow) - LZW with and without the horizontal predictor, one and many
    strips, 8 and 16 bit - plus a hand-written big-endian uncompressed plane, with the pixel arrays they hold."""
    import struct

    from PIL import Image

    rng = np.random.default_rng(4)
    a = (np.cumsum(rng.integers(-3, 4, (97, 131)), axis=1) + 2000).clip(0, 65535).astype(np.uint16)
    a[10:20, 30:60] = rng.integers(0, 65535, (10, 30))
    Image.fromarray(a).save(os.path.join(GOLD, "tiff_lzw16.tif"), compression="tiff_lzw")
    Image.fromarray(a).save(os.path.join(GOLD, "tiff_lzw16_pred.tif"), compression="tiff_lzw", tiffinfo={317: 2})
    Image.fromarray(a).save(os.path.join(GOLD, "tiff_lzw16_strips.tif"), compression="tiff_lzw", tiffinfo={278: 10})
    b = rng.integers(0, 256, (180, 200)).astype(np.uint8)      # > 4094 codes: the table fills up and is reset
    b[50:120] = 7                                              # long runs: KwKwK codes
    Image.fromarray(b).save(os.path.join(GOLD, "tiff_lzw8.tif"), compression="tiff_lzw")
    c = rng.integers(0, 65535, (9, 13)).astype(np.uint16)
    data = c.astype(">u2").tobytes()
    tags = [(256, 3, 1, 13), (257, 3, 1, 9), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1),
            (278, 3, 1, 9), (279, 4, 1, len(data))]
    with open(os.path.join(GOLD, "tiff_be16.tif"), "wb") as fh:
        fh.write(b"MM" + struct.pack(">HI", 42, 8 + len(data)))
        fh.write(data)
        fh.write(struct.pack(">H", len(tags)))
        for tag, typ, cnt, val in tags:
            fh.write(struct.pack(">HHI", tag, typ, cnt) + (struct.pack(">HH", val, 0) if typ == 3 else struct.pack(">I", val)))
        fh.write(struct.pack(">I", 0))
    np.savez_compressed(os.path.join(GOLD, "tiff_expected.npz"), lzw16=a, lzw8=b, be16=c)
    for f in ("tiff_lzw16.tif", "tiff_lzw16_pred.tif", "tiff_lzw16_strips.tif", "tiff_lzw8.tif", "tiff_be16.tif"):
        im = Image.open(os.path.join(GOLD, f))
        t = dict(im.tag_v2)
        print(f, os.path.getsize(os.path.join(GOLD, f)), "bytes", im.size, "compression", t.get(259), "predictor", t.get(317),
              "rows/strip", t.get(278), "strips", len(t.get(273, ())))
        assert (np.array(im) == {"tiff_lzw8.tif": b, "tiff_be16.tif": c}.get(f, a)).all()


def golden_tiff_variants():
    """The TIFF variants libtiff-based readers (the reference's cv2 / tifffile / skimage) accept beyond LZW strips: deflate
    (with and without the predictor) and BigTIFF written by libtiff through Pillow; tiled planes (uncompressed, and deflate
    with the predictor; edge tiles padded) written by hand here - Pillow cannot write tiles - and read back through libtiff
    before they are kept."""
    import struct
    import zlib

    from PIL import Image

    rng = np.random.default_rng(14)
    a = (np.cumsum(rng.integers(-3, 4, (70, 50)), axis=1) + 3000).clip(0, 65535).astype(np.uint16)
    a[20:30, 10:40] = rng.integers(0, 65535, (10, 30))
    b = rng.integers(0, 256, (45, 67)).astype(np.uint8)
    Image.fromarray(a).save(os.path.join(GOLD, "tiff_deflate16.tif"), compression="tiff_adobe_deflate")
    Image.fromarray(a).save(os.path.join(GOLD, "tiff_deflate16_pred.tif"), compression="tiff_adobe_deflate", tiffinfo={317: 2, 278: 16})
    Image.fromarray(b).save(os.path.join(GOLD, "tiff_big8.tif"), big_tiff=True)  # (Pillow writes BigTIFF uncompressed only)

    def write_bigtiff_deflate(path, img, rows_per_strip):
        H, W = img.shape
        strips = [zlib.compress(img[r:r + rows_per_strip].astype("<u2").tobytes(), 6) for r in range(0, H, rows_per_strip)]
        off = 16
        offs = []
        for t in strips:
            offs.append(off)
            off += len(t) + (-len(t) % 8)
        tab = off                                   # StripOffsets (LONG8), StripByteCounts (LONG8)
        ifd = tab + 16 * len(strips)
        tags = [(256, 4, 1, W), (257, 4, 1, H), (258, 3, 1, 16), (259, 3, 1, 8), (262, 3, 1, 1), (273, 16, len(strips), tab),
                (277, 3, 1, 1), (278, 4, 1, rows_per_strip), (279, 16, len(strips), tab + 8 * len(strips))]
        with open(path, "wb") as fh:
            fh.write(b"II" + struct.pack("<HHHQ", 43, 8, 0, ifd))
            for t in strips:
                fh.write(t + b"\0" * (-len(t) % 8))
            fh.write(struct.pack("<%dQ" % len(strips), *offs))
            fh.write(struct.pack("<%dQ" % len(strips), *[len(t) for t in strips]))
            fh.write(struct.pack("<Q", len(tags)))
            for tag, typ, cnt, val in tags:
                fh.write(struct.pack("<HHQQ", tag, typ, cnt, val))
            fh.write(struct.pack("<Q", 0))

    write_bigtiff_deflate(os.path.join(GOLD, "tiff_big16_deflate.tif"), a, 24)

    def write_tiled(path, img, tw, th, compression, predictor):
        bps = img.dtype.itemsize
        H, W = img.shape
        nx, ny = (W + tw - 1) // tw, (H + th - 1) // th
        tiles = []
        for ty in range(ny):
            for tx in range(nx):
                t = np.zeros((th, tw), dtype=img.dtype)
                part = img[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
                t[:part.shape[0], :part.shape[1]] = part
                if predictor == 2:
                    t = np.concatenate([t[:, :1], np.diff(t.astype(np.int64), axis=1).astype(img.dtype)], axis=1)
                raw = t.astype("<u%d" % bps).tobytes()
                tiles.append(zlib.compress(raw, 6) if compression == 8 else raw)
        off = 8
        offs = []
        for t in tiles:
            offs.append(off)
            off += len(t) + (len(t) & 1)
        tab_off = off                      # TileOffsets, then TileByteCounts (LONG arrays)
        ifd = tab_off + 8 * len(tiles)
        tags = [(256, 3, 1, W), (257, 3, 1, H), (258, 3, 1, 8 * bps), (259, 3, 1, compression), (262, 3, 1, 1), (277, 3, 1, 1),
                (317, 3, 1, predictor), (322, 3, 1, tw), (323, 3, 1, th), (324, 4, len(tiles), tab_off),
                (325, 4, len(tiles), tab_off + 4 * len(tiles))]
        with open(path, "wb") as fh:
            fh.write(b"II" + struct.pack("<HI", 42, ifd))
            for t in tiles:
                fh.write(t + (b"\0" if len(t) & 1 else b""))
            fh.write(struct.pack("<%dI" % len(tiles), *offs))
            fh.write(struct.pack("<%dI" % len(tiles), *[len(t) for t in tiles]))
            fh.write(struct.pack("<H", len(tags)))
            for tag, typ, cnt, val in tags:
                if cnt == 1 and typ == 3:
                    fh.write(struct.pack("<HHIHH", tag, typ, cnt, val, 0))
                elif cnt == 1:
                    fh.write(struct.pack("<HHII", tag, typ, cnt, offs[0] if tag == 324 else (len(tiles[0]) if tag == 325 else val)))
                else:
                    fh.write(struct.pack("<HHII", tag, typ, cnt, val))
            fh.write(struct.pack("<I", 0))

    write_tiled(os.path.join(GOLD, "tiff_tiled16.tif"), a, 32, 32, 1, 1)
    write_tiled(os.path.join(GOLD, "tiff_tiled16_deflate_pred.tif"), a, 16, 48, 8, 2)
    write_tiled(os.path.join(GOLD, "tiff_tiled8.tif"), b, 64, 16, 8, 1)
    np.savez_compressed(os.path.join(GOLD, "tiff_expected_variants.npz"), a16=a, b8=b)
    for f, want in (("tiff_deflate16.tif", a), ("tiff_deflate16_pred.tif", a), ("tiff_big16_deflate.tif", a), ("tiff_big8.tif", b),
                    ("tiff_tiled16.tif", a), ("tiff_tiled16_deflate_pred.tif", a), ("tiff_tiled8.tif", b)):
        im = Image.open(os.path.join(GOLD, f))
        t = dict(im.tag_v2)
        print(f, os.path.getsize(os.path.join(GOLD, f)), "bytes", im.size, "compression", t.get(259), "predictor", t.get(317),
              "tile", t.get(322), t.get(323), "magic", open(os.path.join(GOLD, f), "rb").read(4))
        assert (np.array(im) == want).all(), f   # libtiff decodes the fixture to the expected pixels


if __name__ == "__main__" and os.environ.get("DELIVR_GOLDEN_ONLY") == "tiff_variants":
    golden_tiff_variants()
elif __name__ == "__main__" and os.environ.get("DELIVR_GOLDEN_ONLY") == "unet_c1":
    golden_unet_c1()
elif __name__ == "__main__" and os.environ.get("DELIVR_GOLDEN_ONLY") == "tiff":
    golden_tiff()
elif __name__ == "__main__" and os.environ.get("DELIVR_GOLDEN_ONLY") == "atlas":
    golden_atlas()
elif __name__ == "__main__" and os.environ.get("DELIVR_GOLDEN_ONLY") == "paint":
    golden_paint()
elif __name__ == "__main__" and os.environ.get("DELIVR_GOLDEN_ONLY") == "swc":
    golden_swc()
elif __name__ == "__main__":
    main()
