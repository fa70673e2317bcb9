"""Host-side logic of the package on the CPU: integer rules, file formats, the shard plan and the
world_size-2 seam exchange over gloo.  The compute engine is replaced by the oracle here (tests may);
the product path itself never runs without the HIP library + a GPU."""
import os
import struct
import sys

import numpy as np
import pytest

from delivr_cfos_amd import hostlogic
from delivr_cfos_amd.parallel import make_plan
from oracle import delivr_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arrayterator_zblock_matches_numpy():
    for shape, buf in (((40, 70, 66), 12 * 70 * 66), ((10, 7, 5), 1000**3), ((9, 4, 5), 41), ((64, 8, 8), 64 * 5 + 3)):
        blocks = [b.shape[0] for b in np.lib.Arrayterator(np.zeros(shape, np.uint8), buf)]
        zb = hostlogic.arrayterator_zblock(shape, buf)
        assert blocks[0] == min(zb, shape[0]), (shape, buf, blocks, zb)
    assert hostlogic.arrayterator_zblock((1024, 2048, 2048)) == 238
    with pytest.raises(NotImplementedError):
        hostlogic.arrayterator_zblock((4, 40000, 40000))


def test_pass_schedule_collapses_the_13_reference_passes():
    ref = orc.pass_schedule(True)
    assert len(ref) == 13
    mine = hostlogic.pass_schedule(True)
    assert sum(r for _, r in mine) == 13
    for flip, rep in mine:
        assert ref.count(flip) == rep
    assert hostlogic.pass_schedule(False) == [(None, 1)]


def test_padded_shape_and_ratios():
    assert hostlogic.padded_shape((100, 70, 50), (32, 32, 16)) == (128, 96, 64) == orc.padded_shape((100, 70, 50), (32, 32, 16))
    steps = {"original_um_x": 1.62, "original_um_y": 1.62, "original_um_z": 6.0, "downsample_um_x": 25.0,
             "downsample_um_y": 25.0, "downsample_um_z": 25.0}
    assert hostlogic.downsample_ratios(steps) == (4, 15, 15)
    c = hostlogic.scale_cell_coords([[10.0, 30.0, 45.0]], (1024, 2048, 2048), (256, 137, 137), "down")
    np.testing.assert_allclose(c, orc.scale_coords([[10.0, 30.0, 45.0]], (1024, 2048, 2048), (256, 137, 137)))
    np.testing.assert_allclose(hostlogic.scale_cell_coords(c, (1024, 2048, 2048), (256, 137, 137), "up"), [[10.0, 30.0, 45.0]])


def test_csv_text_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_csv.npz"))
    c = np.load(os.path.join(golden_dir, "orc_ccl.npz"))
    stats = {"voxel_counts": c["gt0_counts"], "centroids": c["gt0_centroids"]}
    assert hostlogic.cells_csv_text(stats, int(c["gt0_n"])) == str(g["csv_text"])
    assert hostlogic.csv_name(tuple(c["gt0_shape"]), "brainA.nii.gz") == str(g["csv_name"])


def test_random_state_dict_fits_the_architecture():
    from delivr_cfos_amd.weights import random_state_dict

    net = orc.build_unet(seed=None)
    sd = {k.replace("module.", ""): v for k, v in random_state_dict(3).items()}
    net.load_state_dict(sd, strict=True)
    assert sum(v.numel() for v in sd.values()) == orc.N_PARAMS


def test_shard_plan_is_a_partition():
    for shape, roi, world in (((1024, 2048, 2048), (128, 128, 128), 8), ((512, 512, 512), (128, 128, 128), 4),
                              ((64, 64, 32), (32, 32, 16), 2), ((128, 64, 64), (64, 64, 64), 3), ((64, 64, 64), (64, 64, 64), 2)):
        starts = orc.window_list(shape, roi, 0.5)
        for weights in (None, np.random.default_rng(world).choice([1.0, 0.02], size=len(starts), p=[0.6, 0.4])):
            plan = make_plan(starts, roi[0], shape[0], world, weights)
            if weights is not None and len(starts) >= 4 * world:  # balanced to within ~one window
                loads = [weights[b:e].sum() for b, e in plan.win_ranges]
                assert max(loads) - min(loads) <= 2.0, loads
            assert plan.win_ranges[0][0] == 0 and plan.win_ranges[-1][1] == len(starts)
            assert all(plan.win_ranges[r][1] == plan.win_ranges[r + 1][0] for r in range(world - 1))
            assert plan.z_owned[0][0] == 0 and plan.z_owned[-1][1] == shape[0]
            assert all(plan.z_owned[r][1] == plan.z_owned[r + 1][0] for r in range(world - 1))
            # every plane a rank computed is either owned by it or sent to exactly one owner
            for r in range(world):
                lo, hi = plan.z_computed[r]
                covered = np.zeros(shape[0], dtype=int)
                olo, ohi = plan.z_owned[r]
                covered[max(lo, olo):min(hi, ohi)] += 1
                for dst, a, b in plan.sends(r):
                    covered[a:b] += 1
                    assert (r, a, b) in plan.recvs(dst)
                assert np.all(covered[lo:hi] == 1)


def test_shard_plan_of_the_c_abi_equals_the_python_plan_and_slabs_cover_what_a_rank_needs():
    """dlv_shard_plan_make / dlv_shard_slab (host-only entry points of the C ABI: what examples/c_host.c --gpus N uses)
    against parallel.make_plan / ShardPlan.slab, for equal and weighted windows, clamped last windows, more ranks than
    tile rows and empty ranks."""
    import ctypes as C

    from delivr_cfos_amd import _lib
    from delivr_cfos_amd.parallel import plan_from_params

    lib = _lib.load()
    rng = np.random.default_rng(7)
    cases = (((1024, 2048, 2048), (128, 128, 128), 0.5, 8), ((512, 512, 512), (128, 128, 128), 0.5, 4), ((64, 64, 32), (32, 32, 16), 0.5, 2),
             ((128, 64, 64), (64, 64, 64), 0.5, 3), ((64, 64, 64), (64, 64, 64), 0.5, 2), ((100, 70, 50), (32, 32, 16), 0.25, 5),
             ((96, 96, 128), (96, 96, 64), 0.5, 4), ((256, 64, 64), (32, 32, 32), 0.6, 16), ((40, 48, 56), (32, 32, 32), 0.5, 7))
    for shape, roi, ov, world in cases:
        p = _lib.SwParams()
        p.Zp, p.Yp, p.Xp = shape
        for k in range(3):
            p.roi[k] = roi[k]
        p.overlap = ov
        p.flip_dim = -1
        n = C.c_int64()
        assert lib.dlv_sw_num_windows(C.byref(p), C.byref(n)) == 0
        starts = np.zeros((n.value, 3), dtype=np.int64)
        assert lib.dlv_sw_window_starts(C.byref(p), starts.ctypes.data_as(C.POINTER(C.c_int64)), n.value) == 0
        np.testing.assert_array_equal(starts, orc.window_list(shape, roi, ov))
        for weights in (None, rng.choice([1.0, 0.02], size=n.value, p=[0.6, 0.4]).astype(np.float32), np.zeros(n.value, dtype=np.float32)):
            a = plan_from_params(p, world, weights)
            b = make_plan(starts, roi[0], shape[0], world, None if weights is None else weights.astype(np.float64))
            assert a.win_ranges == b.win_ranges and a.z_computed == b.z_computed and a.z_owned == b.z_owned, (shape, roi, world)
            plan_c = _lib.ShardPlanC()
            wp = None if weights is None else weights.ctypes.data_as(C.POINTER(C.c_float))
            assert lib.dlv_shard_plan_make(C.byref(p), world, wp, C.byref(plan_c)) == 0
            for Z, er, nb in ((shape[0], 30, 0), (shape[0] - 3, 30, 17), (shape[0], 5, 40)):
                for r in range(world):
                    z0, nz = C.c_int(), C.c_int()
                    assert lib.dlv_shard_slab(C.byref(plan_c), r, Z, er, nb, C.byref(z0), C.byref(nz)) == 0
                    lo, hi = a.slab(r, Z, er, nb)
                    assert (z0.value, z0.value + nz.value) == (lo, hi)
                    clo, chi = a.z_computed[r]
                    olo, ohi = a.z_owned[r][0], min(a.z_owned[r][1], Z)
                    if chi > clo:
                        assert lo <= clo and hi >= chi
                    if ohi > olo:  # owned planes + erosion margin inside their z-blocks
                        blk = nb if nb > 0 else Z
                        assert lo <= max(olo - er, (olo // blk) * blk) and hi >= min(ohi + er, ((ohi - 1) // blk + 1) * blk, Z)
    bad = _lib.SwParams()
    assert lib.dlv_shard_plan_make(C.byref(bad), 2, None, C.byref(_lib.ShardPlanC())) != 0  # empty volume
    assert lib.dlv_shard_plan_make(C.byref(p), 99, None, C.byref(_lib.ShardPlanC())) != 0  # too many ranks


def _gloo_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from delivr_cfos_amd.parallel import exchange_seams, gather_slabs, make_plan
    from oracle import delivr_oracle as orc2

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    vol = rng.integers(1, 4000, size=(96, 32, 32)).astype(np.uint16)
    vol[:, :, 20:] = 0
    roi = (32, 32, 16)
    starts = orc2.window_list(vol.shape, roi, 0.5)
    wts = np.array([1.0 if vol[z:z + 32, y:y + 32, x:x + 16].max() > 0 else 0.02 for z, y, x in starts])
    plan = make_plan(starts, roi[0], vol.shape[0], world, wts)
    det = lambda x: (x - 2000.0) / 1000.0  # noqa: E731
    # slab-resident: this rank holds only the planes [slo, shi) of the accumulator (and would of the volume)
    slo, shi = plan.slab(rank, vol.shape[0], 0, 0)
    acc = np.zeros((shi - slo,) + vol.shape[1:], dtype=np.float32)
    wb, we = plan.win_ranges[rank]
    for z, y, x in starts[wb:we]:
        win = vol[z:z + 32, y:y + 32, x:x + 16].astype(np.float32)
        acc[z - slo:z - slo + 32, y:y + 32, x:x + 16] += det(win) if win.max() > 0 else -1000.0
    t = torch.from_numpy(acc)
    exchange_seams(t, plan, rank, dist, z0=slo)
    lo, hi = plan.z_owned[rank]
    slab = (t[lo - slo:hi - slo] >= 0).to(torch.uint8)
    out = torch.zeros(vol.shape, dtype=torch.uint8) if rank == 0 else None
    gather_slabs(slab, plan, rank, dist, out=out)
    if rank == 0:
        np.save(os.path.join(tmp, "mask.npy"), out.numpy())
        np.save(os.path.join(tmp, "acc0.npy"), t[lo - slo:hi - slo].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pass_equals_single_rank_over_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_gloo_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    vol = rng.integers(1, 4000, size=(96, 32, 32)).astype(np.uint16)
    vol[:, :, 20:] = 0
    ref = np.zeros(vol.shape, dtype=np.float32)
    orc.sliding_window_pass(vol, (32, 32, 16), lambda x: (x - 2000.0) / 1000.0, ref, None, 0.5, None, 1, fp16=False)
    mask = np.load(tmp_path / "mask.npy")
    np.testing.assert_array_equal(mask, (ref >= 0).astype(np.uint8))


class _OracleCclEngine:
    """Stand-in for HipEngine in the CPU run of ccl_sharded (tests may use the oracle; the product may not): the
    four device operations of the seam merge restated with numpy on CPU tensors."""

    def __init__(self, orc2):
        self.orc = orc2

    def ccl26(self, mask):
        import torch
        lab, n = self.orc.ccl26(mask.numpy())
        return torch.from_numpy(lab.astype(np.int32)), n

    def seam_pairs(self, a, b):
        a, b = a.numpy(), b.numpy()
        Y, X = a.shape
        out = set()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ya0, ya1 = max(0, -dy), min(Y, Y - dy)
                xa0, xa1 = max(0, -dx), min(X, X - dx)
                aa = a[ya0:ya1, xa0:xa1]
                bb = b[ya0 + dy:ya1 + dy, xa0 + dx:xa1 + dx]
                m = (aa > 0) & (bb > 0)
                out.update(zip(aa[m].tolist(), bb[m].tolist()))
        return np.array(sorted(out), dtype=np.uint32).reshape(-1, 2)

    def relabel(self, labels, lut):
        import torch
        labels.copy_(torch.from_numpy(lut.astype(np.int64)[labels.numpy()].astype(np.int32)))

    def cc_stats_raw(self, labels, n):
        lab = labels.numpy()
        counts = np.bincount(lab.ravel(), minlength=n + 1).astype(np.uint32)
        bbmin = np.full((n + 1, 3), 0xFFFFFFFF, dtype=np.uint32)
        bbmax = np.zeros((n + 1, 3), dtype=np.uint32)
        sums = np.zeros((n + 1, 3), dtype=np.uint64)
        idx = np.indices(lab.shape)
        for k in range(3):
            c = idx[k].ravel().astype(np.int64)
            sums[:, k] = np.bincount(lab.ravel(), weights=c.astype(np.float64), minlength=n + 1).astype(np.uint64)
            lo = np.full(n + 1, 0xFFFFFFFF, dtype=np.int64)
            hi = np.zeros(n + 1, dtype=np.int64)
            np.minimum.at(lo, lab.ravel(), c)
            np.maximum.at(hi, lab.ravel(), c)
            bbmin[:, k], bbmax[:, k] = lo, hi
        sums[0] = 0
        return {"counts": counts, "bbmin": bbmin, "bbmax": bbmax, "sums": sums}


def _ccl_mask(seed=3):
    rng = np.random.default_rng(seed)
    m = (rng.random((37, 24, 20)) < 0.22).astype(np.uint8)
    m[10:30, 5, 5] = 1          # a column crossing every seam
    m[:, 20:, :] = 0
    m[18] = 0                   # an empty plane: components must NOT merge across it
    m[25:, 0, :] = 1
    return m


def _gloo_ccl_worker(rank, world, port, tmp, cuts):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from delivr_cfos_amd.parallel import ccl_sharded
    from oracle import delivr_oracle as orc2

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _ccl_mask()
    slabs = [(cuts[r], cuts[r + 1]) for r in range(world)]
    lo, hi = slabs[rank]
    slab = torch.from_numpy(m[lo:hi].copy()) if hi > lo else None
    labels, n, stats = ccl_sharded(_OracleCclEngine(orc2), slab, slabs, rank, dist, m.shape)
    if labels is not None:
        np.save(os.path.join(tmp, f"labels_{rank}.npy"), labels.numpy())
    if rank == 0:
        np.savez(os.path.join(tmp, "stats.npz"), n=n, **stats)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cuts", [(0, 19, 37), (0, 11, 11, 37), (0, 5, 18, 37), (0, 1, 36, 37)])
def test_sharded_ccl_equals_single_volume_over_gloo(tmp_path, cuts):
    """Labels and statistics of the slab-wise labelling + seam merge are those of the whole volume, for 2 and 3
    ranks, an empty slab, a cut next to an empty plane and one-plane slabs."""
    import torch.multiprocessing as mp

    world = len(cuts) - 1
    port = 31500 + (os.getpid() % 2000) + 7 * world + cuts[1]
    mp.spawn(_gloo_ccl_worker, args=(world, port, str(tmp_path), cuts), nprocs=world, join=True)
    m = _ccl_mask()
    ref, n = orc.ccl26(m)
    out = np.zeros(m.shape, dtype=np.int64)
    for r in range(world):
        if cuts[r + 1] > cuts[r]:
            out[cuts[r]:cuts[r + 1]] = np.load(tmp_path / f"labels_{r}.npy")
    np.testing.assert_array_equal(out, ref.astype(np.int64))
    st = np.load(tmp_path / "stats.npz")
    assert int(st["n"]) == n
    want = orc.cc_stats(ref, n)
    np.testing.assert_array_equal(st["voxel_counts"], want["voxel_counts"])
    np.testing.assert_array_equal(st["bounding_boxes"], want["bounding_boxes"])
    np.testing.assert_array_equal(st["centroids"], want["centroids"])


def _write_tiff(path, arr):
    h, w = arr.shape
    data = arr.astype("<u2").tobytes()
    tags = [(256, 3, 1, w), (257, 3, 1, h), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8),
            (277, 3, 1, 1), (278, 3, 1, h), (279, 4, 1, len(data))]
    ifd_off = 8 + len(data)
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, ifd_off))
        fh.write(data)
        fh.write(struct.pack("<H", len(tags)))
        for tag, typ, cnt, val in tags:
            fh.write(struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val)))
        fh.write(struct.pack("<I", 0))


def test_tiff_header_reader_and_get_real_size(tmp_path):
    from delivr_cfos_amd.downsample.downsample_and_mask import get_real_size, read_tiff_plane

    rng = np.random.default_rng(1)
    planes = [rng.integers(0, 65535, size=(7, 11)).astype(np.uint16) for _ in range(3)]
    for i, p in enumerate(planes):
        _write_tiff(tmp_path / f"Z{i:04d}.tif", p)
    assert get_real_size(str(tmp_path)) == (3, 7, 11)
    np.testing.assert_array_equal(read_tiff_plane(str(tmp_path / "Z0001.tif")), planes[1])


def test_product_modules_never_import_the_oracle():
    """The oracle is test infrastructure: nothing under delivr_cfos_amd/ may reference it."""
    pkg = os.path.join(ROOT, "delivr_cfos_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), encoding="utf-8", errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)


def test_device_api_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from delivr_cfos_amd.engine import HipEngine

    with pytest.raises(RuntimeError):
        HipEngine(0)


def test_swc_writer_matches_reference_golden(golden_dir, tmp_path):
    """the CSV -> SWC rewrite the atlas step performs (automate_mBrainaligner.py:75-197),
    pinned by the reference's own function (run under pandas-1.4 regex semantics)."""
    from delivr_cfos_amd.swc import rewrite_swc, sampling_factors, split_parameters

    g = np.load(os.path.join(golden_dir, "ref_csv.npz"))
    s = np.load(os.path.join(golden_dir, "ref_swc.npz"))
    p = os.path.join(tmp_path, str(g["csv_name"]))
    open(p, "w").write(str(g["csv_text"]))
    one = rewrite_swc(p, str(tmp_path))
    assert os.path.basename(one[0]) == str(s["single_name"])
    assert open(one[0]).read() == str(s["single_text"])
    chunks = rewrite_swc(p, str(tmp_path), parallel_processing=True, n_chunks=4)
    assert [os.path.basename(c) for c in chunks] == [str(x) for x in s["chunk_names"]]
    for c, t in zip(chunks, s["chunk_texts"]):
        assert open(c).read() == str(t)
    assert split_parameters(p) == [int(v) for v in s["params"]]
    assert sampling_factors(p, (25, 10, 10)) == (10.0, 10.0, 4.0)


def test_native_tiff_reader_decodes_libtiff_planes(golden_dir):
    """csrc/tiffio.hip (host code in libdelivr_hip.so) against planes written by libtiff: LZW, LZW + horizontal
    predictor, many strips, 8 bit with table resets, big-endian uncompressed; unsupported files are refused loudly."""
    from delivr_cfos_amd.downsample.downsample_and_mask import get_real_size, read_tiff_plane

    want = np.load(os.path.join(golden_dir, "tiff_expected.npz"))
    for name, key in (("tiff_lzw16.tif", "lzw16"), ("tiff_lzw16_pred.tif", "lzw16"), ("tiff_lzw16_strips.tif", "lzw16"),
                      ("tiff_lzw8.tif", "lzw8"), ("tiff_be16.tif", "be16")):
        got = read_tiff_plane(os.path.join(golden_dir, name))
        assert got.dtype == np.uint16
        np.testing.assert_array_equal(got, want[key])
    with pytest.raises(FileNotFoundError):
        read_tiff_plane(os.path.join(golden_dir, "no_such_plane.tif"))
    with pytest.raises(NotImplementedError):
        read_tiff_plane(os.path.join(golden_dir, "ref_csv.npz"))       # not a TIFF


def test_merge_components_random_slabs_vs_whole_volume():
    """parallel.merge_components on random masks, random cuts (incl. empty slabs): per-slab labelling + seam pairs +
    union + renumbering == labelling of the whole volume (pure host logic, no process group)."""
    from delivr_cfos_amd.parallel import merge_components, merge_stats

    rng = np.random.default_rng(0)
    eng = _OracleCclEngine(orc)
    import torch
    for trial in range(12):
        Z, Y, X = int(rng.integers(6, 26)), int(rng.integers(4, 16)), int(rng.integers(4, 16))
        m = (rng.random((Z, Y, X)) < rng.choice([0.05, 0.2, 0.5])).astype(np.uint8)
        full, n_full = orc.ccl26(m)
        cuts = sorted(set([0, Z] + [int(v) for v in rng.integers(0, Z + 1, size=int(rng.integers(1, 4)))]))
        slabs = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        if trial % 3 == 0:
            slabs.insert(1, (slabs[0][1], slabs[0][1]))   # an empty slab in the middle
        labs, counts, raws = [], [], []
        for lo, hi in slabs:
            if hi > lo:
                lab, n = eng.ccl26(torch.from_numpy(m[lo:hi].copy()))
                labs.append(lab)
                counts.append(n)
                raws.append(eng.cc_stats_raw(lab, n))
            else:
                labs.append(None)
                counts.append(0)
                raws.append(None)
        live = [r for r, (lo, hi) in enumerate(slabs) if hi > lo]
        seams = []
        for k, r in enumerate(live[:-1]):
            pr = eng.seam_pairs(labs[r][-1], labs[live[k + 1]][0])
            if len(pr):
                seams.append((r, live[k + 1], pr))
        luts, n = merge_components(counts, seams)
        assert n == n_full
        out = np.zeros(m.shape, dtype=np.int64)
        for (lo, hi), lab, lut in zip(slabs, labs, luts):
            if hi > lo:
                out[lo:hi] = lut.astype(np.int64)[lab.numpy()]
        np.testing.assert_array_equal(out, full.astype(np.int64))
        st = merge_stats(luts, raws, [s[0] for s in slabs], m.shape, n)
        want = orc.cc_stats(full, n_full)
        np.testing.assert_array_equal(st["voxel_counts"], want["voxel_counts"])
        np.testing.assert_array_equal(st["bounding_boxes"], want["bounding_boxes"])
        np.testing.assert_array_equal(st["centroids"], want["centroids"])


def test_native_tiff_writer_roundtrip_and_libtiff_readback(tmp_path, golden_dir):
    """LZW and uncompressed planes written by csrc/tiffio.hip decode to the same pixels with the native reader and -
    when Pillow/libtiff is present (test-only dependency) - with libtiff: noise (table fills and resets), long runs
    (KwKwK), a constant plane, one row, odd sizes, 8 and 16 bit."""
    from delivr_cfos_amd.downsample.downsample_and_mask import read_tiff_plane
    from delivr_cfos_amd.tiffio import write_tiff_plane

    rng = np.random.default_rng(8)
    want = np.load(os.path.join(golden_dir, "tiff_expected.npz"))
    planes = {
        "noise16": rng.integers(0, 65536, (300, 257)).astype(np.uint16),
        "noise8": rng.integers(0, 256, (513, 300)).astype(np.uint8),
        "runs8": np.repeat(rng.integers(0, 4, (40, 11)).astype(np.uint8), 37, axis=1),
        "const16": np.full((64, 64), 1234, dtype=np.uint16),
        "row": np.arange(1000, dtype=np.uint16)[None, :],
        "golden": want["lzw16"],
    }
    try:
        from PIL import Image
    except ImportError:  # pragma: no cover
        Image = None
    for name, a in planes.items():
        for comp in ("lzw", None):
            p = str(tmp_path / f"{name}_{comp}.tif")
            write_tiff_plane(p, a, compression=comp)
            np.testing.assert_array_equal(read_tiff_plane(p), a)
            if Image is not None:
                b = np.array(Image.open(p))
                assert b.dtype == a.dtype
                np.testing.assert_array_equal(b, a)
    big = str(tmp_path / "big.tif")
    write_tiff_plane(big, planes["runs8"])
    assert os.path.getsize(big) < planes["runs8"].nbytes // 4     # it does compress


def test_native_tiff_reader_variants_libtiff_accepts(golden_dir):
    """Deflate (+ predictor), BigTIFF (uncompressed by libtiff, deflate with LONG8 strip tables), tiles (uncompressed,
    deflate + predictor, 8 bit; edge tiles padded): fixtures of oracle/make_goldens.py golden_tiff_variants, each
    decoded by libtiff to the expected pixels when it was written."""
    from delivr_cfos_amd.downsample.downsample_and_mask import read_tiff_plane

    want = np.load(os.path.join(golden_dir, "tiff_expected_variants.npz"))
    for name, key in (("tiff_deflate16.tif", "a16"), ("tiff_deflate16_pred.tif", "a16"), ("tiff_big16_deflate.tif", "a16"),
                      ("tiff_big8.tif", "b8"), ("tiff_tiled16.tif", "a16"), ("tiff_tiled16_deflate_pred.tif", "a16"),
                      ("tiff_tiled8.tif", "b8")):
        got = read_tiff_plane(os.path.join(golden_dir, name))
        assert got.dtype == np.uint16
        np.testing.assert_array_equal(got, want[key], err_msg=name)


def test_native_tiff_reader_survives_corrupt_files(tmp_path, golden_dir):
    """Random byte corruption and truncation of the fixtures: the reader either decodes or refuses with an exception -
    no crash, no exception across the C ABI; a header that claims a 900000 x 900000 plane is refused."""
    import struct
    from delivr_cfos_amd.downsample.downsample_and_mask import read_tiff_plane

    rng = np.random.default_rng(0)
    p = str(tmp_path / "f.tif")
    for name in ("tiff_lzw16.tif", "tiff_lzw16_strips.tif", "tiff_lzw8.tif", "tiff_be16.tif", "tiff_lzw16_pred.tif",
                 "tiff_deflate16_pred.tif", "tiff_big16_deflate.tif", "tiff_big8.tif", "tiff_tiled16.tif",
                 "tiff_tiled16_deflate_pred.tif"):
        data = bytearray(open(os.path.join(golden_dir, name), "rb").read())
        for it in range(60):
            d = bytearray(data)
            for _ in range(int(rng.integers(1, 8))):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
            if it % 5 == 0:
                d = d[: int(rng.integers(8, len(d)))]
            open(p, "wb").write(d)
            try:
                read_tiff_plane(p)
            except (NotImplementedError, FileNotFoundError, OSError, MemoryError):
                pass
    tags = [(256, 4, 1, 900000), (257, 4, 1, 900000), (258, 3, 1, 16), (259, 3, 1, 5), (262, 3, 1, 1), (273, 4, 1, 8),
            (277, 3, 1, 1), (278, 4, 1, 900000), (279, 4, 1, 4)]
    with open(p, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, 12) + b"\\x80\\x00\\x00\\x00")
        fh.write(struct.pack("<H", len(tags)))
        for tag, typ, cnt, val in tags:
            fh.write(struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val)))
        fh.write(struct.pack("<I", 0))
    with pytest.raises(NotImplementedError):
        read_tiff_plane(p)


def _gloo_round4_worker(rank, world, port, tmp):
    """p2p self-test of the seam transport and the all-ranks error exchange of the sharded count_blobs (gloo)."""
    import torch
    import torch.distributed as dist

    from delivr_cfos_amd.count_blobs import _raise_if_any_failed
    from delivr_cfos_amd.parallel import p2p_selftest

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p2p_selftest(dist, torch.device("cpu"), rank, world, 1 << 16)
    _raise_if_any_failed(dist, None, "nothing")  # nobody failed: returns on every rank
    try:
        _raise_if_any_failed(dist, "rank 1: disk full" if rank == 1 else None, "writing the label slabs")
        outcome = "no error"
    except RuntimeError as e:
        outcome = str(e)
    with open(os.path.join(tmp, f"outcome_{rank}.txt"), "w") as fh:
        fh.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_p2p_selftest_and_error_exchange_over_gloo(tmp_path, world):
    """Round 4: (i) the ring exchange that bench.py / run_inference run before a multi-rank job delivers every word (at one
    rank: to itself); (ii) a failure on ONE rank of the sharded count_blobs' file writes reaches EVERY rank as the same
    error instead of leaving the others in a barrier."""
    import torch.multiprocessing as mp

    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_gloo_round4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [open(tmp_path / f"outcome_{r}.txt").read() for r in range(world)]
    if world == 1:
        assert outs == ["no error"]
    else:
        assert all("writing the label slabs failed: rank 1: disk full" in o for o in outs), outs


def test_find_cached_returns_the_last_matching_entry_like_the_reference(tmp_path, monkeypatch):
    """count_blobs.py:10-34 of the reference: the LAST directory entry (os.listdir order) that holds the suffix and the brain
    name, False without a match - one helper now serves the three look-ups."""
    from delivr_cfos_amd import count_blobs as cb

    names = ["zz_other-3-cc3d.npy", "brainA-7-cc3d.npy", "brainA-stats.pickle", "brainA-9-cc3d.npy", "notes.txt"]
    monkeypatch.setattr(cb.os, "listdir", lambda p: list(names))
    settings = {"postprocessing": {"output_location": str(tmp_path)}}
    assert cb.load_cached_brain(settings, "brainA") == os.path.join(str(tmp_path), "brainA-9-cc3d.npy")
    assert cb.load_cached_stats(settings, "brainA") == os.path.join(str(tmp_path), "brainA-stats.pickle")
    assert cb.load_cached_brain(settings, "brainB") is False


def test_streaming_host_arithmetic():
    from delivr_cfos_amd import streaming as st

    assert st.even_slabs(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert st.inference_bytes_per_voxel(False, False, False) == 9 and st.inference_bytes_per_voxel(True, True, True) == 17
    assert st.forward_workspace_bytes((128, 128, 128), "fp16") > 30 * 2**30  # 3 lanes x 2^25 patch voxels x 340 B
    assert st.forward_workspace_bytes((32, 32, 32), "fp16") < 4 * 2**30
    assert st.hbm_budget_bytes(None, {"mi355x": {"hbm_budget_gb": 1.5}}) == int(1.5 * 2**30)


def test_range_guard_maps_the_named_layer_to_the_blocks_feeding_it():
    """range_guard.producers: MONAI BasicUNet's wiring (inference/inference.py:190-197) - the block whose InstanceNorm sums were
    not finite names its INPUT's producers; next_shifts turns the library's per-block |mean| + 8 sigma into a power of two."""
    from delivr_cfos_amd.range_guard import MAX_SHIFT, next_shifts, producers

    assert producers(1) == [0] and producers(2) == [1] and producers(9) == [8]
    assert producers(10) == [7, 9] and producers(12) == [5, 11] and producers(14) == [3, 13] and producers(16) == [1, 15]
    assert producers(11) == [10] and producers(17) == [16] and producers(18) == [17] and producers(0) == []
    peaks = [0.0] * 18
    shifts = [0] * 18
    # a hint on one of two candidates: only that block moves, far enough to bring the peak to <= 1024
    peaks[13] = 3.0e6
    plan = next_shifts(14, peaks, shifts)
    assert plan == {13: 12} and 3.0e6 / 2 ** 12 <= 1024 < 3.0e6 / 2 ** 11
    # no hint: every candidate moves by 6 bits; block 16 itself when its folded up half overflowed before its statistics exist
    assert next_shifts(14, [0.0] * 18, shifts) == {3: 6, 13: 6}
    assert next_shifts(16, [0.0] * 18, shifts) == {16: 6}
    assert next_shifts(18, [0.0] * 18, [0] * 17 + [MAX_SHIFT]) is None  # nothing left to try
    assert next_shifts(0, [0.0] * 18, shifts) is None


def test_native_cell_table_writer_writes_pythons_text(golden_dir):
    """dlv_cells_csv (host-only C: the reference's count_blobs.py:98-114 text without 0.8 s of Python formatting per brain) against
    hostlogic.cells_csv_text - which the golden ref_csv.npz pins to the text the reference's own count_blobs() wrote - on centroids
    that exercise every branch of Python's float repr: integral values, quarters, long fractions, < 1e-4, >= 1e16, zero."""
    from delivr_cfos_amd.hostlogic import cells_csv_bytes, cells_csv_text

    rng = np.random.default_rng(0)
    n = 5000
    cent = rng.random((n + 1, 3)) * 2048
    cent[5] = [12.0, 0.0, 2047.0]
    cent[6] = [5e-5, 1e-4, 9.999e-5]
    cent[7] = [1 / 3, 2 / 3, 1e16]
    cent[8] = [1e15, 123456789.125, 0.1]
    cent[9] = [1e22, 1.5e-7, 3.0e-300]
    cent[10:1000] = np.round(cent[10:1000] * 4) / 4
    cent[1000:2000] = rng.integers(0, 100000, (1000, 3)) / rng.integers(1, 5000, (1000, 3))
    stats = {"centroids": cent, "voxel_counts": rng.integers(1, 2**32 - 1, n + 1).astype(np.uint32)}
    for k in (n, 2, 1, 0):
        assert cells_csv_bytes(stats, k) == cells_csv_text(stats, k).encode(), k
    # ... and the golden itself: the text the reference's own count_blobs() wrote for the statistics of a gt patch
    g = np.load(os.path.join(golden_dir, "ref_csv.npz"))
    c = np.load(os.path.join(golden_dir, "orc_ccl.npz"))
    st = {"voxel_counts": c["gt0_counts"], "centroids": c["gt0_centroids"]}
    assert cells_csv_bytes(st, int(c["gt0_n"])) == str(g["csv_text"]).encode()


def test_range_recovery_spares_small_blocks_stops_after_a_futile_blind_step_and_takes_it_back():
    """range_guard.run_with_range_recovery on a fake engine: (i) a block whose recorded |mean| + 8 sigma would fall below 1 is not
    moved on no evidence; (ii) a blind step that leaves the same layer overflowing is not repeated - the format falls back to
    bf16_all with the futile shift taken back; (iii) a hinted step cures the overflow in the format asked for."""
    from delivr_cfos_amd._lib import DLV_ERANGE, DelivrHipError
    from delivr_cfos_amd.range_guard import next_shifts, run_with_range_recovery

    peaks = [0.0] * 18
    peaks[3], peaks[13] = 20.0, 300.0   # both small: 20 * 2^-6 < 1 (spared), 300 * 2^-6 = 4.7 (may move)
    assert next_shifts(14, peaks, [0] * 18) == {13: 6}
    peaks[13] = 30.0
    assert next_shifts(14, peaks, [0] * 18) is None

    class FakeEngine:
        def __init__(self, layer, peaks, cured_by):
            self.layer, self.peaks, self.cured_by = layer, peaks, cured_by
            self.shifts, self.log = [0] * 18, []

        def range_report(self):
            return self.layer, list(self.peaks)

        def conv_shifts(self):
            return list(self.shifts)

        def set_conv_shift(self, p, k):
            self.log.append((p, k))
            self.shifts[p] = k

    def runner(eng, ran):
        def run(prec):
            ran.append((prec, list(eng.shifts)))
            if prec != "bf16_all" and not eng.cured_by(eng.shifts):
                raise DelivrHipError(DLV_ERANGE, "fp16 range exceeded")
        return run

    # (ii) nothing recorded for the producers of layer 18 -> one blind step of 6 bits on block 17, still overflowing -> stop
    eng, ran, resets = FakeEngine(18, [0.0] * 18, lambda s: False), [], []
    out = run_with_range_recovery(eng, "fp16", runner(eng, ran), lambda: resets.append(1), log=lambda m: None)
    assert out == "bf16_all" and [p for p, _ in ran] == ["fp16", "fp16", "bf16_all"]
    assert eng.log == [(17, 6), (17, 0)] and ran[-1][1] == [0] * 18  # the futile shift is taken back before the last resort
    # (iii) a hint: block 15 reported 3e6 -> 12 bits, cured, the mixed format stays the mixed format
    pk = [0.0] * 18
    pk[15] = 3.0e6
    eng, ran = FakeEngine(16, pk, lambda s: s[15] >= 12), []
    out = run_with_range_recovery(eng, "bf16", runner(eng, ran), lambda: None, log=lambda m: None)
    assert out == "bf16" and eng.log == [(15, 12)] and [p for p, _ in ran] == ["bf16", "bf16"]
    # bf16_all never recovers: its range errors are raised
    eng = FakeEngine(5, [0.0] * 18, lambda s: False)
    import pytest

    with pytest.raises(DelivrHipError):
        def always(prec):
            raise DelivrHipError(DLV_ERANGE, "x")
        run_with_range_recovery(eng, "bf16_all", always, lambda: None, log=lambda m: None)


def test_c_abi_range_policy_equals_the_python_one():
    """dlv_range_next_shifts (api.hip; what dlv_range_recover / dlv_comm_range_recover apply for hosts without run_inference) against
    range_guard.next_shifts on every layer x a set of peak / shift patterns (pure host logic: no GPU)."""
    import ctypes as C

    from delivr_cfos_amd import _lib
    from delivr_cfos_amd.range_guard import next_shifts

    lib = _lib.load()
    rng = np.random.default_rng(5)
    n = _lib.N_CONV
    cases = 0
    for layer in range(-1, 20):
        for trial in range(40):
            peaks = np.where(rng.random(n) < 0.3, 2.0 ** rng.uniform(12.0, 40.0, n), 0.0).astype(np.float32)
            if trial % 2:  # the library records EVERY block's peak: small ones (never moved below 1) beside the hints
                peaks = np.where(peaks > 0, peaks, 2.0 ** rng.uniform(-3.0, 11.9, n)).astype(np.float32)
            if trial % 4 == 0:
                peaks[:] = 0
            if trial % 7 == 0:
                peaks = np.where(peaks > 0, np.float32(4097.0), peaks)  # just above the reporting threshold: one bit... (ceil(log2(4.0009)) = 3)
            shifts = rng.integers(0, 41, n).astype(np.int32) if trial % 3 else np.zeros(n, np.int32)
            out = (C.c_int * n)()
            changed = lib.dlv_range_next_shifts(layer, peaks.ctypes.data_as(C.POINTER(C.c_float)), shifts.ctypes.data_as(C.POINTER(C.c_int)), out)
            want = next_shifts(layer, [float(v) for v in peaks], [int(v) for v in shifts]) if 0 <= layer <= 18 else None
            exp = [int(v) for v in shifts]
            for p, k in (want or {}).items():
                exp[p] = k
            assert list(out) == exp, (layer, peaks, shifts, list(out), want)
            assert changed == len(want or {})
            cases += 1
    assert cases > 500
    assert lib.dlv_range_next_shifts(3, None, None, None) == -1
