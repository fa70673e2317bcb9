"""hostio.py: the file <-> HBM pipelines of the step boundaries (masked_nifti.npy in, binaries.npy / labels out;
reference inference/inference.py:234,312-318, count_blobs.py:45-65,86-88) move exactly the bytes np.load / np.save would."""
import io
import os

import numpy as np
import pytest


def test_file_backing_of_memmap_views(tmp_path):
    """CPU: a slice of a memmapped .npy resolves to (file, byte offset of its first element)"""
    from delivr_cfos_amd import hostio

    a = np.arange(4 * 6 * 8, dtype=np.uint16).reshape(1, 1, 4, 6, 8)
    path = str(tmp_path / "v.npy")
    np.save(path, a)
    mm = np.memmap(path, dtype=np.uint16, mode="r", shape=a.shape, offset=128)
    f, off = hostio._file_backing(mm[0, 0])
    assert f == path and off == 128
    f, off = hostio._file_backing(mm[0, 0][1:3])
    assert f == path and off == 128 + 1 * 6 * 8 * 2
    assert hostio._file_backing(mm[0, 0][:, 1:3]) is None  # (not contiguous: copied by the reader threads instead)
    assert hostio._file_backing(np.zeros(4)) is None
    mm2 = np.load(path, mmap_mode="r")
    f, off = hostio._file_backing(mm2[0, 0, 2:])
    assert f == path and off == mm2.offset + 2 * 6 * 8 * 2
    with open(path, "rb") as fh:
        fh.seek(off)
        assert np.array_equal(np.frombuffer(fh.read(2 * 6 * 8 * 2), dtype=np.uint16), a[0, 0, 2:].reshape(-1))


def test_split_and_create_npy(tmp_path):
    from delivr_cfos_amd import hostio

    for n, parts in ((1, 16), (64 << 20, 16), ((64 << 20) + 5, 7), (3 << 20, 2)):
        pieces = hostio._split(n, parts)
        assert pieces[0][0] == 0 and pieces[-1][1] == n and len(pieces) <= parts
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    path = str(tmp_path / "x.npy")
    off = hostio.create_npy(path, np.uint8, (3, 5, 7))
    buf = io.BytesIO()
    np.save(buf, np.zeros((3, 5, 7), np.uint8))
    with open(path, "rb") as fh:
        assert fh.read(off) == buf.getvalue()[:off]  # numpy's own header, byte for byte
    assert os.path.getsize(path) == off + 3 * 5 * 7 and off == 128


@pytest.mark.gpu
@pytest.mark.parametrize("threads,write_mode", [("1", "mmap"), ("5", "mmap"), ("5", "pwrite")])
def test_upload_and_download_round_trip(tmp_path, monkeypatch, threads, write_mode):
    import torch

    from delivr_cfos_amd import hostio
    from delivr_cfos_amd.engine import HipEngine

    monkeypatch.setenv("DLV_IO_THREADS", threads)
    monkeypatch.setattr(hostio, "WRITE_MODE", write_mode)
    eng = HipEngine(0)
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 65535, size=(1, 1, 37, 130, 257), dtype=np.uint16)  # (chunks of 3 MiB below: ragged last chunk)
    path = str(tmp_path / "masked_nifti.npy")
    np.save(path, vol)
    mm = np.memmap(path, dtype=np.uint16, mode="r", shape=vol.shape, offset=128)
    for src in (mm[0, 0], mm[0, 0][5:29], vol[0, 0], vol[0, 0][:, ::2]):
        t = hostio.upload(eng, src, chunk_bytes=3 << 20)
        assert t.dtype == torch.uint16 and np.array_equal(t.cpu().numpy(), np.asarray(src))
    t = hostio.upload(eng, path, np.uint16, vol.shape, offset=128, chunk_bytes=1 << 20)
    assert np.array_equal(t.cpu().numpy(), vol)
    assert hostio.last_transfer["h2d"]["bytes"] == vol.nbytes
    slab = eng.upload_volume(mm, 3, 30, chunk_bytes=1 << 20)  # the slab upload of a sharded rank
    assert tuple(slab.shape) == (1, 1, 27, 130, 257) and np.array_equal(slab.cpu().numpy()[0, 0], vol[0, 0, 3:30])
    # download: into an array, into a file region, as a whole .npy (header = np.save's)
    dev = t[0, 0].contiguous()
    host = np.empty(dev.shape, dtype=np.uint16)
    hostio.download(eng, dev, host, chunk_bytes=3 << 20)
    assert np.array_equal(host, vol[0, 0])
    out = str(tmp_path / "out.npy")
    hostio.save_npy(eng, dev, out, np.uint16, partial=True)
    assert not os.path.exists(out + ".partial")
    ref = str(tmp_path / "ref.npy")
    np.save(ref, vol[0, 0])
    assert open(out, "rb").read() == open(ref, "rb").read()
    lab = torch.arange(dev.numel(), dtype=torch.int32, device=eng.device).reshape(dev.shape)  # uint32 labels live in int32 tensors
    hostio.save_npy(eng, lab, out, np.uint32)
    back = np.load(out)
    assert back.dtype == np.uint32 and np.array_equal(back.reshape(-1), np.arange(dev.numel(), dtype=np.uint32))
    # a slab written into the middle of an existing file (the sharded count_blobs)
    off = hostio.create_npy(out, np.uint16, vol.shape[2:])
    hostio.download(eng, dev[10:20].contiguous(), out, offset=off + 10 * 130 * 257 * 2, chunk_bytes=1 << 20)
    got = np.load(out)
    assert np.array_equal(got[10:20], vol[0, 0, 10:20]) and not got[:10].any() and not got[20:].any()
    with pytest.raises(ValueError):
        hostio.download(eng, dev[:, ::2], host)
    # sparse files: blocks of 1 MiB that are zero in HBM are not written - the fresh file reads back identically and holds holes
    big = torch.zeros((40, 512, 1024), dtype=torch.int32, device=eng.device)  # 80 MiB of labels, most of it background
    big[3, 100:140, 17:900] = torch.arange(40 * 883, dtype=torch.int32, device=eng.device).reshape(40, 883) + 1
    big[25:27] = 7
    big[-1, -1, -5:] = 9  # the last bytes of the tensor
    sp = str(tmp_path / "sparse.npy")
    hostio.save_npy(eng, big, sp, np.uint32, what="d2h_sparse")
    assert np.array_equal(np.load(sp), big.cpu().numpy().view(np.uint32))
    tr = hostio.last_transfer["d2h_sparse"]
    assert tr["bytes"] == big.numel() * 4 and tr["zero_blocks_skipped"] >= 0.9 * (80 << 20) // hostio.SPARSE_BLOCK and tr["bytes_written"] <= 12 << 20, tr
    if write_mode == "pwrite":
        assert os.stat(sp).st_blocks * 512 < big.numel() * 4 // 2, "the zero blocks should have stayed holes"
    dense = str(tmp_path / "dense.npy")
    hostio.save_npy(eng, big, dense, np.uint32, sparse=False)
    assert open(dense, "rb").read() == open(sp, "rb").read()
    if write_mode == "mmap":
        with pytest.raises(ValueError):  # the file must already have its size (create_npy)
            hostio.download(eng, dev, out, offset=off + 1)
    eng.close()


@pytest.mark.gpu
def test_count_blobs_writes_uint16_labels_for_few_components_and_reuses_them(tmp_path):
    """fewer than 2^16 components: the label file is uint16 (cc3d's smallest fitting type), converted in HBM; a second call
    finds the cached labelling and widens it in HBM for the statistics - same pickle, same CSV"""
    import pickle

    from delivr_cfos_amd.count_blobs import count_blobs
    from oracle import delivr_oracle as orc

    rng = np.random.default_rng(11)
    mask = (rng.random((40, 64, 72)) < 0.02).astype(np.uint8)
    d = tmp_path / "in" / "brain" / "binary_segmentations"
    os.makedirs(d)
    np.save(str(d / "binaries.npy"), mask)
    settings = {"postprocessing": {"output_location": str(tmp_path / "post") + "/"}}
    n = count_blobs(settings, str(tmp_path / "in"), 0, "brain", (1, 1) + mask.shape)
    lab_ref, n_ref = orc.ccl26(mask)
    assert n == n_ref and n < 2**16
    post = str(tmp_path / "post")
    labels = np.load(os.path.join(post, f"brain-{n}-cc3d.npy"))
    assert labels.dtype == np.uint16 and np.array_equal(labels.astype(np.uint32), lab_ref)
    assert not any(f.endswith(".partial") for f in os.listdir(post))
    first = pickle.load(open(os.path.join(post, "brain-stats.pickle"), "rb"))
    csv = open(os.path.join(post, f"{mask.shape}_brain.csv")).read()
    os.remove(os.path.join(post, "brain-stats.pickle"))
    assert count_blobs(settings, str(tmp_path / "in"), 0, "brain", (1, 1) + mask.shape) == n  # cached labels, fresh statistics
    again = pickle.load(open(os.path.join(post, "brain-stats.pickle"), "rb"))
    for k in first:
        np.testing.assert_array_equal(first[k], again[k])
    assert open(os.path.join(post, f"{mask.shape}_brain.csv")).read() == csv == orc.cells_csv_text(orc.cc_stats(lab_ref, n_ref), n_ref)


@pytest.mark.gpu
def test_reserve_allocates_what_the_pass_asks_for_and_the_shared_engine_is_one_per_device():
    """dlv_reserve_dev (run_inference calls it from a second thread while it reads the volume): after it, a pass and a finalize of
    that geometry allocate (almost) nothing in the library - and the result is the one of an engine that allocated on demand."""
    import torch

    from delivr_cfos_amd.engine import HipEngine, shared_engine
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    assert shared_engine(0) is shared_engine(0) and shared_engine(0).shared
    shape, roi = (64, 128, 192), (64, 64, 64)
    vol = synth_volume_np(shape, seed=3, dense=True)
    sd = random_state_dict(2)
    outs = []
    for reserve in (True, False):
        eng = HipEngine(0)
        eng.load_state_dict({"state_dict": sd})
        v = eng.to_device(vol)
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        mask = torch.empty(shape, dtype=torch.uint8, device="cuda")
        p = eng.make_sw_params(shape, roi, 0.5, None, 0, "fp16")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        if reserve:
            eng.reserve(p, shape)
            free1 = torch.cuda.mem_get_info()[0]
            assert free0 - free1 > 3 * 2**30  # three lanes of (up to) 64 windows x 64^3 voxels x ~400 B of activations
        eng.sw_infer(p, v, acc)
        out = eng.finalize(acc, None, v, shape, 0.5, 30, 0, out=mask)
        eng.sync()
        assert out.data_ptr() == mask.data_ptr()
        if reserve:
            # (what is left: window lists, the HSA queues of the lanes' streams on their first use - against the GBs reserved)
            assert free1 - torch.cuda.mem_get_info()[0] < 512 * 2**20, "the pass allocated what dlv_reserve_dev should have"
        outs.append((acc.cpu().numpy(), mask.cpu().numpy()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    with pytest.raises(ValueError):
        eng2 = HipEngine(0)
        eng2.finalize(torch.zeros(shape, device="cuda"), None, torch.zeros(shape, dtype=torch.uint16, device="cuda"), shape, out=torch.empty((1, 2, 3), dtype=torch.uint8, device="cuda"))


@pytest.mark.gpu
def test_pipelined_brains_write_the_files_of_the_sequential_run(tmp_path):
    """python -m delivr_cfos_amd runs a batch of brains pipelined: run_inference(prefetch=<next masked_nifti.npy>, defer_write=True) reads
    the next volume into HBM while this brain's passes run and lets binaries.npy stream out behind the next brain; count_blobs(
    defer_write=True) lets the label file stream out behind the next labelling.  Same files, byte for byte, as one brain at a time."""
    from delivr_cfos_amd import hostio
    from delivr_cfos_amd.count_blobs import count_blobs
    from delivr_cfos_amd.hostlogic import padded_shape
    from delivr_cfos_amd.inference.inference import run_inference
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    crop = (32, 32, 32)
    sd = {"state_dict": random_state_dict(4)}
    shapes = [(40, 70, 64), (40, 70, 64), (33, 64, 96)]  # (the third brain has another shape: the read-ahead is sized by its file)
    niftis = []
    for k, shp in enumerate(shapes):
        vol = synth_volume_np(shp, seed=20 + k)
        pad = padded_shape(shp, crop)
        d = tmp_path / "mask" / f"b{k}" / "masked_niftis"
        os.makedirs(d)
        out = np.lib.format.open_memmap(str(d / "masked_nifti.npy"), mode="w+", dtype=np.uint16, shape=(1, 1) + pad)
        out[0, 0, : shp[0], : shp[1], : shp[2]] = vol
        out.flush()
        del out
        niftis.append(str(d / "masked_nifti.npy"))

    def run(tag, pipelined):
        blob, post = str(tmp_path / tag / "blob"), str(tmp_path / tag / "post") + "/"
        settings = {"postprocessing": {"output_location": post}}
        for k, shp in enumerate(shapes):
            kw = dict(prefetch=niftis[k + 1] if k + 1 < len(shapes) else None, defer_write=True) if pipelined else {}
            run_inference([niftis[k]], blob, (1, 1) + shp, comment=f"b{k}", crop_size=crop, state_dict=sd, precision="fp32", **kw)
        hostio.wait_deferred()
        ns = [count_blobs(settings, blob, k, f"b{k}", (1, 1) + shp, **({"defer_write": True} if pipelined else {})) for k, shp in enumerate(shapes)]
        hostio.wait_deferred()
        files = {}
        for root, _dirs, names in os.walk(str(tmp_path / tag)):
            for n in names:
                files[os.path.relpath(os.path.join(root, n), str(tmp_path / tag))] = open(os.path.join(root, n), "rb").read()
        return ns, files

    n_seq, f_seq = run("seq", False)
    n_pipe, f_pipe = run("pipe", True)
    assert n_seq == n_pipe and sorted(f_seq) == sorted(f_pipe) and not any(k.endswith(".partial") for k in f_pipe)
    for k in f_seq:
        assert f_seq[k] == f_pipe[k], k
    assert any("cc3d.npy" in k for k in f_seq) and any(k.endswith("binaries.npy") for k in f_seq)
