"""File <-> HBM transfers of the step boundaries: the reference's steps hand volumes over as .npy files
(masked_nifti.npy in, inference/inference.py:234; binaries.npy out, :312-318; the label volume and its re-read,
count_blobs.py:45-65,86-88), so a 5 s GPU pass sits between an 8.6 GB read and a 4.3 GB write, and 20 ms of labelling
between a 4.3 GB read and a 17 GB write.  Everything here is a chunked pipeline through a small ring of PINNED staging
buffers: reader threads pread file bytes straight into the pinned memory (27 GB/s from tmpfs with 16 threads), writer
threads pwrite out of it (both release the GIL; buffered writes of ONE file serialise on its inode lock in the kernel: 4-7
GB/s on tmpfs whatever the thread count - see WRITE_MODE), the copy engine moves the previous chunk in the meantime, and
no whole-volume pageable copy is ever made.

    upload(engine, src, ...)        file / memmap / ndarray  -> tensor in HBM
    download(engine, tensor, dst)   tensor in HBM            -> bytes of a file (at an offset) / ndarray
    create_npy(path, dtype, shape)  an empty .npy with numpy's own header (what np.save / open_memmap write) -> offset

`last_transfer` holds the figures of the most recent call of each kind (bytes, seconds, GB/s) - bench.py reports them.
"""
from __future__ import annotations

import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor, wait
from typing import Optional

import numpy as np

CHUNK_BYTES = 64 << 20
N_STAGE = 4
# "pwrite": positional writes from a few threads; "mmap": memcpy into a MAP_SHARED mapping of the file.  Measured on the GPU
# box's /dev/shm (2 x 64-core EPYC, profiles/r06b_step_probe.json, 8 GB file): pwrite 7.1 / 3.9 / 6.1 / 3.5 / 2.1 GB/s with 8 / 16 /
# 32 / 64 / 128 threads - buffered writes of ONE file serialise on its inode lock, more threads only add hand-overs - and the
# mapping 3.5 / 3.8 / 2.1 / 1.1 / 0.7 GB/s: page faults that allocate tmpfs pages scale negatively.  A single file takes 4-7 GB/s
# from the kernel, whatever the host does; the writers exist to keep the copy engine and the coordinator from waiting.
# (The boxes these figures come from grant the process 16 cores of CPU time - cgroup cpu.max - whatever they show: part of why more
# threads were slower there.  $DLV_IO_WRITE_THREADS overrides the 8 on a host that has its cores to itself.)
WRITE_MODE = "pwrite"
WRITE_THREADS = max(1, int(os.environ.get("DLV_IO_WRITE_THREADS") or 8))
# Files written into a FRESH file (create_npy: ftruncate, every byte reads as zero) skip the blocks that are zero in HBM: the file
# stays sparse there and reads back the same bytes.  A brain fills ~40 % of its box - binaries.npy and the label volume are zero
# outside it - so ~60 % of the 4.3 + 17 GB never cross PCIe or the kernel's one-file write path (the bottleneck of step 3).
# (block size: 1 MiB blocks skipped 23 % of the mask and 38 % of the labels of the benchmark brain - a block of half a plane is
# rarely empty; 128 KiB = 64 mask rows / 16 label rows of 2048 voxels follow the brain's outline)
SPARSE_BLOCK = 128 << 10
_MAX_IO = 1 << 30  # (one pread / pwrite moves at most 0x7ffff000 bytes on Linux)

last_transfer = {}
_pool_lock = threading.Lock()
_pool = None


def io_threads() -> int:
    """threads that read file bytes: DLV_IO_THREADS, else 16 (fewer on small hosts)"""
    env = os.environ.get("DLV_IO_THREADS")
    if env:
        return max(1, int(env))
    return max(2, min(16, (os.cpu_count() or 2)))


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


def io_cpus(device_index: Optional[int] = None):
    """CPUs the I/O threads are bound to: those of the NUMA node the GPU hangs on (a two-socket host: the pinned staging buffers
    and the copy engine's DMA target live there).  Measured (profiles/r06k_io_numa_probe.json, 8 GB from / to tmpfs, four rounds
    interleaved): reads 22.5-24.7 GB/s unbound, 27.5-34.8 bound; writes 3.5-6.6 unbound, 4.3-6.7 bound - the one-file write rate is
    the kernel's and stays noisy.  DLV_IO_NUMA=off disables the binding, =<n> picks the node; None when the node is unknown."""
    mode = os.environ.get("DLV_IO_NUMA", "auto")
    if mode == "off":
        return None
    try:
        if mode == "auto":
            import torch

            pr = torch.cuda.get_device_properties(torch.cuda.current_device() if device_index is None else device_index)  # (one process = one device)
            with open(f"/sys/bus/pci/devices/{int(pr.pci_domain_id):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0/numa_node") as f:
                node = int(f.read().strip())
        else:
            node = int(mode)
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = _parse_cpulist(f.read()) & set(os.sched_getaffinity(0))
        return cpus or None
    except Exception:
        return None


def _bind_io_thread(cpus):
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)  # (0: the calling thread)
        except OSError:
            pass


def _executor() -> ThreadPoolExecutor:
    global _pool
    with _pool_lock:
        if _pool is None or _pool._max_workers != io_threads():
            if _pool is not None:
                _pool.shutdown(wait=False)
            _pool = ThreadPoolExecutor(max_workers=io_threads(), thread_name_prefix="dlv-io", initializer=_bind_io_thread, initargs=(io_cpus(),))
        return _pool


class _Stage:
    """a ring of pinned byte buffers kept by the engine (hipHostMalloc of 256 MB is not free: allocated once)"""

    def __init__(self, torch, n: int, nbytes: int):
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(n)]
        self.views = [memoryview(b.numpy()) for b in self.bufs]
        self.nbytes = nbytes


class _Slots:
    """the staging rings of one direction of one engine: acquire() -> index of a free ring (blocks while all are in use)"""

    def __init__(self, n: int):
        self.free = list(range(n))
        self.cv = threading.Condition()

    def acquire(self) -> int:
        with self.cv:
            while not self.free:
                self.cv.wait()
            return self.free.pop(0)

    def release(self, i: int) -> None:
        with self.cv:
            self.free.append(i)
            self.free.sort()
            self.cv.notify()


# rings per direction.  "up": one (the next brain's volume is read while the previous brain's files stream out of the "down" rings).
# "down": two - buffered writes of ONE file serialise on its inode lock, two files do not: two 8 GB files written at once took
# 6.9-8.7 GB/s together against 3.8-5.7 one after the other (profiles/r06w_two_files_probe.json), so two deferred outputs (the label
# volumes of consecutive brains) may be in flight together.
_RINGS = {"up": 1, "down": 2}


def _slots(engine, kind: str) -> _Slots:
    table = engine.__dict__.setdefault("_io_slots", {})
    with _pool_lock:
        if kind not in table:
            table[kind] = _Slots(_RINGS[kind])
    return table[kind]


# ---- deferred transfers: a step may return while its output file is still streaming out of HBM (the next brain's passes run
# meanwhile); one background worker per engine keeps the files in order, wait_deferred() joins and re-raises -------------------
_deferred = {}


def submit_deferred(engine, fn, *args, **kwargs):
    """run fn(*args, **kwargs) on one of the engine's background workers (as many as there are "down" rings; the outputs are
    independent files, so their order does not matter) -> Future"""
    with _pool_lock:
        ent = _deferred.get(id(engine))
        if ent is None:
            ent = _deferred[id(engine)] = {"pool": ThreadPoolExecutor(max_workers=_RINGS["down"], thread_name_prefix="dlv-deferred"), "futs": []}
    fut = ent["pool"].submit(fn, *args, **kwargs)
    ent["futs"].append(fut)
    return fut


def wait_deferred(engine=None) -> None:
    """join every deferred transfer (of one engine, or of all); the first exception any of them ran into is raised here"""
    with _pool_lock:
        ents = [e for k, e in _deferred.items() if engine is None or k == id(engine)]
    first = None
    for ent in ents:
        futs, ent["futs"] = ent["futs"], []
        for f in futs:
            try:
                f.result()
            except BaseException as exc:  # noqa: BLE001 (collected, raised below)
                first = first or exc
    if first is not None:
        raise first


def _settle(futures) -> None:
    """wait for reader / writer tasks that are still running when a transfer ends in an error: the file descriptor and the staging
    buffers they use are about to go"""
    for f in futures:
        try:
            f.result()
        except Exception:
            pass


def _stage_of(engine, chunk_bytes: int, kind: str) -> _Stage:
    stages = engine.__dict__.setdefault("_io_stages", {})
    st = stages.get(kind)
    if st is None or st.nbytes < chunk_bytes:
        st = stages[kind] = _Stage(engine.torch, N_STAGE, chunk_bytes)
    return st


def _pread_full(fd: int, mv: memoryview, off: int) -> None:
    done, n = 0, len(mv)
    while done < n:
        got = os.preadv(fd, [mv[done:min(n, done + _MAX_IO)]], off + done)
        if got <= 0:
            raise EOFError(f"short read: {n - done} bytes missing at offset {off + done}")
        done += got


def _pwrite_full(fd: int, mv: memoryview, off: int) -> None:
    done, n = 0, len(mv)
    while done < n:
        done += os.pwrite(fd, mv[done:min(n, done + _MAX_IO)], off + done)


def _split(n: int, parts: int, align: int = 128 << 10):
    """[0, n) in up to `parts` pieces whose cuts sit at multiples of `align`"""
    per = -(-n // parts)
    per = -(-per // align) * align
    return [(s, min(s + per, n)) for s in range(0, n, per)]


def _file_backing(src):
    """(path, byte offset of element 0) when `src` is a C-contiguous np.memmap of a file region, else None"""
    if isinstance(src, np.memmap) and src.flags.c_contiguous and getattr(src, "filename", None) is not None:
        base = src
        while isinstance(getattr(base, "base", None), np.memmap):  # views keep the mmap's own offset: walk to the owner
            base = base.base
        start = src.__array_interface__["data"][0] - base.__array_interface__["data"][0]
        return str(src.filename), int(base.offset) + int(start)
    return None


def upload(engine, src, dtype=None, shape=None, offset: int = 0, out=None, chunk_bytes: int = CHUNK_BYTES, what: str = "h2d"):
    """`src` -> tensor in HBM (a new one, or the contiguous tensor `out`).
    src: a path (with `dtype`, `shape`, byte `offset` of the payload), a C-contiguous np.memmap (read through its file),
    or any ndarray (copied by the reader threads).  dtype: numpy dtype of the elements."""
    torch = engine.torch
    fd, arr = None, None
    if isinstance(src, (str, os.PathLike)):
        path, off0 = os.fspath(src), int(offset)
        dt = np.dtype(dtype)
        shape = tuple(int(v) for v in shape)
    else:
        a = src if isinstance(src, np.ndarray) else np.asarray(src)
        dt, shape = a.dtype, tuple(int(v) for v in a.shape)
        fb = _file_backing(a)
        if fb is not None:
            path, off0 = fb
        else:
            path, arr = None, np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    tdt = {"uint8": torch.uint8, "uint16": torch.uint16, "int16": torch.int16, "int32": torch.int32, "uint32": torch.int32,
           "float32": torch.float32, "float16": torch.float16}[dt.name]
    if out is None:
        out = torch.empty(shape, dtype=tdt, device=engine.device)
    elif not out.is_contiguous() or out.numel() * out.element_size() != nbytes:
        raise ValueError("upload: `out` must be a contiguous tensor of the source's size")
    if nbytes == 0:
        return out
    dst = out.reshape(-1).view(torch.uint8)
    pool, nthr = _executor(), io_threads()
    t0 = time.perf_counter()
    if path is not None:
        fd = os.open(path, os.O_RDONLY)
    pending = []  # (buffer index, byte range, futures) whose reads are in flight
    slots = _slots(engine, "up")
    slot = slots.acquire()
    try:
        st = _stage_of(engine, chunk_bytes, f"up{slot}")
        copy_stream = torch.cuda.Stream(device=engine.device)
        evs = [torch.cuda.Event() for _ in range(N_STAGE)]

        def fill(b, lo, hi):
            mv = st.views[b]
            if fd is not None:
                return [pool.submit(_pread_full, fd, mv[s:e], off0 + lo + s) for s, e in _split(hi - lo, nthr)]
            return [pool.submit(_copy_bytes, mv[s:e], arr[lo + s:lo + e]) for s, e in _split(hi - lo, nthr)]

        def finish(b, lo, hi, futs):
            for f in futs:
                f.result()
            with torch.cuda.stream(copy_stream):
                dst[lo:hi].copy_(st.bufs[b][: hi - lo], non_blocking=True)
                evs[b].record(copy_stream)

        with torch.cuda.device(engine.device):
            for i, lo in enumerate(range(0, nbytes, chunk_bytes)):
                hi, b = min(lo + chunk_bytes, nbytes), i % N_STAGE
                if i >= N_STAGE:
                    evs[b].synchronize()  # the H2D copy that read this buffer has finished
                pending.append((b, lo, hi, fill(b, lo, hi)))
                if len(pending) >= 2:  # reads of chunk i run while chunk i-1 is handed to the copy engine
                    finish(*pending.pop(0))
            while pending:
                finish(*pending.pop(0))
            copy_stream.synchronize()
    finally:
        for item in pending:
            _settle(item[3])
        slots.release(slot)
        if fd is not None:
            os.close(fd)
    dt_s = time.perf_counter() - t0
    last_transfer[what] = {"bytes": nbytes, "s": dt_s, "GBps": nbytes / dt_s / 1e9, "threads": nthr,
                           "source": "file (pread into pinned staging)" if fd is not None else "host array"}
    return out


def _copy_bytes(dst_mv: memoryview, src_arr) -> None:
    np.copyto(np.frombuffer(dst_mv, dtype=np.uint8), src_arr, casting="no")


def _zero_block_flags(torch, src, nblk: int):
    """numpy bool (nblk,): block i of SPARSE_BLOCK bytes of the uint8 device tensor `src` holds a non-zero byte"""
    words = src[: nblk * SPARSE_BLOCK].view(torch.int64).view(nblk, SPARSE_BLOCK // 8)
    out = torch.empty(nblk, dtype=torch.bool, device=src.device)
    step = max(1, (1 << 30) // SPARSE_BLOCK)  # (1 GiB at a time: the comparison's temporary stays at 128 MiB)
    for b0 in range(0, nblk, step):
        out[b0:b0 + step] = words[b0:b0 + step].ne(0).any(dim=1)
    return out.cpu().numpy()


def download(engine, tensor, dst, offset: int = 0, chunk_bytes: int = CHUNK_BYTES, what: str = "d2h", sync_file: bool = False,
             synced: bool = False, sparse: bool = False):
    """contiguous tensor in HBM -> `dst`: a path (bytes written at `offset`; the file must exist, e.g. from create_npy), an
    open file descriptor (int), or a C-contiguous writeable ndarray of the same byte size.  synced=True: the caller has already
    synchronised the engine (a download running in a side thread must not touch the context).  sparse=True (file destinations
    whose target range is known to read as zeros - a file fresh from create_npy): blocks of SPARSE_BLOCK bytes that are zero in
    HBM are neither copied nor written."""
    torch = engine.torch
    if not tensor.is_contiguous():
        raise ValueError("download: the tensor must be contiguous")
    src = tensor.reshape(-1).view(torch.uint8)
    nbytes = int(src.numel())
    fd, own_fd, arr, fmap = None, False, None, None
    if isinstance(dst, (str, os.PathLike)):
        fd, own_fd = os.open(os.fspath(dst), os.O_RDWR), True
    elif isinstance(dst, int):
        fd = dst
    else:
        if not (isinstance(dst, np.ndarray) and dst.flags.c_contiguous and dst.flags.writeable and dst.nbytes == nbytes):
            raise ValueError("download: destination array must be C-contiguous, writeable and of the tensor's size")
        arr = dst.reshape(-1).view(np.uint8)
    if nbytes == 0:
        if own_fd:
            os.close(fd)
        return
    pool, nthr = _executor(), (io_threads() if isinstance(dst, np.ndarray) else min(io_threads(), WRITE_THREADS))
    writes = None
    t0 = time.perf_counter()
    slots = _slots(engine, "down")
    slot = slots.acquire()
    try:
        st = _stage_of(engine, chunk_bytes, f"down{slot}")
        if fd is not None and WRITE_MODE == "mmap":
            import mmap

            if os.fstat(fd).st_size < offset + nbytes:
                raise ValueError("download: the file is smaller than offset + tensor size (create it with create_npy)")
            try:
                fmap = mmap.mmap(fd, offset + nbytes, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
                arr = np.frombuffer(fmap, dtype=np.uint8)[offset:offset + nbytes]
            except (OSError, ValueError):  # (a file system without shared writable mappings: positional writes)
                fmap, arr = None, None
        if not synced:
            engine.sync()
        copy_stream = torch.cuda.Stream(device=engine.device)
        copy_stream.wait_stream(torch.cuda.current_stream(engine.device))
        evs = [torch.cuda.Event() for _ in range(N_STAGE)]
        writes = [[] for _ in range(N_STAGE)]
        inflight = []  # (buffer, lo, hi) whose D2H copy has been queued
        flags, nblk, written = None, 0, 0
        if sparse and fd is not None and chunk_bytes % SPARSE_BLOCK == 0 and nbytes >= 4 * SPARSE_BLOCK:
            with torch.cuda.device(engine.device):
                nblk = nbytes // SPARSE_BLOCK
                flags = _zero_block_flags(torch, src, nblk)  # (the tail beyond the last whole block is always written)

        def runs_of(lo, hi):
            """byte ranges [s, e) relative to `lo` of the chunk [lo, hi) that have to be written"""
            if flags is None:
                return [(0, hi - lo)]
            b0, b1 = lo // SPARSE_BLOCK, min(hi // SPARSE_BLOCK, nblk)
            out, start = [], None
            for k in range(b0, b1):
                if flags[k] and start is None:
                    start = k
                elif not flags[k] and start is not None:
                    out.append((start * SPARSE_BLOCK - lo, k * SPARSE_BLOCK - lo))
                    start = None
            if start is not None:
                out.append((start * SPARSE_BLOCK - lo, b1 * SPARSE_BLOCK - lo))
            if hi > b1 * SPARSE_BLOCK:  # the ragged tail of the tensor
                if out and out[-1][1] == b1 * SPARSE_BLOCK - lo:
                    out[-1] = (out[-1][0], hi - lo)
                else:
                    out.append((b1 * SPARSE_BLOCK - lo, hi - lo))
            return out

        def drain(b, lo, hi):
            nonlocal written
            evs[b].synchronize()
            mv = st.views[b]
            writes[b] = []
            for rs, re_ in runs_of(lo, hi):
                written += re_ - rs
                if arr is None:
                    writes[b] += [pool.submit(_pwrite_full, fd, mv[rs + s:rs + e], offset + lo + rs + s) for s, e in _split(re_ - rs, nthr)]
                else:
                    writes[b] += [pool.submit(_copy_out, arr[lo + rs + s:lo + rs + e], mv[rs + s:rs + e]) for s, e in _split(re_ - rs, nthr)]

        with torch.cuda.device(engine.device):
            k = 0  # chunks that are copied (a staging buffer is reused every N_STAGE of THEM)
            for lo in range(0, nbytes, chunk_bytes):
                hi = min(lo + chunk_bytes, nbytes)
                if flags is not None and not runs_of(lo, hi):
                    continue  # the whole chunk is zero: nothing crosses PCIe, the file keeps its hole
                b, k = k % N_STAGE, k + 1
                for f in writes[b]:
                    f.result()  # the writers have emptied this buffer
                writes[b] = []
                with torch.cuda.stream(copy_stream):
                    st.bufs[b][: hi - lo].copy_(src[lo:hi], non_blocking=True)
                    evs[b].record(copy_stream)
                inflight.append((b, lo, hi))
                if len(inflight) >= 2:  # chunk i is on the copy engine while chunk i-1 goes to the writers
                    drain(*inflight.pop(0))
            while inflight:
                drain(*inflight.pop(0))
            for w in writes:
                for f in w:
                    f.result()
        if fd is not None and sync_file:
            if fmap is not None:
                fmap.flush()
            os.fsync(fd)
    finally:
        for w in (writes or []):
            _settle(w)
        slots.release(slot)
        if fmap is not None:
            arr = writes = None  # (the futures hold slices of the mapping)
            try:
                fmap.close()
            except BufferError:  # a view is still alive somewhere: the mapping goes with it
                pass
        if own_fd:
            os.close(fd)
    dt_s = time.perf_counter() - t0
    last_transfer[what] = {"bytes": nbytes, "s": dt_s, "GBps": nbytes / dt_s / 1e9, "threads": nthr,
                           "bytes_written": written, "zero_blocks_skipped": (int(nblk - flags.sum()) if flags is not None else None),
                           "sink": ("host array" if fd is None else "file (pwrite from pinned staging)" if fmap is None
                                    else "file (memcpy from pinned staging into a shared mapping)")}


def _copy_out(dst_arr, src_mv: memoryview) -> None:
    np.copyto(dst_arr, np.frombuffer(src_mv, dtype=np.uint8), casting="no")


def create_npy(path: str, dtype, shape) -> int:
    """An .npy file of this dtype / shape with numpy's own header and an unwritten payload of the right size (what
    np.lib.format.open_memmap(mode="w+") leaves: the reference creates binaries.npy that way, inference/inference.py:312).
    -> byte offset of the payload."""
    mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.dtype(dtype), shape=tuple(int(v) for v in shape))
    off = int(mm.offset)
    del mm
    return off


def save_npy(engine, tensor, path: str, dtype=None, what: str = "d2h", partial: bool = False, synced: bool = False,
             sparse: bool = True) -> None:
    """np.save(path, tensor) without a host copy of the tensor: header by numpy, payload streamed from HBM.  `dtype`: the
    numpy dtype the file declares (same item size as the tensor's: e.g. uint32 for labels held in an int32 tensor).
    partial: write under `<path>.partial` and rename when complete (a killed run leaves no complete-looking file)."""
    npdt = np.dtype(dtype) if dtype is not None else np.dtype(str(tensor.dtype).replace("torch.", ""))
    if npdt.itemsize != tensor.element_size():
        raise ValueError("save_npy: dtype must have the tensor's item size")
    target = path + ".partial" if partial else path
    if tensor.numel() == 0:  # (nothing to map or stream: numpy writes the header of the empty array)
        with open(target, "wb") as fh:
            np.save(fh, np.empty(tuple(tensor.shape), dtype=npdt))
        if partial:
            os.replace(target, path)
        return
    off = create_npy(target, npdt, tuple(tensor.shape))
    download(engine, tensor, target, offset=off, what=what, synced=synced, sparse=sparse)  # (a fresh file: zero blocks stay holes)
    if partial:
        os.replace(target, path)
