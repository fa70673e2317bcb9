import json, os, shutil, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
import torch
from delivr_cfos_amd.downsample import downsample_and_mask as dm
from delivr_cfos_amd.engine import shared_engine
from delivr_cfos_amd.synth import synth_planes_torch
from delivr_cfos_amd.tiffio import write_tiff_plane
Z = 256
d = tempfile.mkdtemp(prefix="dlv_tp2_", dir="/dev/shm")
out = {}
try:
    eng = shared_engine(0)
    with ThreadPoolExecutor(32) as ex:
        for lo in range(0, Z, 64):
            blk = synth_planes_torch((1024, 2048, 2048), 2, eng.device, 384 + lo, 384 + lo + 64).cpu().numpy()
            list(ex.map(lambda i: write_tiff_plane(os.path.join(d, f"Z{lo + i:04d}.tif"), blk[i]), range(64)))
    planes = sorted(os.path.join(d, f) for f in os.listdir(d))
    # single-thread decode of one plane
    ts = []
    for p in planes[:8]:
        t0 = time.perf_counter(); dm.read_tiff_plane(p); ts.append(time.perf_counter() - t0)
    out["one_plane_decode_ms"] = [round(t * 1e3, 1) for t in ts]
    # python threads calling the single-plane reader (ctypes releases the GIL): scaling of the decode alone, no staging / copies
    for n in (8, 16, 32, 64):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(n) as ex:
            list(ex.map(dm.read_tiff_plane, planes))
        dt = time.perf_counter() - t0
        out[f"decode_only_{n}thr_s"] = round(dt, 3)
    # file reads alone
    t0 = time.perf_counter()
    with ThreadPoolExecutor(32) as ex:
        list(ex.map(lambda p: len(open(p, "rb").read()), planes))
    out["file_reads_32thr_s"] = round(time.perf_counter() - t0, 3)
    dst = torch.empty((Z, 2048, 2048), dtype=torch.uint16, device=eng.device)
    for n_pl in (32, 64, 128, 256):
        for rep in range(2):
            t0 = time.perf_counter(); dm.load_stack_to_device(eng, planes[:n_pl], out=dst, n_threads=32); eng.sync()
            out[f"stack_{n_pl}planes_32thr_s_rep{rep}"] = round(time.perf_counter() - t0, 3)
    # pinned allocation alone
    import ctypes
    t0 = time.perf_counter(); x = torch.empty(256 << 20, dtype=torch.uint8).pin_memory(); out["pin_256MB_s"] = round(time.perf_counter() - t0, 3)
    out["cpu_count"] = os.cpu_count(); out["sched_affinity"] = len(os.sched_getaffinity(0))
    try:
        out["cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError as e:
        out["cpu_max"] = str(e)
finally:
    shutil.rmtree(d, ignore_errors=True)
print(json.dumps(out))
