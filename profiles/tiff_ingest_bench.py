"""Throughput of the z-plane ingest (dlv_tiff_stack_to_device) on LZW planes of a synthetic brain, against reading the
same planes one by one with libtiff through Pillow (what the reference's cv2.imread loop amounts to).
Usage (GPU box): python profiles/tiff_ingest_bench.py [planes=128] [edge=2048]"""
import os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.synth import synth_volume_torch
from delivr_cfos_amd.downsample.downsample_and_mask import load_stack_to_device

Z = int(sys.argv[1]) if len(sys.argv) > 1 else 128
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = HipEngine(0)
vol = synth_volume_torch((Z, E, E), 2, eng.device).cpu().numpy()
td = tempfile.mkdtemp(prefix="dlv_tiff_")
paths = []
t0 = time.perf_counter()
for z in range(Z):
    p = os.path.join(td, f"plane_{z:04d}.tif")
    Image.fromarray(vol[z]).save(p, compression="tiff_lzw")
    paths.append(p)
t_write = time.perf_counter() - t0
file_bytes = sum(os.path.getsize(p) for p in paths)
raw_bytes = vol.nbytes
res = {"planes": Z, "edge": E, "raw_MB": raw_bytes / 1e6, "file_MB": file_bytes / 1e6, "cores": os.cpu_count()}
t0 = time.perf_counter()
n_ref = min(Z, 16)
for p in paths[:n_ref]:
    a = np.array(Image.open(p))
t_ref = (time.perf_counter() - t0) / n_ref
res["pillow_libtiff_1thread_MBps"] = raw_bytes / Z / t_ref / 1e6
for nt in (1, 8, 32, 0):
    load_stack_to_device(eng, paths[:8], n_threads=nt)  # warm
    t0 = time.perf_counter()
    out = load_stack_to_device(eng, paths, n_threads=nt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[f"native_{nt if nt else 'auto'}threads_MBps"] = raw_bytes / dt / 1e6
assert (out.cpu().numpy() == vol).all()
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tiff_ingest.json"), "w"), indent=1)
for p in paths:
    os.remove(p)
os.rmdir(td)
