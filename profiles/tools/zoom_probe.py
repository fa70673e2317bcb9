import sys, time, torch
sys.path.insert(0, '/root/repo')
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.synth import synth_volume_torch
eng = HipEngine(0)
shape = (1024, 2048, 2048)
vol = synth_volume_torch(shape, 2, eng.device)
ds = eng.block_mean_u16(vol, (4, 15, 15))
small = (ds.to(torch.int32) > 0).to(torch.uint8)
ones = torch.ones_like(small)
del vol
def t(fn, n=3):
    fn(); eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        o = fn(); del o
    eng.sync(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
print("real mask", t(lambda: eng.zoom_spline2_u8(small, shape)))
print("all ones ", t(lambda: eng.zoom_spline2_u8(ones, shape)))
out = torch.empty(shape, dtype=torch.uint8, device=eng.device)
print("torch fill 4.3 GB", t(lambda: out.fill_(1)))
