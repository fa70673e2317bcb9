"""The REAL RCCL on the one GPU of a test box, before a multi-GPU node ever runs it (replaces torch.nn.DataParallel,
inference/inference.py:217-219):

  * the C-ABI communicator (csrc/multi.hip) with DLV_FORCE_RCCL=1: dlopen of librccl.so, every dlsym, ncclCommInitAll(1), a
    1-rank ncclBroadcast of the weight blob, a grouped ncclSend/ncclRecv to itself compared word for word, and the sharded
    pass through that communicator against the plain pass;
  * the one-process-per-GPU launch of bench.py (`python -m torch.distributed.run --nproc-per-node 1`) with
    DLV_BENCH_FORCE_DIST=1: torch's "nccl" backend (= RCCL) initialised at world size 1, broadcast_weights, balanced_plan's
    all_gather_object, a batch_isend_irecv self exchange, all_reduce and barrier - the calls an 8-GPU run makes.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _maps_rccl():
    with open("/proc/self/maps") as f:
        return sorted({l.split()[-1] for l in f if "librccl" in l})


def test_one_rank_rccl_communicator_runs_broadcast_selftest_and_the_sharded_pass(monkeypatch):
    import torch
    from delivr_cfos_amd.engine import HipComm, HipEngine
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    shape, roi = (96, 64, 64), (32, 32, 32)
    vol = synth_volume_np(shape, seed=33, dense=True)
    vol[:, :10] = 0
    sd = random_state_dict(3)
    one = HipEngine(0)
    one.load_state_dict({"state_dict": sd})
    v = one.to_device(vol)
    acc1 = torch.zeros(shape, dtype=torch.float32, device="cuda")
    cnt1 = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    p = one.make_sw_params(shape, roi, 0.5, None, 0, "fp16")
    st1 = one.sw_infer(p, v, acc1, cnt1)
    one.sync()

    plain = HipComm([0])  # default: one rank needs no transport, librccl is not even loaded by the library
    assert not plain.uses_rccl
    plain.close()
    monkeypatch.setenv("DLV_FORCE_RCCL", "1")
    comm = HipComm([0])
    monkeypatch.delenv("DLV_FORCE_RCCL")
    assert comm.uses_rccl, "DLV_FORCE_RCCL=1 must give a 1-rank RCCL communicator (ncclCommInitAll)"
    assert _maps_rccl(), "librccl.so is not mapped into the process"
    print("RCCL mapped:", _maps_rccl())
    comm.engines[0].load_state_dict({"state_dict": sd})
    comm.bcast_weights(0)                 # ncclBroadcast of the packed blob (in place at one rank)
    comm.selftest(32 * 256 * 256 * 4)     # grouped ncclSend/ncclRecv to itself, one seam of a 256x256 slab; then a broadcast
    plan = comm.make_plan(p, None)
    acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    st = comm.sw_infer_sharded(p, plan, [(0, shape[0])], [v], [acc], [cnt])
    torch.cuda.synchronize()
    assert st[0]["n_windows"] == st1["n_windows"] and st[0]["n_skipped"] == st1["n_skipped"]
    assert torch.equal(acc, acc1) and torch.equal(cnt, cnt1)  # same windows, same order, same device: bit for bit
    comm.close()
    one.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_at_world_one_on_the_nccl_backend_matches_the_plain_line():
    common = ["--steps", "1", "--warmup", "0", "--workload", "tiny", "--no-cpu-baseline", "--no-extras", "--no-isolated"]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout + one.stderr
    j1 = _json_line(one.stdout)
    assert j1["config"]["dist_backend"] is None
    env["DLV_BENCH_FORCE_DIST"] = "1"
    # torch.distributed.run starts a fresh child before anything touches the GPU
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", *common],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout + run.stderr
    j = _json_line(run.stdout)
    assert j["config"]["dist_backend"].startswith("nccl"), j["config"]
    assert j["n_gpus"] == 1 and j["value"] > 0
    c1, c = j1["config"], j["config"]
    assert c["windows"] == c1["windows"] and c["windows_skipped"] == c1["windows_skipped"]
    assert c["mask_voxels"] == c1["mask_voxels"] and c["mask_checksum"] == c1["mask_checksum"]


_NCCL_SELF_SNIPPET = r"""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from delivr_cfos_amd import parallel
from delivr_cfos_amd.engine import HipEngine
from delivr_cfos_amd.synth import synth_volume_np
from delivr_cfos_amd.weights import random_state_dict
from oracle import delivr_oracle as orc

eng = HipEngine(0)
eng.load_state_dict({"state_dict": random_state_dict(3)})
parallel.broadcast_weights(eng, dist, 0)                       # ncclBroadcast of the meta words and of the blob (in place)
parallel.p2p_selftest(dist, eng.device, 0, 1, 8 << 20)

class SelfSeam(parallel.ShardPlan):                            # rank 0 "computed" planes [lo, hi) that rank 0 owns: a seam to itself
    def sends(self, rank):
        return [(0, 8, 24)]
    def recvs(self, rank):
        return [(0, 8, 24)]

shape, roi = (64, 64, 96), (32, 32, 32)
vol = synth_volume_np(shape, seed=8, dense=True)
vol[:, :12] = 0
p = eng.make_sw_params(shape, roi, 0.5, None, 0, "fp16")
plan, slo, shi, v = parallel.balanced_plan(eng, p, lambda lo, hi: eng.to_device(vol[lo:hi]), 1, 0, dist, shape[0], 30, 0)  # all_gather_object on nccl
assert (slo, shi) == (0, shape[0]) and plan.win_ranges == [(0, plan.n_windows)]
acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
cnt = torch.zeros(shape, dtype=torch.uint8, device="cuda")
eng.sw_infer(p, v, acc, cnt)
eng.sync()
before, cbefore = acc.clone(), cnt.clone()
self_plan = SelfSeam(plan.world, plan.n_windows, plan.win_ranges, plan.z_computed, plan.z_owned)
parallel.exchange_seams(acc, self_plan, 0, dist)               # isend / irecv of HBM tensors through RCCL, then the ordered add
parallel.exchange_seams(cnt, self_plan, 0, dist)               # ... and of the uint8 count map
torch.cuda.synchronize()
want = before.clone(); want[8:24] += before[8:24]
cwant = cbefore.clone(); cwant[8:24] += cbefore[8:24]
assert torch.equal(acc, want) and torch.equal(cnt, cwant), "the seam planes did not arrive intact"
acc.copy_(before); cnt.copy_(cbefore)
parallel.exchange_seams(acc, plan, 0, dist)                    # the real plan of one rank: nothing to exchange
slab, _, own = parallel.finalize_owned(eng, plan, 0, acc, cnt, v, shape, 0.5, 30)
full = torch.empty(shape, dtype=torch.uint8, device="cuda")
parallel.gather_slabs(slab, plan, 0, dist, out=full)
one = eng.finalize(acc, cnt, v, shape, 0.5, 30, 0)
assert torch.equal(full, one)
labels, n, stats = parallel.ccl_sharded(eng, full.contiguous(), [(0, shape[0])], 0, dist, shape)   # all_gather_object / gather_object on nccl
lab_ref, n_ref = orc.ccl26(one.cpu().numpy())
assert n == n_ref and np.array_equal(labels.cpu().numpy().view(np.uint32), lab_ref)
ref_stats = orc.cc_stats(lab_ref, n_ref)
np.testing.assert_array_equal(stats["voxel_counts"][1:], ref_stats["voxel_counts"][1:])
t = torch.tensor([3.0], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
dist.destroy_process_group()
print("NCCL_SELF_OK", n)
"""


def test_parallel_py_on_the_nccl_backend_at_world_size_one_with_a_seam_to_itself():
    """Every torch.distributed call of parallel.py on GPU tensors through the nccl (= RCCL) backend, as far as ONE device allows:
    broadcast_weights, balanced_plan's all_gather_object, exchange_seams with a seam the rank sends to ITSELF (isend / irecv of
    fp32 and uint8 HBM planes, the ordered add), finalize_owned, gather_slabs, ccl_sharded's all_gather_object / gather_object,
    all_reduce, barrier.  What stays unexercised until a multi-GPU node runs it: a transfer between two devices (RCCL refuses
    two ranks on one device) - the irecv branch of gather_slabs and the boundary-plane exchange of ccl_sharded."""
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, "-c", _NCCL_SELF_SNIPPET, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "NCCL_SELF_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
