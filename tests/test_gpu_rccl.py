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
