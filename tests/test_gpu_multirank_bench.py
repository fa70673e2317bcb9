"""The N > 1 launch of bench.py as the driver issues it - `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
- as a fresh child process on the ONE GPU of a test box: DLV_BENCH_SAME_DEVICE=1 puts both ranks on cuda:0 and swaps RCCL for
gloo (RCCL refuses two ranks on one device); everything else - the balanced shard plan, slab-resident volume / accumulator,
weight broadcast, seam exchange, slab finalize, the JSON line - is the code an 8-GPU run takes.  Replaces the reference's
torch.nn.DataParallel (inference/inference.py:217-219)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "1", "--warmup", "0", "--workload", "tiny", "--no-cpu-baseline", "--no-extras", "--no-isolated"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_two_ranks_through_torch_distributed_run_match_the_single_rank_line():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *COMMON], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout + one.stderr
    j1 = _json_line(one.stdout)
    env["DLV_BENCH_SAME_DEVICE"] = "1"
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stdout + two.stderr
    j2 = _json_line(two.stdout)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2
    assert j2["metric"] == j1["metric"] and j2["unit"] == "voxels/s" and j2["scaling"] == "strong"
    c1, c2 = j1["config"], j2["config"]
    assert c2["windows"] == c1["windows"] and c2["windows_skipped"] == c1["windows_skipped"]  # summed over the ranks
    # same mask: the seam sums associate differently in fp32, so a voxel whose mean logit is within rounding of 0 may flip
    assert abs(c2["mask_voxels"] - c1["mask_voxels"]) <= 2, (c1["mask_voxels"], c2["mask_voxels"])
    if c2["mask_voxels"] == c1["mask_voxels"]:
        assert c2["mask_checksum"] == c1["mask_checksum"]
    assert j2["value"] > 0 and j2["ms_per_step"] > 0
    assert j2["cpu_baseline"] is None  # the CPU leg is reported at N = 1 only
