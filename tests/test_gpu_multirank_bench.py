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
    # N = 1: the file -> file walls of the two steps ran on this volume and left the files the next step reads
    w = j1["step_walls"]
    assert w and w.get("files_ok") is True and w["step2_wall_s"] > 0 and w["step3_wall_next_brain_s"] > 0, w
    assert w["first_brain"]["components"] == w["next_brain"]["components"] == w["components"]
    assert j2["step_walls"] is None
    # N = 2: where each rank's step went
    ph = c2["per_rank_phases"]
    assert len(ph) == 2 and all({"windows_ms", "seam_ms", "finalize_ms", "window_max_gather_ms", "slab_planes"} <= set(p) for p in ph), ph


def test_padded_volume_with_tta_schedule_sharded_three_ways_matches_the_single_rank_line():
    """The shape of the `default` workload (the reference's shipped configuration: windows 96 x 96 x 64, TTA as 3 distinct passes,
    config.json:24-28,63) at plumbing size: a stack of 100 x 180 x 150 zero-padded to 192^3, on one rank and on three."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["--steps", "1", "--warmup", "0", "--workload", "tiny_default", "--no-cpu-baseline", "--no-extras", "--no-isolated", "--no-step-walls"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *args], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout + one.stderr
    j1 = _json_line(one.stdout)
    env["DLV_BENCH_SAME_DEVICE"] = "1"
    env["DLV_LANES"] = "1"
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "3", *args],
                           capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert three.returncode == 0, three.stdout + three.stderr
    j3 = _json_line(three.stdout)
    c1, c3 = j1["config"], j3["config"]
    assert c1["padded_zyx"] == [192, 192, 192] and c1["volume_zyx"] == [100, 180, 150] and c1["passes_per_step"] == 3 and c1["pass_weights"] == [5, 4, 4]
    assert c1["windows"] == 3 * 3 * 5 and c3["windows"] == c1["windows"] and c3["windows_skipped"] == c1["windows_skipped"]
    assert j1["value"] == pytest.approx(100 * 180 * 150 / (j1["ms_per_step"] * 1e-3), rel=1e-6)  # the STACK's voxels, not the padding's
    assert abs(c3["mask_voxels"] - c1["mask_voxels"]) <= 3, (c1["mask_voxels"], c3["mask_voxels"])
    assert c1["mask_voxels"] > 0


def test_eight_ranks_launched_by_bench_itself_match_the_single_rank_line():
    """`python bench.py --gpus 8` with no launcher in the environment: bench.py starts the contract's torch.distributed.run line
    as a child process before it touches the GPU.  Eight ranks on cuda:0 (DLV_BENCH_SAME_DEVICE=1, gloo): the world size of the
    driver's scaling run, on the tiny workload - thin slabs, ranks with few or no windows that run the network."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *COMMON], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout + one.stderr
    j1 = _json_line(one.stdout)
    env["DLV_BENCH_SAME_DEVICE"] = "1"
    env["DLV_LANES"] = "1"  # eight contexts share one device: keep their workspaces small
    eight = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", *COMMON], capture_output=True, text=True,
                           timeout=1500, env=env, cwd=ROOT)
    assert eight.returncode == 0, eight.stdout + eight.stderr
    assert "torch.distributed.run" in eight.stderr  # the child launch line is announced
    j8 = _json_line(eight.stdout)
    assert j8["n_gpus"] == 8 and j8["scaling"] == "strong"
    c1, c8 = j1["config"], j8["config"]
    assert c8["windows"] == c1["windows"] and c8["windows_skipped"] == c1["windows_skipped"]
    per_rank = c8["per_rank_windows"]
    assert len(per_rank) == 8 and sum(w for w, _ in per_rank) == c1["windows"] and sum(s for _, s in per_rank) == c1["windows_skipped"]
    assert abs(c8["mask_voxels"] - c1["mask_voxels"]) <= 8, (c1["mask_voxels"], c8["mask_voxels"])


def test_more_ranks_than_devices_fails_with_a_clear_message():
    """`python bench.py --gpus 2` on a one-GPU box (no DLV_BENCH_SAME_DEVICE): the child ranks say what is missing."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DLV_BENCH_SAME_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "needs 2 devices" in r.stdout + r.stderr, r.stdout + r.stderr
