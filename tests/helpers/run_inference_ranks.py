"""Worker of tests/test_gpu_range_guard.py::test_sharded_run_inference_recovers_in_fp16_on_every_rank: one rank of
`python -m torch.distributed.run --nproc-per-node N run_inference_ranks.py <nifti> <out> <z> <y> <x>` - every rank on cuda:0
(a one-GPU test box), gloo instead of RCCL.  The checkpoint (seeded random, ONE conv block scaled by 1e6: it overflows fp16) is
built on every rank from the same seed; run_inference loads it on rank 0 and broadcasts it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from delivr_cfos_amd.inference import run_inference
    from delivr_cfos_amd.weights import random_state_dict

    nifti, out = sys.argv[1], sys.argv[2]
    shape = tuple(int(v) for v in sys.argv[3:6])
    os.environ["LOCAL_RANK"] = "0"  # every rank on device 0
    dist.init_process_group("gloo")
    sd = {k: v.clone() for k, v in random_state_dict(6).items()}
    sd["module.down_1.convs.conv_0.conv.weight"] *= 1.0e6
    sd["module.down_1.convs.conv_0.conv.bias"] *= 1.0e6
    settings = {"blob_detection": {"window_dimensions": {"window_dim_0": 32, "window_dim_1": 32, "window_dim_2": 32}},
                "mi355x": {"precision": "fp16"}}
    run_inference([nifti], out, (1, 1) + shape, comment="b", tta=False, crop_size=(32, 32, 32), state_dict={"state_dict": sd},
                  settings=settings)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
