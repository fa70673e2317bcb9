#!/bin/bash
# windows per launch of the `default` workload (96 x 96 x 64 windows: 12 workgroups of the 16-row z-reg conv per window): 56 (the
# 2^25-voxel rule) gives 672 workgroups = 2.6 rounds of 256 CUs, 64 gives 768 = 3.0, 42 gives 504 = 1.97
mkdir -p gpurun_out
out=gpurun_out/${1:-r06n}_batch_ab.txt
: > $out
for b in 56 64 42 64 56 42; do
  python bench.py --workload default --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-extras --no-step-walls --no-isolated --sw-batch $b 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels']
print('sw_batch $b: ms_per_step', round(d['ms_per_step'], 1), ' hot', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'], 1), 'us  mask', d['config']['mask_voxels'], d['config']['mask_checksum'])" >> $out
done
cat $out
