#!/bin/bash
for b in 2 4 8 16 32; do
  python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --sw-batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('batch $b two-lane', round(d['ms_per_step'],1), 'norm', round(k['norm_mish_bf16']['total_ms']/2,1), 'zm32', round(k['conv3_zmarch_bf16_c32x32']['total_ms']/2,1))"
  DLV_ONE_LANE=1 python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --sw-batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('batch $b one-lane', round(d['ms_per_step'],1), 'norm', round(k['norm_mish_bf16']['total_ms']/2,1), 'zm32', round(k['conv3_zmarch_bf16_c32x32']['total_ms']/2,1))"
done
