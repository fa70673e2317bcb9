#!/bin/bash
# A/B of kernel variants in ONE gpurun (same device): DLV_ZM_VARIANT=<v> ..., interleaved twice
for rep in 1 2; do for v in "$@"; do
  DLV_ZM_VARIANT=$v DLV_ONE_LANE=1 python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', round(d['ms_per_step'],1), {k:v['avg_us'] for k,v in d['kernels'].items() if 'zmarch' in k})"
done; done
