#!/bin/bash
tag=r07s
out=gpurun_out/${tag}_ab.txt
mkdir -p gpurun_out; : > $out
run() {
  name=$(echo "$*" | tr ' =' '__' | tr -d '-')
  python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-extras --no-step-walls --no-isolated "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
  python - "$*" gpurun_out/${tag}_${name}.json >> $out <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:32s} ms_per_step {d['ms_per_step']:9.1f}  mask {d['config']['mask_voxels']} {d['config']['mask_checksum']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
for rep in 1 2; do
  run
  run --diag fuse_layers=131072
  run --diag fuse_layers=65536
  run --diag fuse_layers=32768
  run --diag fuse_layers=49152
done
cat $out
