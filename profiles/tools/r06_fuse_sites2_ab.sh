#!/bin/bash
tag=r07w
out=gpurun_out/${tag}_ab.txt
mkdir -p gpurun_out; : > $out
run() {
  name=$(echo "$*" | tr ' =' '__' | tr -d '-')
  python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-extras --no-step-walls --no-isolated "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
  python - "$*" gpurun_out/${tag}_${name}.json >> $out <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:36s} ms_per_step {d['ms_per_step']:9.1f}  mask {d['config']['mask_voxels']} {d['config']['mask_checksum']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
for rep in 1 2; do
  run                                   # default: block 17
  run --diag fuse_layers=131080         # 17 + 3  (down_1.conv_1 activates the raw output of down_1.conv_0)
  run --diag fuse_layers=163840         # 17 + 15 (upcat_2.conv_1)
  run --diag fuse_layers=0              # nothing fused
done
cat $out
