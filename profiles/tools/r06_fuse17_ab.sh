#!/bin/bash
# upcat_1.conv_1 (conv block 17) activating its input itself - the one site where activation on load pays - on the other workloads
tag=${1:-r07t}
out=gpurun_out/${tag}_ab.txt
mkdir -p gpurun_out; : > $out
run() {
  wl=$1; steps=$2; shift 2
  name=$(echo "$wl $*" | tr ' =' '__' | tr -d '-')
  python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-dense --no-extras --no-step-walls --no-isolated "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
  python - "$wl $*" gpurun_out/${tag}_${name}.json >> $out <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:40s} ms_per_step {d['ms_per_step']:9.1f}  mask {d['config']['mask_voxels']} {d['config']['mask_checksum']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
for rep in 1 2; do
  run default 1
  run default 1 --diag fuse_layers=131072
  run legacy 2
  run legacy 2 --diag fuse_layers=131072
  run c2 5
  run c2 5 --diag fuse_layers=131072
  run c3 2 --precision bf16
  run c3 2 --precision bf16 --diag fuse_layers=131072
done
cat $out
