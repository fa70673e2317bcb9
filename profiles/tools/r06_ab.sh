#!/bin/bash
# round 6 A/B on one box: activate-on-load in the z-reg convs (fuse_levels), the deep kernel on sub-tile levels (deep_small)
# usage: bash profiles/tools/r06_ab.sh <tag>   -> gpurun_out/<tag>_ab.txt (one line per run) + the bench JSONs
tag=${1:-r06c}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
: > $out
run() {  # workload steps diag...
  wl=$1; steps=$2; shift 2
  name=$(echo "$wl $*" | tr ' =' '__' | tr -d '-')
  python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-dense --no-extras --no-step-walls --no-isolated "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
  python - "$wl" "$*" gpurun_out/${tag}_${name}.json >> $out <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    k = d["kernels"]
    tot = sum(v["total_ms"] for v in k.values())
    print(f"{sys.argv[1]:8s} {sys.argv[2]:28s} ms_per_step {d['ms_per_step']:9.1f}  profiled {d['ms_per_step_profiled']:9.1f}  mask {d['config']['mask_voxels']} {d['config']['mask_checksum']}  kernel-sum {tot:9.1f}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
P
}
for rep in 1 2; do
  run c3 2
  run c3 2 --diag fuse_levels=1
  run c3 2 --diag fuse_levels=2
  run c3 2 --diag fuse_levels=3
done
run default 1
run default 1 --diag deep_small=0
run default 1 --diag fuse_levels=3
run default 1
run legacy 1
run legacy 1 --diag deep_small=0
run legacy 1 --diag fuse_levels=3
run legacy 1
cat $out
