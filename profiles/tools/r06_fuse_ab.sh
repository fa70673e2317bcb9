#!/bin/bash
# activate-on-load with the fine-grained staging schedule (conv_zreg_kernel.h: act_sub / flat): C3 pass, interleaved on one box
# usage: bash profiles/tools/r06_fuse_ab.sh <tag> [reps]  -> gpurun_out/<tag>_ab.txt + the bench JSONs (with the per-kernel tables)
tag=${1:-r07}
reps=${2:-2}
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
    z = {n: round(v["total_ms"] / max(v["launches"], 1) * 1000) for n, v in k.items() if n.startswith("conv3_zreg") and "d128" in n}
    print(f"{sys.argv[1]:8s} {sys.argv[2]:28s} ms_per_step {d['ms_per_step']:9.1f}  profiled {d['ms_per_step_profiled']:9.1f}  mask {d['config']['mask_voxels']} {d['config']['mask_checksum']}  kernel-sum {tot:9.1f}  zreg d128 us {z}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
P
}
for rep in $(seq $reps); do
  run c3 2
  run c3 2 --diag fuse_levels=1
  run c3 2 --diag fuse_levels=2
  run c3 2 --diag fuse_levels=3
done
cat $out
