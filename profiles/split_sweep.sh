#!/bin/bash
# CU split x lanes sweep on one box: bench.py lines only (no CPU leg, no extras, no profiling legs)
#   bash profiles/split_sweep.sh <tag> <workload> "<split:lanes[:sw_batch] ...>"
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-sweep}; WL=${2:-c3}; CONFIGS=${3:-"0:3 8:3 8:4 8:6 6:4 10:4"}
mkdir -p gpurun_out/$TAG
for cfg in $CONFIGS; do
  IFS=: read split lanes swb <<< "$cfg"
  out=gpurun_out/$TAG/split${split}_lanes${lanes}_b${swb:-0}.json
  DLV_CU_SPLIT=$split DLV_LANES=$lanes timeout 600 python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras \
      --no-prof --no-isolated --sw-batch ${swb:-0} > $out 2> gpurun_out/$TAG/split${split}_lanes${lanes}_b${swb:-0}.err
  python3 - "$out" "$cfg" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:>10s}  ms_per_step {j['ms_per_step']:9.1f}  dense {j.get('ms_per_step_dense') or 0:9.1f}  mask_voxels {j['config']['mask_voxels']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
