#!/bin/bash
# A/B of environment settings on one box, interleaved twice: bash profiles/env_ab.sh <tag> <workload> "<VAR=val|-> ..."
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-envab}; WL=${2:-c3}; shift; shift
mkdir -p gpurun_out/$TAG
for round in 1 2; do
for setting in "$@"; do
  out=gpurun_out/$TAG/$(echo $setting | tr '= ' '__')_r$round.json
  if [ "$setting" = "-" ]; then
    timeout 900 python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-prof --no-isolated > $out 2>/dev/null
  else
    env $setting timeout 900 python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-prof --no-isolated > $out 2>/dev/null
  fi
  python3 - "$out" "$setting" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:>24s}  ms_per_step {j['ms_per_step']:9.1f}  dense {j.get('ms_per_step_dense') or 0:9.1f}  mask {j['config']['mask_voxels']} {j['config']['mask_checksum']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
