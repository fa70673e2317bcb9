#!/bin/bash
# A/B of library builds on one box: bash profiles/lib_ab.sh <tag> <workload> "<libA.so libB.so ...>"
# per build, twice interleaved: the bench line (3 lanes, timed region) and the per-kernel view of one lane (HIP events)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-ab}; WL=${2:-c3}; LIBS=${3:-"libdelivr_hip_prev.so libdelivr_hip.so"}
mkdir -p gpurun_out/$TAG
for round in 1 2; do
for L in $LIBS; do
  out=gpurun_out/$TAG/${L%.so}_r$round.json
  DLV_LIB=$L timeout 900 python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-dense > $out 2> gpurun_out/$TAG/${L%.so}_r$round.err
  python3 - "$out" "$L" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{sys.argv[2]:>28s}  ms_per_step {j['ms_per_step']:9.1f}   dominant {r['kernel']} {r['avg_launch_us']:.0f} us frac {r['frac']:.3f}  forward(1 lane) {r.get('forward_tflops') or 0:.0f} TFLOP/s")
    ks = j["kernels"]
    print("      " + "  ".join(f"{k.replace('_f16','')[:26]}:{v['avg_us']:.0f}" for k, v in list(ks.items())[:16]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
