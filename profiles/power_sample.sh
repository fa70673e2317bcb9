#!/bin/bash
# shader clock and socket power while the benchmark's passes run: rocm-smi sampled every 0.5 s beside bench.py
#   bash profiles/power_sample.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-power}; shift
mkdir -p gpurun_out/$TAG
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use"; sleep 0.5; done ) > gpurun_out/$TAG/smi.txt &
SAMPLER=$!
python3 bench.py --no-cpu-baseline --no-extras --no-prof --no-isolated --steps 3 --warmup 1 "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
kill $SAMPLER
python3 - gpurun_out/$TAG/smi.txt <<'PY'
import re, sys, statistics
txt = open(sys.argv[1]).read()
sclk = [int(m) for m in re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)]
pw = [float(m) for m in re.findall(r"Power \(W\): ([\d.]+)", txt)]
use = [int(m) for m in re.findall(r"GPU use \(%\): (\d+)", txt)]
busy = [i for i, u in enumerate(use) if u >= 90]
def pick(a):
    return [a[i] for i in busy if i < len(a)] or a
for name, a in (("sclk MHz", sclk), ("power W", pw)):
    b = pick(a)
    if b:
        print(f"{name}: samples {len(a)}, while busy: median {statistics.median(b):.0f}  min {min(b):.0f}  max {max(b):.0f}")
PY
