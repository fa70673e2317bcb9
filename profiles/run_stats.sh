#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (C2 and C3), copied into profiles/ by hand
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c2_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/c3_bench.log 2>&1
python3 $R/bench.py --workload c3 --steps 2 --warmup 1 > $OUT/c3_bench_plain.json 2> $OUT/c3_bench_plain.err
python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c2_bench_plain.json 2> $OUT/c2_bench_plain.err
find $OUT -name "*kernel_stats.csv"; rm -f $OUT/*/*/*kernel_trace.csv
du -sh $OUT
