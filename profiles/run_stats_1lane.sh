#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command on ONE lane (kernel durations not stretched by overlap)
set -u
export TMPDIR=/tmp
export DLV_LANES=1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-r01_1lane}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c2_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/c3_bench.log 2>&1
tail -1 $OUT/c3_bench.log | cut -c1-300
rm -f $OUT/*/*/*kernel_trace.csv
find $OUT -name "*kernel_stats.csv"
