#!/bin/bash
# round-2 measurement set: plain bench lines (4 lanes, 1 lane), rocprofv3 --kernel-trace --stats of the bench command
# (C2 on 1 lane and 4 lanes, C3 on 1 lane); summaries are copied into profiles/ by hand
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/c3_bench.json 2> $OUT/c3_bench.err
DLV_LANES=1 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c3_bench_1lane.json 2> $OUT/c3_bench_1lane.err
python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c2_bench.json 2> $OUT/c2_bench.err
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_1lane -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-prof > $OUT/c2_1lane_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-prof > $OUT/c2_prof.log 2>&1
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_1lane -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof > $OUT/c3_1lane_prof.log 2>&1
rm -f $OUT/*/*/*kernel_trace.csv
find $OUT -name "*kernel_stats.csv"
cut -c1-600 $OUT/c3_bench.json; tail -3 $OUT/c3_bench.err
