#!/bin/bash
# round-4 measurement set at HEAD: the default bench line, the 1-lane line, rocprofv3 --kernel-trace --stats of the bench
# command (C3 on 1 and 3 lanes; C2 on 1 and 3 lanes), the bf16 line, PMC traffic (C2).  Summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r04m}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/c3_bench.json 2> $OUT/c3_bench.err
DLV_LANES=1 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/c3_bench_1lane.json 2> $OUT/c3_bench_1lane.err
python3 $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/c3_bench_bf16.json 2> $OUT/c3_bench_bf16.err
python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c2_bench.json 2> $OUT/c2_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_3lane -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras > $OUT/c3_3lane_prof.log 2>&1
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_1lane -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras > $OUT/c3_1lane_prof.log 2>&1
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_1lane -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras > $OUT/c2_1lane_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras > $OUT/c2_prof.log 2>&1
rm -f $OUT/*/*/*kernel_trace.csv
find $OUT -name "*kernel_stats.csv"
cut -c1-400 $OUT/c3_bench.json; tail -3 $OUT/c3_bench.err
cd $R
bash profiles/run_pmc_traffic.sh ${TAG}_traffic c2 fp16
