#!/bin/bash
# round-6 measurement set at HEAD on ONE box: the default bench line (c3, with the step walls), the 1-lane line, the bf16 line,
# C2, the reference's shipped configuration (96 x 96 x 64 windows + TTA) and run_inference's own default (64 x 64 x 32),
# rocprofv3 --kernel-trace --stats of the bench command (c3 on 1 and 3 lanes, `default` on 1 lane), one --marker-trace run
# (the roctx ranges of a forward's layers), PMC traffic (C2).  Summaries are copied into profiles/ by hand.
#   DLV_GIT_HEAD=$(git rev-parse --short HEAD) bash profiles/run_r06.sh <tag> [quick]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/c3_bench.json 2> $OUT/c3_bench.err
DLV_LANES=1 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-step-walls > $OUT/c3_bench_1lane.json 2> $OUT/c3_bench_1lane.err
python3 $R/bench.py --workload default --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-step-walls > $OUT/default_bench.json 2> $OUT/default_bench.err
if [ "${2:-}" != "quick" ]; then
python3 $R/bench.py --workload legacy --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-step-walls > $OUT/legacy_bench.json 2> $OUT/legacy_bench.err
python3 $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-step-walls > $OUT/c3_bench_bf16.json 2> $OUT/c3_bench_bf16.err
python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-step-walls > $OUT/c2_bench.json 2> $OUT/c2_bench.err
LIGHT="--steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras --no-step-walls"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_3lane -- python3 $R/bench.py --workload c3 $LIGHT > $OUT/c3_3lane_prof.log 2>&1
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_1lane -- python3 $R/bench.py --workload c3 $LIGHT > $OUT/c3_1lane_prof.log 2>&1
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default_1lane -- python3 $R/bench.py --workload default $LIGHT > $OUT/default_1lane_prof.log 2>&1
DLV_LANES=1 DLV_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/tiny_markers -- python3 $R/bench.py --workload tiny $LIGHT > $OUT/tiny_markers_prof.log 2>&1
rm -f $OUT/*/*/*kernel_trace.csv
find $OUT -name "*stats.csv"
cd $R
bash profiles/run_pmc_traffic.sh ${TAG}_traffic c2 fp16
fi
cut -c1-400 $OUT/c3_bench.json; tail -3 $OUT/c3_bench.err
