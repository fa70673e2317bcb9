#!/bin/bash
# the round's closing measurement set on ONE box, in the order that lets every bench line carry `roofline.traffic`: first the PMC
# traffic passes of the library as built (copied over profiles/traffic_r06_c2.json on the box: bench.py shows the figures only
# for the library whose sha256 the file records), then the default bench line (c3, every leg), the shipped configuration
# (96 x 96 x 64 + TTA), run_inference's own default (64 x 64 x 32), C2, and rocprofv3 --kernel-trace --stats of the bench
# command on one lane (c3).
#   DLV_GIT_HEAD=$(git rev-parse --short HEAD) bash profiles/run_r06_final.sh <tag>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06final}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
bash profiles/run_pmc_traffic.sh ${TAG}_traffic c2 fp16 > $OUT/traffic.log 2>&1
cp $R/gpurun_out/${TAG}_traffic/traffic_r06_c2.json $R/profiles/traffic_r06_c2.json
cd /tmp
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/c3_bench.json 2> $OUT/c3_bench.err
python3 $R/bench.py --workload default --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-step-walls > $OUT/default_bench.json 2> $OUT/default_bench.err
python3 $R/bench.py --workload legacy --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-step-walls > $OUT/legacy_bench.json 2> $OUT/legacy_bench.err
python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-step-walls > $OUT/c2_bench.json 2> $OUT/c2_bench.err
LIGHT="--steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras --no-step-walls"
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_1lane -- python3 $R/bench.py --workload c3 $LIGHT > $OUT/c3_1lane_prof.log 2>&1
rm -f $OUT/*/*/*kernel_trace.csv
find $OUT -name "*stats.csv"
cut -c1-600 $OUT/c3_bench.json; tail -3 $OUT/c3_bench.err
