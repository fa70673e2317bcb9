#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) of the default bench command
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
WL=${2:-c3}
OUT=$R/gpurun_out/${1:-traffic}
mkdir -p $OUT
cd /tmp
export DLV_ONE_LANE=1   # PMC passes serialise kernels anyway; rocprofv3 segfaulted with the second lane
ARGS="$R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-prof"
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- python3 $ARGS > $OUT/write.log 2>&1
python3 $R/profiles/make_traffic.py $OUT $WL > $OUT/traffic_$WL.json
cat $OUT/traffic_$WL.json | head -c 1500
rm -f $OUT/*/*/*kernel_trace.csv
