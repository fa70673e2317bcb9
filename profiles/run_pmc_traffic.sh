#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) of the default bench command
#   DLV_GIT_HEAD=$(git rev-parse --short HEAD) bash profiles/run_pmc_traffic.sh <tag> <workload> <precision>
# (the JSON records the sha256 of the library that ran and of its sources: bench.py shows the figures only for that library)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
WL=${2:-c2}
PREC=${3:-fp16}
OUT=$R/gpurun_out/${1:-traffic}
mkdir -p $OUT
cd /tmp
export DLV_LANES=1   # PMC passes serialise kernels anyway; one lane = the launch log is in stream order
ARGS="$R/bench.py --workload $WL --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-dense --no-extras --no-step-walls"
DLV_LAUNCH_LOG=$OUT/fetch.launches rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- python3 $ARGS > $OUT/fetch.log 2>&1
DLV_LAUNCH_LOG=$OUT/write.launches rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- python3 $ARGS > $OUT/write.log 2>&1
python3 $R/profiles/make_traffic.py $OUT $WL $PREC > $OUT/traffic_r06_$WL.json
python3 -c "
import json; j=json.load(open('$OUT/traffic_r06_$WL.json')); print('unmatched', j['unmatched_launches'])
for k,v in j['kernels'].items(): print(f\"{k:28s} n={v['launches']:5d} traffic={v['traffic_bytes']/1e6:9.1f} MB alg={v['algorithmic_bytes']/1e6:9.1f} MB ratio={v['traffic_over_algorithmic']:.2f}\")
"
rm -f $OUT/*/*/*kernel_trace.csv
gzip -f $OUT/*/*/*counter_collection.csv 2>/dev/null
