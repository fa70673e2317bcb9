#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-ccl_prof}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/profiles/ccl_prof.py 512 2048 2048 > $OUT/log.txt 2>&1
grep "^rep" $OUT/log.txt
rm -f $OUT/*/*/*kernel_trace.csv
python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/run/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-60s %5s %10.1f us %6s%%'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
