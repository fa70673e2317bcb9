#!/bin/bash
# rocprofv3 kernel stats of one 16-window forward batch for each library build named on the command line (files under
# delivr_cfos_amd/lib/): bash profiles/tools/lib_ab_stats.sh <pattern> libA.so libB.so ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
PAT=$1; shift
cd /tmp
for L in "$@"; do
  echo "=== $L"
  rm -rf /tmp/ab_$L
  DLV_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$L -- python3 $R/profiles/zreg_ab.py 0 3 128,128,2048 > /dev/null 2>&1
  python3 - "$PAT" /tmp/ab_$L <<'PY'
import csv, glob, re, sys
pat, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
    if re.search(pat, n):
        print(f"  {n[:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
