#!/bin/bash
# round-5 measurement set at HEAD: the default bench line, the 1-lane line, the bf16 line, C2, rocprofv3 --kernel-trace --stats of
# the bench command (C3 on 1 and 3 lanes), the deep-level kernel A/B on this box (DLV_DEEP_MASK=0: the round-4 kernels), PMC
# traffic (C2).  Summaries are copied into profiles/ by hand.
#   DLV_GIT_HEAD=$(git rev-parse --short HEAD) bash profiles/run_r05.sh <tag> [quick]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r05}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
LIGHT="--no-cpu-baseline --no-extras --no-dense --no-isolated --no-prof"
for rep in 1 2; do
  for M in 0 2; do
    DLV_DEEP_MASK=$M python3 $R/bench.py --steps 2 --warmup 1 $LIGHT 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DLV_DEEP_MASK=$M rep $rep: ms_per_step', round(j['ms_per_step'],1))"
  done
done > $OUT/deep_mask_ab.txt
cat $OUT/deep_mask_ab.txt
DLV_LANES=1 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/c3_bench_1lane.json 2> $OUT/c3_bench_1lane.err
if [ "${2:-}" != "quick" ]; then
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/c3_bench.json 2> $OUT/c3_bench.err
python3 $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/c3_bench_bf16.json 2> $OUT/c3_bench_bf16.err
python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c2_bench.json 2> $OUT/c2_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_3lane -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras > $OUT/c3_3lane_prof.log 2>&1
DLV_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_1lane -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-prof --no-extras > $OUT/c3_1lane_prof.log 2>&1
rm -f $OUT/*/*/*kernel_trace.csv
find $OUT -name "*kernel_stats.csv"
cut -c1-400 $OUT/c3_bench.json; tail -3 $OUT/c3_bench.err
cd $R
bash profiles/run_pmc_traffic.sh ${TAG}_traffic c2 fp16
fi
