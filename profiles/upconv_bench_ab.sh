#!/bin/bash
# C3 pass with the folded UpCat conv (default) and without (DLV_NO_UPCONV=1), alternating, on one box: ms_per_step of bench.py
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-r04v_upconv_bench_ab}
mkdir -p $OUT
cd /tmp
F="--steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-dense --no-prof --no-isolated"
for i in 1 2; do
  python3 $R/bench.py $F > $OUT/fold_$i.json 2> $OUT/fold_$i.err
  DLV_NO_UPCONV=1 python3 $R/bench.py $F > $OUT/nofold_$i.json 2> $OUT/nofold_$i.err
  DLV_UPCONV_SIMPLE=1 python3 $R/bench.py $F > $OUT/simple_$i.json 2> $OUT/simple_$i.err
done
for f in $OUT/*.json; do echo "$(basename $f) $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"; done | tee $OUT/summary.txt
