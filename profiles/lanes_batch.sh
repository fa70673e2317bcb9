#!/bin/bash
for wl in c2 c3; do for cfg in "4 16" "4 8" "2 16" "3 12"; do set -- $cfg
  DLV_LANES=$1 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --sw-batch $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl lanes $1 batch $2', round(d['ms_per_step'],1))"
done; done
