#!/bin/bash
# pass time against the number of pipeline lanes (streams with their own activation workspace); C3, 2 steps each
for rep in 1 2; do for l in 1 2 3 4; do
  DLV_LANES=$l python bench.py --workload ${1:-c3} --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-dense 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $l', round(d['ms_per_step'],1))"
done; done
