#!/bin/bash
# how do the kernels scale with the number of CUs they may use?  (HSA_CU_MASK applies to every queue of the process)
# 16 dense 128^3 windows, one lane; prints wall time and the level-0 kernels per mask
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for n in 255 223 191 127; do
  echo "== CUs 0-$n"
  HSA_CU_MASK=0:0-$n python3 profiles/zreg_ab.py 0 2 128,128,2048 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=j['variants']['0']
print('wall_ms', round(v['wall_ms_med'],2))
for k,e in sorted(v['kernels'].items(), key=lambda kv:-kv[1]['us_med'])[:9]:
    print(f\"  {k:32s} {e['us_med']:9.1f} us {e['tflops_med']:8.1f}\")
"
done
