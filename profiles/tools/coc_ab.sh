#!/bin/bash
# A/B of the cout-complete z-reg waves (DLV_ZREG_COC = tile rows: 12 / 8; 0 = the default kernels): per-label HIP-event times of
# one dense pass (one lane, 16 windows of 128^3 per launch), interleaved
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for C in ${1:-0 12 8 0 12 8}; do
  echo "=== DLV_ZREG_COC=$C"
  DLV_ZREG_COC=$C python3 profiles/zreg_ab.py 0 3 128,256,2048 ${2:-fp16} 2>&1 | grep -E "wall|conv3_zreg" | grep -v "^{"
done
