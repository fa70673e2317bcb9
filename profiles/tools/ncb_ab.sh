#!/bin/bash
# generic conv (levels 2-4): cout blocks per workgroup (DLV_GENERIC_NCB) A/B
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for v in 0 2 1; do
  echo "=== DLV_GENERIC_NCB=$v"
  if [ $v = 0 ]; then unset DLV_GENERIC_NCB; else export DLV_GENERIC_NCB=$v; fi
  python3 profiles/zreg_ab.py 0 3 128,128,2048 2>/dev/null | grep -E "wall|conv3_mfma" | grep -v "^{"
done
