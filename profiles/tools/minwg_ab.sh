#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for v in 512 256 1024; do
  echo "=== DLV_ZREG_MINWG=$v"
  DLV_ZREG_MINWG=$v python3 profiles/zreg_ab.py 0 3 128,128,2048 2>/dev/null | grep -E "wall|conv3_zreg" | grep -v "^{"
done
