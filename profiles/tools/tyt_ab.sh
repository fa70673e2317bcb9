#!/bin/bash
# Cin=32 z-reg conv with 8-row vs 16-row tiles: correctness of one block against torch, then kernel times (one lane, 16 windows)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for t in 8 16; do
  echo "=== DLV_ZREG_TYT=$t"
  DLV_ZREG_TYT=$t python3 profiles/tools/zreg_debug.py 1 48,48,64 0 2 2>&1 | tail -6
  DLV_ZREG_TYT=$t python3 profiles/tools/zreg_debug.py 17 128,128,128 0 1 2>&1 | tail -4
  DLV_ZREG_TYT=$t python3 profiles/zreg_ab.py 0 3 128,128,2048 2>/dev/null | grep -E "wall|zreg" 
done
