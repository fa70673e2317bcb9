#!/bin/bash
# 16-row tiles for the 32-channel convs of level 1 (64^3 per window): 128 workgroups of full columns become 256 of 32 planes
# (diagnostic library: DLV_ZREG_TYT forces the tile height).  Per-label times, one lane, twice interleaved.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for T in 0 16 0 16; do
  echo "=== DLV_ZREG_TYT=$T"
  DLV_LIB=libdelivr_hip_diag.so DLV_ZREG_TYT=$T python3 profiles/zreg_ab.py 0 3 128,256,2048 fp16 2>&1 | grep -E "wall|conv3_zreg" | grep -v "^{"
done
