#!/bin/bash
# timing-only ablations of the z-reg conv (make -C delivr_cfos_amd/csrc abl ABL=<mask>): which part of a z step costs what?
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for L in libdelivr_hip.so $(cd delivr_cfos_amd/lib && ls libdelivr_hip_abl*.so | sort -V); do
  [ -f delivr_cfos_amd/lib/$L ] || continue
  echo "=== $L"
  DLV_ALLOW_WRONG_RESULTS=1 DLV_LIB=$L python3 profiles/zreg_ab.py 0 3 128,128,2048 2>&1 | grep -E "conv3_zreg.*d128|ignored|Error" | grep -v "^{"
done
