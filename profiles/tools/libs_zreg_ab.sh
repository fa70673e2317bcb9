#!/bin/bash
# per-label HIP-event times of the z-reg convs for several library builds (files under delivr_cfos_amd/lib/), interleaved:
#   bash profiles/tools/libs_zreg_ab.sh "libdelivr_hip.so libdelivr_hip_x.so" [rounds]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in $(seq 1 ${2:-2}); do
for L in ${1:-libdelivr_hip.so}; do
  echo "=== $L"
  DLV_LIB=$L python3 profiles/zreg_ab.py 0 3 128,256,2048 fp16 2>&1 | grep -E "conv3_zreg" | grep -v "^{"
done
done
