#!/bin/bash
# A/B of the deep-level conv kernel (conv_deep.hip) against the kernels it replaces: per-label HIP-event times of one dense pass
# (one lane, 16 windows of 128^3 per launch).  DLV_DEEP_MASK: 0 = off, 1 = only the layers the LDS-weights z-march took, 3 = all
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for M in ${1:-0 3}; do
  echo "=== DLV_DEEP_MASK=$M ${3:-}"
  DLV_DEEP_MASK=$M python3 profiles/zreg_ab.py 0 3 128,256,2048 ${2:-fp16} 2>&1 | grep -E "wall|conv3_(deep|mfma|zmarch)|deconv|norm|stats" | grep -v "^{"
done
