#!/bin/bash
# compile one .hip of delivr_cfos_amd/csrc for gfx950 into gpurun_out/tmp with -save-temps and print the resource usage
# of every kernel; usage: profiles/tools/isa.sh conv_zreg.hip [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$1; shift
mkdir -p "$ROOT/gpurun_out/tmp" && cd "$ROOT/gpurun_out/tmp"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -fno-gpu-rdc \
  -save-temps -Rpass-analysis=kernel-resource-usage "$@" -c "$ROOT/delivr_cfos_amd/csrc/$SRC" -o "${SRC%.hip}.o" 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|error|warning: " |
  sed -e 's/.*Function Name: /== /' -e 's/remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste -sd' ' | sed 's/== /\n== /g'
echo
