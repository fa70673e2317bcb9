#!/bin/bash
# SQ counters of the folded-upcat kernels (two passes), one forward set on a 256x256x512 dense volume
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc_upconv}
mkdir -p $OUT
cd /tmp
ARGS="$R/profiles/upconv_ab.py 1"
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -- python3 $ARGS > $OUT/sq2.log 2>&1
python3 $R/profiles/pmc_summary.py $OUT 2>&1 | grep -A18 "upconv2_kernel<PF16"
rm -rf $OUT/sq/*/*kernel_trace.csv $OUT/sq2/*/*kernel_trace.csv
