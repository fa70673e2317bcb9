#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp
export DLV_LANES=1
ARGS="$R/bench.py --workload c2 --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-dense --no-extras --no-step-walls"
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -- python3 $ARGS > $OUT/sq2.log 2>&1
du -sh $OUT
