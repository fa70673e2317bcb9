#!/bin/bash
# after isa.sh: per region between two s_barrier of one kernel (name pattern $1): MFMAs, scratch ops, branches
cd "$(dirname "$0")/../../gpurun_out/tmp"
S=$(ls *-hip-amdgcn-amd-amdhsa-gfx950.s | head -1)
L0=$(grep -n "^_ZN.*$1.*:" $S | head -1 | cut -d: -f1); L1=$(awk -v s=$L0 'NR>s && /s_endpgm/{print NR; exit}' $S)
sed -n "${L0},${L1}p" $S > k.s
awk '/s_barrier/{if (m>100) print NR": mfma "m" scratch "sc" branches "br" accrd "ar" accwr "aw" vmov "vm" valu "va" waitcnt "wc; m=0;sc=0;br=0;ar=0;aw=0;vm=0;va=0;wc=0} /v_mfma/{m++} /scratch_/{sc++} /s_cbranch/{br++} /v_accvgpr_read/{ar++} /v_accvgpr_write/{aw++} /v_mov_b32/{vm++} /^\tv_/{va++} /s_waitcnt/{wc++}' k.s
