// conv_zreg_inst.hip - one instantiation of conv3_zreg_kernel per object file: compiled with
//   -DZR_INST_NAME=<entry point> -DZR_INST_P=<PF16|PBf16> -DZR_INST_CIN=<32|64> -DZR_INST_TYT=<8|16> -DZR_INST_ACT=<0|1> [-DZR_INST_ADD=1]
// (see the Makefile)
#include "conv_zreg_kernel.h"

#ifndef ZR_INST_ADD
#define ZR_INST_ADD 0
#endif
int ZR_INST_NAME(dlv_ctx* ctx, const ZrArgs& a) { return zr_launch<ZR_INST_P, ZR_INST_CIN, ZR_INST_TYT, (ZR_INST_ACT != 0), ZR_INST_ADD>(ctx, a); }
