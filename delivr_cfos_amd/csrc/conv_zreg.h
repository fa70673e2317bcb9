// conv_zreg.h - launch arguments of the register-resident-weights z-march conv and its per-instantiation entry points
// (every instantiation of the kernel is its own translation unit, conv_zreg_inst.hip compiled with -DZR_INST_*: each
// takes about a minute to compile, make -j builds them side by side).
#pragma once
struct dlv_ctx;
struct ZrArgs {
    const void *in1, *ss1, *in2, *ss2, *wpk16;
    void* out;
    float* partials;
    char* trash;
    int c1_8, c2_8, D, H, W, tilesX, zseg, nseg, cout8, dbg;
    unsigned gx, gy, gz;
};
// tile rows per configuration (launcher and instances must agree)
#define ZR_TYT_CIN32 8
#define ZR_TYT_CIN64 8
#define ZR_DECLARE(name) int name(dlv_ctx* ctx, const ZrArgs& a)
ZR_DECLARE(dlv_zr_f16_c32_a0);
ZR_DECLARE(dlv_zr_f16_c32_a1);
ZR_DECLARE(dlv_zr_f16_c64_a0);
ZR_DECLARE(dlv_zr_f16_c64_a1);
ZR_DECLARE(dlv_zr_bf16_c32_a0);
ZR_DECLARE(dlv_zr_bf16_c32_a1);
ZR_DECLARE(dlv_zr_bf16_c64_a0);
ZR_DECLARE(dlv_zr_bf16_c64_a1);
