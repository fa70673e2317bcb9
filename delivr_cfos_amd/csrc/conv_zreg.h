// conv_zreg.h - launch arguments of the register-resident-weights z-march conv and its per-instantiation entry points
// (every instantiation of the kernel is its own translation unit, conv_zreg_inst.hip compiled with -DZR_INST_*: each
// takes about a minute to compile, make -j builds them side by side).
#pragma once
struct dlv_ctx;
struct ZrArgs {
    const void *in1, *ss1, *in2, *ss2, *wpk16;
    const void* addend = nullptr;  // ADD instantiations: 16-bit tensor in the layout of `out`, added before statistics and pack
    void* out;
    float* partials;
    char* trash;
    int c1_8, c2_8, D, H, W, tilesX, zseg, nseg, cout8, dbg;
    unsigned gx, gy, gz;
};
// entry points dlv_zr_<format>_c<Cin>_t<tile rows>_a<activate the first input while staging>; the launcher
// (conv_zreg.hip) picks the tile height: 16 rows for 32-channel inputs whenever the launch still fills the chip with
// long z-columns (twice the MFMAs per staged plane and per barrier, 1.20x instead of 1.33x halo re-reads), else 8
#define ZR_DECLARE(name) int name(dlv_ctx* ctx, const ZrArgs& a)
ZR_DECLARE(dlv_zr_f16_c32_t8_a0);
ZR_DECLARE(dlv_zr_f16_c32_t8_a1);
ZR_DECLARE(dlv_zr_f16_c32_t16_a0);
ZR_DECLARE(dlv_zr_f16_c32_t16_a1);
ZR_DECLARE(dlv_zr_f16_c64_t8_a0);
ZR_DECLARE(dlv_zr_f16_c64_t8_a1);
ZR_DECLARE(dlv_zr_bf16_c32_t8_a0);
ZR_DECLARE(dlv_zr_bf16_c32_t8_a1);
ZR_DECLARE(dlv_zr_bf16_c32_t16_a0);
ZR_DECLARE(dlv_zr_bf16_c32_t16_a1);
ZR_DECLARE(dlv_zr_bf16_c64_t8_a0);
ZR_DECLARE(dlv_zr_bf16_c64_t8_a1);
// Cin 32 with an addend (the up half of an UpCat conv, upconv.hip)
ZR_DECLARE(dlv_zr_f16_c32_t8_add);
ZR_DECLARE(dlv_zr_f16_c32_t16_add);
ZR_DECLARE(dlv_zr_bf16_c32_t8_add);
ZR_DECLARE(dlv_zr_bf16_c32_t16_add);
// ... whose 32-channel input is still raw (scale/shift + Mish applied while staging)
ZR_DECLARE(dlv_zr_f16_c32_t8_adda1);
ZR_DECLARE(dlv_zr_f16_c32_t16_adda1);
ZR_DECLARE(dlv_zr_bf16_c32_t8_adda1);
ZR_DECLARE(dlv_zr_bf16_c32_t16_adda1);
