// conv_zreg.hip - host side of the register-resident-weights z-march conv (kernel: conv_zreg_kernel.h, one translation
// unit per instantiation: conv_zreg_inst.hip): weight packing for v_mfma_f32_16x16x32 and the launch dispatcher.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "prec16.h"
#include "conv_zreg.h"

namespace {

// A-fragment pack for v_mfma_f32_16x16x32: out[(((cb16*27 + t)*KS + ks)*64 + lane)*8 + j] =
//     W[cout = cb16*16 + (lane & 15)][cin = ks*32 + 8*(lane >> 4) + j][t]
// (ctot, c0): the weight tensor has ctot input channels and the pack takes channels [c0, c0 + cin) of it - the skip half of an
// UpCat conv whose up half is folded into upconv.hip
template <class P>
__global__ void pack_conv_w16_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin, int ctot, int c0, float wscale) {
    const int KS = cin / 32;
    const long long n = (long long)cout * cin * 27;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int ks = (int)(r % KS);
        r /= KS;
        const int t = (int)(r % 27);
        const int cb = (int)(r / 27);
        const int co = cb * 16 + (lane & 15);
        const int ci = ks * 32 + 8 * (lane >> 4) + j;
        const float v = w[((long long)co * ctot + c0 + ci) * 27 + t] * wscale;  // (a power of two: exact)
        out[i] = (uint16_t)(P::pack2(v, 0.f) & 0xffffu);
    }
}

}  // namespace

int dlv_pack_conv_w16(dlv_ctx* ctx, bool f16, const float* w_f32, uint16_t* out, int cout, int cin, int ctot, int c0, float wscale) {
    if (ctot <= 0) ctot = cin;
    if (f16)
        hipLaunchKernelGGL(pack_conv_w16_kernel<PF16>, dim3(256), dim3(256), 0, ctx->stream, w_f32, out, cout, cin, ctot, c0, wscale);
    else
        hipLaunchKernelGGL(pack_conv_w16_kernel<PBf16>, dim3(256), dim3(256), 0, ctx->stream, w_f32, out, cout, cin, ctot, c0, wscale);
    DLV_LAUNCH_CHECK(ctx, "pack_conv_w16_kernel");
    return DLV_OK;
}

bool dlv_conv3_zreg_supports(int cin, int cout, int c1, int c2, int W) {
    return cout % 32 == 0 && cout > 0 && W >= 32 &&
           ((cin == 32 && c1 == 32 && c2 == 0) || (cin == 64 && ((c1 == 32 && c2 == 32) || (c1 == 64 && c2 == 0))));
}

// the register-resident-weights conv.  ss1 / ss2: InstanceNorm scale/shift [n][C] of the layer that produced in1 / in2
// (applied with Mish while staging), or nullptr for an input that is already final (stem output, pooled tensor, deconv
// output).  Returns the number of partial-sum rows per sample in *nparts.
int dlv_conv3_zreg_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in1, int c1, const void* ss1, const void* in2,
                          int c2, const void* ss2, const void* wpk16, void* out, float* partials, int B, int D, int H, int W,
                          int* nparts, const void* addend) {
    if (!dlv_conv3_zreg_supports(cin, cout, c1, c2, W))
        return dlv_fail(ctx, DLV_EUNSUP, "z-reg conv: needs Cout %% 32 == 0, W >= 32 and inputs of 32, 32+32 or 64 channels");
    if (addend && cin != 32) return dlv_fail(ctx, DLV_EUNSUP, "z-reg conv: an addend needs Cin 32");
    if (ss2 != nullptr || (ss1 != nullptr && c1 != 32))
        return dlv_fail(ctx, DLV_EUNSUP, "z-reg conv: only a 32-channel first input can be activated while staging");
    if ((long long)D * H * W >= (1ll << 26)) return dlv_fail(ctx, DLV_EUNSUP, "z-reg conv: window too large for 32-bit plane offsets (buffer resources of 2 x 16 B x voxels)");
    const int ncb = cout / 32;
    const bool act = ss1 != nullptr || ss2 != nullptr;
    // tile height: 16 rows (Cin = 32 only) for windows large enough that 16 of them fill the chip with z-columns of at
    // least 64 planes.  The choice depends on the window shape only, never on the batch size: the InstanceNorm partial
    // sums are per tile, so a window's result must not depend on how many windows share its launch
    int tyt = 8;
    if (cin == 32 && H % 16 == 0 && (long long)(H / 16) * dlv_cdiv(W, 32) * ncb * dlv_cdiv(D, 64) >= 16) tyt = 16;
#ifdef DLV_DIAG  // A/B switches of the diagnostic library (profiles/tools/tyt_ab.sh, minwg_ab.sh)
    static const int force_tyt = getenv("DLV_ZREG_TYT") ? atoi(getenv("DLV_ZREG_TYT")) : 0;
    if (force_tyt == 8 || (force_tyt == 16 && cin == 32)) tyt = force_tyt;
    static const int min_wg = getenv("DLV_ZREG_MINWG") ? atoi(getenv("DLV_ZREG_MINWG")) : 256;
#else
    const int min_wg = 256;
#endif
    const int tilesY = dlv_cdiv(H, tyt), tilesX = dlv_cdiv(W, 32);
    // split long columns (in multiples of 16 planes) so that small batches still fill 256 CUs; one full-length column per
    // CU beats two half-length ones (16 windows of 64^3: 395 -> 361 us for 64->32, 247 -> 220 us for 32->32)
    int zseg = ((D + 15) / 16) * 16;
    while ((long long)B * tilesY * tilesX * ncb * dlv_cdiv(D, zseg) < min_wg && zseg > 16) zseg = std::max(16, ((zseg / 2 + 15) / 16) * 16);
    const int nseg = dlv_cdiv(D, zseg);
    *nparts = tilesY * tilesX * ((D + 15) / 16);
    char* trash;  // target of the masked-out stores of edge steps / partial tiles
    DLV_TRY(dlv_ws_get(ctx, WS_MISC, 65536, (void**)&trash));
    const int dbg = ctx->zreg_dbg;  // (dlv_diag_set "zreg_dbg") 1 = no interior steps
    ZrArgs a;
    a.in1 = in1; a.ss1 = ss1; a.in2 = in2; a.ss2 = ss2; a.wpk16 = wpk16; a.addend = addend;
    a.out = out; a.partials = partials; a.trash = trash;
    a.c1_8 = c1 / 8; a.c2_8 = c2 / 8; a.D = D; a.H = H; a.W = W; a.tilesX = tilesX; a.zseg = zseg; a.nseg = nseg; a.cout8 = cout / 8;
    a.dbg = dbg;
    a.gx = (unsigned)(tilesY * tilesX); a.gy = (unsigned)(nseg * ncb); a.gz = (unsigned)B;
    if (addend) {
        if (act) {
            if (f16) return tyt == 16 ? dlv_zr_f16_c32_t16_adda1(ctx, a) : dlv_zr_f16_c32_t8_adda1(ctx, a);
            return tyt == 16 ? dlv_zr_bf16_c32_t16_adda1(ctx, a) : dlv_zr_bf16_c32_t8_adda1(ctx, a);
        }
        if (f16) return tyt == 16 ? dlv_zr_f16_c32_t16_add(ctx, a) : dlv_zr_f16_c32_t8_add(ctx, a);
        return tyt == 16 ? dlv_zr_bf16_c32_t16_add(ctx, a) : dlv_zr_bf16_c32_t8_add(ctx, a);
    }
    if (f16) {
        if (cin == 32) {
            if (act) return tyt == 16 ? dlv_zr_f16_c32_t16_a1(ctx, a) : dlv_zr_f16_c32_t8_a1(ctx, a);
            return tyt == 16 ? dlv_zr_f16_c32_t16_a0(ctx, a) : dlv_zr_f16_c32_t8_a0(ctx, a);
        }
        return act ? dlv_zr_f16_c64_t8_a1(ctx, a) : dlv_zr_f16_c64_t8_a0(ctx, a);
    }
    if (cin == 32) {
        if (act) return tyt == 16 ? dlv_zr_bf16_c32_t16_a1(ctx, a) : dlv_zr_bf16_c32_t8_a1(ctx, a);
        return tyt == 16 ? dlv_zr_bf16_c32_t16_a0(ctx, a) : dlv_zr_bf16_c32_t8_a0(ctx, a);
    }
    return act ? dlv_zr_bf16_c64_t8_a1(ctx, a) : dlv_zr_bf16_c64_t8_a0(ctx, a);
}
