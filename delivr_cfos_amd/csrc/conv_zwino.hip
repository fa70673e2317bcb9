// conv_zwino.hip - host side of the Winograd F(2,3)-along-x variant of the register-resident-weights z-march conv
// (kernel: conv_zwino_kernel.h): weight transform + A-fragment pack, launch.  fp16 operands only (bf16 has no packed
// add on gfx950 and 8 significant bits do not survive the input transform); Cin = 32 -> Cout blocks of 32.
// Reference: the Conv3d blocks of MONAI's BasicUNet at the two top levels (inference/inference.py:190-197).
#include "conv_zwino_kernel.h"

namespace {

// out[(((cb16*36 + t)*64 + lane)*8 + j] = U_nu[cout = cb16*16 + (lane & 15)][cin = 8*(lane >> 4) + j][kz][ky],
// t = (kz*3 + ky)*4 + nu;  U = G g along kx in fp32: (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2), rounded to fp16 once
__global__ void pack_conv_wino_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin) {
    const long long n = (long long)cout * cin * 36;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int t = (int)(r % 36);
        const int cb = (int)(r / 36);
        const int nu = t & 3, kzy = t >> 2;
        const int co = cb * 16 + (lane & 15);
        const int ci = 8 * (lane >> 4) + j;
        const float* g = w + ((long long)co * cin + ci) * 27 + kzy * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        float u;
        if (nu == 0) u = g0;
        else if (nu == 1) u = __fmul_rn(__fadd_rn(__fadd_rn(g0, g1), g2), 0.5f);
        else if (nu == 2) u = __fmul_rn(__fadd_rn(__fsub_rn(g0, g1), g2), 0.5f);
        else u = g2;
        out[i] = (uint16_t)(PF16::pack2(u, 0.f) & 0xffffu);
    }
}

}  // namespace

int dlv_pack_conv_wino(dlv_ctx* ctx, const float* w_f32, uint16_t* out, int cout, int cin) {
    if (cin != 32 || cout % 16) return dlv_fail(ctx, DLV_EUNSUP, "winograd pack: Cin 32, Cout %% 16 == 0");
    hipLaunchKernelGGL(pack_conv_wino_kernel, dim3(64), dim3(256), 0, ctx->stream, w_f32, out, cout, cin);
    DLV_LAUNCH_CHECK(ctx, "pack_conv_wino_kernel");
    return DLV_OK;
}

bool dlv_conv3_zwino_supports(int cin, int cout, int c1, int c2, int W) {
    return cout % 32 == 0 && cout > 0 && W >= 32 && W % 2 == 0 && cin == 32 && c1 == 32 && c2 == 0;
}

int dlv_conv3_zwino_launch(dlv_ctx* ctx, int cin, int cout, const void* in1, const void* wwino, void* out, float* partials, int B, int D,
                           int H, int W, int* nparts) {
    if (!dlv_conv3_zwino_supports(cin, cout, cin, 0, W)) return dlv_fail(ctx, DLV_EUNSUP, "winograd z-reg conv: Cin 32, Cout %% 32 == 0, even W >= 32");
    if ((long long)D * H * W >= (1ll << 26)) return dlv_fail(ctx, DLV_EUNSUP, "winograd z-reg conv: window too large for 32-bit plane offsets");
    const int ncb = cout / 32;
    const int tilesY = dlv_cdiv(H, ZW_TYT), tilesX = dlv_cdiv(W, 32);
    int zseg = ((D + 15) / 16) * 16;
    while ((long long)B * tilesY * tilesX * ncb * dlv_cdiv(D, zseg) < 256 && zseg > 16) zseg = std::max(16, ((zseg / 2 + 15) / 16) * 16);
    const int nseg = dlv_cdiv(D, zseg);
    *nparts = tilesY * tilesX * ((D + 15) / 16);
    char* trash;
    DLV_TRY(dlv_ws_get(ctx, WS_MISC, 65536, (void**)&trash));
    static const int dbg = getenv("DLV_ZREG_DBG") ? atoi(getenv("DLV_ZREG_DBG")) : 0;  // development: 1 = no interior steps
    static dlv_attr_bits attr_set{0};
    if (!dlv_attr_is_set(attr_set, ctx->device)) {
        DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_zwino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZwCfg::LDS_BYTES));
        dlv_attr_mark(attr_set, ctx->device);
    }
    hipLaunchKernelGGL(conv3_zwino_kernel, dim3(tilesY * tilesX, nseg * ncb, B), dim3(256), ZwCfg::LDS_BYTES, ctx->stream, (const uint4*)in1,
                       (const uint4*)wwino, (uint4*)out, partials, D, H, W, tilesX, zseg, nseg, cout / 8, dbg, trash);
    DLV_LAUNCH_CHECK(ctx, "conv3_zwino_kernel");
    return DLV_OK;
}
