/* A host in plain C driving the whole path through the C ABI of libdelivr_hip.so - no Python, no PyTorch:
 * weights -> HBM, uint16 volume -> HBM, one sliding-window pass, threshold + eroded re-mask, CCL-26 + statistics.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -o c_host -Ldelivr_cfos_amd/lib -ldelivr_hip \
 *       -Wl,-rpath,$PWD/delivr_cfos_amd/lib -Wl,--allow-shlib-undefined -lm
 *   ./c_host            # needs an MI355X; prints the number of mask voxels and components
 *
 * Weights: a deterministic LCG stands in for a checkpoint (same topology as MONAI BasicUNet(3,1,1,
 * (32,32,64,128,256,32), act=mish, norm=instance-affine); a real host passes the arrays of its state_dict). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "delivr_hip.h"

static uint32_t lcg_state = 12345u;
static float lcg_uniform(void) { /* (-1, 1) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((lcg_state >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}
static float* filled(size_t n, float scale, float offset) {
    float* p = (float*)malloc(n * sizeof(float));
    size_t i;
    if (!p) exit(3);
    for (i = 0; i < n; ++i) p[i] = offset + scale * lcg_uniform();
    return p;
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != DLV_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? dlv_last_error(ctx) : "(no context)"); \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

int main(void) {
    dlv_ctx* ctx = NULL;
    const int f[6] = {32, 32, 64, 128, 256, 32};
    /* channels per conv in forward order (see the header comment of dlv_unet_weights) */
    const int cin[DLV_N_CONV] = {1, 32, 32, 32, 32, 64, 64, 128, 128, 256, 256, 128, 128, 64, 64, 32, 64, 32};
    const int cout[DLV_N_CONV] = {32, 32, 32, 32, 64, 64, 128, 128, 256, 256, 128, 128, 64, 64, 32, 32, 32, 32};
    const int dcin[DLV_N_DECONV] = {256, 128, 64, 32}, dcout[DLV_N_DECONV] = {128, 64, 32, 32};
    dlv_unet_weights w;
    int i, k;
    const int Z = 48, Y = 64, X = 64, roi = 32;
    const size_t nvox = (size_t)Z * Y * X;
    uint16_t* vol = (uint16_t*)malloc(nvox * 2);
    uint8_t* mask = (uint8_t*)malloc(nvox);
    void *vol_dev = NULL, *acc_dev = NULL, *mask_dev = NULL, *lab_dev = NULL;
    dlv_sw_params p;
    dlv_sw_stats st;
    uint64_t ncomp = 0;
    size_t fg = 0;

    if (dlv_ctx_create(0, NULL, &ctx) != DLV_OK) {
        fprintf(stderr, "dlv_ctx_create failed: this program needs an MI355X (there is no CPU fallback)\n");
        return 2;
    }
    for (k = 0; k < 6; ++k) w.features[k] = f[k];
    for (i = 0; i < DLV_N_CONV; ++i) {
        const size_t nw = (size_t)cout[i] * cin[i] * 27;
        w.conv_w[i] = filled(nw, 1.0f / (float)(cin[i] * 27 > 27 ? 40 : 6), 0.f);
        w.conv_b[i] = filled((size_t)cout[i], 0.05f, 0.f);
        w.norm_g[i] = filled((size_t)cout[i], 0.2f, 1.0f);
        w.norm_b[i] = filled((size_t)cout[i], 0.2f, 0.f);
    }
    for (i = 0; i < DLV_N_DECONV; ++i) {
        w.deconv_w[i] = filled((size_t)dcin[i] * dcout[i] * 8, 0.05f, 0.f);
        w.deconv_b[i] = filled((size_t)dcout[i], 0.05f, 0.f);
    }
    w.final_w = filled(32, 0.3f, 0.f);
    w.final_b = filled(1, 0.1f, 0.f);
    CHECK(dlv_unet_load(ctx, &w));

    /* a synthetic volume: tissue everywhere but a background margin in x (those windows are skipped) */
    for (i = 0; i < (int)nvox; ++i) vol[i] = (i % X) < 30 ? (uint16_t)(2000 + 1500 * lcg_uniform()) : 0;
    CHECK(dlv_malloc(ctx, nvox * 2, &vol_dev));
    CHECK(dlv_malloc(ctx, nvox * 4, &acc_dev));
    CHECK(dlv_malloc(ctx, nvox, &mask_dev));
    CHECK(dlv_malloc(ctx, nvox * 4, &lab_dev));
    CHECK(dlv_copy_h2d(ctx, vol_dev, vol, nvox * 2));
    CHECK(dlv_memset_dev(ctx, acc_dev, 0, nvox * 4));

    p.Zp = Z; p.Yp = Y; p.Xp = X;
    p.roi[0] = p.roi[1] = p.roi[2] = roi;
    p.overlap = 0.5f;
    p.flip_dim = -1;
    p.skip_threshold = 0;
    p.precision = DLV_PREC_F16;
    p.sw_batch = 0;
    p.win_begin = p.win_end = 0;
    p.z0 = p.nz = 0;
    p.repeat = 1;
    p.blend_mode = DLV_BLEND_CONSTANT;
    p.sigma_scale = 0.f;
    p.wsum_dev = NULL;
    CHECK(dlv_sw_infer_dev(ctx, &p, (const uint16_t*)vol_dev, (float*)acc_dev, NULL, &st));
    CHECK(dlv_finalize_dev(ctx, (const float*)acc_dev, NULL, (const uint16_t*)vol_dev, Y, X, Z, Y, X, 0.5f, 3, 0,
                           (uint8_t*)mask_dev, NULL));
    CHECK(dlv_ccl26_dev(ctx, (const uint8_t*)mask_dev, Z, Y, X, (uint32_t*)lab_dev, &ncomp));
    CHECK(dlv_copy_d2h(ctx, mask, mask_dev, nvox));
    for (i = 0; i < (int)nvox; ++i) fg += mask[i] != 0;
    printf("windows %lld (skipped %lld), mask voxels %zu of %zu, components %llu\n", (long long)st.n_windows,
           (long long)st.n_skipped, fg, nvox, (unsigned long long)ncomp);
    CHECK(dlv_free(ctx, vol_dev));
    CHECK(dlv_free(ctx, acc_dev));
    CHECK(dlv_free(ctx, mask_dev));
    CHECK(dlv_free(ctx, lab_dev));
    dlv_ctx_destroy(ctx);
    return st.n_skipped > 0 ? 0 : 1;
}
