/* A host in plain C driving the whole path through the C ABI of libdelivr_hip.so - no Python, no PyTorch:
 * weights -> HBM, uint16 volume -> HBM, one sliding-window pass, threshold + eroded re-mask, CCL-26 + statistics.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -o c_host -Ldelivr_cfos_amd/lib -ldelivr_hip \
 *       -Wl,-rpath,$PWD/delivr_cfos_amd/lib -Wl,--allow-shlib-undefined -lm
 *   ./c_host            # needs an MI355X; prints the number of mask voxels and components
 *   ./c_host --gpus N [--same-device] [--comm]   # the same pass sharded over N devices (dlv_comm_init_all, ONE weight broadcast,
 *                       # per-rank Z-slabs, one seam exchange); --same-device puts every rank on device 0 (a one-GPU box)
 *   ./c_host --plan N   # prints the shard plan only (host logic, no GPU needed)
 *   ./c_host --hot-block B [...]  # conv block B's weights times 2^16: its raw output leaves fp16's range, the pass returns
 *                       # DLV_ERANGE and the host repeats it after dlv_range_recover / dlv_comm_range_recover (the block's output
 *                       # stored 2^-k times smaller - InstanceNorm removes the factor), bf16 only as the last resort
 *
 * Weights: a deterministic LCG stands in for a checkpoint (same topology as MONAI BasicUNet(3,1,1,
 * (32,32,64,128,256,32), act=mish, norm=instance-affine); a real host passes the arrays of its state_dict). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

static void set_params(dlv_sw_params* p, int Z, int Y, int X, int roi) {
    memset(p, 0, sizeof(*p));
    p->Zp = Z; p->Yp = Y; p->Xp = X;
    p->roi[0] = p->roi[1] = p->roi[2] = roi;
    p->overlap = 0.5f;
    p->flip_dim = -1;
    p->skip_threshold = 0;
    p->precision = DLV_PREC_F16;
    p->repeat = 1;
    p->blend_mode = DLV_BLEND_CONSTANT;
}

/* the pass of main() sharded over n devices: every rank holds only its Z-slab of the volume and of the accumulator */
static int run_sharded(int n, int same_device, const dlv_unet_weights* w, const uint16_t* vol, int Z, int Y, int X, int roi,
                       int erode, uint8_t* mask) {
    dlv_comm* comm = NULL;
    dlv_ctx* ctx = NULL; /* for CHECK's message */
    dlv_shard_plan plan;
    dlv_sw_params p;
    dlv_sw_stats st[DLV_MAX_RANKS];
    int devs[DLV_MAX_RANKS], z0[DLV_MAX_RANKS], nz[DLV_MAX_RANKS], r;
    const uint16_t* vslab[DLV_MAX_RANKS];
    float* aslab[DLV_MAX_RANKS];
    const size_t plane = (size_t)Y * X;
    long long nw = 0, nsk = 0;
    int attempt, recoveries = 0;
    for (r = 0; r < n; ++r) devs[r] = same_device ? 0 : r;
    if (dlv_comm_init_all(n, devs, &comm) != DLV_OK) {
        fprintf(stderr, "dlv_comm_init_all(%d) failed: needs %d MI355X (or --same-device) and librccl.so: %s\n", n, n,
                dlv_comm_last_error(NULL));
        return 2;
    }
#define CCHECK(call)                                                                              \
    do {                                                                                          \
        int rc_ = (call);                                                                         \
        if (rc_ != DLV_OK) {                                                                      \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dlv_comm_last_error(comm));             \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)
    ctx = dlv_comm_ctx(comm, 0);
    CHECK(dlv_unet_load(ctx, w));      /* rank 0 reads the checkpoint ... */
    CCHECK(dlv_bcast_weights(comm, 0)); /* ... the others receive the packed blob with ONE broadcast */
    set_params(&p, Z, Y, X, roi);
    CCHECK(dlv_shard_plan_make(&p, n, NULL, &plan));
    for (r = 0; r < n; ++r) {
        void *v = NULL, *a = NULL;
        ctx = dlv_comm_ctx(comm, r);
        CCHECK(dlv_shard_slab(&plan, r, Z, erode, 0, &z0[r], &nz[r]));
        CHECK(dlv_malloc(ctx, (size_t)(nz[r] > 0 ? nz[r] : 1) * plane * 2, &v));
        CHECK(dlv_malloc(ctx, (size_t)(nz[r] > 0 ? nz[r] : 1) * plane * 4, &a));
        if (nz[r] > 0) {
            CHECK(dlv_copy_h2d(ctx, v, vol + (size_t)z0[r] * plane, (size_t)nz[r] * plane * 2)); /* only ITS planes */
            CHECK(dlv_memset_dev(ctx, a, 0, (size_t)nz[r] * plane * 4));
            CHECK(dlv_sync(ctx));
        }
        vslab[r] = (const uint16_t*)v;
        aslab[r] = (float*)a;
        printf("rank %d: windows [%lld,%lld), holds planes [%d,%d), owns [%d,%d)\n", r, (long long)plan.win_begin[r],
               (long long)plan.win_end[r], z0[r], z0[r] + nz[r], plan.z_own_lo[r], plan.z_own_hi[r]);
    }
    for (attempt = 0;; ++attempt) { /* the fp16 range guard: DLV_ERANGE -> next block shifts on every rank -> repeat */
        int changed = 0;
        const int rc = dlv_sw_infer_sharded(comm, &p, &plan, z0, nz, vslab, aslab, NULL, st);
        if (rc != DLV_ERANGE || p.precision != DLV_PREC_F16) {
            CCHECK(rc);
            break;
        }
        fprintf(stderr, "range guard: %s\n", dlv_comm_last_error(comm));
        if (attempt == 4 || dlv_comm_range_recover(comm, &changed) != DLV_OK) p.precision = DLV_PREC_BF16_ALL;
        else ++recoveries;
        for (r = 0; r < n; ++r)
            if (nz[r] > 0) {
                ctx = dlv_comm_ctx(comm, r);
                CHECK(dlv_memset_dev(ctx, aslab[r], 0, (size_t)nz[r] * plane * 4));
                CHECK(dlv_sync(ctx));
            }
    }
    printf("range recoveries %d, precision %s\n", recoveries, p.precision == DLV_PREC_F16 ? "fp16" : "bf16");
    for (r = 0; r < n; ++r) {
        const int olo = plan.z_own_lo[r], ohi = plan.z_own_hi[r] < Z ? plan.z_own_hi[r] : Z;
        void* m = NULL;
        nw += st[r].n_windows;
        nsk += st[r].n_skipped;
        ctx = dlv_comm_ctx(comm, r);
        if (ohi > olo) { /* threshold + eroded re-mask of the slab; the owned planes go to the host */
            CHECK(dlv_malloc(ctx, (size_t)nz[r] * plane, &m));
            CHECK(dlv_finalize_slab_dev(ctx, aslab[r], NULL, vslab[r], Y, X, z0[r], nz[r], Y, X, 0.5f, erode, 0, (uint8_t*)m, NULL));
            CHECK(dlv_copy_d2h(ctx, mask + (size_t)olo * plane, (const uint8_t*)m + (size_t)(olo - z0[r]) * plane,
                               (size_t)(ohi - olo) * plane));
            CHECK(dlv_free(ctx, m));
        }
        CHECK(dlv_free(ctx, (void*)vslab[r]));
        CHECK(dlv_free(ctx, aslab[r]));
    }
    printf("sharded over %d ranks: windows %lld (skipped %lld)\n", n, nw, nsk);
    dlv_comm_destroy(comm);
    return nsk > 0 ? 0 : 1;
}

int main(int argc, char** argv) {
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
    int gpus = 1, same_device = 0, plan_only = 0, use_comm = 0, hot_block = -1, attempt, recoveries = 0;
    const char* csv_path = NULL;

    for (i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--plan") && i + 1 < argc) { gpus = atoi(argv[++i]); plan_only = 1; }
        else if (!strcmp(argv[i], "--same-device")) same_device = 1;
        else if (!strcmp(argv[i], "--hot-block") && i + 1 < argc) hot_block = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--comm")) use_comm = 1; /* the communicator path also for ONE rank (with DLV_FORCE_RCCL=1: real RCCL) */
        else if (!strcmp(argv[i], "--csv") && i + 1 < argc) csv_path = argv[++i]; /* the cell table as count_blobs.py:98-114 writes it */
    }
    if (gpus < 1 || gpus > DLV_MAX_RANKS) return 3;
    if (plan_only) { /* pure host logic: no device is touched */
        dlv_shard_plan plan;
        set_params(&p, Z, Y, X, roi);
        if (dlv_shard_plan_make(&p, gpus, NULL, &plan) != DLV_OK) return 2;
        for (i = 0; i < gpus; ++i) {
            int z0 = 0, nz = 0;
            dlv_shard_slab(&plan, i, Z, 3, 0, &z0, &nz);
            printf("rank %d: windows [%lld,%lld) computes [%d,%d) owns [%d,%d) holds [%d,%d)\n", i, (long long)plan.win_begin[i],
                   (long long)plan.win_end[i], plan.z_comp_lo[i], plan.z_comp_hi[i], plan.z_own_lo[i], plan.z_own_hi[i], z0, z0 + nz);
        }
        return 0;
    }
    if (gpus == 1 && !use_comm && dlv_ctx_create(0, NULL, &ctx) != DLV_OK) {
        fprintf(stderr, "dlv_ctx_create failed: this program needs an MI355X (there is no CPU fallback)\n");
        return 2;
    }
    for (k = 0; k < 6; ++k) w.features[k] = f[k];
    for (i = 0; i < DLV_N_CONV; ++i) {
        const size_t nw = (size_t)cout[i] * cin[i] * 27;
        float* cw = filled(nw, 1.0f / (float)(cin[i] * 27 > 27 ? 40 : 6), 0.f);
        float* cb = filled((size_t)cout[i], 0.05f, 0.f);
        w.conv_w[i] = cw;
        w.conv_b[i] = cb;
        w.norm_g[i] = filled((size_t)cout[i], 0.2f, 1.0f);
        w.norm_b[i] = filled((size_t)cout[i], 0.2f, 0.f);
        if (i == hot_block) { /* the same block times 2^16 (exact): same network after InstanceNorm, raw output beyond 65504 */
            size_t j;
            for (j = 0; j < nw; ++j) cw[j] *= 65536.0f;
            for (j = 0; j < (size_t)cout[i]; ++j) cb[j] *= 65536.0f;
        }
    }
    for (i = 0; i < DLV_N_DECONV; ++i) {
        w.deconv_w[i] = filled((size_t)dcin[i] * dcout[i] * 8, 0.05f, 0.f);
        w.deconv_b[i] = filled((size_t)dcout[i], 0.05f, 0.f);
    }
    w.final_w = filled(32, 0.3f, 0.f);
    w.final_b = filled(1, 0.1f, 0.f);
    /* a synthetic volume: tissue everywhere but a background margin in x (those windows are skipped) */
    for (i = 0; i < (int)nvox; ++i) vol[i] = (i % X) < 30 ? (uint16_t)(2000 + 1500 * lcg_uniform()) : 0;
    if (gpus > 1 || use_comm) {
        const int rc = run_sharded(gpus, same_device, &w, vol, Z, Y, X, roi, 3, mask);
        for (i = 0; i < (int)nvox; ++i) fg += mask[i] != 0;
        printf("mask voxels %zu of %zu\n", fg, nvox);
        return rc;
    }
    CHECK(dlv_unet_load(ctx, &w));

    CHECK(dlv_malloc(ctx, nvox * 2, &vol_dev));
    CHECK(dlv_malloc(ctx, nvox * 4, &acc_dev));
    CHECK(dlv_malloc(ctx, nvox, &mask_dev));
    CHECK(dlv_malloc(ctx, nvox * 4, &lab_dev));
    CHECK(dlv_copy_h2d(ctx, vol_dev, vol, nvox * 2));
    CHECK(dlv_memset_dev(ctx, acc_dev, 0, nvox * 4));

    set_params(&p, Z, Y, X, roi);
    CHECK(dlv_reserve_dev(ctx, &p, Z, Y, X)); /* (optional) the pass's workspaces and the finalize scratch now, not inside the pass */
    for (attempt = 0;; ++attempt) { /* the fp16 range guard: DLV_ERANGE -> next block shifts -> repeat */
        int changed = 0;
        const int rc = dlv_sw_infer_dev(ctx, &p, (const uint16_t*)vol_dev, (float*)acc_dev, NULL, &st);
        if (rc != DLV_ERANGE || p.precision != DLV_PREC_F16) {
            CHECK(rc);
            break;
        }
        fprintf(stderr, "range guard: %s\n", dlv_last_error(ctx));
        if (attempt == 4 || dlv_range_recover(ctx, &changed) != DLV_OK) p.precision = DLV_PREC_BF16_ALL;
        else ++recoveries;
        CHECK(dlv_memset_dev(ctx, acc_dev, 0, nvox * 4));
    }
    printf("range recoveries %d, precision %s\n", recoveries, p.precision == DLV_PREC_F16 ? "fp16" : "bf16");
    CHECK(dlv_finalize_dev(ctx, (const float*)acc_dev, NULL, (const uint16_t*)vol_dev, Y, X, Z, Y, X, 0.5f, 3, 0,
                           (uint8_t*)mask_dev, NULL));
    CHECK(dlv_ccl26_dev(ctx, (const uint8_t*)mask_dev, Z, Y, X, (uint32_t*)lab_dev, &ncomp));
    if (csv_path) { /* cc3d.statistics + the cell table (count_blobs.py:85, :98-114): statistics on the device, text by the library */
        uint32_t* counts = (uint32_t*)calloc((size_t)ncomp + 1, sizeof(uint32_t));
        uint16_t* boxes = (uint16_t*)calloc(((size_t)ncomp + 1) * 6, sizeof(uint16_t));
        double* cents = (double*)calloc(((size_t)ncomp + 1) * 3, sizeof(double));
        const size_t cap = 64 + 128 * ((size_t)ncomp + 1);
        char* text = (char*)malloc(cap);
        size_t len = 0;
        FILE* fh;
        if (!counts || !boxes || !cents || !text) return 1;
        CHECK(dlv_cc_stats_dev(ctx, (const uint32_t*)lab_dev, Z, Y, X, ncomp, counts, boxes, cents));
        if (dlv_cells_csv(counts, cents, ncomp, text, cap, &len) != DLV_OK) return 1;
        fh = fopen(csv_path, "wb");
        if (!fh || fwrite(text, 1, len, fh) != len || fclose(fh) != 0) return 1;
        printf("cell table: %llu rows, %zu bytes -> %s\n", (unsigned long long)(ncomp ? ncomp - 1 : 0), len, csv_path);
        free(counts); free(boxes); free(cents); free(text);
    }
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
