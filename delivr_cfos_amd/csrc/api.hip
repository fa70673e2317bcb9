// api.hip - context, memory helpers, weight loading and the in-library kernel timer of
// libdelivr_hip.so (C ABI declared in include/delivr_hip.h).
#include <dlfcn.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "common.h"

int dlv_fail(dlv_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int dlv_ws_get(dlv_ctx* ctx, int slot, size_t bytes, void** out) {
    if (slot < 0 || slot >= WS_N_SLOTS) return dlv_fail(ctx, DLV_EINVAL, "bad scratch slot %d", slot);
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            DLV_TRY(dlv_sync_all(ctx));
            DLV_HIP(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        // 12.5 % headroom against regrowth - for the small buffers only: the big ones (activations of a lane, CCL scratch) are sized
        // by the geometry, and a regrowth (hipFree + hipMalloc) is what makes later allocations slow (profiles/r06r_alloc_probe2.json)
        size_t want = bytes + (bytes < ((size_t)256 << 20) ? (bytes >> 3) : 0);
        hipError_t e = hipMalloc(&ctx->ws[slot], want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&ctx->ws[slot], want);
        }
        if (e != hipSuccess)
            return dlv_fail(ctx, DLV_ENOMEM, "scratch slot %d: hipMalloc(%zu) failed: %s", slot, bytes,
                            hipGetErrorString(e));
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return DLV_OK;
}

int dlv_sync_all(dlv_ctx* ctx) {
    DLV_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
    for (int k = 0; k < DLV_MAX_LANES - 1; ++k)
        if (ctx->aux[k]) DLV_HIP(ctx, hipStreamSynchronize(ctx->aux[k]));
    return DLV_OK;
}

// ---- kernel timer ------------------------------------------------------------------------------
// DLV_LAUNCH_LOG=<file>: one line per bracketed launch (label, algorithmic flops and bytes), in launch order.  The PMC
// passes (profiles/run_pmc_traffic.sh) read it next to rocprofv3's dispatch table to attribute the counters of a
// kernel template that serves several layers to the layer's label.
static FILE* launch_log() {
    static FILE* f = [] {
        const char* path = getenv("DLV_LAUNCH_LOG");
        return path && *path ? fopen(path, "w") : (FILE*)nullptr;
    }();
    return f;
}

// roctx (rocprofiler-sdk): ranges named like the kernel labels of dlv_prof_report.  The symbols are taken from a roctx library
// that is already in the process (a profiler's); DLV_ROCTX=1 loads one.  Without either nothing is called.
typedef int (*dlv_roctx_push_t)(const char*);
typedef int (*dlv_roctx_pop_t)(void);
static dlv_roctx_push_t g_roctx_push = nullptr;
static dlv_roctx_pop_t g_roctx_pop = nullptr;
static void roctx_init() {
    static std::once_flag once;
    std::call_once(once, [] {
        void* push = dlsym(RTLD_DEFAULT, "roctxRangePushA");
        void* pop = dlsym(RTLD_DEFAULT, "roctxRangePop");
        if ((!push || !pop) && getenv("DLV_ROCTX")) {
            void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                push = dlsym(h, "roctxRangePushA");
                pop = dlsym(h, "roctxRangePop");
            }
        }
        if (push && pop) {
            g_roctx_push = (dlv_roctx_push_t)push;
            g_roctx_pop = (dlv_roctx_pop_t)pop;
        }
    });
}

DlvProf::~DlvProf() {
    if (ranged) (void)g_roctx_pop();
}

DlvProf::DlvProf(dlv_ctx* c, const char* name, double flops, double bytes) : ctx(c) {
    roctx_init();
    if (g_roctx_push) {
        (void)g_roctx_push(name);
        ranged = true;
    }
    if (FILE* f = launch_log()) {
        fprintf(f, "%s\t%.0f\t%.0f\n", name, flops, bytes);
        fflush(f);
    }
    if (!c->prof_on) return;
    int slot = -1;
    for (size_t i = 0; i < c->prof_slots.size(); ++i)
        if (strncmp(c->prof_slots[i].name, name, sizeof(c->prof_slots[i].name)) == 0) slot = (int)i;
    if (slot < 0) {
        if (c->prof_slots.size() >= DLV_PROF_MAX_KERNELS) return;
        DlvProfSlot s;
        memset(s.name, 0, sizeof(s.name));
        strncpy(s.name, name, sizeof(s.name) - 1);
        c->prof_slots.push_back(s);
        slot = (int)c->prof_slots.size() - 1;
    }
    c->prof_slots[slot].flops += flops;
    c->prof_slots[slot].bytes += bytes;
    DlvProfPending p;
    p.slot = slot;
    for (hipEvent_t* e : {&p.a, &p.b}) {
        if (!c->prof_free.empty()) {
            *e = c->prof_free.back();
            c->prof_free.pop_back();
        } else if (hipEventCreate(e) != hipSuccess) {
            return;
        }
    }
    (void)hipEventRecord(p.a, c->stream);
    c->prof_pending.push_back(p);
    idx = (int)c->prof_pending.size() - 1;
}
void DlvProf::end() {
    if (idx >= 0) (void)hipEventRecord(ctx->prof_pending[idx].b, ctx->stream);
    if (ranged) {
        (void)g_roctx_pop();
        ranged = false;
    }
}

static int prof_drain(dlv_ctx* ctx) {
    if (ctx->prof_pending.empty()) return DLV_OK;
    DLV_TRY(dlv_sync_all(ctx));
    for (auto& p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof_slots[p.slot].total_ms += ms;
            ctx->prof_slots[p.slot].launches += 1;
        }
        ctx->prof_free.push_back(p.a);
        ctx->prof_free.push_back(p.b);
    }
    ctx->prof_pending.clear();
    return DLV_OK;
}

// ---- weight blob layout --------------------------------------------------------------------------
static const int kConvLevel[DLV_N_CONV] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0};

static void conv_channels(const int f[6], int cin[DLV_N_CONV], int cout[DLV_N_CONV], int dcin[DLV_N_DECONV],
                          int dcout[DLV_N_DECONV]) {
    // MONAI BasicUNet topology (features f[0..5]); upcat_1 keeps its channels (halves=False)
    int ci[DLV_N_CONV] = {1, f[0], f[0], f[1], f[1], f[2], f[2], f[3], f[3], f[4],
                          f[3] + f[4] / 2, f[3], f[2] + f[3] / 2, f[2], f[1] + f[2] / 2, f[1], f[0] + f[1], f[5]};
    int co[DLV_N_CONV] = {f[0], f[0], f[1], f[1], f[2], f[2], f[3], f[3], f[4], f[4],
                          f[3], f[3], f[2], f[2], f[1], f[1], f[5], f[5]};
    for (int i = 0; i < DLV_N_CONV; ++i) {
        cin[i] = ci[i];
        cout[i] = co[i];
    }
    int di[DLV_N_DECONV] = {f[4], f[3], f[2], f[1]};
    int dco[DLV_N_DECONV] = {f[4] / 2, f[3] / 2, f[2] / 2, f[1]};
    for (int j = 0; j < DLV_N_DECONV; ++j) {
        dcin[j] = di[j];
        dcout[j] = dco[j];
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// walks the blob layout; if base != nullptr assigns the pointers into ctx
static size_t layout_blob(dlv_ctx* ctx, const int f[6], char* base) {
    int cin[DLV_N_CONV], cout[DLV_N_CONV], dcin[DLV_N_DECONV], dcout[DLV_N_DECONV];
    conv_channels(f, cin, cout, dcin, dcout);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off = align256(off + bytes);
        return p;
    };
    for (int i = 0; i < DLV_N_CONV; ++i) {
        size_t nw = (size_t)cout[i] * cin[i] * 27;
        float* w = (float*)take(nw * 4);
        float* b = (float*)take((size_t)cout[i] * 4);
        float* b16 = (float*)take((size_t)cout[i] * 4);
        float* g = (float*)take((size_t)cout[i] * 4);
        float* be = (float*)take((size_t)cout[i] * 4);
        // the stem (i == 0) keeps a 4-k-step hi/lo A-fragment pack for the MFMA stem (4 KiB)
        uint16_t* wb = (uint16_t*)take(i == 0 ? (size_t)4096 : nw * 2);
        uint16_t* wh = (uint16_t*)take(i == 0 ? (size_t)4096 : nw * 2);
        uint16_t* wb16 = i == 0 ? nullptr : (uint16_t*)take(nw * 2);
        uint16_t* wh16 = i == 0 ? nullptr : (uint16_t*)take(nw * 2);
        // conv 16 = upcat_1.conv_0 when its inputs are a 32-channel skip + the 32 channels of a 32->32 transposed conv
        const int slices = (i == 16 && cin[i] == 64 && cout[i] == 32 && dcin[3] == 32 && dcout[3] == 32) ? 1 : 0;
        const bool fold = slices > 0;
        uint16_t* wsb = fold ? (uint16_t*)take((size_t)32 * 32 * 27 * 2) : nullptr;
        uint16_t* wsh = fold ? (uint16_t*)take((size_t)32 * 32 * 27 * 2) : nullptr;
        uint16_t* wub = fold ? (uint16_t*)take((size_t)slices * 2 * 2 * 4 * 8 * 64 * 8 * 2) : nullptr;
        uint16_t* wuh = fold ? (uint16_t*)take((size_t)slices * 2 * 2 * 4 * 8 * 64 * 8 * 2) : nullptr;
        float* ucr = fold ? (float*)take((size_t)slices * 8 * 8 * 32 * 4) : nullptr;
        if (base) {
            ctx->conv[i].up_slices = slices;
            ctx->conv[i].wskip_bf16 = wsb;
            ctx->conv[i].wskip_f16 = wsh;
            ctx->conv[i].wup_bf16 = wub;
            ctx->conv[i].wup_f16 = wuh;
            ctx->conv[i].up_corr = ucr;
            ctx->conv[i].w16_bf16 = wb16;
            ctx->conv[i].w16_f16 = wh16;
            ctx->conv[i].w_f16 = wh;
            ctx->conv[i].cin = cin[i];
            ctx->conv[i].cout = cout[i];
            ctx->conv[i].w_f32 = w;
            ctx->conv[i].bias = b;
            ctx->conv[i].bias16 = b16;
            ctx->conv[i].gamma = g;
            ctx->conv[i].beta = be;
            ctx->conv[i].w_bf16 = wb;
        }
    }
    for (int j = 0; j < DLV_N_DECONV; ++j) {
        size_t nw = (size_t)dcin[j] * dcout[j] * 8;
        float* w = (float*)take(nw * 4);
        float* b = (float*)take((size_t)dcout[j] * 4);
        uint16_t* wb = (uint16_t*)take(nw * 2);
        uint16_t* wh = (uint16_t*)take(nw * 2);
        uint16_t* wb16 = dcin[j] >= 128 ? (uint16_t*)take(nw * 2) : nullptr;
        uint16_t* wh16 = dcin[j] >= 128 ? (uint16_t*)take(nw * 2) : nullptr;
        if (base) {
            ctx->deconv[j].w16_bf16 = wb16;
            ctx->deconv[j].w16_f16 = wh16;
            ctx->deconv[j].w_f16 = wh;
            ctx->deconv[j].cin = dcin[j];
            ctx->deconv[j].cout = dcout[j];
            ctx->deconv[j].w_f32 = w;
            ctx->deconv[j].bias = b;
            ctx->deconv[j].w_bf16 = wb;
        }
    }
    float* fw = (float*)take((size_t)f[5] * 4);
    float* fb = (float*)take(4);
    if (base) {
        ctx->final_w = fw;
        ctx->final_b = fb;
    }
    return off;
}

static int check_features(dlv_ctx* ctx, const int f[6]) {
    for (int i = 0; i < 6; ++i)
        if (f[i] <= 0 || f[i] % 32 != 0 || f[i] > 256)
            return dlv_fail(ctx, DLV_EUNSUP, "features[%d]=%d: channel counts must be multiples of 32, <= 256", i,
                            f[i]);
    if (f[2] % 2 || f[3] % 2 || f[4] % 2) return dlv_fail(ctx, DLV_EUNSUP, "features 2..4 must be even");
    if ((f[2] / 2) % 32 || (f[3] / 2) % 32 || (f[4] / 2) % 32)
        return dlv_fail(ctx, DLV_EUNSUP, "halved up-sampling channels must be multiples of 32");
    return DLV_OK;
}

static int alloc_blob(dlv_ctx* ctx, const int f[6]) {
    DLV_TRY(check_features(ctx, f));
    size_t bytes = layout_blob(ctx, f, nullptr);
    if (ctx->blob) {
        DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        DLV_HIP(ctx, hipFree(ctx->blob));
        ctx->blob = nullptr;
    }
    DLV_HIP(ctx, hipMalloc(&ctx->blob, bytes));
    DLV_HIP(ctx, hipMemsetAsync(ctx->blob, 0, bytes, ctx->stream));
    ctx->blob_bytes = bytes;
    memcpy(ctx->features, f, sizeof(int) * 6);
    layout_blob(ctx, f, (char*)ctx->blob);
    for (int i = 0; i < DLV_N_CONV; ++i) ctx->conv[i].shift = 0;  // a new checkpoint starts unshifted
    return DLV_OK;
}

extern "C" {

int dlv_abi_version(void) { return DLV_ABI_VERSION; }

int dlv_ctx_create(int device_id, void* stream, dlv_ctx** out) {
    if (!out) return DLV_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return DLV_EHIP;
    dlv_ctx* ctx = new (std::nothrow) dlv_ctx();
    if (!ctx) return DLV_ENOMEM;
    ctx->device = device_id;
#ifdef DLV_DIAG  // the product library takes no kernel variant from the environment (diagnostic builds: make diag)
    if (const char* e = getenv("DLV_ZM_VARIANT")) ctx->zm_variant = atoi(e);
#endif
#ifdef DLV_DIAG  // timing-only ablations (WRONG results): the diagnostic library only (make diag), never the product
    ctx->upconv_dbg = getenv("DLV_UPCONV_DBG") ? atoi(getenv("DLV_UPCONV_DBG")) : 0;
#endif
    if (hipSetDevice(device_id) != hipSuccess) {
        delete ctx;
        return DLV_EHIP;
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return DLV_EHIP;
        }
        ctx->own_stream = true;
    }
    ctx->main_stream = ctx->stream;
    if (hipMalloc(&ctx->zero_page, 256) == hipSuccess) (void)hipMemset(ctx->zero_page, 0, 256);
    if (hipMalloc((void**)&ctx->range_flag, 256) != hipSuccess || hipMemset(ctx->range_flag, 0, 256) != hipSuccess) {
        (void)dlv_ctx_destroy(ctx);
        return DLV_ENOMEM;
    }
    {
        bool ok = true;
        for (int k = 0; k < DLV_MAX_LANES - 1 && ok; ++k) ok = hipStreamCreateWithFlags(&ctx->aux[k], hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < DLV_MAX_LANES + 1 && ok; ++k) ok = hipEventCreateWithFlags(&ctx->ev_lane[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (int k = 0; k < DLV_MAX_LANES - 1; ++k)
                if (ctx->aux[k]) {
                    (void)hipStreamDestroy(ctx->aux[k]);
                    ctx->aux[k] = nullptr;
                }
        }
        ctx->aux_stream = ctx->aux[0];
    }
    if (const char* e = getenv("DLV_LANES")) ctx->lanes_wanted = std::max(1, std::min(DLV_MAX_LANES, atoi(e)));
    *out = ctx;
    return DLV_OK;
}

int dlv_ctx_destroy(dlv_ctx* ctx) {
    if (!ctx) return DLV_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)dlv_sync_all(ctx);
    for (auto& p : ctx->prof_pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : ctx->prof_free) (void)hipEventDestroy(e);
    for (int i = 0; i < WS_N_SLOTS; ++i)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    if (ctx->blob) (void)hipFree(ctx->blob);
    if (ctx->zero_page) (void)hipFree(ctx->zero_page);
    if (ctx->range_flag) (void)hipFree(ctx->range_flag);
    for (int k = 0; k < DLV_MAX_LANES - 1; ++k)
        if (ctx->aux[k]) {
            (void)hipStreamSynchronize(ctx->aux[k]);
            (void)hipStreamDestroy(ctx->aux[k]);
        }
    for (int k = 0; k < DLV_MAX_LANES + 1; ++k)
        if (ctx->ev_lane[k]) (void)hipEventDestroy(ctx->ev_lane[k]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->main_stream);
    delete ctx;
    return DLV_OK;
}

const char* dlv_last_error(dlv_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int dlv_sync(dlv_ctx* ctx) {
    if (!ctx) return DLV_EINVAL;
    DLV_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
    return DLV_OK;
}

void* dlv_stream(dlv_ctx* ctx) { return ctx ? (void*)ctx->main_stream : nullptr; }

int dlv_malloc(dlv_ctx* ctx, size_t bytes, void** out_dev) {
    if (!ctx || !out_dev) return DLV_EINVAL;
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DLV_HIP(ctx, hipMalloc(out_dev, bytes));
    return DLV_OK;
}
int dlv_free(dlv_ctx* ctx, void* p_dev) {
    if (!ctx) return DLV_EINVAL;
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DLV_HIP(ctx, hipFree(p_dev));
    return DLV_OK;
}
int dlv_memset_dev(dlv_ctx* ctx, void* p_dev, int value, size_t bytes) {
    if (!ctx) return DLV_EINVAL;
    DLV_HIP(ctx, hipMemsetAsync(p_dev, value, bytes, ctx->stream));
    return DLV_OK;
}
int dlv_copy_h2d(dlv_ctx* ctx, void* dst_dev, const void* src, size_t bytes) {
    if (!ctx) return DLV_EINVAL;
    DLV_HIP(ctx, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DLV_OK;
}
int dlv_copy_d2h(dlv_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
    if (!ctx) return DLV_EINVAL;
    DLV_HIP(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DLV_OK;
}

int dlv_unet_alloc_blob(dlv_ctx* ctx, const int features[6]) {
    if (!ctx || !features) return DLV_EINVAL;
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DLV_TRY(alloc_blob(ctx, features));
    ctx->weights_loaded = true;  // content arrives by broadcast into dlv_unet_blob_dev()
    return DLV_OK;
}

int dlv_unet_load(dlv_ctx* ctx, const dlv_unet_weights* w) {
    if (!ctx || !w) return DLV_EINVAL;
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < DLV_N_CONV; ++i)
        if (!w->conv_w[i] || !w->conv_b[i] || !w->norm_g[i] || !w->norm_b[i])
            return dlv_fail(ctx, DLV_EINVAL, "conv layer %d has a null parameter pointer", i);
    for (int j = 0; j < DLV_N_DECONV; ++j)
        if (!w->deconv_w[j] || !w->deconv_b[j]) return dlv_fail(ctx, DLV_EINVAL, "deconv %d has a null pointer", j);
    if (!w->final_w || !w->final_b) return dlv_fail(ctx, DLV_EINVAL, "final conv has a null pointer");
    DLV_TRY(alloc_blob(ctx, w->features));
    auto up = [&](void* dst, const void* src, size_t bytes) -> int {
        DLV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return DLV_OK;
    };
    for (int i = 0; i < DLV_N_CONV; ++i) {
        DlvConvLayer& L = ctx->conv[i];
        DLV_TRY(up(L.w_f32, w->conv_w[i], (size_t)L.cout * L.cin * 27 * 4));
        DLV_TRY(up(L.bias, w->conv_b[i], (size_t)L.cout * 4));
        DLV_TRY(up(L.gamma, w->norm_g[i], (size_t)L.cout * 4));
        DLV_TRY(up(L.beta, w->norm_b[i], (size_t)L.cout * 4));
    }
    for (int j = 0; j < DLV_N_DECONV; ++j) {
        DlvDeconvLayer& L = ctx->deconv[j];
        DLV_TRY(up(L.w_f32, w->deconv_w[j], (size_t)L.cin * L.cout * 8 * 4));
        DLV_TRY(up(L.bias, w->deconv_b[j], (size_t)L.cout * 4));
    }
    DLV_TRY(up(ctx->final_w, w->final_w, (size_t)ctx->features[5] * 4));
    DLV_TRY(up(ctx->final_b, w->final_b, 4));
    DLV_TRY(dlv_pack_weights_bf16(ctx));
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->weights_loaded = true;
    return DLV_OK;
}

int dlv_unet_set_conv_shift(dlv_ctx* ctx, int layer, int shift) {
    if (!ctx) return DLV_EINVAL;
    if (layer < 0 || layer >= DLV_N_CONV || shift < 0 || shift > 40) return dlv_fail(ctx, DLV_EINVAL, "conv shift: layer 0..17, shift 0..40");
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "dlv_unet_set_conv_shift before dlv_unet_load");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DLV_TRY(dlv_sync_all(ctx));
    ctx->conv[layer].shift = shift;
    DLV_TRY(dlv_pack_weights_bf16(ctx));  // (both 16-bit formats, every pack of every layer: milliseconds)
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DLV_OK;
}
int dlv_unet_get_conv_shift(dlv_ctx* ctx, int layer, int* shift) {
    if (!ctx || !shift || layer < 0 || layer >= DLV_N_CONV) return DLV_EINVAL;
    *shift = ctx->conv[layer].shift;
    return DLV_OK;
}
int dlv_unet_note_conv_shifts(dlv_ctx* ctx, const int* shifts) {
    if (!ctx || !shifts) return DLV_EINVAL;
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "dlv_unet_note_conv_shifts before the blob was allocated");
    for (int i = 0; i < DLV_N_CONV; ++i) {
        if (shifts[i] < 0 || shifts[i] > 40) return dlv_fail(ctx, DLV_EINVAL, "conv shift %d of block %d", shifts[i], i);
        ctx->conv[i].shift = shifts[i];
    }
    return DLV_OK;
}
int dlv_range_report(dlv_ctx* ctx, int* layer, float* peaks) {
    if (!ctx) return DLV_EINVAL;
    if (layer) *layer = ctx->range_last;
    if (peaks)
        for (int i = 0; i < DLV_N_CONV; ++i) peaks[i] = ctx->range_peak[i];
    return DLV_OK;
}

// The range guard's policy as host logic (what run_inference does about DLV_ERANGE: delivr_cfos_amd/range_guard.py holds the
// same rules; a CPU test compares the two).  Conv block `layer` (18 = the logits) saw a non-finite input: the blocks whose
// stored raw output reaches it - MONAI BasicUNet's wiring, inference/inference.py:190-197 - get a larger shift.
static int range_producers(int layer, int* out) {
    if (layer == 18) { out[0] = 17; return 1; }
    if (layer == 10 || layer == 12 || layer == 14 || layer == 16) {  // upcat_l.conv_0: the skip tensor of its level + the block below
        const int level = 3 - (layer - 10) / 2;
        out[0] = 2 * level + 1;
        out[1] = layer - 1;
        return 2;
    }
    if (layer >= 1 && layer < DLV_N_CONV) { out[0] = layer - 1; return 1; }
    return 0;
}
int dlv_range_next_shifts(int layer, const float* peaks, const int* shifts, int* out) {
    if (!peaks || !shifts || !out) return -1;
    for (int i = 0; i < DLV_N_CONV; ++i) out[i] = shifts[i];
    int cand[2];
    int nc = range_producers(layer, cand);
    bool hinted = false;
    for (int i = 0; i < nc; ++i) hinted |= peaks[cand[i]] > 4096.0f;
    if (layer == 16 && !hinted) {  // the folded up half P of upcat_1.conv_0 is stored before the block's statistics exist
        cand[0] = 16;
        nc = 1;
    }
    int changed = 0;
    for (int i = 0; i < nc; ++i) {
        const int p = cand[i];
        const bool hint = hinted && peaks[p] > 4096.0f;
        if (hinted && !hint) continue;
        // |mean| + 8 sigma of the stored tensor to <= 1024; without a hint 6 bits at a time - but never a block whose recorded
        // |mean| + 8 sigma would fall below 1 (sigma below 2^-3: its small values would leave fp16's normal range)
        const int step = hint ? std::max(1, (int)std::ceil(std::log2((double)peaks[p] / 1024.0))) : 6;
        if (!hint && peaks[p] > 0.0f && (double)peaks[p] * std::exp2(-(double)step) < 1.0) continue;
        const int k = std::min(40, shifts[p] + step);
        if (k != shifts[p]) {
            out[p] = k;
            ++changed;
        }
    }
    return changed;
}
}  // extern "C"
// (internal, C++ linkage: common.h)
int dlv_unet_apply_conv_shifts(dlv_ctx* ctx, const int* shifts) {
    if (!ctx || !shifts) return DLV_EINVAL;
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "conv shifts before dlv_unet_load");
    for (int i = 0; i < DLV_N_CONV; ++i)
        if (shifts[i] < 0 || shifts[i] > 40) return dlv_fail(ctx, DLV_EINVAL, "conv shift %d of block %d", shifts[i], i);
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DLV_TRY(dlv_sync_all(ctx));
    int old[DLV_N_CONV];
    for (int i = 0; i < DLV_N_CONV; ++i) {
        old[i] = ctx->conv[i].shift;
        ctx->conv[i].shift = shifts[i];
    }
    int rc = dlv_pack_weights_bf16(ctx);
    if (rc == DLV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = dlv_fail(ctx, DLV_EHIP, "repack of the 16-bit weights failed");
    if (rc != DLV_OK) {  // eps and the packs must agree: back to the shifts the packs were made with (best effort: repack those)
        for (int i = 0; i < DLV_N_CONV; ++i) ctx->conv[i].shift = old[i];
        (void)dlv_pack_weights_bf16(ctx);
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    ctx->range_last = -1;  // the report belongs to the pass before the repack: a second recover without a pass must not reuse it
    for (int i = 0; i < DLV_N_CONV; ++i) ctx->range_peak[i] = 0.f;
    return DLV_OK;
}
// the next shifts for (layer, peaks) from this context's current ones -> number of blocks that change (0: nothing left);
// *blind: no block feeding the layer reported a peak (a 6-bit step on no evidence)
int dlv_range_plan(dlv_ctx* ctx, int layer, const float* peaks, int* nxt, bool* blind) {
    int cur[DLV_N_CONV];
    for (int i = 0; i < DLV_N_CONV; ++i) cur[i] = ctx->conv[i].shift;
    int cand[2];
    const int nc = range_producers(layer, cand);
    *blind = true;
    for (int i = 0; i < nc; ++i) *blind = *blind && !(peaks[cand[i]] > 4096.0f);
    return dlv_range_next_shifts(layer, peaks, cur, nxt);
}
int dlv_range_step(dlv_ctx* ctx, int layer, const float* peaks, int* n_changed, const int* force_next) {
    int nxt[DLV_N_CONV];
    bool blind = false;
    int changed = dlv_range_plan(ctx, layer, peaks, nxt, &blind);
    // a blind step that did not move the overflow (the same layer is named again, again without a hint) is not repeated: the
    // value that leaves the range is not a stored conv output (e.g. the un-normalised transposed conv) and no shift reaches it
    if (changed > 0 && blind && ctx->range_seq && ctx->range_blind_layer == layer) changed = 0;
    if (changed <= 0) {
        if (ctx->range_seq) {  // leave the packs as the caller's next format finds them best: the shifts of before the sequence
            (void)dlv_unet_apply_conv_shifts(ctx, ctx->range_base);
            ctx->range_seq = false;
            ctx->range_blind_layer = -1;
        }
        return dlv_fail(ctx, DLV_ERANGE, "range guard: no block shift left to try for conv block %d (repeat the passes with DLV_PREC_BF16_ALL)", layer);
    }
    if (!ctx->range_seq) {
        for (int i = 0; i < DLV_N_CONV; ++i) ctx->range_base[i] = ctx->conv[i].shift;
        ctx->range_seq = true;
    }
    DLV_TRY(dlv_unet_apply_conv_shifts(ctx, force_next ? force_next : nxt));
    ctx->range_blind_layer = blind ? layer : -1;
    if (n_changed) *n_changed = changed;
    return DLV_OK;
}
extern "C" {
int dlv_range_recover(dlv_ctx* ctx, int* n_changed) {
    if (!ctx) return DLV_EINVAL;
    if (n_changed) *n_changed = 0;
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "dlv_range_recover before dlv_unet_load");
    if (ctx->range_last < 0) return dlv_fail(ctx, DLV_ESTATE, "dlv_range_recover: the last 16-bit pass did not end with DLV_ERANGE");
    float peaks[DLV_N_CONV];
    for (int i = 0; i < DLV_N_CONV; ++i) peaks[i] = ctx->range_peak[i];
    return dlv_range_step(ctx, ctx->range_last, peaks, n_changed, nullptr);
}

int dlv_unet_blob_size(dlv_ctx* ctx, size_t* bytes) {
    if (!ctx || !bytes) return DLV_EINVAL;
    if (!ctx->blob) return dlv_fail(ctx, DLV_ESTATE, "no weights loaded/allocated");
    *bytes = ctx->blob_bytes;
    return DLV_OK;
}
int dlv_unet_blob_dev(dlv_ctx* ctx, void** blob_dev) {
    if (!ctx || !blob_dev) return DLV_EINVAL;
    if (!ctx->blob) return dlv_fail(ctx, DLV_ESTATE, "no weights loaded/allocated");
    *blob_dev = ctx->blob;
    return DLV_OK;
}

int dlv_unet_forward_dev(dlv_ctx* ctx, const float* x_dev, float* logits_dev, int B, int d, int h, int w,
                         int precision) {
    if (!ctx || !x_dev || !logits_dev) return DLV_EINVAL;
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "dlv_unet_forward_dev before dlv_unet_load");
    if (B <= 0 || d <= 0 || h <= 0 || w <= 0) return dlv_fail(ctx, DLV_EINVAL, "empty batch/patch");
    if (d < 16 || h < 16 || w < 16 || (long long)(d >> 4) * (h >> 4) * (w >> 4) < 2)
        return dlv_fail(ctx, DLV_EUNSUP, "patch %dx%dx%d: every dimension must be at least 16 and level 4 (each dimension / 16, rounded down) must hold more "
                        "than one voxel - InstanceNorm3d has no statistics of a single value and torch raises there; any size from there on: "
                        "levels with an odd size are pooled and padded like MONAI's MaxPool3d / UpCat", d, h, w);
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    if (precision == DLV_PREC_F32) return dlv_unet_forward_f32(ctx, x_dev, logits_dev, B, d, h, w);
    if (precision == DLV_PREC_BF16 || precision == DLV_PREC_F16 || precision == DLV_PREC_BF16_ALL)
        return dlv_unet_forward_bf16(ctx, x_dev, logits_dev, B, d, h, w, dlv_fmt16(precision));
    return dlv_fail(ctx, DLV_EINVAL, "unknown precision %d", precision);
}

// ---- the cell table's text (count_blobs.py:98-114) -----------------------------------------------------------------------------
// one float as Python's repr writes it: shortest round-trip digits (std::to_chars without a precision gives exactly those),
// fixed notation with ".0" on integral values for 1e-4 <= |v| < 1e16, exponent form outside (float_repr_style 'short')
static char* py_float_repr(char* p, char* end, double v) {
    if (std::isnan(v)) return p + snprintf(p, (size_t)(end - p), "nan");
    if (std::isinf(v)) return p + snprintf(p, (size_t)(end - p), v < 0 ? "-inf" : "inf");
    const double a = std::fabs(v);
    if (a != 0.0 && (a < 1e-4 || a >= 1e16)) {
        const auto r = std::to_chars(p, end, v, std::chars_format::scientific);
        return r.ptr;
    }
    const auto r = std::to_chars(p, end, v, std::chars_format::fixed);
    char* q = r.ptr;
    bool dot = false;
    for (char* c = p; c < q; ++c) dot |= (*c == '.');
    if (!dot && q + 2 <= end) {
        *q++ = '.';
        *q++ = '0';
    }
    return q;
}

int dlv_cells_csv(const uint32_t* voxel_counts, const double* centroids, uint64_t n, char* out, size_t cap, size_t* len_out) {
    if (!voxel_counts || !centroids || !out || !len_out) return DLV_EINVAL;
    static const char header[] = ",Blob,Coords,Size\n";
    if (cap < sizeof(header)) return DLV_EINVAL;
    char* p = out;
    char* const end = out + cap;
    memcpy(p, header, sizeof(header) - 1);
    p += sizeof(header) - 1;
    for (uint64_t i = 1; i < n; ++i) {  // (range(1, N): the reference drops the last label)
        if ((size_t)(end - p) < 128) return DLV_EINVAL;
        p += snprintf(p, (size_t)(end - p), "0,%llu,\"[", (unsigned long long)i);
        for (int k = 0; k < 3; ++k) {
            p = py_float_repr(p, end, centroids[3 * i + k]);
            if (k < 2) {
                *p++ = ',';
                *p++ = ' ';
            }
        }
        p += snprintf(p, (size_t)(end - p), "]\",%u\n", voxel_counts[i]);
    }
    *len_out = (size_t)(p - out);
    return DLV_OK;
}

// test / A-B switches (include/delivr_hip_diag.h): kernel selection per context, never from the environment
int dlv_diag_set(dlv_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return DLV_EINVAL;
    const std::string n(name);
    if (n == "no_zmarch") ctx->no_zmarch = value != 0;
    else if (n == "no_upconv") ctx->fold_up = value ? 0 : 1;
    else if (n == "upconv_simple") ctx->upconv_simple = value ? 1 : 0;
    else if (n == "fuse_levels") ctx->fuse_levels = value;
    else if (n == "fuse_layers") ctx->fuse_layers = value;
    else if (n == "zreg_mask") ctx->zreg_mask = value;
    else if (n == "deep_mask") ctx->deep_mask = value;
    else if (n == "generic_ncb") ctx->generic_ncb = value;
    else if (n == "zreg_dbg") ctx->zreg_dbg = value;
    else if (n == "deep_small") ctx->deep_small = value;
    else if (n == "pool_rows_off") ctx->pool_rows_off = value != 0;
    else if (n == "erode_xy_split") ctx->erode_xy_split = value != 0;
    else if (n == "erode_z_two_sweeps") ctx->erode_z_two_sweeps = value != 0;
    else if (n == "ccl_simple") ctx->ccl_simple = value != 0;
    else if (n == "resample_simple") ctx->resample_simple = value != 0;
    else if (n == "resample_run16") ctx->resample_run16 = value != 0;
    else if (n == "tiff_chunk") ctx->tiff_chunk = value > 0 ? value : 0;
    else return dlv_fail(ctx, DLV_EINVAL, "dlv_diag_set: unknown switch '%s'", name);
    return DLV_OK;
}

int dlv_debug_set_zm_variant(dlv_ctx* ctx, int variant) {
    if (!ctx) return DLV_EINVAL;
#ifndef DLV_DIAG
    // product library: 0 / 50 = register-resident-weights conv (default), 51 = the LDS-resident-weights kernel for every
    // z-march layer (A/B); the timing-only, stamped and experimental builds exist in libdelivr_hip_diag.so only
    if (variant != 0 && variant != 50 && variant != 51)
        return dlv_fail(ctx, DLV_EUNSUP, "z-march variant %d is a diagnostic build: load libdelivr_hip_diag.so (make -C delivr_cfos_amd/csrc diag)", variant);
#endif
    ctx->zm_variant = variant;
    return DLV_OK;
}

int dlv_debug_stamps(dlv_ctx* ctx, void* buf_dev) {
    if (!ctx) return DLV_EINVAL;
    ctx->stamp_buf = buf_dev;
    return DLV_OK;
}

int dlv_debug_set_format(dlv_ctx* ctx, int precision) {
    if (!ctx || (precision != DLV_PREC_BF16_ALL && precision != DLV_PREC_F16)) return DLV_EINVAL;  // (one layer: one format)
    ctx->debug_f16 = precision == DLV_PREC_F16;
    return DLV_OK;
}

int dlv_set_lanes(dlv_ctx* ctx, int lanes) {
    if (!ctx || lanes < 1 || lanes > DLV_MAX_LANES) return DLV_EINVAL;
    ctx->lanes_wanted = lanes;
    return DLV_OK;
}

int dlv_prof_enable(dlv_ctx* ctx, int on) {
    if (!ctx) return DLV_EINVAL;
    if (!on) DLV_TRY(prof_drain(ctx));
    ctx->prof_on = on != 0;
    return DLV_OK;
}
int dlv_prof_reset(dlv_ctx* ctx) {
    if (!ctx) return DLV_EINVAL;
    DLV_TRY(prof_drain(ctx));
    ctx->prof_slots.clear();
    return DLV_OK;
}
int dlv_prof_report(dlv_ctx* ctx, dlv_prof_entry* entries, int capacity, int* n_out) {
    if (!ctx || !n_out) return DLV_EINVAL;
    DLV_TRY(prof_drain(ctx));
    int n = (int)ctx->prof_slots.size();
    *n_out = n;
    for (int i = 0; i < n && i < capacity && entries; ++i) {
        memcpy(entries[i].name, ctx->prof_slots[i].name, sizeof(entries[i].name));
        entries[i].launches = ctx->prof_slots[i].launches;
        entries[i].total_ms = ctx->prof_slots[i].total_ms;
        entries[i].flops = ctx->prof_slots[i].flops;
        entries[i].bytes = ctx->prof_slots[i].bytes;
    }
    return DLV_OK;
}

}  // extern "C"
