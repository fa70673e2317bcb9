// common.h - context, error plumbing, scratch arena and the in-library kernel timer shared by
// every translation unit of libdelivr_hip.so.  gfx950 only; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/delivr_hip.h"

#define DLV_WAVE 64
#define DLV_MAX_LANES 6

struct DlvProfSlot {
    char name[48];
    int64_t launches = 0;
    double total_ms = 0.0, flops = 0.0, bytes = 0.0;
};
struct DlvProfPending {
    int slot;
    hipEvent_t a, b;
};

// one packed layer of the U-Net in HBM
struct DlvConvLayer {
    int cin = 0, cout = 0;
    float* w_f32 = nullptr;     // (Cout,Cin,27) fp32, as in the checkpoint
    float* bias = nullptr;      // (Cout)
    float* bias16 = nullptr;    // (Cout) bias * 2^-shift: what the 16-bit kernels add (dlv_unet_set_conv_shift)
    int shift = 0;              // the 16-bit packs hold W * 2^-shift: the raw output is stored 2^-shift times smaller, InstanceNorm
                                // (eps * 4^-shift) gives the same normalised value - the fp16 range guard's remedy
    float* gamma = nullptr;     // (Cout)
    float* beta = nullptr;      // (Cout)
    uint16_t* w_bf16 = nullptr; // MFMA A-operand fragment order (see unet_bf16.hip)
    uint16_t* w_f16 = nullptr;  // same order, IEEE half
    uint16_t* w16_bf16 = nullptr;  // v_mfma_f32_16x16x32 A-fragment order (conv_zreg.hip): 16-channel output blocks
    uint16_t* w16_f16 = nullptr;
    // UpCat's first conv with a 32-channel skip and a 32->32 transposed conv (upcat_1): the skip half as a 32-channel z-reg pack
    // and the folded up half (upconv.hip: 2 x 2 x 4 x 8 A-fragments of 1 KiB, corr [8][8][32] fp32)
    uint16_t* wskip_bf16 = nullptr;
    uint16_t* wskip_f16 = nullptr;
    uint16_t* wup_bf16 = nullptr;
    uint16_t* wup_f16 = nullptr;
    float* up_corr = nullptr;
    int up_slices = 0;          // 1: this conv's up half is folded with its transposed conv (upcat_1)
};
struct DlvDeconvLayer {
    int cin = 0, cout = 0;
    float* w_f32 = nullptr;     // (Cin,Cout,8)
    float* bias = nullptr;      // (Cout)
    uint16_t* w_bf16 = nullptr; // fragment order, per output parity
    uint16_t* w_f16 = nullptr;
    uint16_t* w16_bf16 = nullptr;  // Cin >= 128: v_mfma_f32_16x16x32 A-fragment order [parity][cout/16][cin/32][lane][8] (conv_deep.hip)
    uint16_t* w16_f16 = nullptr;
};

enum DlvWsSlot {
    WS_F32_ACT = 0,   // fp32 path activations
    WS_BF16_ACT,      // bf16 path activations
    WS_STATS,         // per-(n,c) sums / scale-shift
    WS_TILE_IN,       // gathered fp32 tiles
    WS_TILE_OUT,      // logits of a batch
    WS_TILE_META,     // window starts / maxima
    WS_BLEND_W,       // Gaussian blend factors (d+h+w floats)
    WS_ERODE,         // distance maps
    WS_CCL,           // CCL scratch
    WS_MISC,
    WS_LANE_ACT0,     // extra pipeline lanes (aux streams): activations, then stats, DLV_MAX_LANES-1 each
    WS_LANE_STATS0 = WS_LANE_ACT0 + DLV_MAX_LANES - 1,
    WS_N_SLOTS = WS_LANE_STATS0 + DLV_MAX_LANES - 1
};

struct dlv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream the NEXT launch goes to (main or aux lane)
    hipStream_t main_stream = nullptr;  // the ctx stream proper
    hipStream_t aux_stream = nullptr;   // lane 1 (kept as a named alias of aux[0])
    hipStream_t aux[DLV_MAX_LANES - 1] = {nullptr};  // extra lanes: batches rotate over the lanes so that the
                                        // HBM-bound kernels of one batch overlap the MFMA kernels of another
    hipEvent_t ev_lane[DLV_MAX_LANES + 1] = {nullptr};  // one per lane + one for the main stream
    int lane = 0;
    int lanes_wanted = 3;  // C3: 1 lane 6.86 s, 2: 6.58, 3: 6.56, 4: 6.63 per pass (profiles/lanes_sweep.sh, r02)
    bool own_stream = false;
    std::string err;
    // weights
    bool weights_loaded = false;
    int features[6] = {0, 0, 0, 0, 0, 0};
    // Gaussian blend of the current dlv_sw_infer_dev call (null = constant weights): d+h+w normalised 1-D factors in HBM
    const float* blend_w = nullptr;
    float blend_min = 0.f;
    float* blend_wsum = nullptr;
    int upconv_dbg = 0;         // DLV_UPCONV_DBG, diagnostic library only (timing, WRONG results): 1 = no stores, 2 = no halo loads
    int upconv_simple = 0;      // dlv_diag_set "upconv_simple": the one-tile-per-workgroup upconv kernel for every shape (A/B, tests)
    int fold_up = 1;            // fold the transposed conv into the first conv of upcat_1 (upconv.hip); dlv_diag_set "no_upconv": the unfolded path
    int zm_variant = 0;         // kernel variant of the z-march conv (0 = default; others: A/B and diagnostic builds)
    void* stamp_buf = nullptr;  // dlv_debug_stamps: timeline buffer of the diagnostic z-march build (DLV_ZM_VARIANT=30)
    int* range_flag = nullptr;  // device words: [0] 0, or 100 - (first layer whose InstanceNorm sums were not finite; 18 = logits);
                                // [1 + layer] float bits of the largest |mean| + 8 sigma of the layer's raw output that exceeded 4096
    int range_last = -1;        // layer of the last DLV_ERANGE (-1: none)
    float range_peak[DLV_N_CONV] = {0};  // ... and the peaks read back with it
    bool range_seq = false;              // dlv_range_recover has changed shifts since the last pass that came to its end
    int range_base[DLV_N_CONV] = {0};    // ... the shifts before its first step (restored when nothing is left to try)
    int range_blind_layer = -1;          // layer whose last step was blind (no block reported a peak): not repeated
    void* zero_page = nullptr;  // 256 zero bytes: source of out-of-window lanes of LDS-DMA loads
    void* blob = nullptr;  // one allocation holding every packed parameter
    size_t blob_bytes = 0;
    DlvConvLayer conv[DLV_N_CONV];
    DlvDeconvLayer deconv[DLV_N_DECONV];
    float* final_w = nullptr;
    float* final_b = nullptr;
    // scratch
    void* ws[WS_N_SLOTS] = {nullptr};
    size_t ws_bytes[WS_N_SLOTS] = {0};
    // timing
    bool debug_f16 = false;  // format used by dlv_debug_layer_bf16
    bool no_zmarch = false;  // test switch: force the generic conv kernel
    // Kernel-selection switches of tests and A/B runs (same results, other kernels).  The library takes NONE of them from the
    // environment (a stray DLV_* in a user's shell must not change kernels): dlv_diag_set (include/delivr_hip_diag.h) sets them
    // per context.  What the library does read from the environment: DLV_LANES, DLV_LAUNCH_LOG, DLV_RCCL_PATH / ROCM_PATH /
    // DLV_FORCE_RCCL (transport), nothing else.
    // bit li: conv block li applies the InstanceNorm + Mish of its first input itself while it stages the planes (no normalisation
    // pass over that tensor).  Default: block 17 (upcat_1.conv_1) only - the one site where it pays: -1.4...-1.7 % of a pass on every
    // workload, masks identical; block 16 (the raw skip tensor into the addend conv) +2.2 %, the level-1 sites +0.3...+0.9 %
    // (profiles/r06z_fuse_sites_ab.txt).  The level-wise switch of rounds 2-6 saw the two level-0 sites cancel.
    int fuse_layers = 1 << 17;
    int fuse_levels = 0;     // bit l: raw tensors of level l are activated by the z-reg conv that stages them (no norm pass)
    int zreg_mask = 3;       // 1 = Cin 32, 2 = Cin 64 layers may take the register-resident-weights conv
    int deep_mask = 2;       // conv_deep.hip: bit 0 = the layers the LDS-weights z-march also takes, bit 1 = the others
    int generic_ncb = 0;     // cout blocks per workgroup of the generic conv (0: its own choice)
    int zreg_dbg = 0;        // 1 = edge-step code on every plane of the z-reg conv
    int deep_small = 1;      // 0 = levels smaller than a tile of conv_deep.hip and its 32-output-channel layers take the generic conv (A/B)
    bool pool_rows_off = false, erode_xy_split = false, erode_z_two_sweeps = false, ccl_simple = false;
    bool resample_simple = false, resample_run16 = false;
    int tiff_chunk = 0;      // planes per staging chunk of dlv_tiff_stack_to_device (0: ~256 MB; tests: the double-buffer hand-over on small planes)
    bool prof_on = false;
    std::vector<DlvProfSlot> prof_slots;
    std::vector<DlvProfPending> prof_pending;
    std::vector<hipEvent_t> prof_free;
};

int dlv_fail(dlv_ctx* ctx, int code, const char* fmt, ...);

#define DLV_HIP(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return dlv_fail((ctx), _e == hipErrorOutOfMemory ? DLV_ENOMEM : DLV_EHIP, "%s failed: %s (%s:%d)", \
                            #expr, hipGetErrorString(_e), __FILE__, __LINE__);                      \
    } while (0)

#define DLV_LAUNCH_CHECK(ctx, what)                                                                 \
    do {                                                                                            \
        hipError_t _e = hipGetLastError();                                                          \
        if (_e != hipSuccess)                                                                       \
            return dlv_fail((ctx), DLV_EHIP, "launch of %s failed: %s", (what), hipGetErrorString(_e)); \
    } while (0)

#define DLV_TRY(expr)                \
    do {                             \
        int _rc = (expr);            \
        if (_rc != DLV_OK) return _rc; \
    } while (0)

// grow-only scratch slot
int dlv_ws_get(dlv_ctx* ctx, int slot, size_t bytes, void** out);
// waits for every stream the context launches on (main, lanes)
int dlv_sync_all(dlv_ctx* ctx);

// kernel timer: DlvProf p(ctx, "name", flops, bytes); <launch>; p.end();
// ... and, when a roctx library is in the process (rocprofv3 --marker-trace; or DLV_ROCTX=1 loads it), a roctx range of the same
// name around the launch(es) of the layer: an external trace shows a forward's layers by name
struct DlvProf {
    dlv_ctx* ctx;
    int idx = -1;
    bool ranged = false;
    DlvProf(dlv_ctx* c, const char* name, double flops, double bytes);
    ~DlvProf();
    void end();
};

static inline int dlv_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// "this kernel's dynamic-LDS attribute is set on this device": one bit per device in a function-local static.  The ranks of
// dlv_sw_infer_sharded are host threads that go through the same launchers, so the bits are atomic (a lost race only
// repeats hipFuncSetAttribute, which is idempotent).
typedef std::atomic<unsigned long long> dlv_attr_bits;
static inline bool dlv_attr_is_set(const dlv_attr_bits& bits, int device) { return (bits.load(std::memory_order_acquire) >> (device & 63)) & 1ull; }
static inline void dlv_attr_mark(dlv_attr_bits& bits, int device) { bits.fetch_or(1ull << (device & 63), std::memory_order_release); }

// XCD-aware tile order for kernels whose gridDim.x enumerates spatial tiles (row-major: x, then y, then z neighbours):
// workgroups go round-robin to the 8 XCDs (each with its own L2), so with gridDim.x a multiple of 8 the XCD of a workgroup
// is blockIdx.x % 8; every XCD gets a contiguous run of tiles instead of every 8th one, neighbouring tiles run at the same
// time and find each other's halo lines in their L2 (z-reg conv, PMC: 1.27x / 1.47x -> 1.06x / 1.08x of the algorithmic bytes)
#ifdef __HIPCC__
// Non-temporal accesses for tensors that are streamed once and are far larger than L2 + MALL (the level-0 tensors of a batch:
// 2.1 GB): `nt` keeps them from displacing the lines other kernels - or the same kernel's halo - will re-read.  Measured
// (profiles/microbench/nt_probe.hip): the in-place norm + Mish pass over 2.15 GB 743 -> 642 us, upconv's P stores 602 -> 540 us.
#ifndef DLV_NT
#define DLV_NT 1
#endif
typedef unsigned dlv_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned dlv_u32x2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ uint4 dlv_ld16(const uint4* p) {
    if constexpr (NT && DLV_NT) {
        const dlv_u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const dlv_u32x4_t*>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void dlv_st16(uint4* p, uint4 v) {
    if constexpr (NT && DLV_NT) __builtin_nontemporal_store(dlv_u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<dlv_u32x4_t*>(p));
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ void dlv_st8(uint2* p, uint2 v) {
    if constexpr (NT && DLV_NT) __builtin_nontemporal_store(dlv_u32x2_t{v.x, v.y}, reinterpret_cast<dlv_u32x2_t*>(p));
    else *p = v;
}
__device__ __forceinline__ int dlv_xcd_tile(unsigned bx, unsigned gx) {
    return (gx % 8 == 0) ? (int)((bx % 8) * (gx / 8) + bx / 8) : (int)bx;
}
#endif

// ---- internal engine entry points (defined in the .hip files) ---------------------------------
int dlv_unet_forward_f32(dlv_ctx* ctx, const float* x, float* logits, int B, int d, int h, int w);
// fmt16 (the 16-bit format of a forward): 0 = bf16 everywhere, 1 = fp16 everywhere, 2 = fp16 at level 0 + bf16 below
static inline int dlv_fmt16(int precision) { return precision == DLV_PREC_F16 ? 1 : (precision == DLV_PREC_BF16 ? 2 : 0); }
int dlv_unet_forward_bf16(dlv_ctx* ctx, const float* x, float* logits, int B, int d, int h, int w, int fmt16);
// bf16 path fused with the tiler: reads the uint16 volume at the given window starts, adds the
// logits into acc (see sw_infer.hip)
int dlv_unet_tiles_bf16(dlv_ctx* ctx, const uint16_t* vol, int Yp, int Xp, const int* starts_dev, int B, int d,
                        int h, int w, int flip_dim, float scale, float* acc, int fmt16);
int dlv_pack_weights_bf16(dlv_ctx* ctx);
int dlv_unet_reserve_16(dlv_ctx* ctx, int B, int d, int h, int w, int lanes);
// z-marching conv for Cout in {32, 64, ...} (blocks of 32), Cin in {32, 64} (conv_zmarch.hip)
int dlv_conv3_zmarch_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in1, int c1, const void* in2, int c2,
                            const void* wpk, const float* bias, void* out, float* partials, int B, int D, int H, int W,
                            int* nparts);
// deep-level conv (conv_deep.hip): LDS-shared weights, persistent workgroups; wpk16 = the 16-channel A-fragment pack
bool dlv_conv3_deep_supports(int cin, int cout, int c1, int c2, int D, int H, int W);
int dlv_conv3_deep_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in1, int c1, const void* in2, int c2, const void* wpk16,
                          void* out, float* partials, int B, int D, int H, int W, int* nparts);
// transposed conv of the deep levels (conv_deep.hip): Cin 128 / 256, LDS-shared weights; `in` holds final (activated) values
bool dlv_deconv2_deep_supports(int cin, int cout, int D, int H, int W);
int dlv_pack_deconv_w16(dlv_ctx* ctx, bool f16, const float* w_f32, uint16_t* out, int cin, int cout);
int dlv_deconv2_deep_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in, const void* wpk16, const float* bias, void* out, int B,
                            int D, int H, int W);
// register-resident-weights z-march conv (conv_zreg.hip): Cout blocks of 32, Cin = 32, 32+32 or 64; ss1 / ss2 =
// InstanceNorm scale/shift of the layer that produced in1 / in2 (applied with Mish while staging) or nullptr
int dlv_conv3_zreg_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in1, int c1, const void* ss1, const void* in2,
                          int c2, const void* ss2, const void* wpk16, void* out, float* partials, int B, int D, int H, int W,
                          int* nparts, const void* addend = nullptr);
int dlv_pack_conv_w16(dlv_ctx* ctx, bool f16, const float* w_f32, uint16_t* out, int cout, int cin, int ctot = 0, int c0 = 0, float wscale = 1.f);
// folded up half of an UpCat block's first conv (upconv.hip): Weff / corr from the conv's and the transposed conv's fp32
// tensors; P = conv3(up-sampled tensor) - const, computed from the ACTIVATED coarse tensor
int dlv_pack_upconv(dlv_ctx* ctx, bool f16, const float* wc, int ctot, int cs, const float* wd, const float* bd, uint16_t* wpk, float* corr,
                    int ci0 = 0, int with_corr = 1, float wscale = 1.f);
bool dlv_upconv2_persistent(const dlv_ctx* ctx, int Dc, int Hc, int Wc);
int dlv_upconv2_launch(dlv_ctx* ctx, bool f16, const void* in, const void* wpk, const float* corr, void* out, int B, int Dc, int Hc, int Wc,
                       int cstride = 4, int c0 = 0);
bool dlv_conv3_zreg_supports(int cin, int cout, int c1, int c2, int W);
// range guard of the 16-bit formats (unet_bf16.hip): reset before a pass / forward, check after it (synchronises the stream)
int dlv_range_reset(dlv_ctx* ctx);
int dlv_range_check(dlv_ctx* ctx, int fmt16);
// api.hip: all 18 block shifts at once - committed only when the repack succeeded (rolled back otherwise); clears the report
int dlv_unet_apply_conv_shifts(dlv_ctx* ctx, const int* shifts);
// api.hip: one step of the recovery policy with the layer / peaks given (dlv_range_recover: the context's own report;
// dlv_comm_range_recover: the maxima over the ranks) -> DLV_OK + *n_changed, or DLV_ERANGE when nothing is left (shifts restored)
int dlv_range_step(dlv_ctx* ctx, int layer, const float* peaks, int* n_changed, const int* force_next /* or nullptr */);
int dlv_range_plan(dlv_ctx* ctx, int layer, const float* peaks, int* nxt, bool* blind);
size_t dlv_bf16_pack_bytes(const int features[6]);
#if defined(__HIPCC__)
// Sum of a value over the 32 lanes of each wave half (lanes 0-31, lanes 32-63) with DPP adds only (five VALU
// instructions, no LDS permute): the total is valid in lanes 16-31 resp. 48-63 - read it from lane 31 / 63.
__device__ __forceinline__ float dlv_half_sum32(float v) {
#define DLV_DPP_ADD(ctrl, rmask)                                                                                      \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))
    DLV_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
    DLV_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
    DLV_DPP_ADD(0x141, 0xf);  // row_half_mirror
    DLV_DPP_ADD(0x140, 0xf);  // row_mirror: every lane holds its row's (16 lanes) total
    DLV_DPP_ADD(0x142, 0xa);  // row_bcast15 into rows 1 and 3: they now hold the 32-lane totals
#undef DLV_DPP_ADD
    return v;
}
#endif


