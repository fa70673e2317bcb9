// conv_deep.hip - the 3x3x3 convolutions of the deep levels (32^3, 16^3, 8^3 voxels per 128^3 window; Cin 32..256,
// Cout 64..256) as an output-stationary implicit GEMM on v_mfma_f32_16x16x32 whose WEIGHTS ARE SHARED THROUGH LDS.
//
// What bounded the kernel this one replaces (conv3_mfma_kernel, unet_bf16.hip): every wave fetched its own A (weight)
// fragments from L2 - one 1 KiB fragment per two MFMAs per wave = the 64 B per clock a CU gets from the fabric - so the
// matrix pipe ran at 0.12-0.32 of its peak and the same weights crossed the fabric once per workgroup (PMC traffic
// 1.4-4x the algorithmic bytes).  Here
//   workgroup  = 512 threads = 8 waves = 2 per SIMD (<= 256 registers each: while one wave of a SIMD waits - barrier,
//                LDS, the staging writes - the other issues MFMAs), ONE per CU, persistent: it walks over work items
//                (cout tile, window, spatial tile), XCD k over a contiguous eighth of them (neighbouring tiles share halo
//                lines and all of them the weights in that XCD's L2)
//   item       = 512 voxels (4 x 8 x 16, or the whole 8 x 8 x 8 window of the deepest level) x 16*NCB output channels
//   wave       = 64 voxels (4 blocks of 16) x all 16*NCB channels: NCB x 4 accumulators of v_mfma_f32_16x16x32
//   K loop     = 32 input channels at a time ("slab"): the slab's halo tile (6 x 10 x 18 voxels x 4 chunks, 68 KiB) in LDS,
//                x 3 kz groups of 9 taps whose A fragments (9 x NCB KiB) sit in a double-buffered LDS stage: fetched from
//                L2 ONCE per workgroup and group by LDS-DMA (no registers) - two thirds of a group (96 MFMAs per wave)
//                ahead - and read by all 8 waves
//   LDS reads  : per (kz, kx) a wave reads the 6 input rows its 4 output rows see through ky = 0..2 ONCE (9 row pairs at
//                the 8-wide level) and 3 x NCB A fragments: 18 ds_read_b128 per 48 MFMAs (0.375 per MFMA; 1.0 before)
//   staging    : the next slab's halo tile (of the next ITEM after the last slab: the walk is one software pipeline) is
//                fetched into registers while this slab is multiplied and written to LDS between two barriers
//   epilogue   : 16-bit pack + 8-byte stores into the chunk-planar output, InstanceNorm partial sums per (item, channel)
//                in a fixed order (DPP row sums, the 8 waves through LDS).  No bias: every 3x3x3 conv of the network is
//                followed by InstanceNorm, which removes a per-channel constant exactly (as conv_zreg_kernel.h).
// Measured and dropped (profiles/README.md round 5): the same walk with ONE wave per SIMD (4 waves x 128 voxels, 22 reads per 96
// MFMAs, every read issued a phase ahead in the source): 1.3x SLOWER - hipcc's schedule of an intrinsic-MFMA loop needs the
// second wave to fill its waits; the weight stage filled through registers (global_load + ds_write in pieces) instead of
// LDS-DMA: 1.35x slower (the in-order vmcnt ties the pieces to the halo prefetch, 3 spilled registers).
// Weights: the 16-channel A-fragment pack of conv_zreg.hip ([cout/16][tap][cin/32][lane][8]).
// Reference: the Conv3d -> InstanceNorm3d -> Mish blocks down_2..down_4 / upcat_4..upcat_2 of MONAI's BasicUNet
// (inference/inference.py:190-197; call site inference/sliding_window_inferer.py:222).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "prec16.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float cd_f32x4;

template <class P>
__device__ __forceinline__ cd_f32x4 cd_mfma(const uint4& a, const uint4& b, const cd_f32x4& c) {
    if constexpr (P::IS_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(typename P::v8, a), __builtin_bit_cast(typename P::v8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(typename P::v8, a), __builtin_bit_cast(typename P::v8, b), c, 0, 0, 0);
}

// sum over the 16 lanes of a DPP row: every lane of the row ends up with the total (fixed order)
__device__ __forceinline__ float cd_row_sum16(float v) {
#define CD_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    CD_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    CD_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    CD_DPP_ADD(0x141);  // row_half_mirror
    CD_DPP_ADD(0x140);  // row_mirror
#undef CD_DPP_ADD
    return v;
}

template <int TX, int NCB>
struct CdCfg {
    static constexpr int NT = 512;
    static constexpr int RPB = 16 / TX;             // rows of a 16-voxel block (1: one row of 16, 2: two rows of 8)
    static constexpr int TY = 8;
    static constexpr int WPP = TY / (4 * RPB);      // waves per z plane of the tile (a wave owns 4 blocks)
    static constexpr int TZ = 8 / WPP;              // 4 (TX 16) or 8 (TX 8)
    static constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
    static constexpr int HV = HZ * HY * HX;         // halo voxels
    static constexpr int CS = ((HV + 15) / 16) * 16;  // chunk stride in LDS (uint4): a multiple of 16 keeps the four 16-lane
                                                      // groups of a ds_read_b128 (one per k-group = chunk) on distinct 16-byte slots
    static constexpr int NPF = (4 * HV + NT - 1) / NT;      // staged uint4 per thread and slab
    static constexpr int WG_ELEMS = 9 * NCB * 64;           // uint4 of one kz group's A fragments
    static constexpr int NFR = 4 * RPB + 2;         // first rows a wave's blocks see through ky = 0..2
    static constexpr size_t LDS_BYTES = (size_t)(4 * CS + 2 * WG_ELEMS) * 16 + 8 * NCB * 16 * 2 * 4;
};

template <class P, int TX, int NCB>
__global__ void __launch_bounds__(512, 2)
conv3_deep_kernel(const uint4* __restrict__ in1, int c1_8, const uint4* __restrict__ in2, int c2_8, const uint4* __restrict__ wpk,
                  uint4* __restrict__ out, float* __restrict__ partials, int cout, int D, int H, int W, int tilesY, int tilesX,
                  int ntiles, int B) {
    using C = CdCfg<TX, NCB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* tile = reinterpret_cast<uint4*>(smem_raw);            // [4 chunks][CS]
    uint4* wbuf = tile + 4 * C::CS;                              // [2][9 taps][NCB][64 lanes]
    float* red = reinterpret_cast<float*>(wbuf + 2 * C::WG_ELEMS);  // [8 waves][NCB*16 couts][2]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l16 = lane & 15, q = lane >> 4;
    const int zl = wave / C::WPP, yb = (wave % C::WPP) * 4 * C::RPB;  // this wave's plane and first row inside the tile
    const int cin8 = c1_8 + c2_8;
    const int nslab = cin8 / 4, KS = nslab;
    const int vox = D * H * W;  // (deep levels: < 2^24)
    const int nct = cout / (16 * NCB);
    const int nitems = nct * B * ntiles;

    // ---- this workgroup's items: XCD k (= blockIdx.x % 8 under round-robin placement) walks a contiguous eighth ----------
    const int G = gridDim.x;
    int it_begin, it_stride, it_end;
    if (G % 8 == 0) {
        const int x = blockIdx.x & 7, wi = blockIdx.x >> 3;
        const int lo = (int)((long long)nitems * x / 8), hi = (int)((long long)nitems * (x + 1) / 8);
        it_begin = lo + wi;
        it_stride = G / 8;
        it_end = hi;
    } else {
        it_begin = blockIdx.x;
        it_stride = G;
        it_end = nitems;
    }
    if (it_begin >= it_end) return;
    const int nmy = (it_end - it_begin + it_stride - 1) / it_stride;

    struct Item {
        int n, ct, z0, y0, x0, tile;
    };
    auto decode = [&](int item) {
        Item r;
        r.tile = item % ntiles;
        const int rest = item / ntiles;
        r.n = rest % B;
        r.ct = rest / B;
        const int tx = r.tile % tilesX, ty = (r.tile / tilesX) % tilesY, tz = r.tile / (tilesX * tilesY);
        r.z0 = tz * C::TZ;
        r.y0 = ty * C::TY;
        r.x0 = tx * TX;
        return r;
    };

    // ---- staging of one slab's halo tile: element i = threadIdx.x + 512 j -> (chunk, zh, yh, xh).  Fetched with BUFFER loads
    // through a resource that covers exactly the slab's four chunks: an element outside the window carries an offset beyond it
    // and the hardware returns zeros (no select, no branch) ----------------------------------------------------------------------
    unsigned poff[C::NPF];  // byte offset: (chunk * vox + voxel inside the window) * 16, or out of range
    unsigned pexist = 0;
    auto map_item = [&](const Item& it) {
#pragma unroll
        for (int j = 0; j < C::NPF; ++j) {
            const int i = threadIdx.x + C::NT * j;
            poff[j] = 0xfffffff0u;
            if (i < 4 * C::HV) {
                const int c = i / C::HV, r = i % C::HV;
                const int xh = r % C::HX, yh = (r / C::HX) % C::HY, zh = r / (C::HX * C::HY);
                const int gz = it.z0 + zh - 1, gy = it.y0 + yh - 1, gx = it.x0 + xh - 1;
                if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                    poff[j] = (unsigned)(c * vox + (gz * H + gy) * W + gx) * 16u;
            }
        }
    };
#pragma unroll
    for (int j = 0; j < C::NPF; ++j)
        if (threadIdx.x + C::NT * j < 4 * C::HV) pexist |= 1u << j;
    typedef unsigned cd_u32x4 __attribute__((ext_vector_type(4)));
    uint4 pf[C::NPF];
    auto fetch_in = [&](const Item& it, int sl) __attribute__((always_inline)) {
        const int cg = sl * 4;  // a 32-channel slab lies entirely in one of the two sources (c1 % 32 == 0)
        const uint4* src = cg < c1_8 ? in1 + ((long long)it.n * c1_8 + cg) * vox : in2 + ((long long)it.n * c2_8 + (cg - c1_8)) * vox;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(src)), 0, 4 * vox * 16, 0x00020000);
#pragma unroll
        for (int j = 0; j < C::NPF; ++j) {
            const cd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)poff[j], 0, 0);
            pf[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto write_in = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < C::NPF; ++j)
            if ((pexist >> j) & 1u) {
                const int i = threadIdx.x + C::NT * j;
                tile[(i / C::HV) * C::CS + i % C::HV] = pf[j];
            }
    };

    // ---- weights of one kz group: [9 taps][NCB][64 lanes] <- pack [cout/16][27][KS][64]: 9 * NCB fragments of 1 KiB, each one
    // LDS-DMA instruction (`buffer_load_dwordx4 ... lds`: L2 -> LDS without registers), dealt round-robin to the 8 waves.  Inline
    // asm: for the builtin hipcc drains vmcnt in front of EVERY later ds_read (it cannot tell that the stage being filled is not
    // the one being read), which would put the L2 round trip in front of each group instead of beside its MFMAs.  The pieces are
    // waited for by the explicit vmcnt(0) in front of the barrier that publishes them. --------------------------------------------
    typedef unsigned cd_u32x4 __attribute__((ext_vector_type(4)));
    cd_u32x4 wrs;
    {
        const unsigned long long wa = (unsigned long long)reinterpret_cast<uintptr_t>(wpk);
        wrs.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wa);
        wrs.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wa >> 32)) & 0xffffu;  // stride 0: raw buffer
        wrs.z = (unsigned)((long long)cout * cin8 * 8 * 27 * 2);
        wrs.w = 0x00020000u;
    }
    const unsigned wbuf_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wbuf;
    const unsigned dma_voff = (unsigned)lane * 16u;
    auto dma_w = [&](int ct, int sl, int kz, int buf) __attribute__((always_inline)) {
        for (int f = wave; f < 9 * NCB; f += 8) {
            const int cb = f % NCB, t9 = f / NCB;
            const unsigned frag = (unsigned)__builtin_amdgcn_readfirstlane((((ct * NCB + cb) * 27 + kz * 9 + t9) * KS + sl) * 1024);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(wbuf_lds + (unsigned)(buf * C::WG_ELEMS + f * 64) * 16u));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(dma_voff), "s"(wrs), "s"(frag) : "memory", "m0");
        }
    };

    cd_f32x4 acc[NCB][4];
    const int lbase = q * C::CS + (zl * C::HY + yb + (l16 / TX)) * C::HX + (l16 % TX);
    auto compute_group = [&](int kz, int buf, bool wnext, int ct2, int sl2, int kz2) __attribute__((always_inline)) {
        const uint4* wb = wbuf + buf * C::WG_ELEMS + lane;
        const uint4* tb = tile + lbase + kz * C::HY * C::HX;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            uint4 bf[C::NFR];
#pragma unroll
            for (int fr = 0; fr < C::NFR; ++fr) {
                // (the 8-wide level reads row pairs 2 blk + ky: first row 9 is never one)
                if (C::RPB == 2 && fr == C::NFR - 1) continue;
                bf[fr] = tb[fr * C::HX + kx];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                uint4 af[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) af[cb] = wb[((ky * 3 + kx) * NCB + cb) * 64];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[cb][blk] = cd_mfma<P>(af[cb], bf[blk * C::RPB + ky], acc[cb][blk]);
            }
            // the next group's LDS-DMA pieces are issued HERE, a third into the group: right behind the barrier all 8 waves would
            // issue them at once (an LDS-DMA instruction holds its wave's issue for ~100 cycles) in front of their first MFMAs -
            // measured 4-6 % slower on the 32^3 / 16^3 layers (profiles/README.md round 5); two thirds of a group (96 MFMAs per
            // wave) still cover the L2 round trip
            if (kx == 0 && wnext) dma_w(ct2, sl2, kz2, buf ^ 1);
        }
    };

    // ---- the walk: steps = (item, slab), groups = (step, kz); group g multiplies from wbuf[g & 1] --------------------------
    // the walk: steps = (item, slab), groups = (step, kz); group g multiplies from stage g & 1 while the LDS-DMA of group g + 1
    // fills the other one
    Item cur = decode(it_begin);
    map_item(cur);
    fetch_in(cur, 0);
    dma_w(cur.ct, 0, 0, 0);
    int g = 0;
    for (int k = 0; k < nmy; ++k) {
        const bool has_next = k + 1 < nmy;
        const Item nxt = has_next ? decode(it_begin + (k + 1) * it_stride) : cur;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) acc[cb][blk] = cd_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < nslab; ++sl) {
            // (the barrier that ended the previous group: everybody is done with the tile)
            write_in();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the very first group's fragments)
            __syncthreads();  // tile visible
            if (sl + 1 < nslab) fetch_in(cur, sl + 1);
            else if (has_next) {
                map_item(nxt);
                fetch_in(nxt, 0);
            }
#pragma unroll 1
            for (int kz = 0; kz < 3; ++kz) {
                {  // group g + 1's fragments -> the stage group g - 1 read (a barrier ago), in flight during this group's MFMAs
                    int kz2 = kz + 1, sl2 = sl, ct2 = cur.ct;
                    bool any = true;
                    if (kz2 == 3) {
                        kz2 = 0;
                        sl2 = sl + 1;
                        if (sl2 == nslab) {
                            sl2 = 0;
                            ct2 = nxt.ct;
                            any = has_next;
                        }
                    }
                    compute_group(kz, g & 1, any, ct2, sl2, kz2);
                }
                ++g;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces have landed
                __syncthreads();
            }
        }
        // ---- epilogue of the item: stores + InstanceNorm partial sums ---------------------------------------------------
        {
            const int cout8 = cout / 8;
            const int oz = cur.z0 + zl;
            float s[NCB][4], qq[NCB][4];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[cb][r] = qq[cb][r] = 0.f;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const int oy = cur.y0 + yb + blk * C::RPB + (l16 / TX), ox = cur.x0 + (l16 % TX);
                const bool ok = oz < D && oy < H && ox < W;
                const long long o = ((long long)oz * H + oy) * W + ox;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const cd_f32x4 v = acc[cb][blk];
                    if (ok) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s[cb][r] += v[r];
                            qq[cb][r] = fmaf(v[r], v[r], qq[cb][r]);
                        }
                        uint2 u;
                        u.x = P::pack2(v[0], v[1]);
                        u.y = P::pack2(v[2], v[3]);
                        // lane holds couts 4q..4q+3 of its 16-channel block: chunk (q >> 1), bytes (q & 1) * 8 of the voxel's 16
                        uint2* dst = reinterpret_cast<uint2*>(out + ((long long)cur.n * cout8 + (cur.ct * NCB + cb) * 2 + (q >> 1)) * vox + o);
                        dst[q & 1] = u;
                    }
                }
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = cd_row_sum16(s[cb][r]), b = cd_row_sum16(qq[cb][r]);
                    if (l16 == 0) {
                        red[((wave * NCB + cb) * 16 + q * 4 + r) * 2] = a;
                        red[((wave * NCB + cb) * 16 + q * 4 + r) * 2 + 1] = b;
                    }
                }
            __syncthreads();
            if (threadIdx.x < NCB * 32) {
                const int i = threadIdx.x;  // (cout of the tile, sum / sum of squares)
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w * NCB * 32 + i];
                partials[(((long long)cur.n * ntiles + cur.tile) * cout + cur.ct * NCB * 16 + (i >> 1)) * 2 + (i & 1)] = v;
            }
            // (red is next written after three more group barriers, the tile after this item's last group barrier: no barrier here)
        }
        cur = nxt;
    }
}

template <class P, int TX, int NCB>
int cd_launch(dlv_ctx* ctx, const void* in1, int c1, const void* in2, int c2, const void* wpk16, void* out, float* partials, int cout,
              int B, int D, int H, int W, int* nparts) {
    using C = CdCfg<TX, NCB>;
    static dlv_attr_bits attr_set{0};
    if (!dlv_attr_is_set(attr_set, ctx->device)) {
        DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_deep_kernel<P, TX, NCB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        dlv_attr_mark(attr_set, ctx->device);
    }
    const int tilesZ = dlv_cdiv(D, C::TZ), tilesY = dlv_cdiv(H, C::TY), tilesX = dlv_cdiv(W, TX);
    const int ntiles = tilesZ * tilesY * tilesX;
    const long long nitems = (long long)(cout / (16 * NCB)) * B * ntiles;
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const unsigned grid = (unsigned)std::min<long long>(nitems, ncu);
    *nparts = ntiles;
    hipLaunchKernelGGL((conv3_deep_kernel<P, TX, NCB>), dim3(grid), dim3(512), C::LDS_BYTES, ctx->stream, (const uint4*)in1, c1 / 8,
                       (const uint4*)in2, c2 / 8, (const uint4*)wpk16, (uint4*)out, partials, cout, D, H, W, tilesY, tilesX, ntiles, B);
    DLV_LAUNCH_CHECK(ctx, "conv3_deep_kernel");
    return DLV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The transposed convs of the deep levels (ConvTranspose3d k2 s2, Cin 256 -> 128 at 8^3 and 128 -> 64 at 16^3 per 128^3 window):
// eight 1x1x1 channel GEMMs (one per output parity) of few voxels and many weights.  The kernel they ran in before
// (deconv2_wst_kernel: a wave keeps the fragments of ONE (parity pair, 32-channel block) and walks 4 row segments) re-reads its
// 32 KiB of weights per 128 MFMAs and wave and wastes half of every 16-voxel segment at the 8-wide level: 53 + 45 us per 16
// windows for 4 + 8 GFLOP.  Here
//   workgroup = 8 waves (2 per SIMD) x TWO blocks of 16 consecutive voxels (linear order: a row, or two rows of 8): their
//               input fragments (2 x Cin/32 x 1 KiB per wave) stay in registers for the whole walk;
//   walk      = the (parity, 64-channel output group) pairs of this workgroup's share of the output channels: a pair's A
//               fragments (4 x Cin/32 KiB) are staged in a ring of three LDS stages by LDS-DMA - once per workgroup, read by all
//               8 waves - TWO pairs ahead behind a counted vmcnt (one pair ahead the L2 round trip was in front of every pair:
//               28 instead of 57 us, not the 10 the arithmetic asks for); 8 x Cin/32 MFMAs (v_mfma_f32_16x16x32) per wave and
//               pair, then 8 x 8-byte buffer stores per lane (lanes without a voxel: out-of-range offsets).
// Weights: [parity 8][cout/16][cin/32][lane 64][8] (pack_deconv_w16_kernel).  Input: final (activated) values.
// Reference: the ConvTranspose3d of MONAI's UpCat blocks upcat_4 / upcat_3 (inference/inference.py:190-197).
// ---------------------------------------------------------------------------------------------------------------------------
template <class P>
__global__ void pack_deconv_w16_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cin, int cout) {
    const int KS = cin / 32, CB = cout / 16;
    const long long n = (long long)cin * cout * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int ks = (int)(r % KS);
        r /= KS;
        const int cb = (int)(r % CB);
        const int par = (int)(r / CB);
        const int co = cb * 16 + (lane & 15);
        const int ci = ks * 32 + 8 * (lane >> 4) + j;
        out[i] = (uint16_t)(P::pack2(w[((long long)ci * cout + co) * 8 + par], 0.f) & 0xffffu);
    }
}

template <class P, int KS>
__global__ void __launch_bounds__(512, 2)
deconv2_deep_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ bias, uint4* __restrict__ out,
                    int cout, int D, int H, int W, int groups_per_wg) {
    constexpr int STAGE = 4 * KS * 64;  // uint4 of one (parity, 64-channel group): 4 output blocks of 16 x KS fragments
    constexpr int NST = 3, NBLK = 2;    // LDS stages (a pair is fetched TWO pairs ahead), voxel blocks per wave
    constexpr int ND = 4 * KS / 8;      // LDS-DMA instructions per wave and pair
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* wst = reinterpret_cast<uint4*>(smem_raw);  // [NST][4 cb16][KS][64 lanes]
    float* bias_l = reinterpret_cast<float*>(wst + NST * STAGE);  // [cout]: read with ds_read - a global load inside the walk would
                                                                  // make hipcc drain vmcnt, i.e. wait for the LDS-DMA just issued
    for (int i = threadIdx.x; i < cout; i += 512) bias_l[i] = bias[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l16 = lane & 15, q = lane >> 4;
    const int n = blockIdx.z;
    const int vox = D * H * W, ovox = vox * 8;
    const int cin8 = KS * 4, cout8 = cout / 8, CB = cout / 16;
    const int g0 = blockIdx.y * groups_per_wg;  // this workgroup's 64-channel groups [g0, g0 + groups_per_wg)
    const int OH = 2 * H, OW = 2 * W;
    // input fragments: lane (q, l16) holds channels ks*32 + 8q .. +7 of voxel l16 of block blk
    uint4 bf[NBLK][KS];
    unsigned obase[NBLK];  // byte offset of the voxel's parity-0 output inside this sample's first output chunk, or out of range
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        const int v = ((blockIdx.x * 8 + wave) * NBLK + blk) * 16 + l16;
        const bool vok = v < vox;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bf[blk][ks] = vok ? in[((long long)n * cin8 + ks * 4 + q) * vox + v] : make_uint4(0u, 0u, 0u, 0u);
        const int x = v % W, y = (v / W) % H, z = v / (W * H);
        obase[blk] = vok ? (unsigned)(((2 * z) * OH + 2 * y) * OW + 2 * x) * 16u : 0xf0000000u;  // (+ 16 for the x-parity-1 voxel: odd rows)
    }
    // (the fragments are in their registers BEFORE the first LDS-DMA is issued: hipcc's own wait for them - placed at their
    // first use - would otherwise also wait for every DMA piece in flight at that point)
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(bf[blk][ks].x), "v"(bf[blk][ks].y), "v"(bf[blk][ks].z), "v"(bf[blk][ks].w) : "memory");
    // stores through a buffer resource over this sample's output (a lane without a voxel carries an offset beyond it: the
    // hardware drops its store) - every wave issues the same number of vector-memory instructions per pair, which is what the
    // counted vmcnt below relies on
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out + (long long)n * cout8 * ovox), 0,
                                                                         (int)((long long)cout8 * ovox * 16), 0x00020000);

    typedef unsigned cd_u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned cd_u32x2 __attribute__((ext_vector_type(2)));
    cd_u32x4 wrs;
    {
        const unsigned long long wa = (unsigned long long)reinterpret_cast<uintptr_t>(wpk);
        wrs.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wa);
        wrs.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wa >> 32)) & 0xffffu;
        wrs.z = (unsigned)((long long)cout * KS * 32 * 8 * 2);
        wrs.w = 0x00020000u;
    }
    const unsigned wst_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)wst;
    const unsigned dma_voff = (unsigned)lane * 16u;
    // pair p of this workgroup: x parity p & 1, group g0 + (p >> 1) % groups_per_wg, (z, y) parity p / (2 groups_per_wg): the two x
    // parities of one (z, y, group) follow each other, so that their results - neighbouring output voxels - leave in 16-byte
    // stores (below); a pair's fragments are contiguous in the pack: [par][cb16 = 4 g .. 4 g + 3][ks] = 4 * KS KiB
    const int npairs = 8 * groups_per_wg;
    auto pair_of = [&](int p, int& par, int& g) __attribute__((always_inline)) {
        par = (p / (2 * groups_per_wg)) * 2 + (p & 1);
        g = g0 + (p >> 1) % groups_per_wg;
    };
    auto dma_pair = [&](int p, int st) __attribute__((always_inline)) {
        int par, g;
        pair_of(p, par, g);
        const unsigned base = (unsigned)((par * CB + 4 * g) * KS) * 1024u;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int f = wave + 8 * i;
            const unsigned src = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + (unsigned)f * 1024u));
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(wst_lds + (unsigned)(st * STAGE + f * 64) * 16u));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(dma_voff), "s"(wrs), "s"(src) : "memory", "m0");
        }
    };
    dma_pair(0, 0);
    if (npairs > 1) dma_pair(1, 1);
    if (npairs > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // pair 0 has landed
    cd_u32x2 keep[NBLK][4];  // the packed x-parity-0 result of the even iteration
    int st = 0;
    for (int p = 0; p < npairs; ++p) {
        // (stage (p + 2) % 3 was read in iteration p - 1: everybody passed the barrier that ended it)
        if (p + 2 < npairs) dma_pair(p + 2, st == 0 ? 2 : st - 1);
        int par, g;
        pair_of(p, par, g);
        const uint4* wb = wst + st * STAGE + lane;
        cd_f32x4 acc[NBLK][4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_l + (4 * g + cb) * 16 + 4 * q);
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) acc[blk][cb] = cd_f32x4{b4.x, b4.y, b4.z, b4.w};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const uint4 a = wb[(cb * KS + ks) * 64];
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) acc[blk][cb] = cd_mfma<P>(a, bf[blk][ks], acc[blk][cb]);
            }
        // Stores: a lane holds 8 bytes (channels 4q .. 4q+3) of its voxel's 16-byte chunk word, its row neighbour (q ^ 1) the other 8;
        // the x-parity-1 result of the same voxel is the NEXT output voxel.  The even iteration keeps its packed result; the odd one
        // swaps rows with v_permlane16_swap (odd rows of the first operand <-> even rows of the second): even-q lanes then hold
        // the whole 16-byte word of the parity-0 voxel, odd-q lanes that of the parity-1 voxel - one 16-byte store per lane, a lane
        // pair writes 32 contiguous bytes, a row pair 512 (8-byte stores at a 32-byte stride took 2/3 of the kernel's time)
        if ((p & 1) == 0) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) keep[blk][cb] = cd_u32x2{P::pack2(acc[blk][cb][0], acc[blk][cb][1]), P::pack2(acc[blk][cb][2], acc[blk][cb][3])};
            if (p + 2 < npairs) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const unsigned poff = (unsigned)(((par >> 2) * OH + ((par >> 1) & 1)) * OW) * 16u + (unsigned)(q & 1) * 16u;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const unsigned o0 = P::pack2(acc[blk][cb][0], acc[blk][cb][1]), o1 = P::pack2(acc[blk][cb][2], acc[blk][cb][3]);
                    const auto sx = __builtin_amdgcn_permlane16_swap(keep[blk][cb][0], o0, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap(keep[blk][cb][1], o1, false, false);
                    // (lane of an even row: sx[0], sy[0] = its own parity-0 half, sx[1], sy[1] = the odd row's parity-0 half; odd row:
                    // sx[0], sy[0] = the even row's parity-1 half, sx[1], sy[1] = its own)
                    __builtin_amdgcn_raw_buffer_store_b128(cd_u32x4{sx[0], sy[0], sx[1], sy[1]}, ors,
                                                           (int)(obase[blk] + poff + (unsigned)(q >> 1) * (unsigned)ovox * 16u),
                                                           (int)((unsigned)((4 * g + cb) * 2) * (unsigned)ovox * 16u), 0);
                }
            // this wave's vector-memory instructions in program order: DMA(p+1), DMA(p+2), stores(p): pair p + 1 has landed when all
            // but the youngest ND + 4 NBLK are done
            if (p + 2 < npairs) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND + 4 * NBLK) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NBLK) : "memory");
        }
        __syncthreads();
        st = st == 2 ? 0 : st + 1;
    }
}

template <class P, int KS>
int dd_launch(dlv_ctx* ctx, const void* in, const void* wpk16, const float* bias, void* out, int cout, int B, int D, int H, int W) {
    const int vox = D * H * W, G = cout / 64;
    // output channels are split over workgroups until a launch has a few hundred of them (a property of the layer and window
    // shape, not of the batch: every output element is computed by exactly one workgroup either way)
    int gpw = G;
    while (gpw > 1 && (long long)dlv_cdiv(vox, 256) * (G / gpw) < 16) gpw >>= 1;
    const size_t lds = (size_t)3 * 4 * KS * 64 * 16 + (size_t)cout * 4;
    static dlv_attr_bits attr_set{0};
    if (!dlv_attr_is_set(attr_set, ctx->device)) {
        DLV_HIP(ctx, hipFuncSetAttribute((const void*)deconv2_deep_kernel<P, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dlv_attr_mark(attr_set, ctx->device);
    }
    hipLaunchKernelGGL((deconv2_deep_kernel<P, KS>), dim3(dlv_cdiv(vox, 256), G / gpw, B), dim3(512), lds, ctx->stream, (const uint4*)in,
                       (const uint4*)wpk16, bias, (uint4*)out, cout, D, H, W, gpw);
    DLV_LAUNCH_CHECK(ctx, "deconv2_deep_kernel");
    return DLV_OK;
}

}  // namespace

// which layers the kernel takes: the deep levels of a window (at most 32^3 voxels), channel counts in multiples of 32.  Levels
// smaller than a tile (6 x 6 x 4 and 12 x 12 x 8 of the reference's default 96 x 96 x 64 window, config.json:24-28; 8 x 8 x 4 and
// 4 x 4 x 2 of run_inference's own (64, 64, 32), inference/inference.py:119) run as partly filled tiles - out-of-window lanes
// read zeros and store nothing: still 3-4x the generic kernel, whose waves each pull their own weight fragments from L2 - and
// so do the 32-output-channel layers of a level whose rows are shorter than the 32 voxels the z-reg / z-march tiles need
bool dlv_conv3_deep_supports(int cin, int cout, int c1, int c2, int D, int H, int W) {
    return cin % 32 == 0 && c1 % 32 == 0 && c2 % 32 == 0 && c1 + c2 == cin && cout % 32 == 0 && cout >= 32 && W >= 2 && H >= 2 && D >= 2 &&
           (long long)D * H * W <= 32768;
}

// raw conv output + InstanceNorm partial sums partials[((n * nparts + part) * cout + co) * 2 + {sum, sum of squares}]
int dlv_conv3_deep_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in1, int c1, const void* in2, int c2, const void* wpk16,
                          void* out, float* partials, int B, int D, int H, int W, int* nparts) {
    if (!dlv_conv3_deep_supports(cin, cout, c1, c2, D, H, W)) return dlv_fail(ctx, DLV_EUNSUP, "deep conv: unsupported layer shape");
    // 16-wide tiles of 4 x 8 x 16 voxels; windows whose rows are 8 voxels (the deepest level of a 128^3 window) take 8 x 8 x 8.
    // Output channels per workgroup: 64, or 32 when 64 would leave most of the chip without an item (property of the layer and
    // window shape, never of the batch: the partial sums are per spatial tile either way)
    const bool tx16 = W >= 16;
    const int ntiles = tx16 ? dlv_cdiv(D, 4) * dlv_cdiv(H, 8) * dlv_cdiv(W, 16) : dlv_cdiv(D, 8) * dlv_cdiv(H, 8) * dlv_cdiv(W, 8);
    const bool ncb4 = cout % 64 == 0 && (long long)ntiles * (cout / 64) >= 8;
#define CD_GO(P_, TX_, NCB_) return cd_launch<P_, TX_, NCB_>(ctx, in1, c1, in2, c2, wpk16, out, partials, cout, B, D, H, W, nparts)
    if (f16) {
        if (tx16) {
            if (ncb4) CD_GO(PF16, 16, 4);
            CD_GO(PF16, 16, 2);
        }
        if (ncb4) CD_GO(PF16, 8, 4);
        CD_GO(PF16, 8, 2);
    }
    if (tx16) {
        if (ncb4) CD_GO(PBf16, 16, 4);
        CD_GO(PBf16, 16, 2);
    }
    if (ncb4) CD_GO(PBf16, 8, 4);
    CD_GO(PBf16, 8, 2);
#undef CD_GO
}

// ---- transposed conv of the deep levels -------------------------------------------------------------------------------------
bool dlv_deconv2_deep_supports(int cin, int cout, int D, int H, int W) {
    return (cin == 128 || cin == 256) && cout % 64 == 0 && cout > 0 && (long long)D * H * W <= 32768 && W >= 1;
}
int dlv_pack_deconv_w16(dlv_ctx* ctx, bool f16, const float* w_f32, uint16_t* out, int cin, int cout) {
    if (f16) hipLaunchKernelGGL(pack_deconv_w16_kernel<PF16>, dim3(64), dim3(256), 0, ctx->stream, w_f32, out, cin, cout);
    else hipLaunchKernelGGL(pack_deconv_w16_kernel<PBf16>, dim3(64), dim3(256), 0, ctx->stream, w_f32, out, cin, cout);
    DLV_LAUNCH_CHECK(ctx, "pack_deconv_w16_kernel");
    return DLV_OK;
}
// in: FINAL (activated) values; out: raw transposed-conv output (bias added), chunk-planar (B, cout, 2D, 2H, 2W)
int dlv_deconv2_deep_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in, const void* wpk16, const float* bias, void* out, int B,
                            int D, int H, int W) {
    if (!dlv_deconv2_deep_supports(cin, cout, D, H, W)) return dlv_fail(ctx, DLV_EUNSUP, "deep deconv: unsupported layer shape");
    if (f16) return cin == 256 ? dd_launch<PF16, 8>(ctx, in, wpk16, bias, out, cout, B, D, H, W) : dd_launch<PF16, 4>(ctx, in, wpk16, bias, out, cout, B, D, H, W);
    return cin == 256 ? dd_launch<PBf16, 8>(ctx, in, wpk16, bias, out, cout, B, D, H, W) : dd_launch<PBf16, 4>(ctx, in, wpk16, bias, out, cout, B, D, H, W);
}
