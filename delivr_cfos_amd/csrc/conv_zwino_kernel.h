// conv_zwino_kernel.h - the hot 3x3x3 convolutions (Cin = 32 -> 32 output channels per block, fp16 operands) as the
// register-resident-weights z-march of conv_zreg_kernel.h with the x direction evaluated as WINOGRAD F(2,3):
//
//     two adjacent outputs of a row from four inputs,  Y = A^T [ (G g) (.) (B^T d) ]   per (ky, kz) tap and channel pair
//     U = G g   = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)        weights, transformed in fp32 on the host side of the launch,
//                                                              rounded to fp16 once (pack_conv_wino_kernel)
//     V = B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)                 input, one v_pk_add_f16 per element pair: a single rounding
//     M_nu     += U_nu[ky][kz] x V_nu                           fp32 MFMA accumulation, 4 positions x 9 taps = 36 MFMAs per
//                                                              (row, 32 voxels, 16 couts) instead of 54
//     Y0 = M0 + M1 + M2,  Y1 = M1 - M2 - M3                    fp32, in the epilogue
//
// 1.5x fewer MFMAs for the layers that hold 91 % of the network's FLOPs.  What it costs in accuracy was measured BEFORE the
// kernel was written (oracle/winograd_gate.py, profiles/r04a_*: the emulated fp16 forward through the reference-arithmetic
// IoU chain: direct 529 flipped voxels of 16.8 M, F(2,3)-x 706, F(2x2,3x3) 853; IoU 0.99977 / 0.99969 / 0.99963), what it
// buys in profiles/microbench/wino_mix.hip (profiles/r04b_*).
//
//   workgroup  = 8 rows x 32 columns of one window, marching along z; 4 waves = 2 output-channel halves x 2 row groups
//   wave       = 16 output channels x (4 rows x 16 tiles of 2 voxels) x 3 rotating accumulator sets (kz) x 4 positions
//   MFMA       : A = U_nu[ky][kz] [16 cout][32 cin] (36 fragments = 144 AGPRs), B = V_nu [32 cin][16 tiles]
//   LDS        : halo planes (10 x 34 voxels x Cin, chunk-planar) with every halo ROW split into its even and its odd
//                voxels (17 + 17): d0 = E[n], d1 = O[n], d2 = E[n+1], d3 = O[n+1] of tile n are four conflict-free
//                ds_read_b128 whose lanes read consecutive 16-byte elements
//   epilogue   : output transform, InstanceNorm partial sums, pack; two v_permlane16_swap turn the (voxel pair x 4 couts)
//                a lane holds into 8 couts of ONE voxel: one 16-byte store per lane, 512 contiguous bytes per chunk
//   schedule, hazards, edge steps: as conv_zreg_kernel.h (hand-placed side work between asm MFMAs, branch-free interior
//                steps, masked edge steps padded with s_nop).
// Reference: the Conv3d -> InstanceNorm3d -> Mish blocks of MONAI's BasicUNet (inference/inference.py:190-197; call site
// inference/sliding_window_inferer.py:222).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "prec16.h"
#include "conv_zreg.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 zw_h2;

constexpr int ZW_HX = 34;   // halo row: 17 even voxels, then 17 odd voxels
constexpr int ZW_TYT = 8;
#ifndef ZW_RA
#define ZW_RA 2             // output rows (of 4) whose accumulators live in AGPRs: 144 weights + 48 * ZW_RA
#endif
#ifndef ZW_NOP
#define ZW_NOP "s_nop 1"
#endif

struct ZwCfg {
    static constexpr int HY = ZW_TYT + 2;
    static constexpr int PL = HY * ZW_HX;
    static constexpr int CS = ((PL + 15) / 16) * 16;
    static constexpr int NCH = 4;
    static constexpr int NIT = (PL + 63) / 64;
    static constexpr int BUF = NCH * CS;
    static constexpr int RW = ZW_TYT / 2;
    static constexpr int NG = RW + 2;
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 16 + 4 * 64 * 4;
};

__device__ __forceinline__ float zw_row_sum16(float v) {
#define ZW_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    ZW_DPP_ADD(0xB1);
    ZW_DPP_ADD(0x4E);
    ZW_DPP_ADD(0x141);
    ZW_DPP_ADD(0x140);
#undef ZW_DPP_ADD
    return v;
}

template <bool AGPR_ACC, bool FIRST, bool PAD>
__device__ __forceinline__ void zw_mfma(f32x4& acc, const u32x4& w, const u32x4& b) {
    if constexpr (PAD) asm volatile(ZW_NOP);
    if constexpr (AGPR_ACC) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc) : "a"(w), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(b));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
    }
}

__device__ __forceinline__ unsigned zw_pk_add(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (zw_h2)(__builtin_bit_cast(zw_h2, a) + __builtin_bit_cast(zw_h2, b)));
}
__device__ __forceinline__ unsigned zw_pk_sub(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (zw_h2)(__builtin_bit_cast(zw_h2, a) - __builtin_bit_cast(zw_h2, b)));
}

template <int N>
using ZwIC = std::integral_constant<int, N>;

__global__ void __launch_bounds__(256, 1)
conv3_zwino_kernel(const uint4* __restrict__ in1, const uint4* __restrict__ wpk, uint4* __restrict__ out, float* __restrict__ partials,
                   int D, int H, int W, int tilesX, int zseg, int nseg, int cout8, int dbg, char* __restrict__ trash) {
    using C = ZwCfg;
    constexpr int RW = C::RW, RA = ZW_RA, RV = RW - RA, NG = C::NG, TYT = ZW_TYT;
    using P = PF16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);
    float* red = reinterpret_cast<float*>(lds + 2 * C::BUF);

    const int n = blockIdx.z;
    const int seg = blockIdx.y % nseg, cb = blockIdx.y / nseg;
    const int tile = dlv_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int y0 = ty * TYT, x0 = tx * 32;
    const int zs = seg * zseg, ze = min(zs + zseg, D);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wave & 1, rg = wave >> 1;
    const int l16 = lane & 15, q = lane >> 4;
    const int plane = H * W;
    const long long vox = (long long)D * plane;
    const bool full_tile = (y0 + TYT <= H) && (x0 + 32 <= W) && !(dbg & 1);

    // ---- this wave's transformed weights -> AGPRs: fragment t = (kz*3 + ky)*4 + nu --------------------------------
    u32x4 wf[36];
    {
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wpk) + ((size_t)(cb * 2 + half) * 36) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            wf[i] = wsrc[(size_t)i * 64];
            asm volatile("" : "+a"(wf[i]));
        }
    }

    // ---- staging map: wave w stages chunk w of the halo plane; element e of the chunk plane = halo voxel (e / 34, e % 34)
    // goes to LDS slot (row, parity, index / 2).  Out-of-window lanes carry a buffer offset beyond the plane: zeros.
    unsigned goff[C::NIT], wpos[C::NIT];
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int e = it * 64 + lane;
        goff[it] = 0xfffffff0u;
        wpos[it] = 0;
        if (e < C::PL) {
            const int xh = e % ZW_HX, yh = e / ZW_HX;
            const int gy = y0 + yh - 1, gx = x0 + xh - 1;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) goff[it] = (unsigned)(gy * W + gx) * 16u;
            wpos[it] = (unsigned)(wave * C::CS + yh * ZW_HX + (xh & 1) * 17 + (xh >> 1));
        }
    }
    const char* const src = reinterpret_cast<const char*>(in1 + ((long long)n * 4 + wave) * vox);
    u32x4 pre[C::NIT];
    const long long plane_b = (long long)plane * 16;
    auto load_piece = [&](int p, int it) __attribute__((always_inline)) {
#ifdef ZW_DIAG
        if (dbg & 4) return;
#endif
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src + (long long)p * plane_b), 0, (int)plane_b, 0x00020000);
        pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[it], 0, 0);
    };
    auto store_piece = [&](auto INT_, int buf, int it, bool plane_ok) __attribute__((always_inline)) {
        u32x4 v = pre[it];
        if (!decltype(INT_)::value && !plane_ok) v = u32x4{0u, 0u, 0u, 0u};
        if (it * 64 + 63 < C::PL || it * 64 + lane < C::PL) lds[wpos[it] + buf * C::BUF] = v;
    };

    // ---- accumulators: 3 rotating output planes x RW rows x 4 transform positions ------------------------------------
    f32x4 accv[3][RV > 0 ? RV : 1][4], acca[3][RA > 0 ? RA : 1][4];
    const f32x4 fzero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int r = 0; r < RV; ++r) accv[s][r][b] = fzero;
#pragma unroll
            for (int r = 0; r < RA; ++r) acca[s][r][b] = fzero;
        }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    // fragment reads: lane (tile l16, k-group q) reads chunk q, halo row j of the wave's row group, slots E[n], O[n], E[n+1], O[n+1]
    const unsigned lbase = (unsigned)(q * C::CS + rg * RW * ZW_HX + l16);
    // output: after the two swaps lane (n, q) holds couts cb*32 + half*16 + 8*(q >> 1) + {0..7} of voxel x0 + 2n + (q & 1):
    // one uint4 of chunk cb*4 + half*2 + (q >> 1)
    char* const obase = reinterpret_cast<char*>(out + ((long long)n * cout8 + cb * 4 + half * 2) * vox);
    const unsigned ooff = ((unsigned)(q >> 1) * (unsigned)vox + (unsigned)((y0 + rg * RW) * W + x0 + 2 * l16 + (q & 1))) * 16u;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)(2u * (unsigned)vox * 16u), 0x00020000);
    const bool xok0 = x0 + 2 * l16 < W, xok1 = x0 + 2 * l16 + 1 < W;  // this lane's two voxels before the swap
    const bool xoks = x0 + 2 * l16 + (q & 1) < W;                     // the voxel it stores
    const unsigned toff = (unsigned)(threadIdx.x * 16u + (blockIdx.x & 15u) * 4096u);  // masked-out stores land here (64 KB)

    const int nzc = (D + 15) / 16;
    auto flush_stats = [&](int zc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = zw_row_sum16(ssum[r]);
            const float b = zw_row_sum16(ssq[r]);
            if (l16 == 0) {
                red[(wave * 16 + q * 4 + r) * 2] = a;
                red[(wave * 16 + q * 4 + r) * 2 + 1] = b;
            }
            ssum[r] = ssq[r] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int i = threadIdx.x;
            const int co = i >> 1, hf = co >> 4;
            const float v = red[((hf)*16 + (co & 15)) * 2 + (i & 1)] + red[((hf + 2) * 16 + (co & 15)) * 2 + (i & 1)];
            const long long nparts = (long long)gridDim.x * nzc;
            const long long part = (long long)zc * gridDim.x + tile;
            partials[(((long long)n * nparts + part) * (cout8 * 8) + cb * 32 + co) * 2 + (i & 1)] = v;
        }
        __syncthreads();
    };
    auto flush_check = [&](int ozf) __attribute__((always_inline)) {
        if (ozf >= zs && ozf < ze && ((ozf & 15) == 15 || ozf == ze - 1)) flush_stats(ozf >> 4);
    };

    // epilogue micro-ops of one finished output row (k = 0: accumulators -> VGPRs; 1..4: output transform of element e;
    // 5..8: statistics of element e; 9: pack, swap, store).  Masked steps: same code with data masks and a trash line.
    f32x4 epi_m[4] = {fzero, fzero, fzero, fzero};
    f32x4 y0v = fzero, y1v = fzero;
    constexpr int EPI_OPS = 10;
    auto epi_op = [&](auto INT_, int set, int r, int k, int oz) __attribute__((always_inline)) {
        constexpr bool INT = decltype(INT_)::value;
        const int oy = y0 + rg * RW + r;
        const bool uok = INT || (oz >= zs && oz < ze && oy < H);
        if (k == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (r < RA) {
                    asm volatile("" : "+a"(acca[set][r < RA ? r : 0][b]));
                    epi_m[b] = acca[set][r < RA ? r : 0][b];
                    asm volatile("" : "+v"(epi_m[b]));
                } else {
                    asm volatile("" : "+v"(accv[set][r >= RA ? r - RA : 0][b]));
                    epi_m[b] = accv[set][r >= RA ? r - RA : 0][b];
                }
            }
        } else if (k <= 4) {
            const int e = k - 1;
            float a, b2;
            asm volatile("v_add_f32 %0, %2, %3\n\tv_sub_f32 %1, %3, %4\n\tv_add_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %5"
                         : "=&v"(a), "=&v"(b2)
                         : "v"(epi_m[0][e]), "v"(epi_m[1][e]), "v"(epi_m[2][e]), "v"(epi_m[3][e]));
            y0v[e] = a;
            y1v[e] = b2;
        } else if (k <= 8) {
            const int e = k - 5;
            const float t0 = (INT || (uok && xok0)) ? y0v[e] : 0.f;
            const float t1 = (INT || (uok && xok1)) ? y1v[e] : 0.f;
            asm volatile("v_add_f32 %0, %2, %0\n\tv_fmac_f32 %1, %2, %2\n\tv_add_f32 %0, %3, %0\n\tv_fmac_f32 %1, %3, %3"
                         : "+v"(ssum[e]), "+v"(ssq[e])
                         : "v"(t0), "v"(t1));
        } else {
            const unsigned a0 = P::pack2(y0v[0], y0v[1]), a1 = P::pack2(y0v[2], y0v[3]);  // voxel 2n,     couts 4q .. 4q+3
            const unsigned b0 = P::pack2(y1v[0], y1v[1]), b1 = P::pack2(y1v[2], y1v[3]);  // voxel 2n + 1
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            const u32x4 u = {s0[0], s1[0], s0[1], s1[1]};
            if constexpr (INT) {
#ifdef ZW_DIAG  // timing-only diagnostics (wrong results): dbg bit 1 = no output stores, bit 2 = no global loads
                if (!(dbg & 2))
#endif
                __builtin_amdgcn_raw_buffer_store_b128(u, ors, (int)ooff, (int)((unsigned)(oz * plane + r * W) * 16u), 0);
            } else {
                char* const real = obase + ((long long)oz * plane + (long long)r * W) * 16 + ooff;
                *reinterpret_cast<u32x4*>((uok && xoks) ? real : trash + toff) = u;
            }
        }
    };

    // ---- one z step (plane p in buffer p & 1; kz=2 -> set A (out[p-1]), kz=1 -> set B (out[p]), kz=0 -> set C (out[p+1]))
    // Group g = input row j of the wave's row group: MFMAs of the output rows j-2 (complete afterwards), j-1, j.  Side work
    // of group g: raw fragment reads of group g+1, epilogue of the row finished one group earlier (row RW-1 of the previous
    // plane in group 0), the input transform of group g+1, piece g of the next plane (LDS write, then the load of the plane
    // after next).
    auto step = [&](auto INT_, int p, auto SA_, auto SB_, auto SC_) __attribute__((always_inline)) {
        constexpr bool INT = decltype(INT_)::value;
        constexpr int SA = decltype(SA_)::value, SB = decltype(SB_)::value, SC = decltype(SC_)::value;
        const bool wr_ok = INT || (p + 1 >= 0 && p + 1 < D && p + 1 <= ze);
        const int pld = INT ? p + 2 : min(max(p + 2, 0), D - 1);
        const int rb = (p & 1) * C::BUF, wb = ((p + 1) & 1);
        u32x4 fb[2][4], raw[4];
        auto load_frag = [&](int g, int i) __attribute__((always_inline)) {
            raw[i] = lds[lbase + rb + g * ZW_HX + (i & 1) * 17 + (i >> 1)];
        };
        auto xform = [&](int g, int e) __attribute__((always_inline)) {  // dword e (two channels) of the four positions
            fb[g & 1][0][e] = zw_pk_sub(raw[0][e], raw[2][e]);
            fb[g & 1][1][e] = zw_pk_add(raw[1][e], raw[2][e]);
            fb[g & 1][2][e] = zw_pk_sub(raw[2][e], raw[1][e]);
            fb[g & 1][3][e] = zw_pk_sub(raw[1][e], raw[3][e]);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) load_frag(0, i);
#pragma unroll
        for (int e = 0; e < 4; ++e) xform(0, e);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = g;
            const int n_frag = (g + 1 < NG) ? 4 : 0;
            const int n_x = (g + 1 < NG) ? 4 : 0;
            const int piece = g < C::NIT ? g : -1;
            const int n_piece = piece >= 0 ? 2 : 0;
            const int erow = (g == 0) ? RW - 1 : (j >= 3 ? j - 3 : -1);
            const int n_epi = erow >= 0 ? EPI_OPS : 0;
            const int n_side = n_frag + n_epi + n_x + n_piece;
            auto side = [&](int k) __attribute__((always_inline)) {
                if (k < n_frag) {
                    load_frag(g + 1, k);
                } else if (k < n_frag + n_epi) {
                    if (g == 0) epi_op(INT_, SC, RW - 1, k - n_frag, p - 2);
                    else epi_op(INT_, SA, erow, k - n_frag, p - 1);
                } else if (k < n_frag + n_epi + n_x) {
                    xform(g + 1, k - n_frag - n_epi);
                } else if (k == n_frag + n_epi + n_x) {
                    store_piece(INT_, wb, piece, wr_ok);
                } else {
                    load_piece(pld, piece);
                }
            };
            int n_mfma = 0;
#pragma unroll
            for (int ky = 2; ky >= 0; --ky)
                if (j - ky >= 0 && j - ky < RW) n_mfma += 12;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (INT) asm volatile("s_nop 1");  // the last side op of the previous group may have written fb[]
#pragma unroll
            for (int ky = 2; ky >= 0; --ky) {
                const int r = j - ky;
                if (r < 0 || r >= RW) continue;
                int vr = 0;
#pragma unroll
                for (int k2 = 2; k2 > ky; --k2)
                    if (j - k2 >= 0 && j - k2 < RW) ++vr;
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    const u32x4 bf = fb[g & 1][nu];
#pragma unroll
                    for (int kz = 0; kz < 3; ++kz) {
                        const u32x4 w = wf[(kz * 3 + ky) * 4 + nu];
                        const int set = kz == 0 ? SC : (kz == 1 ? SB : SA);
                        const bool first = (kz == 0 && ky == 0);
                        if (r < RA) {
                            if (first) zw_mfma<true, true, !INT>(acca[set][r < RA ? r : 0][nu], w, bf);
                            else zw_mfma<true, false, !INT>(acca[set][r < RA ? r : 0][nu], w, bf);
                        } else {
                            if (first) zw_mfma<false, true, !INT>(accv[set][r >= RA ? r - RA : 0][nu], w, bf);
                            else zw_mfma<false, false, !INT>(accv[set][r >= RA ? r - RA : 0][nu], w, bf);
                        }
                        const int m = (vr * 4 + nu) * 3 + kz;
                        const int lo = m * n_side / n_mfma, hi = (m + 1) * n_side / n_mfma;
#pragma unroll
                        for (int t = 0; t < 2; ++t)  // at most 20 side ops per >= 12 MFMAs
                            if (lo + t < hi) side(lo + t);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (g == 0) flush_check(p - 2);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };

    // prologue: the input plane of the first step (zs-1; zeros when zs == 0) into its buffer, the plane after it into the
    // staging registers
    {
        const int p0 = zs - 1;
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) load_piece(max(p0, 0), it);
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) store_piece(std::false_type{}, p0 & 1, it, p0 >= 0);
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) load_piece(min(zs, D - 1), it);
        __syncthreads();
    }
    using T = std::true_type;
    using F = std::false_type;
    const int pmax = min(ze, D - 1);
    auto int_ok = [&](int p) { return full_tile && p - 2 >= zs && p + 4 <= pmax && p + 1 < ze; };
    auto pin_accs = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int r = 0; r < RA; ++r) asm volatile("" : "+a"(acca[st][r][b]));
#pragma unroll
                for (int r = 0; r < RV; ++r) asm volatile("" : "+v"(accv[st][r][b]));
            }
        asm volatile("s_nop 3");
    };
    int p = zs - 1;
    for (;;) {
        for (; p <= ze + 1 && !int_ok(p); p += 3) {
            step(F{}, p + 0, ZwIC<0>{}, ZwIC<1>{}, ZwIC<2>{});
            step(F{}, p + 1, ZwIC<1>{}, ZwIC<2>{}, ZwIC<0>{});
            step(F{}, p + 2, ZwIC<2>{}, ZwIC<0>{}, ZwIC<1>{});
        }
        if (p > ze + 1) break;
        for (; int_ok(p); p += 3) {
            pin_accs();
            step(T{}, p + 0, ZwIC<0>{}, ZwIC<1>{}, ZwIC<2>{});
            step(T{}, p + 1, ZwIC<1>{}, ZwIC<2>{}, ZwIC<0>{});
            step(T{}, p + 2, ZwIC<2>{}, ZwIC<0>{}, ZwIC<1>{});
        }
    }
}

}  // namespace
