// conv_zreg_kernel.h - the hot 3x3x3 convolutions (Cout blocks of 32, Cin = 32 or 64, W >= 32) as a z-marching implicit GEMM
// whose WEIGHTS LIVE IN REGISTERS: one wave per SIMD (256 threads, 512 registers per lane), every wave keeps the
// 27 x Cin x 16 weights of ITS 16 output channels as MFMA A-fragments in the accumulator half of the register file
// (108 / 216 AGPRs - "MFMA A/B operands may be AGPRs") for the whole column, so the only LDS traffic of the inner loop is
// the input fragment: one ds_read_b128 per 9 (interior) v_mfma_f32_16x16x32 - against 1.33 reads per MFMA of the
// LDS-resident-weights kernel in conv_zmarch.hip, whose matrix pipe ran at 45 instead of 32 cycles per MFMA.
//
//   workgroup  = TYT rows x 32 columns of one window, marching along z; 4 waves = 2 output-channel halves x 2 row groups
//   wave       = 16 output channels x (TYT/2 rows x 32 columns) x 3 rotating accumulators (kz = 0,1,2)
//   MFMA       : A = weights [16 cout][32 cin of one tap], B = input [32 cin][16 voxels], D[cout][voxel]
//   LDS        : two halo planes ((TYT+2) x 34 voxels x Cin, chunk-planar): plane p+1 is written while plane p is
//                multiplied - ONE barrier per plane
//   staging    : HBM -> registers (issued a whole step before use) -> [InstanceNorm scale/shift + Mish of the PRODUCER
//                layer, applied here so that no separate normalisation pass over the tensor exists] -> LDS
//   epilogue   : per finished output row: InstanceNorm partial sums (fp32, flushed every 16 planes in a fixed order),
//                16-bit pack, 8-byte stores into the chunk-planar output.  No bias: every 3x3x3 conv of the network is
//                followed by InstanceNorm, which removes a per-channel constant exactly.
//   schedule   : hipcc will not place an MFMA A operand in an AGPR (it copies it to a VGPR first, or spills), so the
//                MFMAs are inline asm with explicit register classes, and because an asm statement is opaque to the
//                scheduler the order of a step is written out by hand: after every MFMA a small piece ("side op") of the
//                step's other work - fragment reads of the next group, staging of the next plane, epilogue of the row
//                finished one group earlier - pinned with sched_barrier.  Steps whose guards are all true (the bulk) run
//                from a branch-free instantiation.
//   hazards    : an asm MFMA is invisible to hipcc's hazard recognizer.  The only MFMA result read by other code is a
//                finished accumulator row, and the order below puts >= 36 MFMAs between the row's last MFMA and its first
//                reader (rows that complete in a group are issued FIRST in that group, their epilogue runs in the NEXT group).
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
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // native vector: usable with inline-asm register constraints

constexpr int ZR_HX = 34;
#ifndef ZR_RA16
#define ZR_RA16 4
#endif
#ifndef ZR_RA8
#define ZR_RA8 2
#endif
// wait states in front of every asm MFMA: a compiler-generated VALU write of an MFMA operand (register copy, spill
// reload) right before the asm would otherwise be read stale - hipcc pads nothing for an asm statement
// addend rows in flight (ADD instantiations): the fetch of a row waits, through the in-order vmcnt, for every store issued before it
#ifndef ZR_ADD_AHEAD
#define ZR_ADD_AHEAD 4
#endif
// cache policy of the addend loads (read once) and of the interior steps' output stores (2 = nt)
#ifndef ZR_ADD_AUX
#define ZR_ADD_AUX (DLV_NT ? 2 : 0)
#endif
#ifndef ZR_STORE_AUX
#define ZR_STORE_AUX 0
#endif
#ifndef ZR_LOAD_AUX
#define ZR_LOAD_AUX 0  // the staged halo planes (2 = nt: measured, profiles/README.md)
#endif
#ifndef ZR_NOP
#define ZR_NOP "s_nop 1"
#endif

template <int CIN, int TYT>
struct ZrCfg {
    static constexpr int HY = TYT + 2;
    static constexpr int PL = HY * ZR_HX;             // voxels of one halo plane
    static constexpr int CS = ((PL + 15) / 16) * 16;  // chunk stride in LDS (uint4): a multiple of 16 keeps the four
                                                      // 16-lane groups of a ds_read_b128 on distinct 16-byte slots
    static constexpr int NCH = CIN / 8;               // 8-channel chunks
    static constexpr int KS = CIN / 32;               // k-steps (32 input channels) per tap
    static constexpr int SPW = NCH / 4;               // chunks staged per wave
    static constexpr int NIT = (PL + 63) / 64;        // 64-lane pieces of one chunk plane
    static constexpr int BUF = NCH * CS;              // uint4 per plane buffer
    static constexpr int RW = TYT / 2;                // output rows per wave
    static constexpr int RA = CIN == 32 ? (RW <= 4 ? ZR_RA8 : ZR_RA16) : 1;  // of which accumulate in AGPRs (the rest in VGPRs):
                                                      // Cin 32: 108 weights + 120 = 228 AGPR; Cin 64: 216 + 24 = 240 AGPR
    static constexpr int NG = (RW + 2) * KS;          // MFMA groups per step: (input row, k-step)
    static constexpr int NPIECE = SPW * NIT;          // staged pieces per step (<= NG: one per group)
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 16 + 4 * 64 * 4;
};

// y * tanh(softplus(y)) exactly as unet_bf16.hip's mish_fast (one v_exp + one v_rcp)
__device__ __forceinline__ float zr_mish(float y) {
    const float n = __builtin_amdgcn_exp2f(y * 1.44269504f);
    const float d = fmaf(n, n + 2.f, 2.f);
    return fmaf(-2.f * y, __builtin_amdgcn_rcpf(d), y);
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4): every lane of the row ends up with the total
__device__ __forceinline__ float zr_row_sum16(float v) {
#define ZR_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    ZR_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    ZR_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    ZR_DPP_ADD(0x141);  // row_half_mirror
    ZR_DPP_ADD(0x140);  // row_mirror
#undef ZR_DPP_ADD
    return v;
}

// acc (+)= W (AGPR) x B (VGPR); FIRST: acc = W x B (zero C operand, starts a new output plane)
template <class P, bool AGPR_ACC, bool FIRST, bool PAD>
__device__ __forceinline__ void zr_mfma(f32x4& acc, const u32x4& w, const u32x4& b) {
    if constexpr (PAD) asm volatile(ZR_NOP);  // (adjacent volatile asm statements keep their order)
    if constexpr (P::IS_F16) {
        if constexpr (AGPR_ACC) {
            if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc) : "a"(w), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(b));
        } else {
            if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
        }
    } else {
        if constexpr (AGPR_ACC) {
            if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(w), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(b));
        } else {
            if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
        }
    }
}

template <int N>
using IC = std::integral_constant<int, N>;

// ADD: `addend` (16-bit, the layout of `out`) is added to the accumulators in the epilogue, before the statistics and the
// pack: the "up" half of an UpCat block's first conv, computed from the COARSE tensor by upconv.hip - the 64-channel conv
// becomes this 32-channel one and the up-sampled tensor never exists.  The addend of an output row is fetched ZR_ADD_AHEAD epilogue
// rows ahead (the rows of a z column form one sequence: plane by plane, row by row).
template <class P, int CIN, int TYT, bool ACT, int ADD = 0>
__global__ void __launch_bounds__(256, 1)
conv3_zreg_kernel(const uint4* __restrict__ in1, int c1_8, const float2* __restrict__ ss1, const uint4* __restrict__ in2,
                  int c2_8, const float2* __restrict__ ss2, const uint4* __restrict__ wpk, uint4* __restrict__ out,
                  float* __restrict__ partials, int D, int H, int W, int tilesX, int zseg, int nseg, int cout8, int dbg,
                  char* __restrict__ trash, const uint4* __restrict__ addend) {
    using C = ZrCfg<CIN, TYT>;
    constexpr int RW = C::RW, RA = C::RA, RV = RW - RA, KS = C::KS, NG = C::NG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);                 // two plane buffers of C::BUF
    float* red = reinterpret_cast<float*>(lds + 2 * C::BUF);         // [4 waves][16 couts][2] statistics scratch

    const int n = blockIdx.z;
    const int seg = blockIdx.y % nseg, cb = blockIdx.y / nseg;       // z segment, 32-channel output block
    const int tile = dlv_xcd_tile(blockIdx.x, gridDim.x);  // XCD-aware tile order (common.h)
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int y0 = ty * TYT, x0 = tx * 32;
    const int zs = seg * zseg, ze = min(zs + zseg, D);               // output planes [zs, ze)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wave & 1, rg = wave >> 1;                       // output-channel half, row group
    const int l16 = lane & 15, q = lane >> 4;                        // voxel within a block / k-group = cout quad
    const int plane = H * W;
    const long long vox = (long long)D * plane;
    // every guard of a step is statically true when the tile is full and the step is far enough from the segment ends
    const bool full_tile = (y0 + TYT <= H) && (x0 + 32 <= W) && !(dbg & 1);

    // ---- this wave's weights -> AGPRs (A fragments: lane = [k-group q][cout l16]) ---------------------------------
    u32x4 wf[27 * KS];
    {
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wpk) + ((size_t)(cb * 2 + half) * 27 * KS) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 27 * KS; ++i) {
            wf[i] = wsrc[(size_t)i * 64];
            asm volatile("" : "+a"(wf[i]));
        }
    }

    // ---- staging map: wave w stages chunks w, w+4 of the halo plane; constant along z -------------------------------
    // Zero padding of the halo: without activation on load (the default instantiations) the plane is fetched with BUFFER
    // loads whose resource covers exactly one chunk plane - a lane outside the window carries an offset beyond it and the
    // hardware returns zeros, so the staged piece goes to LDS as it is (no per-dword select in front of every ds_write).
    // With activation on load the zeros must be those of the ACTIVATED tensor: global loads + a select after the Mish.
    constexpr bool BUF = !ACT;
    unsigned goff[C::NIT];  // byte offset of this lane's element within a chunk plane (out-of-window lanes: 0 / out of range)
    unsigned valid = 0;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int e = it * 64 + lane;
        goff[it] = BUF ? 0xfffffff0u : 0u;
        if (e < C::PL) {
            const int xh = e % ZR_HX, yh = e / ZR_HX;
            const int gy = y0 + yh - 1, gx = x0 + xh - 1;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                goff[it] = (unsigned)(gy * W + gx) * 16u;
                valid |= 1u << it;
            }
        }
    }
    const char* src[C::SPW];
    float sc[C::SPW][8], sh[C::SPW][8];  // sc: wave-uniform -> scalar registers
    bool act[C::SPW];
#pragma unroll
    for (int s = 0; s < C::SPW; ++s) {
        const int c = wave + 4 * s;  // chunk of the concatenated input
        const bool first = c < c1_8;
        src[s] = reinterpret_cast<const char*>(first ? in1 + ((long long)n * c1_8 + c) * vox : in2 + ((long long)n * c2_8 + (c - c1_8)) * vox);
        const float2* ss = first ? ss1 : ss2;
        act[s] = ACT && s == 0 && ss != nullptr;  // slot 0 = in1 (the launcher refuses other activation patterns)
        const int cc = first ? c : c - c1_8;
        const int ctot = first ? c1_8 * 8 : c2_8 * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sc[s][k] = 1.f;
            sh[s][k] = 0.f;
            if (act[s]) {
                const float2 v = ss[n * ctot + cc * 8 + k];
                sc[s][k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v.x)));
                sh[s][k] = v.y;
            }
        }
    }
    u32x4 pre[C::SPW][C::NIT];
    const long long plane_b = (long long)plane * 16;
    auto load_piece = [&](int p, int s, int it) __attribute__((always_inline)) {
        if constexpr (BUF) {  // (the lane offset goes into the instruction as it is: no address arithmetic in the VALU)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src[s] + (long long)p * plane_b), 0,
                                                                                (int)plane_b, 0x00020000);
            pre[s][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[it], 0, ZR_LOAD_AUX);
        } else {
            unsigned o = goff[it];
            asm volatile("" : "+v"(o));  // keeps the zero-extension next to the load: SGPR base + 32-bit VGPR offset form
            pre[s][it] = *reinterpret_cast<const u32x4*>(src[s] + (long long)p * plane_b + o);
        }
    };
    const unsigned wbase = (unsigned)(wave * C::CS + lane);  // + buf*BUF + 4*s*CS + it*64
    // one staged piece: 64 consecutive elements of one chunk plane: [norm + Mish] -> zero padding -> LDS
    auto act_elem = [&](int s, int it, int k2) __attribute__((always_inline)) {  // elements 2*k2, 2*k2+1 of the piece
        const unsigned u = pre[s][it][k2];
        const float a = zr_mish(fmaf(P::lo(u), sc[s][2 * k2], sh[s][2 * k2]));
        const float b = zr_mish(fmaf(P::hi(u), sc[s][2 * k2 + 1], sh[s][2 * k2 + 1]));
        pre[s][it][k2] = P::pack2(a, b);
    };
    auto store_piece = [&](auto INT_, int buf, int s, int it, bool plane_ok) __attribute__((always_inline)) {
        u32x4 v = pre[s][it];
        if constexpr (BUF) {  // out-of-window lanes already hold zeros; only a plane that does not exist needs the mask
            if (!decltype(INT_)::value && !plane_ok) v = u32x4{0u, 0u, 0u, 0u};
        } else if (!(((valid >> it) & 1u) && plane_ok)) {
            v = u32x4{0u, 0u, 0u, 0u};  // zero padding of the ACTIVATED tensor
        }
        if (it * 64 + 63 < C::CS || it * 64 + lane < C::CS) lds[wbase + buf * C::BUF + 4 * s * C::CS + it * 64] = v;
    };

    // ---- accumulators: 3 rotating output planes x RW rows x 2 column blocks of 16 voxels ---------------------------
    f32x4 accv[3][RV > 0 ? RV : 1][2], acca[3][RA][2];
    const f32x4 fzero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < RV; ++r) accv[s][r][b] = fzero;
#pragma unroll
            for (int r = 0; r < RA; ++r) acca[s][r][b] = fzero;
        }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    const unsigned lbase = (unsigned)(q * C::CS + rg * RW * ZR_HX + l16);  // + buf*BUF + (ks*4)*CS + j*HX + blk*16 + kx
    // output: lane holds couts cb*32 + half*16 + 4q + {0..3} of voxel (row, blk*16 + l16): 8 bytes at
    // chunk (cb*4 + half*2 + (q >> 1)), byte (q & 1) * 8 of the voxel's uint4
    // (wave-uniform base pointer + 32-bit lane offset; (q >> 1) * vox * 16 < 2^31 is guaranteed by the launcher)
    char* const obase = reinterpret_cast<char*>(out + ((long long)n * cout8 + cb * 4 + half * 2) * vox);
    const unsigned ooff = ((unsigned)(q >> 1) * (unsigned)vox + (unsigned)((y0 + rg * RW) * W + x0 + l16)) * 16u + (unsigned)(q & 1) * 8u;
    // interior steps store through a buffer resource over this wave's two output chunks: lane offset (per column block) in a
    // VGPR that never changes, (plane, row) offset in an SGPR - no per-store address arithmetic in the VALU
    const unsigned ooff_b[2] = {ooff, ooff + 256u};
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)(2u * (unsigned)vox * 16u), 0x00020000);
    const bool xok[2] = {x0 + l16 < W, x0 + 16 + l16 < W};
    // ADD: the addend's two chunks through a buffer resource with the output's offsets (an offset outside it - a plane of a
    // masked step - reads zeros); pbuf[slot][block]: slot = row % AHEAD
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        ADD != 0 ? const_cast<char*>(reinterpret_cast<const char*>(addend + ((long long)n * cout8 + cb * 4 + half * 2) * vox)) : obase, 0,
        (int)(2u * (unsigned)vox * 16u), 0x00020000);
    constexpr int AHEAD = ZR_ADD_AHEAD;  // rows between an addend's fetch and its use (RW % AHEAD == 0: a row's slot is r % AHEAD)
    static_assert(RW % AHEAD == 0, "addend slots");
    u32x2 pbuf[AHEAD][2] = {};
    auto add_fetch = [&](bool valid, int r, int b, int oz) __attribute__((always_inline)) {  // row r (0..RW-1) of plane oz
        // (the scalar offset is not part of the buffer's range check: a row / plane that does not exist reads row 0 of plane 0
        // instead - its value is never used)
        if constexpr (ADD != 0)
            pbuf[r % AHEAD][b] = __builtin_amdgcn_raw_buffer_load_b64(ars, (int)ooff_b[b], valid ? (int)((unsigned)(oz * plane + r * W) * 16u) : 0, ZR_ADD_AUX);
    };
    const unsigned toff = (unsigned)(threadIdx.x * 8u + (blockIdx.x & 31u) * 2048u);  // masked-out stores land here (64 KB)

    const int nzc = (D + 15) / 16;
    auto flush_stats = [&](int zc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = zr_row_sum16(ssum[r]);
            const float b = zr_row_sum16(ssq[r]);
            if (l16 == 0) {
                red[(wave * 16 + q * 4 + r) * 2] = a;
                red[(wave * 16 + q * 4 + r) * 2 + 1] = b;
            }
            ssum[r] = ssq[r] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // value i: cout (i >> 1) of the 32-channel block, sum / sum of squares (i & 1); waves hf and hf+2 hold its rows
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

    // epilogue micro-ops of (row r, block b) of output plane oz, accumulator set `set`: k = 0..3 sums, 4..7 sums of
    // squares, 8 pack + store.  INT: no guards.  Otherwise (steps near the segment ends, partial tiles) the same code with
    // data masks instead of branches: masked-out values add 0 to the statistics and are stored to a trash line.
    // a finished AGPR accumulator is read into VGPRs ONCE (k == 0) and the nine micro-ops of its epilogue use that copy: left to
    // the compiler every use (sum, sum of squares, pack) re-reads the AGPRs - 96 v_accvgpr_read per step instead of 32
    f32x4 epi_v[2] = {fzero, fzero};
    auto epi_op = [&](auto INT_, int set, int r, int b, int k, int oz) __attribute__((always_inline)) {
        constexpr bool INT = decltype(INT_)::value;
        const int oy = y0 + rg * RW + r;
        const bool uok = INT || (oz >= zs && oz < ze && oy < H);  // wave-uniform
        const bool ok = INT || (uok && xok[b]);
        if (k == 0) {
            // Pin the first read of the finished accumulator HERE: left alone, the optimiser hoists the element extracts
            // up to the asm MFMA that produced the value - i.e. in front of the wait states an MFMA result needs before
            // anything but another MFMA may read it (asm MFMAs are invisible to the hazard recognizer).
            if (r < RA) {
                asm volatile("" : "+a"(acca[set][r < RA ? r : 0][b]));
                epi_v[b] = acca[set][r < RA ? r : 0][b];
                asm volatile("" : "+v"(epi_v[b]));
            } else {
                asm volatile("" : "+v"(accv[set][r >= RA ? r - RA : 0][b]));
            }
        }
        if constexpr (ADD != 0) {
            if (k == 0) {  // accumulator + addend -> the copy every later micro-op of this (row, block) uses
                f32x4 a = r < RA ? epi_v[b] : accv[set][r >= RA ? r - RA : 0][b];
                const u32x2 pv = pbuf[r % AHEAD][b];
                a[0] += P::lo(pv[0]);
                a[1] += P::hi(pv[0]);
                a[2] += P::lo(pv[1]);
                a[3] += P::hi(pv[1]);
                epi_v[b] = a;
                // the row AHEAD further on in the sequence (plane by plane, row by row): same slot, now free
                const int r2 = (r + AHEAD) % RW, oz2 = oz + ((r + AHEAD) >= RW ? 1 : 0);
                add_fetch(INT || (oz2 >= 0 && oz2 < D && y0 + rg * RW + r2 < H), r2, b, oz2);
            }
        }
        const f32x4 v = (ADD != 0 || r < RA) ? epi_v[b] : accv[set][r >= RA ? r - RA : 0][b];
        if (k < 4) {
            // pinned (volatile asm keeps its place between the MFMAs; plain C++ adds are re-associated and sunk to the
            // end of the step by the optimiser, where nothing overlaps them)
            float t = ok ? v[k] : 0.f;
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(ssum[k]) : "v"(t));
        } else if (k < 8) {
            float t = ok ? v[k - 4] : 0.f;
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(ssq[k - 4]) : "v"(t));
        } else {
            uint2 u;
            u.x = P::pack2(v[0], v[1]);
            u.y = P::pack2(v[2], v[3]);
            if constexpr (INT) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{u.x, u.y}, ors, (int)ooff_b[b], (int)((unsigned)(oz * plane + r * W) * 16u), ZR_STORE_AUX);
            } else {  // per-lane address: the real voxel or this lane's slot of the trash line
                char* const real = obase + ((long long)oz * plane + (long long)r * W) * 16 + ooff + (unsigned)b * 256u;
                *reinterpret_cast<uint2*>(ok ? real : trash + toff) = u;
            }
        }
    };
    constexpr int EPI_OPS = 18;  // per row: 2 blocks x 9
    auto epi_row_op = [&](auto INT_, int set, int r, int idx, int oz) __attribute__((always_inline)) {
        epi_op(INT_, set, r, idx / 9, idx % 9, oz);
    };

    // ---- one z step ---------------------------------------------------------------------------------------------------
    // Plane p sits in buffer p & 1.  kz=2 -> set A (out[p-1]), kz=1 -> set B (out[p]), kz=0 -> set C (out[p+1]).
    // Group g = (input row j of the wave's row group, k-step ks): MFMAs of the output rows j-2 (ky=2, these rows are
    // complete afterwards), j-1, j.  Side work of group g: fragment reads of group g+1, piece g of the next plane
    // (activation, LDS write, then the load of the plane after next into the freed registers), epilogue of row j-3
    // (complete since the previous group).  Row RW-1 of a plane is emitted in group 0 of the next step (its accumulator
    // set - then set C - is not restarted before group RW-1), its statistics flush follows group 0.
    // Planes that do not exist (p < 0, p >= D) or lie beyond the segment are staged as zeros and multiplied like any
    // other: the edge steps run the same straight-line code as the interior ones, only with the data masks switched on.
    auto step = [&](auto INT_, int p, auto SA_, auto SB_, auto SC_) __attribute__((always_inline)) {
        constexpr bool INT = decltype(INT_)::value;
        constexpr int SA = decltype(SA_)::value, SB = decltype(SB_)::value, SC = decltype(SC_)::value;
        const bool wr_ok = INT || (p + 1 >= 0 && p + 1 < D && p + 1 <= ze);  // plane p+1 exists (else: zeros)
        const int pld = INT ? p + 2 : min(max(p + 2, 0), D - 1);             // plane fetched for the step after next
        const int rb = (p & 1) * C::BUF, wb = ((p + 1) & 1);
        u32x4 fb[2][6];
        auto load_frag = [&](int g, int i) __attribute__((always_inline)) {
            const int j = g / KS, ks = g % KS, kx = i / 2, b = i % 2;
#ifdef ZR_ABL  // bit 7: every fragment read is issued, but into registers no MFMA reads (the MFMAs multiply stale ones): what is
               // left is the reads' issue slots and LDS traffic without the dependency of an MFMA on a read
            if (ZR_ABL & 128) {
                u32x4 t = lds[lbase + rb + (ks * 4) * C::CS + j * ZR_HX + b * 16 + kx];
                asm volatile("" ::"v"(t));
                return;
            }
#endif
            fb[g & 1][i] = lds[lbase + rb + (ks * 4) * C::CS + j * ZR_HX + b * 16 + kx];
        };
#pragma unroll
        for (int i = 0; i < 6; ++i) load_frag(0, i);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = g / KS, ks = g % KS;
            // side work of this group, as a list of micro-ops
            const int n_frag = (g + 1 < NG) ? 6 : 0;
            const int piece = g < C::NPIECE ? g : -1;
            const int ps = piece >= 0 ? piece / C::NIT : 0, pit = piece >= 0 ? piece % C::NIT : 0;
            const int n_act = (piece >= 0 && ACT && ps == 0) ? 4 : 0;
            const int n_piece = piece >= 0 ? 2 : 0;  // LDS write, next load
            const int erow = (g == 0) ? RW - 1 : ((ks == 0 && j >= 3) ? j - 3 : -1);
            const int n_epi = erow >= 0 ? EPI_OPS : 0;
            const int n_side = n_frag + n_act + n_piece + n_epi;
            auto side = [&](int k) __attribute__((always_inline)) {
#ifdef ZR_ABL  // timing-only ablation builds (WRONG results; `make abl`, never part of libdelivr_hip.so): bit 0 no barrier,
               // bit 1 no epilogue (statistics, pack, store), bit 2 no global loads of the plane after next, bit 3 no fragment
               // reads, bit 4 no LDS writes of the next plane, bit 5 / bit 6: 2 of 6 / 3 of 6 fragment reads dropped (the gate of a
               // cout-complete wave layout, which would issue 0.6x the reads: profiles/README.md round 5)
                if ((ZR_ABL & 8) && k < n_frag) return;
                if ((ZR_ABL & 32) && k < n_frag && (k == 2 || k == 5)) return;
                if ((ZR_ABL & 64) && k < n_frag && (k & 1)) return;
                if ((ZR_ABL & 2) && k >= n_frag && k < n_frag + n_epi) return;
                if ((ZR_ABL & 16) && k >= n_frag + n_epi && k <= n_frag + n_epi + n_act) return;
                if ((ZR_ABL & 4) && k > n_frag + n_epi + n_act) return;
#endif
                if (k < n_frag) {
                    load_frag(g + 1, k);
                } else if (k < n_frag + n_epi) {
                    const int i = k - n_frag;
                    if (g == 0) epi_row_op(INT_, SC, RW - 1, i, p - 2);
                    else epi_row_op(INT_, SA, erow, i, p - 1);
                } else if (k < n_frag + n_epi + n_act) {
                    if (ps == 0) act_elem(ps, pit, k - n_frag - n_epi);  // (ACT instantiation: in1 is always raw)
                } else if (k == n_frag + n_epi + n_act) {
                    store_piece(INT_, wb, ps, pit, wr_ok);
                } else {
                    load_piece(pld, ps, pit);
                }
            };
            // MFMAs: rows j-2 (complete afterwards) first, then j-1, then j
            int n_mfma = 0;
#pragma unroll
            for (int ky = 2; ky >= 0; --ky)
                if (j - ky >= 0 && j - ky < RW) n_mfma += 18;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 2; ky >= 0; --ky) {
                const int r = j - ky;
                if (r < 0 || r >= RW) continue;
                int vr = 0;  // index of this row among the group's rows
#pragma unroll
                for (int k2 = 2; k2 > ky; --k2)
                    if (j - k2 >= 0 && j - k2 < RW) ++vr;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int kz = 0; kz < 3; ++kz) {
                        // the two column blocks innermost: consecutive MFMAs share the WEIGHT fragment (A) and alternate between two
                        // input fragments; with kz innermost (one input fragment for three MFMAs, a new weight fragment for each) the
                        // step is 0.9-1.2 % slower, interleaved over 7 rounds (profiles/README.md r05z_mfma_order_ab*)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const u32x4 bf = fb[g & 1][kx * 2 + b];
                            const u32x4 w = wf[((kz * 3 + ky) * 3 + kx) * KS + ks];
                            const int set = kz == 0 ? SC : (kz == 1 ? SB : SA);
                            const bool first = (kz == 0 && ky == 0 && ks == 0 && kx == 0);
                            if (r < RA) {
                                if (first) zr_mfma<P, true, true, !INT>(acca[set][r < RA ? r : 0][b], w, bf);
                                else zr_mfma<P, true, false, !INT>(acca[set][r < RA ? r : 0][b], w, bf);
                            } else {
                                if (first) zr_mfma<P, false, true, !INT>(accv[set][r >= RA ? r - RA : 0][b], w, bf);
                                else zr_mfma<P, false, false, !INT>(accv[set][r >= RA ? r - RA : 0][b], w, bf);
                            }
                            // the side ops that belong behind MFMA m of this group
                            const int m = ((vr * 3 + kx) * 3 + kz) * 2 + b;
                            const int lo = m * n_side / n_mfma, hi = (m + 1) * n_side / n_mfma;
#pragma unroll
                            for (int t = 0; t < 3; ++t)  // constant trip count (at most 30 side ops per >= 18 MFMAs)
                                if (lo + t < hi) side(lo + t);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
            if (g == 0) flush_check(p - 2);  // plane p-2 is complete (its last row was emitted in this group)
        }
#ifdef ZR_ABL
        if (ZR_ABL & 16) {  // the loads stay alive although nothing writes them to LDS
#pragma unroll
            for (int q2 = 0; q2 < C::NPIECE; ++q2) asm volatile("" ::"v"(pre[q2 / C::NIT][q2 % C::NIT]));
        }
#endif
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes of plane p+1 have landed
#ifdef ZR_ABL
        if (!(ZR_ABL & 1))
#endif
        __builtin_amdgcn_s_barrier();        // plane p+1 visible; everybody is done reading plane p
    };

    // prologue: the input plane of the first step (zs-1; zeros when zs == 0) into its buffer, the plane after it into
    // the staging registers
    {
        const int p0 = zs - 1;
#pragma unroll
        for (int q2 = 0; q2 < C::NPIECE; ++q2) load_piece(max(p0, 0), q2 / C::NIT, q2 % C::NIT);
#pragma unroll
        for (int q2 = 0; q2 < C::NPIECE; ++q2) {
            const int s = q2 / C::NIT, it = q2 % C::NIT;
            if (ACT && s == 0) {
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) act_elem(s, it, k2);
            }
            store_piece(std::false_type{}, p0 & 1, s, it, p0 >= 0);
        }
#pragma unroll
        for (int q2 = 0; q2 < C::NPIECE; ++q2) load_piece(min(zs, D - 1), q2 / C::NIT, q2 % C::NIT);
        __syncthreads();
    }
    // three steps per iteration: the accumulator roles are static; steps ze and ze+1 only finish what is pending
    // (ze: kz = 2 contributions of plane ze to out[ze-1]; ze+1: the last row of out[ze-1])
    using T = std::true_type;
    using F = std::false_type;
    const int pmax = min(ze, D - 1);  // last plane that is fetched
    auto int_ok = [&](int p) { return full_tile && p - 2 >= zs && p + 4 <= pmax && p + 1 < ze; };
    // every accumulator is (re)defined at this point: whatever register copies the allocator needs at a control-flow join
    // (loop entry / back edge) land in front of it, and the wait states behind it cover them - inside the interior loop
    // the asm MFMAs carry no padding
    auto pin_accs = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int r = 0; r < RA; ++r) asm volatile("" : "+a"(acca[st][r][b]));
#pragma unroll
                for (int r = 0; r < RV; ++r) asm volatile("" : "+v"(accv[st][r][b]));
            }
        asm volatile("s_nop 3");
    };
    int p = zs - 1;
    for (;;) {  // at most two rounds: edge steps, interior steps, edge steps
        for (; p <= ze + 1 && !int_ok(p); p += 3) {
            step(F{}, p + 0, IC<0>{}, IC<1>{}, IC<2>{});
            step(F{}, p + 1, IC<1>{}, IC<2>{}, IC<0>{});
            step(F{}, p + 2, IC<2>{}, IC<0>{}, IC<1>{});
        }
        if (p > ze + 1) break;
        for (; int_ok(p); p += 3) {
            pin_accs();
            step(T{}, p + 0, IC<0>{}, IC<1>{}, IC<2>{});
            step(T{}, p + 1, IC<1>{}, IC<2>{}, IC<0>{});
            step(T{}, p + 2, IC<2>{}, IC<0>{}, IC<1>{});
        }
    }
}


// host side of one instantiation: LDS attribute + launch
template <class P, int CIN, int TYT, bool ACT, int ADD = 0>
int zr_launch(dlv_ctx* ctx, const ZrArgs& a) {
    static dlv_attr_bits attr_set{0};  // bit per device
    if (!dlv_attr_is_set(attr_set, ctx->device)) {
        DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_zreg_kernel<P, CIN, TYT, ACT, ADD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)ZrCfg<CIN, TYT>::LDS_BYTES));
        dlv_attr_mark(attr_set, ctx->device);
    }
    hipLaunchKernelGGL((conv3_zreg_kernel<P, CIN, TYT, ACT, ADD>), dim3(a.gx, a.gy, a.gz), dim3(256), (ZrCfg<CIN, TYT>::LDS_BYTES), ctx->stream,
                       (const uint4*)a.in1, a.c1_8, (const float2*)a.ss1, (const uint4*)a.in2, a.c2_8, (const float2*)a.ss2,
                       (const uint4*)a.wpk16, (uint4*)a.out, a.partials, a.D, a.H, a.W, a.tilesX, a.zseg, a.nseg, a.cout8, a.dbg,
                       a.trash, (const uint4*)a.addend);
    DLV_LAUNCH_CHECK(ctx, "conv3_zreg_kernel");
    return DLV_OK;
}

}  // namespace
