// upconv.hip - the "up" half of an UpCat block's first convolution WITHOUT the up-sampled tensor.
//
// MONAI's UpCat (BasicUNet, inference/inference.py:190-197): u = ConvTranspose3d(k2,s2)(x_coarse); y = Conv3d(k3)(cat[skip, u]).
// By linearity  y = conv3(skip; Wc[:, :Cs]) + conv3(u; Wc[:, Cs:]),  and the second term is a linear map of the COARSE tensor:
// a fine voxel v with parity p (per axis) sees, through its 27 taps, the fine voxels v + t, t in {-1,0,1}^3, i.e. the coarse
// voxels (v >> 1) + d with d in {-1,0} (p = 0) or {0,+1} (p = 1) per axis - a 2x2x2 coarse neighbourhood - each through its
// own transposed-conv parity.  Folding the two weight sets,
//     Weff[p][d][co][ci] = sum over fine taps t that map to d, sum over c'  Wc[co][Cs + c'][t] * Wd[ci][c'][(p + t) & 1]
// gives  conv3(u)[co, v] = sum_d Weff[p(v)][d] x_coarse[(v >> 1) + d]  (zero outside the window: a fine voxel is outside
// exactly when its coarse voxel is)  + the transposed conv's bias seen through the taps that lie inside the window.
//   * 8 coarse taps instead of 27 fine ones for this half: 8192 instead of 27648 MAC per output voxel,
//   * the up-sampled tensor (2.1 GB per 16 windows of 128^3) is neither written nor read, the transposed-conv kernel is gone,
//   * one rounding less: Weff is formed in fp32 and rounded to the 16-bit format once.
// This kernel writes P = conv3(u) - const (16-bit, chunk-planar, fine resolution); the z-reg conv of the skip half adds it in
// its epilogue (conv_zreg_kernel.h, ADD).  The constant is the interior bias term sum_27 Wc bd, the same for every voxel of
// a channel: the InstanceNorm that follows removes it exactly; voxels on the window's faces, where some taps fall outside,
// get the difference (`corr`).
//
//   workgroup = coarse tile 4 x 8 x 16 (fine 8 x 16 x 32), halo tile of the ACTIVATED coarse tensor in LDS
//   4 waves   = 2 output-channel halves x 2 plane parities; a wave keeps its 4 (py, px) x 8 taps = 32 A-fragments in registers
//   MFMA      : v_mfma_f32_16x16x32, A = Weff[p][d] [16 cout][32 cin], B = coarse row segment [32 cin][16 voxels]
//   epilogue  : the even-x and the odd-x result of a lane are interleaved with two v_permlane16_swap: one 16-byte store
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "prec16.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int UC_TZ = 4, UC_TY = 8, UC_TX = 16;
constexpr int UC_HZ = UC_TZ + 2, UC_HY = UC_TY + 2, UC_HX = UC_TX + 2;
constexpr int UC_CS = ((UC_HZ * UC_HY * UC_HX + 15) / 16) * 16;  // chunk stride (uint4): a multiple of 16 keeps the 16-lane groups apart

// Weff / corr from the fp32 checkpoint tensors.  wc: (Cout, Ctot, 27) of the conv, wd: (Cin, Cup, 8) of the transposed conv, bd: (Cup).
// One pack covers 32 input channels [ci0, ci0 + 32) of the transposed conv: a 64-channel coarse tensor (upcat_2) is two packs, two
// launches and two addends (linearity in the input channels).
// out A-fragments: [pz][half][pyx 4][tap 8][lane 64][8]:  cout = half*16 + (lane & 15), cin = 8*(lane >> 4) + j, tap = (dz*2 + dy)*2 + dx
// corr: [p 8][mask 8][cout 32] = - sum over the taps d that leave the window under `mask` (bit 2: z, 1: y, 0: x; the tap that
// leaves is index 0 on an even, index 1 on an odd coordinate)  of  sum_{t -> d} sum_c' Wc[co][Cs+c'][t] bd[c']
template <class P>
__global__ void pack_upconv_kernel(const float* __restrict__ wc, int ctot, int cs, const float* __restrict__ wd, const float* __restrict__ bd,
                                   uint16_t* __restrict__ wpk, float* __restrict__ corr, int ci0, int with_corr, float wscale) {
    const int cup = 32, cin = 32;
    // taps of one axis that map (parity p) to coarse index i: t in {-1,0,1} with floor((p + t) / 2) + 1 - p == i
    auto maps = [](int p, int t, int i) { return ((p + t + 2) >> 1) - 1 + 1 - p == i; };
    const int total = 2 * 2 * 4 * 8 * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, tap = (e >> 9) & 7, pyx = (e >> 12) & 3, half = (e >> 14) & 1, pz = (e >> 15) & 1;
        const int co = half * 16 + (lane & 15), ci = 8 * (lane >> 4) + j;
        const int p[3] = {pz, pyx >> 1, pyx & 1}, d[3] = {tap >> 2, (tap >> 1) & 1, tap & 1};
        float s = 0.f;
        for (int tz = -1; tz <= 1; ++tz)
            for (int ty = -1; ty <= 1; ++ty)
                for (int tx = -1; tx <= 1; ++tx) {
                    if (!maps(p[0], tz, d[0]) || !maps(p[1], ty, d[1]) || !maps(p[2], tx, d[2])) continue;
                    const int t = ((tz + 1) * 3 + (ty + 1)) * 3 + (tx + 1);
                    const int par = (((p[0] + tz) & 1) * 2 + ((p[1] + ty) & 1)) * 2 + ((p[2] + tx) & 1);
                    for (int c = 0; c < cup; ++c)
                        s = __fmaf_rn(wc[((long long)co * ctot + cs + c) * 27 + t], wd[((long long)(ci0 + ci) * cup + c) * 8 + par], s);
                }
        wpk[e] = (uint16_t)(P::pack2(s * wscale, 0.f) & 0xffffu);  // (wscale: the conv's 2^-shift, a linear factor of the whole product)
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 8 * 8 * 32; e += gridDim.x * blockDim.x) {
        const int co = e & 31, mask = (e >> 5) & 7, pi = e >> 8;
        const int p[3] = {pi >> 2, (pi >> 1) & 1, pi & 1}, out[3] = {(mask >> 2) & 1, (mask >> 1) & 1, mask & 1};
        float s = 0.f;
        for (int tz = -1; tz <= 1; ++tz)
            for (int ty = -1; ty <= 1; ++ty)
                for (int tx = -1; tx <= 1; ++tx) {
                    const int tt[3] = {tz, ty, tx};
                    bool gone = false;
                    for (int a = 0; a < 3; ++a) {
                        const int di = ((p[a] + tt[a] + 2) >> 1) - 1 + 1 - p[a];  // coarse index 0 / 1 of this tap
                        if (out[a] && di == p[a]) gone = true;                 // the leaving one: index 0 (even), 1 (odd)
                    }
                    if (!gone) continue;
                    const int t = ((tz + 1) * 3 + (ty + 1)) * 3 + (tx + 1);
                    for (int c = 0; c < cup; ++c) s = __fmaf_rn(wc[((long long)co * ctot + cs + c) * 27 + t], bd[c], s);
                }
        corr[e] = with_corr ? -s * wscale : 0.f;  // (a later K-slice of the same transposed conv: the bias terms are in the first slice's table)
    }
}

template <class P>
__global__ void __launch_bounds__(256) upconv2_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ corr,
                                                      uint4* __restrict__ out, int Dc, int Hc, int Wc, int tilesY, int tilesX, int cstride, int c0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char uc_smem[];
    uint4* tile = reinterpret_cast<uint4*>(uc_smem);  // [4 chunks][UC_CS]: halo tile of the activated coarse tensor
    // the face corrections come from LDS: a global load inside the loop would wait (vmcnt counts stores too) for every store the
    // wave has in flight - 64 times per workgroup on the tiles that touch an x face (the first version: 60 of its 61 us)
    float* corr_l = reinterpret_cast<float*>(tile + 4 * UC_CS);  // [8][8][32]
    for (int i = threadIdx.x; i < 8 * 8 * 32; i += 256) corr_l[i] = corr[i];
    const int n = blockIdx.z;
    const int t = dlv_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = t % tilesX, ty = (t / tilesX) % tilesY, tz = t / (tilesX * tilesY);
    const int z0 = tz * UC_TZ, y0 = ty * UC_TY, x0 = tx * UC_TX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wave & 1, pz = wave >> 1;
    const int l16 = lane & 15, q = lane >> 4;
    const long long voxc = (long long)Dc * Hc * Wc;
    const int D = 2 * Dc, H = 2 * Hc, W = 2 * Wc;
    const long long vox = (long long)D * H * W;

    // ---- this wave's folded weights: (py, px) x 8 taps -----------------------------------------------------------
    uint4 wf[32];
    {
        const uint4* wsrc = wpk + ((size_t)(pz * 2 + half) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) wf[i] = wsrc[(size_t)i * 64];
    }
    // ---- halo tile: zeros outside the window.  All loads are issued before the first LDS write: a load -> wait -> write chain
    // per element exposed the memory latency 17 times per workgroup (half of the first version's 61 us per workgroup) -------
    {
        constexpr int NEL = 4 * UC_HZ * UC_HY * UC_HX, NIT = (NEL + 255) / 256;
        uint4 v[NIT];
        int dst[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + 256 * it;
            const int xh = i % UC_HX;
            int r = i / UC_HX;
            const int yh = r % UC_HY;
            r /= UC_HY;
            const int zh = r % UC_HZ, c = r / UC_HZ;
            const int gz = z0 + zh - 1, gy = y0 + yh - 1, gx = x0 + xh - 1;
            v[it] = make_uint4(0u, 0u, 0u, 0u);
            dst[it] = i < NEL ? c * UC_CS + (zh * UC_HY + yh) * UC_HX + xh : -1;
            if (i < NEL && (unsigned)gz < (unsigned)Dc && (unsigned)gy < (unsigned)Hc && (unsigned)gx < (unsigned)Wc)
                v[it] = in[((long long)n * cstride + c0 + c) * voxc + ((long long)gz * Hc + gy) * Wc + gx];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (dst[it] >= 0) tile[dst[it]] = v[it];
    }
    __syncthreads();
    const int lbase = q * UC_CS + l16;
    char* const obase = reinterpret_cast<char*>(out + ((long long)n * 4 + half * 2) * vox);
    const bool xlo = (x0 + l16 == 0), xhi = (x0 + l16 == Wc - 1);
    for (int cz = 0; cz < UC_TZ; ++cz) {
        const int gz = z0 + cz;
        if (gz >= Dc) break;  // (workgroup-uniform)
        const int fz = 2 * gz + pz;
        const bool zout = pz == 0 ? gz == 0 : gz == Dc - 1;
        for (int cy = 0; cy < UC_TY; ++cy) {
            const int gy = y0 + cy;
            if (gy >= Hc) break;
            // the 18 fragments both row parities need (2 coarse planes x 3 rows x 3 x offsets) are read up front: one LDS latency per
            // 32 MFMAs instead of one per 4 (the first version waited for three reads in front of every group of four MFMAs: 973 us)
            uint4 bf[2][3][3];
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int yy = 0; yy < 3; ++yy)
#pragma unroll
                    for (int xs = 0; xs < 3; ++xs) bf[dz][yy][xs] = tile[lbase + ((cz + pz + dz) * UC_HY + cy + yy) * UC_HX + xs];
            f32x4 accs[2][2];
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) accs[py][px] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int py = 0; py < 2; ++py)
#pragma unroll
                            for (int px = 0; px < 2; ++px) {  // (four independent accumulators between two uses of one)
                                const uint4 a = wf[(py * 2 + px) * 8 + (dz * 2 + dy) * 2 + dx];
                                const uint4 bb = bf[dz][py + dy][px + dx];
                                if constexpr (P::IS_F16)
                                    accs[py][px] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(typename P::v8, a),
                                                                                          __builtin_bit_cast(typename P::v8, bb), accs[py][px], 0, 0, 0);
                                else
                                    accs[py][px] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(typename P::v8, a),
                                                                                           __builtin_bit_cast(typename P::v8, bb), accs[py][px], 0, 0, 0);
                            }
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                f32x4 acc[2] = {accs[py][0], accs[py][1]};
                // faces of the window: the bias terms of the taps that fall outside are missing relative to the interior constant
                const bool yout = py == 0 ? gy == 0 : gy == Hc - 1;
                if (zout || yout || xlo || xhi) {
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int mask = (zout ? 4 : 0) | (yout ? 2 : 0) | ((px == 0 ? xlo : xhi) ? 1 : 0);
                        if (mask) {
                            const float* cr = corr_l + (((pz * 2 + py) * 2 + px) * 8 + mask) * 32 + half * 16 + 4 * q;
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[px][r] += cr[r];
                        }
                    }
                }
                const unsigned a0 = P::pack2(acc[0][0], acc[0][1]), a1 = P::pack2(acc[0][2], acc[0][3]);  // fine x = 2n
                const unsigned b0 = P::pack2(acc[1][0], acc[1][1]), b1 = P::pack2(acc[1][2], acc[1][3]);  // fine x = 2n + 1
                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                // lane (n, q): fine voxel x = 2 (x0 + n) + (q & 1), couts half*16 + 8 (q >> 1) .. + 7
                const int fy = 2 * gy + py, fx = 2 * (x0 + l16) + (q & 1);
                if (x0 + l16 < Wc)
                    *reinterpret_cast<uint4*>(obase + ((long long)(q >> 1) * vox + ((long long)fz * H + fy) * W + fx) * 16) =
                        make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
}


// ---- the persistent, hand-scheduled form (the one that runs when the coarse dims are multiples of the tile) ---------------
// The kernel above spends its time around the MFMAs (1 wave per SIMD: every LDS wait, every exec-masked branch of the face
// correction, the weights + halo tile fetched anew by each of the 4096 workgroups of a batch: SQ_INSTS_VALU 3.8 per MFMA,
// matrix pipe busy 0.29 - profiles/r04t_upconv_pmc.txt).  Here, in the manner of conv_zreg_kernel.h:
//   * one workgroup per CU walks over tiles: the 32 A-fragments of a wave are loaded ONCE and live in AGPRs (asm MFMAs),
//   * the halo tile of the NEXT tile is staged while this one is multiplied (two LDS tiles; wave w stages chunk w: 17 pieces
//     of 64 voxels by LDS-DMA - `buffer_load_dwordx4 ... lds`, one per iteration: a lane outside the window carries an offset
//     beyond the resource and the hardware writes zeros (profiles/microbench/lds_dma_oob.hip); no staging registers, one counted
//     vmcnt + one barrier per tile.  Register staging with 4 pieces in flight measured the same: the kernel is bound by its HBM
//     mix, profiles/r04u_upconv_diag.txt),
//   * XCD k walks over a contiguous eighth of the tile sequence, its workgroups over consecutive tiles of it,
//   * an iteration = one coarse row segment (cz, cy) = 32 MFMAs, all 32 iterations of a tile unrolled; the B fragments sit in
//     a ring of 4 row sets (row = 2 planes x 3 x offsets), the row set of the next iteration is read one iteration ahead;
//     where the plane changes (cy = 7) the three new row sets go into the slots as the MFMA order (rows yy = 0, 1, 2) frees them,
//   * no branches: the face corrections are the INITIAL value of the accumulators (read from the LDS table one iteration
//     ahead, row 0 of the table = zeros), the epilogue of iteration i (pack, two v_permlane16_swap, one 16-byte buffer store
//     per row parity) sits between the first MFMAs of iteration i + 1 (two accumulator sets).
// Hazards (asm MFMAs are invisible to hipcc): an accumulator's first non-MFMA reader comes >= 5 MFMAs after its last MFMA;
// a ring slot is overwritten by a ds_read issued after the slot's last MFMA; "s_nop 1" in front of every MFMA covers a
// compiler-generated VALU write of an operand.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#ifndef UC_STORE_AUX
#define UC_STORE_AUX (DLV_NT ? 2 : 0)  // cache policy of the P stores: nt (602 -> 540 us per 16 windows, profiles/README.md)
#endif
constexpr int UM_NP = UC_CS / 64;  // staged pieces per chunk (17)
constexpr int UM_DEP = 4;          // staged pieces in flight

template <class P>
__device__ __forceinline__ void um_mfma(f32x4& acc, const u32x4& w, const u32x4& b) {
    if constexpr (P::IS_F16) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}

template <class P>
__global__ void __launch_bounds__(256, 1)
upconv2m_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk, const float* __restrict__ corr, uint4* __restrict__ out,
                int Dc, int Hc, int Wc, int tilesY, int tilesX, int tilesWin, int total, int dbg, int cstride, int c0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char uc_smem[];
    u32x4* const lds = reinterpret_cast<u32x4*>(uc_smem);                 // two halo tiles [4 chunks][UC_CS]
    float* const corr_l = reinterpret_cast<float*>(lds + 8 * UC_CS);      // [p 8][mask 8][32]
    for (int i = threadIdx.x; i < 8 * 8 * 32; i += 256) corr_l[i] = corr[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wave & 1, pz = wave >> 1;
    const int l16 = lane & 15, q = lane >> 4;
    const int voxc = Dc * Hc * Wc;
    const int H = 2 * Hc, W = 2 * Wc;
    const unsigned vox = 8u * (unsigned)voxc;

    u32x4 wf[32];
    {
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(wpk) + ((size_t)(pz * 2 + half) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) wf[i] = wsrc[(size_t)i * 64];
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(wf[i]));  // (after ALL loads are out: one latency, not 32)
    }
    // staging map of this lane: piece p = elements p*64 + lane of the chunk's halo tile: byte offset relative to the tile's
    // first halo voxel (bits 0-23) | which faces of the halo it lies on (bits 24-30; bit 30: padding of the chunk stride)
    unsigned tab[UM_NP];
#pragma unroll
    for (int p = 0; p < UM_NP; ++p) {
        const int e = p * 64 + lane;
        unsigned fl = 64u, rel = 0u;
        if (e < UC_HZ * UC_HY * UC_HX) {
            const int xh = e % UC_HX, r = e / UC_HX, yh = r % UC_HY, zh = r / UC_HY;
            rel = (unsigned)((zh * Hc + yh) * Wc + xh) * 16u;
            fl = (zh == 0 ? 1u : 0u) | (zh == UC_HZ - 1 ? 2u : 0u) | (yh == 0 ? 4u : 0u) | (yh == UC_HY - 1 ? 8u : 0u) | (xh == 0 ? 16u : 0u) |
                 (xh == UC_HX - 1 ? 32u : 0u);
        }
        tab[p] = rel | (fl << 24);
    }
    struct TileAt {
        int n, z0, y0, x0;
    };
    auto decode = [&](int t) __attribute__((always_inline)) {
        TileAt a;
        a.n = t / tilesWin;
        int r = t - a.n * tilesWin;
        a.x0 = (r % tilesX) * UC_TX;
        r /= tilesX;
        a.y0 = (r % tilesY) * UC_TY;
        a.z0 = (r / tilesY) * UC_TZ;
        return a;
    };
    // staging of tile `a` (this wave's chunk): resource over the chunk, offset of the tile's first halo voxel (may be "negative":
    // 32-bit wrap-around, the lanes it would send below the chunk are exactly the ones the face mask turns into zeros)
    auto stage_rsrc = [&](const TileAt& a) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(in + ((long long)a.n * cstride + c0 + wave) * voxc)), 0,
                                                 (dbg & 2) ? 0 : voxc * 16, 0x00020000);  // (dbg: timing-only builds of profiles/upconv_ab.py - a resource of zero records drops every access)
    };
    auto stage_base = [&](const TileAt& a) __attribute__((always_inline)) {
        return (unsigned)((((a.z0 - 1) * Hc + (a.y0 - 1)) * Wc + (a.x0 - 1)) * 16);
    };
    auto stage_mask = [&](const TileAt& a) __attribute__((always_inline)) {
        return 64u | (a.z0 == 0 ? 1u : 0u) | (a.z0 + UC_TZ == Dc ? 2u : 0u) | (a.y0 == 0 ? 4u : 0u) | (a.y0 + UC_TY == Hc ? 8u : 0u) |
               (a.x0 == 0 ? 16u : 0u) | (a.x0 + UC_TX == Wc ? 32u : 0u);
    };
    auto stage_off = [&](unsigned base, unsigned tm, int p) __attribute__((always_inline)) {
        const unsigned e = tab[p];
        unsigned off = (e & 0xffffffu) + base;
        if ((e >> 24) & tm) off = 0xfffffff0u;
        return off;
    };
    // tile walk: XCD x (= workgroups x, x + 8, ...: the dispatcher deals workgroups round-robin) owns the contiguous eighth
    // [x * total/8, (x+1) * total/8) of the tile sequence and its workgroups take consecutive tiles of it at every step - a
    // z-layer of tiles at a time for a 64^2 plane: the halo planes two tiles share are in THAT XCD's L2
    int t, tstep, tend;
    if (gridDim.x % 8 == 0 && total % 8 == 0) {
        const int per = total / 8, x = blockIdx.x & 7;
        tstep = gridDim.x / 8;
        t = x * per + (int)(blockIdx.x >> 3);
        tend = (x + 1) * per;
    } else {
        t = blockIdx.x;
        tstep = gridDim.x;
        tend = total;
    }
    if (t >= tend) return;
    TileAt cur = decode(t);
    typedef __attribute__((address_space(3))) void lds_void;
    {  // the first tile (LDS-DMA: HBM -> LDS without registers; a lane whose offset is out of range writes zeros)
        const __amdgpu_buffer_rsrc_t rs = stage_rsrc(cur);
        const unsigned base = stage_base(cur), tm = stage_mask(cur);
#pragma unroll
        for (int p = 0; p < UM_NP; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + wave * UC_CS + p * 64), 16, (int)stage_off(base, tm, p), 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned lane_out = ((unsigned)(q >> 1) * vox + (unsigned)(2 * l16 + (q & 1))) * 16u;
    const unsigned lrow = (unsigned)(q * UC_CS + l16 + pz * UC_HY * UC_HX);
    const f32x4* const cbase = reinterpret_cast<const f32x4*>(corr_l + pz * 4 * 256 + half * 16 + 4 * q);  // + (py*2+px)*256 + mask*32 floats
    int buf = 0;
    for (;;) {
        const int tn_raw = t + tstep;
        const TileAt nxt = decode(tn_raw < tend ? tn_raw : t);  // (the last tile stages itself again: no branch around the side ops)
        const __amdgpu_buffer_rsrc_t nrs = stage_rsrc(nxt);
        const unsigned nbase = stage_base(nxt), ntm = stage_mask(nxt);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(out + ((long long)cur.n * 4 + half * 2) * (long long)vox), 0, (dbg & 1) ? 0 : (int)(2u * vox * 16u), 0x00020000);
        const u32x4* const rd = lds + buf * 4 * UC_CS + lrow;                      // + ((cz + dz) * HY + row) * HX + xs
        u32x4* const wr = lds + (buf ^ 1) * 4 * UC_CS + wave * UC_CS;              // + p * 64: this wave's chunk of the next halo tile
        // face corrections of this tile: float offsets of the mask bits
        const int zb0 = (pz == 0 && cur.z0 == 0) ? 4 : 0, zb3 = (pz == 1 && cur.z0 + UC_TZ == Dc) ? 4 : 0;
        const int yb0 = cur.y0 == 0 ? 2 : 0, yb7 = cur.y0 + UC_TY == Hc ? 2 : 0;
        const int xb[2] = {(cur.x0 == 0 && l16 == 0) ? 1 : 0, (cur.x0 + UC_TX == Wc && l16 == 15) ? 1 : 0};
        auto corr_init = [&](int T, f32x4 (&a)[2][2]) __attribute__((always_inline)) {
            const int cz = T >> 3, cy = T & 7;
            const int zb = cz == 0 ? zb0 : (cz == UC_TZ - 1 ? zb3 : 0);
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int yb = (py == 0 && cy == 0) ? yb0 : ((py == 1 && cy == UC_TY - 1) ? yb7 : 0);
#pragma unroll
                for (int px = 0; px < 2; ++px) a[py][px] = cbase[((py * 2 + px) * 256 + (zb + yb + xb[px]) * 32) / 4];
            }
        };
        u32x4 ring[4][2][3];
        auto ring_read = [&](int cz, int row) __attribute__((always_inline)) {
            const int s = (2 * cz + row) & 3;
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int xs = 0; xs < 3; ++xs) ring[s][dz][xs] = rd[((cz + dz) * UC_HY + row) * UC_HX + xs];
        };
        f32x4 acc[2][2][2];
        uint2 pk[2];
        // epilogue of iteration T, row parity py, in two pieces: pack + interleave, store
        auto epi_pack = [&](int T, int py) __attribute__((always_inline)) {
            f32x4(&a)[2][2] = acc[T & 1];
            asm volatile("" : "+v"(a[py][0]), "+v"(a[py][1]));  // (the first read of the finished accumulators stays HERE)
            const unsigned a0 = P::pack2(a[py][0][0], a[py][0][1]), a1 = P::pack2(a[py][0][2], a[py][0][3]);  // fine x = 2n
            const unsigned b0 = P::pack2(a[py][1][0], a[py][1][1]), b1 = P::pack2(a[py][1][2], a[py][1][3]);  // fine x = 2n + 1
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            pk[0] = make_uint2(s0[0], s1[0]);
            pk[1] = make_uint2(s0[1], s1[1]);
        };
        auto epi_store = [&](int T, int py) __attribute__((always_inline)) {
            const int cz = T >> 3, cy = T & 7;
            const int fz = 2 * (cur.z0 + cz) + pz, fy = 2 * (cur.y0 + cy) + py;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk[0].x, pk[0].y, pk[1].x, pk[1].y}, ors, (int)lane_out,
                                                   (int)((unsigned)((fz * H + fy) * W + 2 * cur.x0) * 16u), UC_STORE_AUX);
        };
        ring_read(0, 0);
        ring_read(0, 1);
        ring_read(0, 2);
        corr_init(0, acc[0]);
#pragma unroll
        for (int T = 0; T < UC_TZ * UC_TY; ++T) {
            const int cz = T >> 3, cy = T & 7;
            const bool last = T == UC_TZ * UC_TY - 1;
            // the row set of the next iteration
            if (cy < UC_TY - 1) ring_read(cz, cy + 3);
            else if (!last) ring_read(cz + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int g = m >> 3, k = m & 7;
                const int py = g >> 1, dy = g & 1;  // input rows in the order yy = py + dy = 0, 1, 1, 2: a slot is free as soon as its row is done
                const int dz = k >> 2, dx = (k >> 1) & 1, px = k & 1;
                um_mfma<P>(acc[T & 1][py][px], wf[(py * 2 + px) * 8 + (dz * 2 + dy) * 2 + dx], ring[(2 * cz + cy + py + dy) & 3][dz][px + dx]);
                // ---- side work ----
                bool side = true;
                if (m == 1 && T > 0) epi_pack(T - 1, 0);
                else if (m == 3 && T > 0) epi_store(T - 1, 0);
                else if (m == 5 && T > 0) epi_pack(T - 1, 1);
                else if (m == 7 && T > 0) epi_store(T - 1, 1);
                else if (m == 8 && cy == UC_TY - 1 && !last) ring_read(cz + 1, 1);
                else if (m == 10 && !last) corr_init(T + 1, acc[(T + 1) & 1]);
                else if (m == 12 && T < UM_NP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(nrs, (lds_void*)(wr + T * 64), 16, (int)stage_off(nbase, ntm, T), 0, 0, 0);
                else if (m == 24 && cy == UC_TY - 1 && !last) ring_read(cz + 1, 2);
                else side = false;
                if (side) __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7");  // the last MFMAs' results
        epi_pack(UC_TZ * UC_TY - 1, 0);
        epi_store(UC_TZ * UC_TY - 1, 0);
        epi_pack(UC_TZ * UC_TY - 1, 1);
        epi_store(UC_TZ * UC_TY - 1, 1);
        t = tn_raw;
        if (t >= tend) break;
        // the next halo tile is complete (the 17 DMA loads are older than the 30 stores of the iterations behind them: a counted
        // wait - vmcnt(0) would drain the store queue once per tile), nobody reads this one any more
        asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = nxt;
        buf ^= 1;
    }
}

}  // namespace

int dlv_pack_upconv(dlv_ctx* ctx, bool f16, const float* wc, int ctot, int cs, const float* wd, const float* bd, uint16_t* wpk, float* corr, int ci0,
                    int with_corr, float wscale) {
    if (f16)
        hipLaunchKernelGGL(pack_upconv_kernel<PF16>, dim3(64), dim3(256), 0, ctx->stream, wc, ctot, cs, wd, bd, wpk, corr, ci0, with_corr, wscale);
    else
        hipLaunchKernelGGL(pack_upconv_kernel<PBf16>, dim3(64), dim3(256), 0, ctx->stream, wc, ctot, cs, wd, bd, wpk, corr, ci0, with_corr, wscale);
    DLV_LAUNCH_CHECK(ctx, "pack_upconv_kernel");
    return DLV_OK;
}

// does the persistent kernel take this shape?  Full tiles only; 24-bit halo offsets, 31-bit offsets within a wave's two output chunks
bool dlv_upconv2_persistent(const dlv_ctx* ctx, int Dc, int Hc, int Wc) {
    const long long voxc = (long long)Dc * Hc * Wc;
    return !ctx->upconv_simple && Dc > 0 && Dc % UC_TZ == 0 && Hc % UC_TY == 0 && Wc % UC_TX == 0 &&
           ((long long)(UC_HZ - 1) * Hc + UC_HY) * Wc * 16 < (1 << 24) && voxc * 8 * 16 * 2 < (1LL << 31);
}

// in: activated coarse tensor (B, 8 * cstride ch, Dc, Hc, Wc) chunk-planar, of which the four chunks from c0 are this launch's 32
// input channels; out: P (B, 32 ch, 2Dc, 2Hc, 2Wc)
int dlv_upconv2_launch(dlv_ctx* ctx, bool f16, const void* in, const void* wpk, const float* corr, void* out, int B, int Dc, int Hc, int Wc,
                       int cstride, int c0) {
    if (Wc % 2 || Dc <= 0 || Hc <= 0 || Wc <= 0) return dlv_fail(ctx, DLV_EUNSUP, "upconv: even coarse width expected");
    const int tilesX = dlv_cdiv(Wc, UC_TX), tilesY = dlv_cdiv(Hc, UC_TY), tilesZ = dlv_cdiv(Dc, UC_TZ);
    if (dlv_upconv2_persistent(ctx, Dc, Hc, Wc)) {
        static int ncu = 0;
        if (!ncu) DLV_HIP(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
        int cus = ncu;
        const int tilesWin = tilesX * tilesY * tilesZ, total = tilesWin * B;
        int grid = dlv_cdiv(total, dlv_cdiv(total, cus));  // every workgroup walks over the same number of tiles (+-1)
        if (total % 8 == 0) grid = std::min(cus / 8 * 8, dlv_cdiv(grid, 8) * 8);  // the same number of workgroups on every XCD
        const size_t lds = (size_t)8 * UC_CS * 16 + 8 * 8 * 32 * 4;
        static dlv_attr_bits mattr_f16{0}, mattr_bf16{0};
        if (f16) {
            if (!dlv_attr_is_set(mattr_f16, ctx->device)) {
                DLV_HIP(ctx, hipFuncSetAttribute((const void*)upconv2m_kernel<PF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                dlv_attr_mark(mattr_f16, ctx->device);
            }
            hipLaunchKernelGGL(upconv2m_kernel<PF16>, dim3(grid), dim3(256), lds, ctx->stream, (const uint4*)in, (const uint4*)wpk, corr, (uint4*)out,
                               Dc, Hc, Wc, tilesY, tilesX, tilesWin, total, ctx->upconv_dbg, cstride, c0);
        } else {
            if (!dlv_attr_is_set(mattr_bf16, ctx->device)) {
                DLV_HIP(ctx, hipFuncSetAttribute((const void*)upconv2m_kernel<PBf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                dlv_attr_mark(mattr_bf16, ctx->device);
            }
            hipLaunchKernelGGL(upconv2m_kernel<PBf16>, dim3(grid), dim3(256), lds, ctx->stream, (const uint4*)in, (const uint4*)wpk, corr, (uint4*)out,
                               Dc, Hc, Wc, tilesY, tilesX, tilesWin, total, ctx->upconv_dbg, cstride, c0);
        }
        DLV_LAUNCH_CHECK(ctx, "upconv2m_kernel");
        return DLV_OK;
    }
    const size_t lds = (size_t)4 * UC_CS * 16 + 8 * 8 * 32 * 4;
    static dlv_attr_bits attr_f16{0}, attr_bf16{0};
    if (f16) {
        if (!dlv_attr_is_set(attr_f16, ctx->device)) {
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)upconv2_kernel<PF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            dlv_attr_mark(attr_f16, ctx->device);
        }
        hipLaunchKernelGGL(upconv2_kernel<PF16>, dim3(tilesX * tilesY * tilesZ, 1, B), dim3(256), lds, ctx->stream, (const uint4*)in,
                           (const uint4*)wpk, corr, (uint4*)out, Dc, Hc, Wc, tilesY, tilesX, cstride, c0);
    } else {
        if (!dlv_attr_is_set(attr_bf16, ctx->device)) {
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)upconv2_kernel<PBf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            dlv_attr_mark(attr_bf16, ctx->device);
        }
        hipLaunchKernelGGL(upconv2_kernel<PBf16>, dim3(tilesX * tilesY * tilesZ, 1, B), dim3(256), lds, ctx->stream, (const uint4*)in,
                           (const uint4*)wpk, corr, (uint4*)out, Dc, Hc, Wc, tilesY, tilesX, cstride, c0);
    }
    DLV_LAUNCH_CHECK(ctx, "upconv2_kernel");
    return DLV_OK;
}
