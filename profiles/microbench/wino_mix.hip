// Gate B of the Winograd z-reg convolution: the instruction mix of one z step of the
// register-resident-weights 3x3x3 conv (Cin = 32 -> 32 of 64 output channels halves, 4 waves = 2 channel halves x 2 row
// groups, one wave per SIMD), registers + LDS only, random data, on all 256 CUs - the shipped DIRECT mix against the
// Winograd F(2,3)-along-x mix, in output voxels per second under the chip's power cap.
//
//   direct (TYT 16, what conv3_zreg_kernel<PF16,32,16> runs per z step and wave): 10 row groups x 6 ds_read_b128,
//       432 MFMA (27 taps x 8 rows x 2 column blocks), per output row 2 x (4 accumulator reads where the set lives in
//       AGPRs + 4 add + 4 fmac + 2 cvt_pk + 1 8-byte store), 10 staging writes; 512 voxels x 32 couts per workgroup step
//   wino_x (TYT 8): 6 row groups x (4 ds_read_b128 of the even/odd-split halo row + 16 v_pk_add_f16 = the input transform
//       V = B^T d), 144 MFMA (4 positions x 9 (ky,kz) x 4 rows), per output row: 16 accumulator reads (AGPR sets), the
//       output transform Y = A^T M (16 fp32 adds), 8 add + 8 fmac statistics, 4 cvt_pk, 2 v_permlane16_swap, ONE 16-byte
//       store, 6 staging writes; 256 voxels x 32 couts per workgroup step
//
// The global loads/stores of the real kernel are replaced by LDS writes of the same width (the judge's definition of the
// gate: registers/LDS only).  Pass: wino_x >= 1.35 x direct in voxels/s.
//     hipcc -O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=4000000 wino_mix.hip -o wino_mix && ./wino_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int N>
using IC = std::integral_constant<int, N>;

constexpr int HX = 34;

template <bool AG, bool FIRST>
__device__ __forceinline__ void mfma(f32x4& acc, const u32x4& w, const u32x4& b) {
    if constexpr (AG) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc) : "a"(w), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(b));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
    }
}

__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (h2)(__builtin_bit_cast(h2, a) + __builtin_bit_cast(h2, b)));
}
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (h2)(__builtin_bit_cast(h2, a) - __builtin_bit_cast(h2, b)));
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}

// WINO false: RW output rows per wave, NB = 2 column blocks of 16 voxels, 27 weight fragments
// WINO true : RW output rows per wave, NB = 4 transform positions over 16 tiles of 2 voxels, 36 weight fragments
// ABL (timing-only ablations, wrong numbers): 1 asm output transform (not an ablation), 2 no input transform, 4 no output
// transform, 8 no statistics, 16 no pack / swap / store, 32 no staging, 64 no fragment reads, 128 no barrier
template <bool WINO, int RW, int RA, int ABL>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, long long* clk) {
    constexpr bool XASM = ABL & 1;
    constexpr int TYT = 2 * RW, HY = TYT + 2, PL = HY * HX, CS = ((PL + 15) / 16) * 16, BUF = 4 * CS;
    constexpr int NB = WINO ? 4 : 2, NW = WINO ? 36 : 27, NG = RW + 2, RV = RW - RA;
    constexpr int NPIECE = (PL + 63) / 64;  // staging writes per wave and step (one chunk of the plane)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);  // 2 plane buffers + an "output" scratch line per wave
    u32x4* scratch = lds + 2 * BUF;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rg = wave >> 1, l16 = lane & 15, q = lane >> 4;
    auto rnd16 = [&](unsigned i) {
        unsigned h = i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 13;
        h *= 0x5bd1e995u;
        h ^= h >> 15;
        return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f)));
    };
    for (int i = threadIdx.x; i < 2 * BUF; i += 256) {
        u32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = rnd16(i * 8 + 2 * e) | (rnd16(i * 8 + 2 * e + 1) << 16);
        lds[i] = v;
    }
    u32x4 wf[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        u32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = rnd16(7777 + (i * 64 + lane) * 8 + 2 * e) | (rnd16(9999 + (i * 64 + lane) * 8 + 2 * e) << 16);
        wf[i] = v;
        asm volatile("" : "+a"(wf[i]));
    }
    __syncthreads();
    f32x4 accv[3][RV > 0 ? RV : 1][NB], acca[3][RA > 0 ? RA : 1][NB];
    const f32x4 fzero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < RV; ++r) accv[s][r][b] = fzero;
#pragma unroll
            for (int r = 0; r < RA; ++r) acca[s][r][b] = fzero;
        }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    // direct: lane reads voxel (row, b*16 + l16 + kx) of chunk q;  wino: halo row split into even / odd voxels (17 each):
    // d0 = E[n], d1 = O[n], d2 = E[n+1], d3 = O[n+1] for tile n = l16
    const unsigned lbase = (unsigned)(q * CS + rg * RW * HX + l16);
    u32x4 pre[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) pre[i] = lds[wave * CS + i * 64 + lane];
    const unsigned wbase = (unsigned)(wave * CS + lane);
    const unsigned sbase = (unsigned)(wave * 64 + lane);

    f32x4 epi_m[NB];
    f32x4 y0 = fzero, y1 = fzero;
    auto acc_ref = [&](int set, int r, int b) -> f32x4& { return r < RA ? acca[set][r < RA ? r : 0][b] : accv[set][r >= RA ? r - RA : 0][b]; };
    // epilogue micro-ops of one finished output row
    constexpr int EPI_OPS = WINO ? 10 : 18;
    auto epi = [&](int set, int r, int k) __attribute__((always_inline)) {
        if constexpr (WINO) {
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
            } else if (k <= 4) {  // output transform of element e: Y0 = M0 + M1 + M2, Y1 = M1 - M2 - M3
                const int e = k - 1;
                if constexpr (ABL & 4) {
                    y0[e] = epi_m[0][e];
                    y1[e] = epi_m[1][e];
                } else if constexpr (XASM) {
                    float a, b2;
                    asm volatile("v_add_f32 %0, %2, %3\n\tv_sub_f32 %1, %3, %4\n\tv_add_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %5"
                                 : "=&v"(a), "=&v"(b2) : "v"(epi_m[0][e]), "v"(epi_m[1][e]), "v"(epi_m[2][e]), "v"(epi_m[3][e]));
                    y0[e] = a;
                    y1[e] = b2;
                } else {
                    y0[e] = (epi_m[0][e] + epi_m[1][e]) + epi_m[2][e];
                    y1[e] = (epi_m[1][e] - epi_m[2][e]) - epi_m[3][e];
                }
            } else if (k <= 8) {
                const int e = k - 5;
                if constexpr (!(ABL & 8)) asm volatile("v_add_f32 %0, %2, %0\n\tv_fmac_f32 %1, %2, %2\n\tv_add_f32 %0, %3, %0\n\tv_fmac_f32 %1, %3, %3"
                             : "+v"(ssum[e]), "+v"(ssq[e]) : "v"(y0[e]), "v"(y1[e]));
            } else if constexpr (ABL & 16) {
                asm volatile("" ::"v"(y0), "v"(y1));
            } else {
                unsigned a0 = pack2(y0[0], y0[1]), a1 = pack2(y0[2], y0[3]);  // voxel 2n, couts 4q..4q+3
                unsigned b0 = pack2(y1[0], y1[1]), b1 = pack2(y1[2], y1[3]);  // voxel 2n+1
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                // lane (n, q): voxel 2n + (q & 1), couts 8 (q >> 1) .. + 7: 16 bytes
                scratch[sbase] = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        } else {
            const int b = k / 9, kk = k % 9;
            if (kk == 0) {
                if (r < RA) {
                    asm volatile("" : "+a"(acca[set][r < RA ? r : 0][b]));
                    epi_m[b] = acca[set][r < RA ? r : 0][b];
                    asm volatile("" : "+v"(epi_m[b]));
                } else {
                    asm volatile("" : "+v"(accv[set][r >= RA ? r - RA : 0][b]));
                    epi_m[b] = accv[set][r >= RA ? r - RA : 0][b];
                }
            }
            if (kk < 4) {
                if constexpr (!(ABL & 8)) asm volatile("v_add_f32 %0, %1, %0" : "+v"(ssum[kk]) : "v"(epi_m[b][kk]));
            } else if (kk < 8) {
                if constexpr (!(ABL & 8)) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(ssq[kk - 4]) : "v"(epi_m[b][kk - 4]));
            } else {
                uint2 u;
                u.x = pack2(epi_m[b][0], epi_m[b][1]);
                u.y = pack2(epi_m[b][2], epi_m[b][3]);
                if constexpr (!(ABL & 16)) reinterpret_cast<uint2*>(scratch)[sbase * 2 + b] = u;
                else asm volatile("" ::"v"(u.x), "v"(u.y));
            }
        }
    };

    auto step = [&](int p, auto SA_, auto SB_, auto SC_) __attribute__((always_inline)) {
        constexpr int SA = decltype(SA_)::value, SB = decltype(SB_)::value, SC = decltype(SC_)::value;
        const int rb = (p & 1) * BUF, wb = ((p + 1) & 1) * BUF;
        constexpr int NF = WINO ? 4 : 6;
        u32x4 fb[2][NF], raw[4];
        auto load_frag = [&](int g, int i) __attribute__((always_inline)) {
            if constexpr (WINO) raw[i] = lds[lbase + rb + g * HX + (i & 1) * 17 + (i >> 1)];
            else fb[g & 1][i] = lds[lbase + rb + g * HX + (i % 2) * 16 + i / 2];
        };
        auto xform = [&](int g, int e) __attribute__((always_inline)) {  // dword e of the four positions
            if constexpr (ABL & 2) {
                for (int i = 0; i < 4; ++i) fb[g & 1][i][e] = raw[i][e];
                return;
            }
            fb[g & 1][0][e] = pk_sub(raw[0][e], raw[2][e]);
            fb[g & 1][1][e] = pk_add(raw[1][e], raw[2][e]);
            fb[g & 1][2][e] = pk_sub(raw[2][e], raw[1][e]);
            fb[g & 1][3][e] = pk_sub(raw[1][e], raw[3][e]);
        };
#pragma unroll
        for (int i = 0; i < NF; ++i) load_frag(0, i);
        if constexpr (WINO) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xform(0, e);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = g;
            const int n_frag = (g + 1 < NG) ? NF : 0;
            const int n_x = (WINO && g + 1 < NG) ? 4 : 0;
            const int piece = g < NPIECE ? g : -1;
            const int n_piece = piece >= 0 ? 2 : 0;
            const int erow = (g == 0) ? RW - 1 : (j >= 3 ? j - 3 : -1);
            const int n_epi = erow >= 0 ? EPI_OPS : 0;
            const int n_side = n_frag + n_epi + n_x + n_piece;
            auto side = [&](int kk) __attribute__((always_inline)) {
                if (kk < n_frag) {
                    if constexpr (!(ABL & 64)) load_frag(g + 1, kk);
                } else if (kk < n_frag + n_epi) {
                    if (g == 0) epi(SC, RW - 1, kk - n_frag);
                    else epi(SA, erow, kk - n_frag);
                } else if (kk < n_frag + n_epi + n_x) {
                    xform(g + 1, kk - n_frag - n_epi);
                } else if constexpr (ABL & 32) {
                } else if (kk == n_frag + n_epi + n_x) {
                    lds[wbase + wb + piece * 64] = pre[piece];
                } else {
                    pre[piece] = lds[wbase + rb + piece * 64];  // stands in for the global load
                }
            };
            int n_mfma = 0;
#pragma unroll
            for (int ky = 2; ky >= 0; --ky)
                if (j - ky >= 0 && j - ky < RW) n_mfma += WINO ? 12 : 18;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 2; ky >= 0; --ky) {
                const int r = j - ky;
                if (r < 0 || r >= RW) continue;
                int vr = 0;
#pragma unroll
                for (int k2 = 2; k2 > ky; --k2)
                    if (j - k2 >= 0 && j - k2 < RW) ++vr;
                constexpr int NX = WINO ? 4 : 3, NBB = WINO ? 1 : 2;
#pragma unroll
                for (int kx = 0; kx < NX; ++kx)
#pragma unroll
                    for (int b = 0; b < NBB; ++b) {
                        const u32x4 bf = fb[g & 1][WINO ? kx : kx * 2 + b];
                        const int ab = WINO ? kx : b;  // accumulator: transform position / column block
#pragma unroll
                        for (int kz = 0; kz < 3; ++kz) {
                            const u32x4 w = wf[(kz * 3 + ky) * NX + kx];
                            const int set = kz == 0 ? SC : (kz == 1 ? SB : SA);
                            const bool first = WINO ? (kz == 0 && ky == 0) : (kz == 0 && ky == 0 && kx == 0);
                            if (r < RA) {
                                if (first) mfma<true, true>(acca[set][r < RA ? r : 0][ab], w, bf);
                                else mfma<true, false>(acca[set][r < RA ? r : 0][ab], w, bf);
                            } else {
                                if (first) mfma<false, true>(accv[set][r >= RA ? r - RA : 0][ab], w, bf);
                                else mfma<false, false>(accv[set][r >= RA ? r - RA : 0][ab], w, bf);
                            }
                            const int m = ((vr * NX + kx) * NBB + b) * 3 + kz;
                            const int lo = m * n_side / n_mfma, hi = (m + 1) * n_side / n_mfma;
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (lo + t < hi) side(lo + t);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if constexpr (!(ABL & 128)) __builtin_amdgcn_s_barrier();
    };
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int r = 0; r < RA; ++r) asm volatile("" : "+a"(acca[s][r][b]));
#pragma unroll
                for (int r = 0; r < RV; ++r) asm volatile("" : "+v"(accv[s][r][b]));
            }
        asm volatile("s_nop 3");
        step(3 * it + 0, IC<0>{}, IC<1>{}, IC<2>{});
        step(3 * it + 1, IC<1>{}, IC<2>{}, IC<0>{});
        step(3 * it + 2, IC<2>{}, IC<0>{}, IC<1>{});
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = c1 - c0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += ssum[e] + ssq[e];
    out[blockIdx.x * 256 + threadIdx.x] = s + __builtin_bit_cast(float, scratch[sbase][0]);
}

template <bool WINO, int RW, int RA, int ABL>
double run(const char* name, int grid, int iters) {
    constexpr int TYT = 2 * RW, PL = (TYT + 2) * HX, CS = ((PL + 15) / 16) * 16, BUF = 4 * CS;
    const size_t ldsb = (size_t)(2 * BUF + 256) * 16;
    hipFuncSetAttribute((const void*)k<WINO, RW, RA, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    float* out;
    long long* clk;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&clk, (size_t)grid * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double rate = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WINO, RW, RA, ABL>), dim3(grid), dim3(256), ldsb, 0, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double vox = (double)grid * iters * 3 * TYT * 32;  // output voxels (x 32 couts) per launch
        const double mf = (double)grid * 4 * iters * 3 * RW * (WINO ? 36 : 54) * 16384.0;
        rate = vox / (ms * 1e-3);
        if (rep == 3) {
            long long h[2];
            hipMemcpy(h, clk + (grid / 2) * 2, 16, hipMemcpyDeviceToHost);
            const double cyc_step = (double)h[0] / (iters * 3.0), ghz = (double)h[0] / ((double)h[1] * 10.0);  // realtime: 100 MHz
            printf("%-26s grid %4d %8.2f ms %6.2f Gvoxel/s %7.1f MFMA-TFLOP/s (direct-equivalent %7.1f)  %6.0f cyc/step = %5.2f cyc/voxel @ %.2f GHz\n",
                   name, grid, ms, rate / 1e9, mf / (ms * 1e-3) / 1e12, rate * 2.0 * 27 * 32 * 32 / 1e12, cyc_step,
                   cyc_step / (TYT * 32), ghz);
        }
    }
    if (hipGetLastError() != hipSuccess) printf("HIP error\n");
    hipFree(out);
    return rate;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int grid = 256, it = iters;
    const double d16 = run<false, 8, 4, 1>("direct TYT16 (shipped)", grid, it / 2);
    const double d8 = run<false, 4, 2, 1>("direct TYT8", grid, it);
    const double w2a = run<true, 4, 2, 1>("wino_x RA2 asm-xf", grid, it);
    const double w1c = run<true, 4, 1, 0>("wino_x RA1 C-xf", grid, it);
    printf("  wino_x / direct TYT16: %.3f (RA2 asm)  %.3f (RA1 C);  direct TYT8 / TYT16 = %.3f\n", w2a / d16, w1c / d16, d8 / d16);
    printf("timing-only ablations of wino_x RA1 (wrong numbers):\n");
    run<true, 4, 1, 2>("  - input transform", grid, it);
    run<true, 4, 1, 4>("  - output transform", grid, it);
    run<true, 4, 1, 8>("  - statistics", grid, it);
    run<true, 4, 1, 16>("  - pack/swap/store", grid, it);
    run<true, 4, 1, 32>("  - staging", grid, it);
    run<true, 4, 1, 64>("  - fragment reads", grid, it);
    run<true, 4, 1, 128>("  - barrier", grid, it);
    run<true, 4, 1, 2 + 4 + 8 + 16>("  - all VALU side work", grid, it);
    run<true, 4, 1, 2 + 4 + 8 + 16 + 32 + 64>("  MFMA only (+barrier)", grid, it);
    run<false, 8, 4, 1 + 8 + 16 + 32 + 64>("direct TYT16 MFMA only", grid, it / 2);
    return 0;
}
