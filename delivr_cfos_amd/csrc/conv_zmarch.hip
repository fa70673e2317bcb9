// conv_zmarch.hip - the hot 3x3x3 convolutions of the U-Net (Cout = 32, Cin = 32 or 32+32 at levels
// 0 and 1: 91 % of the network's FLOPs) as a z-marching, input-stationary implicit GEMM on
// v_mfma_f32_32x32x16_bf16.
//
// One workgroup owns an (8 rows x 32 columns) in-plane tile of one sample and marches along z
// ("z-major slabs"): for every input plane p it
//     - has the halo plane (10 x 34 voxels x Cin channels) in LDS, staged through registers from the
//       coalesced chunk-planar tensor while the previous plane is being computed (issue-early /
//       write-late),
//     - multiplies it with ALL 27 taps: the kz = 0/1/2 slices of the weights feed three rotating
//       accumulators (output planes p+1, p, p-1), so each input fragment read from LDS is used by
//       three MFMAs and only ONE plane has to be resident,
//     - emits output plane p-1 (bias, bf16 store, InstanceNorm partial sums kept per lane and
//       reduced once per column).
// The complete weight set (27 x Cin x 32 bf16 = 54 / 108 KiB) is LDS-resident in MFMA A-fragment
// order.  LDS reads per MFMA: 0.83 x ds_read_b128 (weights shared by the wave's two voxel blocks).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "prec16.h"

namespace {

#define AS_FRAG(x) (x)

constexpr int ZM_TY = 8, ZM_TX = 32, ZM_HY = 10, ZM_HX = 34;
constexpr int ZM_PLANE = ZM_HY * ZM_HX;  // 340 voxels

template <int CIN, int TYT>
struct ZmCfg {
    static constexpr int HY = TYT + 2;
    static constexpr int PLANE = HY * ZM_HX;     // halo plane voxels
    static constexpr int C8 = CIN / 8;           // chunks
    static constexpr int KP = CIN / 16;          // k-steps per tap
    static constexpr int WELEMS = 27 * KP * 64;  // uint4 elements of weights in LDS
    static constexpr int PELEMS = C8 * PLANE; // uint4 elements of one halo plane
    static constexpr size_t LDS_BYTES = (size_t)(WELEMS + PELEMS) * 16 + 2048;  // + 8x64 floats for the stats flush
};

template <class P, int CIN, int VB, int MINW, int TYT, bool PIN, int DIST, int ABL = 0, bool STAG = false>
__global__ void __launch_bounds__(64 * TYT / VB, MINW) conv3_zmarch_kernel(const uint4* __restrict__ in1, int c1_8,
                                                           const uint4* __restrict__ in2, int c2_8,
                                                           const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                           uint4* __restrict__ out, float* __restrict__ partials, int D,
                                                           int H, int W, int tilesY, int tilesX, int zseg,
                                                           unsigned long long* __restrict__ stamps, int nseg, int cout8) {
    using C = ZmCfg<CIN, TYT>;
    constexpr int NT = 64 * TYT / VB;            // threads: TYT rows / VB rows per wave
    constexpr int NPRE = (C::PELEMS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* lds_w = reinterpret_cast<uint4*>(smem_raw);
    uint4* lds_p = lds_w + C::WELEMS;
    float* red = reinterpret_cast<float*>(lds_p + C::PELEMS);  // [4 waves][32][2]

    const int n = blockIdx.z;
    // blockIdx.y = z segment + nseg * output-channel block (32 channels each; Cout = 64 layers run two blocks
    // over the same input)
    const int seg = blockIdx.y % nseg, cb = blockIdx.y / nseg;
    const int tile = dlv_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int y0 = ty * TYT, x0 = tx * ZM_TX;
    const int zs = seg * zseg, ze = min(zs + zseg, D);  // output planes [zs, ze)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long plane = (long long)H * W;
    const long long vox = (long long)D * plane;

    // ---- weights -> LDS (A-fragment order, lane-linear) ------------------------------------------------
    for (int i = threadIdx.x; i < C::WELEMS; i += NT) lds_w[i] = wpk[(size_t)cb * C::WELEMS + i];

    // ---- per-thread staging map of the halo plane (constant along z) ----------------------------------
    long long goff[NPRE];
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int i = threadIdx.x + NT * j;
        goff[j] = -1;
        if (i < C::PELEMS) {
            const int xh = i % ZM_HX, yh = (i / ZM_HX) % C::HY, c = i / C::PLANE;
            const int gy = y0 + yh - 1, gx = x0 + xh - 1;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const long long base = c < c1_8 ? ((long long)n * c1_8 + c) * vox : ((long long)n * c2_8 + (c - c1_8)) * vox;
                goff[j] = base + (long long)gy * W + gx;
                valid |= 1u << j;
            }
        }
    }
    auto src_of = [&](int j) -> const uint4* {
        const int c = (threadIdx.x + NT * j) / C::PLANE;
        return c < c1_8 ? in1 : in2;
    };
    uint4 pre[NPRE];
    // loads are unconditional (invalid lanes read element 0 of their source and are zeroed by a select):
    // no exec-masked branch per load, all of a plane's loads are in flight together
    auto issue_loads = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const bool ok = (valid >> j) & 1u;
            const uint4* src = ok ? src_of(j) : in1;  // lanes without an element read (and discard) in1[0]
            pre[j] = src[ok ? goff[j] + (long long)p * plane : 0];
        }
    };
    // the zero-select of out-of-window lanes happens here, at the first use of the loaded registers, so
    // that the wait for the loads sits in front of the LDS write and not in front of the MFMAs
    auto write_plane = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = threadIdx.x + NT * j;
            const bool ok = (valid >> j) & 1u;
            if (i < C::PELEMS) lds_p[i] = ok ? pre[j] : make_uint4(0, 0, 0, 0);
        }
    };

    // per-lane LDS offsets: voxel block v = row 2*wave + v, column col; chunk half h
    int lb[VB];
#pragma unroll
    for (int v = 0; v < VB; ++v) lb[v] = (h * C::HY + (VB * wave + v)) * ZM_HX + col;

    f32x16 fzero;
#pragma unroll
    for (int r = 0; r < 16; ++r) fzero[r] = 0.f;
    f32x16 a0[VB], a1[VB], a2[VB];
#pragma unroll
    for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[v][r] = a1[v][r] = a2[v][r] = 0.f;
    float bs[16], ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bs[r] = bias[cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        ssum[r] = ssq[r] = 0.f;
    }

    // InstanceNorm partial sums are flushed every 16 output planes (absolute z / 16), so that their
    // grouping - and with it every bit of the statistics - does not depend on zseg or the batch size
    const int nzc = (D + 15) / 16;
    auto flush_stats = [&](int zc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = ssum[r], b = ssq[r];
            a = dlv_half_sum32(a);  // DPP adds; totals valid in lanes 16-31 / 48-63
            b = dlv_half_sum32(b);
            if (col == 31) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                red[(wave * 32 + co) * 2] = a;
                red[(wave * 32 + co) * 2 + 1] = b;
            }
            ssum[r] = ssq[r] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int i = threadIdx.x;
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < TYT / VB; ++w8) v += red[w8 * 64 + i];
            const long long nparts = (long long)gridDim.x * nzc;
            const long long part = (long long)zc * gridDim.x + tile;
            partials[(((long long)n * nparts + part) * (cout8 * 8) + cb * 32 + (i >> 1)) * 2 + (i & 1)] = v;
        }
        __syncthreads();
    };

    // STAG: the second wave of every SIMD (waves NW/2..NW-1) handles its epilogue one step late, at the start of
    // the next step, so that on each SIMD one wave's VALU/store work overlaps the other wave's MFMAs
    const bool late_wave = STAG && (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= NT / 2);
    auto epilogue = [&](f32x16(&acc)[VB], int oz) __attribute__((always_inline)) {
        const bool emit = !(ABL & 1) && oz >= zs && oz < ze;
        if (ABL & 1) {
#pragma unroll
            for (int v = 0; v < VB; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[v][r]));
        }
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            const int oy = y0 + VB * wave + v, ox = x0 + col;
            const bool ok = emit && oy < H && ox < W;
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                val[r] = acc[v][r] + bs[r];
                if (ok) {
                    ssum[r] += val[r];
                    ssq[r] = fmaf(val[r], val[r], ssq[r]);
                }
            }
            if (ok) {
                const long long o = (long long)oz * plane + (long long)oy * W + ox;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = P::pack2(val[4 * g + 0], val[4 * g + 1]);
                    u.y = P::pack2(val[4 * g + 2], val[4 * g + 3]);
                    uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * cout8 + cb * 4 + g) * vox + o);
                    if (ABL & 8) {  // streaming store: the output is not re-read by this kernel
                        __builtin_nontemporal_store(u.x, &dst[h].x);
                        __builtin_nontemporal_store(u.y, &dst[h].y);
                    } else {
                        dst[h] = u;
                    }
                }
            }
        }
    };

    // diagnostic build only (ABL bit 4): s_memtime stamps of the phases of every step, written to a buffer of
    // their own that nothing else reads (profiles/zm_timeline.py); no stamp executes in the product kernels
    constexpr bool STAMP = (ABL & 16) != 0;
    const bool stamp_on = STAMP && n == 0 && seg == 0 && stamps != nullptr && lane == 0;
    auto stamp = [&](int p, int slot) __attribute__((always_inline)) {
        if (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (stamp_on) stamps[(((long long)tile * (NT / 64) + wave) * (D + 4) + (p + 1)) * 8 + slot] = t;
        }
    };
    // one z step: plane p is in LDS (when 0 <= p < D).  kz=2 -> accA (out[p-1]), kz=1 -> accB (out[p]),
    // kz=0 -> accC (out[p+1], started here).  Then out[p-1] is emitted from accA.
    auto step = [&](int p, f32x16(&accA)[VB], f32x16(&accB)[VB], f32x16(&accC)[VB]) __attribute__((always_inline)) {
        const bool next_needed = !(ABL & 2) && (p + 1 <= ze) && (p + 1 >= 0) && (p + 1 < D);
        stamp(p, 0);
        if (STAMP && stamp_on) stamps[(((long long)tile * (NT / 64) + wave) * (D + 4) + (p + 1)) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        if (next_needed) issue_loads(p + 1);
        stamp(p, 1);
        if (STAG) {
            if (late_wave) epilogue(accC, p - 2);  // accC still holds the plane finished one step ago
            // every wave has now added exactly the planes <= p-2: the only cut at which a statistics chunk is
            // complete for both wave groups (keeps the partial sums independent of zseg / batch size)
            const int ozf = p - 2;
            if (!(ABL & 1) && ozf >= zs && ozf < ze && ((ozf & 15) == 15 || ozf == ze - 1)) flush_stats(ozf >> 4);
        }
        if (p >= 0 && p < D && p <= ze) {
            // software-pipelined over the 9*KP (ky,kx,ks) groups: the 5 LDS fragment reads of group g+1
            // are issued before the 6 MFMAs of group g (one wave per SIMD: nothing else hides LDS latency)
            constexpr int NG = 9 * C::KP;
            uint4 fb[DIST + 1][VB], fw[DIST + 1][3];
            auto load_group = [&](int g, uint4(&b)[VB], uint4(&w)[3]) __attribute__((always_inline)) {
                const int ks = g % C::KP, kx = (g / C::KP) % 3, ky = g / (3 * C::KP);
#pragma unroll
                for (int v = 0; v < VB; ++v) b[v] = lds_p[lb[v] + (ks * 2 * C::HY + ky) * ZM_HX + kx];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) w[kz] = lds_w[(((kz * 3 + ky) * 3 + kx) * C::KP + ks) * 64 + lane];
            };
#pragma unroll
            for (int g = 0; g < DIST; ++g) load_group(g, fb[g], fw[g]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = g % (DIST + 1);
                if (g + DIST < NG) load_group(g + DIST, fb[(g + DIST) % (DIST + 1)], fw[(g + DIST) % (DIST + 1)]);
                // pin the order: left alone, hipcc sinks every ds_read next to its MFMA (read, wait, mfma),
                // which exposes the full LDS latency on each MFMA
                if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < VB; ++v) {
                    const uint4 bv = AS_FRAG(fb[cur][v]);
                    // accC starts a new output plane: its first MFMA takes a zero C operand
                    accC[v] = P::mfma(AS_FRAG(fw[cur][0]), bv,
                                                                      g == 0 ? fzero : accC[v], 0, 0, 0);
                    accB[v] = P::mfma(AS_FRAG(fw[cur][1]), bv, accB[v], 0, 0, 0);
                    accA[v] = P::mfma(AS_FRAG(fw[cur][2]), bv, accA[v], 0, 0, 0);
                }
                if (PIN) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int v = 0; v < VB; ++v) accC[v] = fzero;
        }
        stamp(p, 2);
        // early waves (and every wave when !STAG) finish output plane p-1 here; late waves did plane p-2 above
        if (!STAG || !late_wave) epilogue(accA, p - 1);
        if (!STAG) {
            const int ozf = p - 1;
            if (!(ABL & 1) && ozf >= zs && ozf < ze && ((ozf & 15) == 15 || ozf == ze - 1)) flush_stats(ozf >> 4);
        }
        stamp(p, 3);
        __syncthreads();  // every wave is done reading plane p
        stamp(p, 4);
        if (next_needed) write_plane();
        stamp(p, 5);
        __syncthreads();
        stamp(p, 6);
    };

    // prologue: first input plane of the segment (zs-1, or zs when zs == 0)
    {
        const int p0 = zs - 1;
        if (p0 >= 0) {
            issue_loads(p0);
            write_plane();
        }
        __syncthreads();
    }
    // steps beyond ze are no-ops (compute, loads and emit are all guarded), so the triple needs no branches
    for (int p = zs - 1; p <= ze + (STAG ? 1 : 0); p += 3) {
        step(p, a0, a1, a2);
        step(p + 1, a1, a2, a0);
        step(p + 2, a2, a0, a1);
    }

}


// ---------------------------------------------------------------------------------------------------
// v2: double-buffered 32-channel half-planes.  Every (plane p, source s) pair is one sub-step with its own
// 10x34x32ch halo plane (21.8 KB); two LDS buffers alternate, so the next sub-step's data is written
// while the current one is being multiplied and ONE barrier per sub-step suffices.  A 64-channel
// (concat) layer is simply two sub-steps per plane, which also makes its LDS budget fit:
// weights 108 KB + 2 x 21.8 KB.  8 waves (one output row each), 2 waves per SIMD.
// ---------------------------------------------------------------------------------------------------
template <int NSRC>
struct Zm2Cfg {
    static constexpr int KP = 2 * NSRC;              // k-steps per tap
    static constexpr int WELEMS = 27 * KP * 64;      // uint4
    static constexpr int PELEMS = 4 * ZM_PLANE;      // one 32-channel halo plane, uint4
    static constexpr int NT = 512;
    static constexpr int NPRE = (PELEMS + NT - 1) / NT;  // 3
    static constexpr size_t LDS_BYTES = (size_t)(WELEMS + 2 * (((PELEMS + 63) / 64) * 64)) * 16 + 2048;
};

template <class P, int NSRC, int ABL = 0, bool DMA = false, bool PIPE = false>
__global__ void __launch_bounds__(512, 2) conv3_zmarch2_kernel(const uint4* __restrict__ in1, const uint4* __restrict__ in2,
                                                              const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                              uint4* __restrict__ out, float* __restrict__ partials, int D,
                                                              int H, int W, int tilesY, int tilesX, int zseg,
                                                              const uint4* __restrict__ zero16) {
    using C = Zm2Cfg<NSRC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* lds_w = reinterpret_cast<uint4*>(smem_raw);
    constexpr int PSTRIDE = ((C::PELEMS + 63) / 64) * 64;  // whole 64-element wave pieces (LDS-DMA writes 1 KiB per wave)
    uint4* lds_p = lds_w + C::WELEMS;  // two buffers of PSTRIDE
    float* red = reinterpret_cast<float*>(lds_p + 2 * PSTRIDE);

    const int n = blockIdx.z, seg = blockIdx.y, tile = blockIdx.x;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int y0 = ty * ZM_TY, x0 = tx * ZM_TX;
    const int zs = seg * zseg, ze = min(zs + zseg, D);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long plane = (long long)H * W;
    const long long vox = (long long)D * plane;

    for (int i = threadIdx.x; i < C::WELEMS; i += C::NT) lds_w[i] = wpk[i];

    // staging map (constant along z and identical for both sources: each has 4 chunks of 8 channels)
    long long goff[C::NPRE];
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < C::NPRE; ++j) {
        const int i = threadIdx.x + C::NT * j;
        goff[j] = 0;
        if (i < C::PELEMS) {
            const int xh = i % ZM_HX, yh = (i / ZM_HX) % ZM_HY, c = i / ZM_PLANE;
            const int gy = y0 + yh - 1, gx = x0 + xh - 1;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                goff[j] = ((long long)n * 4 + c) * vox + (long long)gy * W + gx;
                valid |= 1u << j;
            }
        }
    }
    uint4 pre[C::NPRE];
    auto issue_loads = [&](int p, int s) __attribute__((always_inline)) {
        const uint4* src = (NSRC == 2 && s == 1) ? in2 : in1;
#pragma unroll
        for (int j = 0; j < C::NPRE; ++j) {
            const bool ok = (valid >> j) & 1u;
            pre[j] = src[ok ? goff[j] + (long long)p * plane : 0];
        }
    };
    // DMA variant: the halo plane goes HBM -> LDS directly (global_load_lds_dwordx4: no staging VGPRs, no ds_write);
    // each wave-instruction writes 64 consecutive elements, out-of-window lanes read a 16-byte block of zeros
    auto dma_plane = [&](int p, int s, int buf) __attribute__((always_inline)) {
        const uint4* src = (NSRC == 2 && s == 1) ? in2 : in1;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int j = 0; j < C::NPRE; ++j) {
            const int e0 = C::NT * j + wv * 64;  // first element of this wave's piece
            if (e0 < C::PELEMS) {
                const bool ok = (valid >> j) & 1u;
                const uint4* g = ok ? src + goff[j] + (long long)p * plane : zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds_p + buf * PSTRIDE + e0), 16, 0, 0);
            }
        }
    };
    auto write_plane = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < C::NPRE; ++j) {
            const int i = threadIdx.x + C::NT * j;
            const bool ok = (valid >> j) & 1u;
            if (i < C::PELEMS) lds_p[buf * PSTRIDE + i] = ok ? pre[j] : make_uint4(0, 0, 0, 0);
        }
    };

    const int lb = (h * ZM_HY + wave) * ZM_HX + col;  // this lane's voxel (row = wave, column col), chunk half h
    f32x16 fzero;
#pragma unroll
    for (int r = 0; r < 16; ++r) fzero[r] = 0.f;
    f32x16 a0 = fzero, a1 = fzero, a2 = fzero;
    float bs[16], ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bs[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * h];
        ssum[r] = ssq[r] = 0.f;
    }
    const int nzc = (D + 15) / 16;
    auto flush_stats = [&](int zc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = ssum[r], b = ssq[r];
            a = dlv_half_sum32(a);  // DPP adds; totals valid in lanes 16-31 / 48-63
            b = dlv_half_sum32(b);
            if (col == 31) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                red[(wave * 32 + co) * 2] = a;
                red[(wave * 32 + co) * 2 + 1] = b;
            }
            ssum[r] = ssq[r] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int i = threadIdx.x;
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += red[w8 * 64 + i];
            const long long nparts = (long long)gridDim.x * nzc;
            const long long part = (long long)zc * gridDim.x + tile;
            partials[(((long long)n * nparts + part) * 32 + (i >> 1)) * 2 + (i & 1)] = v;
        }
        __syncthreads();
    };

    // PIPE: the finished plane is only copied out of its accumulator (+bias) at the end of a step; its statistics,
    // conversion and stores are issued in four pieces BETWEEN the MFMAs of the next step, and the fetch of the next
    // halo plane is issued after the first MFMA group, so that neither sits between two MFMA phases
    float pv[16];
    bool pok = false, pflush = false;
    long long po = 0;
    int pzc = 0;
    auto piece = [&](int q) __attribute__((always_inline)) {
        if (pok) {
#pragma unroll
            for (int r = 4 * q; r < 4 * q + 4; ++r) {
                ssum[r] += pv[r];
                ssq[r] = fmaf(pv[r], pv[r], ssq[r]);
            }
            uint2 u;
            u.x = P::pack2(pv[4 * q + 0], pv[4 * q + 1]);
            u.y = P::pack2(pv[4 * q + 2], pv[4 * q + 3]);
            uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * 4 + q) * vox + po);
            dst[h] = u;
        }
    };
    // one sub-step: (plane p, source S) sits in buffer BUF; the data of the NEXT sub-step is fetched and written
    // into the other buffer meanwhile.  kz=2 -> accA (out[p-1]), kz=1 -> accB (out[p]), kz=0 -> accC (out[p+1]).
    auto substep = [&](int p, auto S_, auto BUF_, f32x16& accA, f32x16& accB, f32x16& accC) __attribute__((always_inline)) {
        constexpr int S = decltype(S_)::value, BUF = decltype(BUF_)::value;
        constexpr bool LAST = (S == NSRC - 1);
        // next sub-step's data
        const int pn = LAST ? p + 1 : p;
        constexpr int sn = LAST ? 0 : S + 1;
        const bool next_needed = !(ABL & 2) && pn <= ze && pn >= 0 && pn < D;
        auto fetch_next = [&]() __attribute__((always_inline)) {
            if (next_needed) {
                if (DMA) dma_plane(pn, sn, BUF ^ 1);
                else issue_loads(pn, sn);
            }
        };
        if (!PIPE) fetch_next();
        if (p >= 0 && p < D && p <= ze) {
            constexpr int NG = 18;  // 9 (ky,kx) x 2 k-steps of this source
            const uint4* pb = lds_p + BUF * PSTRIDE;
            uint4 fb[2], fw[2][3];
            auto load_group = [&](int g, uint4& b, uint4(&w)[3]) __attribute__((always_inline)) {
                const int ks = g & 1, kx = (g >> 1) % 3, ky = g / 6;
                b = pb[lb + (ks * 2 * ZM_HY + ky) * ZM_HX + kx];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) w[kz] = lds_w[(((kz * 3 + ky) * 3 + kx) * C::KP + S * 2 + ks) * 64 + lane];
            };
            load_group(0, fb[0], fw[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = g & 1;
                if (g + 1 < NG) load_group(g + 1, fb[cur ^ 1], fw[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
                const uint4 bv = AS_FRAG(fb[cur]);
                accC = P::mfma(AS_FRAG(fw[cur][0]), bv,
                                                               (S == 0 && g == 0) ? fzero : accC, 0, 0, 0);
                accB = P::mfma(AS_FRAG(fw[cur][1]), bv, accB, 0, 0, 0);
                accA = P::mfma(AS_FRAG(fw[cur][2]), bv, accA, 0, 0, 0);
                if (PIPE) {
                    if (g == 0) fetch_next();
                    if (S == 0 && g >= 2 && g <= 8 && (g & 1) == 0) piece(g / 2 - 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (S == 0) accC = fzero;
            if (PIPE) {
                fetch_next();
                if (S == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) piece(q);
                }
            }
        }
        if (PIPE && S == 0) {
            pok = false;
            if (pflush) flush_stats(pzc);
            pflush = false;
        }
        if (PIPE && LAST) {
            const int oz = p - 1;
            const bool emit = !(ABL & 1) && oz >= zs && oz < ze;
            const int oy = y0 + wave, ox = x0 + col;
            pok = emit && oy < H && ox < W;
            po = (long long)oz * plane + (long long)oy * W + ox;
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = accA[r];  // no bias: InstanceNorm removes any per-channel constant
            pflush = emit && ((oz & 15) == 15 || oz == ze - 1);
            pzc = oz >> 4;
        }
        if (!PIPE && LAST) {
            const int oz = p - 1;
            const bool emit = !(ABL & 1) && oz >= zs && oz < ze;
            if (ABL & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(accA[r]));
            }
            const int oy = y0 + wave, ox = x0 + col;
            const bool ok = emit && oy < H && ox < W;
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                val[r] = accA[r] + bs[r];
                if (ok) {
                    ssum[r] += val[r];
                    ssq[r] = fmaf(val[r], val[r], ssq[r]);
                }
            }
            if (ok) {
                const long long o = (long long)oz * plane + (long long)oy * W + ox;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = P::pack2(val[4 * g + 0], val[4 * g + 1]);
                    u.y = P::pack2(val[4 * g + 2], val[4 * g + 3]);
                    uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * 4 + g) * vox + o);
                    dst[h] = u;
                }
            }
            if (emit && ((oz & 15) == 15 || oz == ze - 1)) flush_stats(oz >> 4);
        }
        if (next_needed && !DMA) write_plane(BUF ^ 1);
        __syncthreads();  // next data visible (the barrier's fence drains a pending LDS-DMA); everybody is done reading BUF
    };

    // prologue: data of the first sub-step (plane zs-1, source 0) into buffer 0
    if (zs - 1 >= 0 && !(ABL & 2)) {
        if (DMA) {
            dma_plane(zs - 1, 0, 0);
        } else {
            issue_loads(zs - 1, 0);
            write_plane(0);
        }
    }
    __syncthreads();
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (NSRC == 1) {
        // 6 planes per iteration: buffer parity and accumulator roles are both static
        for (int p = zs - 1; p <= ze; p += 6) {
            substep(p + 0, I0{}, I0{}, a0, a1, a2);
            substep(p + 1, I0{}, I1{}, a1, a2, a0);
            substep(p + 2, I0{}, I0{}, a2, a0, a1);
            substep(p + 3, I0{}, I1{}, a0, a1, a2);
            substep(p + 4, I0{}, I0{}, a1, a2, a0);
            substep(p + 5, I0{}, I1{}, a2, a0, a1);
        }
    } else {
        for (int p = zs - 1; p <= ze; p += 3) {
            substep(p + 0, I0{}, I0{}, a0, a1, a2);
            substep(p + 0, I1{}, I1{}, a0, a1, a2);
            substep(p + 1, I0{}, I0{}, a1, a2, a0);
            substep(p + 1, I1{}, I1{}, a1, a2, a0);
            substep(p + 2, I0{}, I0{}, a2, a0, a1);
            substep(p + 2, I1{}, I1{}, a2, a0, a1);
        }
    }
    if (PIPE) {  // the last finished plane is still pending
#pragma unroll
        for (int q = 0; q < 4; ++q) piece(q);
        if (pflush) flush_stats(pzc);
    }
}


// ---------------------------------------------------------------------------------------------------
// v4: v2's double-buffered half-planes with a software-pipelined step.  Measured with in-kernel stamps
// (profiles/zm_timeline.py), v1 spends ~2500 of its ~5700 cycles per step OUTSIDE the MFMA loop with the
// matrix pipe idle: epilogue 1170, two barriers 320, plane write 485, issue of the next plane's loads 590.
// Here nothing but "copy the finished accumulator out, write the next plane, one barrier" sits between two
// MFMA phases:
//   * the fetch of the next halo plane is issued after the first MFMA group of the step,
//   * the finished plane is packed to 16 bit (8 registers) at the end of its step; its InstanceNorm sums and
//     its stores are issued in four pieces between the MFMA groups of the NEXT step,
//   * steps whose every guard is statically true (the bulk) run from a branch-free instantiation.
// The conv bias is not applied: every 3x3x3 conv of the network is followed by InstanceNorm, which removes
// any per-channel constant exactly (the sums are taken on the same bias-free values).  The sums are taken on
// the 16-bit rounded values that are stored, i.e. exactly on what the normalisation pass reads back.
// Requires H % 8 == 0 and W % 32 == 0 (launcher falls back to v1 otherwise).
// ---------------------------------------------------------------------------------------------------
template <class P, int NSRC, bool STAMP = false, int ABL = 0>
__global__ void __launch_bounds__(512, 2) conv3_zmarch4_kernel(const uint4* __restrict__ in1, const uint4* __restrict__ in2,
                                                              const uint4* __restrict__ wpk, uint4* __restrict__ out,
                                                              float* __restrict__ partials, int D, int H, int W,
                                                              int tilesY, int tilesX, int zseg, unsigned long long* __restrict__ stamps) {
    using C = Zm2Cfg<NSRC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // LDS: the two plane buffers first (every fragment address = one per-lane base + an immediate offset < 64 KB),
    // then the statistics scratch, then the weights
    constexpr int PSTRIDE = ((C::PELEMS + 63) / 64) * 64;
    uint4* lds_p = reinterpret_cast<uint4*>(smem_raw);  // two buffers of PSTRIDE
    float* red = reinterpret_cast<float*>(lds_p + 2 * PSTRIDE);
    uint4* lds_w = lds_p + 2 * PSTRIDE + 128;

    const int n = blockIdx.z, seg = blockIdx.y, tile = blockIdx.x;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int y0 = ty * ZM_TY, x0 = tx * ZM_TX;
    const int zs = seg * zseg, ze = min(zs + zseg, D);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int plane = H * W;
    const long long vox = (long long)D * plane;

    for (int i = threadIdx.x; i < C::WELEMS; i += C::NT) lds_w[i] = wpk[i];

    // staging map: element i of the halo plane <- chunk c, row gy, column gx of the window (BYTE offsets relative
    // to the window's first chunk; the launcher guarantees 4 chunks x D*H*W x 16 B < 2^32)
    const uint4* base1 = in1 + (long long)n * 4 * vox;
    const uint4* base2 = NSRC == 2 ? in2 + (long long)n * 4 * vox : nullptr;
    unsigned goff[C::NPRE];
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < C::NPRE; ++j) {
        const int i = threadIdx.x + C::NT * j;
        goff[j] = 0;
        if (i < C::PELEMS) {
            const int xh = i % ZM_HX, yh = (i / ZM_HX) % ZM_HY, c = i / ZM_PLANE;
            const int gy = y0 + yh - 1, gx = x0 + xh - 1;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                goff[j] = (unsigned)(((long long)c * vox + (long long)gy * W + gx) * 16);  // bytes
                valid |= 1u << j;
            }
        }
    }
    uint4 pre[C::NPRE];
    auto issue_loads = [&](int p, int s) __attribute__((always_inline)) {
        // wave-uniform plane pointer + constant per-lane offsets (goff is 0 for out-of-window lanes): SGPR-base
        // loads, no per-step vector address arithmetic
        const char* sp = reinterpret_cast<const char*>(((NSRC == 2 && s == 1) ? base2 : base1) + (long long)p * plane);
#pragma unroll
        for (int j = 0; j < C::NPRE; ++j) {
            unsigned o = goff[j];
            asm volatile("" : "+v"(o));  // keeps the zero-extension next to the load: SGPR base + 32-bit VGPR offset form
            pre[j] = *reinterpret_cast<const uint4*>(sp + o);
        }
    };
    auto write_plane = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < C::NPRE; ++j) {
            const int i = threadIdx.x + C::NT * j;
            const bool ok = (valid >> j) & 1u;
            if (i < C::PELEMS) lds_p[buf * PSTRIDE + i] = ok ? pre[j] : make_uint4(0, 0, 0, 0);
        }
    };

    const int lb = (h * ZM_HY + wave) * ZM_HX + col;  // this lane's voxel (row = wave, column col), chunk half h
    f32x16 fzero;
#pragma unroll
    for (int r = 0; r < 16; ++r) fzero[r] = 0.f;
    f32x16 a0 = fzero, a1 = fzero, a2 = fzero;
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum[r] = ssq[r] = 0.f;
    const int nzc = (D + 15) / 16;
    auto flush_stats = [&](int zc) __attribute__((always_inline)) {
        // the scratch addresses are derived from a value the optimiser cannot see through, so that it does not
        // hoist 16 of them out of the z loop (they would cost registers in the MFMA loop for a 1-in-16 event)
        int rbase = (wave * 32 + 4 * h) * 2;
        asm volatile("" : "+v"(rbase));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = ssum[r], b = ssq[r];
            a = dlv_half_sum32(a);  // DPP adds; totals valid in lanes 16-31 / 48-63
            b = dlv_half_sum32(b);
            if (col == 31) {
                const int co = (r & 3) + 8 * (r >> 2);
                red[rbase + co * 2] = a;
                red[rbase + co * 2 + 1] = b;
            }
            ssum[r] = ssq[r] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int i = threadIdx.x;
            float v = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += red[w8 * 64 + i];
            const long long nparts = (long long)gridDim.x * nzc;
            const long long part = (long long)zc * gridDim.x + tile;
            partials[(((long long)n * nparts + part) * 32 + (i >> 1)) * 2 + (i & 1)] = v;
        }
        __syncthreads();
    };

    // the pending (finished, not yet emitted) output plane: 16 values per lane, packed
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool pvalid = false, pflush = false;
    int pzc = 0;
    uint4* const outn = out + (long long)n * 4 * vox;
    uint4* pout = outn;  // wave-uniform: chunk 0, plane of the pending output
    const unsigned ooff = ((unsigned)((y0 + wave) * W + x0 + col) * 2u + (unsigned)h) * 8u;  // byte offset of this lane's uint2 within a plane
    // ABL (timing-only ablation builds, wrong results): 1 no statistics/stores, 2 no plane fetch/write, 4 no pack
    auto piece = [&](int q, bool on) __attribute__((always_inline)) {
        if (on && !(ABL & 1)) {
            const unsigned u0 = pk[2 * q], u1 = pk[2 * q + 1];
            const float v0 = P::lo(u0), v1 = P::hi(u0), v2 = P::lo(u1), v3 = P::hi(u1);
            ssum[4 * q + 0] += v0;
            ssum[4 * q + 1] += v1;
            ssum[4 * q + 2] += v2;
            ssum[4 * q + 3] += v3;
            ssq[4 * q + 0] = fmaf(v0, v0, ssq[4 * q + 0]);
            ssq[4 * q + 1] = fmaf(v1, v1, ssq[4 * q + 1]);
            ssq[4 * q + 2] = fmaf(v2, v2, ssq[4 * q + 2]);
            ssq[4 * q + 3] = fmaf(v3, v3, ssq[4 * q + 3]);
            unsigned o = ooff;
            asm volatile("" : "+v"(o));
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(pout + (long long)q * vox) + o) = make_uint2(u0, u1);
        }
    };

    // diagnostic instantiation only: s_memtime stamps per step phase (profiles/zm_timeline.py)
    const bool stamp_on = STAMP && n == 0 && seg == 0 && stamps != nullptr && lane == 0;
    auto stamp = [&](int p, int s, int slot) __attribute__((always_inline)) {
        if (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (stamp_on) stamps[((((long long)tile * 8 + wave) * (D + 4) + (p + 1)) * NSRC + s) * 8 + slot] = t;
        }
    };
    // one sub-step: (plane p, source S) sits in buffer BUF.  kz=2 -> accA (out[p-1]), kz=1 -> accB (out[p]),
    // kz=0 -> accC (out[p+1]).  INTERIOR: every guard below is statically true.
    // Staging runs two sub-steps ahead: the registers `pre` hold the halo data of the NEXT sub-step (fetched during
    // the previous one); they are written into the other buffer - free since the last barrier - after the second
    // MFMA group, and the fetch of the sub-step after next is issued after the fourth.  Between two MFMA phases only
    // "pack the finished accumulator, barrier" remains, and the first weight fragments of the next sub-step are
    // already on their way when the barrier is reached.
    uint4 fw0[3];  // weight fragments of group 0 of the coming sub-step (do not depend on the barrier)
    auto load_w = [&](int S, int g, uint4(&w)[3]) __attribute__((always_inline)) {
        const int ks = g & 1, kx = (g >> 1) % 3, ky = g / 6;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) w[kz] = lds_w[(((kz * 3 + ky) * 3 + kx) * C::KP + S * 2 + ks) * 64 + lane];
    };
    auto substep = [&](int p, auto S_, auto BUF_, auto INT_, f32x16& accA, f32x16& accB, f32x16& accC) __attribute__((always_inline)) {
        constexpr int S = decltype(S_)::value, BUF = decltype(BUF_)::value;
        constexpr bool INTERIOR = decltype(INT_)::value;
        constexpr bool LAST = (S == NSRC - 1);
        const int pn = LAST ? p + 1 : p;  // next sub-step (pn, sn): its data is in `pre`
        constexpr int sn = LAST ? 0 : S + 1;
        const int pnn = NSRC == 1 ? p + 2 : p + 1;  // the one after: (pnn, S)
        const bool next_needed = !(ABL & 2) && (INTERIOR || (pn <= ze && pn >= 0 && pn < D));
        const bool nn_needed = !(ABL & 2) && (INTERIOR || (pnn <= ze && pnn >= 0 && pnn < D));
        const bool on = INTERIOR || pvalid;
        stamp(p, S, 0);
        if (STAMP && stamp_on) stamps[((((long long)tile * 8 + wave) * (D + 4) + (p + 1)) * NSRC + S) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        if (INTERIOR || (p >= 0 && p < D && p <= ze)) {
            constexpr int NG = 18;  // 9 (ky,kx) x 2 k-steps of this source
            const uint4* pb = lds_p + BUF * PSTRIDE;
            uint4 fb[2], fw[2][3];
            auto load_b = [&](int g, uint4& bfr) __attribute__((always_inline)) {
                const int ks = g & 1, kx = (g >> 1) % 3, ky = g / 6;
                bfr = pb[lb + (ks * 2 * ZM_HY + ky) * ZM_HX + kx];
            };
            load_b(0, fb[0]);
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) fw[0][kz] = fw0[kz];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = g & 1;
                if (g + 1 < NG) {
                    load_b(g + 1, fb[cur ^ 1]);
                    load_w(S, g + 1, fw[cur ^ 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint4 bv = AS_FRAG(fb[cur]);
                if (g == NG - 1) {  // accA first: its pack right after the loop then waits for less
                    accA = P::mfma(AS_FRAG(fw[cur][2]), bv, accA, 0, 0, 0);
                    accB = P::mfma(AS_FRAG(fw[cur][1]), bv, accB, 0, 0, 0);
                    accC = P::mfma(AS_FRAG(fw[cur][0]), bv, accC, 0, 0, 0);
                } else {
                    accC = P::mfma(AS_FRAG(fw[cur][0]), bv, (S == 0 && g == 0) ? fzero : accC, 0, 0, 0);
                    accB = P::mfma(AS_FRAG(fw[cur][1]), bv, accB, 0, 0, 0);
                    accA = P::mfma(AS_FRAG(fw[cur][2]), bv, accA, 0, 0, 0);
                }
                if (g == 1 && next_needed) write_plane(BUF ^ 1);
                if (g == 3 && nn_needed) issue_loads(pnn, S);
                if (S == 0 && g >= 4 && g <= 10 && (g & 1) == 0) piece(g / 2 - 2, on);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (S == 0) accC = fzero;
            if (next_needed) write_plane(BUF ^ 1);
            if (nn_needed) issue_loads(pnn, S);
            if (S == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) piece(q, on);
            }
        }
        stamp(p, S, 1);
        load_w(sn, 0, fw0);
        if (S == 0) {
            pvalid = false;
            if (pflush) flush_stats(pzc);
            pflush = false;
        }
        stamp(p, S, 2);
        stamp(p, S, 3);
        if (LAST) {
            const int oz = p - 1;
            const bool emit = INTERIOR || (oz >= zs && oz < ze);
            pvalid = emit;
            pout = outn + (long long)oz * plane;
            if (ABL & 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(accA[r]));
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) pk[q] = P::pack2(accA[2 * q], accA[2 * q + 1]);
            }
            pflush = emit && ((oz & 15) == 15 || oz == ze - 1);
            pzc = oz >> 4;
        }
        stamp(p, S, 4);
        __syncthreads();  // next data visible; everybody is done reading BUF
        stamp(p, S, 5);
    };

    // prologue: data of the first sub-step (plane zs-1, source 0) into buffer 0, data of the second into `pre`
    if (!(ABL & 2)) {
        if (zs - 1 >= 0) {
            issue_loads(zs - 1, 0);
            write_plane(0);
        }
        const int p1 = NSRC == 1 ? zs : zs - 1;
        if (p1 >= 0 && p1 < D && p1 <= ze) issue_loads(p1, NSRC == 1 ? 0 : 1);
    }
    load_w(0, 0, fw0);
    __syncthreads();
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;
    const int pmax = min(ze, D - 1);  // last plane that is fetched
    if (NSRC == 1) {
        // 6 planes per iteration: buffer parity and accumulator roles are both static
        for (int p = zs - 1; p <= ze; p += 6) {
            if (p >= zs + 2 && p + 7 <= pmax) {
                substep(p + 0, I0{}, I0{}, BT{}, a0, a1, a2);
                substep(p + 1, I0{}, I1{}, BT{}, a1, a2, a0);
                substep(p + 2, I0{}, I0{}, BT{}, a2, a0, a1);
                substep(p + 3, I0{}, I1{}, BT{}, a0, a1, a2);
                substep(p + 4, I0{}, I0{}, BT{}, a1, a2, a0);
                substep(p + 5, I0{}, I1{}, BT{}, a2, a0, a1);
            } else {
                substep(p + 0, I0{}, I0{}, BF{}, a0, a1, a2);
                substep(p + 1, I0{}, I1{}, BF{}, a1, a2, a0);
                substep(p + 2, I0{}, I0{}, BF{}, a2, a0, a1);
                substep(p + 3, I0{}, I1{}, BF{}, a0, a1, a2);
                substep(p + 4, I0{}, I0{}, BF{}, a1, a2, a0);
                substep(p + 5, I0{}, I1{}, BF{}, a2, a0, a1);
            }
        }
    } else {
        for (int p = zs - 1; p <= ze; p += 3) {
            if (p >= zs + 2 && p + 3 <= pmax) {
                substep(p + 0, I0{}, I0{}, BT{}, a0, a1, a2);
                substep(p + 0, I1{}, I1{}, BT{}, a0, a1, a2);
                substep(p + 1, I0{}, I0{}, BT{}, a1, a2, a0);
                substep(p + 1, I1{}, I1{}, BT{}, a1, a2, a0);
                substep(p + 2, I0{}, I0{}, BT{}, a2, a0, a1);
                substep(p + 2, I1{}, I1{}, BT{}, a2, a0, a1);
            } else {
                substep(p + 0, I0{}, I0{}, BF{}, a0, a1, a2);
                substep(p + 0, I1{}, I1{}, BF{}, a0, a1, a2);
                substep(p + 1, I0{}, I0{}, BF{}, a1, a2, a0);
                substep(p + 1, I1{}, I1{}, BF{}, a1, a2, a0);
                substep(p + 2, I0{}, I0{}, BF{}, a2, a0, a1);
                substep(p + 2, I1{}, I1{}, BF{}, a2, a0, a1);
            }
        }
    }
    // the last finished plane is still pending
#pragma unroll
    for (int q = 0; q < 4; ++q) piece(q, pvalid);
    if (pflush) flush_stats(pzc);
}

}  // namespace

// returns the number of partial-sum rows per sample (columns) or a negative error
int dlv_conv3_zmarch_launch(dlv_ctx* ctx, bool f16, int cin, int cout, const void* in1, int c1, const void* in2, int c2,
                            const void* wpk, const float* bias, void* out, float* partials, int B, int D, int H, int W,
                            int* nparts) {
    if (cout % 32 || cout <= 0) return dlv_fail(ctx, DLV_EUNSUP, "z-march conv: Cout must be a multiple of 32");
    const int ncb = cout / 32;
    const int variant = ctx->zm_variant;  // DLV_ZM_VARIANT at context creation, or dlv_debug_set_zm_variant
    // 16-row tiles (8 waves x 2 rows) only as A/B variant 3: measured equal/slower than 8 rows x 1 (profiles/README.md)
#ifdef DLV_DIAG
    const int tyt = (cin == 32 && variant == 3) ? 16 : 8;
#else
    const int tyt = 8;
#endif
    const int tilesY = dlv_cdiv(H, tyt), tilesX = dlv_cdiv(W, ZM_TX);
    // split long columns (in multiples of 16 planes) so that small batches still fill 256 CUs
    int zseg = ((D + 15) / 16) * 16;
    while ((long long)B * tilesY * tilesX * ncb * dlv_cdiv(D, zseg) < 512 && zseg > 16) zseg = std::max(16, ((zseg / 2 + 15) / 16) * 16);
    const int nseg = dlv_cdiv(D, zseg);
    dim3 grid(tilesY * tilesX, nseg * ncb, B);
    *nparts = tilesY * tilesX * ((D + 15) / 16);
    // kernel variants (DLV_ZM_VARIANT selects one for A/B timing): VB rows per wave, TYT tile rows -> TYT/VB waves
#define DLV_ZM_LAUNCH_P(P_, CIN_, VB_, MINW_, TYT_, PIN_, DIST_, ABL_, STAG_)                                                                                  \
    do {                                                                                                                 \
        static dlv_attr_bits attr_set{0}; /* bit per device */                                                               \
        if (!dlv_attr_is_set(attr_set, ctx->device)) {                                                                                                 \
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_zmarch_kernel<P_, CIN_, VB_, MINW_, TYT_, PIN_, DIST_, ABL_, STAG_>,                         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZmCfg<CIN_, TYT_>::LDS_BYTES));  \
            dlv_attr_mark(attr_set, ctx->device);                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((conv3_zmarch_kernel<P_, CIN_, VB_, MINW_, TYT_, PIN_, DIST_, ABL_, STAG_>), grid, dim3(64 * TYT_ / VB_), (ZmCfg<CIN_, TYT_>::LDS_BYTES),       \
                           ctx->stream, (const uint4*)in1, c1 / 8, (const uint4*)in2, c2 / 8, (const uint4*)wpk, bias,   \
                           (uint4*)out, partials, D, H, W, tilesY, tilesX, zseg, (unsigned long long*)ctx->stamp_buf, nseg, cout / 8); \
    } while (0)
#define DLV_ZM2_LAUNCH(P_, NSRC_, ABL_, DMA_, PIPE_)                                                                              \
    do {                                                                                                                 \
        static dlv_attr_bits attr_set2{0}; /* bit per device */                                                              \
        if (!dlv_attr_is_set(attr_set2, ctx->device)) {                                                                                                \
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_zmarch2_kernel<P_, NSRC_, ABL_, DMA_, PIPE_>,                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)Zm2Cfg<NSRC_>::LDS_BYTES)); \
            dlv_attr_mark(attr_set2, ctx->device);                                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((conv3_zmarch2_kernel<P_, NSRC_, ABL_, DMA_, PIPE_>), grid, dim3(512), Zm2Cfg<NSRC_>::LDS_BYTES,  \
                           ctx->stream, (const uint4*)in1, (const uint4*)in2, (const uint4*)wpk, bias, (uint4*)out,      \
                           partials, D, H, W, tilesY, tilesX, zseg, (const uint4*)ctx->zero_page);                       \
    } while (0)
#ifdef DLV_DIAG  // A/B, stamped and timing-only builds: libdelivr_hip_diag.so only (make diag)
    if (cout == 32 && variant >= 40 && variant <= 45 && H % 8 == 0 && W % 32 == 0 && (long long)4 * D * H * W * 16 < (1ll << 32) &&
        ((cin == 32 && c1 == 32) || (cin == 64 && c1 == 32 && c2 == 32))) {
#define DLV_ZM4_LAUNCH(P_, NSRC_, ST_, ABL_)                                                                                       \
    do {                                                                                                                 \
        static dlv_attr_bits attr_set4{0}; /* bit per device */                                                              \
        if (!dlv_attr_is_set(attr_set4, ctx->device)) {                                                                                                \
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_zmarch4_kernel<P_, NSRC_, ST_, ABL_>,                               \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)Zm2Cfg<NSRC_>::LDS_BYTES)); \
            dlv_attr_mark(attr_set4, ctx->device);                                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((conv3_zmarch4_kernel<P_, NSRC_, ST_, ABL_>), grid, dim3(512), Zm2Cfg<NSRC_>::LDS_BYTES, ctx->stream,    \
                           (const uint4*)in1, (const uint4*)in2, (const uint4*)wpk, (uint4*)out, partials, D, H, W,      \
                           tilesY, tilesX, zseg, (unsigned long long*)ctx->stamp_buf);                                   \
    } while (0)
#define DLV_ZM4_PICK(NSRC_)                                            \
    do {                                                               \
        if (variant == 41) DLV_ZM4_LAUNCH(PF16, NSRC_, true, 0);       \
        else if (variant == 42) DLV_ZM4_LAUNCH(PF16, NSRC_, false, 1); \
        else if (variant == 43) DLV_ZM4_LAUNCH(PF16, NSRC_, false, 2); \
        else if (variant == 44) DLV_ZM4_LAUNCH(PF16, NSRC_, false, 4); \
        else if (variant == 45) DLV_ZM4_LAUNCH(PF16, NSRC_, false, 7); \
        else if (f16) DLV_ZM4_LAUNCH(PF16, NSRC_, false, 0);           \
        else DLV_ZM4_LAUNCH(PBf16, NSRC_, false, 0);                   \
    } while (0)
        if (cin == 32) DLV_ZM4_PICK(1);
        else DLV_ZM4_PICK(2);
#undef DLV_ZM4_PICK
#undef DLV_ZM4_LAUNCH
    } else
    // variants 20 (double-buffered half-planes, register staging), 24 (LDS-DMA staging), 25 (20 + pipelined epilogue)
    if (cout == 32 && (variant == 20 || variant == 24 || variant == 25) && ((cin == 32 && c1 == 32) || (cin == 64 && c1 == 32 && c2 == 32))) {
#define DLV_ZM2_PICK(P_, NSRC_)                                   \
    do {                                                          \
        if (variant == 24) DLV_ZM2_LAUNCH(P_, NSRC_, 0, true, false);   \
        else if (variant == 25) DLV_ZM2_LAUNCH(P_, NSRC_, 0, false, true); \
        else DLV_ZM2_LAUNCH(P_, NSRC_, 0, false, false);          \
    } while (0)
        if (cin == 32) {
            if (f16) DLV_ZM2_PICK(PF16, 1);
            else DLV_ZM2_PICK(PBf16, 1);
        } else {
            if (f16) DLV_ZM2_PICK(PF16, 2);
            else DLV_ZM2_PICK(PBf16, 2);
        }
#undef DLV_ZM2_PICK
    } else
#endif
    // default (measured fastest on C2, profiles/README.md): one row per wave, 8 waves, 2 waves per SIMD
#define DLV_ZM_LAUNCH(...) DLV_ZM_LAUNCH_P(PBf16, __VA_ARGS__)
    // the fp16 format runs the default kernel only (the A/B variants below are bf16)
    if (f16) {
        if (cin == 32) {
#ifdef DLV_DIAG
            if (variant == 6) DLV_ZM_LAUNCH_P(PF16, 32, 1, 2, 8, true, 1, 8, false);
            else if (variant == 30) DLV_ZM_LAUNCH_P(PF16, 32, 1, 2, 8, true, 1, 16, false);
            else if (variant == 31) DLV_ZM_LAUNCH_P(PF16, 32, 1, 2, 8, true, 1, 19, false);  // stamped, no staging, no epilogue
            else
#endif
            DLV_ZM_LAUNCH_P(PF16, 32, 1, 2, 8, true, 1, 0, false);
        } else if (cin == 64) {
#ifdef DLV_DIAG
            if (variant == 6) DLV_ZM_LAUNCH_P(PF16, 64, 1, 2, 8, true, 1, 8, false);
            else if (variant == 30) DLV_ZM_LAUNCH_P(PF16, 64, 1, 2, 8, true, 1, 16, false);
            else
#endif
            DLV_ZM_LAUNCH_P(PF16, 64, 1, 2, 8, true, 1, 0, false);
        }
        else return dlv_fail(ctx, DLV_EUNSUP, "z-march conv: Cin must be 32 or 64");
    } else
    // variants 11/12/13 are timing-only ablations (no epilogue / no staging / neither): wrong results
    if (cin == 32) {
#ifdef DLV_DIAG
        if (variant == 11) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 1, false);
        else if (variant == 12) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 2, false);
        else if (variant == 13) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 3, false);
        else if (variant == 3) DLV_ZM_LAUNCH(32, 2, 2, 16, true, 1, 0, false);
        else if (variant == 4) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 0, true);
        else
#endif
        DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 0, false);
    } else if (cin == 64) {
#ifdef DLV_DIAG
        if (variant == 11) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 1, false);
        else if (variant == 12) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 2, false);
        else if (variant == 13) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 3, false);
        else if (variant == 4) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 0, true);
        else
#endif
        DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 0, false);
    } else {
        return dlv_fail(ctx, DLV_EUNSUP, "z-march conv: Cin must be 32 or 64");
    }
#undef DLV_ZM_LAUNCH
#undef DLV_ZM_LAUNCH_P
#undef DLV_ZM2_LAUNCH
    DLV_LAUNCH_CHECK(ctx, "conv3_zmarch_kernel");
    return DLV_OK;
}
