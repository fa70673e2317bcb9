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
                                                           int H, int W, int tilesY, int tilesX, int zseg) {
    using C = ZmCfg<CIN, TYT>;
    constexpr bool LATE = false;
    constexpr int NT = 64 * TYT / VB;            // threads: TYT rows / VB rows per wave
    constexpr int NPRE = (C::PELEMS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* lds_w = reinterpret_cast<uint4*>(smem_raw);
    uint4* lds_p = lds_w + C::WELEMS;
    float* red = reinterpret_cast<float*>(lds_p + C::PELEMS);  // [4 waves][32][2]

    const int n = blockIdx.z;
    const int seg = blockIdx.y;
    const int tile = blockIdx.x;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int y0 = ty * TYT, x0 = tx * ZM_TX;
    const int zs = seg * zseg, ze = min(zs + zseg, D);  // output planes [zs, ze)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long plane = (long long)H * W;
    const long long vox = (long long)D * plane;

    // ---- weights -> LDS (A-fragment order, lane-linear) ------------------------------------------------
    for (int i = threadIdx.x; i < C::WELEMS; i += NT) lds_w[i] = wpk[i];

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
        bs[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * h];
        ssum[r] = ssq[r] = 0.f;
    }

    // InstanceNorm partial sums are flushed every 16 output planes (absolute z / 16), so that their
    // grouping - and with it every bit of the statistics - does not depend on zseg or the batch size
    const int nzc = (D + 15) / 16;
    auto flush_stats = [&](int zc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = ssum[r], b = ssq[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            if (col == 0) {
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
            partials[(((long long)n * nparts + part) * 32 + (i >> 1)) * 2 + (i & 1)] = v;
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
                    uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * 4 + g) * vox + o);
                    dst[h] = u;
                }
            }
        }
    };

    // one z step: plane p is in LDS (when 0 <= p < D).  kz=2 -> accA (out[p-1]), kz=1 -> accB (out[p]),
    // kz=0 -> accC (out[p+1], started here).  Then out[p-1] is emitted from accA.
    auto step = [&](int p, f32x16(&accA)[VB], f32x16(&accB)[VB], f32x16(&accC)[VB]) __attribute__((always_inline)) {
        const bool next_needed = !(ABL & 2) && (p + 1 <= ze) && (p + 1 >= 0) && (p + 1 < D);
        if (next_needed) issue_loads(p + 1);
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
        // early waves (and every wave when !STAG) finish output plane p-1 here; late waves did plane p-2 above
        if (!STAG || !late_wave) epilogue(accA, p - 1);
        if (!STAG) {
            const int ozf = p - 1;
            if (!(ABL & 1) && ozf >= zs && ozf < ze && ((ozf & 15) == 15 || ozf == ze - 1)) flush_stats(ozf >> 4);
        }
        __syncthreads();  // every wave is done reading plane p
        if (next_needed) write_plane();
        __syncthreads();
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
    static constexpr size_t LDS_BYTES = (size_t)(WELEMS + 2 * PELEMS) * 16 + 2048;
};

template <class P, int NSRC, int ABL = 0>
__global__ void __launch_bounds__(512, 2) conv3_zmarch2_kernel(const uint4* __restrict__ in1, const uint4* __restrict__ in2,
                                                              const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                              uint4* __restrict__ out, float* __restrict__ partials, int D,
                                                              int H, int W, int tilesY, int tilesX, int zseg) {
    using C = Zm2Cfg<NSRC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* lds_w = reinterpret_cast<uint4*>(smem_raw);
    uint4* lds_p = lds_w + C::WELEMS;  // two buffers of PELEMS
    float* red = reinterpret_cast<float*>(lds_p + 2 * C::PELEMS);

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
    auto write_plane = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < C::NPRE; ++j) {
            const int i = threadIdx.x + C::NT * j;
            const bool ok = (valid >> j) & 1u;
            if (i < C::PELEMS) lds_p[buf * C::PELEMS + i] = ok ? pre[j] : make_uint4(0, 0, 0, 0);
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
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            if (col == 0) {
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

    // one sub-step: (plane p, source S) sits in buffer BUF; the data of the NEXT sub-step is fetched and written
    // into the other buffer meanwhile.  kz=2 -> accA (out[p-1]), kz=1 -> accB (out[p]), kz=0 -> accC (out[p+1]).
    auto substep = [&](int p, auto S_, auto BUF_, f32x16& accA, f32x16& accB, f32x16& accC) __attribute__((always_inline)) {
        constexpr int S = decltype(S_)::value, BUF = decltype(BUF_)::value;
        constexpr bool LAST = (S == NSRC - 1);
        // next sub-step's data
        const int pn = LAST ? p + 1 : p;
        constexpr int sn = LAST ? 0 : S + 1;
        const bool next_needed = !(ABL & 2) && pn <= ze && pn >= 0 && pn < D;
        if (next_needed) issue_loads(pn, sn);
        if (p >= 0 && p < D && p <= ze) {
            constexpr int NG = 18;  // 9 (ky,kx) x 2 k-steps of this source
            const uint4* pb = lds_p + BUF * C::PELEMS;
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
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (S == 0) {
            accC = fzero;
        }
        if (LAST) {
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
        if (next_needed) write_plane(BUF ^ 1);
        __syncthreads();  // next data visible; everybody is done reading BUF
    };

    // prologue: data of the first sub-step (plane zs-1, source 0) into buffer 0
    if (zs - 1 >= 0 && !(ABL & 2)) {
        issue_loads(zs - 1, 0);
        write_plane(0);
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
}

}  // namespace

// returns the number of partial-sum rows per sample (columns) or a negative error
int dlv_conv3_zmarch_launch(dlv_ctx* ctx, bool f16, int cin, const void* in1, int c1, const void* in2, int c2, const void* wpk,
                            const float* bias, void* out, float* partials, int B, int D, int H, int W, int* nparts) {
    static int variant = -1;
    if (variant < 0) {
        const char* e = getenv("DLV_ZM_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    // 16-row tiles (8 waves x 2 rows) for the single-source layers unless a variant asks otherwise
    // 16-row tiles (8 waves x 2 rows) only as A/B variant 3: measured equal/slower than 8 rows x 1 (profiles/README.md)
    const int tyt = (cin == 32 && variant == 3) ? 16 : 8;
    const int tilesY = dlv_cdiv(H, tyt), tilesX = dlv_cdiv(W, ZM_TX);
    // split long columns (in multiples of 16 planes) so that small batches still fill 256 CUs
    int zseg = ((D + 15) / 16) * 16;
    while ((long long)B * tilesY * tilesX * dlv_cdiv(D, zseg) < 512 && zseg > 16) zseg = std::max(16, ((zseg / 2 + 15) / 16) * 16);
    const int nseg = dlv_cdiv(D, zseg);
    dim3 grid(tilesY * tilesX, nseg, B);
    *nparts = tilesY * tilesX * ((D + 15) / 16);
    // kernel variants (DLV_ZM_VARIANT selects one for A/B timing): VB rows per wave, TYT tile rows -> TYT/VB waves
#define DLV_ZM_LAUNCH_P(P_, CIN_, VB_, MINW_, TYT_, PIN_, DIST_, ABL_, STAG_)                                                                                  \
    do {                                                                                                                 \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_zmarch_kernel<P_, CIN_, VB_, MINW_, TYT_, PIN_, DIST_, ABL_, STAG_>,                         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZmCfg<CIN_, TYT_>::LDS_BYTES));  \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((conv3_zmarch_kernel<P_, CIN_, VB_, MINW_, TYT_, PIN_, DIST_, ABL_, STAG_>), grid, dim3(64 * TYT_ / VB_), (ZmCfg<CIN_, TYT_>::LDS_BYTES),       \
                           ctx->stream, (const uint4*)in1, c1 / 8, (const uint4*)in2, c2 / 8, (const uint4*)wpk, bias,   \
                           (uint4*)out, partials, D, H, W, tilesY, tilesX, zseg);                                        \
    } while (0)
    // default (measured fastest on C2, profiles/README.md): one row per wave, 8 waves, 2 waves per SIMD
#define DLV_ZM_LAUNCH(...) DLV_ZM_LAUNCH_P(PBf16, __VA_ARGS__)
    // the fp16 format runs the default kernel only (the A/B variants below are bf16)
    if (f16) {
        if (cin == 32) DLV_ZM_LAUNCH_P(PF16, 32, 1, 2, 8, true, 1, 0, false);
        else if (cin == 64) DLV_ZM_LAUNCH_P(PF16, 64, 1, 2, 8, true, 1, 0, false);
        else return dlv_fail(ctx, DLV_EUNSUP, "z-march conv: Cin must be 32 or 64");
    } else
    // variants 11/12/13 are timing-only ablations (no epilogue / no staging / neither): wrong results
    if (cin == 32) {
        if (variant == 11) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 1, false);
        else if (variant == 12) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 2, false);
        else if (variant == 13) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 3, false);
        else if (variant == 3) DLV_ZM_LAUNCH(32, 2, 2, 16, true, 1, 0, false);
        else if (variant == 4) DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 0, true);
        else DLV_ZM_LAUNCH(32, 1, 2, 8, true, 1, 0, false);
    } else if (cin == 64) {
        if (variant == 11) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 1, false);
        else if (variant == 12) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 2, false);
        else if (variant == 13) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 3, false);
        else if (variant == 4) DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 0, true);
        else DLV_ZM_LAUNCH(64, 1, 2, 8, true, 1, 0, false);
    } else {
        return dlv_fail(ctx, DLV_EUNSUP, "z-march conv: Cin must be 32 or 64");
    }
#undef DLV_ZM_LAUNCH
#undef DLV_ZM_LAUNCH_P
#undef DLV_ZM2_LAUNCH
    DLV_LAUNCH_CHECK(ctx, "conv3_zmarch_kernel");
    return DLV_OK;
}
