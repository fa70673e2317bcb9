// unet_bf16.hip - bf16 MFMA U-Net forward for gfx950: the throughput path.
//
// Same arithmetic as unet_f32.hip / MONAI BasicUNet.forward (inference/sliding_window_inferer.py:222,
// ctor inference/inference.py:190-197) with bf16 activations + weights, fp32 MFMA accumulation and
// fp32 InstanceNorm statistics taken from the un-rounded accumulators.
//
// Data layout in HBM ("chunk-planar", z-major slabs per 8-channel chunk):
//     act[n][C/8][D][H][W] of uint4  (one uint4 = 8 consecutive channels of one voxel, bf16)
// so that (a) consecutive x voxels of a chunk are consecutive 16-byte elements (coalesced 1 KiB
// wave loads/stores), and (b) an MFMA B-operand fragment (8 input channels of one voxel per lane)
// is exactly one ds_read_b128 / global_load_dwordx4.
//
// 3x3x3 convolutions are implicit GEMMs on v_mfma_f32_32x32x16_bf16 with
//     A = weights  (rows = 32 output channels, k = 16 input channels of one tap)
//     B = input    (cols = 32 voxels,          k = same 16 channels, shifted by the tap)
//     D[row = cout][col = voxel]  ->  each lane ends up with 4 consecutive output channels of its
//                                     voxel per register quad = one 8-byte bf16 store.
// K runs over 27 taps x Cin/16.  Weights are pre-packed in A-fragment order (one coalesced 1 KiB
// load per fragment, L2-resident); the input halo tile is staged through LDS.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "prec16.h"

namespace {

#define AS_FRAG(x) (x)

__device__ __forceinline__ float mish_fast(float y) {
    // y * tanh(softplus(y)) = y * t / (t + 2) = y - 2y / d,  d = t + 2 = n (n + 2) + 2,  n = e^y.
    // In this form nothing has to be clamped: for large y, n and d overflow to +inf, 1/d = 0 and the result is y exactly (torch's
    // softplus threshold does the same from y = 20).  For very negative y the difference y - 2y/d cancels: its absolute error is
    // |y| * 2^-23 (about 1e-6 at y = -10, where the exact value is -4.5e-4 and fp16's half ulp 2.4e-7: a relative 2e-3 of a value
    // that is itself 5e-4 of the tensor's scale - inside the 1e-3 rel. RMS the 16-bit forward is held to, not below the store's
    // rounding as an earlier comment claimed).  One v_exp_f32 + one v_rcp_f32 (1 ulp) and four plain
    // VALU ops; __fdividef would expand to the 10-instruction IEEE division sequence.
    const float n = __builtin_amdgcn_exp2f(y * 1.44269504f);
    const float d = fmaf(n, n + 2.f, 2.f);
    return fmaf(-2.f * y, __builtin_amdgcn_rcpf(d), y);
}

// two values at once: the plain operations become packed-f32 instructions (v_pk_mul/add/fma_f32, two lanes' worth of
// work per issue slot); element for element the same operations as mish_fast, i.e. the same bits
__device__ __forceinline__ f32x2_t mish_fast2(f32x2_t y) {
    const f32x2_t l2e = {1.44269504f, 1.44269504f}, two = {2.f, 2.f}, m2 = {-2.f, -2.f};
    const f32x2_t e = y * l2e;
    const f32x2_t n = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const f32x2_t d = __builtin_elementwise_fma(n, n + two, two);
    const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return __builtin_elementwise_fma(m2 * y, r, y);
}
__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }

// ---------------------------------------------------------------------------------------------------
// weight packing (device side, once per dlv_unet_load)
// ---------------------------------------------------------------------------------------------------
// conv:   out[((cb*27 + t)*KP + kp)*64 + lane][j] = W[cout = cb*32 + (lane&31)][cin = kp*16 + 8*(lane>>5) + j][t]
template <class P>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin, float wscale) {
    const int KP = cin / 16;
    const long long n = (long long)cout * cin * 27;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int kp = (int)(r % KP);
        r /= KP;
        const int t = (int)(r % 27);
        const int cb = (int)(r / 27);
        const int co = cb * 32 + (lane & 31);
        const int ci = kp * 16 + 8 * (lane >> 5) + j;
        const float v = w[((long long)co * cin + ci) * 27 + t] * wscale;  // (2^-shift: exact)
        out[i] = (uint16_t)(P::pack2(v, 0.f) & 0xffffu);
    }
}
// stem (Cin = 1, Cout = 32): K = 64 = {hi byte, lo byte} x 32 tap slots (27 used).  The uint16 input is split
// exactly into x = 256*hi + lo (both exact in bf16), the weights carry the factor 256 for the hi half:
//   out[(s*64 + lane)*8 + j]: tap = 8 s + 4 (lane>>5) + (j>>1), part = j&1 (0: lo byte, 1: hi byte), cout = lane&31
template <class P>
__global__ void pack_stem_w_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, float wscale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, s = i >> 9;
    const int co = lane & 31;
    float v = 0.f;
    // k-step s, lane half h: taps 8s + 4h + (j >> 1), low byte (j even) then high byte (j odd) of the same tap - the
    // (lo, hi) pair of one tap is one 32-bit word of the staged tile, i.e. one register of the MFMA operand
    const int tap = 8 * s + 4 * (lane >> 5) + (j >> 1);
    if (tap < 27) v = w[co * 27 + tap] * ((j & 1) ? 256.f : 1.f) * P::STEM_SCALE * wscale;
    out[i] = (uint16_t)(P::pack2(v, 0.f) & 0xffffu);
}
__global__ void scale_copy_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * f;
}
// deconv: out[((par*CB + cb)*KP + kp)*64 + lane][j] = W[cin = kp*16 + 8*(lane>>5) + j][cout = cb*32 + (lane&31)][par]
template <class P>
__global__ void pack_deconv_w_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cin, int cout) {
    const int KP = cin / 16, CB = cout / 32;
    const long long n = (long long)cin * cout * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        long long r = i >> 9;
        const int kp = (int)(r % KP);
        r /= KP;
        const int cb = (int)(r % CB);
        const int par = (int)(r / CB);
        const int co = cb * 32 + (lane & 31);
        const int ci = kp * 16 + 8 * (lane >> 5) + j;
        const float v = w[((long long)ci * cout + co) * 8 + par];
        out[i] = (uint16_t)(P::pack2(v, 0.f) & 0xffffu);
    }
}

// ---------------------------------------------------------------------------------------------------
// stem: Conv3d(1 -> C0, k3, p1) in fp32 on the VALU, straight from the uint16 volume window
// (gather + cast + flip of inference/sliding_window_inferer.py:181-195,218-219 fused in) or from an
// fp32 patch.  Writes raw (pre-norm) bf16 + per-block partial sums for the InstanceNorm.
// ---------------------------------------------------------------------------------------------------
constexpr int STEM_ZR = 4;  // z-run per thread

template <class P, bool FROM_VOLUME>
__global__ void __launch_bounds__(256) stem_conv_kernel(const float* __restrict__ xf, const uint16_t* __restrict__ vol,
                                                        int Yp, int Xp, const int* __restrict__ starts, int flip_dim,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        uint4* __restrict__ out, float* __restrict__ partials, int D,
                                                        int H, int W, float oscale) {
    __shared__ float wl[27 * 32];
    __shared__ float red[4][64];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) {
        const int t = i >> 5, co = i & 31;
        wl[i] = w[co * 27 + t];
    }
    __syncthreads();
    const int n = blockIdx.z;
    const int hw = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool pvalid = p < hw;
    const int y = pvalid ? p / W : 0, x = pvalid ? p % W : 0;
    const int zb = blockIdx.y * STEM_ZR;
    int z0 = 0, y0 = 0, x0 = 0;
    if (FROM_VOLUME) {
        z0 = starts[3 * n];
        y0 = starts[3 * n + 1];
        x0 = starts[3 * n + 2];
    }
    auto fetch = [&](int zz, int yy, int xx) -> float {
        if ((unsigned)zz >= (unsigned)D || (unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) return 0.f;
        if (FROM_VOLUME) {
            if (flip_dim == 2) zz = D - 1 - zz;
            if (flip_dim == 3) yy = H - 1 - yy;
            if (flip_dim == 4) xx = W - 1 - xx;
            return (float)vol[((long long)(z0 + zz) * Yp + (y0 + yy)) * Xp + (x0 + xx)];
        }
        return xf[(long long)n * D * hw + ((long long)zz * H + yy) * W + xx];
    };
    float s[32], q[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) s[c] = q[c] = 0.f;
#pragma unroll 1
    for (int zi = 0; zi < STEM_ZR; ++zi) {
        const int z = zb + zi;
        if (z >= D || !pvalid) continue;
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = bias[c];
#pragma unroll 1
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = fetch(z + dz - 1, y + dy - 1, x + dx - 1);
                    const float4* wr = reinterpret_cast<const float4*>(wl + ((dz * 3 + dy) * 3 + dx) * 32);
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const float4 ww = wr[c4];
                        acc[4 * c4 + 0] = fmaf(v, ww.x, acc[4 * c4 + 0]);
                        acc[4 * c4 + 1] = fmaf(v, ww.y, acc[4 * c4 + 1]);
                        acc[4 * c4 + 2] = fmaf(v, ww.z, acc[4 * c4 + 2]);
                        acc[4 * c4 + 3] = fmaf(v, ww.w, acc[4 * c4 + 3]);
                    }
                }
        if (oscale != 1.0f) {  // (STEM_SCALE * 2^-shift of layer 0)
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c] *= oscale;
        }
        const long long vox = (long long)D * hw;
        const long long o = (long long)z * hw + p;
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            uint4 u;
            u.x = P::pack2(acc[8 * c8 + 0], acc[8 * c8 + 1]);
            u.y = P::pack2(acc[8 * c8 + 2], acc[8 * c8 + 3]);
            u.z = P::pack2(acc[8 * c8 + 4], acc[8 * c8 + 5]);
            u.w = P::pack2(acc[8 * c8 + 6], acc[8 * c8 + 7]);
            out[((long long)n * 4 + c8) * vox + o] = u;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            s[c] += acc[c];
            q[c] = fmaf(acc[c], acc[c], q[c]);
        }
    }
    // block reduction -> partials[n][block][32][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        float a = s[c], b = q[c];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (lane == 0) {
            red[wave][2 * c] = a;
            red[wave][2 * c + 1] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const long long blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        const long long nblk = (long long)gridDim.x * gridDim.y;
        partials[((long long)n * nblk + blk) * 64 + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// MFMA stem for the fused sliding-window path: the uint16 window is staged (with flip and the
// zero padding of the window border) as a halo tile in LDS, every voxel already split into its (lo, hi) bytes as a
// pair of 16-bit floats (exact in bf16 and fp16); every lane gathers the taps of its voxel - one ds_read_b32 per tap
// is one register of the operand, no VALU - and feeds 4 MFMAs (K = 64) per 32-voxel block.
//   workgroup: 4 (z) x 8 (y) x 32 (x) output voxels; wave w = z-slice w, 8 row blocks
// ---------------------------------------------------------------------------------------------------
constexpr int SM_TZ = 4, SM_TY = 8, SM_TX = 32, SM_HZ = 6, SM_HY = 10, SM_HX = 34;
constexpr int SM_ZC = 8;                                   // z-chunks of SM_TZ planes one workgroup walks
constexpr int SM_NT = SM_HZ * SM_HY * SM_HX;               // halo tile voxels
constexpr int SM_NS = (SM_NT + 255) / 256;                 // staged voxels per thread

// MODE 0: store raw + statistics; MODE 1: statistics only (first pass of the two-pass stem); MODE 2: recompute,
// apply InstanceNorm scale/shift + Mish and store the ACTIVATED tensor (no raw tensor, no separate norm pass:
// the K = 64 MFMA work is cheap next to 268 MB of avoided traffic per 128^3 window).
// A workgroup owns an 8 x 32 (y, x) column and walks SM_ZC chunks of 4 planes: the per-thread staging addresses, the
// weights and the statistics registers are set up once per 8192 voxels, the halo tile is double-buffered (the loads of
// the next chunk fly during the MFMAs of this one, one barrier per chunk), one reduction at the end.
template <class P, int MODE>
__global__ void __launch_bounds__(256) stem_mfma_kernel(const uint16_t* __restrict__ vol, int Yp, int Xp,
                                                        const int* __restrict__ starts, int flip_dim,
                                                        const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                        uint4* __restrict__ out, float* __restrict__ partials,
                                                        const float2* __restrict__ ss, int D, int H, int W, int tilesY,
                                                        int tilesX) {
    // the halo tile, already split: word = (lo byte, hi byte) of the voxel as two 16-bit floats (exact in bf16 and fp16)
    __shared__ unsigned tile[2][SM_NT];
    __shared__ float red[4 * 64];
    const int n = blockIdx.z;
    const int t = dlv_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = t % tilesX, ty = (t / tilesX) % tilesY, tg = t / (tilesX * tilesY);
    const int zbase = tg * (SM_ZC * SM_TZ), y0 = ty * SM_TY, x0 = tx * SM_TX;
    const int nzc = min(SM_ZC, (D - zbase + SM_TZ - 1) / SM_TZ);
    const int wz = starts[3 * n], wy = starts[3 * n + 1], wx = starts[3 * n + 2];
    // this thread's staged voxels: in-plane offset into the volume (window origin, flip and y/x validity folded in) and
    // the halo plane; only the z coordinate moves from chunk to chunk
    int soff[SM_NS], szh[SM_NS];
#pragma unroll
    for (int k = 0; k < SM_NS; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int xh = i % SM_HX, yh = (i / SM_HX) % SM_HY, zh = i / (SM_HX * SM_HY);
        int gy = y0 + yh - 1, gx = x0 + xh - 1;
        const bool ok = i < SM_NT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        if (flip_dim == 3) gy = H - 1 - gy;
        if (flip_dim == 4) gx = W - 1 - gx;
        soff[k] = ok ? (wy + gy) * Xp + (wx + gx) : -1;
        szh[k] = zh - 1;
    }
    const long long plane = (long long)Yp * Xp;
    unsigned sv[SM_NS];
    auto stage_load = [&](int zc) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < SM_NS; ++k) {
            int gz = zbase + zc * SM_TZ + szh[k];
            const bool ok = soff[k] >= 0 && (unsigned)gz < (unsigned)D;
            if (flip_dim == 2) gz = D - 1 - gz;
            sv[k] = ok ? (unsigned)vol[(long long)(wz + gz) * plane + soff[k]] : 0u;
        }
    };
    auto stage_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < SM_NS; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < SM_NT) tile[buf][i] = P::pack2((float)(sv[k] & 255u), (float)(sv[k] >> 8));
        }
    };
    stage_load(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    uint4 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = AS_FRAG(wpk[s * 64 + lane]);
    f32x16 bsv;  // the bias is the C operand of the first MFMA of every row
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bsv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * h] * P::STEM_SCALE;
        ssum[r] = ssq[r] = 0.f;
    }
    float nsc[16], nsh[16];
    if (MODE == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 v = ss[n * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
            nsc[r] = v.x;
            nsh[r] = v.y;
        }
    }
    stage_store(0);
    __syncthreads();
    const long long vox = (long long)D * H * W;
    // per-lane LDS offsets of this lane's 16 taps (k-step s, register q: tap 8s + 4h + q), row 0 of buffer 0; the row
    // loop is unrolled so that the row offset is an immediate of the ds_read_b32
    int tap_off[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int tp = 8 * (i >> 2) + 4 * h + (i & 3);
        const int tt = tp < 27 ? tp : 0;  // padding slots: any finite value (their weights are 0)
        tap_off[i] = (wave * SM_HY * SM_HX + col) + ((tt / 9) * SM_HY + (tt / 3) % 3) * SM_HX + tt % 3;
    }
#pragma unroll 1
    for (int zc = 0; zc < nzc; ++zc) {
        if (zc + 1 < nzc) stage_load(zc + 1);
        const unsigned* tl = tile[zc & 1];
        const int oz = zbase + zc * SM_TZ + wave;
#pragma unroll
        for (int row = 0; row < SM_TY; ++row) {
            unsigned b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = tl[tap_off[i] + row * SM_HX];
            f32x16 acc = P::mfma(a[0], AS_FRAG(make_uint4(b[0], b[1], b[2], b[3])), bsv, 0, 0, 0);
            acc = P::mfma(a[1], AS_FRAG(make_uint4(b[4], b[5], b[6], b[7])), acc, 0, 0, 0);
            acc = P::mfma(a[2], AS_FRAG(make_uint4(b[8], b[9], b[10], b[11])), acc, 0, 0, 0);
            acc = P::mfma(a[3], AS_FRAG(make_uint4(b[12], b[13], b[14], b[15])), acc, 0, 0, 0);
            const int oy = y0 + row, ox = x0 + col;
            const bool ok = oz < D && oy < H && ox < W;
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = acc[r];
            if (MODE != 2 && ok) {  // one exec-masked block (per-element selects cost two v_cndmask per value)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ssum[r] += val[r];
                    ssq[r] = fmaf(val[r], val[r], ssq[r]);
                }
            }
            if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t m = mish_fast2(fma2(f32x2_t{val[r], val[r + 1]}, f32x2_t{nsc[r], nsc[r + 1]}, f32x2_t{nsh[r], nsh[r + 1]}));
                    val[r] = m.x;
                    val[r + 1] = m.y;
                }
            }
            if (MODE != 1 && ok) {
                const long long o = ((long long)oz * H + oy) * W + ox;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = P::pack2(val[4 * g + 0], val[4 * g + 1]);
                    u.y = P::pack2(val[4 * g + 2], val[4 * g + 3]);
                    uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * 4 + g) * vox + o);
                    dlv_st8<true>(dst + h, u);
                }
            }
        }
        if (zc + 1 < nzc) stage_store((zc + 1) & 1);
        __syncthreads();
    }
    if (MODE == 2) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float sa = ssum[r], sb = ssq[r];
        sa = dlv_half_sum32(sa);  // DPP adds; totals valid in lanes 16-31 / 48-63
        sb = dlv_half_sum32(sb);
        if (col == 31) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
            red[(wave * 32 + co) * 2] = sa;
            red[(wave * 32 + co) * 2 + 1] = sb;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        partials[((long long)n * gridDim.x + t) * 64 + i] = red[i] + red[64 + i] + red[128 + i] + red[192 + i];
    }
}

// ---------------------------------------------------------------------------------------------------
// generic 3x3x3 convolution, implicit GEMM on MFMA
//   workgroup: 256 output voxels (4 z-slices x 64 voxels) x 32*NCB output channels
//   wave w   : z-slice w, two 32-voxel blocks, NCB cout blocks  -> 2*NCB accumulator tiles
//   loop     : input channels in slabs of 32 (halo tile staged in LDS) x 27 taps x 2 k-steps
// ---------------------------------------------------------------------------------------------------
template <int TX>
struct ConvTile {
    static constexpr int TZ = 4;
    static constexpr int TY = 64 / TX;       // 4 (TX=16) or 8 (TX=8)
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    static constexpr int SLAB = 4 * HZ * HY * HX;  // uint4 elements per 32-channel slab
    static constexpr int RV = 32 / TX;       // rows per 32-voxel block
};

template <class P, int NCB, int TX, bool WLDS>
__global__ void __launch_bounds__(256) conv3_mfma_kernel(const uint4* __restrict__ in1, int c1_8,
                                                         const uint4* __restrict__ in2, int c2_8,
                                                         const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                         uint4* __restrict__ out, float* __restrict__ partials, int cout,
                                                         int D, int H, int W, int tilesY, int tilesX) {
    using T = ConvTile<TX>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* slab = reinterpret_cast<uint4*>(smem_raw);
    uint4* wlds = slab + T::SLAB;  // WLDS: this slab's weights, [cb][tap][k-step][lane]
    const int n = blockIdx.z;
    const int tile = dlv_xcd_tile(blockIdx.x, gridDim.x);
    const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY, tz = tile / (tilesX * tilesY);
    const int z0 = tz * T::TZ, y0 = ty * T::TY, x0 = tx * TX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int vr = col / TX, vx = col % TX;  // row / x of this lane's voxel inside a 32-voxel block
    const int cin8 = c1_8 + c2_8;
    const int KP = cin8 / 2;
    const int cbg0 = blockIdx.y * NCB;
    const long long vox = (long long)D * H * W;

    f32x16 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][v][r] = 0.f;

    // per-lane LDS element offset of tap (0,0,0) for the two voxel blocks, chunk h of k-step 0
    int lbase[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) lbase[v] = ((h * T::HZ + wave) * T::HY + (v * T::RV + vr)) * T::HX + vx;

    const int nslab = cin8 / 4;
    // staging map of one 32-channel slab (the same for every slab: only the base pointer moves): element i of the
    // halo tile <- chunk c, voxel (gz,gy,gx).  The next slab is fetched into registers while the current one is being
    // multiplied (issue early / write late), so the HBM/L2 latency of the staging no longer sits between two MFMA phases.
    constexpr int NPF = (T::SLAB + 255) / 256;
    int poff[NPF];
    unsigned pvalid = 0;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int i = threadIdx.x + 256 * j;
        poff[j] = 0;
        if (i < T::SLAB) {
            const int xh = i % T::HX;
            int r = i / T::HX;
            const int yh = r % T::HY;
            r /= T::HY;
            const int zh = r % T::HZ;
            const int c = r / T::HZ;
            const int gz = z0 + zh - 1, gy = y0 + yh - 1, gx = x0 + xh - 1;
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                poff[j] = (int)((long long)c * vox + ((long long)gz * H + gy) * W + gx);
                pvalid |= 1u << j;
            }
        }
    }
    uint4 pf[NPF];
    auto fetch_slab = [&](int sl) __attribute__((always_inline)) {
        const int cg = sl * 4;  // a 32-channel slab lies entirely in one of the two sources (c1 % 32 == 0)
        const uint4* src = cg < c1_8 ? in1 + ((long long)n * c1_8 + cg) * vox : in2 + ((long long)n * c2_8 + (cg - c1_8)) * vox;
#pragma unroll
        for (int j = 0; j < NPF; ++j) pf[j] = src[poff[j]];
    };
    fetch_slab(0);
    // weight fragments straight from L2 (!WLDS) run through a rolling queue PD k-steps deep that continues across slab
    // boundaries: the load of step g + PD is issued when step g's fragments are consumed, so an L2 round trip is covered
    // by PD groups of MFMAs instead of sitting in front of each group (128^3 windows, 16 per launch: 128+128->64 at 32^3
    // 357 -> 296 us, 256->128 at 16^3 196 -> 166 us).  The same queue for the LDS operand made it slower (registers).
    constexpr int PD = WLDS ? 1 : (NCB >= 4 ? 2 : 6);  // 54 k-steps per slab: PD divides 54 (NCB 4: 256 VGPRs allow no more)
    uint4 aq[PD][NCB];
    const long long nsteps = (long long)nslab * 54;
    auto wfetch = [&](long long g, uint4 (&dst)[NCB]) __attribute__((always_inline)) {
        const int sl2 = (int)(g / 54), st = (int)(g % 54);
        const int t = st >> 1, kp = sl2 * 2 + (st & 1);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) dst[cb] = wpk[(((long long)(cbg0 + cb) * 27 + t) * KP + kp) * 64 + lane];
    };
    if (!WLDS) {
#pragma unroll
        for (int q = 0; q < PD; ++q)
            if (q < nsteps) wfetch(q, aq[q]);
    }
    for (int sl = 0; sl < nslab; ++sl) {
        __syncthreads();  // previous slab fully consumed
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < T::SLAB) slab[i] = ((pvalid >> j) & 1u) ? pf[j] : make_uint4(0, 0, 0, 0);
        }
        if (WLDS) {
            // the slab's weights are fetched cooperatively in one coalesced sweep (deep levels have few
            // workgroups: per-k-step fragment loads from L2 are latency-bound there)
            for (int i = threadIdx.x; i < NCB * 27 * 2 * 64; i += 256) {
                const int l = i & 63, ks = (i >> 6) & 1, r = i >> 7;
                const int t = r % 27, cb = r / 27;
                wlds[i] = wpk[(((long long)(cbg0 + cb) * 27 + t) * KP + sl * 2 + ks) * 64 + l];
            }
        }
        __syncthreads();
        if (sl + 1 < nslab) fetch_slab(sl + 1);  // in flight during this slab's MFMAs
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int t = (kz * 3 + ky) * 3 + kx;
                    const int toff = (kz * T::HY + ky) * T::HX + kx;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int st = t * 2 + ks;
                        uint4 a[NCB];
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
                            a[cb] = AS_FRAG(WLDS ? wlds[((cb * 27 + t) * 2 + ks) * 64 + lane] : aq[st % PD][cb]);
                        if (!WLDS) {
                            const long long g = (long long)sl * 54 + st + PD;
                            if (g < nsteps) wfetch(g, aq[st % PD]);
                        }
                        uint4 b[2];
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const uint4 u = slab[lbase[v] + toff + ks * 2 * T::HZ * T::HY * T::HX];
                            b[v] = AS_FRAG(u);
                        }
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                            for (int v = 0; v < 2; ++v)
                                acc[cb][v] = P::mfma(a[cb], b[v], acc[cb][v], 0, 0, 0);
                    }
                }
    }

    // ---- epilogue: bias, bf16 store, InstanceNorm partial sums -------------------------------------
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);  // [4 waves][NCB*32][2]
    const int oz = z0 + wave;
    const int cout8 = cout / 8;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        float bs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bs[r] = bias[(cbg0 + cb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        float s[16], q[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = q[r] = 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int oy = y0 + v * T::RV + vr, ox = x0 + vx;
            const bool ok = oz < D && oy < H && ox < W;
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                val[r] = acc[cb][v][r] + bs[r];
                if (ok) {
                    s[r] += val[r];
                    q[r] = fmaf(val[r], val[r], q[r]);
                }
            }
            if (ok) {
                const long long o = ((long long)oz * H + oy) * W + ox;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = P::pack2(val[4 * g + 0], val[4 * g + 1]);
                    u.y = P::pack2(val[4 * g + 2], val[4 * g + 3]);
                    uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * cout8 + (cbg0 + cb) * 4 + g) * vox + o);
                    dst[h] = u;
                }
            }
        }
        // reduce over the 32 voxels (lanes with equal h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = s[r], b = q[r];
            a = dlv_half_sum32(a);  // DPP adds; totals valid in lanes 16-31 / 48-63
            b = dlv_half_sum32(b);
            if (col == 31) {
                const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                red[(wave * NCB * 32 + co) * 2] = a;
                red[(wave * NCB * 32 + co) * 2 + 1] = b;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < NCB * 32 * 2) {
        const int i = threadIdx.x;
        const float v = red[i] + red[NCB * 64 + i] + red[2 * NCB * 64 + i] + red[3 * NCB * 64 + i];
        const int co = cbg0 * 32 + (i >> 1);
        partials[(((long long)n * gridDim.x + tile) * cout + co) * 2 + (i & 1)] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// InstanceNorm statistics: partial sums -> per (n,c) scale/shift   y = x*scale + shift
// one wave per (n,c); fixed summation order (bitwise reproducible)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) stats_finalize_kernel(const float* __restrict__ partials, int nparts, int C,
                                                            double inv_count, float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float2* __restrict__ ss,
                                                            int* __restrict__ range_flag, int layer) {
    const int c = blockIdx.x % C, n = blockIdx.x / C;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) {
        const float2 v = *reinterpret_cast<const float2*>(partials + (((long long)n * nparts + i) * C + c) * 2);
        s += v.x;
        q += v.y;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        q += __shfl_down(q, o, 64);
    }
    if (threadIdx.x == 0) {
        // Range guard: a raw value beyond the 16-bit format's range is stored as Inf, the next normalisation pass turns it into
        // Inf or NaN, and the next convolution's sums - these - stop being finite.  Detected here for free; without it the mask
        // silently becomes zeros (NaN >= 0 is false).  The first such layer wins (atomicMax of 100 - layer).
        if (!(fabs(s) <= 1.0e300 && fabs(q) <= 1.0e300)) atomicMax(range_flag, 100 - layer);
        const double mean = s * inv_count;
        double var = q * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        // ... and which layer is the large one: the sums are those of the fp32 accumulators, so they stay finite when the STORED
        // 16-bit value overflows.  The largest |mean| + 8 sigma of every block is recorded (one atomic per (window, channel) of a
        // 64-thread workgroup); after a DLV_ERANGE the host reads it as the hint for dlv_unet_set_conv_shift: how far to move a
        // block that reported > 4096, and which blocks are too SMALL to be moved at all (positive floats order like their bit patterns)
        // (a plain read first: one atomic per (window, channel) on ONE address cost 40-160 us per launch - 16 k workgroups of a
        // 64-window batch queue up on it, +24 % on a pass of 64 x 64 x 32 windows, measured; after the first few workgroups the
        // word already holds a larger value and the others only read it.  A stale read is harmless: atomicMax decides.)
        const float peak = (float)(fabs(mean) + 8.0 * sqrt(var));
        if (peak > 0.f && peak < 3.0e38f && __float_as_int(peak) > __builtin_nontemporal_load(range_flag + 1 + layer))
            atomicMax(range_flag + 1 + layer, __float_as_int(peak));
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gamma[c];
        ss[n * C + c] = make_float2(sc, beta[c] - (float)mean * sc);
    }
}

// ---------------------------------------------------------------------------------------------------
// InstanceNorm apply + Mish (+ MaxPool3d(2) into a second tensor), in place on the raw bf16 tensor
// ---------------------------------------------------------------------------------------------------
// P: the format the raw tensor is stored in; PO: the format of the activated value (the mixed 16-bit mode changes format between
// levels 0 and 1: DLV_PREC_BF16 keeps fp16 at full resolution, section 5 of DESIGN.md)
template <class P, class PO = P>
__device__ __forceinline__ uint4 norm_mish8(uint4 u, const float* sc, const float* sh, float* mx) {
    float v[8] = {P::lo(u.x), P::hi(u.x), P::lo(u.y), P::hi(u.y), P::lo(u.z), P::hi(u.z), P::lo(u.w), P::hi(u.w)};
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2_t m = mish_fast2(fma2(f32x2_t{v[k], v[k + 1]}, f32x2_t{sc[k], sc[k + 1]}, f32x2_t{sh[k], sh[k + 1]}));
        v[k] = m.x;
        v[k + 1] = m.y;
        if (mx) {
            mx[k] = fmaxf(mx[k], v[k]);
            mx[k + 1] = fmaxf(mx[k + 1], v[k + 1]);
        }
    }
    uint4 r;
    r.x = PO::pack2(v[0], v[1]);
    r.y = PO::pack2(v[2], v[3]);
    r.z = PO::pack2(v[4], v[5]);
    r.w = PO::pack2(v[6], v[7]);
    return r;
}

// WB: write the activated tensor back in place.  POOL && !WB: only the pooled tensor is produced - the full-resolution
// tensor stays raw and every consumer applies scale/shift + Mish while it loads (conv_zreg.hip's staging, the
// transposed conv below, the final 1x1x1 conv)
// PW: format of the written-back tensor, PQ: format of the pooled tensor (both P except at the format seam of the mixed mode)
template <class P, bool POOL, bool WB, bool NT = false, class PW = P, class PQ = P>
__global__ void __launch_bounds__(256) norm_mish_kernel(uint4* __restrict__ x, const float2* __restrict__ ss, int C,
                                                        int D, int H, int W, uint4* __restrict__ pooled) {
    const int c8 = blockIdx.y, n = blockIdx.z;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 v = ss[n * C + c8 * 8 + k];
        sc[k] = v.x;
        sh[k] = v.y;
    }
    const long long vox = (long long)D * H * W;
    uint4* p = x + ((long long)n * (C / 8) + c8) * vox;
    if (!POOL) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vox; i += (long long)gridDim.x * 256)
            dlv_st16<NT>(p + i, norm_mish8<P, PW>(dlv_ld16<NT>(p + i), sc, sh, nullptr));
    } else {
        const int d2 = D / 2, h2 = H / 2, w2 = W / 2;
        const long long pv = (long long)d2 * h2 * w2;
        uint4* q = pooled + ((long long)n * (C / 8) + c8) * pv;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pv; i += (long long)gridDim.x * 256) {
            const int xx = (int)(i % w2), yy = (int)((i / w2) % h2), zz = (int)(i / ((long long)w2 * h2));
            float mx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mx[k] = -INFINITY;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const long long o = ((long long)(2 * zz + a) * H + (2 * yy + b)) * W + 2 * xx;
                    // (default cache policy here: a wave's two loads / stores each touch every other 16 bytes of the same lines -
                    // with `nt` the second one misses again: 1060 -> 1340 us per forward)
                    const uint4 r0 = norm_mish8<P, PW>(p[o], sc, sh, mx), r1 = norm_mish8<P, PW>(p[o + 1], sc, sh, mx);
                    if (WB) {
                        p[o] = r0;
                        p[o + 1] = r1;
                    }
                }
            uint4 r;
            r.x = PQ::pack2(mx[0], mx[1]);
            r.y = PQ::pack2(mx[2], mx[3]);
            r.z = PQ::pack2(mx[4], mx[5]);
            r.w = PQ::pack2(mx[6], mx[7]);
            q[i] = r;
        }
    }
}

// The pooling pass by full lines (W % 64 == 0: levels 0 and 1): a wave owns 64 consecutive fine voxels of the four rows
// (2 planes x 2 rows) under 32 pooled voxels - every load / store instruction covers one contiguous KiB (the kernel above reads
// every other 16 bytes per instruction and needs the lines to survive in cache between its two loads), so the non-temporal policy
// applies; the x pair is reduced with one DPP max per value, even lanes store the pooled voxel.  Same values bit for bit (max and
// the 16-bit rounding commute).
template <class P, bool WB, bool NT, class PQ = P>
__global__ void __launch_bounds__(256) norm_mish_pool_rows_kernel(uint4* __restrict__ x, const float2* __restrict__ ss, int C, int D, int H,
                                                                  int W, uint4* __restrict__ pooled) {
    const int c8 = blockIdx.y, n = blockIdx.z;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 v = ss[n * C + c8 * 8 + k];
        sc[k] = v.x;
        sh[k] = v.y;
    }
    const long long vox = (long long)D * H * W;
    const int d2 = D / 2, h2 = H / 2, w2 = W / 2, nseg = W / 64;
    uint4* p = x + ((long long)n * (C / 8) + c8) * vox;
    uint4* q = pooled + ((long long)n * (C / 8) + c8) * ((long long)d2 * h2 * w2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long items = (long long)d2 * h2 * nseg;
    for (long long it = (long long)blockIdx.x * 4 + wave; it < items; it += (long long)gridDim.x * 4) {
        const int xs = (int)(it % nseg), yy = (int)((it / nseg) % h2), zz = (int)(it / ((long long)nseg * h2));
        uint4 u[4];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) u[ab] = dlv_ld16<NT>(p + ((long long)(2 * zz + (ab >> 1)) * H + (2 * yy + (ab & 1))) * W + xs * 64 + lane);
        float mx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) mx[k] = -INFINITY;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const uint4 r = norm_mish8<P>(u[ab], sc, sh, mx);
            if (WB) dlv_st16<NT>(p + ((long long)(2 * zz + (ab >> 1)) * H + (2 * yy + (ab & 1))) * W + xs * 64 + lane, r);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)  // the neighbour of the x pair: quad_perm [1,0,3,2]
            mx[k] = fmaxf(mx[k], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx[k]), 0xB1, 0xf, 0xf, true)));
        if (!(lane & 1)) {
            uint4 r;
            r.x = PQ::pack2(mx[0], mx[1]);
            r.y = PQ::pack2(mx[2], mx[3]);
            r.z = PQ::pack2(mx[4], mx[5]);
            r.w = PQ::pack2(mx[6], mx[7]);
            dlv_st16<NT>(q + ((long long)zz * h2 + yy) * w2 + xs * 32 + (lane >> 1), r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d k2 s2 on MFMA: for each of the 8 output parities a (Cin x Cout) channel GEMM
//   wave: 32 consecutive input voxels (B fragments straight from HBM, no LDS), all parities/couts
// ---------------------------------------------------------------------------------------------------
template <class P, int KP>  // Cin / 16
__global__ void __launch_bounds__(256) deconv2_mfma_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk,
                                                           const float* __restrict__ bias, uint4* __restrict__ out,
                                                           int cout, int D, int H, int W) {
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long vox = (long long)D * H * W;
    const long long v = ((long long)blockIdx.x * 4 + wave) * 32 + col;
    const bool ok = v < vox;
    const long long vc = ok ? v : 0;
    uint4 b[KP];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        uint4 u = in[((long long)n * (2 * KP) + 2 * kp + h) * vox + vc];
        if (!ok) u = make_uint4(0, 0, 0, 0);
        b[kp] = AS_FRAG(u);
    }
    const int x = (int)(vc % W), y = (int)((vc / W) % H), z = (int)(vc / ((long long)W * H));
    const int CB = cout / 32, cout8 = cout / 8;
    const int OH = 2 * H, OW = 2 * W;
    const long long ovox = vox * 8;
    for (int par = 0; par < 8; ++par) {
        const long long o = ((long long)(2 * z + (par >> 2)) * OH + (2 * y + ((par >> 1) & 1))) * OW + 2 * x + (par & 1);
        for (int cb = 0; cb < CB; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias[cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                const uint4 u = wpk[(((long long)par * CB + cb) * KP + kp) * 64 + lane];
                acc = P::mfma(AS_FRAG(u), b[kp], acc, 0, 0, 0);
            }
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = P::pack2(acc[4 * g + 0], acc[4 * g + 1]);
                    u.y = P::pack2(acc[4 * g + 2], acc[4 * g + 3]);
                    uint2* dst = reinterpret_cast<uint2*>(out + ((long long)n * cout8 + cb * 4 + g) * ovox + o);
                    dst[h] = u;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d k2 s2, row-contiguous form: the 32 MFMA columns are 32 CONSECUTIVE OUTPUT voxels of one
// output row (input voxel = column >> 1, x-parity = column & 1).  The parity-dependent weights are applied
// with two MFMAs per k-step on parity-masked copies of the input fragment, and permlane32_swap joins the
// two half-wave channel quads, so that every store is 16 bytes per lane and 512 contiguous bytes per
// half-wave (the per-parity form above writes 8-byte halves at a stride of two voxels).
// ---------------------------------------------------------------------------------------------------
template <class P, int KP>
__global__ void __launch_bounds__(256) deconv2_rows_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk,
                                                           const float* __restrict__ bias, uint4* __restrict__ out,
                                                           int cout, int D, int H, int W, int segs,
                                                           const float2* __restrict__ ss) {
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long vox = (long long)D * H * W;
    const long long item = (long long)blockIdx.x * 4 + wave;  // (z, y, x-segment of 16 input voxels)
    const long long nitems = (long long)D * H * segs;
    if (item >= nitems) return;
    const int sg = (int)(item % segs), y = (int)((item / segs) % H), z = (int)(item / ((long long)segs * H));
    const int xi = sg * 16 + (col >> 1);
    const bool ok = xi < W;
    const bool odd = col & 1;
    const long long vin = ((long long)z * H + y) * W + (ok ? xi : 0);
    uint4 b0[KP], b1[KP];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        uint4 u = in[((long long)n * (2 * KP) + 2 * kp + h) * vox + vin];
        if (ss) {  // the input is the raw output of a conv: its InstanceNorm + Mish are applied here (wave-uniform branch)
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 v = ss[n * (16 * KP) + (2 * kp + h) * 8 + k];
                sc[k] = v.x;
                sh[k] = v.y;
            }
            u = norm_mish8<P>(u, sc, sh, nullptr);
        }
        b0[kp] = AS_FRAG((ok && !odd) ? u : zero4);
        b1[kp] = AS_FRAG((ok && odd) ? u : zero4);
    }
    const int CB = cout / 32, cout8 = cout / 8;
    const int OH = 2 * H, OW = 2 * W;
    const long long ovox = vox * 8;
    const int ox = 2 * sg * 16 + col;
    for (int ab = 0; ab < 4; ++ab) {
        const long long o = ((long long)(2 * z + (ab >> 1)) * OH + (2 * y + (ab & 1))) * OW + ox;
        for (int cb = 0; cb < CB; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias[cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                const uint4 w0 = wpk[(((long long)(ab * 2 + 0) * CB + cb) * KP + kp) * 64 + lane];
                const uint4 w1 = wpk[(((long long)(ab * 2 + 1) * CB + cb) * KP + kp) * 64 + lane];
                acc = P::mfma(AS_FRAG(w0), b0[kp], acc, 0, 0, 0);
                acc = P::mfma(AS_FRAG(w1), b1[kp], acc, 0, 0, 0);
            }
            unsigned px[4], py[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                px[g] = P::pack2(acc[4 * g + 0], acc[4 * g + 1]);
                py[g] = P::pack2(acc[4 * g + 2], acc[4 * g + 3]);
            }
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                // lanes 0-31 end up with all 8 channels of chunk gp, lanes 32-63 with chunk gp+1
                const auto sx = __builtin_amdgcn_permlane32_swap(px[gp], px[gp + 1], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(py[gp], py[gp + 1], false, false);
                if (ok) out[((long long)n * cout8 + cb * 4 + gp + h) * ovox + o] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        }
    }
}

// The same op for the two large levels (Cout = 32, Cin <= 64: 2.1 GB of output per 16 windows at the top level, an
// HBM-write-bound kernel): all 8 taps' weights stay in registers (32 x KP VGPRs), every wave walks DC_IPW consecutive
// row segments and fetches the next segment's input while the MFMAs and stores of the current one are in flight - the
// per-segment kernel above re-reads 16 KB of weights through L1 for 8 KB of output and exposes every load latency.
constexpr int DC_IPW = 8;  // row segments (16 input voxels -> 4 x 32 output voxels x 32 channels = 8 KB) per wave

template <class P, int KP>
__global__ void __launch_bounds__(256) deconv2_regw_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk,
                                                           const float* __restrict__ bias, uint4* __restrict__ out, int D,
                                                           int H, int W, int segs, const float2* __restrict__ ss) {
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform by construction: lets z / y /
                                                                                 // the row offsets live in SGPRs)
    const int h = lane >> 5, col = lane & 31;
    const long long vox = (long long)D * H * W;
    const unsigned nitems = (unsigned)D * (unsigned)H * (unsigned)segs;  // row segments of one window: 32-bit index math
    const unsigned item0 = ((unsigned)blockIdx.x * 4u + (unsigned)wave) * (unsigned)DC_IPW;
    if (item0 >= nitems) return;
    const bool odd = col & 1;
    uint4 w0[4][KP], w1[4][KP];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab)
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            w0[ab][kp] = wpk[((long long)(ab * 2 + 0) * KP + kp) * 64 + lane];
            w1[ab][kp] = wpk[((long long)(ab * 2 + 1) * KP + kp) * 64 + lane];
        }
    f32x16 bsv;  // the bias is the C operand of each parity's first MFMA (no 16 moves per parity to seed an accumulator)
#pragma unroll
    for (int r = 0; r < 16; ++r) bsv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * h];
    float sc[KP][8], sh[KP][8];
    if (ss) {
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 v = ss[n * (16 * KP) + (2 * kp + h) * 8 + k];
                sc[kp][k] = v.x;
                sh[kp][k] = v.y;
            }
    }
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int OH = 2 * H, OW = 2 * W;
    const long long ovox = vox * 8;
    // output through a buffer resource over this sample's four chunks (4 * ovox * 16 B < 2^32: the launcher checks): the
    // chunk part of an address is an SGPR offset, the voxel part a 32-bit lane offset; lanes beyond the row end carry an
    // out-of-range offset and the hardware drops their store
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out + (long long)n * 4 * ovox, 0, (int)(unsigned)(4 * ovox * 16), 0x00020000);
    const unsigned chunk_b = (unsigned)ovox * 16u;
    const uint4* inb = in + ((long long)n * (2 * KP) + h) * vox;
    auto fetch = [&](unsigned item, uint4 (&u)[KP]) __attribute__((always_inline)) {
        const int sg = (int)(item % (unsigned)segs);
        const unsigned zy = item / (unsigned)segs;  // z * H + y
        const int xi = sg * 16 + (col >> 1);
        const long long vin = (long long)zy * W + (xi < W ? xi : 0);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) u[kp] = inb[(long long)(2 * kp) * vox + vin];
    };
    uint4 cur[KP], nxt[KP];
    fetch(item0, cur);
#pragma unroll 1
    for (int it = 0; it < DC_IPW; ++it) {
        const unsigned item = item0 + (unsigned)it;
        if (item >= nitems) break;
        if (item + 1 < nitems && it + 1 < DC_IPW) fetch(item + 1, nxt);
        const int sg = (int)(item % (unsigned)segs), y = (int)((item / (unsigned)segs) % (unsigned)H), z = (int)(item / ((unsigned)segs * (unsigned)H));
        const bool ok = sg * 16 + (col >> 1) < W;
        uint4 b0[KP], b1[KP];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            uint4 u = cur[kp];
            if (ss) u = norm_mish8<P>(u, sc[kp], sh[kp], nullptr);  // the input is a raw conv output (wave-uniform branch)
            b0[kp] = AS_FRAG((ok && !odd) ? u : zero4);
            b1[kp] = AS_FRAG((ok && odd) ? u : zero4);
        }
        const unsigned ox = (unsigned)(2 * sg * 16 + col);
        const unsigned lane_b = ok ? (ox + (unsigned)h * (unsigned)ovox) * 16u : 0xfffffff0u;  // chunk gp + h: h in the lane part
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const unsigned row_b = (unsigned)(((2 * z + (ab >> 1)) * OH + (2 * y + (ab & 1))) * OW) * 16u;  // wave-uniform
            f32x16 acc = P::mfma(AS_FRAG(w0[ab][0]), b0[0], bsv, 0, 0, 0);
            acc = P::mfma(AS_FRAG(w1[ab][0]), b1[0], acc, 0, 0, 0);
#pragma unroll
            for (int kp = 1; kp < KP; ++kp) {
                acc = P::mfma(AS_FRAG(w0[ab][kp]), b0[kp], acc, 0, 0, 0);
                acc = P::mfma(AS_FRAG(w1[ab][kp]), b1[kp], acc, 0, 0, 0);
            }
            unsigned px[4], py[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                px[g] = P::pack2(acc[4 * g + 0], acc[4 * g + 1]);
                py[g] = P::pack2(acc[4 * g + 2], acc[4 * g + 3]);
            }
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                // lanes 0-31 end up with all 8 channels of chunk gp, lanes 32-63 with chunk gp+1
                const auto sx = __builtin_amdgcn_permlane32_swap(px[gp], px[gp + 1], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(py[gp], py[gp + 1], false, false);
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(u32x4v{sx[0], sy[0], sx[1], sy[1]}, ors, (int)lane_b,
                                                       (int)(row_b + (unsigned)gp * chunk_b), 0);
            }
        }
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) cur[kp] = nxt[kp];
    }
}

// The deep levels (Cin >= 128): few voxels, many weights - the per-segment kernel re-reads all 8 x Cin x Cout weights for
// every 16 input voxels (524 KB per wave at 256 -> 128).  Here a wave keeps the fragments of ONE (output parity pair ab,
// 32-channel output block cb) in registers (2 x KP) and walks DW_IPW row segments with them; grid.y enumerates (ab, cb).
constexpr int DW_IPW = 4;

template <class P, int KP>
__global__ void __launch_bounds__(256) deconv2_wst_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk,
                                                          const float* __restrict__ bias, uint4* __restrict__ out, int cout,
                                                          int D, int H, int W, int segs, const float2* __restrict__ ss) {
    const int n = blockIdx.z;
    const int CB = cout / 32, cout8 = cout / 8;
    const int ab = blockIdx.y / CB, cb = blockIdx.y % CB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long vox = (long long)D * H * W;
    const long long nitems = (long long)D * H * segs;
    const long long item0 = ((long long)blockIdx.x * 4 + wave) * DW_IPW;
    if (item0 >= nitems) return;
    const bool odd = col & 1;
    uint4 w0[KP], w1[KP];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        w0[kp] = wpk[(((long long)(ab * 2 + 0) * CB + cb) * KP + kp) * 64 + lane];
        w1[kp] = wpk[(((long long)(ab * 2 + 1) * CB + cb) * KP + kp) * 64 + lane];
    }
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[r] = bias[cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int OH = 2 * H, OW = 2 * W;
    const long long ovox = vox * 8;
    const uint4* inb = in + ((long long)n * (2 * KP) + h) * vox;
#pragma unroll 1
    for (int it = 0; it < DW_IPW; ++it) {
        const long long item = item0 + it;
        if (item >= nitems) break;
        const int sg = (int)(item % segs), y = (int)((item / segs) % H), z = (int)(item / ((long long)segs * H));
        const int xi = sg * 16 + (col >> 1);
        const bool ok = xi < W;
        const long long vin = ((long long)z * H + y) * W + (ok ? xi : 0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bs[r];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            uint4 u = inb[(long long)(2 * kp) * vox + vin];
            if (ss) {  // the input is the raw output of a conv: its InstanceNorm + Mish are applied here (wave-uniform branch)
                float sc[8], sh[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float2 v = ss[n * (16 * KP) + (2 * kp + h) * 8 + k];
                    sc[k] = v.x;
                    sh[k] = v.y;
                }
                u = norm_mish8<P>(u, sc, sh, nullptr);
            }
            acc = P::mfma(AS_FRAG(w0[kp]), AS_FRAG((ok && !odd) ? u : zero4), acc, 0, 0, 0);
            acc = P::mfma(AS_FRAG(w1[kp]), AS_FRAG((ok && odd) ? u : zero4), acc, 0, 0, 0);
        }
        const long long o = ((long long)(2 * z + (ab >> 1)) * OH + (2 * y + (ab & 1))) * OW + 2 * sg * 16 + col;
        unsigned px[4], py[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            px[g] = P::pack2(acc[4 * g + 0], acc[4 * g + 1]);
            py[g] = P::pack2(acc[4 * g + 2], acc[4 * g + 3]);
        }
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
            const auto sx = __builtin_amdgcn_permlane32_swap(px[gp], px[gp + 1], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(py[gp], py[gp + 1], false, false);
            if (ok) out[((long long)n * cout8 + cb * 4 + gp + h) * ovox + o] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// final: InstanceNorm + Mish of the last block, Conv3d(C5 -> 1, k1), then either plain logits or
// the blend accumulate of inference/sliding_window_inferer.py:232-251 (acc[window] += logit, un-flipped)
// ---------------------------------------------------------------------------------------------------
template <class P, bool BLEND>
__global__ void __launch_bounds__(256) final_conv_kernel(const uint4* __restrict__ x, const float2* __restrict__ ss,
                                                         const float* __restrict__ wf, const float* __restrict__ bf,
                                                         float* __restrict__ logits, const int* __restrict__ starts,
                                                         int flip_dim, int Yp, int Xp, float scale, float* __restrict__ acc,
                                                         int D, int H, int W, const float* __restrict__ bw, float bmin,
                                                         float* __restrict__ wsum, int* __restrict__ range_flag) {
    const int n = blockIdx.y;
    bool bad = false;  // range guard of the last block's raw tensor (no later InstanceNorm would see it): a non-finite logit
    // per-sample scale/shift and the 32 weights are uniform over the workgroup: scalar loads, SGPR operands
    f32x2_t sc[16], sh[16], ww[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float2 v0 = ss[n * 32 + 2 * c], v1 = ss[n * 32 + 2 * c + 1];
        sc[c] = f32x2_t{v0.x, v1.x};
        // 96 uniform values exceed the SGPR file, and a packed FMA reads at most ONE scalar pair (constant bus): the scales stay
        // in SGPRs, shifts and weights live in VGPRs - no per-use v_mov_b64 / v_readlane of a spilled pair in the loop
        float h0 = v0.y, h1 = v1.y;
        asm volatile("" : "+v"(h0), "+v"(h1));
        sh[c] = f32x2_t{h0, h1};
        float w0 = wf[2 * c], w1 = wf[2 * c + 1];
        asm volatile("" : "+v"(w0), "+v"(w1));
        ww[c] = f32x2_t{w0, w1};
    }
    const long long vox = (long long)D * H * W;
    const float b0 = bf[0];
    int z0 = 0, y0 = 0, x0 = 0;
    if (BLEND) {
        z0 = starts[3 * n];
        y0 = starts[3 * n + 1];
        x0 = starts[3 * n + 2];
    }
    // Software pipeline: the four chunk words and (plain blend) the accumulator word of iteration i + 1 are in flight while the 32
    // Mish evaluations of iteration i run.  Without it a wave alternates between waiting for its loads and ~1700 cycles of
    // arithmetic, and neither the VALU (408 us of work per 16 windows) nor HBM (420 us) is kept busy: 588 us
    // (profiles/microbench/final_probe.hip: 447 us pipelined at 8 iterations per thread).
    // (32-bit voxel indices - the launcher refuses windows of 2^31 voxels - and the window coordinates advanced by the grid
    // stride with carries instead of three 64-bit divisions per voxel: those were a third of the loop's instructions)
    const unsigned nvox = (unsigned)vox, step = gridDim.x * 256u;
    const unsigned sx = step % (unsigned)W, sy = (step / (unsigned)W) % (unsigned)H, sz = step / ((unsigned)W * (unsigned)H);
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < nvox) {
        unsigned xx = i % (unsigned)W, yy = (i / (unsigned)W) % (unsigned)H, zz = i / ((unsigned)W * (unsigned)H);
        auto out_index = [&](unsigned z, unsigned y, unsigned xc) -> long long {
            const int zf = flip_dim == 2 ? D - 1 - (int)z : (int)z, yf = flip_dim == 3 ? H - 1 - (int)y : (int)y,
                      xf = flip_dim == 4 ? W - 1 - (int)xc : (int)xc;
            return ((long long)(z0 + zf) * Yp + (y0 + yf)) * Xp + (x0 + xf);
        };
        const bool plain = BLEND && !bw;  // (Gaussian weights: two read-modify-writes per voxel, not prefetched)
        const uint4* xb = x + (long long)n * 4 * vox;
        uint4 u[4];
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) u[c8] = dlv_ld16<true>(xb + (long long)c8 * vox + i);  // (read once, 64 B per voxel)
        long long o = BLEND ? out_index(zz, yy, xx) : 0;
        float av = plain ? acc[o] : 0.f;
        for (; i < nvox; i += step) {
            uint4 un[4] = {u[0], u[1], u[2], u[3]};
            long long on = o;
            float avn = 0.f;
            const unsigned zc = zz, yc = yy, xc = xx;  // this iteration's coordinates (Gaussian weights)
            const unsigned in = i + step;
            if (in < nvox && in > i) {
                xx += sx;
                const unsigned cx = xx >= (unsigned)W ? 1u : 0u;
                xx -= cx ? (unsigned)W : 0u;
                yy += sy + cx;
                const unsigned cy = yy >= (unsigned)H ? 1u : 0u;
                yy -= cy ? (unsigned)H : 0u;
                zz += sz + cy;
#pragma unroll
                for (int c8 = 0; c8 < 4; ++c8) un[c8] = dlv_ld16<true>(xb + (long long)c8 * vox + in);
                if (BLEND) on = out_index(zz, yy, xx);
                if (plain) avn = acc[on];
            }
            f32x2_t a2 = {b0, 0.f};
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                const unsigned uu[4] = {u[c8].x, u[c8].y, u[c8].z, u[c8].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x2_t v = {P::lo(uu[k]), P::hi(uu[k])};
                    a2 = fma2(mish_fast2(fma2(v, sc[4 * c8 + k], sh[4 * c8 + k])), ww[4 * c8 + k], a2);
                }
            }
            const float a = a2.x + a2.y;
            bad |= !(fabsf(a) <= 3.0e38f);
            if (!BLEND) {
                logits[(long long)n * vox + i] = a;
            } else if (bw) {  // Gaussian importance map, indexed in volume orientation (after the un-flip)
                const int zf = flip_dim == 2 ? D - 1 - (int)zc : (int)zc, yf = flip_dim == 3 ? H - 1 - (int)yc : (int)yc,
                          xf = flip_dim == 4 ? W - 1 - (int)xc : (int)xc;
                const float wgt = fmaxf(bw[zf] * bw[D + yf] * bw[D + H + xf], bmin) * scale;
                acc[o] += wgt * a;
                if (wsum) wsum[o] += wgt;
            } else {
                acc[o] = av + scale * a;
            }
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) u[c8] = un[c8];
            o = on;
            av = avn;
            if (in <= i) break;  // (32-bit wrap-around of the index)
        }
    }
    if (bad) atomicMax(range_flag, 100 - 18);
}

// ---------------------------------------------------------------------------------------------------
// debug / test conversions: fp32 NCDHW <-> bf16 chunk-planar
// ---------------------------------------------------------------------------------------------------
// MONAI UpCat's replicate padding (monai/networks/nets/basic_unet.py, UpCat.forward, is_pad=True; call site
// inference/inference.py:190-197): a level whose skip tensor has an ODD size gets an up-sampled tensor that is one voxel short in
// that dimension (2 * floor(n / 2) = n - 1); it is padded by one at the far end with the edge value.  Windows whose dimensions
// are multiples of 16 never come here.  Chunk-planar tensors: [n][C/8][D][H][W] of uint4.
__global__ void __launch_bounds__(256) replicate_pad_cp_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int Di, int Hi, int Wi,
                                                               int Do, int Ho, int Wo) {
    const long long vo = (long long)Do * Ho * Wo, vi = (long long)Di * Hi * Wi;
    const long long plane = blockIdx.y;  // (n, chunk)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vo; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), z = (int)(i / ((long long)Wo * Ho));
        out[plane * vo + i] = in[plane * vi + ((long long)min(z, Di - 1) * Hi + min(y, Hi - 1)) * Wi + min(x, Wi - 1)];
    }
}

template <class P>
__global__ void f32_to_cp_kernel(const float* __restrict__ in, uint4* __restrict__ out, int C, long long vox) {
    const int c8 = blockIdx.y, n = blockIdx.z;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vox; i += (long long)gridDim.x * 256) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = in[((long long)n * C + c8 * 8 + k) * vox + i];
        uint4 r;
        r.x = P::pack2(v[0], v[1]);
        r.y = P::pack2(v[2], v[3]);
        r.z = P::pack2(v[4], v[5]);
        r.w = P::pack2(v[6], v[7]);
        out[((long long)n * (C / 8) + c8) * vox + i] = r;
    }
}
template <class P>
__global__ void cp_to_f32_kernel(const uint4* __restrict__ in, float* __restrict__ out, int C, long long vox) {
    const int c8 = blockIdx.y, n = blockIdx.z;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vox; i += (long long)gridDim.x * 256) {
        const uint4 u = in[((long long)n * (C / 8) + c8) * vox + i];
        const float v[8] = {P::lo(u.x), P::hi(u.x), P::lo(u.y), P::hi(u.y), P::lo(u.z), P::hi(u.z), P::lo(u.w), P::hi(u.w)};
#pragma unroll
        for (int k = 0; k < 8; ++k) out[((long long)n * C + c8 * 8 + k) * vox + i] = v[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct Dims {
    int D, H, W;
    long long vox() const { return (long long)D * H * W; }
};

template <class P>
const uint16_t* wpack(const DlvConvLayer& L) { return P::IS_F16 ? L.w_f16 : L.w_bf16; }
template <class P>
const uint16_t* wpack(const DlvDeconvLayer& L) { return P::IS_F16 ? L.w_f16 : L.w_bf16; }

// a tensor of the forward: chunk-planar data + the scale/shift it still awaits (nullptr: final values)
struct Act16 {
    uint4* p;
    int C;
    const float2* ss;
};

template <class P>
struct Net16 {
    dlv_ctx* ctx;
    int B;
    float* partials;
    size_t partials_floats;
    float2* ss_base;  // [DLV_N_CONV][B][256]: scale/shift of every conv layer of this forward (skip tensors stay raw)
    float2* ss_of(int li) const { return ss_base + (size_t)li * B * 256; }
    static size_t ss_bytes(int B) { return (size_t)DLV_N_CONV * B * 256 * sizeof(float2); }

    using Act = Act16;

    int grid1d(long long n) const { return (int)std::min<long long>((n + 255) / 256, 256LL * 16); }

    // does the register-resident-weights conv run this layer (and with which inputs may it apply the activation itself)?
    bool zreg_runs(int li, int c1, int c2, Dims d) const {
        const DlvConvLayer& L = ctx->conv[li];
        const int zreg_mask = ctx->zreg_mask;  // (dlv_diag_set) 1 = Cin 32, 2 = Cin 64
        return (ctx->zm_variant == 0 || ctx->zm_variant == 50) && !ctx->no_zmarch && d.vox() > 32768 &&
               dlv_conv3_zreg_supports(L.cin, L.cout, c1, c2, d.W) && ((L.cin == 32 ? 1 : 2) & zreg_mask);
    }
    bool fuses_first_input(int li, int c1, int c2, Dims d) const {
        // which raw tensors are activated by the consuming conv while it stages them: per conv block (ctx->fuse_layers, default
        // block 17 = upcat_1.conv_1: common.h) or per level (fuse_levels, bit l = level l; A/B).  The Mish costs the staging conv
        // issue cycles (one wave per SIMD), the separate pass costs HBM time: it pays where the pass it removes is a whole
        // read + write of a level-0 tensor and nothing else changes; the transposed convs and the final 1x1x1 conv always
        // activate on load.
        const int fuse_levels = ctx->fuse_levels;
        static const int level_of[DLV_N_CONV] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0};
        return (((fuse_levels >> level_of[li]) & 1) || ((ctx->fuse_layers >> li) & 1)) && c1 == 32 && zreg_runs(li, c1, c2, d);
    }
    // InstanceNorm + Mish in place (the tensor becomes final); PW: the format the activated tensor is written in (P, except
    // where the mixed mode hands a bf16 level-1 tensor to the fp16 level 0)
    template <class PW = P>
    int materialise(Act& t, Dims d) {
        if (!t.ss) return DLV_OK;
        DLV_TRY((norm_mish<PW, P>(t.p, t.C, d, nullptr, t.ss, true)));
        t.ss = nullptr;
        return DLV_OK;
    }

    int stats(int nparts, int li, Dims d) {
        const DlvConvLayer& L = ctx->conv[li];
        // the fp16 format stores the raw stem output scaled by 2^-8: eps scales with its square (same normalised value)
        // ... and a layer stored 2^-shift times smaller (dlv_unet_set_conv_shift) with 4^-shift
        const float eps = (li == 0 ? 1e-5f * P::STEM_SCALE * P::STEM_SCALE : 1e-5f) * exp2f(-2.f * (float)L.shift);
        hipLaunchKernelGGL(stats_finalize_kernel, dim3(B * L.cout), dim3(64), 0, ctx->stream, partials, nparts, L.cout,
                           1.0 / (double)d.vox(), eps, L.gamma, L.beta, ss_of(li), ctx->range_flag, li);
        DLV_LAUNCH_CHECK(ctx, "stats_finalize_kernel");
        return DLV_OK;
    }

    // is conv `li` (the first conv of an UpCat block) run as skip-half conv + folded up half?
    bool folds_up(int li, int cskip, Dims d) const {
        const DlvConvLayer& L = ctx->conv[li];
        if (!(ctx->fold_up && L.up_corr != nullptr && cskip == 32 && (ctx->zm_variant == 0 || ctx->zm_variant == 50) && !ctx->no_zmarch &&
              d.vox() > 32768 && dlv_conv3_zreg_supports(32, L.cout, 32, 0, d.W) && d.D % 2 == 0 && d.H % 2 == 0 && d.W % 2 == 0))
            return false;
        return true;
    }
    int conv_folded(int li, Act& sk, Act& coarse, uint4* pbuf, uint4* out, Dims d, Dims dc) {
        const DlvConvLayer& L = ctx->conv[li];
        DLV_TRY(materialise(coarse, dc));  // the folded weights multiply the ACTIVATED coarse tensor
        if (!fuses_first_input(li, 32, 32, d)) DLV_TRY(materialise(sk, d));  // (else: activated by the conv while it stages the planes)
        if (coarse.C != 32) return dlv_fail(ctx, DLV_ESTATE, "folded conv %d: %d coarse channels, expected 32", li, coarse.C);
        {
            char nm[48];
            snprintf(nm, sizeof(nm), "upconv2%s_%s_c32x32_d%d", dlv_upconv2_persistent(ctx, dc.D, dc.H, dc.W) ? "m" : "", P::IS_F16 ? "f16" : "bf16", dc.D);
            DlvProf pr(ctx, nm, 2.0 * 8 * 32 * 32 * (double)d.vox() * B, 2.0 * 32 * ((double)dc.vox() + (double)d.vox()) * B);
            DLV_TRY(dlv_upconv2_launch(ctx, P::IS_F16, coarse.p, P::IS_F16 ? L.wup_f16 : L.wup_bf16, L.up_corr, pbuf, B, dc.D, dc.H, dc.W, coarse.C / 8, 0));
            pr.end();
        }
        char zname[48];
        snprintf(zname, sizeof(zname), "conv3_zreg_%s_c32x%d_d%d_add%s", P::IS_F16 ? "f16" : "bf16", L.cout, d.D, sk.ss ? "_act" : "");
        DlvProf zp(ctx, zname, 2.0 * 27 * 32 * L.cout * (double)d.vox() * B, 2.0 * (double)d.vox() * B * (32 + 2 * L.cout));
        int np = 0;
        if ((size_t)B * dlv_cdiv(d.H, 8) * dlv_cdiv(d.W, 32) * dlv_cdiv(d.D, 16) * L.cout * 2 > partials_floats)
            return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (zreg)");
        DLV_TRY(dlv_conv3_zreg_launch(ctx, P::IS_F16, 32, L.cout, sk.p, 32, sk.ss, nullptr, 0, nullptr, P::IS_F16 ? L.wskip_f16 : L.wskip_bf16, out,
                                      partials, B, d.D, d.H, d.W, &np, pbuf));
        zp.end();
        return stats(np, li, d);
    }

    // raw conv output + its InstanceNorm scale/shift into ss_of(li).  Inputs that still await their activation are either
    // activated by the kernel while it stages them (conv_zreg.hip) or made final by a normalisation pass first.
    int conv(int li, Act& a1, Act* a2, uint4* out, Dims d) {
        const DlvConvLayer& L = ctx->conv[li];
        const int c1 = a1.C, c2 = a2 ? a2->C : 0;
        if (c1 + c2 != L.cin) return dlv_fail(ctx, DLV_ESTATE, "conv %d: %d+%d input channels, expected %d", li, c1, c2, L.cin);
        if (c1 % 32 || c2 % 32) return dlv_fail(ctx, DLV_EUNSUP, "conv %d: concat parts must be multiples of 32 channels", li);
        if (a2) DLV_TRY(materialise(*a2, d));
        if (!fuses_first_input(li, c1, c2, d)) DLV_TRY(materialise(a1, d));
        const uint4* in1 = a1.p;
        const uint4* in2 = a2 ? a2->p : nullptr;
        if (zreg_runs(li, c1, c2, d)) {
            char zname[48];
            snprintf(zname, sizeof(zname), "conv3_zreg_%s_c%dx%d_d%d%s", P::IS_F16 ? "f16" : "bf16", L.cin, L.cout, d.D, a1.ss ? "_act" : "");
            DlvProf zp(ctx, zname, 2.0 * 27 * L.cin * L.cout * (double)d.vox() * B, 2.0 * (double)d.vox() * B * (L.cin + L.cout));
            int np = 0;
            if ((size_t)B * dlv_cdiv(d.H, 8) * dlv_cdiv(d.W, 32) * dlv_cdiv(d.D, 16) * L.cout * 2 > partials_floats)
                return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (zreg)");
            DLV_TRY(dlv_conv3_zreg_launch(ctx, P::IS_F16, L.cin, L.cout, in1, c1, a1.ss, in2, c2, nullptr,
                                          P::IS_F16 ? L.w16_f16 : L.w16_bf16, out, partials, B, d.D, d.H, d.W, &np));
            zp.end();
            return stats(np, li, d);
        }
        // deep levels (conv_deep.hip): weights shared through LDS, persistent workgroups.  deep_mask (A/B, dlv_diag_set): bit 0 = the layers
        // the LDS-weights z-march below takes (Cin, Cout <= 64 at the 32^3 level: 64->64 equal, 32->64 69 vs 78 us - they
        // stay with the z-march), bit 1 = the others (Cin or Cout >= 128: 1.4-1.5x the generic kernel's rate)
        const int deep_mask = ctx->deep_mask;
        const bool zmarch_ok = (L.cout == 32 || L.cout == 64) && (L.cin == 32 || L.cin == 64) && d.W >= 32;
        const bool deep_full = L.cout >= 64 && d.W >= 8 && d.H >= 8 && d.D >= 4;  // (what the kernel took before round 6: A/B switch "deep_small")
        if (!ctx->no_zmarch && ((zmarch_ok ? 1 : 2) & deep_mask) && (deep_full || ctx->deep_small) &&
            dlv_conv3_deep_supports(L.cin, L.cout, c1, c2, d.D, d.H, d.W)) {
            char zname[48];
            snprintf(zname, sizeof(zname), "conv3_deep_%s_c%dx%d_d%d", P::IS_F16 ? "f16" : "bf16", L.cin, L.cout, d.D);
            // (algorithmic bytes: activations in and out + the weights once - at these levels they are 10-50 % of the activations)
            DlvProf zp(ctx, zname, 2.0 * 27 * L.cin * L.cout * (double)d.vox() * B, 2.0 * (double)d.vox() * B * (L.cin + L.cout) + 2.0 * 27 * L.cin * L.cout);
            int np = 0;
            if ((size_t)B * dlv_cdiv(d.D, 4) * dlv_cdiv(d.H, 8) * dlv_cdiv(d.W, 8) * L.cout * 2 > partials_floats)
                return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (deep)");
            DLV_TRY(dlv_conv3_deep_launch(ctx, P::IS_F16, L.cin, L.cout, in1, c1, in2, c2, P::IS_F16 ? L.w16_f16 : L.w16_bf16, out, partials, B,
                                          d.D, d.H, d.W, &np));
            zp.end();
            return stats(np, li, d);
        }
        if (zmarch_ok && !ctx->no_zmarch) {  // LDS-weights z-march (conv_zmarch.hip)
            char zname[48];
            snprintf(zname, sizeof(zname), "conv3_zmarch_%s_c%dx%d_d%d", P::IS_F16 ? "f16" : "bf16", L.cin, L.cout, d.D);
            DlvProf zp(ctx, zname, 2.0 * 27 * L.cin * L.cout * (double)d.vox() * B, 2.0 * (double)d.vox() * B * (L.cin + L.cout));
            int np = 0;
            if ((size_t)B * dlv_cdiv(d.H, 8) * dlv_cdiv(d.W, 32) * dlv_cdiv(d.D, 16) * L.cout * 2 > partials_floats)
                return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (zmarch)");
            DLV_TRY(dlv_conv3_zmarch_launch(ctx, P::IS_F16, L.cin, L.cout, in1, c1, in2, c2, wpack<P>(L), L.bias16, out, partials, B,
                                            d.D, d.H, d.W, &np));
            zp.end();
            if ((size_t)B * np * L.cout * 2 > partials_floats) return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (zmarch)");
            return stats(np, li, d);
        }
        const bool tx16 = d.W >= 16;
        const int TX = tx16 ? 16 : 8, TY = 64 / TX;
        const int tZ = dlv_cdiv(d.D, 4), tY = dlv_cdiv(d.H, TY), tX = dlv_cdiv(d.W, TX);
        const int ntiles = tZ * tY * tX;
        // Two weight paths (A/B in profiles/README.md): fragments straight from L2 (levels 2-3: enough workgroups to
        // hide the latency; up to 4 cout blocks per workgroup) or the slab's weights staged through LDS in one
        // coalesced sweep (the 8^3 level: few workgroups, per-k-step fragment loads are latency-bound there).
        const bool wlds = d.vox() <= 1024;
        int ncb = wlds ? (L.cout >= 64 ? 2 : 1) : (L.cout >= 128 ? 4 : (L.cout >= 64 ? 2 : 1));
        while (ncb > 1 && (long long)B * ntiles * (L.cout / (32 * ncb)) < 512) ncb >>= 1;
        if (ctx->generic_ncb) {  // A/B switch (dlv_diag_set; profiles/README.md)
            const int want = ctx->generic_ncb;
            if ((want == 1 || want == 2 || (want == 4 && !wlds)) && L.cout % (32 * want) == 0) ncb = want;
        }
        if ((size_t)B * ntiles * L.cout * 2 > partials_floats) return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small");
        const size_t slab_bytes = (size_t)(tx16 ? ConvTile<16>::SLAB : ConvTile<8>::SLAB) * 16;
        const size_t lds = std::max<size_t>(slab_bytes + (wlds ? (size_t)ncb * 27 * 2 * 64 * 16 : 0), (size_t)4 * ncb * 32 * 2 * 4);
        dim3 grid(ntiles, L.cout / (32 * ncb), B);
        const double flops = 2.0 * 27 * L.cin * L.cout * (double)d.vox() * B;
        const double bytes = 2.0 * (double)d.vox() * B * (L.cin + L.cout);
        char name[48];
        snprintf(name, sizeof(name), "conv3_mfma_%s_c%dx%d_d%d", P::IS_F16 ? "f16" : "bf16", L.cin, L.cout, d.D);
        DlvProf pr(ctx, name, flops, bytes);
#define DLV_CONV_LAUNCH(NCB_, TX_, WLDS_)                                                                                \
    do {                                                                                                                 \
        static dlv_attr_bits attr_done{0}; /* bit per device */                                                              \
        if (!dlv_attr_is_set(attr_done, ctx->device)) {                                                                                                \
            DLV_HIP(ctx, hipFuncSetAttribute((const void*)conv3_mfma_kernel<P, NCB_, TX_, WLDS_>,                           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
            dlv_attr_mark(attr_done, ctx->device);                                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((conv3_mfma_kernel<P, NCB_, TX_, WLDS_>), grid, dim3(256), lds, ctx->stream, in1, c1 / 8, in2,   \
                           c2 / 8, reinterpret_cast<const uint4*>(wpack<P>(L)), L.bias16, out, partials, L.cout, d.D, d.H,    \
                           d.W, tY, tX);                                                                                 \
    } while (0)
        if (wlds) {
            if (tx16) {
                if (ncb == 1) DLV_CONV_LAUNCH(1, 16, true);
                else DLV_CONV_LAUNCH(2, 16, true);
            } else {
                if (ncb == 1) DLV_CONV_LAUNCH(1, 8, true);
                else DLV_CONV_LAUNCH(2, 8, true);
            }
        } else if (tx16) {
            if (ncb == 1) DLV_CONV_LAUNCH(1, 16, false);
            else if (ncb == 2) DLV_CONV_LAUNCH(2, 16, false);
            else DLV_CONV_LAUNCH(4, 16, false);
        } else {
            if (ncb == 1) DLV_CONV_LAUNCH(1, 8, false);
            else if (ncb == 2) DLV_CONV_LAUNCH(2, 8, false);
            else DLV_CONV_LAUNCH(4, 8, false);
        }
#undef DLV_CONV_LAUNCH
        pr.end();
        DLV_LAUNCH_CHECK(ctx, "conv3_mfma_kernel");
        return stats(ntiles, li, d);
    }

    // InstanceNorm apply + Mish (+ MaxPool into `pooled`); writeback = false (pool only): x stays raw for consumers that
    // activate while loading
    // PW / PQ: formats of the written-back and of the pooled tensor (norm_mish_kernel)
    template <class PW = P, class PQ = P>
    int norm_mish(uint4* x, int C, Dims d, uint4* pooled, const float2* ss, bool writeback) {
        const long long work = pooled ? d.vox() / 8 : d.vox();
        // a tensor far beyond L2 + MALL is streamed with the non-temporal policy and two grid-stride iterations per thread
        // (profiles/microbench/nt_probe.hip: 2.15 GB in place 743 us at 2048 x default, 642 us at 4096 x nt); the small levels
        // keep the default policy - their tensors are still on chip when the consumer starts
        const bool nt = (double)d.vox() * B * C * 2 > 768.0 * (1 << 20);
        dim3 grid(std::max(1, std::min(grid1d(work), nt && !pooled ? 4096 : 2048)), C / 8, B);
        constexpr bool seam = !std::is_same<PW, P>::value || !std::is_same<PQ, P>::value;  // (the mixed mode's format change)
        DlvProf pr(ctx, pooled ? (writeback ? (P::IS_F16 ? (seam ? "norm_mish_pool_f16_to_bf16" : "norm_mish_pool_f16") : "norm_mish_pool_bf16")
                                            : (P::IS_F16 ? "pool_act_f16" : "pool_act_bf16"))
                               : (P::IS_F16 ? "norm_mish_f16" : (seam ? "norm_mish_bf16_to_f16" : "norm_mish_bf16")), 0.0,
                   (double)d.vox() * B * C * 2 * (writeback ? 2 : 1) + (pooled ? (double)d.vox() / 8 * B * C * 2 : 0.0));
#define DLV_NM_LAUNCH(POOL_, WB_)                                                                                                  \
    do {                                                                                                                           \
        if (nt)                                                                                                                    \
            hipLaunchKernelGGL((norm_mish_kernel<P, POOL_, WB_, true, PW, PQ>), grid, dim3(256), 0, ctx->stream, x, ss, C, d.D, d.H, d.W, pooled); \
        else                                                                                                                       \
            hipLaunchKernelGGL((norm_mish_kernel<P, POOL_, WB_, false, PW, PQ>), grid, dim3(256), 0, ctx->stream, x, ss, C, d.D, d.H, d.W, pooled); \
    } while (0)
        const bool pool_rows_off = ctx->pool_rows_off;  // A/B + tests (dlv_diag_set): the pooled-voxel-per-thread kernel everywhere
        if (pooled && d.W % 64 == 0 && !pool_rows_off) {
            const long long items = (long long)(d.D / 2) * (d.H / 2) * (d.W / 64);
            dim3 g2((unsigned)std::max<long long>(1, std::min<long long>((items + 3) / 4, 4096)), C / 8, B);  // (one item per wave: 4 / 8 / 16 items per workgroup 902 / 914 / 939 us)
#define DLV_NP_LAUNCH(WB_, NT_) \
    hipLaunchKernelGGL((norm_mish_pool_rows_kernel<P, WB_, NT_, PQ>), g2, dim3(256), 0, ctx->stream, x, ss, C, d.D, d.H, d.W, pooled)
            if (writeback && nt) DLV_NP_LAUNCH(true, true);
            else if (writeback) DLV_NP_LAUNCH(true, false);
            else if (nt) DLV_NP_LAUNCH(false, true);
            else DLV_NP_LAUNCH(false, false);
#undef DLV_NP_LAUNCH
        } else if (pooled && writeback) DLV_NM_LAUNCH(true, true);
        else if (pooled) DLV_NM_LAUNCH(true, false);
        else DLV_NM_LAUNCH(false, true);
#undef DLV_NM_LAUNCH
        pr.end();
        DLV_LAUNCH_CHECK(ctx, "norm_mish_kernel");
        return DLV_OK;
    }

    // the transposed conv of an UpCat block into `out` with the SKIP tensor's dimensions `dskip`: where they are 2 x the input's,
    // straight into it; where the skip tensor is odd (windows that are not multiples of 16), into `tmp` and from there
    // replicate-padded by one voxel at the far end (MONAI's UpCat)
    int deconv_to(int j, Act& a, uint4* out, uint4* tmp, Dims din, Dims dskip) {
        const Dims du{2 * din.D, 2 * din.H, 2 * din.W};
        if (du.D == dskip.D && du.H == dskip.H && du.W == dskip.W) return deconv(j, a, out, din);
        if (dskip.D - du.D > 1 || dskip.H - du.H > 1 || dskip.W - du.W > 1 || dskip.D < du.D || dskip.H < du.H || dskip.W < du.W)
            return dlv_fail(ctx, DLV_ESTATE, "deconv %d: skip tensor %dx%dx%d against an up-sampled %dx%dx%d", j, dskip.D, dskip.H, dskip.W, du.D, du.H, du.W);
        DLV_TRY(deconv(j, a, tmp, din));
        const int cout = ctx->deconv[j].cout;
        DlvProf pr(ctx, P::IS_F16 ? "replicate_pad_f16" : "replicate_pad_bf16", 0.0, 16.0 * B * (cout / 8) * ((double)du.vox() + (double)dskip.vox()));
        hipLaunchKernelGGL(replicate_pad_cp_kernel, dim3(std::max(1, std::min(grid1d(dskip.vox()), 1024)), B * (cout / 8)), dim3(256), 0, ctx->stream,
                           tmp, out, du.D, du.H, du.W, dskip.D, dskip.H, dskip.W);
        pr.end();
        DLV_LAUNCH_CHECK(ctx, "replicate_pad_cp_kernel");
        return DLV_OK;
    }

    int deconv(int j, Act& a, uint4* out, Dims din) {
        const DlvDeconvLayer& L = ctx->deconv[j];
        const uint4* w = reinterpret_cast<const uint4*>(wpack<P>(L));
        const bool rows = !ctx->no_zmarch;
        // the per-parity kernel has no activation on load; the weight-stationary kernel of the deep levels (Cin >= 128) would
        // repeat it for every (parity, output block) it enumerates: there the (small) input is activated by one norm pass
        if (!rows || L.cin >= 128) DLV_TRY(materialise(a, din));
        const uint4* in = a.p;
        const float2* ssin = a.ss;
        const int segs = dlv_cdiv(din.W, 16);
        dim3 grid(rows ? dlv_cdiv((long long)din.D * din.H * segs, 4) : dlv_cdiv(din.vox(), 128), B);
        // register-resident weights + segment pipeline where the weights fit (Cout = 32, Cin <= 64) and a window has enough
        // row segments (a property of the window shape, not of the batch)
        const bool regw = rows && L.cout == 32 && L.cin <= 64 && (long long)din.D * din.H * segs >= 4 * DC_IPW * 64 &&
                          din.vox() * 8 * 4 * 16 < (1ll << 32);  // (its stores address one sample's output with 32-bit offsets)
        if (regw) grid.x = dlv_cdiv((long long)din.D * din.H * segs, 4 * DC_IPW);
        char dname[48];
        const bool deep_off = ctx->deep_mask == 0;  // (A/B: the round-4 kernels)
        const bool deep = rows && !deep_off && L.w16_f16 && a.ss == nullptr && dlv_deconv2_deep_supports(L.cin, L.cout, din.D, din.H, din.W);
        snprintf(dname, sizeof(dname), "deconv2_%s_%s_c%dx%d_d%d", deep ? "deep" : "mfma", P::IS_F16 ? "f16" : "bf16", L.cin, L.cout, din.D);
        DlvProf pr(ctx, dname, 2.0 * 8 * L.cin * L.cout * (double)din.vox() * B,
                   2.0 * (double)din.vox() * B * (L.cin + 8.0 * L.cout));
        if (deep) {  // Cin 128 / 256 at the deep levels: weights shared through LDS (conv_deep.hip)
            DLV_TRY(dlv_deconv2_deep_launch(ctx, P::IS_F16, L.cin, L.cout, in, P::IS_F16 ? L.w16_f16 : L.w16_bf16, L.bias, out, B, din.D, din.H, din.W));
            pr.end();
            return DLV_OK;
        }
#define DLV_DECONV(KP_)                                                                                                  \
    do {                                                                                                                 \
        if (rows)                                                                                                        \
            hipLaunchKernelGGL((deconv2_rows_kernel<P, KP_>), grid, dim3(256), 0, ctx->stream, in, w, L.bias, out, L.cout,     \
                               din.D, din.H, din.W, segs, ssin);                                                               \
        else                                                                                                             \
            hipLaunchKernelGGL((deconv2_mfma_kernel<P, KP_>), grid, dim3(256), 0, ctx->stream, in, w, L.bias, out, L.cout,     \
                               din.D, din.H, din.W);                                                                     \
    } while (0)
        switch (L.cin / 16) {
            case 2:
                if (regw) hipLaunchKernelGGL((deconv2_regw_kernel<P, 2>), grid, dim3(256), 0, ctx->stream, in, w, L.bias, out, din.D, din.H, din.W, segs, ssin);
                else DLV_DECONV(2);
                break;
            case 4:
                if (regw) hipLaunchKernelGGL((deconv2_regw_kernel<P, 4>), grid, dim3(256), 0, ctx->stream, in, w, L.bias, out, din.D, din.H, din.W, segs, ssin);
                else DLV_DECONV(4);
                break;
            case 8:
                if (rows) hipLaunchKernelGGL((deconv2_wst_kernel<P, 8>), dim3(dlv_cdiv((long long)din.D * din.H * segs, 4 * DW_IPW), 4 * (L.cout / 32), B), dim3(256), 0, ctx->stream, in, w, L.bias, out, L.cout, din.D, din.H, din.W, segs, ssin);
                else DLV_DECONV(8);
                break;
            case 16:
                if (rows) hipLaunchKernelGGL((deconv2_wst_kernel<P, 16>), dim3(dlv_cdiv((long long)din.D * din.H * segs, 4 * DW_IPW), 4 * (L.cout / 32), B), dim3(256), 0, ctx->stream, in, w, L.bias, out, L.cout, din.D, din.H, din.W, segs, ssin);
                else DLV_DECONV(16);
                break;
            default: return dlv_fail(ctx, DLV_EUNSUP, "deconv %d: Cin=%d not in {32,64,128,256}", j, L.cin);
        }
#undef DLV_DECONV
        pr.end();
        DLV_LAUNCH_CHECK(ctx, "deconv2 kernel");
        return DLV_OK;
    }
};

// the whole forward; the stem reads either xf (fp32 patches) or the uint16 volume windows, the final
// layer writes either logits or blends into acc
// workspace of one 16-bit forward of B windows (one pipeline lane): the activations (4 buffers per level), the InstanceNorm partial
// sums + scale/shift tables.  Shared by forward_16 and dlv_unet_reserve_16 (the same arithmetic, or the reservation is useless).
struct Ws16 {
    size_t act, stats, pfloats;
};
static Ws16 ws16_bytes(const int* f, int B, int d, int h, int w, size_t (*offs)[4]) {
    const int lvlC[5] = {32, f[1], f[2], f[3], f[4]};
    size_t off = 0;
    for (int l = 0; l < 5; ++l)
        for (int k = 0; k < 4; ++k) {
            if (offs) offs[l][k] = off;
            off += (size_t)B * lvlC[l] * ((size_t)(d >> l) * (h >> l) * (w >> l)) * 2;
            off = (off + 255) & ~(size_t)255;
        }
    // partial sums: the level-0 convs have the most tiles (256 voxels each); the stem has fewer blocks
    const long long max_tiles = (long long)dlv_cdiv(d, 4) * dlv_cdiv(h, 4) * dlv_cdiv(w, 8) + 64;
    const size_t pfloats = (size_t)B * max_tiles * 64 * 2;
    return Ws16{off, pfloats * 4 + (size_t)DLV_N_CONV * B * 256 * sizeof(float2) + 256, pfloats};
}

// P0: the format of level 0 (full resolution: stem, conv_0, upcat_1, final conv), PD: of levels 1-4.  P0 = PD: one format
// throughout (fp16 / bf16 everywhere); P0 = fp16 with PD = bf16 is the mixed mode DLV_PREC_BF16 stands for (DESIGN section 5:
// the 8 bits of bf16 are lost at full resolution - fp16 there lifts the margin-free mask IoU from 0.998 to 0.9994).  The
// format changes in two normalisation passes: the pooling pass 0 -> 1 (raw fp16 in, pooled bf16 out) and the pass that
// activates the level-1 decoder output for upcat_1 (raw bf16 in, activated fp16 out).
template <class P0, class PD = P0>
int forward_16(dlv_ctx* ctx, const float* xf, const uint16_t* vol, int Yp, int Xp, const int* starts_dev, int flip_dim,
                 float scale, float* logits, float* acc, int B, int d, int h, int w) {
    const int* f = ctx->features;
    if (f[0] != 32 || f[5] != 32)
        return dlv_fail(ctx, DLV_EUNSUP, "bf16 path: features[0] and features[5] must be 32 (got %d, %d)", f[0], f[5]);
    // the stem, the final conv and the norm passes index a window's voxels with 32 bits (whatever conv kernel runs in between)
    if ((long long)d * h * w >= (1ll << 31)) return dlv_fail(ctx, DLV_EUNSUP, "16-bit path: a window of %d x %d x %d voxels exceeds the 2^31 voxel index range", d, h, w);
    Dims dm[5];
    for (int l = 0; l < 5; ++l) dm[l] = Dims{d >> l, h >> l, w >> l};
    size_t offs[5][4];
    Ws16 wsz = ws16_bytes(f, B, d, h, w, offs);
    const size_t off = wsz.act, pfloats = wsz.pfloats;
    char* base;
    DLV_TRY(dlv_ws_get(ctx, ctx->lane ? WS_LANE_ACT0 + (ctx->lane - 1) : WS_BF16_ACT, off, (void**)&base));
    char* sbase;
    DLV_TRY(dlv_ws_get(ctx, ctx->lane ? WS_LANE_STATS0 + (ctx->lane - 1) : WS_STATS, wsz.stats, (void**)&sbase));
    Net16<P0> net{ctx, B, (float*)sbase, pfloats, (float2*)(sbase + ((pfloats * 4 + 255) & ~(size_t)255))};  // level 0
    Net16<PD> netd{ctx, B, net.partials, pfloats, net.ss_base};                                                // levels 1-4
    constexpr bool mixed = !std::is_same<P0, PD>::value;
    using P = P0;
    using Act = Act16;
    auto buf = [&](int l, int k) { return (uint4*)(base + offs[l][k]); };
    enum { A = 0, Bf = 1, S = 2, U = 3 };

    // stem
    Act x0{buf(0, A), 32, nullptr};
    {
        dim3 grid(dlv_cdiv((long long)h * w, 256), dlv_cdiv(d, STEM_ZR), B);
        int nblk = grid.x * grid.y;
        bool two_pass_stem = false;
        if ((size_t)B * nblk * 64 > pfloats) return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (stem)");
        const DlvConvLayer& L = ctx->conv[0];
        DlvProf pr(ctx, (vol && !ctx->no_zmarch) ? "stem_mfma_u16" : "stem_conv_f32", 2.0 * 27 * 32 * (double)dm[0].vox() * B, (double)dm[0].vox() * B * (2 + 64));
        if (vol && !ctx->no_zmarch) {
            const int tY = dlv_cdiv(h, SM_TY), tX = dlv_cdiv(w, SM_TX), tZ = dlv_cdiv(d, SM_TZ * SM_ZC);
            grid = dim3(tZ * tY * tX, 1, B);
            nblk = grid.x;
            if ((size_t)B * nblk * 64 > pfloats) return dlv_fail(ctx, DLV_ESTATE, "partials buffer too small (stem)");
            const uint4* wst = reinterpret_cast<const uint4*>(wpack<P>(L));
            hipLaunchKernelGGL((stem_mfma_kernel<P, 1>), grid, dim3(256), 0, ctx->stream, vol, Yp, Xp, starts_dev, flip_dim, wst,
                               L.bias16, buf(0, A), net.partials, (const float2*)nullptr, d, h, w, tY, tX);
            DLV_LAUNCH_CHECK(ctx, "stem_mfma_kernel<1>");
            DLV_TRY(net.stats(nblk, 0, dm[0]));
            hipLaunchKernelGGL((stem_mfma_kernel<P, 2>), grid, dim3(256), 0, ctx->stream, vol, Yp, Xp, starts_dev, flip_dim, wst,
                               L.bias16, buf(0, A), net.partials, (const float2*)net.ss_of(0), d, h, w, tY, tX);
            two_pass_stem = true;
        } else if (vol)
            hipLaunchKernelGGL((stem_conv_kernel<P, true>), grid, dim3(256), 0, ctx->stream, nullptr, vol, Yp, Xp, starts_dev,
                               flip_dim, L.w_f32, L.bias, buf(0, A), net.partials, d, h, w, P::STEM_SCALE * exp2f(-(float)L.shift));
        else
            hipLaunchKernelGGL((stem_conv_kernel<P, false>), grid, dim3(256), 0, ctx->stream, xf, nullptr, 0, 0, nullptr, -1,
                               L.w_f32, L.bias, buf(0, A), net.partials, d, h, w, P::STEM_SCALE * exp2f(-(float)L.shift));
        pr.end();
        DLV_LAUNCH_CHECK(ctx, "stem_conv_kernel");
        if (!two_pass_stem) {
            DLV_TRY(net.stats(nblk, 0, dm[0]));
            x0.ss = net.ss_of(0);  // raw: the first conv activates it while staging, or it is made final first
        }
    }
    const int encC[5] = {32, f[1], f[2], f[3], f[4]};
    // encoder.  skip[l] = output of level l (consumed pooled by level l+1 and, as the skip connection, by upcat_{l+1}).
    // Where both consumers of a raw tensor activate on load, no normalised copy of it is ever written: only the pooled
    // tensor is produced (levels 0 and 1 of the default windows); elsewhere the normalisation pass writes it back.
    Act skip[5];
    {
        Act s0{buf(0, S), 32, nullptr};
        DLV_TRY(net.conv(1, x0, nullptr, s0.p, dm[0]));
        s0.ss = net.ss_of(1);
        skip[0] = s0;
    }
    for (int l = 1; l <= 4; ++l) {
        // pool level l-1 into the input of level l
        Act& up = skip[l - 1];
        const int li_cat = 18 - 2 * l;  // upcat conv that takes skip[l-1]: 16, 14, 12, 10
        const int c_up = ctx->deconv[4 - l].cout;
        const bool keep_raw = net.fuses_first_input(li_cat, up.C, c_up, dm[l - 1]);
        const bool odd = ((dm[l - 1].D | dm[l - 1].H | dm[l - 1].W) & 1) != 0;
        // a level with an odd size (windows that are not multiples of 16): MaxPool3d(2) drops its last plane / row / column, so the
        // pooling pass - which writes back only what it pools - cannot be the pass that makes the tensor final: pool only, then a
        // full normalisation pass (unless every consumer activates on load)
        if (l == 1) {  // (level 0 stays P0, pooled: PD)
            DLV_TRY((net.template norm_mish<P0, PD>(up.p, up.C, dm[0], buf(1, A), up.ss, !keep_raw && !odd)));
            if (odd && !keep_raw) DLV_TRY(net.norm_mish(up.p, up.C, dm[0], nullptr, up.ss, true));
        } else {
            DLV_TRY(netd.norm_mish(up.p, up.C, dm[l - 1], buf(l, A), up.ss, !keep_raw && !odd));
            if (odd && !keep_raw) DLV_TRY(netd.norm_mish(up.p, up.C, dm[l - 1], nullptr, up.ss, true));
        }
        if (!keep_raw) up.ss = nullptr;
        Act a{buf(l, A), encC[l - 1], nullptr};
        Act b{buf(l, Bf), encC[l], nullptr};
        DLV_TRY(netd.conv(2 * l, a, nullptr, b.p, dm[l]));
        b.ss = net.ss_of(2 * l);
        Act s{buf(l, S), encC[l], nullptr};
        DLV_TRY(netd.conv(2 * l + 1, b, nullptr, s.p, dm[l]));
        s.ss = net.ss_of(2 * l + 1);
        skip[l] = s;
    }
    // decoder: transposed conv (activates its input on load), concat [skip, up], two convs
    Act cur = skip[4];
    for (int j = 0; j < 4; ++j) {
        const int l = 3 - j;
        const int li = 10 + 2 * j;
        Act b{buf(l, Bf), ctx->conv[li].cout, nullptr};
        if (l >= 1) {
            DLV_TRY(netd.deconv_to(j, cur, buf(l, U), buf(l, A), dm[l + 1], dm[l]));  // (buf(l, A): free until this block's second conv writes it)
            Act u{buf(l, U), ctx->deconv[j].cout, nullptr};
            DLV_TRY(netd.conv(li, skip[l], &u, b.p, dm[l]));
            b.ss = net.ss_of(li);
            Act o{buf(l, A), ctx->conv[li + 1].cout, nullptr};
            DLV_TRY(netd.conv(li + 1, b, nullptr, o.p, dm[l]));
            o.ss = net.ss_of(li + 1);
            cur = o;
            continue;
        }
        // level 0.  Mixed mode: the level-1 output is activated here and WRITTEN in the level-0 format (one pass either way: the
        // folded conv below needs the activated tensor, and the plain transposed conv then finds nothing left to activate)
        if (mixed) DLV_TRY((netd.template materialise<P0>(cur, dm[1])));
        if (net.folds_up(li, skip[l].C, dm[l])) {
            // upcat_1: the transposed conv folded into the conv (upconv.hip): P from the activated coarse tensor, then the
            // 32-channel conv of the skip half with P as its addend - no up-sampled tensor, 8 coarse taps instead of 27 fine ones
            DLV_TRY(net.conv_folded(li, skip[l], cur, buf(l, U), b.p, dm[l], dm[l + 1]));
        } else {
            DLV_TRY(net.deconv_to(j, cur, buf(l, U), buf(l, A), dm[l + 1], dm[l]));  // (buf(0, A): the stem's output, consumed)
            Act u{buf(l, U), ctx->deconv[j].cout, nullptr};
            DLV_TRY(net.conv(li, skip[l], &u, b.p, dm[l]));
        }
        b.ss = net.ss_of(li);
        Act o{buf(l, A), ctx->conv[li + 1].cout, nullptr};
        DLV_TRY(net.conv(li + 1, b, nullptr, o.p, dm[l]));
        o.ss = net.ss_of(li + 1);
        cur = o;
    }
    {
        dim3 grid(std::min(net.grid1d(dm[0].vox()), 512), B);  // (16 iterations per thread at 128^3 - the software pipeline wants a long loop: 1024 / 512 / 256 workgroups 515 / 487 / 484 us)
        DlvProf pr(ctx, acc ? "final_conv_blend" : "final_conv_logits", 2.0 * 32 * (double)dm[0].vox() * B,
                   (double)dm[0].vox() * B * (64 + (acc ? 8 : 4)));
        if (acc)
            hipLaunchKernelGGL((final_conv_kernel<P, true>), grid, dim3(256), 0, ctx->stream, cur.p, cur.ss, ctx->final_w,
                               ctx->final_b, nullptr, starts_dev, flip_dim, Yp, Xp, scale, acc, d, h, w, ctx->blend_w, ctx->blend_min,
                               ctx->blend_wsum, ctx->range_flag);
        else
            hipLaunchKernelGGL((final_conv_kernel<P, false>), grid, dim3(256), 0, ctx->stream, cur.p, cur.ss, ctx->final_w,
                               ctx->final_b, logits, nullptr, -1, 0, 0, 1.f, nullptr, d, h, w, nullptr, 0.f, nullptr, ctx->range_flag);
        pr.end();
        DLV_LAUNCH_CHECK(ctx, "final_conv_kernel");
    }
    return DLV_OK;
}

template <class P>
int pack_weights_16(dlv_ctx* ctx) {
    auto dst = [](auto& L) { return const_cast<uint16_t*>(wpack<P>(L)); };
    hipLaunchKernelGGL(pack_stem_w_kernel<P>, dim3(8), dim3(256), 0, ctx->stream, ctx->conv[0].w_f32, dst(ctx->conv[0]), exp2f(-(float)ctx->conv[0].shift));
    DLV_LAUNCH_CHECK(ctx, "pack_stem_w_kernel");
    for (int i = 0; i < DLV_N_CONV; ++i) {  // the bias the 16-bit kernels add: scaled like the weights
        const DlvConvLayer& L = ctx->conv[i];
        hipLaunchKernelGGL(scale_copy_kernel, dim3(dlv_cdiv(L.cout, 256)), dim3(256), 0, ctx->stream, L.bias, L.bias16, L.cout, exp2f(-(float)L.shift));
        DLV_LAUNCH_CHECK(ctx, "scale_copy_kernel");
    }
    for (int i = 1; i < DLV_N_CONV; ++i) {
        const DlvConvLayer& L = ctx->conv[i];
        if (L.cin % 32 || L.cout % 32) return dlv_fail(ctx, DLV_EUNSUP, "conv %d: %d->%d not multiples of 32", i, L.cin, L.cout);
        const float ws = exp2f(-(float)L.shift);
        hipLaunchKernelGGL(pack_conv_w_kernel<P>, dim3(256), dim3(256), 0, ctx->stream, L.w_f32, dst(ctx->conv[i]), L.cout, L.cin, ws);
        DLV_LAUNCH_CHECK(ctx, "pack_conv_w_kernel");
        DLV_TRY(dlv_pack_conv_w16(ctx, P::IS_F16, L.w_f32, P::IS_F16 ? L.w16_f16 : L.w16_bf16, L.cout, L.cin, 0, 0, ws));
        if (L.up_corr) {  // upcat_1.conv_0: skip half as a 32-channel pack, up half folded with the transposed conv (upconv.hip)
            const DlvDeconvLayer& Dl = ctx->deconv[3];
            DLV_TRY(dlv_pack_conv_w16(ctx, P::IS_F16, L.w_f32, P::IS_F16 ? L.wskip_f16 : L.wskip_bf16, L.cout, 32, L.cin, 0, ws));
            DLV_TRY(dlv_pack_upconv(ctx, P::IS_F16, L.w_f32, L.cin, 32, Dl.w_f32, Dl.bias, P::IS_F16 ? L.wup_f16 : L.wup_bf16, L.up_corr, 0, 1, ws));
        }
    }
    for (int j = 0; j < DLV_N_DECONV; ++j) {
        const DlvDeconvLayer& L = ctx->deconv[j];
        if (L.cin % 32 || L.cout % 32) return dlv_fail(ctx, DLV_EUNSUP, "deconv %d: %d->%d not multiples of 32", j, L.cin, L.cout);
        hipLaunchKernelGGL(pack_deconv_w_kernel<P>, dim3(64), dim3(256), 0, ctx->stream, L.w_f32, dst(ctx->deconv[j]), L.cin, L.cout);
        DLV_LAUNCH_CHECK(ctx, "pack_deconv_w_kernel");
        if (L.w16_f16) DLV_TRY(dlv_pack_deconv_w16(ctx, P::IS_F16, L.w_f32, P::IS_F16 ? L.w16_f16 : L.w16_bf16, L.cin, L.cout));
    }
    return DLV_OK;
}

template <class P>
int debug_layer_16(dlv_ctx* ctx, int kind, int index, const float* in1_dev, int c1, const float* in2_dev, int c2,
                   float* out_dev, int B, int D, int H, int W) {
    const long long vox = (long long)D * H * W;
    if (kind == 0 || kind == 2 || kind == 3) {  // 2: raw conv output (no InstanceNorm / Mish), 3: the layer's scale/shift pairs: kernel debugging
        if (index < 1 || index >= DLV_N_CONV) return dlv_fail(ctx, DLV_EINVAL, "conv index must be 1..17");
        const DlvConvLayer& L = ctx->conv[index];
        if (c1 + c2 != L.cin || c1 % 32 || c2 % 32) return dlv_fail(ctx, DLV_EINVAL, "bad channel split");
        const size_t b1 = (size_t)B * c1 * vox * 2, b2 = (size_t)B * c2 * vox * 2, bo = (size_t)B * L.cout * vox * 2;
        const long long tiles = (long long)dlv_cdiv(D, 4) * dlv_cdiv(H, 4) * dlv_cdiv(W, 8);
        const size_t pf = (size_t)B * tiles * L.cout * 2;
        char* base;
        DLV_TRY(dlv_ws_get(ctx, WS_BF16_ACT, b1 + b2 + bo + 1024, (void**)&base));
        char* sbase;
        DLV_TRY(dlv_ws_get(ctx, WS_STATS, pf * 4 + Net16<P>::ss_bytes(B) + 256, (void**)&sbase));
        uint4 *i1 = (uint4*)base, *i2 = (uint4*)(base + ((b1 + 255) & ~(size_t)255)),
              *o = (uint4*)(base + ((b1 + 255) & ~(size_t)255) + ((b2 + 255) & ~(size_t)255));
        Net16<P> net{ctx, B, (float*)sbase, pf, (float2*)(sbase + ((pf * 4 + 255) & ~(size_t)255))};
        const int g = net.grid1d(vox);
        hipLaunchKernelGGL(f32_to_cp_kernel<P>, dim3(g, c1 / 8, B), dim3(256), 0, ctx->stream, in1_dev, i1, c1, vox);
        if (c2) hipLaunchKernelGGL(f32_to_cp_kernel<P>, dim3(g, c2 / 8, B), dim3(256), 0, ctx->stream, in2_dev, i2, c2, vox);
        typename Net16<P>::Act t1{i1, c1, nullptr}, t2{i2, c2, nullptr};
        DLV_TRY(net.conv(index, t1, c2 ? &t2 : nullptr, o, Dims{D, H, W}));
        if (kind == 0) DLV_TRY(net.norm_mish(o, L.cout, Dims{D, H, W}, nullptr, net.ss_of(index), true));
        if (kind == 3) {
            DLV_HIP(ctx, hipMemcpyAsync(out_dev, net.ss_of(index), (size_t)B * L.cout * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
            return DLV_OK;
        }
        hipLaunchKernelGGL(cp_to_f32_kernel<P>, dim3(g, L.cout / 8, B), dim3(256), 0, ctx->stream, o, out_dev, L.cout, vox);
        DLV_LAUNCH_CHECK(ctx, "debug conv");
        return DLV_OK;
    }
    if (kind == 1) {
        if (index < 0 || index >= DLV_N_DECONV) return dlv_fail(ctx, DLV_EINVAL, "deconv index must be 0..3");
        const DlvDeconvLayer& L = ctx->deconv[index];
        if (c1 != L.cin || c2 != 0) return dlv_fail(ctx, DLV_EINVAL, "bad channels");
        const size_t b1 = (size_t)B * c1 * vox * 2, bo = (size_t)B * L.cout * vox * 8 * 2;
        char* base;
        DLV_TRY(dlv_ws_get(ctx, WS_BF16_ACT, b1 + bo + 1024, (void**)&base));
        uint4 *i1 = (uint4*)base, *o = (uint4*)(base + ((b1 + 255) & ~(size_t)255));
        Net16<P> net{ctx, B, nullptr, 0, nullptr};
        hipLaunchKernelGGL(f32_to_cp_kernel<P>, dim3(net.grid1d(vox), c1 / 8, B), dim3(256), 0, ctx->stream, in1_dev, i1, c1, vox);
        typename Net16<P>::Act t1{i1, c1, nullptr};
        DLV_TRY(net.deconv(index, t1, o, Dims{D, H, W}));
        hipLaunchKernelGGL(cp_to_f32_kernel<P>, dim3(net.grid1d(vox * 8), L.cout / 8, B), dim3(256), 0, ctx->stream, o, out_dev,
                           L.cout, vox * 8);
        DLV_LAUNCH_CHECK(ctx, "debug deconv");
        return DLV_OK;
    }
    return dlv_fail(ctx, DLV_EINVAL, "kind must be 0 (conv block) or 1 (deconv)");
}

}  // namespace

int dlv_pack_weights_bf16(dlv_ctx* ctx) {
    DLV_TRY(pack_weights_16<PBf16>(ctx));
    return pack_weights_16<PF16>(ctx);
}

// the workspaces `lanes` pipeline lanes of a 16-bit forward of B windows of d x h x w will ask for (dlv_reserve_dev): a pass needs
// ~10 GB per lane, and a large hipMalloc that follows a release of device memory can take seconds (profiles/r06r_alloc_probe2.json)
int dlv_unet_reserve_16(dlv_ctx* ctx, int B, int d, int h, int w, int lanes) {
    if (!ctx->weights_loaded && ctx->features[1] == 0) return DLV_OK;  // (channel counts unknown before dlv_unet_load / alloc_blob)
    const Ws16 wsz = ws16_bytes(ctx->features, B, d, h, w, nullptr);
    void* p;
    for (int lane = 0; lane < std::max(1, std::min(lanes, DLV_MAX_LANES)); ++lane) {
        DLV_TRY(dlv_ws_get(ctx, lane ? WS_LANE_ACT0 + (lane - 1) : WS_BF16_ACT, wsz.act, &p));
        DLV_TRY(dlv_ws_get(ctx, lane ? WS_LANE_STATS0 + (lane - 1) : WS_STATS, wsz.stats, &p));
    }
    return DLV_OK;
}

int dlv_range_reset(dlv_ctx* ctx) {
    DLV_HIP(ctx, hipMemsetAsync(ctx->range_flag, 0, sizeof(int) * (1 + DLV_N_CONV), ctx->main_stream));
    return DLV_OK;
}

// after everything of the pass / forward has been ordered behind the main stream: read the guard word back
int dlv_range_check(dlv_ctx* ctx, int fmt16) {
    const bool f16 = fmt16 != 0;  // (the mixed mode: only its fp16 level 0 can leave the range)
    int words[1 + DLV_N_CONV] = {0};
    DLV_HIP(ctx, hipMemcpyAsync(words, ctx->range_flag, sizeof(words), hipMemcpyDeviceToHost, ctx->main_stream));
    DLV_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
    const int flag = words[0];
    for (int i = 0; i < DLV_N_CONV; ++i) memcpy(&ctx->range_peak[i], &words[1 + i], sizeof(float));
    ctx->range_last = flag == 0 ? -1 : 100 - flag;
    if (flag == 0) {
        ctx->range_seq = false;  // a pass came to its end: the recovery sequence (if any) is over
        ctx->range_blind_layer = -1;
        return DLV_OK;
    }
    const int layer = 100 - flag;
    static const char* const names[DLV_N_CONV] = {"conv_0.conv_0", "conv_0.conv_1", "down_1.conv_0", "down_1.conv_1", "down_2.conv_0",
                                                  "down_2.conv_1", "down_3.conv_0", "down_3.conv_1", "down_4.conv_0", "down_4.conv_1",
                                                  "upcat_4.conv_0", "upcat_4.conv_1", "upcat_3.conv_0", "upcat_3.conv_1", "upcat_2.conv_0",
                                                  "upcat_2.conv_1", "upcat_1.conv_0", "upcat_1.conv_1"};
    if (layer >= 0 && layer < DLV_N_CONV)
        return dlv_fail(ctx, DLV_ERANGE, "%s range exceeded: the InstanceNorm sums of conv block %d (%s) are not finite - a value of its input "
                        "(the block before it or the transposed conv feeding it) left the format%s", fmt16 == 2 ? "fp16 (level 0 of DLV_PREC_BF16)" : (f16 ? "fp16" : "bf16"), layer, names[layer],
                        f16 ? "; use dlv_unet_set_conv_shift on the producing block (dlv_range_report) or DLV_PREC_BF16_ALL (8 exponent bits) for this checkpoint" : "");
    return dlv_fail(ctx, DLV_ERANGE, "%s range exceeded: non-finite logits (raw output of the last conv block, upcat_1.conv_1)%s",
                    fmt16 == 2 ? "fp16 (level 0 of DLV_PREC_BF16)" : (f16 ? "fp16" : "bf16"), f16 ? "; use dlv_unet_set_conv_shift on upcat_1.conv_1 (dlv_range_report) or DLV_PREC_BF16_ALL (8 exponent bits) for this checkpoint" : "");
}

int dlv_unet_forward_bf16(dlv_ctx* ctx, const float* x, float* logits, int B, int d, int h, int w, int fmt16) {
    DLV_TRY(dlv_range_reset(ctx));
    if (fmt16 == 1) DLV_TRY(forward_16<PF16>(ctx, x, nullptr, 0, 0, nullptr, -1, 1.f, logits, nullptr, B, d, h, w));
    else if (fmt16 == 2) DLV_TRY((forward_16<PF16, PBf16>(ctx, x, nullptr, 0, 0, nullptr, -1, 1.f, logits, nullptr, B, d, h, w)));
    else DLV_TRY(forward_16<PBf16>(ctx, x, nullptr, 0, 0, nullptr, -1, 1.f, logits, nullptr, B, d, h, w));
    return dlv_range_check(ctx, fmt16);  // (dlv_unet_forward_dev is synchronous)
}

int dlv_unet_tiles_bf16(dlv_ctx* ctx, const uint16_t* vol, int Yp, int Xp, const int* starts_dev, int B, int d, int h,
                        int w, int flip_dim, float scale, float* acc, int fmt16) {
    if (fmt16 == 1) return forward_16<PF16>(ctx, nullptr, vol, Yp, Xp, starts_dev, flip_dim, scale, nullptr, acc, B, d, h, w);
    if (fmt16 == 2) return forward_16<PF16, PBf16>(ctx, nullptr, vol, Yp, Xp, starts_dev, flip_dim, scale, nullptr, acc, B, d, h, w);
    return forward_16<PBf16>(ctx, nullptr, vol, Yp, Xp, starts_dev, flip_dim, scale, nullptr, acc, B, d, h, w);
}

// test hook: one conv block (raw conv + InstanceNorm + Mish) or one deconv of the 16-bit path on fp32
// NCDHW tensors (converted on the device); precision DLV_PREC_BF16 or DLV_PREC_F16
extern "C" int dlv_debug_layer_bf16(dlv_ctx* ctx, int kind, int index, const float* in1_dev, int c1, const float* in2_dev,
                                    int c2, float* out_dev, int B, int D, int H, int W) {
    if (!ctx || !in1_dev || !out_dev) return DLV_EINVAL;
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "no weights");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->debug_f16) return debug_layer_16<PF16>(ctx, kind, index, in1_dev, c1, in2_dev, c2, out_dev, B, D, H, W);
    return debug_layer_16<PBf16>(ctx, kind, index, in1_dev, c1, in2_dev, c2, out_dev, B, D, H, W);
}
