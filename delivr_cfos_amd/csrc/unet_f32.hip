// unet_f32.hip - fp32 (VALU) U-Net forward in NCDHW: the PARITY mode of the path.
//
// Restates MONAI BasicUNet.forward (call site inference/sliding_window_inferer.py:222, ctor
// inference/inference.py:190-197) op by op with plain fp32 arithmetic so that the result can be
// compared with the torch-fp32 oracle to ~1e-5.  It is deliberately simple (direct convolution,
// one thread per output voxel x 8 output channels); the throughput path is unet_bf16.hip.
#include "common.h"

namespace {

constexpr int COB = 8;  // output channels per thread

// y[n,co,z,y,x] = b[co] + sum_ci sum_taps W[co,ci,t] * x[n,ci,z+dz,y+dy,x+dx]   (zero padding)
// the input is the channel concatenation [in1 (c1 channels), in2 (c2 channels)]
__global__ void __launch_bounds__(256) conv3_f32_kernel(const float* __restrict__ in1, int c1,
                                                        const float* __restrict__ in2, int c2,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int cout, int D, int H, int W) {
    const long long vox = (long long)D * H * W;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int co0 = blockIdx.y * COB;
    const int n = blockIdx.z;
    if (v >= vox) return;
    const int x = (int)(v % W);
    const int y = (int)((v / W) % H);
    const int z = (int)(v / ((long long)W * H));
    const int cin = c1 + c2;
    float acc[COB];
#pragma unroll
    for (int k = 0; k < COB; ++k) acc[k] = bias[co0 + k];
    bool okz[3], oky[3], okx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        okz[k] = (unsigned)(z + k - 1) < (unsigned)D;
        oky[k] = (unsigned)(y + k - 1) < (unsigned)H;
        okx[k] = (unsigned)(x + k - 1) < (unsigned)W;
    }
    for (int ci = 0; ci < cin; ++ci) {
        const float* src = ci < c1 ? in1 + ((long long)n * c1 + ci) * vox : in2 + ((long long)n * c2 + (ci - c1)) * vox;
        const float* wp = w + ((long long)co0 * cin + ci) * 27;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float xv = 0.f;
                    if (okz[dz] && oky[dy] && okx[dx])
                        xv = src[((long long)(z + dz - 1) * H + (y + dy - 1)) * W + (x + dx - 1)];
                    const int t = (dz * 3 + dy) * 3 + dx;
#pragma unroll
                    for (int k = 0; k < COB; ++k) acc[k] = fmaf(xv, wp[(long long)k * cin * 27 + t], acc[k]);
                }
    }
#pragma unroll
    for (int k = 0; k < COB; ++k) out[((long long)n * cout + co0 + k) * vox + v] = acc[k];
}

// per (n,c): mean and 1/sqrt(var+eps) with biased variance, accumulated in fp64
__global__ void __launch_bounds__(256) inorm_stats_f32_kernel(const float* __restrict__ x, long long vox,
                                                              float eps, float2* __restrict__ stats) {
    const float* p = x + (long long)blockIdx.x * vox;
    double s = 0.0, ss = 0.0;
    for (long long i = threadIdx.x; i < vox; i += blockDim.x) {
        const double v = p[i];
        s += v;
        ss += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        ss += __shfl_down(ss, o, 64);
    }
    __shared__ double sh[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sh[wave] = s;
        sh[4 + wave] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = sh[0] + sh[1] + sh[2] + sh[3];
        ss = sh[4] + sh[5] + sh[6] + sh[7];
        const double mean = s / (double)vox;
        double var = ss / (double)vox - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[blockIdx.x] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
}

__device__ __forceinline__ float mish_f32(float y) {
    // torch.nn.Mish: y * tanh(softplus(y)), softplus threshold 20
    const float sp = y > 20.f ? y : log1pf(expf(y));
    return y * tanhf(sp);
}

// in place: x = mish((x-mean)*rstd*gamma + beta)
__global__ void __launch_bounds__(256) norm_mish_f32_kernel(float* __restrict__ x, long long vox, int C,
                                                            const float2* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta) {
    const int nc = blockIdx.y;
    const int c = nc % C;
    const float2 st = stats[nc];
    const float sc = st.y * gamma[c];
    const float sh = beta[c] - st.x * sc;
    float* p = x + (long long)nc * vox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vox; i += (long long)gridDim.x * blockDim.x)
        p[i] = mish_f32(fmaf(p[i], sc, sh));
}

__global__ void __launch_bounds__(256) maxpool2_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int D, int H, int W) {
    const int d = D / 2, h = H / 2, w = W / 2;
    const long long ovox = (long long)d * h * w;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= ovox) return;
    const int nc = blockIdx.y;
    const int x = (int)(v % w), y = (int)((v / w) % h), z = (int)(v / ((long long)w * h));
    const float* p = in + (long long)nc * D * H * W;
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) m = fmaxf(m, p[((long long)(2 * z + a) * H + (2 * y + b)) * W + 2 * x + c]);
    out[(long long)nc * ovox + v] = m;
}

// ConvTranspose3d k2 s2: out[n,co,2z+a,2y+b,2x+c] = bias[co] + sum_ci in[n,ci,z,y,x] * W[ci,co,a,b,c]
// UpCat's replicate padding of an up-sampled tensor whose skip tensor has an odd size (MONAI basic_unet.py UpCat.forward; windows
// that are not multiples of 16 reach it): one voxel at the far end of each short dimension, edge value
__global__ void __launch_bounds__(256) replicate_pad_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int Di, int Hi, int Wi,
                                                                int Do, int Ho, int Wo) {
    const long long vo = (long long)Do * Ho * Wo, vi = (long long)Di * Hi * Wi;
    const long long plane = blockIdx.y;  // (n, c)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vo; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), z = (int)(i / ((long long)Wo * Ho));
        out[plane * vo + i] = in[plane * vi + ((long long)min(z, Di - 1) * Hi + min(y, Hi - 1)) * Wi + min(x, Wi - 1)];
    }
}

__global__ void __launch_bounds__(256) deconv2_f32_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int cin, int cout, int D, int H, int W) {
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    const long long ovox = (long long)OD * OH * OW;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= ovox) return;
    const int co0 = blockIdx.y * COB;
    const int n = blockIdx.z;
    const int ox = (int)(v % OW), oy = (int)((v / OW) % OH), oz = (int)(v / ((long long)OW * OH));
    const int par = ((oz & 1) * 2 + (oy & 1)) * 2 + (ox & 1);
    const long long ivox = (long long)D * H * W;
    const long long iv = ((long long)(oz >> 1) * H + (oy >> 1)) * W + (ox >> 1);
    float acc[COB];
#pragma unroll
    for (int k = 0; k < COB; ++k) acc[k] = bias[co0 + k];
    for (int ci = 0; ci < cin; ++ci) {
        const float xv = in[((long long)n * cin + ci) * ivox + iv];
        const float* wp = w + ((long long)ci * cout + co0) * 8 + par;
#pragma unroll
        for (int k = 0; k < COB; ++k) acc[k] = fmaf(xv, wp[k * 8], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < COB; ++k) out[((long long)n * cout + co0 + k) * ovox + v] = acc[k];
}

__global__ void __launch_bounds__(256) conv1_f32_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int cin, long long vox) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= vox) return;
    const int n = blockIdx.y;
    float acc = bias[0];
    for (int ci = 0; ci < cin; ++ci) acc = fmaf(in[((long long)n * cin + ci) * vox + v], w[ci], acc);
    out[(long long)n * vox + v] = acc;
}

struct F32Net {
    dlv_ctx* ctx;
    int B;
    float2* stats;

    int conv(int li, const float* in1, int c1, const float* in2, int c2, float* out, int D, int H, int W) {
        const DlvConvLayer& L = ctx->conv[li];
        if (c1 + c2 != L.cin) return dlv_fail(ctx, DLV_ESTATE, "conv %d: %d+%d input channels, expected %d", li, c1, c2, L.cin);
        const long long vox = (long long)D * H * W;
        dim3 grid(dlv_cdiv(vox, 256), L.cout / COB, B);
        DlvProf p(ctx, "conv3_f32", 2.0 * 27 * L.cin * L.cout * vox * B, 4.0 * vox * B * (L.cin + L.cout));
        hipLaunchKernelGGL(conv3_f32_kernel, grid, dim3(256), 0, ctx->stream, in1, c1, in2, c2, L.w_f32, L.bias, out,
                           L.cout, D, H, W);
        p.end();
        DLV_LAUNCH_CHECK(ctx, "conv3_f32_kernel");
        // InstanceNorm3d(affine, eps 1e-5) -> Dropout (identity in eval) -> Mish
        hipLaunchKernelGGL(inorm_stats_f32_kernel, dim3(B * L.cout), dim3(256), 0, ctx->stream, out, vox, 1e-5f, stats);
        DLV_LAUNCH_CHECK(ctx, "inorm_stats_f32_kernel");
        dim3 g2((unsigned)std::min<long long>(dlv_cdiv(vox, 256), 1024), B * L.cout);
        hipLaunchKernelGGL(norm_mish_f32_kernel, g2, dim3(256), 0, ctx->stream, out, vox, L.cout, stats, L.gamma, L.beta);
        DLV_LAUNCH_CHECK(ctx, "norm_mish_f32_kernel");
        return DLV_OK;
    }
    int pool(const float* in, float* out, int C, int D, int H, int W) {
        dim3 grid(dlv_cdiv((long long)(D / 2) * (H / 2) * (W / 2), 256), B * C);
        hipLaunchKernelGGL(maxpool2_f32_kernel, grid, dim3(256), 0, ctx->stream, in, out, D, H, W);
        DLV_LAUNCH_CHECK(ctx, "maxpool2_f32_kernel");
        return DLV_OK;
    }
    // MONAI UpCat's replicate padding: `in` (B, C, Di, Hi, Wi) -> `out` (B, C, Do, Ho, Wo), Do - Di etc. in {0, 1}, edge values repeated
    int pad(const float* in, float* out, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
        hipLaunchKernelGGL(replicate_pad_f32_kernel, dim3(std::max(1, std::min(dlv_cdiv((long long)Do * Ho * Wo, 256), 1024)), B * C), dim3(256), 0,
                           ctx->stream, in, out, Di, Hi, Wi, Do, Ho, Wo);
        DLV_LAUNCH_CHECK(ctx, "replicate_pad_f32_kernel");
        return DLV_OK;
    }

    int deconv(int j, const float* in, float* out, int D, int H, int W) {
        const DlvDeconvLayer& L = ctx->deconv[j];
        dim3 grid(dlv_cdiv((long long)D * H * W * 8, 256), L.cout / COB, B);
        hipLaunchKernelGGL(deconv2_f32_kernel, grid, dim3(256), 0, ctx->stream, in, L.w_f32, L.bias, out, L.cin, L.cout,
                           D, H, W);
        DLV_LAUNCH_CHECK(ctx, "deconv2_f32_kernel");
        return DLV_OK;
    }
};

}  // namespace

int dlv_unet_forward_f32(dlv_ctx* ctx, const float* x, float* logits, int B, int d, int h, int w) {
    const int* f = ctx->features;
    const int lvlC[5] = {std::max(f[0], f[5]), f[1], f[2], f[3], f[4]};
    long long vox[5];
    for (int l = 0; l < 5; ++l) vox[l] = (long long)(d >> l) * (h >> l) * (w >> l);
    // four buffers per level (A: conv scratch / up-path result, Bf: conv scratch, S: skip, U: up-sampled)
    size_t off = 0, offs[5][4];
    for (int l = 0; l < 5; ++l)
        for (int k = 0; k < 4; ++k) {
            offs[l][k] = off;
            off += (size_t)B * lvlC[l] * vox[l] * sizeof(float);
            off = (off + 255) & ~(size_t)255;
        }
    char* base;
    DLV_TRY(dlv_ws_get(ctx, WS_F32_ACT, off, (void**)&base));
    float2* stats;
    DLV_TRY(dlv_ws_get(ctx, WS_STATS, (size_t)B * 256 * sizeof(float2), (void**)&stats));
    auto buf = [&](int l, int k) { return (float*)(base + offs[l][k]); };
    enum { A = 0, Bf = 1, S = 2, U = 3 };
    F32Net net{ctx, B, stats};
    int D[5], H[5], W[5];
    for (int l = 0; l < 5; ++l) {
        D[l] = d >> l;
        H[l] = h >> l;
        W[l] = w >> l;
    }
    // encoder
    DLV_TRY(net.conv(0, x, 1, nullptr, 0, buf(0, A), D[0], H[0], W[0]));
    DLV_TRY(net.conv(1, buf(0, A), f[0], nullptr, 0, buf(0, S), D[0], H[0], W[0]));
    const int encC[5] = {f[0], f[1], f[2], f[3], f[4]};
    for (int l = 1; l <= 4; ++l) {
        DLV_TRY(net.pool(buf(l - 1, S), buf(l, A), encC[l - 1], D[l - 1], H[l - 1], W[l - 1]));
        DLV_TRY(net.conv(2 * l, buf(l, A), encC[l - 1], nullptr, 0, buf(l, Bf), D[l], H[l], W[l]));
        DLV_TRY(net.conv(2 * l + 1, buf(l, Bf), encC[l], nullptr, 0, buf(l, S), D[l], H[l], W[l]));
    }
    // decoder: upcat_4 .. upcat_1  (cat order [skip, up-sampled])
    const float* cur = buf(4, S);
    for (int j = 0; j < 4; ++j) {
        const int l = 3 - j;  // output level
        if (2 * D[l + 1] == D[l] && 2 * H[l + 1] == H[l] && 2 * W[l + 1] == W[l]) {
            DLV_TRY(net.deconv(j, cur, buf(l, U), D[l + 1], H[l + 1], W[l + 1]));
        } else {  // an odd level: up-sample into the free buffer of the level, replicate-pad into U (MONAI's UpCat)
            DLV_TRY(net.deconv(j, cur, buf(l, A), D[l + 1], H[l + 1], W[l + 1]));
            DLV_TRY(net.pad(buf(l, A), buf(l, U), ctx->deconv[j].cout, 2 * D[l + 1], 2 * H[l + 1], 2 * W[l + 1], D[l], H[l], W[l]));
        }
        const int li = 10 + 2 * j;
        DLV_TRY(net.conv(li, buf(l, S), encC[l], buf(l, U), ctx->deconv[j].cout, buf(l, Bf), D[l], H[l], W[l]));
        DLV_TRY(net.conv(li + 1, buf(l, Bf), ctx->conv[li].cout, nullptr, 0, buf(l, A), D[l], H[l], W[l]));
        cur = buf(l, A);
    }
    dim3 grid(dlv_cdiv(vox[0], 256), B);
    hipLaunchKernelGGL(conv1_f32_kernel, grid, dim3(256), 0, ctx->stream, cur, ctx->final_w, ctx->final_b, logits, f[5],
                       vox[0]);
    DLV_LAUNCH_CHECK(ctx, "conv1_f32_kernel");
    return DLV_OK;
}
