// What bounds the final 1x1x1 conv (norm + Mish of 32 channels, dot product, blend RMW)?  The kernel's loop with its parts
// switched off one at a time: 16 windows of 128^3, chunk-planar fp16 input (2.15 GB read), fp32 accumulator volume RMW (0.27 GB).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2_t mish_fast2(f32x2_t y) {
    const f32x2_t c20 = {20.f, 20.f}, l2e = {1.44269504f, 1.44269504f}, two = {2.f, 2.f};
    const f32x2_t e = __builtin_elementwise_min(y, c20) * l2e;
    const f32x2_t n = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const f32x2_t t = n * (n + two);
    const f32x2_t d = t + two;
    const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return y * (t * r);
}
// MODE bit 0: no Mish; bit 1: no RMW (plain store of the logit); bit 2: only 1 of the 4 chunk loads (bytes / 4)
template <int MODE>
__global__ void __launch_bounds__(256) k_final(const uint4* __restrict__ x, const float2* __restrict__ ss, const float* __restrict__ wf,
                                               float* __restrict__ acc, long long vox) {
    const int n = blockIdx.y;
    f32x2_t sc[16], sh[16], ww[16];
    for (int c = 0; c < 16; ++c) {
        const float2 v0 = ss[n * 32 + 2 * c], v1 = ss[n * 32 + 2 * c + 1];
        sc[c] = f32x2_t{v0.x, v1.x};
        sh[c] = f32x2_t{v0.y, v1.y};
        float w0 = wf[2 * c], w1 = wf[2 * c + 1];
        asm volatile("" : "+v"(w0), "+v"(w1));
        ww[c] = f32x2_t{w0, w1};
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vox; i += (long long)gridDim.x * 256) {
        f32x2_t a2 = {0.1f, 0.f};
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            const int cc = (MODE & 4) ? 0 : c8;
            const u32x4_t u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(x + ((long long)n * 4 + cc) * vox + i));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const h2_t h = __builtin_bit_cast(h2_t, u[k]);
                f32x2_t v = __builtin_elementwise_fma(f32x2_t{(float)h.x, (float)h.y}, sc[4 * c8 + k], sh[4 * c8 + k]);
                if (!(MODE & 1)) v = mish_fast2(v);
                a2 = __builtin_elementwise_fma(v, ww[4 * c8 + k], a2);
            }
        }
        const float a = a2.x + a2.y;
        const long long o = (long long)n * vox + i;
        if (MODE & 2) acc[o] = a; else acc[o] += 0.25f * a;
    }
}

// the same loop software-pipelined: the four chunk loads and the accumulator word of iteration i + 1 are in flight while
// iteration i is computed
__global__ void __launch_bounds__(256) k_final_pipe(const uint4* __restrict__ x, const float2* __restrict__ ss, const float* __restrict__ wf,
                                                    float* __restrict__ acc, long long vox) {
    const int n = blockIdx.y;
    f32x2_t sc[16], sh[16], ww[16];
    for (int c = 0; c < 16; ++c) {
        const float2 v0 = ss[n * 32 + 2 * c], v1 = ss[n * 32 + 2 * c + 1];
        sc[c] = f32x2_t{v0.x, v1.x};
        sh[c] = f32x2_t{v0.y, v1.y};
        float w0 = wf[2 * c], w1 = wf[2 * c + 1];
        asm volatile("" : "+v"(w0), "+v"(w1));
        ww[c] = f32x2_t{w0, w1};
    }
    const long long step = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= vox) return;
    u32x4_t u[4];
    float av;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) u[c8] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(x + ((long long)n * 4 + c8) * vox + i));
    av = acc[(long long)n * vox + i];
    for (; i < vox; i += step) {
        u32x4_t un[4];
        float avn = 0.f;
        const long long in = i + step;
        if (in < vox) {
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) un[c8] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(x + ((long long)n * 4 + c8) * vox + in));
            avn = acc[(long long)n * vox + in];
        }
        f32x2_t a2 = {0.1f, 0.f};
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const h2_t h = __builtin_bit_cast(h2_t, u[c8][k]);
                f32x2_t v = __builtin_elementwise_fma(f32x2_t{(float)h.x, (float)h.y}, sc[4 * c8 + k], sh[4 * c8 + k]);
                v = mish_fast2(v);
                a2 = __builtin_elementwise_fma(v, ww[4 * c8 + k], a2);
            }
        acc[(long long)n * vox + i] = av + 0.25f * (a2.x + a2.y);
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) u[c8] = un[c8];
        av = avn;
    }
}
int main() {
    const int B = 16; const long long vox = 128LL * 128 * 128;
    uint4* x; float* acc; hipMalloc(&x, (size_t)B * 4 * vox * 16); hipMalloc(&acc, (size_t)B * vox * 4);
    std::vector<unsigned short> h(1 << 20);
    for (auto& v : h) { __half t = __float2half((float)(rand() % 2000 - 1000) / 250.f); v = *reinterpret_cast<unsigned short*>(&t); }
    for (size_t o = 0; o < (size_t)B * 4 * vox * 16; o += h.size() * 2) hipMemcpy((char*)x + o, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemset(acc, 0, (size_t)B * vox * 4);
    std::vector<float2> ss(B * 32, make_float2(1.01f, -0.02f)); float2* dss; hipMalloc(&dss, ss.size() * 8); hipMemcpy(dss, ss.data(), ss.size() * 8, hipMemcpyHostToDevice);
    std::vector<float> w(32, 0.05f); float* dw; hipMalloc(&dw, 128); hipMemcpy(dw, w.data(), 128, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* nm, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r) best = ms < best ? ms : best; }
        printf("%-36s %8.1f us\n", nm, best * 1e3);
    };
    for (int gx : {1024, 2048, 4096}) {
        printf("grid %d x %d\n", gx, B);
        run("full", [&] { hipLaunchKernelGGL(k_final<0>, dim3(gx, B), dim3(256), 0, 0, x, dss, dw, acc, vox); });
        run("no Mish", [&] { hipLaunchKernelGGL(k_final<1>, dim3(gx, B), dim3(256), 0, 0, x, dss, dw, acc, vox); });
        run("no RMW (plain store)", [&] { hipLaunchKernelGGL(k_final<2>, dim3(gx, B), dim3(256), 0, 0, x, dss, dw, acc, vox); });
        run("no Mish, no RMW", [&] { hipLaunchKernelGGL(k_final<3>, dim3(gx, B), dim3(256), 0, 0, x, dss, dw, acc, vox); });
        run("full, software-pipelined", [&] { hipLaunchKernelGGL(k_final_pipe, dim3(gx, B), dim3(256), 0, 0, x, dss, dw, acc, vox); });
        run("a quarter of the input bytes", [&] { hipLaunchKernelGGL(k_final<4>, dim3(gx, B), dim3(256), 0, 0, x, dss, dw, acc, vox); });
    }
    return 0;
}
