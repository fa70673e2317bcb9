// Does a spatial split of the chip pay?  Stream A (CU mask: the first NA CUs... of every XCD) runs the register-only MFMA
// loop, stream B (the remaining CUs) streams memory (read + write, HBM-bound).  Times: each alone on its mask, both
// concurrently, each alone on the whole chip.     hipcc -O3 --offload-arch=gfx950 cu_partition.hip -o cu_partition
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 1) mfma_k(float* out, int iters) {
    __shared__ h8 lds[512];
    for (int i = threadIdx.x; i < 512; i += 256) {
        h8 v;
        for (int e = 0; e < 8; ++e) {
            unsigned h = (unsigned)(i * 8 + e) * 2654435761u + blockIdx.x * 40503u;
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            v[e] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f));
        }
        lds[i] = v;
    }
    __syncthreads();
    h8 a = lds[threadIdx.x], b = lds[threadIdx.x + 256];
    f4 acc[12];
    for (int j = 0; j < 12; ++j) acc[j] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 36; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j % 12]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int j = 0; j < 12; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) stream_k(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        v.x ^= 1u;
        out[i] = v;
    }
}

static int g_interleaved = 1;  // mask bit b <-> XCD b % 8, CU b / 8 of that XCD (1) or XCD b / 32, CU b % 32 (0)
static hipStream_t masked_stream(int cu_lo, int cu_hi) {  // CUs [cu_lo, cu_hi) of every XCD
    std::vector<uint32_t> mask(8, 0);
    for (int b = 0; b < 256; ++b) {
        const int cu = g_interleaved ? b / 8 : b % 32;
        if (cu >= cu_lo && cu < cu_hi) mask[b / 32] |= 1u << (b % 32);
    }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) {
        printf("hipExtStreamCreateWithCUMask failed\n");
        exit(1);
    }
    return s;
}

int main(int argc, char** argv) {
    if (argc > 1) g_interleaved = atoi(argv[1]);
    const size_t n = (size_t)1 << 27;  // 2 GiB in, 2 GiB out
    uint4 *in, *out;
    float* mo;
    hipMalloc(&in, n * 16);
    hipMalloc(&out, n * 16);
    hipMalloc(&mo, 256 * 256 * 4);
    hipMemset(in, 1, n * 16);
    const int iters = 12000;
    hipEvent_t e0, e1, f0, f1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    for (int na : {32, 24, 20, 16}) {
        hipStream_t sa = na < 32 ? masked_stream(0, na) : nullptr, sb = na < 32 ? masked_stream(na, 32) : nullptr;
        if (!sa) { hipStreamCreate(&sa); hipStreamCreate(&sb); }
        const int ga = na * 8, gb = (na < 32 ? (32 - na) : 32) * 8 * 8;
        float ta = 0, tb = 0, ca = 0, cb = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, sa); hipLaunchKernelGGL(mfma_k, dim3(ga), dim3(256), 0, sa, mo, iters); hipEventRecord(e1, sa);
            hipStreamSynchronize(sa); hipEventElapsedTime(&ta, e0, e1);
            hipEventRecord(f0, sb); hipLaunchKernelGGL(stream_k, dim3(gb), dim3(256), 0, sb, in, out, n); hipEventRecord(f1, sb);
            hipStreamSynchronize(sb); hipEventElapsedTime(&tb, f0, f1);
            // concurrently (the streaming kernel is repeated so that it covers the MFMA kernel)
            hipEventRecord(e0, sa); hipLaunchKernelGGL(mfma_k, dim3(ga), dim3(256), 0, sa, mo, iters); hipEventRecord(e1, sa);
            hipEventRecord(f0, sb);
            for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(stream_k, dim3(gb), dim3(256), 0, sb, in, out, n);
            hipEventRecord(f1, sb);
            hipStreamSynchronize(sa); hipStreamSynchronize(sb);
            hipEventElapsedTime(&ca, e0, e1); hipEventElapsedTime(&cb, f0, f1);
        }
        const double fl = (double)ga * 4 * iters * 36 * 16384.0, by = 2.0 * n * 16;
        printf("MFMA on %3d CUs, stream on %3d CUs: alone %7.1f TFLOP/s | %6.2f TB/s ; together %7.1f TFLOP/s | %6.2f TB/s\n", na * 8,
               na < 32 ? (32 - na) * 8 : 256, fl / (ta * 1e-3) / 1e12, by / (tb * 1e-3) / 1e12, fl / (ca * 1e-3) / 1e12,
               3 * by / (cb * 1e-3) / 1e12);
    }
    return 0;
}
