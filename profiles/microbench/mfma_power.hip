// Sustained MFMA rate of the whole chip, one wave per SIMD (the occupancy of the register-resident conv kernel):
// 16x16x32 vs 32x32x16 f16, registers only, and with one ds_read_b128 per R MFMAs.  Answers: which instruction shape
// holds the higher clock under the power limit?     hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int LDS_EVERY>  // SHAPE 0: 16x16x32 (12 accumulators), 1: 32x32x16 (6 accumulators)
__global__ void __launch_bounds__(256, 1) k(float* out, int iters) {
    __shared__ h8 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) {  // pseudo-random operands: constant data draws less power
        h8 v;
        for (int e = 0; e < 8; ++e) {
            unsigned h = (unsigned)(i * 8 + e) * 2654435761u + blockIdx.x * 40503u;
            h ^= h >> 13;
            h *= 0x5bd1e995u;
            h ^= h >> 15;
            v[e] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f));
        }
        lds[i] = v;
    }
    __syncthreads();
    h8 a = lds[threadIdx.x], b = lds[threadIdx.x + 256];
    float s = 0.f;
    if (SHAPE == 0) {
        f4 acc[12];
        for (int j = 0; j < 12; ++j) acc[j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 36; ++j) {
                if (LDS_EVERY > 0 && j % LDS_EVERY == 0) b = lds[(threadIdx.x + j * 7 + it) & 1023];
                // (inline asm: the builtin form makes hipcc shuffle the twelve accumulators through v_accvgpr_mov every iteration)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j % 12]) : "v"(a), "v"(b));
            }
        }
        for (int j = 0; j < 12; ++j) s += acc[j][0] + acc[j][3];
    } else {
        f16v acc[6];
        for (int j = 0; j < 6; ++j)
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                if (LDS_EVERY > 0 && j % LDS_EVERY == 0) b = lds[(threadIdx.x + j * 7 + it) & 1023];
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j % 6]) : "v"(a), "v"(b));
            }
        }
        for (int j = 0; j < 6; ++j) s += acc[j][0] + acc[j][9];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int LDS_EVERY>
void run(const char* name, int grid) {
    float* out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDS_EVERY>), dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 36 * 16384.0;  // both shapes: 36 x 16x16x32 == 18 x 32x32x16 per iteration
        if (rep == 2) printf("%-28s grid %4d  %8.2f ms  %8.1f TFLOP/s\n", name, grid, ms, flops / (ms * 1e-3) / 1e12);
    }
    hipFree(out);
}

int main(int argc, char** argv) {
    for (int grid : {256, 192, 128, 64}) {
        run<0, 0>("16x16x32 regs only", grid);
        run<1, 0>("32x32x16 regs only", grid);
        run<0, 9>("16x16x32 + ds_read/9", grid);
        run<1, 6>("32x32x16 + ds_read/6", grid);
    }
    return 0;
}
