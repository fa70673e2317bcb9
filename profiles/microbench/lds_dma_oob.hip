#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void k(const uint4* in, uint4* out, int nrec) {
    __shared__ __attribute__((aligned(16))) uint4 buf[128];
    buf[threadIdx.x] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    buf[threadIdx.x + 64] = make_uint4(0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu, 0xBBBBBBBBu);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(in), 0, nrec, 0x00020000);
    unsigned off = threadIdx.x * 16u;
    if (threadIdx.x % 3 == 1) off = 0xfffffff0u;   // out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)buf, 16, (int)off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    out[threadIdx.x] = buf[threadIdx.x];
    out[threadIdx.x + 64] = buf[threadIdx.x + 64];
}
int main() {
    std::vector<uint4> h(64);
    for (int i = 0; i < 64; ++i) h[i] = make_uint4(i + 1, i + 1, i + 1, i + 1);
    uint4 *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024);
    std::vector<uint4> r(128);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("lane %d: %08x %08x\n", i, r[i].x, r[i].w);
    printf("second half untouched: %08x\n", r[64].x);
    return 0;
}
