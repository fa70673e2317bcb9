// paint.hip - stats-driven blob painting: the loop of blob_highlighter.py:108-125 / :150-158
//     for cc_id in cell_csv['connected_component_id']:
//         bb = pad_bb(stats['bounding_boxes'][cc_id]);  IMG[bb] = bin_img[bb] * value(cc_id)
// paints every cell's padded bounding box in CSV order, so a voxel ends up with the value of the LAST listed cell
// whose box contains it (boxes of neighbouring blobs overlap; the reference's comment calls this "might accidentally
// re-color other blobs close by") - times bin_img.  Restated order-free: owner[v] = max{ i+1 : box_i contains v },
// built with atomicMax by one wave per box (boxes are tiny: a cell is 10-40 voxels), then IMG[v] = bin[v] * value[owner[v]-1].
// Integer only; identical to the sequential loop for any box list (tests/test_gpu_parity.py, tests/golden/ref_paint.npz).
#include "common.h"

// scipy-exact fp64 arithmetic below (Gaussian filter, distance transform): no a*b+c may become an fma - HIP contracts by
// default and the __d*_rn "intrinsics" are plain operators in its headers
#pragma clang fp contract(off)


namespace {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr long long PAINT_BIG = 1 << 16;  // boxes above this volume get a launch of their own

__device__ __forceinline__ void paint_box(const uint8_t* __restrict__ bin, u32* __restrict__ owner, int Y, int X,
                                          const int* __restrict__ b, u32 tag, long long first, long long step) {
    const int z0 = b[0], y0 = b[2], x0 = b[4];
    const long long dy = b[3] - b[2], dx = b[5] - b[4];
    const long long vol = (long long)(b[1] - b[0]) * dy * dx;
    for (long long i = first; i < vol; i += step) {
        const int x = x0 + (int)(i % dx);
        const int y = y0 + (int)((i / dx) % dy);
        const int z = z0 + (int)(i / (dx * dy));
        const u64 v = ((u64)z * Y + y) * X + x;
        if (bin[v]) atomicMax(owner + v, tag);
    }
}

// one wave per box; boxes of PAINT_BIG voxels or more are left to paint_big_kernel
__global__ void __launch_bounds__(256) paint_small_kernel(const uint8_t* __restrict__ bin, u32* __restrict__ owner, int Y,
                                                          int X, const int* __restrict__ boxes, long long n) {
    const int lane = threadIdx.x & 63;
    for (long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (long long)gridDim.x * 4) {
        const int* b = boxes + 6 * i;
        const long long vol = (long long)(b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]);
        // an empty or inverted extent on ANY axis paints nothing (two inverted axes would give a positive volume)
        if (b[1] <= b[0] || b[3] <= b[2] || b[5] <= b[4] || vol >= PAINT_BIG) continue;
        paint_box(bin, owner, Y, X, b, (u32)(i + 1), lane, 64);
    }
}

__global__ void __launch_bounds__(256) paint_big_kernel(const uint8_t* __restrict__ bin, u32* __restrict__ owner, int Y,
                                                        int X, const int* __restrict__ boxes, long long i) {
    paint_box(bin, owner, Y, X, boxes + 6 * i, (u32)(i + 1), (long long)blockIdx.x * blockDim.x + threadIdx.x,
              (long long)gridDim.x * blockDim.x);
}

template <class T>
__global__ void __launch_bounds__(256) paint_apply_kernel(const u32* __restrict__ owner, const uint8_t* __restrict__ bin,
                                                          u64 n, const T* __restrict__ values, T* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 o = owner[i];
        // numpy: bin (uint8, or .astype(uint16)) * value, stored into a T array -> wraps modulo 2^bits like T arithmetic
        out[i] = o ? (T)((T)bin[i] * values[o - 1]) : (T)0;
    }
}

}  // namespace

extern "C" {

int dlv_paint_owner_dev(dlv_ctx* ctx, const uint8_t* bin_dev, int Z, int Y, int X, const int32_t* boxes_dev,
                        const int32_t* boxes_host, uint64_t n_boxes, uint32_t* owner_dev) {
    if (!ctx || !bin_dev || !owner_dev || (n_boxes && (!boxes_dev || !boxes_host))) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0) return dlv_fail(ctx, DLV_EINVAL, "empty volume");
    if (n_boxes >= 0xffffffffull) return dlv_fail(ctx, DLV_EUNSUP, "more than 2^32-2 boxes");
    for (uint64_t i = 0; i < n_boxes; ++i) {
        const int32_t* b = boxes_host + 6 * i;
        if (b[0] < 0 || b[2] < 0 || b[4] < 0 || b[1] > Z || b[3] > Y || b[5] > X)
            return dlv_fail(ctx, DLV_EINVAL, "box %llu leaves the volume", (unsigned long long)i);
    }
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const u64 nvox = (u64)Z * Y * X;
    DLV_HIP(ctx, hipMemsetAsync(owner_dev, 0, nvox * 4, ctx->stream));
    if (!n_boxes) return DLV_OK;
    const int gs = (int)std::min<u64>((n_boxes + 3) / 4, (u64)256 * 64);
    hipLaunchKernelGGL(paint_small_kernel, dim3(gs), dim3(256), 0, ctx->stream, bin_dev, owner_dev, Y, X, boxes_dev,
                       (long long)n_boxes);
    DLV_LAUNCH_CHECK(ctx, "paint_small_kernel");
    for (uint64_t i = 0; i < n_boxes; ++i) {
        const int32_t* b = boxes_host + 6 * i;
        if (b[1] <= b[0] || b[3] <= b[2] || b[5] <= b[4]) continue;  // empty / inverted: nothing to paint
        const long long vol = (long long)(b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]);
        if (vol < PAINT_BIG) continue;
        const int gb = (int)std::min<long long>((vol + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(paint_big_kernel, dim3(gb), dim3(256), 0, ctx->stream, bin_dev, owner_dev, Y, X, boxes_dev, (long long)i);
        DLV_LAUNCH_CHECK(ctx, "paint_big_kernel");
    }
    return DLV_OK;
}

int dlv_paint_apply_dev(dlv_ctx* ctx, const uint32_t* owner_dev, const uint8_t* bin_dev, uint64_t nvox,
                        const void* values_dev, int elem_bytes, void* out_dev) {
    if (!ctx || !owner_dev || !bin_dev || !values_dev || !out_dev) return DLV_EINVAL;
    if (elem_bytes != 1 && elem_bytes != 2) return dlv_fail(ctx, DLV_EUNSUP, "values must be uint8 or uint16");
    if (!nvox) return DLV_OK;
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const int gs = (int)std::min<u64>((nvox + 255) / 256, (u64)256 * 32);
    if (elem_bytes == 1)
        hipLaunchKernelGGL(paint_apply_kernel<uint8_t>, dim3(gs), dim3(256), 0, ctx->stream, owner_dev, bin_dev, (u64)nvox,
                           (const uint8_t*)values_dev, (uint8_t*)out_dev);
    else
        hipLaunchKernelGGL(paint_apply_kernel<uint16_t>, dim3(gs), dim3(256), 0, ctx->stream, owner_dev, bin_dev, (u64)nvox,
                           (const uint16_t*)values_dev, (uint16_t*)out_dev);
    DLV_LAUNCH_CHECK(ctx, "paint_apply_kernel");
    return DLV_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// cell-density heat map: cells_to_atlas.py:174-200 create_heatmap
//   heatmap[z,y,x] = number of cells at that atlas voxel;  gaussian_filter(heatmap.astype('float32'), sigma=2.25)
// scipy.ndimage.gaussian_filter = three correlate1d passes (axis 0,1,2), each: line converted to double, symmetric
// kernel evaluated as  t = c*w[0]; for j = r..1: t += (in[-j] + in[+j]) * w[j]  (ni_filters.c NI_Correlate1D), boundary
// 'reflect' (d c b a | a b c d | d c b a), result stored as float32 before the next axis.  Same order, same
// roundings (no fma contraction) -> bit-identical to scipy (tests/golden/scipy_heatmap.npz).
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) heat_count_kernel(const int* __restrict__ xyz, long long n, int Z, int Y, int X,
                                                         unsigned int* __restrict__ hist) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if ((unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z)
            atomicAdd(hist + ((long long)z * Y + y) * X + x, 1u);
    }
}

__global__ void __launch_bounds__(256) u32_to_f32_kernel(const unsigned int* __restrict__ in, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    // scipy 'reflect': period 2n, -1 -> 0, n -> n-1 (also for lines shorter than the kernel radius)
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// one thread per output element; `stride` = element stride of the filtered axis, `len` its length
__global__ void __launch_bounds__(256) gauss_axis_kernel(const float* __restrict__ in, float* __restrict__ out, long long n,
                                                         long long stride, int len, const double* __restrict__ w, int radius) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int pos = (int)((i / stride) % len);
        const long long base = i - (long long)pos * stride;
        double t = __dmul_rn((double)in[i], w[0]);
        for (int j = radius; j >= 1; --j) {
            const double a = (double)in[base + (long long)reflect_idx(pos - j, len) * stride];
            const double b = (double)in[base + (long long)reflect_idx(pos + j, len) * stride];
            t = __dadd_rn(t, __dmul_rn(__dadd_rn(a, b), w[j]));
        }
        out[i] = (float)t;
    }
}

}  // namespace

extern "C" {

int dlv_heatmap_counts_dev(dlv_ctx* ctx, const int32_t* xyz_dev, uint64_t n_cells, int Z, int Y, int X, float* heat_dev) {
    if (!ctx || !heat_dev || (n_cells && !xyz_dev)) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0) return dlv_fail(ctx, DLV_EINVAL, "empty grid");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const long long nvox = (long long)Z * Y * X;
    unsigned int* hist;
    DLV_TRY(dlv_ws_get(ctx, WS_MISC, (size_t)nvox * 4, (void**)&hist));
    DLV_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)nvox * 4, ctx->stream));
    if (n_cells) {
        const int gs = (int)std::min<long long>(((long long)n_cells + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(heat_count_kernel, dim3(gs), dim3(256), 0, ctx->stream, xyz_dev, (long long)n_cells, Z, Y, X, hist);
        DLV_LAUNCH_CHECK(ctx, "heat_count_kernel");
    }
    const int gv = (int)std::min<long long>((nvox + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(u32_to_f32_kernel, dim3(gv), dim3(256), 0, ctx->stream, hist, heat_dev, nvox);
    DLV_LAUNCH_CHECK(ctx, "u32_to_f32_kernel");
    return DLV_OK;
}

int dlv_gauss_blur_f32_dev(dlv_ctx* ctx, float* vol_dev, int Z, int Y, int X, const double* weights_host, int radius,
                           float* tmp_dev) {
    if (!ctx || !vol_dev || !tmp_dev || !weights_host) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0 || radius < 0 || radius > 1024) return dlv_fail(ctx, DLV_EINVAL, "bad blur arguments");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const long long nvox = (long long)Z * Y * X;
    double* w;
    DLV_TRY(dlv_ws_get(ctx, WS_MISC, (size_t)(radius + 1) * 8, (void**)&w));
    DLV_HIP(ctx, hipMemcpyAsync(w, weights_host, (size_t)(radius + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    const int gv = (int)std::min<long long>((nvox + 255) / 256, 256 * 64);
    const long long strides[3] = {(long long)Y * X, X, 1};
    const int lens[3] = {Z, Y, X};
    float* src = vol_dev;
    float* dst = tmp_dev;
    for (int ax = 0; ax < 3; ++ax) {
        hipLaunchKernelGGL(gauss_axis_kernel, dim3(gv), dim3(256), 0, ctx->stream, src, dst, nvox, strides[ax], lens[ax], w, radius);
        DLV_LAUNCH_CHECK(ctx, "gauss_axis_kernel");
        std::swap(src, dst);
    }
    // three passes: the result sits in tmp_dev; copy it back so that vol_dev holds the blurred map
    DLV_HIP(ctx, hipMemcpyAsync(vol_dev, tmp_dev, (size_t)nvox * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // weights_host and the workspace copy are consumed
    return DLV_OK;
}

}  // extern "C"

// ---- exact Euclidean distance transform (depth-coded blob map, blob_depthmap.py:160-170) -----------------------
// scipy.ndimage.distance_transform_edt(np.pad(stack, 1), sampling=(sz, sy, sx))[1:-1, 1:-1, 1:-1].astype(uint16): distance
// of every non-zero voxel to the nearest zero voxel, the volume being surrounded by zeros.  Separable exact algorithm
// (Felzenszwalb / Huttenlocher lower envelope of parabolas) in fp64: x by two scans, then y and z by one envelope per
// line; the zeros of the padding are two extra parabolas of height 0 at positions -1 and n of every line.
namespace {

#pragma clang fp contract(off)

// f[z][y][x] = (dx * sx)^2, dx = voxels to the nearest zero along x (the padding counts)
__global__ void __launch_bounds__(256) edt_x_kernel(const uint16_t* __restrict__ in, double* __restrict__ f, long long rows, int X,
                                                    double sx) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const uint16_t* a = in + r * X;
    double* o = f + r * X;
    int d = 0;  // distance to the zero at x = -1 is x + 1
    for (int x = 0; x < X; ++x) {
        d = a[x] ? d + 1 : 0;
        o[x] = (double)d;
    }
    d = 0;
    for (int x = X - 1; x >= 0; --x) {
        d = a[x] ? d + 1 : 0;
        const double m = fmin(o[x], (double)d) * sx;
        o[x] = m * m;
    }
}

// one line of n values at stride `stride`: g[p] = min_q f[q] + ((p - q) * w)^2 over q in -1 .. n with f[-1] = f[n] = 0
__global__ void __launch_bounds__(64) edt_line_kernel(double* __restrict__ f, long long lines, long long inner, long long outer_stride,
                                                      long long stride, int n, double w, int* __restrict__ vbuf,
                                                      double* __restrict__ zbuf, double* __restrict__ gbuf) {
    const long long l = (long long)blockIdx.x * 64 + threadIdx.x;
    if (l >= lines) return;
    double* line = f + (l / inner) * outer_stride + (l % inner);
    // envelope scratch interleaved by line (element i of line l at i * lines + l): the 64 lanes of a wave walk their lines in
    // step, so every access of the wave is one contiguous run
    auto V = [&](int i) -> int& { return vbuf[(long long)i * lines + l]; };
    auto Zb = [&](int i) -> double& { return zbuf[(long long)i * lines + l]; };
    auto G = [&](int i) -> double& { return gbuf[(long long)i * lines + l]; };
    const int m = n + 2;  // extended positions 0 .. n+1 <-> voxel positions -1 .. n
    auto fe = [&](int q) -> double { return (q == 0 || q == m - 1) ? 0.0 : line[(long long)(q - 1) * stride]; };
    const double w2 = w * w;
    int k = 0;
    V(0) = 0;
    Zb(0) = -INFINITY;
    Zb(1) = INFINITY;
    for (int q = 1; q < m; ++q) {
        const double fq = fe(q) + w2 * ((double)q * (double)q);
        double s;
        while (true) {
            const int vk = V(k);
            s = (fq - (fe(vk) + w2 * ((double)vk * (double)vk))) / (2.0 * w2 * (double)(q - vk));
            if (s <= Zb(k) && k > 0) --k;
            else break;
        }
        ++k;
        V(k) = q;
        Zb(k) = s;
        Zb(k + 1) = INFINITY;
    }
    k = 0;
    for (int p = 1; p < m - 1; ++p) {
        while (Zb(k + 1) < (double)p) ++k;
        const int vk = V(k);
        const double dd = (double)(p - vk) * w;
        G(p) = dd * dd + fe(vk);
    }
    for (int p = 1; p < m - 1; ++p) line[(long long)(p - 1) * stride] = G(p);
}

__global__ void __launch_bounds__(256) edt_sqrt_u16_kernel(const double* __restrict__ f, uint16_t* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (uint16_t)(unsigned long long)sqrt(f[i]);  // numpy's astype(uint16): truncation, modulo 2^16
}

}  // namespace

extern "C" int dlv_edt_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int Z, int Y, int X, const double* sampling_zyx,
                               uint16_t* out_dev) {
    if (!ctx || !in_dev || !out_dev || !sampling_zyx) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0) return dlv_fail(ctx, DLV_EINVAL, "bad shape");
    for (int k = 0; k < 3; ++k)
        if (!(sampling_zyx[k] > 0.0) || sampling_zyx[k] > 1e9) return dlv_fail(ctx, DLV_EINVAL, "sampling must be positive");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const long long nvox = (long long)Z * Y * X;
    // envelope scratch of one pass: lines * (n + 3) elements - Z*X lines of Y along y, Y*X lines of Z along z: ~nvox either
    // way (not max(lines) * max(n), which asks for max(Y,Z)/min(Y,Z) times too much on an anisotropic stack)
    const size_t elems = (size_t)std::max((long long)Z * X * (Y + 3), (long long)Y * X * (Z + 3));
    // [f: nvox doubles | z: elems doubles | g: elems doubles | v: elems ints]
    const size_t off_z = (size_t)nvox * 8, off_g = off_z + elems * 8, off_v = off_g + elems * 8;
    char* ws;
    DLV_TRY(dlv_ws_get(ctx, WS_MISC, off_v + elems * 4, (void**)&ws));
    double* f = (double*)ws;
    double* zb = (double*)(ws + off_z);
    double* gb = (double*)(ws + off_g);
    int* vb = (int*)(ws + off_v);
    DlvProf p(ctx, "edt_u16", 0.0, 2.0 * nvox + 6.0 * 8.0 * nvox);
    hipLaunchKernelGGL(edt_x_kernel, dim3((unsigned)(((long long)Z * Y + 255) / 256)), dim3(256), 0, ctx->stream, in_dev, f,
                       (long long)Z * Y, X, sampling_zyx[2]);
    DLV_LAUNCH_CHECK(ctx, "edt_x_kernel");
    // along y: lines (z, x): base = z * Y*X + x, stride X
    hipLaunchKernelGGL(edt_line_kernel, dim3((unsigned)(((long long)Z * X + 63) / 64)), dim3(64), 0, ctx->stream, f, (long long)Z * X,
                       (long long)X, (long long)Y * X, (long long)X, Y, sampling_zyx[1], vb, zb, gb);
    DLV_LAUNCH_CHECK(ctx, "edt_line_kernel(y)");
    // along z: lines (y, x): base = y * X + x, stride Y*X
    hipLaunchKernelGGL(edt_line_kernel, dim3((unsigned)(((long long)Y * X + 63) / 64)), dim3(64), 0, ctx->stream, f, (long long)Y * X,
                       (long long)Y * X, 0LL, (long long)Y * X, Z, sampling_zyx[0], vb, zb, gb);
    DLV_LAUNCH_CHECK(ctx, "edt_line_kernel(z)");
    hipLaunchKernelGGL(edt_sqrt_u16_kernel, dim3((unsigned)std::min<long long>((nvox + 255) / 256, 65536)), dim3(256), 0, ctx->stream,
                       f, out_dev, nvox);
    p.end();
    DLV_LAUNCH_CHECK(ctx, "edt_sqrt_u16_kernel");
    return DLV_OK;
}
