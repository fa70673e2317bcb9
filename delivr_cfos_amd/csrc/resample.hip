// resample.hip - the volume down/up-samplers either side of the inference path.
//
//   block mean   : skimage.transform.downscale_local_mean(chunk,(fz,fy,fx)).astype(uint16)
//                  (downsample/downsample_and_mask.py:44)  == floor(sum over zero-padded block / f)
//   spline-2 zoom: scipy.ndimage.zoom(mask, ratios, output=uint8, order=2, prefilter=False)
//                  (downsample/downsample_and_mask.py:299); float64 arithmetic restated in the exact
//                  operation order that reproduces scipy bit for bit (see oracle.zoom_spline2_f64)
//   mask + pad   : img *= mask_us[i]; masked_nii[0,0,i,:Y,:X] = img  (downsample_and_mask.py:396-417)
//   trilinear    : north-star extension (no reference counterpart)
// All HBM-bound: one coalesced pass over z-major slabs, X contiguous.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) block_mean_u16_kernel(const uint16_t* __restrict__ in, int Z, int Y, int X, int fz,
                                                             int fy, int fx, uint16_t* __restrict__ out, int oz, int oy,
                                                             int ox) {
    const long long n = (long long)oz * oy * ox;
    const unsigned long long div = (unsigned long long)fz * fy * fx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox), y = (int)((i / ox) % oy), z = (int)(i / ((long long)ox * oy));
        unsigned long long s = 0;
        for (int a = 0; a < fz; ++a) {
            const int zz = z * fz + a;
            if (zz >= Z) break;
            for (int b = 0; b < fy; ++b) {
                const int yy = y * fy + b;
                if (yy >= Y) break;
                const uint16_t* row = in + ((long long)zz * Y + yy) * X;
                for (int c = 0; c < fx; ++c) {
                    const int xx = x * fx + c;
                    if (xx >= X) break;
                    s += row[xx];
                }
            }
        }
        out[i] = (uint16_t)(s / div);
    }
}

struct Taps {
    int k[3];
    double w[3];
};

__device__ __forceinline__ Taps spline2_taps(int i, int n_in, int n_out) {
    Taps t;
    const double scale = n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 0.0;
    const double x = __dmul_rn((double)i, scale);
    const double c = floor(__dadd_rn(x, 0.5));
    const double d = __dsub_rn(x, c);
    const double w1 = __dsub_rn(0.75, __dmul_rn(d, d));
    const double y = __dsub_rn(0.5, d);
    const double w0 = __dmul_rn(__dmul_rn(0.5, y), y);
    const double w2 = __dsub_rn(__dsub_rn(1.0, w0), w1);
    t.w[0] = w0;
    t.w[1] = w1;
    t.w[2] = w2;
    const int ci = (int)c;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int k = ci + j - 1;
        if (n_in == 1) {
            k = 0;
        } else {
            const int p = 2 * (n_in - 1);
            k %= p;
            if (k < 0) k += p;
            if (k >= n_in) k = p - k;
        }
        t.k[j] = k;
    }
    return t;
}

__global__ void __launch_bounds__(256) zoom_spline2_u8_kernel(const uint8_t* __restrict__ in, int iz, int iy, int ix,
                                                              uint8_t* __restrict__ out, int oz, int oy, int ox) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox), y = (int)((i / ox) % oy), z = (int)(i / ((long long)ox * oy));
        const Taps tz = spline2_taps(z, iz, oz), ty = spline2_taps(y, iy, oy), tx = spline2_taps(x, ix, ox);
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const uint8_t* row = in + ((long long)tz.k[a] * iy + ty.k[b]) * ix;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double v = (double)row[tx.k[c]];
                    t = __dadd_rn(t, __dmul_rn(__dmul_rn(__dmul_rn(v, tz.w[a]), ty.w[b]), tx.w[c]));
                }
            }
        // scipy CASE_INTERP_OUT_UINT: t > 0 ? t + 0.5 : 0, clip to [0, 255], truncate
        double r = t > 0.0 ? __dadd_rn(t, 0.5) : 0.0;
        r = r > 255.0 ? 255.0 : r;
        out[i] = (uint8_t)r;
    }
}

__global__ void __launch_bounds__(256) mask_pad_u16_kernel(const uint16_t* __restrict__ raw, const uint8_t* __restrict__ mask,
                                                           int threshold, int Z, int Y, int X, uint16_t* __restrict__ out,
                                                           int Zp, int Yp, int Xp) {
    const long long n = (long long)Zp * Yp * Xp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Xp), y = (int)((i / Xp) % Yp), z = (int)(i / ((long long)Xp * Yp));
        uint16_t v = 0;
        if (z < Z && y < Y && x < X) {
            const long long s = ((long long)z * Y + y) * X + x;
            v = raw[s];
            if (mask) v = (uint16_t)(v * mask[s]);  // uint16 *= uint8 (wraps like numpy)
            else if ((int)v < threshold) v = 0;
        }
        out[i] = v;
    }
}

// Both resamplers below compute in fp64 with contraction off, operation by operation as oracle/delivr_oracle.py states
// them: bit-exact against the numpy restatement (they are HBM-bound; the fp64 arithmetic is hidden).
struct Affine34 {
    double m[12];  // row-major 3x4: (z,y,x) of the INPUT = m[0:3].(z,y,x of the output) + m[3], ...
};

#pragma clang fp contract(off)
__device__ __forceinline__ double lerp3_u16(const uint16_t* __restrict__ in, int iz, int iy, int ix, double fz, double fy,
                                            double fx, bool zero_outside) {
    const double z0f = floor(fz), y0f = floor(fy), x0f = floor(fx);
    const int z0 = (int)z0f, y0 = (int)y0f, x0 = (int)x0f;
    const double tz = fz - z0f, ty = fy - y0f, tx = fx - x0f;
    auto at = [&](int a, int b, int c) -> double {
        if (zero_outside) {
            if ((unsigned)a >= (unsigned)iz || (unsigned)b >= (unsigned)iy || (unsigned)c >= (unsigned)ix) return 0.0;
        } else {
            a = min(max(a, 0), iz - 1);
            b = min(max(b, 0), iy - 1);
            c = min(max(c, 0), ix - 1);
        }
        return (double)in[((long long)a * iy + b) * ix + c];
    };
    const double c00 = at(z0, y0, x0) * (1.0 - tx) + at(z0, y0, x0 + 1) * tx;
    const double c01 = at(z0, y0 + 1, x0) * (1.0 - tx) + at(z0, y0 + 1, x0 + 1) * tx;
    const double c10 = at(z0 + 1, y0, x0) * (1.0 - tx) + at(z0 + 1, y0, x0 + 1) * tx;
    const double c11 = at(z0 + 1, y0 + 1, x0) * (1.0 - tx) + at(z0 + 1, y0 + 1, x0 + 1) * tx;
    const double c0 = c00 * (1.0 - ty) + c01 * ty, c1 = c10 * (1.0 - ty) + c11 * ty;
    return c0 * (1.0 - tz) + c1 * tz;
}

#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) trilinear_u16_kernel(const uint16_t* __restrict__ in, int iz, int iy, int ix,
                                                            uint16_t* __restrict__ out, int oz, int oy, int ox) {
    const long long n = (long long)oz * oy * ox;
    const double sz = (double)iz / (double)oz, sy = (double)iy / (double)oy, sx = (double)ix / (double)ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox), y = (int)((i / ox) % oy), z = (int)(i / ((long long)ox * oy));
        // align_corners=False: src = (dst + 0.5) * scale - 0.5, clamped to the edge
        const double fz = fmin(fmax(((double)z + 0.5) * sz - 0.5, 0.0), (double)(iz - 1));
        const double fy = fmin(fmax(((double)y + 0.5) * sy - 0.5, 0.0), (double)(iy - 1));
        const double fx = fmin(fmax(((double)x + 0.5) * sx - 0.5, 0.0), (double)(ix - 1));
        const double v = lerp3_u16(in, iz, iy, ix, fz, fy, fx, false);
        out[i] = (uint16_t)fmin(fmax(floor(v + 0.5), 0.0), 65535.0);
    }
}

// out[z,y,x] = trilinear sample of `in` at M.(z,y,x,1) (index space, zero outside the volume), round half up
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) affine_warp_u16_kernel(const uint16_t* __restrict__ in, int iz, int iy, int ix,
                                                              Affine34 A, uint16_t* __restrict__ out, int oz, int oy, int ox) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox), y = (int)((i / ox) % oy), z = (int)(i / ((long long)ox * oy));
        const double dz = (double)z, dy = (double)y, dx = (double)x;
        const double fz = ((A.m[0] * dz + A.m[1] * dy) + A.m[2] * dx) + A.m[3];
        const double fy = ((A.m[4] * dz + A.m[5] * dy) + A.m[6] * dx) + A.m[7];
        const double fx = ((A.m[8] * dz + A.m[9] * dy) + A.m[10] * dx) + A.m[11];
        double v = 0.0;
        if (fz > -1.0 && fz < (double)iz && fy > -1.0 && fy < (double)iy && fx > -1.0 && fx < (double)ix)
            v = lerp3_u16(in, iz, iy, ix, fz, fy, fx, true);
        out[i] = (uint16_t)fmin(fmax(floor(v + 0.5), 0.0), 65535.0);
    }
}

int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 256LL * 32); }

}  // namespace

extern "C" {

int dlv_block_mean_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int Z, int Y, int X, int fz, int fy, int fx,
                           uint16_t* out_dev) {
    if (!ctx || !in_dev || !out_dev) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0 || fz <= 0 || fy <= 0 || fx <= 0) return dlv_fail(ctx, DLV_EINVAL, "bad shape/factors");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const int oz = (Z + fz - 1) / fz, oy = (Y + fy - 1) / fy, ox = (X + fx - 1) / fx;
    DlvProf p(ctx, "block_mean_u16", 0.0, 2.0 * Z * Y * X + 2.0 * oz * oy * ox);
    hipLaunchKernelGGL(block_mean_u16_kernel, dim3(grid_for((long long)oz * oy * ox)), dim3(256), 0, ctx->stream, in_dev, Z,
                       Y, X, fz, fy, fx, out_dev, oz, oy, ox);
    p.end();
    DLV_LAUNCH_CHECK(ctx, "block_mean_u16_kernel");
    return DLV_OK;
}

int dlv_zoom_spline2_u8_dev(dlv_ctx* ctx, const uint8_t* in_dev, int iz, int iy, int ix, uint8_t* out_dev, int oz, int oy,
                            int ox) {
    if (!ctx || !in_dev || !out_dev) return DLV_EINVAL;
    if (iz <= 0 || iy <= 0 || ix <= 0 || oz <= 0 || oy <= 0 || ox <= 0) return dlv_fail(ctx, DLV_EINVAL, "bad shape");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DlvProf p(ctx, "zoom_spline2_u8", 0.0, 1.0 * oz * oy * ox + 1.0 * iz * iy * ix);
    hipLaunchKernelGGL(zoom_spline2_u8_kernel, dim3(grid_for((long long)oz * oy * ox)), dim3(256), 0, ctx->stream, in_dev,
                       iz, iy, ix, out_dev, oz, oy, ox);
    p.end();
    DLV_LAUNCH_CHECK(ctx, "zoom_spline2_u8_kernel");
    return DLV_OK;
}

int dlv_mask_pad_u16_dev(dlv_ctx* ctx, const uint16_t* raw_dev, const uint8_t* mask_dev, int threshold, int Z, int Y, int X,
                         uint16_t* out_dev, int Zp, int Yp, int Xp) {
    if (!ctx || !raw_dev || !out_dev) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0 || Zp < Z || Yp < Y || Xp < X) return dlv_fail(ctx, DLV_EINVAL, "bad shapes");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DlvProf p(ctx, "mask_pad_u16", 0.0, 3.0 * Z * Y * X + 2.0 * Zp * Yp * Xp);
    hipLaunchKernelGGL(mask_pad_u16_kernel, dim3(grid_for((long long)Zp * Yp * Xp)), dim3(256), 0, ctx->stream, raw_dev,
                       mask_dev, threshold, Z, Y, X, out_dev, Zp, Yp, Xp);
    p.end();
    DLV_LAUNCH_CHECK(ctx, "mask_pad_u16_kernel");
    return DLV_OK;
}

int dlv_trilinear_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int iz, int iy, int ix, uint16_t* out_dev, int oz, int oy,
                          int ox) {
    if (!ctx || !in_dev || !out_dev) return DLV_EINVAL;
    if (iz <= 0 || iy <= 0 || ix <= 0 || oz <= 0 || oy <= 0 || ox <= 0) return dlv_fail(ctx, DLV_EINVAL, "bad shape");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DlvProf p(ctx, "trilinear_u16", 0.0, 2.0 * oz * oy * ox + 2.0 * iz * iy * ix);
    hipLaunchKernelGGL(trilinear_u16_kernel, dim3(grid_for((long long)oz * oy * ox)), dim3(256), 0, ctx->stream, in_dev, iz,
                       iy, ix, out_dev, oz, oy, ox);
    p.end();
    DLV_LAUNCH_CHECK(ctx, "trilinear_u16_kernel");
    return DLV_OK;
}

/* the north-star's "affine atlas-space warp" (BASELINE config 5): the reference itself warps cell coordinates with
 * external mBrainAligner binaries (automate_mBrainaligner.py:21-72, :292-435) and has no volume warp; this entry point
 * resamples a volume through a 3x4 affine map given in index space. */
int dlv_affine_warp_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int iz, int iy, int ix, const double* matrix34, uint16_t* out_dev,
                            int oz, int oy, int ox) {
    if (!ctx || !in_dev || !out_dev || !matrix34) return DLV_EINVAL;
    if (iz <= 0 || iy <= 0 || ix <= 0 || oz <= 0 || oy <= 0 || ox <= 0) return dlv_fail(ctx, DLV_EINVAL, "bad shape");
    Affine34 A;
    for (int k = 0; k < 12; ++k) {
        if (!(matrix34[k] == matrix34[k]) || matrix34[k] > 1e12 || matrix34[k] < -1e12) return dlv_fail(ctx, DLV_EINVAL, "matrix entry %d is not finite", k);
        A.m[k] = matrix34[k];
    }
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    DlvProf p(ctx, "affine_warp_u16", 0.0, 2.0 * oz * oy * ox + 2.0 * iz * iy * ix);
    hipLaunchKernelGGL(affine_warp_u16_kernel, dim3(grid_for((long long)oz * oy * ox)), dim3(256), 0, ctx->stream, in_dev, iz, iy,
                       ix, A, out_dev, oz, oy, ox);
    p.end();
    DLV_LAUNCH_CHECK(ctx, "affine_warp_u16_kernel");
    return DLV_OK;
}

}  // extern "C"
