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
#include <cstdlib>

#include <cmath>

#include "common.h"

// Every resampler below is bit-exact against numpy / scipy arithmetic: no a*b+c may become an fma (HIP's default for device
// code is -ffp-contract=fast, and __dmul_rn / __dadd_rn are plain operators in HIP's headers, so they do not prevent it:
// the spline-2 zoom differed from scipy at exact 1/2 ties - e.g. a zoom factor of 3.75 - until this pragma covered it).
#pragma clang fp contract(off)

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

// The same sums with coalesced reads: one workgroup per (output z, output y) strip walks the fz x fy input rows of its
// blocks; a thread owns 8 consecutive columns (one 16-byte load per row), the 8 column sums go to LDS and one thread per
// output voxel adds its fx columns.  x is processed in chunks whose width is a multiple of 8 and of fx, so that a chunk
// starts on a 16-byte boundary and on a block boundary.  Requires X % 8 == 0 and fz * fy <= 65536 (uint32 column sums);
// the kernel above serves every other shape.
template <int CH>  // chunk capacity in columns (multiple of 8), 8 columns per thread
__global__ void __launch_bounds__(CH / 8) block_mean_strip_kernel(const uint16_t* __restrict__ in, int Z, int Y, int X, int fz, int fy,
                                                                int fx, uint16_t* __restrict__ out, int oy, int ox, int chunk_w) {
    __shared__ unsigned col[CH];
    const int z = blockIdx.y, y = blockIdx.x;
    const unsigned long long div = (unsigned long long)fz * fy * fx;
    const int nb = chunk_w / fx;  // output voxels per chunk
    for (int c0 = 0; c0 < X; c0 += chunk_w) {
        const int xc = c0 + (int)threadIdx.x * 8;
        unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((int)threadIdx.x * 8 < chunk_w && xc < X) {
            for (int a = 0; a < fz; ++a) {
                const int zz = z * fz + a;
                if (zz >= Z) break;
                for (int b = 0; b < fy; ++b) {
                    const int yy = y * fy + b;
                    if (yy >= Y) break;
                    const uint4 v = *reinterpret_cast<const uint4*>(in + ((long long)zz * Y + yy) * X + xc);
                    acc[0] += v.x & 0xffffu; acc[1] += v.x >> 16;
                    acc[2] += v.y & 0xffffu; acc[3] += v.y >> 16;
                    acc[4] += v.z & 0xffffu; acc[5] += v.z >> 16;
                    acc[6] += v.w & 0xffffu; acc[7] += v.w >> 16;
                }
            }
        }
        __syncthreads();  // the previous chunk's sums have been read
        if ((int)threadIdx.x * 8 < chunk_w) {
#pragma unroll
            for (int k = 0; k < 8; ++k) col[threadIdx.x * 8 + k] = acc[k];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < nb; j += blockDim.x) {
            const int xo = c0 / fx + j;
            if (xo >= ox) break;
            unsigned long long t = 0;
            for (int c = 0; c < fx; ++c) t += col[j * fx + c];  // columns beyond X hold 0 (the reference zero-pads)
            out[((long long)z * oy + y) * ox + xo] = (uint16_t)(t / div);
        }
    }
}

struct Taps {
    int k[3];
    double w[3];
};

// `scale` = (n_in - 1) / (n_out - 1) in fp64, computed ONCE ON THE HOST and passed in: the quotient decides which side of an
// exact 1/2 tie an output lands on, so it has to be the IEEE quotient numpy computes
__device__ __forceinline__ Taps spline2_taps(int i, int n_in, double scale) {
    Taps t;
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

// the centre index alone (same arithmetic as spline2_taps; `scale` = (n_in-1)/(n_out-1) computed once by the caller - the
// fp64 division is the most expensive operation of the whole kernel)
__device__ __forceinline__ int spline2_centre(int i, double scale) {
    return (int)floor(__dadd_rn(__dmul_rn((double)i, scale), 0.5));
}
// whole-sample mirror of a tap index in [-1, n_in] (the centre lies in [0, n_in-1]): equal to spline2_taps' modulo form
__device__ __forceinline__ int mirror1(int k, int n_in) {
    if (n_in == 1) return 0;
    if (k < 0) k = -k;
    if (k >= n_in) k = 2 * (n_in - 1) - k;
    return k;
}

struct ZoomScales {
    double z, y, x;
};

__global__ void __launch_bounds__(256) zoom_spline2_u8_kernel(const uint8_t* __restrict__ in, int iz, int iy, int ix,
                                                              uint8_t* __restrict__ out, int oz, int oy, int ox, ZoomScales sc) {
    const long long n = (long long)oz * oy * ox;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox), y = (int)((i / ox) % oy), z = (int)(i / ((long long)ox * oy));
        const Taps tz = spline2_taps(z, iz, sc.z), ty = spline2_taps(y, iy, sc.y), tx = spline2_taps(x, ix, sc.x);
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

// The same values, 16 consecutive x outputs per thread.  Up-sampling a small mask by (4,15,15) makes almost every output
// voxel look at 27 equal inputs: if every input the run's taps touch holds one value c, each of its outputs is
// (uint8)(c * S + 0.5) with S = sum of the 27 weight products = 1 +- 1e-14 (the weights of an axis add up to 1 up to two
// roundings), which is c for every uint8 c - no fp64 arithmetic needed and the result is the one the full evaluation
// gives (tests: bit-exact vs scipy on masks with edges, and this kernel against the one above on random data).  Runs
// whose taps see different values take the full evaluation.  One 16-byte store per run when the row layout allows it.
__global__ void __launch_bounds__(256) zoom_spline2_u8_run16_kernel(const uint8_t* __restrict__ in, int iz, int iy, int ix,
                                                                    uint8_t* __restrict__ out, int oz, int oy, int ox, ZoomScales sc) {
    const int runs = (ox + 15) / 16;
    const long long n = (long long)oz * oy * runs;
    const bool aligned = (ox % 16 == 0) && ((reinterpret_cast<unsigned long long>(out) & 15ull) == 0);
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count: the lanes of a wave classify one run each, then share the runs that need the full evaluation
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long rounds = (n + stride - 1) / stride;
    for (long long rd = 0; rd < rounds; ++rd) {
        const long long i = first + rd * stride;
        const bool active = i < n;
        int z = 0, y = 0, x0 = 0, nx = 0;
        bool uniform = false;
        if (active) {
            const int r = (int)(i % runs);
            y = (int)((i / runs) % oy);
            z = (int)(i / ((long long)runs * oy));
            x0 = r * 16;
            nx = min(16, ox - x0);
            // input rows / columns the run touches (indices only: no weights yet)
            int kz[3], ky[3];
            const int cz = spline2_centre(z, sc.z), cy = spline2_centre(y, sc.y);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                kz[t] = mirror1(cz + t - 1, iz);
                ky[t] = mirror1(cy + t - 1, iy);
            }
            int kmin = ix, kmax = -1;
            for (int j = 0; j < nx; ++j) {
                const int ci = spline2_centre(x0 + j, sc.x);
#pragma unroll
                for (int t = -1; t <= 1; ++t) {
                    const int k = mirror1(ci + t, ix);
                    kmin = min(kmin, k);
                    kmax = max(kmax, k);
                }
            }
            uniform = kmax - kmin < 8;
            if (uniform) {
                const unsigned cval = in[((long long)kz[0] * iy + ky[0]) * ix + kmin];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const uint8_t* row = in + ((long long)kz[a] * iy + ky[b]) * ix;
                        for (int k = kmin; k <= kmax; ++k) uniform = uniform && (row[k] == cval);
                    }
                if (uniform) {
                    uint8_t* dst = out + ((long long)z * oy + y) * ox + x0;
                    if (aligned) {
                        const unsigned w = cval * 0x01010101u;
                        *reinterpret_cast<uint4*>(dst) = make_uint4(w, w, w, w);
                    } else {
                        for (int j = 0; j < nx; ++j) dst[j] = (uint8_t)cval;
                    }
                }
            }
        }
        // the other runs (mask edges): four at a time, one output voxel per lane
        unsigned long long todo = __ballot(active && !uniform);
        while (todo) {
            int srcs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                srcs[q] = todo ? __ffsll((long long)todo) - 1 : -1;
                if (todo) todo &= todo - 1;
            }
            const int slot = lane >> 4, j = lane & 15;
            const int src = slot == 0 ? srcs[0] : (slot == 1 ? srcs[1] : (slot == 2 ? srcs[2] : srcs[3]));
            const int sl = src < 0 ? 0 : src;
            const int rz = __shfl(z, sl, 64), ry = __shfl(y, sl, 64), rx0 = __shfl(x0, sl, 64), rnx = __shfl(nx, sl, 64);
            if (src >= 0 && j < rnx) {
                const Taps tz = spline2_taps(rz, iz, sc.z), ty = spline2_taps(ry, iy, sc.y), tx = spline2_taps(rx0 + j, ix, sc.x);
                double t = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const uint8_t* row = in + ((long long)tz.k[a] * iy + ty.k[b]) * ix;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const double val = (double)row[tx.k[c]];
                            t = t + ((val * tz.w[a]) * ty.w[b]) * tx.w[c];
                        }
                    }
                double rr = t > 0.0 ? t + 0.5 : 0.0;  // scipy CASE_INTERP_OUT_UINT
                rr = rr > 255.0 ? 255.0 : rr;
                out[((long long)rz * oy + ry) * ox + rx0 + j] = (uint8_t)rr;
            }
        }
    }
}

// The same values once more, organised by OUTPUT ROWS: one workgroup per (z, group of ZR_ROWS consecutive y).  What the run16
// kernel re-derives per run - the tap range of the run along x (16 fp64 centre evaluations) and the comparison of up to
// 9 x 8 input bytes - is hoisted: the tap ranges depend on x only (zoom_run_table_kernel, once per call), and whether the
// nine input rows (3 kz x 3 ky) agree is decided per INPUT COLUMN once per distinct (cz, cy) of the group (`colval`: the
// common value, or 0x100).  A run is uniform when colval is one valid value over its tap range: a few LDS reads and one
// 16-byte store; the runs of all rows that share a centre row are spread over the 256 threads.  Runs at the mask's edge
// are queued and evaluated afterwards, one output voxel per lane, with the arithmetic of the kernels above (bit-exact vs
// scipy), screened by an fp32 evaluation that decides every voxel whose value is not within 4e-3 of a rounding tie.
constexpr int ZR_ROWS = 32;
struct Taps32 {
    int k[3];
    float w[3];
};
__device__ __forceinline__ Taps32 spline2_taps32(int i, int n_in, double scale);
// fp32 screening taps of every output x and y (spline2_taps32 evaluated once per call instead of once per edge voxel: two fp64
// centre computations and six mirrored indices per voxel were a third of the edge loop): .k = k0 | k1 << 16, k2; .w = weights
struct TapsRow {
    unsigned k01, k2;
    float w0, w1, w2, pad;
};
__global__ void __launch_bounds__(256) zoom_run_table_kernel(short2* __restrict__ runk, int ix, int ox, double scx, TapsRow* __restrict__ xt,
                                                             TapsRow* __restrict__ yt, int iy, int oy, double scy) {
    const int runs = (ox + 15) / 16;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < runs; r += gridDim.x * blockDim.x) {
        const int x0 = r * 16, nx = min(16, ox - x0);
        int kmin = ix, kmax = -1;
        for (int j = 0; j < nx; ++j) {
            const int ci = spline2_centre(x0 + j, scx);
#pragma unroll
            for (int t = -1; t <= 1; ++t) {
                const int k = mirror1(ci + t, ix);
                kmin = min(kmin, k);
                kmax = max(kmax, k);
            }
        }
        runk[r] = make_short2((short)kmin, (short)kmax);
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ox + oy; i += gridDim.x * blockDim.x) {
        const bool isx = i < ox;
        const Taps32 t = isx ? spline2_taps32(i, ix, scx) : spline2_taps32(i - ox, iy, scy);
        TapsRow e;
        e.k01 = (unsigned)t.k[0] | ((unsigned)t.k[1] << 16);
        e.k2 = (unsigned)t.k[2];
        e.w0 = t.w[0];
        e.w1 = t.w[1];
        e.w2 = t.w[2];
        e.pad = 0.f;
        (isx ? xt[i] : yt[i - ox]) = e;
    }
}

// weights and (mirrored) tap indices of output index i: spline2_taps with the centre / mirror helpers (same values)
__device__ __forceinline__ Taps spline2_taps_m(int i, int n_in, double scale) {
    Taps t;
    const double x = __dmul_rn((double)i, scale);
    const double c = floor(__dadd_rn(x, 0.5));
    const double d = __dsub_rn(x, c);
    const double w1 = __dsub_rn(0.75, __dmul_rn(d, d));
    const double y = __dsub_rn(0.5, d);
    const double w0 = __dmul_rn(__dmul_rn(0.5, y), y);
    t.w[0] = w0;
    t.w[1] = w1;
    t.w[2] = __dsub_rn(__dsub_rn(1.0, w0), w1);
    const int ci = (int)c;
#pragma unroll
    for (int j = 0; j < 3; ++j) t.k[j] = mirror1(ci + j - 1, n_in);
    return t;
}

// the fp32 screening form: centre from the fp64 coordinate (the same centre as the exact path), weights in fp32
__device__ __forceinline__ Taps32 spline2_taps32(int i, int n_in, double scale) {
    Taps32 t;
    const double x = __dmul_rn((double)i, scale);
    const double c = floor(__dadd_rn(x, 0.5));
    const float d = (float)__dsub_rn(x, c);
    const float y = 0.5f - d, u = 0.5f + d;
    t.w[0] = 0.5f * y * y;
    t.w[1] = 0.75f - d * d;
    t.w[2] = 0.5f * u * u;
    const int ci = (int)c;
#pragma unroll
    for (int j = 0; j < 3; ++j) t.k[j] = mirror1(ci + j - 1, n_in);
    return t;
}

__global__ void __launch_bounds__(256) zoom_spline2_u8_rows_kernel(const uint8_t* __restrict__ in, int iz, int iy, int ix,
                                                                   uint8_t* __restrict__ out, int oz, int oy, int ox, ZoomScales sc,
                                                                   int aligned, const short2* __restrict__ runk, int nrows_max,
                                                                   const TapsRow* __restrict__ xt, const TapsRow* __restrict__ yt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zr_smem[];
    const int runs = (ox + 15) / 16;
    unsigned short* colval = reinterpret_cast<unsigned short*>(zr_smem);                                       // [ix]
    int* queue = reinterpret_cast<int*>(zr_smem + (((size_t)ix * 2 + 15) & ~(size_t)15));                      // [ZR_ROWS * runs]
    short2* runl = reinterpret_cast<short2*>(queue + (size_t)ZR_ROWS * runs);    // [runs]: the tap ranges, from the table
    uint8_t* tile = reinterpret_cast<uint8_t*>(runl + runs);  // [3 kz][nrows][ix]: the input rows the group's taps touch
    __shared__ int qn;
    const int ygroups = (oy + ZR_ROWS - 1) / ZR_ROWS;
    const int z = blockIdx.x / ygroups, y0 = (blockIdx.x % ygroups) * ZR_ROWS;
    const int y1 = min(y0 + ZR_ROWS, oy);
    const int tid = threadIdx.x;
    if (tid == 0) qn = 0;
    int kz[3];
    const int cz = spline2_centre(z, sc.z);
#pragma unroll
    for (int t = 0; t < 3; ++t) kz[t] = mirror1(cz + t - 1, iz);
    // the input rows (unmirrored index cy0 - 1 .. cyN + 1, three kz planes) the taps of this group's rows can touch: the
    // edge voxels read their 27 inputs from LDS (27 scattered byte loads per voxel from global memory were the bottleneck)
    const int cy0 = spline2_centre(y0, sc.y) - 1;
    const int nrows = min(spline2_centre(y1 - 1, sc.y) + 1 - cy0 + 1, nrows_max);
    for (int r = tid; r < runs; r += 256) runl[r] = runk[r];
    for (int row = 0; row < 3 * nrows; ++row) {
        const int a = row / nrows, rr = row - a * nrows;  // (workgroup-uniform)
        const uint8_t* src = in + ((long long)kz[a] * iy + mirror1(cy0 + rr, iy)) * ix;
        for (int k = tid; k < ix; k += 256) tile[(size_t)row * ix + k] = src[k];
    }
    for (int ya = y0; ya < y1;) {  // segments of rows with one centre row cy (workgroup-uniform)
        const int cy = spline2_centre(ya, sc.y);
        int yb = ya + 1;
        while (yb < y1 && spline2_centre(yb, sc.y) == cy) ++yb;
        const int lr = cy - 1 - cy0;  // local row of ky tap 0
        __syncthreads();  // the tile is staged / the previous segment's readers are done with colval
        for (int k = tid; k < ix; k += 256) {
            const unsigned v0 = tile[(size_t)lr * ix + k];
            bool same = true;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) same = same && (tile[((size_t)a * nrows + lr + b) * ix + k] == v0);
            colval[k] = (unsigned short)(same ? v0 : 0x100u);
        }
        __syncthreads();
        const int nitem = (yb - ya) * runs;
        for (int it = tid; it < nitem; it += 256) {
            const int yy = it / runs, r = it - yy * runs;
            const short2 kk = runl[r];
            const unsigned c = colval[kk.x];
            bool uniform = c < 0x100u;
            for (int k = kk.x + 1; k <= kk.y; ++k) uniform = uniform && (colval[k] == c);
            if (uniform) {
                uint8_t* const dst = out + ((long long)z * oy + ya + yy) * ox + r * 16;
                if (aligned) {
                    const unsigned w = c * 0x01010101u;
                    *reinterpret_cast<uint4*>(dst) = make_uint4(w, w, w, w);
                } else {
                    const int nx = min(16, ox - r * 16);
                    for (int j = 0; j < nx; ++j) dst[j] = (uint8_t)c;
                }
            } else {
                queue[atomicAdd(&qn, 1)] = ((ya + yy - y0) << 16) | r;
            }
        }
        ya = yb;
    }
    __syncthreads();
    // the queued runs (mask edges): 16 at a time, one output voxel per lane
    const int nq = qn;
    const int slot = tid >> 4, j = tid & 15;
    const Taps32 fz = spline2_taps32(z, iz, sc.z);
    for (int q0 = 0; q0 < nq; q0 += 16) {
        const int qi = q0 + slot;
        if (qi >= nq) continue;
        const int e = queue[qi];
        const int y = y0 + (e >> 16), x = (e & 0xffff) * 16 + j;
        if (x >= ox) continue;
        // fp32 first: the same centres (fp64, exact), weights and sum in fp32 (error < 2e-3 for uint8 inputs).  Unless
        // t + 1/2 lands within 4e-3 of an integer the truncation is decided - identical to the fp64 evaluation; the few
        // voxels near a tie (about one per crossing of the mask's edge) take the exact path below.
        const TapsRow ex = xt[x], ey = yt[y];  // (the values spline2_taps32 gives: tabulated once per call)
        Taps32 fx, fy;
        fx.k[0] = (int)(ex.k01 & 0xffffu); fx.k[1] = (int)(ex.k01 >> 16); fx.k[2] = (int)ex.k2;
        fx.w[0] = ex.w0; fx.w[1] = ex.w1; fx.w[2] = ex.w2;
        fy.w[0] = ey.w0; fy.w[1] = ey.w1; fy.w[2] = ey.w2;
        const int lrow = spline2_centre(y, sc.y) - 1 - cy0;  // local row of tap b = 0
        float tf = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const uint8_t* rowp = tile + ((size_t)a * nrows + lrow + b) * ix;
                const float p = fz.w[a] * fy.w[b];
                tf = fmaf((float)rowp[fx.k[0]] * p, fx.w[0], tf);
                tf = fmaf((float)rowp[fx.k[1]] * p, fx.w[1], tf);
                tf = fmaf((float)rowp[fx.k[2]] * p, fx.w[2], tf);
            }
        const float sf = tf + 0.5f, rf = floorf(sf), fr = sf - rf;
        if (fr > 4e-3f && fr < 1.f - 4e-3f) {
            out[((long long)z * oy + y) * ox + x] = (uint8_t)fminf(fmaxf(rf, 0.f), 255.f);
            continue;
        }
        const Taps tz = spline2_taps_m(z, iz, sc.z), ty = spline2_taps_m(y, iy, sc.y), tx = spline2_taps_m(x, ix, sc.x);
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const uint8_t* rowp = tile + ((size_t)a * nrows + lrow + b) * ix;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double val = (double)rowp[tx.k[c]];
                    t = t + ((val * tz.w[a]) * ty.w[b]) * tx.w[c];
                }
            }
        double rr = t > 0.0 ? t + 0.5 : 0.0;  // scipy CASE_INTERP_OUT_UINT
        rr = rr > 255.0 ? 255.0 : rr;
        out[((long long)z * oy + y) * ox + x] = (uint8_t)rr;
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
    // coalesced strip kernel where the layout allows it (16-byte loads need X % 8 == 0 and an aligned base; uint32 column
    // sums need fz * fy <= 65536; a chunk of <= 2048 columns must hold whole blocks of fx columns on 8-column boundaries)
    long long l = fx;
    while (l % 8) l += fx;  // lcm(fx, 8)
    const bool strip = X % 8 == 0 && (reinterpret_cast<unsigned long long>(in_dev) & 15ull) == 0 && (long long)fz * fy <= 65536 &&
                       l <= 2048 && oz <= 65535 && !ctx->resample_simple;
    if (strip) {
        const int chunk_w = (int)((2048 / l) * l);
        hipLaunchKernelGGL((block_mean_strip_kernel<2048>), dim3(oy, oz), dim3(256), 0, ctx->stream, in_dev, Z, Y, X, fz, fy, fx,
                           out_dev, oy, ox, chunk_w);
    } else {
        hipLaunchKernelGGL(block_mean_u16_kernel, dim3(grid_for((long long)oz * oy * ox)), dim3(256), 0, ctx->stream, in_dev, Z,
                           Y, X, fz, fy, fx, out_dev, oz, oy, ox);
    }
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
    ZoomScales sc;
    sc.z = oz > 1 ? (double)(iz - 1) / (double)(oz - 1) : 0.0;
    sc.y = oy > 1 ? (double)(iy - 1) / (double)(oy - 1) : 0.0;
    sc.x = ox > 1 ? (double)(ix - 1) / (double)(ox - 1) : 0.0;
    if (ctx->resample_simple)  // the one-voxel-per-thread kernel (A/B and cross-check in tests: dlv_diag_set)
        hipLaunchKernelGGL(zoom_spline2_u8_kernel, dim3(grid_for((long long)oz * oy * ox)), dim3(256), 0, ctx->stream, in_dev,
                           iz, iy, ix, out_dev, oz, oy, ox, sc);
    else {
        // row-organised kernel whenever its tables fit LDS and 16-bit indices (always, for the pipeline's masks); the
        // run-per-thread kernel otherwise (dlv_diag_set "resample_run16": A/B and cross-check in tests)
        const int runs = (ox + 15) / 16;
        const int nrows_max = (int)std::ceil((ZR_ROWS - 1) * sc.y) + 4;  // centre rows of a group's first and last output row, +-1, +1
        const size_t lds = (((size_t)ix * 2 + 15) & ~(size_t)15) + (size_t)ZR_ROWS * runs * 4 + (size_t)runs * 4 + (size_t)3 * nrows_max * ix;
        const long long groups = (long long)oz * ((oy + ZR_ROWS - 1) / ZR_ROWS);
        if (ix <= 32767 && runs <= 32767 && lds <= 60 * 1024 && groups < (1ll << 31) && !ctx->resample_run16) {
            const int aligned = (ox % 16 == 0) && ((reinterpret_cast<unsigned long long>(out_dev) & 15ull) == 0);
            char* wsp;
            const size_t runk_b = ((size_t)runs * sizeof(short2) + 31) & ~(size_t)31;
            DLV_TRY(dlv_ws_get(ctx, WS_MISC, runk_b + (size_t)(ox + oy) * sizeof(TapsRow), (void**)&wsp));
            short2* runk = reinterpret_cast<short2*>(wsp);
            TapsRow* xt = reinterpret_cast<TapsRow*>(wsp + runk_b);
            TapsRow* yt = xt + ox;
            hipLaunchKernelGGL(zoom_run_table_kernel, dim3((std::max(runs, ox + oy) + 255) / 256), dim3(256), 0, ctx->stream, runk, ix, ox, sc.x,
                               xt, yt, iy, oy, sc.y);
            hipLaunchKernelGGL(zoom_spline2_u8_rows_kernel, dim3((unsigned)groups), dim3(256), lds, ctx->stream, in_dev, iz, iy, ix,
                               out_dev, oz, oy, ox, sc, aligned, runk, nrows_max, xt, yt);
        } else {
            hipLaunchKernelGGL(zoom_spline2_u8_run16_kernel, dim3(grid_for((long long)oz * oy * ((ox + 15) / 16))), dim3(256), 0,
                               ctx->stream, in_dev, iz, iy, ix, out_dev, oz, oy, ox, sc);
        }
    }
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
