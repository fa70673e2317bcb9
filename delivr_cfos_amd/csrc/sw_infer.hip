// sw_infer.hip - the sliding-window pass with the volume resident in HBM.
//
// Restates inference/sliding_window_inferer.py:102-251 (sliding_window_inference):
//   tiler      :140-145 (+ _get_scan_interval :255-276, MONAI dense_patch_slices)
//   gather/cast:181-195    uint16 -> int32 -> fp32, (B,1,d,h,w)
//   skip       :198-202    max(window) <= threshold -> logits := -1000 (decided per window here,
//                          i.e. the reference at sw_batch_size = 1)
//   flip TTA   :218-226
//   blend      :232-251    acc += 1 * logit, cnt += 1   (fp32 here, fp16 in the reference)
//
// Windows are processed colour class by colour class: within a class no two windows overlap, so
// the read-modify-write of the accumulator needs no atomics and every voxel receives its
// contributions in a fixed order (bitwise reproducible, independent of batch size).
#include <algorithm>

#include "common.h"
#include <cmath>

namespace {

struct Tiler {
    int n[3], roi[3], iv[3];
    std::vector<int> st[3];
    int ncol[3];
    std::vector<int> col[3];
    int64_t count() const { return (int64_t)st[0].size() * st[1].size() * st[2].size(); }
};

int build_tiler(dlv_ctx* ctx, const dlv_sw_params* p, Tiler& t) {
    t.n[0] = p->Zp;
    t.n[1] = p->Yp;
    t.n[2] = p->Xp;
    if (!(p->overlap >= 0.f && p->overlap < 1.f)) return dlv_fail(ctx, DLV_EINVAL, "overlap must be >= 0 and < 1");
    for (int k = 0; k < 3; ++k) {
        if (t.n[k] <= 0) return dlv_fail(ctx, DLV_EINVAL, "empty volume");
        int r = p->roi[k] > 0 ? p->roi[k] : t.n[k];  // fall_back_tuple
        if (r > t.n[k])
            return dlv_fail(ctx, DLV_EUNSUP, "roi[%d]=%d exceeds the padded volume (%d): pad the volume first "
                            "(downsample_and_mask.py:390-396 does)", k, r, t.n[k]);
        t.roi[k] = r;
        // _get_scan_interval: python int(r * (1 - overlap)) evaluated in double
        int iv = r == t.n[k] ? r : (int)((double)r * (1.0 - (double)p->overlap));
        if (iv <= 0) iv = 1;
        t.iv[k] = iv;
        // dense_patch_slices
        const int num = (t.n[k] + iv - 1) / iv;
        int scan = num - 1;
        for (int dd = 0; dd < num; ++dd)
            if ((long long)dd * iv + r >= t.n[k]) {
                scan = dd;
                break;
            }
        t.st[k].clear();
        for (int i = 0; i <= scan; ++i) {
            int s = i * iv;
            s -= std::max(s + r - t.n[k], 0);
            t.st[k].push_back(s);
        }
        // colours: windows i and i + c are disjoint when c*iv >= roi; a clamped last window gets its own
        const int c = (r + iv - 1) / iv;
        const int kk = (int)t.st[k].size();
        t.col[k].assign(kk, 0);
        int ncol = std::min(c, kk);
        for (int i = 0; i < kk; ++i) t.col[k][i] = i % c;
        if (kk > 1 && t.st[k][kk - 1] != (kk - 1) * iv && kk > c) {
            t.col[k][kk - 1] = c;
            ncol = c + 1;
        }
        t.ncol[k] = ncol;
    }
    return DLV_OK;
}

// ---- kernels --------------------------------------------------------------------------------------

// per-window maximum of the uint16 volume.  grid (chunks, windows); one atomicMax per block.
__global__ void __launch_bounds__(256) window_max_kernel(const uint16_t* __restrict__ vol, int Yp, int Xp,
                                                         const int* __restrict__ starts, int d, int h, int w,
                                                         int* __restrict__ wmax) {
    const int win = blockIdx.y;
    const int z0 = starts[3 * win], y0 = starts[3 * win + 1], x0 = starts[3 * win + 2];
    const uint16_t* base = vol + ((long long)z0 * Yp + y0) * Xp + x0;
    unsigned m = 0;
    const bool vec = (w % 8 == 0) && (Xp % 8 == 0) && (x0 % 8 == 0);
    if (vec) {
        const int w8 = w / 8;
        const long long n8 = (long long)d * h * w8;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
            const int xx = (int)(i % w8), yy = (int)((i / w8) % h), zz = (int)(i / ((long long)w8 * h));
            const uint4 v = *reinterpret_cast<const uint4*>(base + ((long long)zz * Yp + yy) * Xp + xx * 8);
            const unsigned a = max(max(v.x & 0xffffu, v.x >> 16), max(v.y & 0xffffu, v.y >> 16));
            const unsigned b = max(max(v.z & 0xffffu, v.z >> 16), max(v.w & 0xffffu, v.w >> 16));
            m = max(m, max(a, b));
        }
    } else {
        const long long n = (long long)d * h * w;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const int xx = (int)(i % w), yy = (int)((i / w) % h), zz = (int)(i / ((long long)w * h));
            m = max(m, (unsigned)base[((long long)zz * Yp + yy) * Xp + xx]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(wmax + win, (int)m);
}

// The same maxima with ONE pass over the volume (overlapping windows read every voxel up to 8 times above): the window
// starts and ends cut every axis into cells; cell maxima first (a workgroup owns one (plane, y-cell) strip: 16-byte loads
// along x, per-thread maxima over the strip's rows, LDS atomics per x-cell, then one global atomicMax per x-cell), then
// every window takes the maximum over the cells it covers.  xcell[x / 8] is the x-cell of the 8-voxel chunk at x (all
// x-boundaries are multiples of 8 on this path), zcell[z] the z-cell of plane z or -1 outside every window.
__global__ void __launch_bounds__(256) cell_max_kernel(const uint16_t* __restrict__ vol, int Yp, int Xp, int z_lo,
                                                       const int* __restrict__ zcell, const int* __restrict__ ybound, int ncy,
                                                       const int* __restrict__ xcell, int x_lo8, int x_hi8, int ncx,
                                                       int* __restrict__ cellmax) {
    extern __shared__ int lmax[];  // ncx
    const int cy = blockIdx.x % ncy, z = z_lo + blockIdx.x / ncy;
    const int cz = zcell[z];
    if (cz < 0) return;
    for (int i = threadIdx.x; i < ncx; i += blockDim.x) lmax[i] = 0;
    __syncthreads();
    const int y0 = ybound[cy], y1 = ybound[cy + 1];
    for (int c = x_lo8 + threadIdx.x; c < x_hi8; c += blockDim.x) {
        unsigned m = 0;
        const uint16_t* col = vol + ((long long)z * Yp + y0) * Xp + 8 * c;
        for (int y = y0; y < y1; ++y, col += Xp) {
            const uint4 v = *reinterpret_cast<const uint4*>(col);
            const unsigned a = max(max(v.x & 0xffffu, v.x >> 16), max(v.y & 0xffffu, v.y >> 16));
            const unsigned b = max(max(v.z & 0xffffu, v.z >> 16), max(v.w & 0xffffu, v.w >> 16));
            m = max(m, max(a, b));
        }
        if (m) atomicMax(&lmax[xcell[c]], (int)m);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncx; i += blockDim.x)
        if (lmax[i]) atomicMax(&cellmax[((long long)cz * ncy + cy) * ncx + i], lmax[i]);
}

// ranges: per window 6 ints (cell ranges [lo, hi) along z, y, x)
__global__ void __launch_bounds__(256) window_from_cells_kernel(const int* __restrict__ ranges, long long n, int ncy, int ncx,
                                                                const int* __restrict__ cellmax, int* __restrict__ wmax) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int* r = ranges + 6 * i;
    int m = 0;
    for (int a = r[0]; a < r[1]; ++a)
        for (int b = r[2]; b < r[3]; ++b)
            for (int c = r[4]; c < r[5]; ++c) m = max(m, cellmax[((long long)a * ncy + b) * ncx + c]);
    wmax[i] = m;
}

// (B,1,d,h,w) fp32 <- uint16 volume windows, flipped along flip_dim (2 = Z, 3 = Y, 4 = X) if >= 0
__global__ void __launch_bounds__(256) gather_f32_kernel(const uint16_t* __restrict__ vol, int Yp, int Xp,
                                                         const int* __restrict__ starts, int d, int h, int w,
                                                         int flip_dim, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int z0 = starts[3 * b], y0 = starts[3 * b + 1], x0 = starts[3 * b + 2];
    const long long n = (long long)d * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int xx = (int)(i % w), yy = (int)((i / w) % h), zz = (int)(i / ((long long)w * h));
        if (flip_dim == 2) zz = d - 1 - zz;
        if (flip_dim == 3) yy = h - 1 - yy;
        if (flip_dim == 4) xx = w - 1 - xx;
        out[(long long)b * n + i] = (float)vol[((long long)(z0 + zz) * Yp + (y0 + yy)) * Xp + (x0 + xx)];
    }
}

// acc[window] += logits (un-flipped); cnt[window] += 1.  Windows of one launch never overlap.
__global__ void __launch_bounds__(256) blend_add_kernel(const float* __restrict__ logits, const int* __restrict__ starts,
                                                        int d, int h, int w, int flip_dim, int Yp, int Xp,
                                                        float scale, int rep, float* __restrict__ acc,
                                                        uint8_t* __restrict__ cnt, const float* __restrict__ bw, float bmin,
                                                        float* __restrict__ wsum) {
    const int b = blockIdx.y;
    const int z0 = starts[3 * b], y0 = starts[3 * b + 1], x0 = starts[3 * b + 2];
    const long long n = (long long)d * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w), yy = (int)((i / w) % h), zz = (int)(i / ((long long)w * h));
        int fz = zz, fy = yy, fx = xx;
        if (flip_dim == 2) fz = d - 1 - zz;
        if (flip_dim == 3) fy = h - 1 - yy;
        if (flip_dim == 4) fx = w - 1 - xx;
        const long long o = ((long long)(z0 + zz) * Yp + (y0 + yy)) * Xp + (x0 + xx);
        const float wgt = bw ? fmaxf(bw[zz] * bw[d + yy] * bw[d + h + xx], bmin) * scale : scale;
        acc[o] += wgt * logits[(long long)b * n + ((long long)fz * h + fy) * w + fx];
        if (cnt) cnt[o] += (uint8_t)rep;
        if (wsum) wsum[o] += wgt;
    }
}

// background-skipped windows: acc += value (-1000), cnt += 1
__global__ void __launch_bounds__(256) fill_add_kernel(const int* __restrict__ starts, int d, int h, int w, int Yp,
                                                       int Xp, float value, int rep, float* __restrict__ acc,
                                                       uint8_t* __restrict__ cnt, const float* __restrict__ bw, float bmin,
                                                       float* __restrict__ wsum) {
    const int b = blockIdx.y;
    const int z0 = starts[3 * b], y0 = starts[3 * b + 1], x0 = starts[3 * b + 2];
    const long long n = (long long)d * h * w;
    // the plain case (constant weights, no count map: 8 B of read-modify-write per window voxel, 5400 windows per C3 pass): four
    // voxels per thread as one 16-byte non-temporal word, 32-bit row arithmetic
    if (!cnt && !bw && !wsum && value != 0.f && w % 4 == 0 && Xp % 4 == 0 && x0 % 4 == 0 && (reinterpret_cast<uintptr_t>(acc) & 15) == 0 &&
        n < (1ll << 31)) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const unsigned w4 = (unsigned)w / 4u, n4 = (unsigned)(n / 4);
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
            const unsigned xq = i % w4, r = i / w4, yy = r % (unsigned)h, zz = r / (unsigned)h;
            f4_t* p = reinterpret_cast<f4_t*>(acc + ((long long)(z0 + (int)zz) * Yp + (y0 + (int)yy)) * Xp + x0 + 4 * (int)xq);
            f4_t v = __builtin_nontemporal_load(p);
            v += value;
            __builtin_nontemporal_store(v, p);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w), yy = (int)((i / w) % h), zz = (int)(i / ((long long)w * h));
        const long long o = ((long long)(z0 + zz) * Yp + (y0 + yy)) * Xp + (x0 + xx);
        const float wgt = bw ? fmaxf(bw[zz] * bw[d + yy] * bw[d + h + xx], bmin) : 1.f;
        if (value != 0.f) acc[o] += value * wgt;
        if (cnt) cnt[o] += (uint8_t)rep;
        if (wsum && value != 0.f) wsum[o] += wgt * (float)rep;
    }
}

}  // namespace

// maxima of n windows (starts: 3 ints per window, z relative to vol_dev's first plane) into wmax_host; synchronises
static int window_maxima(dlv_ctx* ctx, const uint16_t* vol_dev, int Yp, int Xp, const int roi[3], const std::vector<int>& starts,
                         int64_t n, int32_t* wmax) {
    // ---- one pass over the volume: cell maxima, then windows from cells --------------------------------------------
    {
        std::vector<int> bz, by, bx;  // cell boundaries: the starts and ends of the windows of this call, per axis
        for (int64_t i = 0; i < n; ++i) {
            bz.push_back(starts[3 * i]);     bz.push_back(starts[3 * i] + roi[0]);
            by.push_back(starts[3 * i + 1]); by.push_back(starts[3 * i + 1] + roi[1]);
            bx.push_back(starts[3 * i + 2]); bx.push_back(starts[3 * i + 2] + roi[2]);
        }
        for (auto* b : {&bz, &by, &bx}) {
            std::sort(b->begin(), b->end());
            b->erase(std::unique(b->begin(), b->end()), b->end());
        }
        bool fast = (Xp % 8 == 0) && ((reinterpret_cast<uintptr_t>(vol_dev) & 15) == 0);
        for (int v : bx) fast = fast && (v % 8 == 0);
        const long long ncz = (long long)bz.size() - 1, ncy = (long long)by.size() - 1, ncx = (long long)bx.size() - 1;
        fast = fast && ncz * ncy * ncx < (1ll << 24) && ncx <= 8192;
        if (fast) {
            const int z_lo = bz.front(), z_hi = bz.back();
            std::vector<int> zcell((size_t)z_hi, -1), xcell((size_t)(bx.back() / 8), 0), ranges((size_t)n * 6);
            for (size_t c = 0; c + 1 < bz.size(); ++c)
                for (int z = bz[c]; z < bz[c + 1]; ++z) zcell[z] = (int)c;
            for (size_t c = 0; c + 1 < bx.size(); ++c)
                for (int x8 = bx[c] / 8; x8 < bx[c + 1] / 8; ++x8) xcell[x8] = (int)c;
            auto cell_of = [](const std::vector<int>& b, int v) { return (int)(std::lower_bound(b.begin(), b.end(), v) - b.begin()); };
            for (int64_t i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) {
                    const std::vector<int>& b = k == 0 ? bz : (k == 1 ? by : bx);
                    ranges[6 * i + 2 * k] = cell_of(b, starts[3 * i + k]);
                    ranges[6 * i + 2 * k + 1] = cell_of(b, starts[3 * i + k] + roi[k]);
                }
            // [zcell | ybound | xcell | ranges | cellmax | wmax]
            const size_t o_y = zcell.size(), o_x = o_y + by.size(), o_r = o_x + xcell.size(), o_c = o_r + ranges.size(),
                         o_w = o_c + (size_t)(ncz * ncy * ncx), total_ints = o_w + (size_t)n;
            int* ws;
            DLV_TRY(dlv_ws_get(ctx, WS_MISC, total_ints * sizeof(int), (void**)&ws));
            std::vector<int> host(o_c);
            std::copy(zcell.begin(), zcell.end(), host.begin());
            std::copy(by.begin(), by.end(), host.begin() + o_y);
            std::copy(xcell.begin(), xcell.end(), host.begin() + o_x);
            std::copy(ranges.begin(), ranges.end(), host.begin() + o_r);
            DLV_HIP(ctx, hipMemcpyAsync(ws, host.data(), o_c * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            DLV_HIP(ctx, hipMemsetAsync(ws + o_c, 0, (size_t)(ncz * ncy * ncx) * sizeof(int), ctx->stream));
            {
                DlvProf pr(ctx, "window_max_u16", 0.0, 2.0 * (double)(z_hi - z_lo) * (by.back() - by.front()) * (bx.back() - bx.front()));
                hipLaunchKernelGGL(cell_max_kernel, dim3((unsigned)((long long)(z_hi - z_lo) * ncy)), dim3(256), (size_t)ncx * sizeof(int),
                                   ctx->stream, vol_dev, Yp, Xp, z_lo, ws, ws + o_y, (int)ncy, ws + o_x, bx.front() / 8,
                                   bx.back() / 8, (int)ncx, ws + o_c);
                hipLaunchKernelGGL(window_from_cells_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ws + o_r,
                                   (long long)n, (int)ncy, (int)ncx, ws + o_c, ws + o_w);
                pr.end();
            }
            DLV_LAUNCH_CHECK(ctx, "cell_max_kernel / window_from_cells_kernel");
            DLV_HIP(ctx, hipMemcpyAsync(wmax, ws + o_w, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (also: the host tables above are consumed)
            return DLV_OK;
        }
    }
    // general case (x boundaries off the 8-voxel grid, unaligned volume): every window reads its own voxels
    int* meta;
    DLV_TRY(dlv_ws_get(ctx, WS_MISC, (size_t)n * 4 * sizeof(int), (void**)&meta));
    DLV_HIP(ctx, hipMemcpyAsync(meta, starts.data(), (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    DLV_HIP(ctx, hipMemsetAsync(meta + n * 3, 0, (size_t)n * sizeof(int), ctx->stream));
    const long long tile_vox = (long long)roi[0] * roi[1] * roi[2];
    const int chunks = (int)std::min<long long>(std::max<long long>(tile_vox / (256 * 8 * 4), 1), 64);
    {
        DlvProf pr(ctx, "window_max_u16", 0.0, 2.0 * tile_vox * n);
        hipLaunchKernelGGL(window_max_kernel, dim3(chunks, (unsigned)n), dim3(256), 0, ctx->stream, vol_dev, Yp, Xp, meta,
                           roi[0], roi[1], roi[2], meta + n * 3);
        pr.end();
    }
    DLV_LAUNCH_CHECK(ctx, "window_max_kernel");
    DLV_HIP(ctx, hipMemcpyAsync(wmax, meta + n * 3, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DLV_OK;
}

extern "C" {

int dlv_sw_num_windows(const dlv_sw_params* p, int64_t* n_windows) {
    if (!p || !n_windows) return DLV_EINVAL;
    Tiler t;
    DLV_TRY(build_tiler(nullptr, p, t));
    *n_windows = t.count();
    return DLV_OK;
}

int dlv_sw_window_starts(const dlv_sw_params* p, int64_t* starts, int64_t capacity) {
    if (!p || !starts) return DLV_EINVAL;
    Tiler t;
    DLV_TRY(build_tiler(nullptr, p, t));
    if (capacity < t.count()) return DLV_EINVAL;
    int64_t i = 0;
    for (int z : t.st[0])
        for (int y : t.st[1])
            for (int x : t.st[2]) {
                starts[3 * i] = z;
                starts[3 * i + 1] = y;
                starts[3 * i + 2] = x;
                ++i;
            }
    return DLV_OK;
}

int dlv_sw_window_max_dev(dlv_ctx* ctx, const dlv_sw_params* p, const uint16_t* vol_dev, int32_t* wmax, int64_t capacity) {
    if (!ctx || !p || !vol_dev || !wmax) return DLV_EINVAL;
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    Tiler t;
    DLV_TRY(build_tiler(ctx, p, t));
    const int64_t total = t.count();
    // the shard of the window list (win_begin/win_end) and the slab of the volume (z0/nz) are honoured: a rank of a
    // sharded run computes the maxima of its own windows from the planes it holds; wmax[i] belongs to window win_begin + i
    const int64_t wb = std::max<int64_t>(p->win_begin, 0);
    const int64_t we = p->win_end > 0 ? std::min<int64_t>(p->win_end, total) : total;
    const int64_t n = std::max<int64_t>(we - wb, 0);
    if (capacity < n) return dlv_fail(ctx, DLV_EINVAL, "wmax capacity %lld < %lld windows", (long long)capacity, (long long)n);
    if (n == 0) return DLV_OK;
    if (n > (int64_t)1 << 30) return dlv_fail(ctx, DLV_EUNSUP, "too many windows");
    const int z0 = p->nz > 0 ? p->z0 : 0, nz = p->nz > 0 ? p->nz : p->Zp;
    const int ny = (int)t.st[1].size(), nx = (int)t.st[2].size();
    std::vector<int> starts((size_t)n * 3);
    for (int64_t g = wb; g < we; ++g) {
        const int iz = (int)(g / ((int64_t)ny * nx)), iy = (int)((g / nx) % ny), ix = (int)(g % nx);
        const int z = t.st[0][iz];
        if (z < z0 || z + t.roi[0] > z0 + nz) return dlv_fail(ctx, DLV_EINVAL, "window %lld outside the slab", (long long)g);
        const int64_t i = g - wb;
        starts[3 * i] = z - z0;
        starts[3 * i + 1] = t.st[1][iy];
        starts[3 * i + 2] = t.st[2][ix];
    }
    return window_maxima(ctx, vol_dev, p->Yp, p->Xp, t.roi, starts, n, wmax);
}

// windows per forward launch when the caller leaves it to the library (dlv_sw_params.sw_batch == 0)
static int default_sw_batch(long long tile_vox) {
    // ~2^25 patch voxels per forward (16 windows of 128^3, 56 of 96x96x64, ...), capped at 64 windows: ~10 GB of 16-bit
    // activations, and enough tiles at the deep levels to fill 256 CUs.  (64 x 64 x 32 windows, measured in round 6: 256 per
    // launch instead of 64 make the pass 20 % SLOWER - the level-0 tensors of a launch grow from 0.5 to 2 GB and stop finding
    // each other's lines in L2 / MALL)
    return (int)std::min<long long>(std::max<long long>(((long long)1 << 25) / tile_vox, 1), 64);
}

int dlv_reserve_dev(dlv_ctx* ctx, const dlv_sw_params* p, int Z, int Y, int X) {
    if (!ctx || !p) return DLV_EINVAL;
    if (p->roi[0] <= 0 || p->roi[1] <= 0 || p->roi[2] <= 0) return dlv_fail(ctx, DLV_EINVAL, "dlv_reserve_dev: window dimensions");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    const long long tile_vox = (long long)p->roi[0] * p->roi[1] * p->roi[2];
    const int B = p->sw_batch > 0 ? p->sw_batch : default_sw_batch(tile_vox);
    if (p->precision != DLV_PREC_F32) {
        const int lanes = ctx->aux[0] != nullptr ? std::max(1, std::min(ctx->lanes_wanted, DLV_MAX_LANES)) : 1;
        DLV_TRY(dlv_unet_reserve_16(ctx, B, p->roi[0], p->roi[1], p->roi[2], lanes));
    }
    if (Z > 0 && Y > 0 && X > 0) {  // the distance map of the eroded re-mask (finalize.hip)
        void* q;
        DLV_TRY(dlv_ws_get(ctx, WS_ERODE, (size_t)Z * Y * X, &q));
    }
    return DLV_OK;
}

int dlv_sw_infer_dev(dlv_ctx* ctx, const dlv_sw_params* p, const uint16_t* vol_dev, float* acc_dev, uint8_t* cnt_dev,
                     dlv_sw_stats* stats) {
    if (!ctx || !p || !vol_dev || !acc_dev) return DLV_EINVAL;
    if (!ctx->weights_loaded) return dlv_fail(ctx, DLV_ESTATE, "dlv_sw_infer_dev before dlv_unet_load");
    if (p->precision != DLV_PREC_F32 && p->precision != DLV_PREC_BF16 && p->precision != DLV_PREC_F16 && p->precision != DLV_PREC_BF16_ALL)
        return dlv_fail(ctx, DLV_EINVAL, "unknown precision %d", p->precision);
    if (!(p->flip_dim == -1 || (p->flip_dim >= 2 && p->flip_dim <= 4)))
        return dlv_fail(ctx, DLV_EINVAL, "flip_dim must be -1, 2, 3 or 4");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    Tiler t;
    DLV_TRY(build_tiler(ctx, p, t));
    const int d = t.roi[0], h = t.roi[1], w = t.roi[2];
    if (d < 16 || h < 16 || w < 16 || (long long)(d >> 4) * (h >> 4) * (w >> 4) < 2)
        return dlv_fail(ctx, DLV_EUNSUP, "window %dx%dx%d: every dimension must be at least 16 and level 4 (each dimension / 16, rounded down) must hold more "
                        "than one voxel - InstanceNorm3d has no statistics of a single value and torch raises there; any size from there on: "
                        "levels with an odd size are pooled and padded like MONAI's MaxPool3d / UpCat", d, h, w);
    const int64_t total = t.count();
    const int64_t wb = std::max<int64_t>(p->win_begin, 0);
    const int64_t we = p->win_end > 0 ? std::min<int64_t>(p->win_end, total) : total;
    const int z0 = p->nz > 0 ? p->z0 : 0;
    const int nz = p->nz > 0 ? p->nz : p->Zp;
    const int Yp = p->Yp, Xp = p->Xp;
    const int rep = p->repeat > 0 ? p->repeat : 1;
    // Gaussian importance map (option): MONAI 1.2.0 compute_importance_map(mode="gaussian") [3P-recall] - a delta at
    // roi//2 filtered by separable truncated (4 sigma) erf-form Gaussians = outer product of three 1-D factors,
    // divided by its maximum, floored at its smallest non-zero value
    ctx->blend_w = nullptr;
    ctx->blend_wsum = nullptr;
    ctx->blend_min = 0.f;
    if (p->blend_mode == DLV_BLEND_GAUSSIAN) {
        if (cnt_dev) return dlv_fail(ctx, DLV_EINVAL, "Gaussian blend: pass wsum_dev (fp32) instead of the uint8 count map");
        const float ss = p->sigma_scale > 0.f ? p->sigma_scale : 0.125f;
        std::vector<float> g((size_t)d + h + w);
        float gmin = 1.f;
        size_t off = 0;
        for (int k = 0; k < 3; ++k) {
            const int len = t.roi[k], center = len / 2;
            const float sigma = (float)len * ss;
            const int tail = (int)(std::max(sigma * 4.0f, 0.5f) + 0.5f);
            const float tt = 0.70710678f / std::fabs(sigma);  // float arithmetic like MONAI's gaussian_1d
            float mx = 0.f, mn = 0.f;
            for (int i = 0; i < len; ++i) {
                const int x = i - center;
                float v = 0.f;
                if (x >= -tail && x <= tail) v = std::max(0.f, 0.5f * (std::erf(tt * ((float)x + 0.5f)) - std::erf(tt * ((float)x - 0.5f))));
                g[off + i] = v;
                mx = std::max(mx, v);
            }
            for (int i = 0; i < len; ++i) {
                g[off + i] /= mx;
                if (g[off + i] > 0.f && (mn == 0.f || g[off + i] < mn)) mn = g[off + i];
            }
            gmin *= mn;
            off += len;
        }
        float* gw;
        DLV_TRY(dlv_ws_get(ctx, WS_BLEND_W, g.size() * sizeof(float), (void**)&gw));
        DLV_HIP(ctx, hipMemcpyAsync(gw, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        DLV_HIP(ctx, hipStreamSynchronize(ctx->stream));  // g is a local
        ctx->blend_w = gw;
        ctx->blend_min = gmin;
        ctx->blend_wsum = p->wsum_dev;
    } else if (p->blend_mode != DLV_BLEND_CONSTANT) {
        return dlv_fail(ctx, DLV_EINVAL, "unknown blend_mode %d", p->blend_mode);
    }
    if (cnt_dev) {
        // the count map is uint8 (the reference's LOAD_ALL_RAM map, inference.py:241): refuse a geometry whose overlap
        // multiplicity times `repeat` cannot be held instead of wrapping silently
        long long mult = 1;
        for (int k = 0; k < 3; ++k) {
            std::vector<int> cov((size_t)t.n[k] + 1, 0);
            for (int st : t.st[k]) {
                cov[st] += 1;
                cov[st + t.roi[k]] -= 1;
            }
            int run = 0, mx = 0;
            for (int i = 0; i < t.n[k]; ++i) {
                run += cov[i];
                mx = std::max(mx, run);
            }
            mult *= mx;
        }
        if (mult * rep > 255)
            return dlv_fail(ctx, DLV_EUNSUP, "uint8 count map would wrap: up to %lld windows cover a voxel, x repeat %d > 255", mult, rep);
    }
    if (stats) {
        stats->n_windows = std::max<int64_t>(we - wb, 0);
        stats->n_skipped = 0;
        stats->n_forward_launches = 0;
    }
    if (we <= wb) return DLV_OK;
    const int64_t nwin = we - wb;
    if (nwin > (int64_t)1 << 30) return dlv_fail(ctx, DLV_EUNSUP, "too many windows in one shard");

    // windows of the shard, grouped by colour class (stable within a class = reference order)
    const int ny = (int)t.st[1].size(), nx = (int)t.st[2].size();
    const int ncolors = t.ncol[0] * t.ncol[1] * t.ncol[2];
    std::vector<std::vector<int>> by_color(ncolors);
    std::vector<int> starts((size_t)nwin * 3);
    for (int64_t g = wb; g < we; ++g) {
        const int iz = (int)(g / ((int64_t)ny * nx)), iy = (int)((g / nx) % ny), ix = (int)(g % nx);
        const int i = (int)(g - wb);
        const int sz = t.st[0][iz], sy = t.st[1][iy], sx = t.st[2][ix];
        if (sz < z0 || sz + d > z0 + nz)
            return dlv_fail(ctx, DLV_EINVAL, "window %lld (z %d..%d) lies outside the slab [%d,%d)", (long long)g, sz,
                            sz + d, z0, z0 + nz);
        starts[3 * i] = sz - z0;
        starts[3 * i + 1] = sy;
        starts[3 * i + 2] = sx;
        const int c = (t.col[0][iz] * t.ncol[1] + t.col[1][iy]) * t.ncol[2] + t.col[2][ix];
        by_color[c].push_back(i);
    }

    // device metadata: the per-launch start lists (the window maxima use their own scratch: window_maxima)
    int* meta;
    DLV_TRY(dlv_ws_get(ctx, WS_TILE_META, (size_t)nwin * 3 * sizeof(int), (void**)&meta));
    int* list_dev = meta;
    const long long tile_vox = (long long)d * h * w;
    std::vector<int> wmax((size_t)nwin);
    DLV_TRY(window_maxima(ctx, vol_dev, Yp, Xp, t.roi, starts, nwin, wmax.data()));

    // ordered launch lists: per colour [active..., skipped...]
    std::vector<int> list;
    list.reserve((size_t)nwin * 3);
    struct Seg { size_t off; int n_active, n_skipped; };
    std::vector<Seg> segs;
    int64_t n_skipped = 0;
    for (int c = 0; c < ncolors; ++c) {
        Seg s{list.size() / 3, 0, 0};
        for (int pass = 0; pass < 2; ++pass)
            for (int i : by_color[c]) {
                const bool skip = wmax[i] <= p->skip_threshold;
                if ((pass == 0) == skip) continue;
                list.push_back(starts[3 * i]);
                list.push_back(starts[3 * i + 1]);
                list.push_back(starts[3 * i + 2]);
                if (skip) ++s.n_skipped; else ++s.n_active;
            }
        n_skipped += s.n_skipped;
        segs.push_back(s);
    }
    DLV_HIP(ctx, hipMemcpyAsync(list_dev, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));

    int sw_batch = p->sw_batch;
    if (sw_batch <= 0) sw_batch = default_sw_batch(tile_vox);
    int64_t launches = 0;
    const int bchunks = (int)std::min<long long>(std::max<long long>(tile_vox / (256 * 16), 1), 256);
    // Pipeline lanes (HIP streams): consecutive batches of the 16-bit path rotate over them so
    // that the HBM-bound kernels of one batch (stem, norm, deconv, blend) overlap the MFMA-bound
    // convolutions of the other.  Windows of one colour class are disjoint, so their blends may run
    // concurrently; the lanes are joined at every class boundary, which keeps the per-voxel summation
    // order (colour by colour) and therefore the bits of the result.
    const int nlanes = (p->precision != DLV_PREC_F32 && ctx->aux[0] != nullptr) ? std::max(1, std::min(ctx->lanes_wanted, DLV_MAX_LANES)) : 1;
    const bool two_lanes = nlanes > 1;
    auto lane_stream = [&](int l) { return l == 0 ? ctx->main_stream : ctx->aux[l - 1]; };
    // the streams that take part in a join: the lanes' streams
    std::vector<hipStream_t> parts;
    for (int l = 0; l < nlanes; ++l) parts.push_back(lane_stream(l));
    auto join_lanes = [&]() -> int {
        if (!two_lanes) return DLV_OK;
        for (size_t l = 0; l < parts.size(); ++l) DLV_HIP(ctx, hipEventRecord(ctx->ev_lane[l], parts[l]));
        for (size_t l = 0; l < parts.size(); ++l)
            for (size_t m = 0; m < parts.size(); ++m)
                if (m != l) DLV_HIP(ctx, hipStreamWaitEvent(parts[l], ctx->ev_lane[m], 0));
        return DLV_OK;
    };
    if (p->precision != DLV_PREC_F32) DLV_TRY(dlv_range_reset(ctx));
    int rc = join_lanes();  // the aux lane starts after everything queued on the main stream so far
    int lane = 0;
    for (const Seg& s : segs) {
        if (rc != DLV_OK) break;
        for (int b0 = 0; b0 < s.n_active && rc == DLV_OK; b0 += sw_batch) {
            const int B = std::min(sw_batch, s.n_active - b0);
            const int* st_dev = list_dev + (s.off + b0) * 3;
            if (p->precision != DLV_PREC_F32) {
                ctx->lane = two_lanes ? lane : 0;
                ctx->stream = lane_stream(ctx->lane);
                rc = dlv_unet_tiles_bf16(ctx, vol_dev, Yp, Xp, st_dev, B, d, h, w, p->flip_dim, (float)rep, acc_dev, dlv_fmt16(p->precision));
                if (rc == DLV_OK && cnt_dev) {
                    hipLaunchKernelGGL(fill_add_kernel, dim3(bchunks, B), dim3(256), 0, ctx->stream, st_dev, d, h, w, Yp,
                                       Xp, 0.0f, rep, acc_dev, cnt_dev, nullptr, 0.f, nullptr);
                    if (hipGetLastError() != hipSuccess) rc = dlv_fail(ctx, DLV_EHIP, "launch of fill_add_kernel(count) failed");
                }
                lane = (lane + 1) % nlanes;
            } else {
                float *tin, *tout;
                rc = dlv_ws_get(ctx, WS_TILE_IN, (size_t)sw_batch * tile_vox * 4, (void**)&tin);
                if (rc == DLV_OK) rc = dlv_ws_get(ctx, WS_TILE_OUT, (size_t)sw_batch * tile_vox * 4, (void**)&tout);
                if (rc != DLV_OK) break;
                hipLaunchKernelGGL(gather_f32_kernel, dim3(bchunks, B), dim3(256), 0, ctx->stream, vol_dev, Yp, Xp,
                                   st_dev, d, h, w, p->flip_dim, tin);
                rc = dlv_unet_forward_f32(ctx, tin, tout, B, d, h, w);
                if (rc != DLV_OK) break;
                hipLaunchKernelGGL(blend_add_kernel, dim3(bchunks, B), dim3(256), 0, ctx->stream, tout, st_dev, d, h, w,
                                   p->flip_dim, Yp, Xp, (float)rep, rep, acc_dev, cnt_dev, ctx->blend_w, ctx->blend_min, ctx->blend_wsum);
                if (hipGetLastError() != hipSuccess) rc = dlv_fail(ctx, DLV_EHIP, "launch of gather/blend kernel failed");
            }
            ++launches;
        }
        ctx->stream = ctx->main_stream;
        ctx->lane = 0;
        if (rc == DLV_OK && s.n_skipped > 0) {
            const int* st_dev = list_dev + (s.off + s.n_active) * 3;
            DlvProf pr(ctx, "skip_fill_f32", 0.0, 8.0 * tile_vox * s.n_skipped);
            hipLaunchKernelGGL(fill_add_kernel, dim3(bchunks, s.n_skipped), dim3(256), 0, ctx->stream, st_dev, d, h, w,
                               Yp, Xp, -1000.0f * rep, rep, acc_dev, cnt_dev, ctx->blend_w, ctx->blend_min, ctx->blend_wsum);
            pr.end();
            if (hipGetLastError() != hipSuccess) rc = dlv_fail(ctx, DLV_EHIP, "launch of fill_add_kernel failed");
        }
        if (rc == DLV_OK) rc = join_lanes();  // colour-class boundary
    }
    ctx->stream = ctx->main_stream;
    ctx->lane = 0;
    if (rc != DLV_OK) return rc;
    if (stats) {
        stats->n_skipped = n_skipped;
        stats->n_forward_launches = launches;
    }
    // range guard of the 16-bit formats: every lane has been joined into the main stream; one read-back per pass (a pass is
    // seconds of queued work, and the next one starts with the read-back of its window maxima anyway)
    if (p->precision != DLV_PREC_F32 && launches > 0) return dlv_range_check(ctx, dlv_fmt16(p->precision));
    return DLV_OK;
}

}  // extern "C"
