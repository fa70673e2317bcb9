// finalize.hip - divide by the count map, threshold, eroded re-mask.
//
// Restates inference/inference.py:285-299 (block-wise divide) and :31-95 (create_nifti_seg):
//   mean = acc / cnt;  fg = sigmoid(float32(mean)) >= threshold
//   keep = binary_erosion(raw > 0, iterations=30, border_value=1)   [default 6-neighbourhood]
//   out  = uint8(fg) * keep, cropped to the unpadded (Z,Y,X) stack
// The erosion equals "taxicab distance to the nearest zero voxel INSIDE the block > iterations"
// (outside the block counts as foreground), so it is evaluated as a separable capped L1 distance
// transform: X by log-step min-plus doubling in LDS, Y and Z by forward/backward scans.  The
// reference erodes per Arrayterator z-block (inference.py:53); `zblock` carries that block size.
#include "common.h"
#ifndef ZXY_PFR
#define ZXY_PFR 1
#endif

namespace {

// distance along X.  One block per (z,y) row; the row lives in LDS as uint8.
__global__ void __launch_bounds__(256) erode_x_kernel(const uint16_t* __restrict__ raw, int Yp, int Xp, int Y, int X,
                                                      int cap, uint8_t* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* a = smem;
    unsigned char* b = smem + X;
    const int y = blockIdx.x % Y, z = blockIdx.x / Y;
    const uint16_t* row = raw + ((long long)z * Yp + y) * Xp;
    for (int x = threadIdx.x; x < X; x += blockDim.x) a[x] = row[x] > 0 ? (unsigned char)cap : 0;
    __syncthreads();
    for (int k = 1; k < cap; k <<= 1) {
        for (int x = threadIdx.x; x < X; x += blockDim.x) {
            int v = a[x];
            if (x - k >= 0) v = min(v, (int)a[x - k] + k);
            if (x + k < X) v = min(v, (int)a[x + k] + k);
            b[x] = (unsigned char)v;
        }
        __syncthreads();
        unsigned char* t = a;
        a = b;
        b = t;
    }
    uint8_t* o = dist + ((long long)z * Y + y) * X;
    for (int x = threadIdx.x; x < X; x += blockDim.x) o[x] = a[x];
}

// the same distance for cap <= 57 (the reference's 30 iterations: cap 31) without the log-step sweeps: the row's zero
// voxels become a bit mask in LDS (one byte per 8 voxels); a thread owns 8 consecutive voxels (one 16-byte load, one
// 8-byte store) and finds the nearest zero on either side of each with a count-leading/trailing-zeros on a 64-bit window
// of that mask (bits left of x: window ending at the thread's last voxel; bits right of x: window starting at its first).
// Voxels outside [0, X) count as foreground (border_value = 1).
__global__ void __launch_bounds__(256) erode_x_bits_kernel(const uint16_t* __restrict__ raw, int Yp, int Xp, int Y, int X,
                                                           int cap, uint8_t* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* zm64 = reinterpret_cast<unsigned long long*>(smem);  // bytes: 8 zero bytes, X/8 mask bytes, 16 zero bytes
    unsigned char* zm = smem + 8;
    const int y = blockIdx.x % Y, z = blockIdx.x / Y;
    const uint16_t* row = raw + ((long long)z * Yp + y) * Xp;
    const int nchunk = (X + 7) / 8;
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    for (int i = threadIdx.x; i < (nchunk + 24 + 7) / 8; i += blockDim.x) zm64[i] = 0ull;
    __syncthreads();
    for (int c = threadIdx.x; c < nchunk; c += blockDim.x) {
        unsigned m = 0;
        if (vec && 8 * c + 8 <= X) {
            const uint4 u = *reinterpret_cast<const uint4*>(row + 8 * c);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m |= ((w[k] & 0xffffu) == 0u ? 1u : 0u) << (2 * k);
                m |= ((w[k] >> 16) == 0u ? 1u : 0u) << (2 * k + 1);
            }
        } else {
            for (int k = 0; k < 8; ++k)
                if (8 * c + k < X && row[8 * c + k] == 0) m |= 1u << k;
        }
        zm[c] = (unsigned char)m;
    }
    __syncthreads();
    uint8_t* o = dist + ((long long)z * Y + y) * X;
    const bool ovec = ((reinterpret_cast<uintptr_t>(o) & 7) == 0);
    for (int c = threadIdx.x; c < nchunk; c += blockDim.x) {
        // bytes c-7 .. c+8 of the mask (smem offset c+1 .. c+16) out of three aligned 64-bit words
        const int off = c + 1;
        const int a = off >> 3, sh = (off & 7) * 8;
        const unsigned long long w0 = zm64[a], w1 = zm64[a + 1], w2 = zm64[a + 2];
        const unsigned long long left = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;    // bits: voxels 8c-56 .. 8c+7
        const unsigned long long right = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;   // bits: voxels 8c+8 .. 8c+71
        const unsigned long long own = left >> 56;                                    // voxels 8c .. 8c+7
        const unsigned long long rwin = own | (right << 8);                           // voxels 8c .. 8c+63
        unsigned long long res = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long L = left & (~0ull >> (7 - k));  // bits at or left of voxel 8c+k
            const unsigned long long R = rwin >> k;                  // bit 0 = voxel 8c+k
            int d = cap;
            if (L) d = min(d, __clzll((long long)L) - (7 - k));
            if (R) d = min(d, (int)__ffsll((long long)R) - 1);
            res |= (unsigned long long)(unsigned)d << (8 * k);
        }
        if (ovec && 8 * c + 8 <= X) {
            *reinterpret_cast<unsigned long long*>(o + 8 * c) = res;
        } else {
            for (int k = 0; k < 8 && 8 * c + k < X; ++k) o[8 * c + k] = (uint8_t)(res >> (8 * k));
        }
    }
}

template <int V>
struct U8V;
template <>
struct U8V<1> {
    using T = unsigned char;
    static __device__ __forceinline__ unsigned ld(const uint8_t* p) { return *p; }
    static __device__ __forceinline__ void st(uint8_t* p, unsigned v) { *p = (unsigned char)v; }
};
template <>
struct U8V<4> {
    // (non-temporal: the distance and mask volumes are streamed once per kernel and are far larger than L2 + MALL)
    static __device__ __forceinline__ unsigned ld(const uint8_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p)); }
    static __device__ __forceinline__ void st(uint8_t* p, unsigned v) { __builtin_nontemporal_store(v, reinterpret_cast<unsigned*>(p)); }
};

// per-byte min(a, b+1) for V packed bytes (values <= 254)
template <int V>
__device__ __forceinline__ unsigned minplus1(unsigned cur, unsigned prev) {
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const unsigned c = (cur >> (8 * k)) & 0xffu, p = ((prev >> (8 * k)) & 0xffu) + 1u;
        r |= min(c, p) << (8 * k);
    }
    return r;
}

// distance along Y, in place: forward then backward scan.  One thread per (z, V consecutive x).
template <int V>
__global__ void __launch_bounds__(256) erode_y_kernel(uint8_t* __restrict__ dist, int Z, int Y, int X) {
    const int xv = X / V;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)Z * xv) return;
    const int x = (int)(t % xv) * V, z = (int)(t / xv);
    uint8_t* col = dist + (long long)z * Y * X + x;
    unsigned prev = U8V<V>::ld(col);
    for (int y = 1; y < Y; ++y) {
        const unsigned cur = minplus1<V>(U8V<V>::ld(col + (long long)y * X), prev);
        U8V<V>::st(col + (long long)y * X, cur);
        prev = cur;
    }
    for (int y = Y - 2; y >= 0; --y) {
        const unsigned cur = minplus1<V>(U8V<V>::ld(col + (long long)y * X), prev);
        U8V<V>::st(col + (long long)y * X, cur);
        prev = cur;
    }
}

// distance along Z inside z-blocks + the final decision.  One thread per (block, y, V x).
template <int V>
__global__ void __launch_bounds__(256) erode_z_final_kernel(uint8_t* __restrict__ dist, const float* __restrict__ acc,
                                                            const uint8_t* __restrict__ cnt, int Yp, int Xp, int Z,
                                                            int Y, int X, int zblock, int zphase, int radius, float threshold,
                                                            uint8_t* __restrict__ out, float* __restrict__ prob) {
    const int xv = X / V;
    const long long per_block = (long long)Y * xv;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // z-blocks at absolute multiples of zblock; local plane z is absolute plane z + zphase (mod zblock)
    const int nblk = (Z + zphase + zblock - 1) / zblock;
    if (t >= per_block * nblk) return;
    const int blk = (int)(t / per_block);
    const long long r = t % per_block;
    const int x = (int)(r % xv) * V, y = (int)(r / xv);
    const int zb0 = max(blk * zblock - zphase, 0), zb1 = min((blk + 1) * zblock - zphase, Z);
    const long long plane = (long long)Y * X;
    uint8_t* col = dist + (long long)y * X + x;
    unsigned prev = U8V<V>::ld(col + zb0 * plane);
    for (int z = zb0 + 1; z < zb1; ++z) {
        const unsigned cur = minplus1<V>(U8V<V>::ld(col + z * plane), prev);
        U8V<V>::st(col + z * plane, cur);
        prev = cur;
    }
    for (int z = zb1 - 1; z >= zb0; --z) {
        unsigned cur = U8V<V>::ld(col + z * plane);
        if (z < zb1 - 1) cur = minplus1<V>(cur, prev);
        prev = cur;
        unsigned res = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const long long po = ((long long)z * Yp + y) * Xp + x + k;
            float m = acc[po];
            if (cnt) m = m / (float)cnt[po];  // 0/0 -> NaN -> background, as in the reference
            const float pr = 1.0f / (1.0f + expf(-m));
            const bool keep = ((cur >> (8 * k)) & 0xffu) > (unsigned)radius;
            if (prob) prob[(long long)z * plane + (long long)y * X + x + k] = pr;
            res |= ((pr >= threshold && keep) ? 1u : 0u) << (8 * k);
        }
        U8V<V>::st(out + (long long)z * plane + (long long)y * X + x, res);
    }
}

// X and Y distances fused: one workgroup per plane z marches down y.  Per row: the raw row -> zero-voxel bit mask in LDS
// (double-buffered: one barrier per row) -> the x distance of the thread's 8 voxels by the count-leading / trailing-zeros trick
// of erode_x_bits_kernel -> forward y scan against the carried previous row -> 8-byte store; then the backward y scan in
// place.  The x distance never goes to HBM: raw 2 B + dist 1 B (forward) + dist 1 + 1 B (backward) = 5 B per voxel instead
// of 3 + 4.  NC = 8-voxel chunks per thread (X <= 2048 * NC); cap <= 57.
template <int NC>
__global__ void __launch_bounds__(256) erode_xy_kernel(const uint16_t* __restrict__ raw, int Yp, int Xp, int Y, int X, int cap,
                                                       uint8_t* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nchunk = (X + 7) / 8;
    const int mstride = ((nchunk + 24 + 7) / 8) * 8;  // bytes per mask buffer: 8 zero bytes, the mask, >= 16 zero bytes
    const int z = blockIdx.x;
    const bool vec = (Xp % 8 == 0) && ((reinterpret_cast<uintptr_t>(raw) & 15) == 0);
    const bool ovec = (X % 8 == 0) && ((reinterpret_cast<uintptr_t>(dist) & 7) == 0);
    for (int i = threadIdx.x; i < 2 * mstride / 8; i += 256) reinterpret_cast<unsigned long long*>(smem)[i] = 0ull;
    __syncthreads();
    unsigned long long prev[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) prev[q] = 0x0101010101010101ull * (unsigned long long)cap;  // above the volume: foreground
    auto bits_of = [](const uint4& u) -> unsigned {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m |= ((w[k] & 0xffffu) == 0u ? 1u : 0u) << (2 * k);
            m |= ((w[k] >> 16) == 0u ? 1u : 0u) << (2 * k + 1);
        }
        return m;
    };
    auto row_load = [&](int y, int c) -> uint4 {  // the 8 voxels of chunk c of row y (a ragged or unaligned row: voxel by voxel)
        const uint16_t* row = raw + ((long long)z * Yp + y) * Xp;
        if (vec && 8 * c + 8 <= X) return dlv_ld16<true>(reinterpret_cast<const uint4*>(row + 8 * c));  // (nt: every byte of these volumes is touched once per kernel)
        unsigned w[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // beyond X: foreground
        for (int k = 0; k < 8; ++k)
            if (8 * c + k < X) w[k >> 1] = (w[k >> 1] & ~(0xffffu << (16 * (k & 1)))) | ((unsigned)row[8 * c + k] << (16 * (k & 1)));
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    constexpr int PFR = ZXY_PFR;  // rows y + 1 .. y + PFR are in flight while row y is processed (4 measured slower than 1 or 2)
    uint4 nxt[PFR][NC];
#pragma unroll
    for (int r = 0; r < PFR; ++r)
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = threadIdx.x + 256 * q;
            nxt[r][q] = (c < nchunk && r < Y) ? row_load(r, c) : make_uint4(~0u, ~0u, ~0u, ~0u);
        }
    auto minplus8 = [](unsigned long long cur, unsigned long long pv) -> unsigned long long {
        return (unsigned long long)minplus1<4>((unsigned)cur, (unsigned)pv) |
               ((unsigned long long)minplus1<4>((unsigned)(cur >> 32), (unsigned)(pv >> 32)) << 32);
    };
    for (int y0 = 0; y0 < Y; y0 += PFR) {
#pragma unroll
      for (int r = 0; r < PFR; ++r) {
        const int y = y0 + r;
        if (y >= Y) break;  // (workgroup-uniform)
        unsigned char* zm = smem + (y & 1) * mstride + 8;
        const unsigned long long* zm64 = reinterpret_cast<const unsigned long long*>(smem + (y & 1) * mstride);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = threadIdx.x + 256 * q;
            if (c < nchunk) {
                zm[c] = (unsigned char)bits_of(nxt[r][q]);
                if (y + PFR < Y) nxt[r][q] = row_load(y + PFR, c);
            }
        }
        __syncthreads();  // (the other buffer is rewritten only after the NEXT barrier: every reader of it has passed this one)
        uint8_t* o = dist + ((long long)z * Y + y) * X;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = threadIdx.x + 256 * q;
            if (c >= nchunk) continue;
            const int off = c + 1;
            const int a = off >> 3, sh = (off & 7) * 8;
            const unsigned long long w0 = zm64[a], w1 = zm64[a + 1], w2 = zm64[a + 2];
            const unsigned long long left = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
            const unsigned long long right = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
            const unsigned long long own = left >> 56;
            const unsigned long long rwin = own | (right << 8);
            unsigned long long dx = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned long long L = left & (~0ull >> (7 - k));
                const unsigned long long R = rwin >> k;
                int d = cap;
                if (L) d = min(d, __clzll((long long)L) - (7 - k));
                if (R) d = min(d, (int)__ffsll((long long)R) - 1);
                dx |= (unsigned long long)(unsigned)d << (8 * k);
            }
            const unsigned long long fwd = minplus8(dx, prev[q]);
            prev[q] = fwd;
            if (ovec && 8 * c + 8 <= X) {
                __builtin_nontemporal_store(fwd, reinterpret_cast<unsigned long long*>(o + 8 * c));
            } else {
                for (int k = 0; k < 8 && 8 * c + k < X; ++k) o[8 * c + k] = (uint8_t)(fwd >> (8 * k));
            }
        }
      }
    }
    // backward y scan (prev holds row Y-1), rows prefetched PFR ahead
    if (ovec) {
        unsigned long long bq[PFR][NC];
#pragma unroll
        for (int r = 0; r < PFR; ++r)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = threadIdx.x + 256 * q, y = Y - 2 - r;
                bq[r][q] = (c < nchunk && y >= 0) ? __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(dist + ((long long)z * Y + y) * X + 8 * c)) : 0ull;
            }
        for (int yb = Y - 2; yb >= 0; yb -= PFR) {
#pragma unroll
            for (int r = 0; r < PFR; ++r) {
                const int y = yb - r;
                if (y < 0) break;
                uint8_t* o = dist + ((long long)z * Y + y) * X;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int c = threadIdx.x + 256 * q;
                    if (c >= nchunk) continue;
                    const unsigned long long cur = minplus8(bq[r][q], prev[q]);
                    if (y - PFR >= 0) bq[r][q] = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(dist + ((long long)z * Y + (y - PFR)) * X + 8 * c));
                    prev[q] = cur;
                    __builtin_nontemporal_store(cur, reinterpret_cast<unsigned long long*>(o + 8 * c));
                }
            }
        }
        return;
    }
    for (int y = Y - 2; y >= 0; --y) {
        uint8_t* o = dist + ((long long)z * Y + y) * X;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = threadIdx.x + 256 * q;
            if (c >= nchunk) continue;
            unsigned long long cur = 0;
            for (int k = 0; k < 8; ++k) cur |= (unsigned long long)(8 * c + k < X ? o[8 * c + k] : cap) << (8 * k);
            cur = minplus8(cur, prev[q]);
            prev[q] = cur;
            for (int k = 0; k < 8 && 8 * c + k < X; ++k) o[8 * c + k] = (uint8_t)(cur >> (8 * k));
        }
    }
}

// The same decision in ONE sweep along z (radius <= 31): the forward scan f(j) = min(d(j), f(j-1) + 1) alone decides, because
// plane j removes exactly the planes z in [j - (radius - f(j)), j] (b(z) = min_k f(z+k) + k <= radius).  Every column keeps a
// 32-bit shift register of "removed" flags for the `radius` + 1 planes that are still pending; the output plane lags the
// distance plane by `radius` planes.  Reads the (x, y) distance once, the sums once, writes the mask once: 6 B per voxel
// instead of 8, and the distance map is not rewritten.
template <int V>
__global__ void __launch_bounds__(256) erode_z_shift_kernel(const uint8_t* __restrict__ dist, const float* __restrict__ acc,
                                                            const uint8_t* __restrict__ cnt, int Yp, int Xp, int Z,
                                                            int Y, int X, int zblock, int zphase, int radius, float threshold,
                                                            uint8_t* __restrict__ out, float* __restrict__ prob) {
    const int xv = X / V;
    const long long per_block = (long long)Y * xv;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nblk = (Z + zphase + zblock - 1) / zblock;
    if (t >= per_block * nblk) return;
    const int blk = (int)(t / per_block);
    const long long r = t % per_block;
    const int x = (int)(r % xv) * V, y = (int)(r / xv);
    const int zb0 = max(blk * zblock - zphase, 0), zb1 = min((blk + 1) * zblock - zphase, Z);
    const long long plane = (long long)Y * X;
    const uint8_t* col = dist + (long long)y * X + x;
    const bool acc_vec = (Xp % 4 == 0) && ((reinterpret_cast<uintptr_t>(acc) & 15) == 0);  // (x is a multiple of V)
    unsigned f = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) f |= (unsigned)(radius + 1) << (8 * k);  // outside the block: foreground
    unsigned kill[V];
#pragma unroll
    for (int k = 0; k < V; ++k) kill[k] = 0u;
    // software pipeline: the distance word of plane j + PF and the sums of output plane z + PF are in flight while plane j /
    // z is processed (without it every iteration exposes a full memory latency: 3.5 TB/s instead of ~5)
    constexpr int PF = 4;
    unsigned dq[PF];
    float aq[PF][V];
    unsigned cq[PF];  // count bytes (V <= 4 per word)
    auto ld_dist = [&](int j) -> unsigned { return j < zb1 ? U8V<V>::ld(col + (long long)j * plane) : 0u; };
    auto ld_acc = [&](int z, float (&a)[V], unsigned& c) {
        c = 0u;
        if (z >= zb0 && z < zb1) {
            const long long po = ((long long)z * Yp + y) * Xp + x;
            if (V == 4 && acc_vec) {  // one 16-byte load per lane: a wave instruction covers whole lines (the compiler's 4 + 12 byte
                typedef float f4_t __attribute__((ext_vector_type(4)));  // split touches every line twice)
                const f4_t t = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(acc + po));
#pragma unroll
                for (int k = 0; k < V; ++k) a[k] = t[k % 4];
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) a[k] = __builtin_nontemporal_load(acc + po + k);
            }
            if (cnt) {
#pragma unroll
                for (int k = 0; k < V; ++k) c |= (unsigned)cnt[po + k] << (8 * k);
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] = 0.f;
        }
    };
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        dq[q] = ld_dist(zb0 + q);
        ld_acc(zb0 - radius + q, aq[q], cq[q]);
    }
    const int jend = zb1 + radius;
    for (int j0 = zb0; j0 < jend; j0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int j = j0 + q;
            if (j >= jend) break;
            const unsigned dcur = dq[q];
            float acur[V];
#pragma unroll
            for (int k = 0; k < V; ++k) acur[k] = aq[q][k];
            const unsigned ccur = cq[q];
            dq[q] = ld_dist(j + PF);
            ld_acc(j + PF - radius, aq[q], cq[q]);
            if (j < zb1) {
                f = minplus1<V>(dcur, f);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const unsigned fk = (f >> (8 * k)) & 0xffu;
                    if (fk <= (unsigned)radius) kill[k] |= (0xffffffffu >> (31 - (radius - (int)fk))) << fk;  // bits [fk, radius]
                }
            }
            const int z = j - radius;
            if (z >= zb0) {
                unsigned res = 0;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float m = acur[k];
                    if (cnt) m = m / (float)((ccur >> (8 * k)) & 0xffu);  // 0/0 -> NaN -> background, as in the reference
                    const float pr = 1.0f / (1.0f + expf(-m));
                    if (prob) prob[(long long)z * plane + (long long)y * X + x + k] = pr;
                    res |= ((pr >= threshold && !(kill[k] & 1u)) ? 1u : 0u) << (8 * k);
                }
                U8V<V>::st(out + (long long)z * plane + (long long)y * X + x, res);
            }
#pragma unroll
            for (int k = 0; k < V; ++k) kill[k] >>= 1;
        }
    }
}

}  // namespace

static int finalize_impl(dlv_ctx* ctx, const float* acc_dev, const uint8_t* cnt_dev, const uint16_t* raw_dev, int Yp, int Xp, int Z,
                         int Y, int X, float threshold, int erode_iters, int zblock, int zphase, uint8_t* out_dev, float* prob_dev);

extern "C" int dlv_finalize_dev(dlv_ctx* ctx, const float* acc_dev, const uint8_t* cnt_dev, const uint16_t* raw_dev,
                                int Yp, int Xp, int Z, int Y, int X, float threshold, int erode_iters, int zblock,
                                uint8_t* out_dev, float* prob_dev) {
    if (Z > 0 && (zblock <= 0 || zblock > Z)) zblock = Z;
    return finalize_impl(ctx, acc_dev, cnt_dev, raw_dev, Yp, Xp, Z, Y, X, threshold, erode_iters, zblock, 0, out_dev, prob_dev);
}

extern "C" int dlv_finalize_slab_dev(dlv_ctx* ctx, const float* acc_dev, const uint8_t* cnt_dev, const uint16_t* raw_dev, int Yp,
                                     int Xp, int z_abs0, int nz, int Y, int X, float threshold, int erode_iters, int zblock,
                                     uint8_t* out_dev, float* prob_dev) {
    if (z_abs0 < 0) return DLV_EINVAL;
    if (zblock <= 0) zblock = 0x3fffffff;  // one block: the whole stack
    return finalize_impl(ctx, acc_dev, cnt_dev, raw_dev, Yp, Xp, nz, Y, X, threshold, erode_iters, zblock, z_abs0 % zblock, out_dev,
                         prob_dev);
}

static int finalize_impl(dlv_ctx* ctx, const float* acc_dev, const uint8_t* cnt_dev, const uint16_t* raw_dev, int Yp, int Xp, int Z,
                         int Y, int X, float threshold, int erode_iters, int zblock, int zphase, uint8_t* out_dev, float* prob_dev) {
    if (!ctx || !acc_dev || !raw_dev || !out_dev) return DLV_EINVAL;
    if (Z <= 0 || Y <= 0 || X <= 0 || Y > Yp || X > Xp) return dlv_fail(ctx, DLV_EINVAL, "bad shapes");
    if (erode_iters < 0 || erode_iters > 253) return dlv_fail(ctx, DLV_EUNSUP, "erode_iters must be in [0,253]");
    if (2 * (size_t)X > 160 * 1024) return dlv_fail(ctx, DLV_EUNSUP, "X=%d exceeds the LDS row buffer", X);
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    if (zblock <= 0) zblock = Z;
    const int cap = erode_iters + 1;
    uint8_t* dist;
    const long long nvox = (long long)Z * Y * X;
    DLV_TRY(dlv_ws_get(ctx, WS_ERODE, (size_t)nvox, (void**)&dist));
    const bool v4 = (X % 4 == 0);
    const bool split_xy = ctx->erode_xy_split;  // A/B + cross-check in tests (dlv_diag_set): separate x and y passes
    const int nchunk8 = (X + 7) / 8;
    if (cap <= 57 && nchunk8 <= 1024 && !split_xy) {
        // x and y fused: the x distance never reaches HBM
        DlvProf p(ctx, "erode_xy_u8", 0.0, 5.0 * nvox);
        const size_t lds = 2 * (size_t)(((nchunk8 + 24 + 7) / 8) * 8);
        if (nchunk8 <= 256)
            hipLaunchKernelGGL(erode_xy_kernel<1>, dim3((unsigned)Z), dim3(256), lds, ctx->stream, raw_dev, Yp, Xp, Y, X, cap, dist);
        else if (nchunk8 <= 512)
            hipLaunchKernelGGL(erode_xy_kernel<2>, dim3((unsigned)Z), dim3(256), lds, ctx->stream, raw_dev, Yp, Xp, Y, X, cap, dist);
        else
            hipLaunchKernelGGL(erode_xy_kernel<4>, dim3((unsigned)Z), dim3(256), lds, ctx->stream, raw_dev, Yp, Xp, Y, X, cap, dist);
        p.end();
        DLV_LAUNCH_CHECK(ctx, "erode_xy_kernel");
    } else {
    {
        DlvProf p(ctx, "erode_x_u8", 0.0, 3.0 * nvox);
        if (cap <= 57)
            hipLaunchKernelGGL(erode_x_bits_kernel, dim3((unsigned)((long long)Z * Y)), dim3(256), (size_t)((X + 7) / 8 + 24 + 8),
                               ctx->stream, raw_dev, Yp, Xp, Y, X, cap, dist);
        else
            hipLaunchKernelGGL(erode_x_kernel, dim3((unsigned)((long long)Z * Y)), dim3(256), 2 * (size_t)X, ctx->stream,
                               raw_dev, Yp, Xp, Y, X, cap, dist);
        p.end();
        DLV_LAUNCH_CHECK(ctx, "erode_x_kernel");
    }
    {
        DlvProf p(ctx, "erode_y_u8", 0.0, 4.0 * nvox);
        if (v4)
            hipLaunchKernelGGL(erode_y_kernel<4>, dim3(dlv_cdiv((long long)Z * (X / 4), 256)), dim3(256), 0, ctx->stream,
                               dist, Z, Y, X);
        else
            hipLaunchKernelGGL(erode_y_kernel<1>, dim3(dlv_cdiv((long long)Z * X, 256)), dim3(256), 0, ctx->stream, dist,
                               Z, Y, X);
        p.end();
        DLV_LAUNCH_CHECK(ctx, "erode_y_kernel");
    }
    }
    {
        const int nblk = (int)(((long long)Z + zphase + zblock - 1) / zblock);
        DlvProf p(ctx, "erode_z_final", 0.0, (4.0 + 4.0 + 1.0) * nvox);
        const bool two_sweeps = ctx->erode_z_two_sweeps;  // A/B + cross-check in tests (dlv_diag_set)
        if (erode_iters <= 31 && !two_sweeps) {  // one sweep with a shift register per column (the reference's 30 iterations)
            if (v4)
                hipLaunchKernelGGL(erode_z_shift_kernel<4>, dim3(dlv_cdiv((long long)nblk * Y * (X / 4), 256)), dim3(256), 0,
                                   ctx->stream, dist, acc_dev, cnt_dev, Yp, Xp, Z, Y, X, zblock, zphase, erode_iters, threshold,
                                   out_dev, prob_dev);
            else
                hipLaunchKernelGGL(erode_z_shift_kernel<1>, dim3(dlv_cdiv((long long)nblk * Y * X, 256)), dim3(256), 0,
                                   ctx->stream, dist, acc_dev, cnt_dev, Yp, Xp, Z, Y, X, zblock, zphase, erode_iters, threshold,
                                   out_dev, prob_dev);
        } else if (v4) {
            hipLaunchKernelGGL(erode_z_final_kernel<4>, dim3(dlv_cdiv((long long)nblk * Y * (X / 4), 256)), dim3(256), 0,
                               ctx->stream, dist, acc_dev, cnt_dev, Yp, Xp, Z, Y, X, zblock, zphase, erode_iters, threshold,
                               out_dev, prob_dev);
        } else {
            hipLaunchKernelGGL(erode_z_final_kernel<1>, dim3(dlv_cdiv((long long)nblk * Y * X, 256)), dim3(256), 0,
                               ctx->stream, dist, acc_dev, cnt_dev, Yp, Xp, Z, Y, X, zblock, zphase, erode_iters, threshold,
                               out_dev, prob_dev);
        }
        p.end();
        DLV_LAUNCH_CHECK(ctx, "erode_z_final_kernel");
    }
    return DLV_OK;
}
