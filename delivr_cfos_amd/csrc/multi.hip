// multi.hip - one process, N devices: the replacement of torch.nn.DataParallel (inference/inference.py:217-219: per-forward
// scatter of the batch, re-broadcast of every parameter, gather through GPU 0) behind the C ABI.
//   dlv_shard_plan_make   static partition of the reference's window list (Z slowest) into contiguous ranges: a rank's
//                         windows form a Z-slab of tile rows; planes are owned by exactly one rank (pure host logic)
//   dlv_comm_init_all     one context (device, streams, scratch) per device + one RCCL communicator per device
//   dlv_bcast_weights     ONE ncclBroadcast of the packed parameter blob (the reference re-broadcasts per forward)
//   dlv_sw_infer_sharded  every rank runs its window range on ITS slab of the volume / accumulator (one host thread per
//                         rank), then ONE point-to-point exchange per seam (ncclSend/ncclRecv, grouped): the planes a
//                         rank computed but another rank owns are added by the owner in source-rank order
// RCCL is loaded with dlopen at dlv_comm_init_all (a single-GPU host never needs librccl.so); ranks that share a device
// (tests on a one-GPU box) exchange with hipMemcpyAsync instead.  No all-reduce exists on this path: xGMI is
// point-to-point, every seam crosses exactly one link.
#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <thread>

#include "common.h"

// The few RCCL declarations this file needs (librccl.so is loaded with dlopen at run time, so the build does not depend on
// the RCCL development headers); values as in rccl/rccl.h of ROCm 7.x = NCCL's public ABI.
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclFloat32 = 7 } ncclDataType_t;

struct dlv_comm {
    int n = 0;
    std::vector<int> devs;
    std::vector<dlv_ctx*> ctx;
    std::vector<ncclComm_t> comm;  // empty: every rank on one device (loopback transport)
    void* rccl = nullptr;
    std::string err;
    // RCCL entry points
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // exchange scratch per rank (grow-only)
    std::vector<void*> stage;
    std::vector<size_t> stage_bytes;
};

namespace {

int comm_fail(dlv_comm* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define DLV_NCCL(c, expr)                                                                                          \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess)                                                                                     \
            return comm_fail((c), DLV_EHIP, "%s failed: %s", #expr, (c)->GetErrorString ? (c)->GetErrorString(_r) : "?"); \
    } while (0)
#define DLV_CHIP(c, expr)                                                                            \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return comm_fail((c), DLV_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

__global__ void __launch_bounds__(256) add_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}
__global__ void __launch_bounds__(256) add_u8_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (uint8_t)(dst[i] + src[i]);
}

int stage_get(dlv_comm* c, int r, size_t bytes, void** out) {
    if (c->stage_bytes[r] < bytes) {
        DLV_CHIP(c, hipSetDevice(c->devs[r]));
        if (c->stage[r]) {
            DLV_CHIP(c, hipStreamSynchronize(c->ctx[r]->main_stream));
            DLV_CHIP(c, hipFree(c->stage[r]));
            c->stage[r] = nullptr;
            c->stage_bytes[r] = 0;
        }
        DLV_CHIP(c, hipMalloc(&c->stage[r], bytes));
        c->stage_bytes[r] = bytes;
    }
    *out = c->stage[r];
    return DLV_OK;
}

}  // namespace

// no C++ exception crosses the C ABI (std::bad_alloc from the vectors, std::system_error from std::thread)
#define DLV_ABI_GUARD_BEGIN try {
#define DLV_ABI_GUARD_END(c)                                                                 \
    }                                                                                        \
    catch (const std::bad_alloc&) { return comm_fail((c), DLV_ENOMEM, "out of host memory"); } \
    catch (const std::exception& e) { return comm_fail((c), DLV_EHIP, "host exception: %s", e.what()); } \
    catch (...) { return comm_fail((c), DLV_EHIP, "unknown host exception"); }

extern "C" {

int dlv_shard_plan_make(const dlv_sw_params* p, int world, const float* weights, dlv_shard_plan* out) {
    if (!p || !out || world < 1 || world > DLV_MAX_RANKS) return DLV_EINVAL;
    DLV_ABI_GUARD_BEGIN
    int64_t n = 0;
    int rc = dlv_sw_num_windows(p, &n);
    if (rc != DLV_OK) return rc;
    std::vector<int64_t> st((size_t)n * 3);
    rc = dlv_sw_window_starts(p, st.data(), n);
    if (rc != DLV_OK) return rc;
    const int roi_z = p->roi[0] > 0 ? std::min(p->roi[0], p->Zp) : p->Zp;
    memset(out, 0, sizeof(*out));
    out->world = world;
    out->n_windows = n;
    std::vector<int64_t> cuts(1, 0);
    if (!weights) {
        // cuts snap to Z tile-row boundaries when there are enough rows: every seam is then one half-tile slab
        std::vector<int64_t> edges(1, 0);
        for (int64_t i = 1; i < n; ++i)
            if (st[3 * i] != st[3 * (i - 1)]) edges.push_back(i);
        edges.push_back(n);
        for (int r = 1; r < world; ++r) {
            const double target = (double)n * r / world;
            int64_t c;
            if ((int64_t)edges.size() - 1 >= world) {
                c = edges[0];
                for (int64_t e : edges)
                    if (std::fabs((double)e - target) < std::fabs((double)c - target)) c = e;  // first minimum, as Python's min()
            } else {
                c = (int64_t)std::nearbyint(target);  // round half to even, as Python's round()
            }
            cuts.push_back(std::max(c, cuts.back()));
        }
    } else {
        // balanced cumulative weight (1 for a window that runs the network, ~0.02 for a background-skipped one)
        std::vector<double> cum((size_t)n + 1, 0.0);
        for (int64_t i = 0; i < n; ++i) cum[i + 1] = cum[i] + (double)weights[i];
        const double total = cum[n];
        for (int r = 1; r < world; ++r) {
            int64_t c;
            if (total > 0) c = std::lower_bound(cum.begin(), cum.end(), total * r / world) - cum.begin();
            else c = (int64_t)std::nearbyint((double)n * r / world);
            cuts.push_back(std::min(std::max(c, cuts.back()), n));
        }
    }
    cuts.push_back(n);
    for (int r = 0; r < world; ++r) {
        out->win_begin[r] = cuts[r];
        out->win_end[r] = cuts[r + 1];
        if (cuts[r + 1] > cuts[r]) {
            int64_t lo = st[3 * cuts[r]], hi = lo;
            for (int64_t i = cuts[r]; i < cuts[r + 1]; ++i) {
                lo = std::min(lo, st[3 * i]);
                hi = std::max(hi, st[3 * i]);
            }
            out->z_comp_lo[r] = (int)lo;
            out->z_comp_hi[r] = (int)hi + roi_z;
        }
    }
    // ownership: [0, Zp) split at the midpoints of the seams between consecutive non-empty ranks
    int lo = 0;
    for (int r = 0; r < world; ++r) {
        if (out->win_end[r] <= out->win_begin[r]) {
            out->z_own_lo[r] = out->z_own_hi[r] = lo;
            continue;
        }
        int nxt = -1;
        for (int s = r + 1; s < world; ++s)
            if (out->win_end[s] > out->win_begin[s]) {
                nxt = s;
                break;
            }
        int hi = p->Zp;
        if (nxt >= 0) {
            // Python's floor division of a non-negative sum
            hi = (out->z_comp_lo[nxt] + out->z_comp_hi[r]) / 2;
            hi = std::min(std::max(hi, lo), p->Zp);
        }
        out->z_own_lo[r] = lo;
        out->z_own_hi[r] = hi;
        lo = hi;
    }
    return DLV_OK;
    DLV_ABI_GUARD_END(nullptr)
}

int dlv_shard_slab(const dlv_shard_plan* plan, int rank, int Z, int erode_iters, int zblock, int* z0, int* nz) {
    if (!plan || !z0 || !nz || rank < 0 || rank >= plan->world) return DLV_EINVAL;
    int lo = plan->z_comp_lo[rank], hi = plan->z_comp_hi[rank];
    const int olo = plan->z_own_lo[rank], ohi = std::min(plan->z_own_hi[rank], Z);
    if (ohi > olo) {
        // the eroded re-mask of an owned plane looks erode_iters planes up and down, never across a z-block boundary
        const int nb = zblock > 0 ? zblock : Z;
        const int elo = std::max(olo - erode_iters, (olo / nb) * nb);
        const int ehi = std::min(ohi + erode_iters, std::min(((ohi - 1) / nb + 1) * nb, Z));
        if (hi <= lo) {
            lo = elo;
            hi = ehi;
        } else {
            lo = std::min(lo, elo);
            hi = std::max(hi, ehi);
        }
    }
    *z0 = lo;
    *nz = std::max(hi - lo, 0);
    return DLV_OK;
}

// librccl.so for a host that is not a PyTorch process: (1) $DLV_RCCL_PATH, (2) a librccl that is already mapped into the
// process (under torch it lives in torch/lib, which no loader path names), (3) $ROCM_PATH/lib (default /opt/rocm/lib), (4) the
// soname through the loader's own search.  `tried` collects what was attempted for dlv_comm_last_error(NULL).
static thread_local std::string g_comm_init_err;
static int dlv_find_mapped_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(out) = info->dlpi_name;
        return 1;
    }
    return 0;
}
static void* dlv_open_rccl(std::string& tried) {
    tried.clear();
    std::vector<std::string> cand;
    if (const char* e = getenv("DLV_RCCL_PATH")) cand.push_back(e);
    std::string mapped;
    dl_iterate_phdr(dlv_find_mapped_rccl, &mapped);
    if (!mapped.empty()) cand.push_back(mapped);
    const char* rocm = getenv("ROCM_PATH");
    cand.push_back(std::string(rocm && rocm[0] ? rocm : "/opt/rocm") + "/lib/librccl.so");
    cand.push_back("librccl.so");
    cand.push_back("librccl.so.1");
    for (const std::string& p : cand) {
        if (void* h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL)) return h;
        const char* why = dlerror();
        tried += (tried.empty() ? "librccl not loadable; tried: " : "; ") + p + " (" + (why ? why : "?") + ")";
    }
    return nullptr;
}

int dlv_comm_init_all(int n, const int* devs, dlv_comm** out) {
    if (!out || !devs || n < 1 || n > DLV_MAX_RANKS) return DLV_EINVAL;
    *out = nullptr;
    dlv_comm* c = new (std::nothrow) dlv_comm();
    if (!c) return DLV_ENOMEM;
    try {
    c->n = n;
    c->devs.assign(devs, devs + n);
    c->ctx.assign(n, nullptr);
    c->stage.assign(n, nullptr);
    c->stage_bytes.assign(n, 0);
    bool distinct = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devs[i] == devs[j]) distinct = false;
    for (int r = 0; r < n; ++r) {
        const int rc = dlv_ctx_create(devs[r], nullptr, &c->ctx[r]);
        if (rc != DLV_OK) {
            dlv_comm_destroy(c);
            return rc;
        }
    }
    // one rank: no transport is needed - unless DLV_FORCE_RCCL=1 asks for the real thing (a 1-rank communicator: every RCCL
    // entry point of this file - ncclCommInitAll, ncclBroadcast, grouped ncclSend/ncclRecv - then runs on a one-GPU box)
    const char* force = getenv("DLV_FORCE_RCCL");
    if (distinct && (n > 1 || (force && force[0] == '1'))) {
        c->rccl = dlv_open_rccl(g_comm_init_err);
        if (!c->rccl) {
            dlv_comm_destroy(c);
            return DLV_EUNSUP;  // several devices without RCCL: dlv_comm_last_error(NULL) lists the paths that were tried
        }
#define DLV_SYM(field, name) c->field = reinterpret_cast<decltype(c->field)>(dlsym(c->rccl, name))
        DLV_SYM(CommInitAll, "ncclCommInitAll");
        DLV_SYM(CommDestroy, "ncclCommDestroy");
        DLV_SYM(Broadcast, "ncclBroadcast");
        DLV_SYM(Send, "ncclSend");
        DLV_SYM(Recv, "ncclRecv");
        DLV_SYM(GroupStart, "ncclGroupStart");
        DLV_SYM(GroupEnd, "ncclGroupEnd");
        DLV_SYM(GetErrorString, "ncclGetErrorString");
#undef DLV_SYM
        if (!c->CommInitAll || !c->CommDestroy || !c->Broadcast || !c->Send || !c->Recv || !c->GroupStart || !c->GroupEnd) {
            dlv_comm_destroy(c);
            return DLV_EUNSUP;
        }
        c->comm.assign(n, nullptr);
        if (c->CommInitAll(c->comm.data(), n, devs) != ncclSuccess) {
            c->comm.clear();
            dlv_comm_destroy(c);
            return DLV_EHIP;
        }
    }
    } catch (...) {
        dlv_comm_destroy(c);
        return DLV_ENOMEM;
    }
    *out = c;
    return DLV_OK;
}

int dlv_comm_destroy(dlv_comm* c) {
    if (!c) return DLV_EINVAL;
    for (int r = 0; r < (int)c->stage.size(); ++r) {
        if (c->stage[r]) {
            (void)hipSetDevice(c->devs[r]);
            (void)hipFree(c->stage[r]);
        }
    }
    for (auto cm : c->comm)
        if (cm && c->CommDestroy) (void)c->CommDestroy(cm);
    for (auto x : c->ctx)
        if (x) (void)dlv_ctx_destroy(x);
    if (c->rccl) dlclose(c->rccl);
    delete c;
    return DLV_OK;
}

int dlv_comm_size(dlv_comm* c) { return c ? c->n : 0; }
int dlv_comm_uses_rccl(dlv_comm* c) { return (c && !c->comm.empty()) ? 1 : 0; }

int dlv_comm_selftest(dlv_comm* c, size_t bytes) {
    if (!c || bytes == 0 || bytes % 4) return DLV_EINVAL;
    DLV_ABI_GUARD_BEGIN
    const int n = c->n;
    const size_t nw = bytes / 4;
    std::vector<uint32_t*> snd(n, nullptr), rcv(n, nullptr);
    std::vector<uint32_t> host(nw), back(nw);
    struct Free {
        dlv_comm* c;
        std::vector<uint32_t*>&a, &b;
        ~Free() {
            for (int r = 0; r < c->n; ++r) {
                (void)hipSetDevice(c->devs[r]);
                if (a[r]) (void)hipFree(a[r]);
                if (b[r]) (void)hipFree(b[r]);
            }
        }
    } guard{c, snd, rcv};
    for (int r = 0; r < n; ++r) {
        DLV_CHIP(c, hipSetDevice(c->devs[r]));
        DLV_CHIP(c, hipMalloc((void**)&snd[r], bytes));
        DLV_CHIP(c, hipMalloc((void**)&rcv[r], bytes));
        for (size_t i = 0; i < nw; ++i) host[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(r * 0x9e3779b9u);
        DLV_CHIP(c, hipMemcpy(snd[r], host.data(), bytes, hipMemcpyHostToDevice));
        DLV_CHIP(c, hipMemset(rcv[r], 0, bytes));
    }
    const bool rccl = !c->comm.empty();
    // (1) ring exchange rank r -> r+1 (one rank: to itself), grouped like the seam exchange of dlv_sw_infer_sharded
    if (rccl) DLV_NCCL(c, c->GroupStart());
    const int xrc = [&]() -> int {
        for (int r = 0; r < n; ++r) {
            const int to = (r + 1) % n, from = (r + n - 1) % n;
            DLV_CHIP(c, hipSetDevice(c->devs[r]));
            if (rccl) {
                DLV_NCCL(c, c->Send(snd[r], nw, ncclFloat32, to, c->comm[r], c->ctx[r]->main_stream));
                DLV_NCCL(c, c->Recv(rcv[r], nw, ncclFloat32, from, c->comm[r], c->ctx[r]->main_stream));
            } else {
                DLV_CHIP(c, hipMemcpyAsync(rcv[r], snd[from], bytes, hipMemcpyDeviceToDevice, c->ctx[r]->main_stream));
            }
        }
        return DLV_OK;
    }();
    if (rccl) {
        const ncclResult_t ge = c->GroupEnd();
        if (xrc != DLV_OK) return xrc;
        DLV_NCCL(c, ge);
    } else if (xrc != DLV_OK) {
        return xrc;
    }
    for (int r = 0; r < n; ++r) {
        const int from = (r + n - 1) % n;
        DLV_CHIP(c, hipSetDevice(c->devs[r]));
        DLV_CHIP(c, hipStreamSynchronize(c->ctx[r]->main_stream));
        DLV_CHIP(c, hipMemcpy(back.data(), rcv[r], bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nw; ++i)
            if (back[i] != ((uint32_t)(i * 2654435761u) ^ (uint32_t)(from * 0x9e3779b9u)))
                return comm_fail(c, DLV_ESTATE, "self-test: rank %d received a wrong word %zu from rank %d (%s transport)", r, i, from,
                                 rccl ? "RCCL" : "device-copy");
    }
    // (2) broadcast of rank 0's buffer into every rank's receive buffer (ncclUint8, like the weight blob)
    if (rccl) {
        DLV_NCCL(c, c->GroupStart());
        const int brc = [&]() -> int {
            for (int r = 0; r < n; ++r) {
                DLV_CHIP(c, hipSetDevice(c->devs[r]));
                DLV_NCCL(c, c->Broadcast(snd[0], rcv[r], bytes, ncclUint8, 0, c->comm[r], c->ctx[r]->main_stream));
            }
            return DLV_OK;
        }();
        const ncclResult_t ge = c->GroupEnd();
        if (brc != DLV_OK) return brc;
        DLV_NCCL(c, ge);
        for (int r = 0; r < n; ++r) {
            DLV_CHIP(c, hipSetDevice(c->devs[r]));
            DLV_CHIP(c, hipStreamSynchronize(c->ctx[r]->main_stream));
            DLV_CHIP(c, hipMemcpy(back.data(), rcv[r], bytes, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nw; ++i)
                if (back[i] != (uint32_t)(i * 2654435761u))
                    return comm_fail(c, DLV_ESTATE, "self-test: rank %d holds a wrong word %zu after the broadcast", r, i);
        }
    }
    return DLV_OK;
    DLV_ABI_GUARD_END(c)
}
dlv_ctx* dlv_comm_ctx(dlv_comm* c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[rank] : nullptr; }
// c == NULL: why the last dlv_comm_init_all of this thread failed (the RCCL paths that were tried)
const char* dlv_comm_last_error(dlv_comm* c) { return c ? c->err.c_str() : (g_comm_init_err.empty() ? "null comm" : g_comm_init_err.c_str()); }

int dlv_bcast_weights(dlv_comm* c, int root) {
    if (!c || root < 0 || root >= c->n) return DLV_EINVAL;
    DLV_ABI_GUARD_BEGIN
    dlv_ctx* src = c->ctx[root];
    if (!src->weights_loaded || !src->blob) return comm_fail(c, DLV_ESTATE, "dlv_bcast_weights: rank %d has no weights (dlv_unet_load first)", root);
    for (int r = 0; r < c->n; ++r) {
        if (r == root) continue;
        const int rc = dlv_unet_alloc_blob(c->ctx[r], src->features);
        if (rc != DLV_OK) return comm_fail(c, rc, "rank %d: %s", r, dlv_last_error(c->ctx[r]));
        if (c->ctx[r]->blob_bytes != src->blob_bytes) return comm_fail(c, DLV_ESTATE, "blob sizes differ");
    }
    DLV_CHIP(c, hipSetDevice(c->devs[root]));
    DLV_CHIP(c, hipStreamSynchronize(src->main_stream));
    if (!c->comm.empty()) {
        DLV_NCCL(c, c->GroupStart());
        const int rc = [&]() -> int {  // an error inside the group must not leave it open
            for (int r = 0; r < c->n; ++r) {
                DLV_CHIP(c, hipSetDevice(c->devs[r]));
                DLV_NCCL(c, c->Broadcast(src->blob, c->ctx[r]->blob, src->blob_bytes, ncclUint8, root, c->comm[r], c->ctx[r]->main_stream));
            }
            return DLV_OK;
        }();
        const ncclResult_t ge = c->GroupEnd();
        if (rc != DLV_OK) return rc;
        DLV_NCCL(c, ge);
    } else {
        for (int r = 0; r < c->n; ++r)
            if (r != root) {
                DLV_CHIP(c, hipSetDevice(c->devs[r]));
                DLV_CHIP(c, hipMemcpyAsync(c->ctx[r]->blob, src->blob, src->blob_bytes, hipMemcpyDeviceToDevice, c->ctx[r]->main_stream));
            }
    }
    for (int r = 0; r < c->n; ++r) {
        DLV_CHIP(c, hipSetDevice(c->devs[r]));
        DLV_CHIP(c, hipStreamSynchronize(c->ctx[r]->main_stream));
    }
    // the packs that travelled were made with the root's per-block shifts (dlv_unet_set_conv_shift): the receivers normalise with
    // the matching eps
    int shifts[DLV_N_CONV];
    for (int i = 0; i < DLV_N_CONV; ++i) DLV_TRY(dlv_unet_get_conv_shift(src, i, &shifts[i]));
    for (int r = 0; r < c->n; ++r)
        if (r != root) {
            const int rc = dlv_unet_note_conv_shifts(c->ctx[r], shifts);
            if (rc != DLV_OK) return comm_fail(c, rc, "rank %d: %s", r, dlv_last_error(c->ctx[r]));
        }
    return DLV_OK;
    DLV_ABI_GUARD_END(c)
}

// dlv_sw_infer_sharded ended with DLV_ERANGE on some rank: every rank gets the same next block shifts (the largest block index
// named and the largest peaks seen by any rank - what the ranks of a torch.distributed job exchange with an all_reduce MAX,
// inference/inference.py of this package), so the ranks' masks keep composing.
int dlv_comm_range_recover(dlv_comm* c, int* n_changed) {
    if (!c) return DLV_EINVAL;
    DLV_ABI_GUARD_BEGIN
    if (n_changed) *n_changed = 0;
    int layer = -1;
    float peaks[DLV_N_CONV] = {};
    for (int r = 0; r < c->n; ++r) {
        int l = -1;
        float pk[DLV_N_CONV];
        DLV_TRY(dlv_range_report(c->ctx[r], &l, pk));
        layer = std::max(layer, l);
        for (int i = 0; i < DLV_N_CONV; ++i) peaks[i] = std::max(peaks[i], pk[i]);
    }
    if (layer < 0) return comm_fail(c, DLV_ESTATE, "dlv_comm_range_recover: no rank's last 16-bit pass ended with DLV_ERANGE");
    // one decision (rank 0's shifts are everybody's), the same step on every rank: each commits only after its repack succeeded
    int changed = 0;
    for (int r = 0; r < c->n; ++r) {
        int ch = 0;
        const int rc = dlv_range_step(c->ctx[r], layer, peaks, &ch, nullptr);
        if (rc != DLV_OK) return comm_fail(c, rc, "rank %d: %s", r, dlv_last_error(c->ctx[r]));
        if (r == 0) changed = ch;
    }
    if (n_changed) *n_changed = changed;
    return DLV_OK;
    DLV_ABI_GUARD_END(c)
}

int dlv_sw_infer_sharded(dlv_comm* c, const dlv_sw_params* p, const dlv_shard_plan* plan, const int* slab_z0, const int* slab_nz,
                         const uint16_t* const* vol_slab_dev, float* const* acc_slab_dev, uint8_t* const* cnt_slab_dev,
                         dlv_sw_stats* stats) {
    if (!c || !p || !plan || !slab_z0 || !slab_nz || !vol_slab_dev || !acc_slab_dev) return DLV_EINVAL;
    DLV_ABI_GUARD_BEGIN
    if (plan->world != c->n) return comm_fail(c, DLV_EINVAL, "plan for %d ranks, communicator has %d", plan->world, c->n);
    const int n = c->n;
    const size_t plane = (size_t)p->Yp * p->Xp;
    for (int r = 0; r < n; ++r) {
        const bool live = plan->win_end[r] > plan->win_begin[r];
        if (live && (slab_z0[r] > plan->z_comp_lo[r] || slab_z0[r] + slab_nz[r] < plan->z_comp_hi[r]))
            return comm_fail(c, DLV_EINVAL, "rank %d: slab [%d,%d) does not hold its windows' planes [%d,%d)", r, slab_z0[r],
                             slab_z0[r] + slab_nz[r], plan->z_comp_lo[r], plan->z_comp_hi[r]);
        if (plan->z_own_hi[r] > plan->z_own_lo[r] && (slab_z0[r] > plan->z_own_lo[r] || slab_z0[r] + slab_nz[r] < plan->z_own_hi[r]))
            return comm_fail(c, DLV_EINVAL, "rank %d: slab does not hold the planes it owns", r);
    }
    // ---- every rank: its window range on its slab (the launches of a pass are thousands: one host thread per rank) ----
    std::vector<int> rcs(n, DLV_OK);
    {
        std::vector<std::thread> pool;
        pool.reserve(n);
        struct Joiner {  // a failed thread start (std::system_error) must not destroy joinable threads: std::terminate
            std::vector<std::thread>& p;
            ~Joiner() {
                for (auto& t : p)
                    if (t.joinable()) t.join();
            }
        } joiner{pool};
        for (int r = 0; r < n; ++r)
            pool.emplace_back([&, r]() {
                if (stats) memset(&stats[r], 0, sizeof(dlv_sw_stats));
                if (plan->win_end[r] <= plan->win_begin[r]) return;
                dlv_sw_params q = *p;
                q.win_begin = plan->win_begin[r];
                q.win_end = plan->win_end[r];
                q.z0 = slab_z0[r];
                q.nz = slab_nz[r];
                rcs[r] = dlv_sw_infer_dev(c->ctx[r], &q, vol_slab_dev[r], acc_slab_dev[r], cnt_slab_dev ? cnt_slab_dev[r] : nullptr,
                                          stats ? &stats[r] : nullptr);
            });
    }
    for (int r = 0; r < n; ++r)
        if (rcs[r] != DLV_OK) return comm_fail(c, rcs[r], "rank %d: %s", r, dlv_last_error(c->ctx[r]));
    if (n == 1) return DLV_OK;
    // ---- seam exchange: planes computed by src, owned by dst ------------------------------------------------------
    struct Seam {
        int src, dst, lo, hi;
        size_t off;  // byte offset in the owner's staging buffer
    };
    std::vector<Seam> seams;
    std::vector<size_t> need(n, 0);
    const size_t bpv = 4 + (cnt_slab_dev ? 1 : 0);  // staged bytes per voxel: fp32 sum (+ uint8 count)
    for (int dst = 0; dst < n; ++dst)
        for (int src = 0; src < n; ++src) {  // increasing source rank = the order of the additions
            if (src == dst || plan->win_end[src] <= plan->win_begin[src]) continue;
            const int lo = std::max(plan->z_comp_lo[src], plan->z_own_lo[dst]), hi = std::min(plan->z_comp_hi[src], plan->z_own_hi[dst]);
            if (lo >= hi) continue;
            seams.push_back({src, dst, lo, hi, need[dst]});
            need[dst] += (((size_t)(hi - lo) * plane * bpv + 255) & ~(size_t)255);
        }
    std::vector<char*> stg(n, nullptr);
    for (int r = 0; r < n; ++r)
        if (need[r]) DLV_TRY(stage_get(c, r, need[r], (void**)&stg[r]));
    for (int r = 0; r < n; ++r) {  // the sums must be complete before they leave
        DLV_CHIP(c, hipSetDevice(c->devs[r]));
        DLV_CHIP(c, hipStreamSynchronize(c->ctx[r]->main_stream));
    }
    const bool rccl = !c->comm.empty();
    if (rccl) DLV_NCCL(c, c->GroupStart());
    const int xrc = [&]() -> int {  // an error inside the group must not leave it open
    for (const Seam& s : seams) {
        const size_t nvox = (size_t)(s.hi - s.lo) * plane;
        const float* sa = acc_slab_dev[s.src] + (size_t)(s.lo - slab_z0[s.src]) * plane;
        float* da = reinterpret_cast<float*>(stg[s.dst] + s.off);
        const uint8_t* sc = cnt_slab_dev ? cnt_slab_dev[s.src] + (size_t)(s.lo - slab_z0[s.src]) * plane : nullptr;
        uint8_t* dc = reinterpret_cast<uint8_t*>(stg[s.dst] + s.off + nvox * 4);
        if (rccl) {
            DLV_CHIP(c, hipSetDevice(c->devs[s.src]));
            DLV_NCCL(c, c->Send(sa, nvox, ncclFloat32, s.dst, c->comm[s.src], c->ctx[s.src]->main_stream));
            if (sc) DLV_NCCL(c, c->Send(sc, nvox, ncclUint8, s.dst, c->comm[s.src], c->ctx[s.src]->main_stream));
            DLV_CHIP(c, hipSetDevice(c->devs[s.dst]));
            DLV_NCCL(c, c->Recv(da, nvox, ncclFloat32, s.src, c->comm[s.dst], c->ctx[s.dst]->main_stream));
            if (sc) DLV_NCCL(c, c->Recv(dc, nvox, ncclUint8, s.src, c->comm[s.dst], c->ctx[s.dst]->main_stream));
        } else {  // ranks on one device (tests): plain device copies on the owner's stream
            DLV_CHIP(c, hipSetDevice(c->devs[s.dst]));
            DLV_CHIP(c, hipMemcpyAsync(da, sa, nvox * 4, hipMemcpyDeviceToDevice, c->ctx[s.dst]->main_stream));
            if (sc) DLV_CHIP(c, hipMemcpyAsync(dc, sc, nvox, hipMemcpyDeviceToDevice, c->ctx[s.dst]->main_stream));
        }
    }
    return DLV_OK;
    }();
    if (rccl) {
        const ncclResult_t ge = c->GroupEnd();
        if (xrc != DLV_OK) return xrc;
        DLV_NCCL(c, ge);
    } else if (xrc != DLV_OK) {
        return xrc;
    }
    for (const Seam& s : seams) {  // owner adds, in increasing source-rank order per destination (the list is sorted so)
        const size_t nvox = (size_t)(s.hi - s.lo) * plane;
        DLV_CHIP(c, hipSetDevice(c->devs[s.dst]));
        const int grid = (int)std::min<size_t>((nvox + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(add_f32_kernel, dim3(grid), dim3(256), 0, c->ctx[s.dst]->main_stream,
                           acc_slab_dev[s.dst] + (size_t)(s.lo - slab_z0[s.dst]) * plane, reinterpret_cast<const float*>(stg[s.dst] + s.off), nvox);
        if (cnt_slab_dev)
            hipLaunchKernelGGL(add_u8_kernel, dim3(grid), dim3(256), 0, c->ctx[s.dst]->main_stream,
                               cnt_slab_dev[s.dst] + (size_t)(s.lo - slab_z0[s.dst]) * plane,
                               reinterpret_cast<const uint8_t*>(stg[s.dst] + s.off + nvox * 4), nvox);
        if (hipGetLastError() != hipSuccess) return comm_fail(c, DLV_EHIP, "launch of the seam add kernel failed");
    }
    for (int r = 0; r < n; ++r) {
        DLV_CHIP(c, hipSetDevice(c->devs[r]));
        DLV_CHIP(c, hipStreamSynchronize(c->ctx[r]->main_stream));
    }
    return DLV_OK;
    DLV_ABI_GUARD_END(c)
}

}  // extern "C"
