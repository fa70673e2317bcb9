/* delivr_hip.h - C ABI of libdelivr_hip.so: the MI355X (gfx950) implementation of DELiVR's tiled
 * 3D-U-Net cFos inference path.
 *
 * The reference (erturklab/delivr_cfos) has no FFI layer: its boundary for this path is Python
 * calls + .npy files.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference root).  The Python host package (delivr_cfos_amd/) binds
 * these with ctypes and mirrors the reference's step API on top; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = DLV_OK, negative = DLV_E*; dlv_last_error(ctx) returns a
 *     message owned by the ctx (valid until the next call on that ctx).
 *   - no C++ exception and no HIP error code crosses the ABI.
 *   - pointers are caller-owned.  "_dev" parameters are device (HBM) pointers on the ctx's
 *     device; all others are host pointers.  Sizes are explicit; nothing allocated by the
 *     library is handed to the caller (scratch lives in the ctx and is freed by dlv_ctx_destroy).
 *   - volumes are C-order (Z,Y,X), X contiguous ("z-major slabs"), exactly the payload of the
 *     reference's .npy files (inference/inference.py:234, count_blobs.py:46).
 *   - a ctx is bound to one device and one HIP stream and is NOT thread-safe; distinct ctxs are
 *     independent.  Calls are asynchronous on the ctx stream unless documented "synchronous";
 *     dlv_sync() waits for the stream.
 */
#ifndef DELIVR_HIP_H
#define DELIVR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLV_OK 0
#define DLV_EINVAL (-1)   /* bad argument */
#define DLV_EHIP (-2)     /* a HIP runtime call failed (message has the HIP error string) */
#define DLV_ENOMEM (-3)   /* device allocation failed */
#define DLV_ESTATE (-4)   /* call order violated (e.g. forward before dlv_unet_load) */
#define DLV_EUNSUP (-5)   /* shape not supported by the kernels */
#define DLV_ERANGE (-6)   /* a 16-bit forward produced a non-finite value: the format's range was exceeded (fp16: use bf16) */

/* 2 (round 6): DLV_PREC_BF16 is the mixed format, DLV_PREC_BF16_ALL added; dlv_set_cu_split / dlv_set_conv_algo gone, the range
 * guard entry points (dlv_unet_set_conv_shift ... dlv_comm_range_recover) added and the weight blob re-laid out in round 5; the
 * dlv_debug_* hooks moved to delivr_hip_diag.h.  A host built against version 1 must be rebuilt: dlv_abi_version() tells. */
#define DLV_ABI_VERSION 2

/* compute precision of the U-Net forward */
#define DLV_PREC_F32 0  /* fp32 VALU kernels, NCDHW: parity mode (matches torch fp32 to ~1e-5) */
#define DLV_PREC_BF16 1 /* MFMA implicit-GEMM kernels, fp32 accumulate + fp32 norm statistics: bf16 operands and storage at levels
                           1-4 of the U-Net, IEEE half at level 0 (full resolution), where bf16's 8 significant bits cost the mask
                           its tolerance - the format changes in two normalisation passes */
#define DLV_PREC_F16 2  /* the same kernels on IEEE-half operands everywhere (11 significant bits, same MFMA rate): the default */
#define DLV_PREC_BF16_ALL 3 /* bf16 at every level (8 significant bits, fp32's exponent range): the last resort of the range guard */

#define DLV_N_CONV 18
#define DLV_N_DECONV 4

typedef struct dlv_ctx dlv_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int dlv_abi_version(void);
/* stream == NULL: the ctx creates (and owns) a non-blocking HIP stream; otherwise it adopts the
 * caller's hipStream_t.  Replaces the device set-up at inference/inference.py:153-160. */
int dlv_ctx_create(int device_id, void* stream, dlv_ctx** out);
int dlv_ctx_destroy(dlv_ctx* ctx);
const char* dlv_last_error(dlv_ctx* ctx);
int dlv_sync(dlv_ctx* ctx);
void* dlv_stream(dlv_ctx* ctx); /* the hipStream_t kernels are launched on */

/* plain device-memory helpers so that a host without torch can drive the library */
int dlv_malloc(dlv_ctx* ctx, size_t bytes, void** out_dev);
int dlv_free(dlv_ctx* ctx, void* p_dev);
int dlv_memset_dev(dlv_ctx* ctx, void* p_dev, int value, size_t bytes);
int dlv_copy_h2d(dlv_ctx* ctx, void* dst_dev, const void* src, size_t bytes); /* synchronous */
int dlv_copy_d2h(dlv_ctx* ctx, void* dst, const void* src_dev, size_t bytes); /* synchronous */

/* ---- U-Net weights ------------------------------------------------------------------------ */
/* MONAI BasicUNet(3,1,1,features,act=mish,norm=instance-affine) parameters as fp32 host arrays
 * taken from the checkpoint's state_dict (inference/inference.py:190-200,222).  Conv layers are
 * listed in forward order:
 *   0 conv_0.conv_0   1 conv_0.conv_1   2,3 down_1.convs.conv_{0,1}   4,5 down_2   6,7 down_3
 *   8,9 down_4   10,11 upcat_4.convs   12,13 upcat_3.convs   14,15 upcat_2.convs   16,17 upcat_1.convs
 * conv_w[i]: (Cout,Cin,3,3,3); norm_g/norm_b: InstanceNorm3d affine weight/bias (Cout).
 * deconv_w[j] (j=0..3 = upcat_4..upcat_1): ConvTranspose3d weight (Cin,Cout,2,2,2).
 * final_w: (1,features[5],1,1,1). */
typedef struct dlv_unet_weights {
    int features[6];
    const float* conv_w[DLV_N_CONV];
    const float* conv_b[DLV_N_CONV];
    const float* norm_g[DLV_N_CONV];
    const float* norm_b[DLV_N_CONV];
    const float* deconv_w[DLV_N_DECONV];
    const float* deconv_b[DLV_N_DECONV];
    const float* final_w;
    const float* final_b;
} dlv_unet_weights;

/* copies (and re-packs for the MFMA kernels) the weights into HBM.  Synchronous. */
int dlv_unet_load(dlv_ctx* ctx, const dlv_unet_weights* w);
/* multi-GPU: after dlv_unet_load on the root rank the packed weight blob can be broadcast
 * (RCCL, by the host) instead of re-reading the checkpoint on every rank
 * (replaces DataParallel's per-forward broadcast_coalesced, inference/inference.py:217-219). */
int dlv_unet_blob_size(dlv_ctx* ctx, size_t* bytes);
int dlv_unet_blob_dev(dlv_ctx* ctx, void** blob_dev); /* device pointer owned by ctx */
int dlv_unet_alloc_blob(dlv_ctx* ctx, const int features[6]); /* non-root ranks: allocate only */

/* ---- U-Net forward (one batch of patches) -------------------------------------------------- */
/* logits = BasicUNet(x): x_dev (B,1,d,h,w) fp32 -> logits_dev (B,1,d,h,w) fp32; d,h,w >= 16 and
 * (d/16)*(h/16)*(w/16) > 1 (DLV_EUNSUP otherwise: torch raises there too); multiples of 16 keep every level
 * even, elsewhere MaxPool3d's dropped plane and UpCat's replicate padding as in MONAI.  Replaces predictor(window_data) at inference/sliding_window_inferer.py:222. */
int dlv_unet_forward_dev(dlv_ctx* ctx, const float* x_dev, float* logits_dev, int B, int d, int h, int w,
                         int precision);

/* ---- sliding-window pass ------------------------------------------------------------------- */
typedef struct dlv_sw_params {
    int Zp, Yp, Xp;      /* padded volume shape (inference/inference.py:229-231) */
    int roi[3];          /* window size (Z,Y,X) = config window_dim_0/1/2 (inference.py:162-168) */
    float overlap;       /* 0.5 in the reference (inference.py:127,211) */
    int flip_dim;        /* -1 none, 2 = Z, 3 = Y, 4 = X (sliding_window_inferer.py:218-226) */
    int skip_threshold;  /* window_data_threshold (sliding_window_inferer.py:52,198): a window whose
                            max <= threshold is not inferred and contributes -1000; decided PER
                            WINDOW (= the reference at sw_batch_size 1) */
    int precision;       /* DLV_PREC_* */
    int sw_batch;        /* windows per forward launch (0 = library default) */
    int64_t win_begin;   /* shard: windows [win_begin, win_end) of the reference's enumeration */
    int64_t win_end;     /* (Z slowest, X fastest); win_end <= 0 means "to the end" */
    int z0, nz;          /* the buffers below hold planes [z0, z0+nz) of the padded volume; every
                            window of the shard must lie inside (nz <= 0: whole volume) */
    int repeat;          /* this call stands for `repeat` identical passes (>= 1): acc += repeat*logit,
                            cnt += repeat.  The reference's 13-pass TTA schedule (inference.py:265-279)
                            has only 3 distinct passes once its <=1e-3 noise is dropped (5:4:4) */
    int blend_mode;      /* DLV_BLEND_CONSTANT (0): weights 1 - what the reference does, its mode="gaussian" argument
                            never reaches compute_importance_map (sliding_window_inferer.py:148).
                            DLV_BLEND_GAUSSIAN (1): MONAI 1.2.0's Gaussian importance map (sigma = sigma_scale*roi,
                            truncated at 4 sigma, erf form, normalised to max 1, floored at its smallest non-zero
                            value): acc += w*logit, wsum += w.  Option; no reference output exists for it. */
    float sigma_scale;   /* Gaussian mode: 0.125 in the reference's constructor call (inference.py:206); <= 0 -> 0.125 */
    float* wsum_dev;     /* Gaussian mode: optional fp32 (nz,Yp,Xp) sum of weights (the float count map); cnt_dev must
                            be NULL in that mode */
} dlv_sw_params;
#define DLV_BLEND_CONSTANT 0
#define DLV_BLEND_GAUSSIAN 1

typedef struct dlv_sw_stats {
    int64_t n_windows;   /* windows in the shard */
    int64_t n_skipped;   /* of which background-skipped */
    int64_t n_forward_launches;
} dlv_sw_stats;

/* number of windows the reference enumerates for this geometry (sliding_window_inferer.py:140-145) */
int dlv_sw_num_windows(const dlv_sw_params* p, int64_t* n_windows);
/* writes the (z0,y0,x0) start of every window in the reference's order: starts[3*n] */
int dlv_sw_window_starts(const dlv_sw_params* p, int64_t* starts, int64_t capacity);

/* Per-window maximum of the uint16 volume, in the reference's window order (the quantity the skip test of
 * sliding_window_inferer.py:198 looks at): wmax[n_windows] on the host.  Lets a multi-rank host balance its
 * shards by NON-background windows before calling dlv_sw_infer_dev.  p->win_begin/win_end and p->z0/nz are honoured
 * (wmax[i] = window win_begin + i, from a slab of the volume): the ranks of a sharded run each compute a part.  Synchronous. */
int dlv_sw_window_max_dev(dlv_ctx* ctx, const dlv_sw_params* p, const uint16_t* vol_dev, int32_t* wmax, int64_t capacity);

/* One pass of sliding_window_inference (inference/sliding_window_inferer.py:161-251) with the
 * volume resident in HBM: gather + cast, per-window background skip, optional flip, U-Net forward,
 * un-flip, acc += logit (fp32; the reference accumulates fp16), cnt += 1.
 * vol_dev: uint16 (nz,Yp,Xp); acc_dev: fp32 (nz,Yp,Xp) in/out; cnt_dev: uint8 (nz,Yp,Xp) in/out or
 * NULL.  Synchronous with respect to the host only for the small skip-list read-back. */
int dlv_sw_infer_dev(dlv_ctx* ctx, const dlv_sw_params* p, const uint16_t* vol_dev, float* acc_dev,
                     uint8_t* cnt_dev, dlv_sw_stats* stats);

/* ---- one process, N devices (replaces torch.nn.DataParallel, inference/inference.py:217-219) ------------------------
 * The reference scatters every sw-batch over the visible GPUs, re-broadcasts all parameters per forward and gathers the
 * logits through GPU 0.  Here: a static partition of the reference's window list (Z slowest) into contiguous ranges - a
 * rank's windows form a Z-slab of tile rows -, ONE broadcast of the packed weights, and ONE point-to-point exchange per
 * seam and pass (the planes a rank computed but another rank owns).  Every rank holds only ITS slab of the volume and of
 * the accumulator.  RCCL (ncclBroadcast, grouped ncclSend/ncclRecv over xGMI) is loaded at dlv_comm_init_all; there is
 * no all-reduce on this path.  The multi-process form of the same plan (one process per GPU, torch.distributed) lives in
 * delivr_cfos_amd/parallel.py. */
#define DLV_MAX_RANKS 16
typedef struct dlv_comm dlv_comm;
typedef struct dlv_shard_plan {
    int world;
    int64_t n_windows;
    int64_t win_begin[DLV_MAX_RANKS], win_end[DLV_MAX_RANKS]; /* [begin, end) of the reference's window enumeration */
    int z_comp_lo[DLV_MAX_RANKS], z_comp_hi[DLV_MAX_RANKS];   /* planes [lo, hi) the rank's windows touch ((0,0): none) */
    int z_own_lo[DLV_MAX_RANKS], z_own_hi[DLV_MAX_RANKS];     /* planes [lo, hi) the rank finalizes: a partition of [0, Zp) */
} dlv_shard_plan;
/* Host-only integer logic.  weights == NULL: equal windows, cuts snap to Z tile-row boundaries; otherwise weights[n_windows]
 * (e.g. 1 for a window that runs the network, 0.02 for a background-skipped one, from dlv_sw_window_max_dev) balance the
 * cumulative work.  Ownership: [0, Zp) split at the midpoints of the seams between consecutive non-empty ranks. */
int dlv_shard_plan_make(const dlv_sw_params* p, int world, const float* weights, dlv_shard_plan* out);
/* planes [*z0, *z0 + *nz) a rank must hold: its windows' planes and the planes it owns extended by erode_iters planes
 * inside their z-blocks (zblock <= 0: one block; Z = unpadded stack depth) - what dlv_finalize_slab_dev needs */
int dlv_shard_slab(const dlv_shard_plan* plan, int rank, int Z, int erode_iters, int zblock, int* z0, int* nz);
/* one context per entry of devs (rank r <-> devs[r]) and one RCCL communicator per device (ncclCommInitAll); ranks that
 * share a device (a test on a one-GPU box) exchange with device copies instead.  DLV_EUNSUP: several devices, no librccl. */
int dlv_comm_init_all(int n, const int* devs, dlv_comm** out);
int dlv_comm_destroy(dlv_comm* c);
int dlv_comm_size(dlv_comm* c);
/* 1 when the communicator moves data with RCCL (several distinct devices, or one device with DLV_FORCE_RCCL=1 in the
 * environment of dlv_comm_init_all: a 1-rank ncclCommInitAll), 0 for the device-copy transport of ranks sharing a device */
int dlv_comm_uses_rccl(dlv_comm* c);
/* Transport check before a long job: every rank sends `bytes` (a multiple of 4) to the next rank of the ring - itself when
 * there is one rank - with the grouped ncclSend/ncclRecv the seam exchange uses, then rank 0's buffer is broadcast
 * (ncclBroadcast, like the weight blob); everything received is compared word for word on the host.  DLV_ESTATE + message
 * on a mismatch.  Synchronous.  No reference counterpart (DataParallel has no such check). */
int dlv_comm_selftest(dlv_comm* c, size_t bytes);
dlv_ctx* dlv_comm_ctx(dlv_comm* c, int rank); /* owned by the communicator */
/* c == NULL: why the calling thread's last dlv_comm_init_all failed - for DLV_EUNSUP the librccl paths that were tried
 * ($DLV_RCCL_PATH, a librccl.so already mapped into the process, $ROCM_PATH/lib/librccl.so, then the soname). */
const char* dlv_comm_last_error(dlv_comm* c);
/* dlv_unet_load was called on rank `root`: every other rank allocates the blob and receives it with ONE ncclBroadcast */
int dlv_bcast_weights(dlv_comm* c, int root);
/* dlv_range_recover for every rank of the communicator (see the range guard below): one decision, the same shifts everywhere. */
int dlv_comm_range_recover(dlv_comm* c, int* n_changed /* or NULL */);
/* One sliding-window pass sharded over the communicator.  Rank r's buffers hold planes [slab_z0[r], slab_z0[r]+slab_nz[r])
 * of the padded volume (vol: uint16, acc: fp32 in/out, cnt: uint8 in/out or cnt_slab_dev == NULL), on device devs[r]; they
 * must cover the rank's z_comp and z_own ranges.  p->win_begin/win_end/z0/nz are ignored (taken from the plan).  After the
 * call the planes a rank OWNS hold the complete sums: a neighbour's partial sum is added as one term, in increasing
 * source-rank order - the result is fixed by the plan and does not depend on arrival order; against the single-device
 * pass it is the same terms in another association (fp32 rounding; the count map is exact).  stats: [world] or NULL.
 * Synchronous. */
int dlv_sw_infer_sharded(dlv_comm* c, const dlv_sw_params* p, const dlv_shard_plan* plan, const int* slab_z0, const int* slab_nz,
                         const uint16_t* const* vol_slab_dev, float* const* acc_slab_dev, uint8_t* const* cnt_slab_dev,
                         dlv_sw_stats* stats);

/* ---- finalize: divide, threshold, eroded re-mask ------------------------------------------- */
/* inference/inference.py:285-299 + create_nifti_seg (:31-95).  mean = acc/cnt; cnt_dev == NULL: acc_dev already holds
 * the MEAN logits - or, for threshold == 0.5 ONLY, any positive multiple of them such as the plain sum (the sign decides);
 * a caller with another threshold must pass the count map (the Python mirror enforces it).  fg = sigmoid(mean) >= threshold, keep = raw>0 eroded
 * by an L1 ball of radius erode_iters evaluated inside z-blocks of zblock planes (0 = whole
 * volume; the reference's Arrayterator rule gives floor(floor(1e9/X)/Y)), out = fg & keep.
 * acc/cnt/raw are (.,Yp,Xp)-strided padded buffers, out_dev is the unpadded (Z,Y,X) uint8
 * binaries.npy payload; prob_dev (nullable) receives sigmoid(mean) as fp32 (Z,Y,X)
 * (network_output.npy, inference.py:312-318). */
int dlv_finalize_dev(dlv_ctx* ctx, const float* acc_dev, const uint8_t* cnt_dev, const uint16_t* raw_dev,
                     int Yp, int Xp, int Z, int Y, int X, float threshold, int erode_iters, int zblock,
                     uint8_t* out_dev, float* prob_dev);
/* the same on a Z-slab: the buffers hold planes [z_abs0, z_abs0 + nz) of the stack and the erosion's z-blocks sit at
 * absolute multiples of zblock (the reference's Arrayterator grid), so that a rank of a sharded run reproduces the
 * single-volume result on the planes it owns.  Planes beyond the slab count as foreground: the caller includes
 * erode_iters planes of margin unless the slab ends on a block boundary (dlv_shard_slab).  out/prob: (nz, Y, X). */
int dlv_finalize_slab_dev(dlv_ctx* ctx, const float* acc_dev, const uint8_t* cnt_dev, const uint16_t* raw_dev, int Yp, int Xp,
                          int z_abs0, int nz, int Y, int X, float threshold, int erode_iters, int zblock, uint8_t* out_dev,
                          float* prob_dev);

/* ---- connected components + statistics ------------------------------------------------------ */
/* cc3d.connected_components(bin_img, return_N=True), connectivity 26 (count_blobs.py:61): labels
 * 1..N in C-raster order of each component's first voxel, 0 = background.  labels_dev: uint32
 * (Z,Y,X).  Synchronous (returns N). */
int dlv_ccl26_dev(dlv_ctx* ctx, const uint8_t* mask_dev, int Z, int Y, int X, uint32_t* labels_dev,
                  uint64_t* n_out);
/* cc3d.statistics(labels, no_slice_conversion=True) (count_blobs.py:85): host outputs sized n+1:
 * voxel_counts uint32; bounding_boxes uint16 (n+1,6) = z0,z1,y0,y1,x0,x1 inclusive; centroids
 * float64 (n+1,3) = coordinate sums / count (row 0 = background).  Synchronous. */
int dlv_cc_stats_dev(dlv_ctx* ctx, const uint32_t* labels_dev, int Z, int Y, int X, uint64_t n,
                     uint32_t* voxel_counts, uint16_t* bounding_boxes, double* centroids);

/* The cell table as the reference writes it (count_blobs.py:98-114: a pandas frame of Blob / Coords / Size rows for labels 1..N-1 -
 * `range(1, N)` drops the last label - through DataFrame.to_csv): header ",Blob,Coords,Size", then per label
 * `0,<label>,"[z, y, x]",<voxel count>` with the centroid as Python writes a list of floats (repr: the shortest digits that
 * round-trip, ".0" on integral values, exponent form below 1e-4 and from 1e16).  Host-only (no context, no GPU): 540 k rows take
 * ~0.1 s instead of 0.8 s of Python string formatting.  voxel_counts / centroids: the outputs of dlv_cc_stats_dev.  out / cap:
 * caller-owned text buffer (128 bytes per label are enough); *len_out = bytes written (no terminator); DLV_EINVAL when the buffer
 * is too small. */
int dlv_cells_csv(const uint32_t* voxel_counts, const double* centroids, uint64_t n, char* out, size_t cap, size_t* len_out);

/* Multi-GPU CCL (one process per GPU, every rank labels its own Z-slab with dlv_ccl26_dev): the pieces of the seam
 * merge that run on the device.  The reference has no counterpart - cc3d labels the whole volume on one core
 * (count_blobs.py:61); the contract is that the merged result is identical to that single-volume labelling.
 *  dlv_seam_pairs_dev: plane_a = labels of the LAST plane of a slab, plane_b = labels of the FIRST plane of the slab
 *    below it (both (Y,X) uint32).  Writes every (label_a, label_b) pair that is 26-adjacent across the seam (with
 *    repeats) to pairs_dev (cap pairs of 2 x uint32) and the number of pairs found to *count_out; pairs_dev == NULL
 *    counts only.  Synchronous.
 *  dlv_relabel_u32_dev: labels[i] = lut[labels[i]] for labels[i] != 0 (lut_len = n_local + 1, lut[0] unused).
 *  dlv_cc_stats_raw_dev: the accumulators behind dlv_cc_stats_dev, host arrays of n+1 rows: counts uint32, bbmin /
 *    bbmax uint32 (n+1,3) in z,y,x (0xffffffff / 0 for an absent label), coordinate sums uint64 (n+1,3); row 0
 *    (background) carries its bounding box only.  Synchronous. */
int dlv_seam_pairs_dev(dlv_ctx* ctx, const uint32_t* plane_a_dev, const uint32_t* plane_b_dev, int Y, int X,
                       uint32_t* pairs_dev, uint64_t cap, uint64_t* count_out);
int dlv_relabel_u32_dev(dlv_ctx* ctx, uint32_t* labels_dev, uint64_t nvox, const uint32_t* lut_dev, uint64_t lut_len);
int dlv_cc_stats_raw_dev(dlv_ctx* ctx, const uint32_t* labels_dev, int Z, int Y, int X, uint64_t n, uint32_t* counts,
                         uint32_t* bbmin, uint32_t* bbmax, uint64_t* sums);

/* ---- TIFF z-plane ingest -------------------------------------------------------- */
/* Replaces the per-plane cv2.imread / skimage.io / tifffile reads of the raw stack
 * (downsample/downsample_and_mask.py:25-30, :36-41, :396-404).  Classic TIFF and BigTIFF, II/MM, strips or tiles, 8/16-bit
 * unsigned single channel, compression 1 (none), 5 (LZW) or 8 / 32946 (deflate), predictor 1 or 2; anything else (JPEG,
 * float or multi-sample planes, ...) is refused with DLV_EUNSUP.
 *  dlv_tiff_plane_size / dlv_tiff_read_plane_u16: host-only (no context, no GPU): header fields / one decoded plane
 *    (8-bit samples widened) into a caller-owned (height,width) uint16 host array; dlv_tiff_last_error() holds the
 *    message of the calling thread's last failure.
 *  dlv_tiff_stack_to_device: planes paths[0..n_planes) are decoded by n_threads host threads (0 = one per core, at
 *    most 32) into pinned staging buffers and copied to vol_dev[(i*plane_stride) + y*row_stride + x] (uint16 elements;
 *    strides let the caller fill the zero-padded (Zp,Yp,Xp) network input directly) while the next chunk is being
 *    decoded.  Synchronous. */
const char* dlv_tiff_last_error(void);
int dlv_tiff_plane_size(const char* path, int* height, int* width, int* bits);
int dlv_tiff_read_plane_u16(const char* path, uint16_t* out_host, int height, int width);
/* egress: one (height,width) plane of 8- or 16-bit samples (host, native little-endian) as a classic TIFF with
 * ~64 KB strips, compression 1 (none) or 5 (LZW) - the plane files of blob_highlighter.py:131-133, :160 and
 * cells_to_atlas.py:320 (tifffile.imwrite(..., compression='lzw')).  Host-only. */
int dlv_tiff_write_plane(const char* path, const void* data_host, int height, int width, int bits, int compression);
int dlv_tiff_stack_to_device(dlv_ctx* ctx, const char* const* paths, int n_planes, int height, int width, uint16_t* vol_dev,
                             long long plane_stride, long long row_stride, int n_threads);

/* ---- blob painting (visualisation step) ------------------------------------------ */
/* The colouring loops of blob_highlighter.py:108-125 (RGB) and :150-158 (region id): cells are visited in CSV order
 * and IMG[box] = bin_img[box] * value is assigned over each cell's padded bounding box, so the last listed box that
 * contains a voxel wins.  dlv_paint_owner_dev computes that winner for every voxel: owner (Z,Y,X) uint32 = 1 + index
 * of the last box containing the voxel, 0 = none / background.  boxes (n,6) int32 = z0,z1,y0,y1,x0,x1 HALF-OPEN, i.e.
 * exactly the slices the reference takes after pad_bb (blob_highlighter.py:18-23); the same list is passed on the host
 * (bounds check, launch plan) and in HBM.  dlv_paint_apply_dev: out[v] = bin[v] * values[owner[v]-1] (0 where owner is
 * 0) in uint8 (elem_bytes 1: the R/G/B images) or uint16 (elem_bytes 2: region ids), numpy's wrap-around included. */
int dlv_paint_owner_dev(dlv_ctx* ctx, const uint8_t* bin_dev, int Z, int Y, int X, const int32_t* boxes_dev,
                        const int32_t* boxes_host, uint64_t n_boxes, uint32_t* owner_dev);
int dlv_paint_apply_dev(dlv_ctx* ctx, const uint32_t* owner_dev, const uint8_t* bin_dev, uint64_t nvox,
                        const void* values_dev, int elem_bytes, void* out_dev);

/* Depth-coded blob map (blob_depthmap.py:160-170): scipy.ndimage.distance_transform_edt(np.pad(stack, 1), sampling)
 * [1:-1,1:-1,1:-1].astype(np.uint16) of the down-sampled masked stack - the distance (in the units of `sampling_zyx`, 3
 * doubles on the host) of every non-zero voxel to the nearest zero voxel, the stack being surrounded by zeros; exact
 * (separable lower-envelope algorithm in fp64), truncated to uint16 like numpy's astype.  The painting itself
 * (blob_depthmap.py:186-198: IMG[box] = bin_img[box] * depth_of_the_cell) is dlv_paint_owner_dev + dlv_paint_apply_dev. */
int dlv_edt_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int Z, int Y, int X, const double* sampling_zyx, uint16_t* out_dev);


/* ---- cell-density heat map in atlas space (region assignment step) --------------- */
/* create_heatmap (cells_to_atlas.py:174-200): heat (Z,Y,X) float32 = number of cells per atlas voxel (cells (n,3) int32
 * = x,y,z as in the reference's columns; cells outside the grid are ignored - the reference drops them before, :139-144),
 * then scipy.ndimage.gaussian_filter(heat, sigma) = dlv_gauss_blur_f32_dev with the host-computed half kernel
 * weights[0..radius] (weights[0] = centre; scipy: radius = int(4*sigma + 0.5), phi / phi.sum()), boundary 'reflect',
 * float32 between the axis passes, in place (tmp_dev: scratch of the same size).  Bit-identical to scipy. */
int dlv_heatmap_counts_dev(dlv_ctx* ctx, const int32_t* xyz_dev, uint64_t n_cells, int Z, int Y, int X, float* heat_dev);
int dlv_gauss_blur_f32_dev(dlv_ctx* ctx, float* vol_dev, int Z, int Y, int X, const double* weights_host, int radius,
                           float* tmp_dev);

/* ---- resamplers (the steps either side of the path) ----------------------------------------- */
/* transform.downscale_local_mean(chunk,(fz,fy,fx)).astype(uint16) (downsample_and_mask.py:44):
 * out (ceil(Z/fz),ceil(Y/fy),ceil(X/fx)) = floor(sum over zero-padded block / (fz*fy*fx)). */
int dlv_block_mean_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int Z, int Y, int X, int fz, int fy, int fx,
                           uint16_t* out_dev);
/* scipy.ndimage.zoom(mask, out/in, output=uint8, order=2, prefilter=False)
 * (downsample_and_mask.py:299). */
int dlv_zoom_spline2_u8_dev(dlv_ctx* ctx, const uint8_t* in_dev, int iz, int iy, int ix, uint8_t* out_dev,
                            int oz, int oy, int ox);
/* img *= mask_us[i]; masked_nii[0,0,i,:Y,:X] = img (downsample_and_mask.py:396-417): writes the
 * zero-padded (Zp,Yp,Xp) uint16 volume from raw (Z,Y,X) uint16 and mask (Z,Y,X) uint8 (nullable:
 * threshold mode, values < threshold -> 0). */
int dlv_mask_pad_u16_dev(dlv_ctx* ctx, const uint16_t* raw_dev, const uint8_t* mask_dev, int threshold, int Z,
                         int Y, int X, uint16_t* out_dev, int Zp, int Yp, int Xp);
/* north-star extension (no reference counterpart): trilinear resample, align_corners=False,
 * clamp-to-edge, uint16 -> uint16 (round half up). */
int dlv_trilinear_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int iz, int iy, int ix, uint16_t* out_dev,
                          int oz, int oy, int ox);
/* north-star extension "affine atlas-space warp" (BASELINE config 5).  The reference has no volume warp: it registers with
 * the external mBrainAligner binaries and moves cell COORDINATES (automate_mBrainaligner.py:21-72, :292-435).
 * out[z,y,x] = trilinear sample of `in` at the index-space point matrix34 . (z,y,x,1) (row-major 3x4, host pointer: maps
 * OUTPUT voxel indices to INPUT voxel indices), zero outside the input, round half up.  fp64 arithmetic, bit-exact
 * against oracle.affine_warp_u16.  hostlogic.affine_apply maps cell coordinates found in the warped volume back. */
int dlv_affine_warp_u16_dev(dlv_ctx* ctx, const uint16_t* in_dev, int iz, int iy, int ix, const double* matrix34,
                            uint16_t* out_dev, int oz, int oy, int ox);

/* ---- in-library kernel timing (bench.py's roofline leg) ------------------------------------- */
#define DLV_PROF_MAX_KERNELS 64
typedef struct dlv_prof_entry {
    char name[48];
    int64_t launches;
    double total_ms;     /* sum of HIP-event durations on the ctx stream */
    double flops;        /* algorithmic FLOPs summed over those launches */
    double bytes;        /* algorithmic HBM bytes summed over those launches */
} dlv_prof_entry;
/* fp16 range guard, the remedy.  fp16 ends at 65504; a checkpoint whose RAW (pre-normalisation) output of conv block `layer`
 * exceeds it makes dlv_sw_infer_dev / dlv_unet_forward_dev return DLV_ERANGE.  Every 3x3x3 conv of the network is followed by
 * InstanceNorm, which is invariant to a scale of its input: dlv_unet_set_conv_shift(ctx, layer, k) packs that block's 16-bit
 * weights (and bias) multiplied by 2^-k (exact) and normalises with eps * 4^-k - the stored raw tensor is 2^k times smaller, the
 * normalised value the same, the format stays fp16 (11 significant bits; bf16, the other way out, has 8).  The fp32 path is
 * not affected.  Shifts are reset by dlv_unet_load / dlv_unet_alloc_blob; a rank that received its weights by broadcast sets
 * the same shifts itself.  dlv_range_report: the layer the last DLV_ERANGE named (-1: none; 18: the logits) - its INPUT
 * overflowed, i.e. the block(s) feeding it - and per conv block the largest |mean| + 8 sigma of its raw output (in stored
 * units; 0: the block did not run) seen since the last pass started: above 4096 it is the hint for k, and it tells which blocks
 * are too small to be moved at all.  inference/inference.py's
 * run_inference applies both before it falls back to bf16.  No reference counterpart (the reference network is fp32:
 * inference/sliding_window_inferer.py:205-229). */
int dlv_unet_set_conv_shift(dlv_ctx* ctx, int layer, int shift);
int dlv_unet_get_conv_shift(dlv_ctx* ctx, int layer, int* shift);
int dlv_range_report(dlv_ctx* ctx, int* layer, float* peaks /* [DLV_N_CONV] or NULL */);
/* The policy between the two, for hosts that do not go through run_inference (same rules as delivr_cfos_amd/range_guard.py; a CPU
 * test compares them).  dlv_range_next_shifts is pure host logic: given the layer and peaks of dlv_range_report and the current
 * shifts it writes the next shifts to out[DLV_N_CONV] and returns how many blocks changed (0: nothing left to try; -1: null
 * argument) - the blocks feeding `layer` (MONAI BasicUNet's wiring, inference/inference.py:190-197) move so that |mean| + 8 sigma
 * of their stored output falls to <= 1024, or by 6 bits where no block reported a peak above 4096 (never a block whose peak would
 * fall below 1).  dlv_range_recover applies it to one context after a DLV_ERANGE: DLV_OK = shifts changed (committed only after
 * the weights were re-packed; the report is cleared), repeat the passes (zero the accumulators first); DLV_ERANGE = nothing left,
 * or a 6-bit step on no evidence did not move the overflow - the shifts are back where the sequence started (the caller's last
 * resort is DLV_PREC_BF16_ALL); run_inference gives up after four steps.  dlv_comm_range_recover does the
 * same for every rank of a dlv_comm with the largest layer / peaks any rank saw, so all ranks keep the same shifts. */
int dlv_range_next_shifts(int layer, const float* peaks, const int* shifts, int* out /* [DLV_N_CONV] each */);
int dlv_range_recover(dlv_ctx* ctx, int* n_changed /* or NULL */);
/* The blob this context RECEIVED (dlv_unet_alloc_blob + a broadcast into dlv_unet_blob_dev) holds 16-bit packs made with these
 * shifts: record them - the InstanceNorm eps of a shifted block scales with 4^-shift - without packing again.  dlv_bcast_weights
 * does it for the ranks of a dlv_comm; parallel.broadcast_weights for the ranks of a torch.distributed job. */
int dlv_unet_note_conv_shifts(dlv_ctx* ctx, const int* shifts /* [DLV_N_CONV] */);
/* 1 = run batches back to back on the ctx stream; 2 .. 6 (default 3) = rotate consecutive batches over that many HIP
 * streams so that HBM-bound and MFMA-bound kernels of neighbouring batches overlap (results are identical). */
int dlv_set_lanes(dlv_ctx* ctx, int lanes);
int dlv_prof_enable(dlv_ctx* ctx, int on); /* on: bracket each kernel launch with hipEvents */
int dlv_prof_reset(dlv_ctx* ctx);
int dlv_prof_report(dlv_ctx* ctx, dlv_prof_entry* entries, int capacity, int* n_out); /* synchronous */

/* Allocates now what a pass with these parameters (window shape, precision, batch; dlv_set_lanes) and the finalize of a Z x Y x X
 * stack (0s: none) will ask for later - ~10 GB of activations per pipeline lane, the erosion's distance map.  Fresh device memory
 * is free to allocate, memory that went back to the driver is not (up to seconds for tens of GB in a long-lived process:
 * profiles/r06r_alloc_probe2.json): a host calls this from a second thread while it reads the volume (inference/inference.py
 * does), and keeps one context per device for all its volumes so that nothing is released in between.  Optional: the pass
 * allocates on demand otherwise.  The ctx must not be used by another thread meanwhile.  No reference counterpart (PyTorch's
 * caching allocator plays this role there, inference/inference.py:240-247). */
int dlv_reserve_dev(dlv_ctx* ctx, const dlv_sw_params* p, int Z, int Y, int X);

/* Test hooks and A/B switches (dlv_debug_*, dlv_diag_set) are declared in delivr_hip_diag.h: they are not part of the drop-in
 * boundary.  The library reads these environment variables and no others: DLV_LANES (default of dlv_set_lanes), DLV_LAUNCH_LOG
 * (file that receives one line per kernel launch: profiles/make_traffic.py), and for the transport of dlv_comm_init_all
 * DLV_RCCL_PATH / ROCM_PATH (where librccl.so is looked for) and DLV_FORCE_RCCL (RCCL also for a one-rank communicator). */

#ifdef __cplusplus
}
#endif
#endif /* DELIVR_HIP_H */
