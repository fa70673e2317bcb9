/* delivr_hip_diag.h - test hooks and A/B switches of libdelivr_hip.so.  NOT part of the drop-in boundary (include/delivr_hip.h):
 * nothing here replaces a reference interface; tests/ and profiles/ use these to run one layer in isolation or to select another
 * kernel for the same result.  Nothing in the library is switched through the environment. */
#ifndef DELIVR_HIP_DIAG_H
#define DELIVR_HIP_DIAG_H
#include "delivr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-selection switches of one context (same results up to the 16-bit rounding of one store, other kernels):
 *   "no_zmarch" 1         the generic conv kernel for every layer (and the VALU stem)
 *   "no_upconv" 1         upcat_1 as transposed conv + 64-channel conv instead of the folded form (upconv.hip)
 *   "upconv_simple" 1     the one-tile-per-workgroup upconv kernel for every shape
 *   "fuse_levels" mask    bit l: raw tensors of level l are activated by the z-reg conv that stages them (no normalisation pass)
 *   "fuse_layers" mask    bit li: conv block li activates its first input itself (default 1 << 17: upcat_1.conv_1, the one site that
 *                         pays; 0 = a normalisation pass in front of every conv)
 *   "zreg_mask" mask      1 = Cin 32, 2 = Cin 64 layers may take the register-resident-weights conv (default 3)
 *   "deep_mask" mask      conv_deep.hip: bit 0 = the layers the LDS-weights z-march also takes, bit 1 = the others (default 2)
 *   "generic_ncb" n       cout blocks per workgroup of the generic conv (0: its own choice)
 *   "zreg_dbg" 1          edge-step code on every plane of the z-reg conv
 *   "deep_small" 0        levels smaller than a tile of conv_deep.hip, and its 32-output-channel layers, take the generic conv
 *   "pool_rows_off" 1     the pooling pass by pooled voxels instead of by full lines
 *   "erode_xy_split" 1, "erode_z_two_sweeps" 1, "ccl_simple" 1, "resample_simple" 1, "resample_run16" 1
 *                         the earlier kernels of finalize / CCL / the resamplers (cross-checks in tests/test_gpu_parity.py)
 *   "tiff_chunk" n        planes per pinned staging chunk of dlv_tiff_stack_to_device (0: ~256 MB): the hand-over between the two
 *                         staging buffers on stacks of small planes (tests/test_gpu_pipeline.py)
 * DLV_EINVAL for an unknown name. */
int dlv_diag_set(dlv_ctx* ctx, const char* name, int value);

/* Runs ONE layer of the bf16 MFMA path on fp32 NCDHW device tensors (converted on the device) so
 * that tests/ can compare each kernel with the oracle in isolation.  kind 0: conv block `index`
 * (1..17: Conv3d k3 + InstanceNorm + Mish) on the channel concatenation [in1 (c1), in2 (c2, may be
 * 0)] -> out (B,Cout,D,H,W); kind 1: ConvTranspose3d `index` (0..3) -> out (B,Cout,2D,2H,2W). */
/* Diagnostic library only (libdelivr_hip_diag.so, `make diag`): selects an A/B, stamped or timing-only build of the
 * LDS-weights z-marching conv (3/4/6 tile, stagger and streaming-store variants; 20/24 double-buffered half-planes; 40
 * software-pipelined step; 11-13, 30, 41-45 timing-only or stamped builds, profiles/README.md).  The PRODUCT library holds
 * none of them: it accepts 0 / 50 (default: register-resident-weights conv) and 51 (the LDS-weights kernel for every
 * z-march layer, an A/B that gives the same results), refuses every other value with DLV_EUNSUP and ignores the
 * DLV_ZM_VARIANT environment variable.  No reference counterpart. */
int dlv_debug_set_zm_variant(dlv_ctx* ctx, int variant);
/* diagnostic: buffer (caller-owned, HBM, >= tiles*8*(D+4)*64 bytes, zeroed) that the stamped build of the z-march
 * conv (DLV_ZM_VARIANT=30) fills with s_memtime stamps of window 0; NULL switches it off.  No reference counterpart. */
int dlv_debug_stamps(dlv_ctx* ctx, void* buf_dev);
/* selects the 16-bit format dlv_debug_layer_bf16 runs in (DLV_PREC_BF16 default, DLV_PREC_F16) */
int dlv_debug_set_format(dlv_ctx* ctx, int precision);
int dlv_debug_layer_bf16(dlv_ctx* ctx, int kind, int index, const float* in1_dev, int c1, const float* in2_dev,
                         int c2, float* out_dev, int B, int D, int H, int W);

#ifdef __cplusplus
}
#endif
#endif /* DELIVR_HIP_DIAG_H */
