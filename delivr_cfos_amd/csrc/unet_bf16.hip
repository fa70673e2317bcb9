// placeholder - replaced by the MFMA implementation
#include "common.h"
int dlv_pack_weights_bf16(dlv_ctx* ctx) { return DLV_OK; }
int dlv_unet_forward_bf16(dlv_ctx* ctx, const float*, float*, int, int, int, int) {
    return dlv_fail(ctx, DLV_EUNSUP, "bf16 path not built yet");
}
int dlv_unet_tiles_bf16(dlv_ctx* ctx, const uint16_t*, int, int, const int*, int, int, int, int, int, float, float*) {
    return dlv_fail(ctx, DLV_EUNSUP, "bf16 path not built yet");
}
