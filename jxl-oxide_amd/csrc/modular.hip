// Modular inverse transforms (placeholder exports until the Squeeze/RCT/Palette kernels land).
#include "common.h"

extern "C" {
int jxlgpu_modular_upload(jxlgpu_ctx* ctx, const JxlGpuModularDesc* desc, jxlgpu_frame** out_frame) {
    if (ctx) ctx->last_error = "modular path not built yet";
    return JXLGPU_ERR_UNSUPPORTED;
}
int jxlgpu_modular_inverse(jxlgpu_ctx* ctx, jxlgpu_frame* frame, void* const* planes_or_null) {
    return JXLGPU_ERR_UNSUPPORTED;
}
int jxlgpu_modular_render(jxlgpu_ctx* ctx, jxlgpu_frame* frame, uint32_t stages, const JxlGpuOut* out) {
    return JXLGPU_ERR_UNSUPPORTED;
}
}
