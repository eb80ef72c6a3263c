// Fused restoration + colour tile kernel (placeholder until the tile kernel lands).
#include "common.h"
#include "pixel_device.h"

bool fused_post_supported(const jxlgpu_frame* f, bool gabor, int epf_iters) { return false; }

void launch_fused_post(hipStream_t s, jxlgpu_frame* f, const float* const in[3], uint32_t in_stride,
                       float* const out[3], uint32_t out_stride, bool gabor, int epf_iters, bool color) {}
