// Frame compositing primitive: blend_single, jxl-render/src/blend.rs:550-728,
// on device planes.  One launch handles a batch of rectangles that do not overlap in the base
// plane; overlapping rectangles (patches) go to later launches so list order is preserved.
// Arithmetic follows the reference statement by statement (-ffp-contract=off; `recip()` is the
// correctly rounded 1.0f / x).
#include <vector>

#include "common.h"

namespace {

struct BlendBatch {
    float* base;
    const float* new_plane;
    uint32_t base_stride, new_stride;
    const JxlGpuBlendRect* rects;  // device copy
};

__device__ __forceinline__ float clamp01_blend(float v) {
    v = v < 0.0f ? 0.0f : v;
    return v > 1.0f ? 1.0f : v;
}

__global__ __launch_bounds__(256) void blend_kernel(BlendBatch b) {
    const JxlGpuBlendRect r = b.rects[blockIdx.z];
    const uint32_t dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= r.width || dy >= r.height) return;
    uint32_t mode = r.mode;
    if (mode == JXLGPU_BLEND_BLEND && !r.new_alpha) mode = JXLGPU_BLEND_REPLACE;
    if (mode == JXLGPU_BLEND_MULADD && !r.new_alpha) mode = JXLGPU_BLEND_ADD;
    float* bp = b.base + (size_t)(r.base_y + dy) * b.base_stride + r.base_x + dx;
    const float nv = b.new_plane[(size_t)(r.new_y + dy) * b.new_stride + r.new_x + dx];
    const float base_a = r.base_alpha ? r.base_alpha[(size_t)(r.base_y + dy) * r.base_alpha_stride + r.base_x + dx] : 0.0f;
    const float new_a = r.new_alpha ? r.new_alpha[(size_t)(r.new_y + dy) * r.new_alpha_stride + r.new_x + dx] : 0.0f;
    const float bv = *bp;
    float out;
    switch (mode) {
        case JXLGPU_BLEND_REPLACE: out = nv; break;
        case JXLGPU_BLEND_ADD: out = bv + nv; break;
        case JXLGPU_BLEND_MUL: out = bv * (r.clamp ? clamp01_blend(nv) : nv); break;
        case JXLGPU_BLEND_BLEND: {
            float base_sample, new_sample, base_alpha, new_alpha;
            if (r.swapped) { base_sample = nv; new_sample = bv; base_alpha = new_a; new_alpha = base_a; }
            else { base_sample = bv; new_sample = nv; base_alpha = base_a; new_alpha = new_a; }
            if (r.clamp) new_alpha = clamp01_blend(new_alpha);
            if (r.premultiplied) {
                out = new_sample + base_sample * (1.0f - new_alpha);
            } else {
                const float base_alpha_rev = 1.0f - base_alpha;
                const float new_alpha_rev = 1.0f - new_alpha;
                const float mixed_alpha = 1.0f - new_alpha_rev * base_alpha_rev;
                const float mixed_alpha_recip = mixed_alpha > 0.0f ? 1.0f / mixed_alpha : 0.0f;
                out = (new_alpha * new_sample + base_alpha * base_sample * new_alpha_rev) * mixed_alpha_recip;
            }
            break;
        }
        case JXLGPU_BLEND_MULADD: {
            float base_sample, new_sample, new_alpha;
            if (r.swapped) { base_sample = nv; new_sample = bv; new_alpha = base_a; }
            else { base_sample = bv; new_sample = nv; new_alpha = new_a; }
            if (r.clamp) new_alpha = clamp01_blend(new_alpha);
            out = base_sample + new_alpha * new_sample;
            break;
        }
        case JXLGPU_BLEND_MIXALPHA: {
            float bb = bv, nn = nv;
            if (r.swapped) { const float t = bb; bb = nn; nn = t; }
            if (r.clamp) nn = clamp01_blend(nn);
            out = bb + nn * (1.0f - bb);
            break;
        }
        default: return;
    }
    *bp = out;
}

bool overlaps(const JxlGpuBlendRect& a, const JxlGpuBlendRect& b) {
    return a.base_x < b.base_x + b.width && b.base_x < a.base_x + a.width && a.base_y < b.base_y + b.height &&
           b.base_y < a.base_y + a.height;
}

}  // namespace

extern "C" int jxlgpu_blend_rects(jxlgpu_ctx* ctx, float* base, uint32_t base_stride, uint32_t base_w, uint32_t base_h,
                                  const float* new_plane, uint32_t new_stride, uint32_t new_w, uint32_t new_h,
                                  const JxlGpuBlendRect* rects, uint32_t num_rects) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    auto bad = [&](const char* msg) { ctx->last_error = msg; return JXLGPU_ERR_INVALID_ARG; };
    if (!base || !new_plane || (num_rects && !rects)) return bad("null plane / rect list");
    if (base_stride < base_w || new_stride < new_w) return bad("stride < width");
    uint32_t max_w = 0, max_h = 0;
    for (uint32_t i = 0; i < num_rects; ++i) {
        const JxlGpuBlendRect& r = rects[i];
        if (r.mode > JXLGPU_BLEND_SKIP) return bad("unknown blend mode");
        if ((uint64_t)r.base_x + r.width > base_w || (uint64_t)r.base_y + r.height > base_h ||
            (uint64_t)r.new_x + r.width > new_w || (uint64_t)r.new_y + r.height > new_h)
            return bad("blend rectangle outside a plane");
        if (r.new_alpha && r.new_alpha_stride < new_w) return bad("new_alpha_stride < width");
        if (r.base_alpha && r.base_alpha_stride < base_w) return bad("base_alpha_stride < width");
        max_w = std::max(max_w, r.width); max_h = std::max(max_h, r.height);
    }
    if (max_h > 65535u) { ctx->last_error = "blend rectangle taller than 65535 rows"; return JXLGPU_ERR_UNSUPPORTED; }
    if (num_rects == 0 || max_w == 0 || max_h == 0) return JXLGPU_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d_rects = nullptr;
    HIP_TRY(ctx, ctx_dev_malloc(ctx, &d_rects, sizeof(JxlGpuBlendRect) * num_rects));
    hipError_t e = hipMemcpy(d_rects, rects, sizeof(JxlGpuBlendRect) * num_rects, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // batches of rectangles that are pairwise disjoint in the base plane, in list order
        uint32_t first = 0;
        while (first < num_rects) {
            uint32_t last = first + 1;
            uint32_t bw = rects[first].width, bh = rects[first].height;
            while (last < num_rects && last - first < 65535) {
                bool clash = false;
                for (uint32_t k = first; k < last && !clash; ++k) clash = overlaps(rects[k], rects[last]);
                if (clash) break;
                bw = std::max(bw, rects[last].width); bh = std::max(bh, rects[last].height);
                ++last;
            }
            if (bw && bh) {
                BlendBatch b{base, new_plane, base_stride, new_stride, static_cast<const JxlGpuBlendRect*>(d_rects) + first};
                hipLaunchKernelGGL(blend_kernel, dim3((bw + 255) / 256, bh, last - first), dim3(256), 0, ctx->stream, b);
            }
            first = last;
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the rect list is released below
    }
    ctx_dev_release(ctx, d_rects);
    if (e != hipSuccess) {
        ctx->last_error = std::string("jxlgpu_blend_rects: ") + hipGetErrorString(e);
        return JXLGPU_ERR_DEVICE;
    }
    return JXLGPU_OK;
}
