/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors);
 * tests/test_oracle_blend.py checks every mode against the compositing formulas in f64.
 *
 * blend_single, jxl-render/src/blend.rs:550-728: the per-sample arithmetic of frame blending and
 * patches on one channel rectangle.  Alpha planes are host pointers here.
 */
#include <stddef.h>

#include "oracle.h"

static float clamp01(float v) { /* f32::clamp(0.0, 1.0): NaN stays */
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}

void orc_blend_rect(float* base, size_t base_stride, const float* new_grid, size_t new_stride, const JxlGpuBlendRect* r) {
    uint32_t mode = r->mode;
    if (mode == JXLGPU_BLEND_BLEND && !r->new_alpha) mode = JXLGPU_BLEND_REPLACE;  /* blend.rs:565 */
    if (mode == JXLGPU_BLEND_MULADD && !r->new_alpha) mode = JXLGPU_BLEND_ADD;     /* blend.rs:578 */
    for (size_t dy = 0; dy < r->height; ++dy) {
        float* base_row = base + (r->base_y + dy) * base_stride + r->base_x;
        const float* new_row = new_grid + (r->new_y + dy) * new_stride + r->new_x;
        const float* base_alpha_row = r->base_alpha ? r->base_alpha + (r->base_y + dy) * (size_t)r->base_alpha_stride + r->base_x : NULL;
        const float* new_alpha_row = r->new_alpha ? r->new_alpha + (r->new_y + dy) * (size_t)r->new_alpha_stride + r->new_x : NULL;
        for (size_t dx = 0; dx < r->width; ++dx) {
            switch (mode) {
                case JXLGPU_BLEND_REPLACE: base_row[dx] = new_row[dx]; break;
                case JXLGPU_BLEND_ADD: base_row[dx] += new_row[dx]; break;
                case JXLGPU_BLEND_MUL: {
                    float new_sample = new_row[dx];
                    if (r->clamp) new_sample = clamp01(new_sample);
                    base_row[dx] *= new_sample;
                    break;
                }
                case JXLGPU_BLEND_BLEND: {
                    float base_sample, new_sample, base_alpha, new_alpha;
                    if (r->swapped) {
                        base_sample = new_row[dx];
                        new_sample = base_row[dx];
                        base_alpha = new_alpha_row[dx];
                        new_alpha = base_alpha_row ? base_alpha_row[dx] : 0.0f;
                    } else {
                        base_sample = base_row[dx];
                        new_sample = new_row[dx];
                        base_alpha = base_alpha_row ? base_alpha_row[dx] : 0.0f;
                        new_alpha = new_alpha_row[dx];
                    }
                    if (r->clamp) new_alpha = clamp01(new_alpha);
                    if (r->premultiplied) {
                        base_row[dx] = new_sample + base_sample * (1.0f - new_alpha);
                    } else {
                        float base_alpha_rev = 1.0f - base_alpha;
                        float new_alpha_rev = 1.0f - new_alpha;
                        float mixed_alpha = 1.0f - new_alpha_rev * base_alpha_rev;
                        float mixed_alpha_recip = mixed_alpha > 0.0f ? 1.0f / mixed_alpha : 0.0f;
                        base_row[dx] = (new_alpha * new_sample + base_alpha * base_sample * new_alpha_rev) * mixed_alpha_recip;
                    }
                    break;
                }
                case JXLGPU_BLEND_MULADD: {
                    float base_sample, new_sample, new_alpha;
                    if (r->swapped) {
                        base_sample = new_row[dx];
                        new_sample = base_row[dx];
                        new_alpha = base_alpha_row ? base_alpha_row[dx] : 0.0f;
                    } else {
                        base_sample = base_row[dx];
                        new_sample = new_row[dx];
                        new_alpha = new_alpha_row[dx];
                    }
                    if (r->clamp) new_alpha = clamp01(new_alpha);
                    base_row[dx] = base_sample + new_alpha * new_sample;
                    break;
                }
                case JXLGPU_BLEND_MIXALPHA: {
                    float b = base_row[dx], n = new_row[dx];
                    if (r->swapped) { float t = b; b = n; n = t; }
                    if (r->clamp) n = clamp01(n);
                    base_row[dx] = b + n * (1.0f - b);
                    break;
                }
                default: break;
            }
        }
    }
}
