/*
 * ORACLE — test infrastructure only (see oracle.h).  Stage driver for what follows the
 * transforms in jxl-render/src/render.rs:76-149 and jxl-render/src/lib.rs:925-998:
 *   apply_gabor_like (render.rs:76-101, filter/gabor.rs:8-41)
 *   apply_epf        (render.rs:103-131, filter/epf.rs:10-104: step order and buffer swaps)
 *   upsample_nonseparable (render.rs:149, image.rs:487-557, features/upsampling.rs:6-43)
 *   render_noise     (render.rs:207-222, features/noise.rs)
 *   ColorTransform::run_with_threads (lib.rs:925-998)
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

int orc_post_stages(float* const pix[3], size_t stride, size_t width, size_t height,
                    const float* sigma, size_t sigma_stride, const JxlGpuFilterParams* fp,
                    const JxlGpuUpsampling* up, const JxlGpuNoiseParams* np, size_t group_dim, float corr_x,
                    float corr_b, const JxlGpuColorParams* cp, uint32_t stages, float* const out[3],
                    uint32_t out_stride) {
    size_t n = width * height;
    float* a[3];
    float* b[3];
    for (int c = 0; c < 3; ++c) {
        a[c] = (float*)malloc(sizeof(float) * n);
        b[c] = (float*)malloc(sizeof(float) * n);
#pragma omp parallel for schedule(static)
        for (long y = 0; y < (long)height; ++y) {
            memcpy(a[c] + (size_t)y * width, pix[c] + (size_t)y * stride, sizeof(float) * width);
            memset(b[c] + (size_t)y * width, 0, sizeof(float) * width);  /* first touch in parallel */
        }
    }
#define SWAP_AB() do { for (int c_ = 0; c_ < 3; ++c_) { float* t_ = a[c_]; a[c_] = b[c_]; b[c_] = t_; } } while (0)
    if ((stages & JXLGPU_STAGE_GABOR) && fp->gab_enabled) {
        for (int c = 0; c < 3; ++c)
            orc_gabor_plane(a[c], width, b[c], width, width, height, fp->gab_weights[c]);
        SWAP_AB();
    }
    if ((stages & JXLGPU_STAGE_EPF) && fp->epf_iters > 0) {
        const float* in[3];
        if (fp->epf_iters == 3) {
            in[0] = a[0]; in[1] = a[1]; in[2] = a[2];
            orc_epf_step(0, in, width, b, width, width, height, sigma, sigma_stride, fp);
            SWAP_AB();
        }
        in[0] = a[0]; in[1] = a[1]; in[2] = a[2];
        orc_epf_step(1, in, width, b, width, width, height, sigma, sigma_stride, fp);
        SWAP_AB();
        if (fp->epf_iters >= 2) {
            in[0] = a[0]; in[1] = a[1]; in[2] = a[2];
            orc_epf_step(2, in, width, b, width, width, height, sigma, sigma_stride, fp);
            SWAP_AB();
        }
    }
    size_t ow = width, oh = height;
    if ((stages & JXLGPU_STAGE_UPSAMPLE) && up->factor > 1) {
        /* features/upsampling.rs:18-41: `factor` there is log2; 8x passes first, then 2x or 4x */
        int log2f = up->factor == 2 ? 1 : up->factor == 4 ? 2 : 3;
        int up8 = log2f / 3, last_up = log2f % 3;
        for (int pass = 0; pass < up8 + (last_up ? 1 : 0); ++pass) {
            int k = pass < up8 ? 8 : (last_up == 1 ? 2 : 4);
            const float* w = k == 8 ? up->up8_weight : k == 2 ? up->up2_weight : up->up4_weight;
            for (int c = 0; c < 3; ++c) {
                float* o = (float*)malloc(sizeof(float) * ow * k * oh * k);
                orc_upsample_inner(a[c], ow, ow, oh, o, ow * k, k, w);
                free(a[c]);
                a[c] = o;
            }
            ow *= k; oh *= k;
        }
    }
    int rc = 0;
    if ((stages & JXLGPU_STAGE_NOISE) && np && np->enabled)
        rc = orc_render_noise(a, ow, ow, oh, group_dim, np, corr_x, corr_b);
    if (stages & JXLGPU_STAGE_COLOR) orc_color_transform(a, ow * oh, cp);
    for (int c = 0; c < 3; ++c) {
#pragma omp parallel for schedule(static)
        for (long y = 0; y < (long)oh; ++y)
            memcpy(out[c] + (size_t)y * out_stride, a[c] + (size_t)y * ow, sizeof(float) * ow);
        free(a[c]);
        free(b[c]);
    }
    return rc;
}
