/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors).
 * CPU restatement of jxl-oxide's restoration filters, generic scalar flavour.
 *
 * Follows:
 *   run_gabor_rows_unsafe         jxl-render/src/filter/gabor.rs:59-116
 *   gabor_row_edge                jxl-render/src/filter/impls/generic/gabor.rs:3-86
 *   run_gabor_row_generic         jxl-render/src/filter/impls/generic/gabor.rs:89-168
 *   run_epf_rows                  jxl-render/src/filter/epf.rs:119-261
 *   epf_kernel/dist_offsets       jxl-render/src/filter/epf.rs:263-291
 *   epf_row<STEP>, weight         jxl-render/src/filter/impls/generic/epf.rs:3-210
 *   mirror                        jxl-render/src/util.rs:376-386
 */
#include <math.h>
#include <stdlib.h>

#include "oracle.h"

/* util.rs:376-386 */
static size_t mirror(ptrdiff_t offset, size_t len) {
    for (;;) {
        if (offset < 0) offset = -(offset + 1);
        else if ((size_t)offset >= len) offset = -(offset + 1) + (ptrdiff_t)(len * 2);
        else return (size_t)offset;
    }
}

/* generic/gabor.rs:3-86 */
static void gabor_row_edge(const float* row_c, const float* row_a, float* out, size_t width,
                           const float weights[2]) {
    float w0 = weights[0], w1 = weights[1];
    float global_weight = 1.0f / (1.0f + w0 * 4.0f + w1 * 4.0f);
    if (row_a) {
        if (width == 1) {
            float u = row_a[0], c = row_c[0];
            out[0] = (c * (1.0f + 3.0f * w0 + 2.0f * w1) + u * (w0 + 2.0f * w1)) * global_weight;
            return;
        }
        {
            float a1 = row_a[0], a0 = row_a[1], c1 = row_c[0], c0 = row_c[1];
            out[0] = (c1 * (1.0f + 2.0f * w0 + w1) + (a1 + c0) * (w0 + w1) + a0 * w1) * global_weight;
        }
        for (size_t i = 0; i + 2 < width; ++i) {
            float a0 = row_a[i], a1 = row_a[i + 1], a2 = row_a[i + 2];
            float c0 = row_c[i], c1 = row_c[i + 1], c2 = row_c[i + 2];
            out[i + 1] = (c1 + (a1 + c0 + c1 + c2) * w0 + (a0 + a2 + c0 + c2) * w1) * global_weight;
        }
        {
            float a0 = row_a[width - 2], a1 = row_a[width - 1];
            float c0 = row_c[width - 2], c1 = row_c[width - 1];
            out[width - 1] =
                (c1 * (1.0f + 2.0f * w0 + w1) + (a1 + c0) * (w0 + w1) + a0 * w1) * global_weight;
        }
    } else {
        if (width == 1) {
            out[0] = row_c[0];
            return;
        }
        float merged_w0 = 1.0f + 2.0f + w0;
        float merged_w1 = w0 + 2.0f * w1;
        {
            float c1 = row_c[0], c0 = row_c[1];
            out[0] = (c1 * (merged_w0 + merged_w1) + c0 * merged_w1) * global_weight;
        }
        for (size_t i = 0; i + 2 < width; ++i) {
            float c0 = row_c[i], c1 = row_c[i + 1], c2 = row_c[i + 2];
            out[i + 1] = (c1 * merged_w0 + (c0 + c2) * merged_w1) * global_weight;
        }
        {
            float c0 = row_c[width - 2], c1 = row_c[width - 1];
            out[width - 1] = (c1 * (merged_w0 + merged_w1) + c0 * merged_w1) * global_weight;
        }
    }
}

/* generic/gabor.rs:89-168 */
static void gabor_row_generic(const float* t, const float* c, const float* b, float* out,
                              size_t width, const float weights[2]) {
    if (width == 0) return;
    float w0 = weights[0], w1 = weights[1];
    float global_weight = 1.0f / (1.0f + w0 * 4.0f + w1 * 4.0f);
    if (width == 1) {
        float sum_side = t[0] + 2.0f * c[0] + b[0];
        float sum_diag = 2.0f * (t[0] + b[0]);
        float unweighted_sum = c[0] + sum_side * w0 + sum_diag * w1;
        out[0] = unweighted_sum * global_weight;
        return;
    }
    {
        float t1 = t[0], c1 = c[0], b1 = b[0], t0 = t[1], c0 = c[1], b0 = b[1];
        float sum_side = t1 + c0 + c1 + b1;
        float sum_diag = t0 + t1 + b0 + b1;
        out[0] = (c1 + sum_side * w0 + sum_diag * w1) * global_weight;
    }
    for (size_t i = 0; i + 2 < width; ++i) {
        float sum_side = t[i + 1] + c[i] + c[i + 2] + b[i + 1];
        float sum_diag = t[i] + t[i + 2] + b[i] + b[i + 2];
        out[i + 1] = (c[i + 1] + sum_side * w0 + sum_diag * w1) * global_weight;
    }
    {
        float t1 = t[width - 1], c1 = c[width - 1], b1 = b[width - 1];
        float t0 = t[width - 2], c0 = c[width - 2], b0 = b[width - 2];
        float sum_side = t1 + c0 + c1 + b1;
        float sum_diag = t0 + t1 + b0 + b1;
        out[width - 1] = (c1 + sum_side * w0 + sum_diag * w1) * global_weight;
    }
}

/* gabor.rs:59-116 */
void orc_gabor_plane(const float* in, size_t in_stride, float* out, size_t out_stride, size_t width,
                     size_t height, const float weights[2]) {
    if (height == 1) {
        gabor_row_edge(in, NULL, out, width, weights);
        return;
    }
    gabor_row_edge(in, in + in_stride, out, width, weights);
    /* gabor.rs:86-90: rayon over 8-row stripes */
#pragma omp parallel for schedule(static)
    for (long y8 = 0; y8 < (long)((height - 2 + 7) / 8); ++y8)
        for (size_t dy = 0; dy < 8 && (size_t)y8 * 8 + dy + 2 < height; ++dy) {
            size_t y_up = (size_t)y8 * 8 + dy;
            gabor_row_generic(in + y_up * in_stride, in + (y_up + 1) * in_stride,
                              in + (y_up + 2) * in_stride, out + (y_up + 1) * out_stride, width,
                              weights);
        }
    gabor_row_edge(in + (height - 1) * in_stride, in + (height - 2) * in_stride,
                   out + (height - 1) * out_stride, width, weights);
}

/* epf.rs:263-291 */
static const int KERNEL_1[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
static const int KERNEL_2[12][2] = {{0, -2}, {-1, -1}, {0, -1}, {1, -1}, {-2, 0}, {-1, 0},
                                    {1, 0},  {2, 0},   {-1, 1}, {0, 1},  {1, 1},  {0, 2}};
static const int DIST_0[5][2] = {{0, -1}, {1, 0}, {0, 0}, {-1, 0}, {0, 1}};
static const int DIST_1[5][2] = {{0, -1}, {0, 0}, {0, 1}, {-1, 0}, {1, 0}};
static const int DIST_2[1][2] = {{0, 0}};

/* generic/epf.rs:207-210 */
static float epf_weight(float scaled_distance, float sigma, float step_multiplier) {
    const float FRAC_1_SQRT_2 = 0.70710678118654752440f;
    float neg_inv_sigma = 6.6f * (FRAC_1_SQRT_2 - 1.0f) / sigma * step_multiplier;
    return fmaxf(1.0f + scaled_distance * neg_inv_sigma, 0.0f);
}

/* epf.rs:119-261 + generic/epf.rs:3-204.  The reference's three x-ranges (left edge, inner,
 * right edge) evaluate the same expression; only the inner one skips the (identity) mirror. */
void orc_epf_step(int step, const float* const in[3], size_t in_stride, float* const out[3],
                  size_t out_stride, size_t width, size_t height, const float* sigma,
                  size_t sigma_stride, const JxlGpuFilterParams* fp) {
    const int (*kernel)[2] = step == 0 ? KERNEL_2 : KERNEL_1;
    int nkernel = step == 0 ? 12 : 4;
    const int (*dist)[2] = step == 0 ? DIST_0 : step == 1 ? DIST_1 : DIST_2;
    int ndist = step == 2 ? 1 : 5;
    float step_multiplier = step == 0 ? fp->epf_pass0_sigma_scale
                          : step == 2 ? fp->epf_pass2_sigma_scale : 1.0f;
    float border_sad_mul = fp->epf_border_sad_mul;

    /* epf.rs:155-172: rayon over 8-row stripes */
#pragma omp parallel for schedule(static)
    for (long y8 = 0; y8 < (long)((height + 7) / 8); ++y8)
    for (size_t y = (size_t)y8 * 8; y < (size_t)y8 * 8 + 8 && y < height; ++y) {
        const float* sigma_row = sigma + (y / 8) * sigma_stride;
        const float* rows[3][7];
        for (int c = 0; c < 3; ++c)
            for (int i = 0; i < 7; ++i)
                rows[c][i] = in[c] + mirror((ptrdiff_t)(y + i) - 3, height) * in_stride;

        int is_y_border = ((y + 1) & 6) == 0;
        float sm[8];
        if (is_y_border) {
            for (int i = 0; i < 8; ++i) sm[i] = step_multiplier * border_sad_mul;
        } else {
            for (int i = 0; i < 8; ++i) sm[i] = step_multiplier;
            sm[0] *= border_sad_mul;
            sm[7] *= border_sad_mul;
        }

        for (size_t dx = 0; dx < width; ++dx) {
            float sigma_val = sigma_row[dx / 8];
            if (sigma_val < 0.3f) {
                for (int c = 0; c < 3; ++c) out[c][y * out_stride + dx] = rows[c][3][dx];
                continue;
            }
            float sum_weights = 1.0f;
            float sum_channels[3] = {rows[0][3][dx], rows[1][3][dx], rows[2][3][dx]};
            for (int k = 0; k < nkernel; ++k) {
                int kx = kernel[k][0], ky = kernel[k][1];
                int kernel_dy = 3 + ky;
                ptrdiff_t kernel_dx = (ptrdiff_t)dx + kx;
                float d = 0.0f;
                for (int c = 0; c < 3; ++c) {
                    float scale = fp->epf_channel_scale[c];
                    float acc = 0.0f;
                    for (int i = 0; i < ndist; ++i) {
                        int ix = dist[i][0], iy = dist[i][1];
                        size_t kdx = mirror(kernel_dx + ix, width);
                        size_t bdx = mirror((ptrdiff_t)dx + ix, width);
                        acc += fabsf(rows[c][kernel_dy + iy][kdx] - rows[c][3 + iy][bdx]);
                    }
                    d += scale * acc;
                }
                float w = epf_weight(d, sigma_val, sm[dx & 7]);
                sum_weights += w;
                size_t kdx = mirror(kernel_dx, width);
                for (int c = 0; c < 3; ++c) sum_channels[c] += w * rows[c][kernel_dy][kdx];
            }
            for (int c = 0; c < 3; ++c) out[c][y * out_stride + dx] = sum_channels[c] / sum_weights;
        }
    }
}
