/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors).
 * CPU restatement of jxl-oxide's non-separable upsampling.
 *
 * Follows:
 *   upsample / upsample_inner<K,NW>   jxl-render/src/features/upsampling.rs:6-132
 *   PaddedGrid::mirror_edges_padding  jxl-render/src/util.rs:423-454
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

void orc_upsample_inner(const float* in, size_t in_stride, size_t grid_width, size_t grid_height,
                        float* out, size_t out_stride, int k, const float* weights) {
    const size_t PADDING = 2;
    int log2k = k == 2 ? 1 : k == 4 ? 2 : 3;
    size_t frame_width = grid_width << log2k, frame_height = grid_height << log2k;
    size_t stride = grid_width + PADDING * 2;
    size_t pheight = grid_height + PADDING * 2;
    float* buf = (float*)calloc(stride * pheight, sizeof(float));
    for (size_t y = 0; y < grid_height; ++y)
        memcpy(buf + (y + PADDING) * stride + PADDING, in + y * in_stride, sizeof(float) * grid_width);
    /* util.rs:423-454, literally (including its behaviour for dimensions < 2) */
    for (size_t y = PADDING; y < grid_height + PADDING; ++y)
        for (size_t x = 0; x < PADDING; ++x) {
            buf[y * stride + x] = buf[y * stride + PADDING * 2 - x - 1];
            buf[(y + 1) * stride - x - 1] = buf[(y + 1) * stride - PADDING * 2 + x];
        }
    for (size_t i = 0; i < PADDING; ++i) /* out rows 0..P <- in rows (P..2P) reversed */
        memcpy(buf + i * stride, buf + (PADDING + (PADDING - 1 - i)) * stride, sizeof(float) * stride);
    {
        /* in_chunk = rows [0, height+P), out_chunk = rows [height+P, height+2P); zip(out, in.rev()) */
        size_t in_rows = grid_height + PADDING;
        for (size_t i = 0; i < PADDING; ++i)
            memcpy(buf + (in_rows + i) * stride, buf + (in_rows - 1 - i) * stride, sizeof(float) * stride);
    }

    int mat_n = k / 2;
    float (*wq)[25] = (float (*)[25])calloc((size_t)(k * k / 4), sizeof(float[25]));
    size_t weight_idx = 0;
    for (int y = 0; y < 5 * mat_n; ++y) {
        int mat_y = y / 5, ky = y % 5;
        for (int x = y; x < 5 * mat_n; ++x) {
            int mat_x = x / 5, kx = x % 5;
            float w = weights[weight_idx++];
            wq[mat_y * mat_n + mat_x][ky * 5 + kx] = w;
            wq[mat_x * mat_n + mat_y][kx * 5 + ky] = w;
        }
    }

#pragma omp parallel for schedule(static)
    for (long yy = 0; yy < (long)frame_height; ++yy) {
        size_t y = (size_t)yy;
        size_t ref_y = y / k;
        int ym = (int)(y % k);
        int mat_y = ym < k - ym - 1 ? ym : k - ym - 1;
        int flip_v = ym >= mat_n;
        for (size_t x = 0; x < frame_width; ++x) {
            size_t ref_x = x / k;
            int xm = (int)(x % k);
            int mat_x = xm < k - xm - 1 ? xm : k - xm - 1;
            int flip_h = xm >= mat_n;
            const float* kernel = wq[mat_y * mat_n + mat_x];
            float sum = 0.0f, mn = INFINITY, mx = -INFINITY;
            for (int iy = 0; iy < 5; ++iy) {
                int ky = flip_v ? 4 - iy : iy;
                for (int ix = 0; ix < 5; ++ix) {
                    int kx = flip_h ? 4 - ix : ix;
                    float sample = buf[(ref_y + iy) * stride + (ref_x + ix)];
                    sum += kernel[ky * 5 + kx] * sample;
                    mn = fminf(mn, sample);
                    mx = fmaxf(mx, sample);
                }
            }
            float r;
            if (!isfinite(mn)) r = NAN;
            else { r = sum; if (r < mn) r = mn; if (r > mx) r = mx; }
            out[y * out_stride + x] = r;
        }
    }
    free(wq);
    free(buf);
}
