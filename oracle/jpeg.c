/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors);
 * tests/test_oracle_jpeg.py checks the chroma upsampling and YCbCr -> RGB against f64 formulas
 * and the subsampled VarDCT path against the 4:4:4 path on equivalent input.
 *
 * the JPEG-recompression flavour of the VarDCT path.
 *   ChannelShift::from_jpeg_upsampling / shift_size   jxl-modular/src/param.rs:105-165
 *   width_rounded / per-channel buffers               jxl-render/src/vardct/mod.rs:82-96, 206-222
 *   LfCoeff channel shifts                            jxl-vardct/src/lf.rs:154-161
 *   HfMetadata block grid rounded to even             jxl-vardct/src/hf_metadata.rs:70-81
 *   for_each_varblocks (shifted positions)            jxl-render/src/vardct/mod.rs:693-730
 *   dequant_hf_varblock_grouped                       jxl-render/src/vardct/mod.rs:442-542
 *   no CfL when subsampled                            jxl-render/src/vardct/mod.rs:184, 355
 *   transform_with_lf_grouped / transform_varblocks   vardct/mod.rs:605-682, transform_common.rs:11-75
 *   ImageWithRegion::upsample_jpeg                    jxl-render/src/image.rs:448-485
 *   apply_jpeg_upsampling_single                      jxl-render/src/filter/ycbcr.rs:6-89
 *   ycbcr_to_rgb (run_generic)                        jxl-color/src/ycbcr.rs:40-56
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* param.rs:105-140 */
void orc_jpeg_shift(const uint32_t jpeg_upsampling[3], int idx, int* hshift, int* vshift, int* has_h, int* has_v) {
    int hscale = 0, vscale = 0;
    for (int i = 0; i < 3; ++i) {
        hscale |= jpeg_upsampling[i] == 1 || jpeg_upsampling[i] == 2;
        vscale |= jpeg_upsampling[i] == 1 || jpeg_upsampling[i] == 3;
    }
    int h, v;
    switch (jpeg_upsampling[idx]) {
        case 0: h = hscale; v = vscale; break;
        case 1: h = 0; v = 0; break;
        case 2: h = 0; v = vscale; break;
        default: h = hscale; v = 0; break;
    }
    *hshift = h; *vshift = v; *has_h = hscale; *has_v = vscale;
}

/* param.rs:142-165, JpegUpsampling arm */
static size_t shift_size1(size_t n, int has, int sub) {
    if (!has) return n;
    size_t size = (n + 1) / 2;
    return sub ? size : size * 2;
}

/* filter/ycbcr.rs:6-89.  `in`: in_w x in_h samples (the channel's downsampled valid region). */
void orc_upsample_jpeg(const float* in, size_t in_stride, size_t in_w, size_t in_h, int hshift, int vshift,
                       float* out, size_t target_width, size_t target_height) {
    const size_t height = in_h;
    const int h_upsampled = hshift == 0, v_upsampled = vshift == 0;
    for (size_t y = 0; y < height && y < target_height; ++y) {
        const float* row = in + y * in_stride;
        float* out_row = out + y * target_width;
        if (h_upsampled) {
            memcpy(out_row, row, sizeof(float) * target_width);
            continue;
        }
        float last_sample = row[in_w - 1];
        float prev_sample = row[0];
        for (size_t i = 0; i < in_w && 2 * i < target_width; ++i) {
            float curr = row[i];
            float next = i + 1 < in_w ? row[i + 1] : last_sample;
            float left = 0.25f * prev_sample + 0.75f * curr;
            float right = 0.75f * curr + 0.25f * next;
            out_row[2 * i] = left;
            if (2 * i + 1 < target_width) out_row[2 * i + 1] = right;
            prev_sample = curr;
        }
    }
    if (!v_upsampled) {
        float* prev_row = (float*)malloc(sizeof(float) * target_width);
        memcpy(prev_row, out + (height - 1) * target_width, sizeof(float) * target_width);
        for (size_t yy = height; yy-- > 0;) {
            size_t idx_base = yy * target_width;
            size_t top_base = idx_base >= target_width ? idx_base - target_width : 0;
            for (size_t x = 0; x < target_width; ++x) {
                float curr_sample = out[idx_base + x];
                /* interpolating bottom-to-top */
                float bottom = 0.25f * prev_row[x] + 0.75f * curr_sample;
                float top = 0.75f * curr_sample + 0.25f * out[top_base + x];
                out[idx_base * 2 + x] = top;
                if (yy * 2 + 1 < target_height) out[idx_base * 2 + target_width + x] = bottom;
                prev_row[x] = curr_sample;
            }
        }
        free(prev_row);
    }
}

/* jxl-color/src/ycbcr.rs:40-56 */
void orc_ycbcr_to_rgb(float* cb_r, float* y_g, float* cr_b, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        float cb = cb_r[i];
        float y = y_g[i] + 128.0f / 255.0f;
        float cr = cr_b[i];
        cb_r[i] = fmaf(cr, 1.402f, y);
        y_g[i] = fmaf(cb, -0.114f * 1.772f / 0.587f, fmaf(cr, -0.299f * 1.402f / 0.587f, y));
        cr_b[i] = fmaf(cb, 1.772f, y);
    }
}

static float powi_f32(float a, int b) { /* compiler-rt __powisf2 (vardct/mod.rs:458-462) */
    const int recip = b < 0;
    float r = 1.0f;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

/* The VarDCT stage with per-channel shifts, then upsample_jpeg; ends where render.rs:70-72 does:
 * three full-resolution planes (width x height, tight) in `full`. */
int orc_vardct_subsampled(const JxlGpuVardctDesc* d, float* const full[3]) {
    if (!d->skip_adaptive_lf_smoothing) return JXLGPU_ERR_UNSUPPORTED; /* per-channel LF sizes differ */
    if (d->coeff_format != JXLGPU_COEFF_DENSE || d->coeff_sample_type != JXLGPU_SAMPLE_I32) return JXLGPU_ERR_INVALID_ARG;
    const size_t W = d->width, H = d->height, gd = d->group_dim, gd8 = gd / 8;
    int hs[3], vs[3], has_h = 0, has_v = 0;
    for (int c = 0; c < 3; ++c) orc_jpeg_shift(d->jpeg_upsampling, c, &hs[c], &vs[c], &has_h, &has_v);
    const size_t W8 = (W + 7) / 8, H8 = (H + 7) / 8;
    const size_t W8r = has_h ? (W8 + 1) / 2 * 2 : W8, H8r = has_v ? (H8 + 1) / 2 * 2 : H8;
    size_t cw[3], ch[3];
    for (int c = 0; c < 3; ++c) { cw[c] = shift_size1(W8, has_h, hs[c]); ch[c] = shift_size1(H8, has_v, vs[c]); }

    /* frame-level block grid (rounded to even, hf_metadata.rs:70-81) */
    uint8_t* kind = (uint8_t*)malloc(W8r * H8r);
    memset(kind, JXLGPU_BLOCK_UNINIT, W8r * H8r);
    int32_t* hf_mul = (int32_t*)calloc(W8r * H8r, sizeof(int32_t));
    const size_t lf_dim = gd * 8, per_row = (W + lf_dim - 1) / lf_dim;
    uint8_t* has_meta = (uint8_t*)calloc(d->num_lf_groups, 1);
    float* lf[3];
    for (int c = 0; c < 3; ++c) lf[c] = (float*)calloc(cw[c] * ch[c], sizeof(float));
    static const int SRC[3] = {1, 0, 2}; /* util.rs:275-298 */
    for (uint32_t g = 0; g < d->num_lf_groups; ++g) {
        const JxlGpuLfGroup* lg = &d->lf_groups[g];
        const size_t gx = g % per_row, gy = g / per_row;
        const size_t gbw = (lg->width_px + 7) / 8, gbh = (lg->height_px + 7) / 8;
        const size_t bw = has_h ? (gbw + 1) / 2 * 2 : gbw, bh = has_v ? (gbh + 1) / 2 * 2 : gbh;
        for (int c = 0; c < 3; ++c) {
            /* LfCoeff channel sizes: shift_size of the group's cell grid (lf.rs:154-161) */
            const size_t lw = shift_size1(gbw, has_h, hs[c]), lh = shift_size1(gbh, has_v, vs[c]);
            const size_t ox = (gx * gd) >> hs[c], oy = (gy * gd) >> vs[c];
            orc_copy_lf_dequant(lf[c] + oy * cw[c] + ox, cw[c], lg->lf_quant[SRC[c]], d->lf_sample_type, lw, lh,
                                d->m_lf[c], d->global_scale, d->quant_lf, lg->extra_precision);
        }
        if (!lg->has_hf_meta) continue;
        has_meta[g] = 1;
        for (size_t y = 0; y < bh; ++y)
            for (size_t x = 0; x < bw; ++x) {
                size_t o = (gy * gd + y) * W8r + gx * gd + x;
                kind[o] = lg->block_kind[y * bw + x];
                hf_mul[o] = lg->hf_mul[y * bw + x];
            }
    }

    /* per-channel coefficient planes -> f32 (the reference reinterprets the same buffer) */
    float* pix[3];
    size_t stride[3];
    for (int c = 0; c < 3; ++c) {
        stride[c] = cw[c] * 8;
        pix[c] = (float*)malloc(sizeof(float) * stride[c] * ch[c] * 8);
        const size_t src_stride = d->coeff_stride >> hs[c];
        for (size_t y = 0; y < ch[c] * 8; ++y)
            memcpy(pix[c] + y * stride[c], (const int32_t*)d->coeff[c] + y * src_stride, sizeof(float) * stride[c]);
    }

    const float qm_scale[3] = {powi_f32(0.8f, (int)d->x_qm_scale - 2), 1.0f, powi_f32(0.8f, (int)d->b_qm_scale - 2)};
    const size_t groups_x = (W + gd - 1) / gd, groups_y = (H + gd - 1) / gd;
    for (size_t gy = 0; gy < groups_y; ++gy)
        for (size_t gx = 0; gx < groups_x; ++gx) {
            const size_t cx0 = gx * gd8, cy0 = gy * gd8;
            const size_t gcw = W8r - cx0 < gd8 ? W8r - cx0 : gd8, gch = H8r - cy0 < gd8 ? H8r - cy0 : gd8;
            const size_t lfg = (cy0 / gd) * per_row + cx0 / gd;
            for (int c = 0; c < 3; ++c) {
                const size_t ox = cx0 >> hs[c], oy = cy0 >> vs[c]; /* group origin in this channel, cells */
                if (!has_meta[lfg]) {
                    /* vardct/mod.rs:655-665: replicate LF over the channel's part of the group */
                    size_t w = shift_size1(gcw, has_h, hs[c]), h = shift_size1(gch, has_v, vs[c]);
                    for (size_t y = 0; y < h * 8 && oy * 8 + y < ch[c] * 8; ++y)
                        for (size_t x = 0; x < w * 8 && ox * 8 + x < cw[c] * 8; ++x)
                            pix[c][(oy * 8 + y) * stride[c] + ox * 8 + x] = lf[c][(oy + y / 8) * cw[c] + ox + x / 8];
                    continue;
                }
                for (size_t by = 0; by < gch; ++by)
                    for (size_t bx = 0; bx < gcw; ++bx) {
                        /* for_each_varblocks, vardct/mod.rs:693-730 */
                        const size_t cell = (cy0 + by) * W8r + cx0 + bx;
                        const int t = kind[cell];
                        if (t > 26) continue;
                        const size_t sbx = bx >> hs[c], sby = by >> vs[c];
                        if (hs[c] || vs[c]) {
                            if ((sbx << hs[c]) != bx || (sby << vs[c]) != by) continue;
                            if (kind[(cy0 + sby) * W8r + cx0 + sbx] > 26) continue;
                        }
                        int bw, bh;
                        orc_dct_select_size(t, &bw, &bh);
                        const size_t width = (size_t)bw * 8, height = (size_t)bh * 8;
                        if ((ox + sbx) * 8 + width > stride[c] || (oy + sby) * 8 + height > ch[c] * 8) {
                            /* the reference's subgrid() would panic: varblock outside the channel */
                            for (int k = 0; k < 3; ++k) { free(pix[k]); free(lf[k]); }
                            free(kind); free(hf_mul); free(has_meta);
                            return JXLGPU_ERR_INVALID_ARG;
                        }
                        float* blk = pix[c] + (oy + sby) * 8 * stride[c] + (ox + sbx) * 8;
                        /* dequant, vardct/mod.rs:513-537 */
                        const float mul = 65536.0f / ((float)d->global_scale * (float)hf_mul[cell]) * qm_scale[c];
                        const float* matrix = d->dequant[t][c];
                        for (size_t y = 0; y < height; ++y)
                            for (size_t x = 0; x < width; ++x) {
                                float* q = &blk[y * stride[c] + x];
                                int32_t qn;
                                memcpy(&qn, q, 4);
                                float v = (float)qn;
                                if (fabsf(v) <= 1.0f) v *= d->quant_bias[c];
                                else v -= d->quant_bias_numerator / v;
                                v *= matrix[y * width + x];
                                v *= mul;
                                *q = v;
                            }
                        /* transform_common.rs:38-71 */
                        orc_inject_llf(blk, stride[c], lf[c] + (oy + sby) * cw[c] + ox + sbx, cw[c], t);
                        orc_transform_block(blk, stride[c], t);
                    }
            }
        }

    /* image.rs:448-485: every channel to the frame's full resolution */
    for (int c = 0; c < 3; ++c) {
        const size_t in_w = shift_size1(W, has_h, hs[c]), in_h = shift_size1(H, has_v, vs[c]);
        orc_upsample_jpeg(pix[c], stride[c], in_w, in_h, hs[c], vs[c], full[c], W, H);
    }
    for (int c = 0; c < 3; ++c) { free(pix[c]); free(lf[c]); }
    free(kind); free(hf_mul); free(has_meta);
    return 0;
}
