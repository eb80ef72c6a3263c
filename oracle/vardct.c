/*
 * ORACLE — test infrastructure only (see dct.h).  CPU restatement of the VarDCT stage of
 * jxl-oxide's render path, generic scalar flavour.  PARITY UNPINNED: no upstream golden vectors
 * for these functions exist in this container; cross-checked against f64
 * analytic transforms in tests/test_oracle_vardct.py.
 *
 * Follows:
 *   copy_lf_dequant               jxl-render/src/vardct/mod.rs:387-412 (caller util.rs:275-298)
 *   chroma_from_luma_lf           jxl-render/src/vardct/mod.rs:544-568
 *   adaptive_lf_smoothing(_impl)  jxl-render/src/vardct/mod.rs:414-440, generic/mod.rs:11-103
 *   dequant_hf_varblock_grouped   jxl-render/src/vardct/mod.rs:442-542
 *   chroma_from_luma_hf_grouped   jxl-render/src/vardct/mod.rs:570-603
 *   transform_with_lf_grouped     jxl-render/src/vardct/mod.rs:605-682
 *   transform_varblocks_inner     jxl-render/src/vardct/transform_common.rs:11-75
 *   transform_dct2/4/hornuss/4x8/afv   jxl-render/src/vardct/generic/transform.rs:14-240
 *   for_each_varblocks            jxl-render/src/vardct/mod.rs:693-730
 *   TransformType tables          jxl-vardct/src/dct_select.rs:52-151
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/jxlgpu.h"
#include "dct.h"
#include "oracle.h"

/* dct_select.rs:52-76: (bw, bh) in 8x8 cells */
static const uint8_t DCT_SELECT_SIZE[27][2] = {
    {1, 1}, {1, 1}, {1, 1}, {1, 1}, {2, 2}, {4, 4}, {1, 2}, {2, 1}, {1, 4},
    {4, 1}, {2, 4}, {4, 2}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {1, 1},
    {8, 8}, {4, 8}, {8, 4}, {16, 16}, {8, 16}, {16, 8}, {32, 32}, {16, 32}, {32, 16}};

void orc_dct_select_size(int t, int* bw, int* bh) {
    *bw = DCT_SELECT_SIZE[t][0];
    *bh = DCT_SELECT_SIZE[t][1];
}

static int is_1x1_type(int t) {
    return t == JXLGPU_HORNUSS || t == JXLGPU_DCT2 || t == JXLGPU_DCT4 || t == JXLGPU_DCT8X4 ||
           t == JXLGPU_DCT4X8 || t == JXLGPU_DCT8 || (t >= JXLGPU_AFV0 && t <= JXLGPU_AFV3);
}

/* compiler-rt __powisf2, what Rust's f32::powi lowers to (vardct/mod.rs:458-462) */
static float powi_f32(float a, int b) {
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

/* ------------------------------------------------------------------ V1 */
void orc_copy_lf_dequant(float* out, size_t out_stride, const void* quant, uint32_t sample_type,
                         size_t width, size_t height, float m_lf, uint32_t global_scale,
                         uint32_t quant_lf, uint32_t extra_precision) {
    int32_t precision_scale = 1 << (9 - extra_precision);
    uint64_t scale_inv = (uint64_t)global_scale * (uint64_t)quant_lf;
    float scale = (float)((double)m_lf * (double)precision_scale / (double)scale_inv);
    for (size_t y = 0; y < height; ++y) {
        for (size_t x = 0; x < width; ++x) {
            int32_t q = sample_type == JXLGPU_SAMPLE_I16 ? ((const int16_t*)quant)[y * width + x]
                                                          : ((const int32_t*)quant)[y * width + x];
            out[y * out_stride + x] = (float)q * scale;
        }
    }
}

/* ------------------------------------------------------------------ V2 */
void orc_chroma_from_luma_lf(float* x, const float* y, float* b, size_t n, uint32_t colour_factor,
                             float base_x, float base_b, uint32_t x_factor_lf, uint32_t b_factor_lf) {
    int32_t x_factor = (int32_t)x_factor_lf - 128;
    int32_t b_factor = (int32_t)b_factor_lf - 128;
    float kx = base_x + ((float)x_factor / (float)colour_factor);
    float kb = base_b + ((float)b_factor / (float)colour_factor);
    for (size_t i = 0; i < n; ++i) {
        float yy = y[i];
        x[i] += kx * yy;
        b[i] += kb * yy;
    }
}

/* ------------------------------------------------------------------ V3 */
void orc_adaptive_lf_smoothing(size_t width, size_t height, float* in_x, float* in_y, float* in_b,
                               const float m_lf[3], uint32_t global_scale, uint32_t quant_lf) {
    const float SCALE_SELF = 0.052262735f;
    const float SCALE_SIDE = 0.2034514f;
    const float SCALE_DIAG = 0.03348292f;
    uint64_t scale_inv = (uint64_t)global_scale * (uint64_t)quant_lf;
    float lf[3];
    for (int c = 0; c < 3; ++c) lf[c] = (float)(512.0 * (double)m_lf[c] / (double)scale_inv);
    if (width <= 2 || height <= 2) return;

    float* planes[3] = {in_x, in_y, in_b};
    float* udsum[3];
    for (int c = 0; c < 3; ++c) {
        udsum[c] = (float*)malloc(sizeof(float) * width * (height - 2));
        for (size_t y = 0; y + 2 < height; ++y)
            for (size_t x = 0; x < width; ++x)
                udsum[c][y * width + x] = planes[c][y * width + x] + planes[c][(y + 2) * width + x];
    }
    for (size_t y = 1; y + 1 < height; ++y) {
        float* row[3] = {in_x + y * width, in_y + y * width, in_b + y * width};
        const float* ud[3] = {udsum[0] + (y - 1) * width, udsum[1] + (y - 1) * width,
                              udsum[2] + (y - 1) * width};
        float prev[3] = {row[0][0], row[1][0], row[2][0]};
        for (size_t x = 1; x + 1 < width; ++x) {
            float self[3], wa[3], gap_t[3];
            for (int c = 0; c < 3; ++c) {
                self[c] = row[c][x];
                float side = prev[c] + row[c][x + 1] + ud[c][x];
                float diag = ud[c][x - 1] + ud[c][x + 1];
                wa[c] = self[c] * SCALE_SELF + side * SCALE_SIDE + diag * SCALE_DIAG;
                gap_t[c] = fabsf(wa[c] - self[c]) / lf[c];
            }
            float gap = fmaxf(fmaxf(fmaxf(0.5f, gap_t[0]), gap_t[1]), gap_t[2]);
            float gap_scale = fmaxf(3.0f - 4.0f * gap, 0.0f);
            for (int c = 0; c < 3; ++c) {
                row[c][x] = (wa[c] - self[c]) * gap_scale + self[c];
                prev[c] = self[c];
            }
        }
    }
    for (int c = 0; c < 3; ++c) free(udsum[c]);
}

/* ------------------------------------------------------------------ V8: 8x8 special transforms */
#define C(x, y) coeff[(size_t)(y) * stride + (size_t)(x)]

/* generic/transform.rs:26-48 */
static void aux_idct2_in_place(float* coeff, size_t stride, int size) {
    int num_2x2 = size / 2;
    float scratch[8][8];
    for (int y = 0; y < num_2x2; ++y) {
        for (int x = 0; x < num_2x2; ++x) {
            float c00 = C(x, y);
            float c01 = C(x + num_2x2, y);
            float c10 = C(x, y + num_2x2);
            float c11 = C(x + num_2x2, y + num_2x2);
            scratch[2 * y][2 * x] = c00 + c01 + c10 + c11;
            scratch[2 * y][2 * x + 1] = c00 + c01 - c10 - c11;
            scratch[2 * y + 1][2 * x] = c00 - c01 + c10 - c11;
            scratch[2 * y + 1][2 * x + 1] = c00 - c01 - c10 + c11;
        }
    }
    for (int y = 0; y < size; ++y)
        for (int x = 0; x < size; ++x) C(x, y) = scratch[y][x];
}

/* generic/transform.rs:50-54 */
static void transform_dct2(float* coeff, size_t stride) {
    aux_idct2_in_place(coeff, stride, 2);
    aux_idct2_in_place(coeff, stride, 4);
    aux_idct2_in_place(coeff, stride, 8);
}

/* generic/transform.rs:56-82 */
static void transform_dct4(float* coeff, size_t stride) {
    aux_idct2_in_place(coeff, stride, 2);
    float scratch[64];
    for (int y = 0; y < 2; ++y) {
        for (int x = 0; x < 2; ++x) {
            float* s = scratch + (y * 2 + x) * 16;
            for (int iy = 0; iy < 4; ++iy)
                for (int ix = 0; ix < 4; ++ix) s[ix * 4 + iy] = C(x + ix * 2, y + iy * 2);
            orc_dct_2d(s, 4, 4, 4, ORC_INVERSE);
        }
    }
    for (int y = 0; y < 2; ++y)
        for (int x = 0; x < 2; ++x) {
            const float* s = scratch + (y * 2 + x) * 16;
            for (int iy = 0; iy < 4; ++iy)
                for (int ix = 0; ix < 4; ++ix) C(x * 4 + ix, y * 4 + iy) = s[iy * 4 + ix];
        }
}

/* generic/transform.rs:84-116 */
static void transform_hornuss(float* coeff, size_t stride) {
    aux_idct2_in_place(coeff, stride, 2);
    float scratch[64];
    for (int y = 0; y < 2; ++y) {
        for (int x = 0; x < 2; ++x) {
            float* s = scratch + (y * 2 + x) * 16;
            for (int iy = 0; iy < 4; ++iy)
                for (int ix = 0; ix < 4; ++ix) s[iy * 4 + ix] = C(x + ix * 2, y + iy * 2);
            float residual_sum = 0.0f; /* Iterator::sum::<f32>() folds from 0.0 */
            for (int i = 1; i < 16; ++i) residual_sum += s[i];
            float avg = s[0] - residual_sum / 16.0f;
            s[0] = s[5];
            s[5] = 0.0f;
            for (int i = 0; i < 16; ++i) s[i] += avg;
        }
    }
    for (int y = 0; y < 2; ++y)
        for (int x = 0; x < 2; ++x) {
            const float* s = scratch + (y * 2 + x) * 16;
            for (int iy = 0; iy < 4; ++iy)
                for (int ix = 0; ix < 4; ++ix) C(x * 4 + ix, y * 4 + iy) = s[iy * 4 + ix];
        }
}

/* generic/transform.rs:118-146 */
static void transform_dct4x8(float* coeff, size_t stride, int tr) {
    float coeff0 = C(0, 0);
    float coeff1 = C(0, 1);
    C(0, 0) = coeff0 + coeff1;
    C(0, 1) = coeff0 - coeff1;
    float scratch[64];
    for (int idx = 0; idx < 2; ++idx) {
        float* s = scratch + idx * 32;
        for (int iy = 0; iy < 4; ++iy)
            for (int ix = 0; ix < 8; ++ix) s[iy * 8 + ix] = C(ix, iy * 2 + idx);
        orc_dct_2d(s, 8, 8, 4, ORC_INVERSE);
    }
    if (tr) {
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) C(y, x) = scratch[y * 8 + x];
    } else {
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) C(x, y) = scratch[y * 8 + x];
    }
}

#include "afv_basis.inc"

/* generic/transform.rs:148-219 */
static void transform_afv(float* coeff, size_t stride, int n) {
    int flip_x = n % 2;
    int flip_y = n / 2;
    float coeff_afv[16];
    coeff_afv[0] = (C(0, 0) + C(1, 0) + C(0, 1)) * 4.0f;
    for (int idx = 1; idx < 16; ++idx) {
        int iy = idx / 4, ix = idx % 4;
        coeff_afv[idx] = C(2 * ix, 2 * iy);
    }
    float samples_afv[16];
    memset(samples_afv, 0, sizeof samples_afv);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            samples_afv[j] = fmaf(coeff_afv[i], ORC_AFV_BASIS[i][j], samples_afv[j]);

    float scratch_4x4[16];
    float scratch_4x8[32];
    memset(scratch_4x4, 0, sizeof scratch_4x4);
    memset(scratch_4x8, 0, sizeof scratch_4x8);
    scratch_4x4[0] = C(0, 0) - C(1, 0) + C(0, 1);
    for (int iy = 0; iy < 4; ++iy)
        for (int ix = 0; ix < 4; ++ix) {
            if ((ix | iy) == 0) continue;
            scratch_4x4[ix * 4 + iy] = C(2 * ix + 1, 2 * iy);
        }
    orc_dct_2d(scratch_4x4, 4, 4, 4, ORC_INVERSE);

    scratch_4x8[0] = C(0, 0) - C(0, 1);
    for (int iy = 0; iy < 4; ++iy)
        for (int ix = 0; ix < 8; ++ix) {
            if ((ix | iy) == 0) continue;
            scratch_4x8[iy * 8 + ix] = C(ix, 2 * iy + 1);
        }
    orc_dct_2d(scratch_4x8, 8, 8, 4, ORC_INVERSE);

    for (int iy = 0; iy < 4; ++iy) {
        int afv_y = flip_y == 0 ? iy : 3 - iy;
        for (int ix = 0; ix < 4; ++ix) {
            int afv_x = flip_x == 0 ? ix : 3 - ix;
            C(flip_x * 4 + ix, flip_y * 4 + iy) = samples_afv[afv_y * 4 + afv_x];
        }
    }
    for (int iy = 0; iy < 4; ++iy) {
        int y = flip_y * 4 + iy;
        for (int ix = 0; ix < 4; ++ix) {
            int x = (1 - flip_x) * 4 + ix;
            C(x, y) = scratch_4x4[iy * 4 + ix];
        }
    }
    for (int iy = 0; iy < 4; ++iy) {
        int y = (1 - flip_y) * 4 + iy;
        for (int ix = 0; ix < 8; ++ix) C(ix, y) = scratch_4x8[iy * 8 + ix];
    }
}

/* generic/transform.rs:225-240 */
void orc_transform_block(float* coeff, size_t stride, int dct_select) {
    int bw, bh;
    orc_dct_select_size(dct_select, &bw, &bh);
    switch (dct_select) {
        case JXLGPU_DCT2: transform_dct2(coeff, stride); break;
        case JXLGPU_DCT4: transform_dct4(coeff, stride); break;
        case JXLGPU_HORNUSS: transform_hornuss(coeff, stride); break;
        case JXLGPU_DCT4X8: transform_dct4x8(coeff, stride, 0); break;
        case JXLGPU_DCT8X4: transform_dct4x8(coeff, stride, 1); break;
        case JXLGPU_AFV0: transform_afv(coeff, stride, 0); break;
        case JXLGPU_AFV1: transform_afv(coeff, stride, 1); break;
        case JXLGPU_AFV2: transform_afv(coeff, stride, 2); break;
        case JXLGPU_AFV3: transform_afv(coeff, stride, 3); break;
        default: orc_dct_2d(coeff, stride, (size_t)bw * 8, (size_t)bh * 8, ORC_INVERSE); break;
    }
}
#undef C

/* transform_common.rs:40-66: LF -> lowest-frequency coefficients of one varblock */
void orc_inject_llf(float* coeff, size_t stride, const float* lf, size_t lf_stride, int dct_select) {
    int bw, bh;
    orc_dct_select_size(dct_select, &bw, &bh);
    if (is_1x1_type(dct_select)) {
        coeff[0] = lf[0];
        return;
    }
    int logbw = __builtin_ctz(bw), logbh = __builtin_ctz(bh);
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) coeff[y * stride + x] = lf[y * lf_stride + x];
    orc_dct_2d(coeff, stride, bw, bh, ORC_FORWARD);
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x)
            coeff[y * stride + x] /= orc_scale_f(y, 5 - logbh) * orc_scale_f(x, 5 - logbw);
}

/* ------------------------------------------------------------------ frame-level driver */
typedef struct {
    size_t w8, h8;       /* cells                        */
    uint8_t* kind;       /* frame-level BlockInfo plane  */
    int32_t* hf_mul;
    float* sigma;
    size_t w64, h64;
    int32_t* xfy;
    int32_t* bfy;
    uint8_t* has_meta;   /* per LF group */
    size_t lf_groups_per_row;
} FrameMeta;

static void build_frame_meta(const JxlGpuVardctDesc* d, FrameMeta* m) {
    size_t w8 = (d->width + 7) / 8, h8 = (d->height + 7) / 8;
    size_t w64 = (d->width + 63) / 64, h64 = (d->height + 63) / 64;
    m->w8 = w8; m->h8 = h8; m->w64 = w64; m->h64 = h64;
    m->kind = (uint8_t*)malloc(w8 * h8);
    memset(m->kind, JXLGPU_BLOCK_UNINIT, w8 * h8);
    m->hf_mul = (int32_t*)calloc(w8 * h8, sizeof(int32_t));
    m->sigma = (float*)malloc(sizeof(float) * w8 * h8);
    for (size_t i = 0; i < w8 * h8; ++i) m->sigma[i] = d->filter.epf_sigma_for_modular;
    m->xfy = (int32_t*)calloc(w64 * h64, sizeof(int32_t));
    m->bfy = (int32_t*)calloc(w64 * h64, sizeof(int32_t));
    size_t lf_dim = (size_t)d->group_dim * 8;
    size_t per_row = (d->width + lf_dim - 1) / lf_dim;
    m->lf_groups_per_row = per_row;
    m->has_meta = (uint8_t*)calloc(d->num_lf_groups, 1);
    for (uint32_t g = 0; g < d->num_lf_groups; ++g) {
        const JxlGpuLfGroup* lg = &d->lf_groups[g];
        if (!lg->has_hf_meta) continue;
        m->has_meta[g] = 1;
        size_t gx = g % per_row, gy = g / per_row;
        size_t bw = (lg->width_px + 7) / 8, bh = (lg->height_px + 7) / 8;
        size_t cw = (lg->width_px + 63) / 64, ch = (lg->height_px + 63) / 64;
        size_t cell0x = gx * d->group_dim, cell0y = gy * d->group_dim;
        /* hf_metadata.rs:70-81: the group's grids are rounded to even cell counts when subsampled */
        size_t gstride = bw;
        {
            int hs_, vs_, has_h = 0, has_v = 0;
            orc_jpeg_shift(d->jpeg_upsampling, 0, &hs_, &vs_, &has_h, &has_v);
            if (has_h) gstride = (bw + 1) / 2 * 2;
        }
        for (size_t y = 0; y < bh; ++y)
            for (size_t x = 0; x < bw; ++x) {
                size_t o = (cell0y + y) * w8 + cell0x + x;
                m->kind[o] = lg->block_kind[y * gstride + x];
                m->hf_mul[o] = lg->hf_mul[y * gstride + x];
                if (lg->epf_sigma) m->sigma[o] = lg->epf_sigma[y * gstride + x];
            }
        size_t t0x = gx * (lf_dim / 64), t0y = gy * (lf_dim / 64);
        for (size_t y = 0; y < ch; ++y)
            for (size_t x = 0; x < cw; ++x) {
                m->xfy[(t0y + y) * w64 + t0x + x] = lg->x_from_y[y * cw + x];
                m->bfy[(t0y + y) * w64 + t0x + x] = lg->b_from_y[y * cw + x];
            }
    }
}

static void free_frame_meta(FrameMeta* m) {
    free(m->kind); free(m->hf_mul); free(m->sigma); free(m->xfy); free(m->bfy); free(m->has_meta);
}

/* V1-V3 on the whole frame: out planes lf[3] are w8 x h8, order X, Y, B. */
void orc_vardct_lf(const JxlGpuVardctDesc* d, float* const lf[3]) {
    size_t w8 = (d->width + 7) / 8, h8 = (d->height + 7) / 8;
    size_t lf_dim = (size_t)d->group_dim * 8;
    size_t per_row = (d->width + lf_dim - 1) / lf_dim;
    if (d->lf_frame[0]) {
        /* vardct/mod.rs:175-179: `lf_frame` given -> the blended LF frame IS lf_xyb; CfL-LF and the
         * adaptive smoothing belong to the other branch of that `if` */
        for (int c = 0; c < 3; ++c)
            for (size_t y = 0; y < h8; ++y)
                memcpy(lf[c] + y * w8, d->lf_frame[c] + y * d->lf_frame_stride, sizeof(float) * w8);
        return;
    }
    for (uint32_t g = 0; g < d->num_lf_groups; ++g) {
        const JxlGpuLfGroup* lg = &d->lf_groups[g];
        size_t gx = g % per_row, gy = g / per_row;
        size_t bw = (lg->width_px + 7) / 8, bh = (lg->height_px + 7) / 8;
        size_t o = gy * d->group_dim * w8 + gx * d->group_dim;
        /* util.rs:275-298: lf_x <- channel 1, lf_y <- channel 0, lf_b <- channel 2 */
        static const int SRC[3] = {1, 0, 2};
        for (int c = 0; c < 3; ++c)
            orc_copy_lf_dequant(lf[c] + o, w8, lg->lf_quant[SRC[c]], d->lf_sample_type, bw, bh,
                                d->m_lf[c], d->global_scale, d->quant_lf, lg->extra_precision);
    }
    orc_chroma_from_luma_lf(lf[0], lf[1], lf[2], w8 * h8, d->colour_factor, d->base_correlation_x,
                            d->base_correlation_b, d->x_factor_lf, d->b_factor_lf);
    if (!d->skip_adaptive_lf_smoothing)
        orc_adaptive_lf_smoothing(w8, h8, lf[0], lf[1], lf[2], d->m_lf, d->global_scale, d->quant_lf);
}

/* V4-V8 for one 256x256 group, in place on f32 planes that initially hold the i32 coefficient
 * bits (the reference reinterprets the same buffer, vardct/mod.rs:262-265, 528-529). */
static void transform_group(const JxlGpuVardctDesc* d, const FrameMeta* m, float* const pix[3],
                            size_t stride, const float* const lf[3], size_t gx, size_t gy) {
    const size_t gd8 = d->group_dim / 8;
    size_t cx0 = gx * gd8, cy0 = gy * gd8;
    size_t cw = m->w8 - cx0 < gd8 ? m->w8 - cx0 : gd8;
    size_t ch = m->h8 - cy0 < gd8 ? m->h8 - cy0 : gd8;
    size_t lf_dim_cells = (size_t)d->group_dim; /* cells per LF group side */
    size_t lfg = (cy0 / lf_dim_cells) * m->lf_groups_per_row + cx0 / lf_dim_cells;
    int has_meta = m->has_meta[lfg];

    if (!has_meta) {
        /* vardct/mod.rs:655-665: no HfMetadata -> replicate LF */
        for (int c = 0; c < 3; ++c)
            for (size_t y = 0; y < ch * 8; ++y)
                for (size_t x = 0; x < cw * 8; ++x)
                    pix[c][(cy0 * 8 + y) * stride + cx0 * 8 + x] =
                        lf[c][(cy0 + y / 8) * m->w8 + cx0 + x / 8];
        return;
    }

    const float qm_scale[3] = {powi_f32(0.8f, (int)d->x_qm_scale - 2), 1.0f,
                               powi_f32(0.8f, (int)d->b_qm_scale - 2)};
    /* V4 dequant */
    for (int c = 0; c < 3; ++c) {
        float quant_bias = d->quant_bias[c];
        for (size_t by = 0; by < ch; ++by)
            for (size_t bx = 0; bx < cw; ++bx) {
                size_t cell = (cy0 + by) * m->w8 + cx0 + bx;
                int t = m->kind[cell];
                if (t > 26) continue;
                int bw, bh;
                orc_dct_select_size(t, &bw, &bh);
                size_t width = (size_t)bw * 8, height = (size_t)bh * 8;
                float mul = 65536.0f / ((float)d->global_scale * (float)m->hf_mul[cell]) * qm_scale[c];
                const float* matrix = d->dequant[t][c];
                float* blk = pix[c] + (cy0 + by) * 8 * stride + (cx0 + bx) * 8;
                for (size_t y = 0; y < height; ++y)
                    for (size_t x = 0; x < width; ++x) {
                        float* q = &blk[y * stride + x];
                        int32_t qn;
                        memcpy(&qn, q, 4);
                        float v = (float)qn;
                        if (fabsf(v) <= 1.0f) v *= quant_bias;
                        else v -= d->quant_bias_numerator / v;
                        v *= matrix[y * width + x];
                        v *= mul;
                        *q = v;
                    }
            }
    }
    /* V5 CfL on HF */
    {
        size_t gw = cw * 8, gh = ch * 8;
        for (size_t y = 0; y < gh; ++y) {
            size_t ty = (cy0 * 8 + y) / 64;
            float* row_x = pix[0] + (cy0 * 8 + y) * stride + cx0 * 8;
            float* row_y = pix[1] + (cy0 * 8 + y) * stride + cx0 * 8;
            float* row_b = pix[2] + (cy0 * 8 + y) * stride + cx0 * 8;
            for (size_t x64 = 0; x64 * 64 < gw; ++x64) {
                size_t tx = (cx0 * 8) / 64 + x64;
                float kx = d->base_correlation_x + ((float)m->xfy[ty * m->w64 + tx] / (float)d->colour_factor);
                float kb = d->base_correlation_b + ((float)m->bfy[ty * m->w64 + tx] / (float)d->colour_factor);
                size_t n = gw - x64 * 64 < 64 ? gw - x64 * 64 : 64;
                for (size_t dx = 0; dx < n; ++dx) {
                    size_t x = x64 * 64 + dx;
                    float cy = row_y[x];
                    row_x[x] += kx * cy;
                    row_b[x] += kb * cy;
                }
            }
        }
    }
    /* V6-V8 */
    for (int c = 0; c < 3; ++c)
        for (size_t by = 0; by < ch; ++by)
            for (size_t bx = 0; bx < cw; ++bx) {
                size_t cell = (cy0 + by) * m->w8 + cx0 + bx;
                int t = m->kind[cell];
                if (t > 26) continue;
                float* blk = pix[c] + (cy0 + by) * 8 * stride + (cx0 + bx) * 8;
                orc_inject_llf(blk, stride, lf[c] + cell, m->w8, t);
                orc_transform_block(blk, stride, t);
            }
}

/* Whole VarDCT render, stages as in jxlgpu.h.  out planes: stride `out_stride`.
 * lf_out (optional) receives the LF image (w8 x h8).  Returns 0 or a JXLGPU_ERR code. */
int jxl_oracle_vardct_render(const JxlGpuVardctDesc* d, uint32_t stages, float* const out[3],
                             uint32_t out_stride, float* const lf_out[3]) {
    if (d->abi != JXLGPU_ABI_VERSION) return JXLGPU_ERR_ABI;
    if (d->jpeg_upsampling[0] | d->jpeg_upsampling[1] | d->jpeg_upsampling[2]) {
        /* chroma-subsampled frame (jpeg.c): per-channel geometry up to upsample_jpeg, then the
         * common tail of render.rs:70-156 on full-resolution planes */
        if (lf_out && lf_out[0]) return JXLGPU_ERR_INVALID_ARG;
        size_t W = d->width, H = d->height, w8s = (W + 7) / 8;
        for (int n = 64, i = 0; n <= 256; n *= 2, ++i)
            if (d->sec_half_large[i]) orc_set_sec_half_large(n, d->sec_half_large[i]);
        float* full[3];
        for (int c = 0; c < 3; ++c) full[c] = (float*)calloc(W * H, sizeof(float));
        int rc = orc_vardct_subsampled(d, full);
        if (rc == 0) {
            FrameMeta m;
            build_frame_meta(d, &m);
            rc = orc_post_stages(full, W, W, H, m.sigma, w8s, &d->filter, &d->upsampling, &d->noise, d->group_dim,
                                 d->base_correlation_x, d->base_correlation_b, &d->color, stages, out, out_stride);
            free_frame_meta(&m);
        }
        for (int c = 0; c < 3; ++c) free(full[c]);
        return rc;
    }
    /* the reference's framebuffer holds dense i32 coefficients (vardct/mod.rs:262-265); the compact
     * transports of the device ABI are rebuilt to exactly that before anything is computed */
    if (d->coeff_format != JXLGPU_COEFF_DENSE || d->coeff_sample_type != JXLGPU_SAMPLE_I32) return JXLGPU_ERR_INVALID_ARG;
    size_t w8 = (d->width + 7) / 8, h8 = (d->height + 7) / 8;
    size_t wr = w8 * 8, hr = h8 * 8;
    for (int n = 64, i = 0; n <= 256; n *= 2, ++i)
        if (d->sec_half_large[i]) orc_set_sec_half_large(n, d->sec_half_large[i]);

    float* lf[3];
    for (int c = 0; c < 3; ++c) lf[c] = (float*)calloc(w8 * h8, sizeof(float));
    orc_vardct_lf(d, lf);
    if (lf_out)
        for (int c = 0; c < 3; ++c)
            if (lf_out[c]) memcpy(lf_out[c], lf[c], sizeof(float) * w8 * h8);
    if (!(stages & JXLGPU_STAGE_TRANSFORM)) {
        for (int c = 0; c < 3; ++c) free(lf[c]);
        return 0;
    }

    FrameMeta m;
    build_frame_meta(d, &m);
    float* pix[3];
    for (int c = 0; c < 3; ++c) {
        pix[c] = (float*)malloc(sizeof(float) * wr * hr);
#pragma omp parallel for schedule(static)
        for (long y = 0; y < (long)hr; ++y)
            memcpy(pix[c] + (size_t)y * wr, (const int32_t*)d->coeff[c] + (size_t)y * d->coeff_stride, sizeof(float) * wr);
    }
    size_t gpr = (d->width + d->group_dim - 1) / d->group_dim;
    size_t gpc = (d->height + d->group_dim - 1) / d->group_dim;
    const float* lfc[3] = {lf[0], lf[1], lf[2]};
    /* vardct/mod.rs:319: one rayon job per group */
#pragma omp parallel for schedule(dynamic)
    for (long g = 0; g < (long)(gpr * gpc); ++g)
        transform_group(d, &m, pix, wr, lfc, (size_t)g % gpr, (size_t)g / gpr);

    /* render.rs:175-180: noise correlates X and B with the frame's base correlations */
    int rc = orc_post_stages(pix, wr, d->width, d->height, m.sigma, w8, &d->filter, &d->upsampling, &d->noise,
                             d->group_dim, d->base_correlation_x, d->base_correlation_b, &d->color, stages, out,
                             out_stride);
    for (int c = 0; c < 3; ++c) { free(pix[c]); free(lf[c]); }
    free_frame_meta(&m);
    return rc;
}


/* Measurement helper for bench.py's cpu_baseline: libgomp reads OMP_NUM_THREADS once, when the
 * first OpenMP runtime in the process is loaded (torch loads one long before this library). */
#ifdef _OPENMP
#include <omp.h>
#endif
int jxl_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
