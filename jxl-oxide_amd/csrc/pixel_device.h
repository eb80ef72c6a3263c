// Per-pixel device functions shared by the stage kernels and the fused tile kernel: Gabor-like
// 3x3, edge-preserving filter taps, XYB -> display colour.  Arithmetic order follows the
// reference's generic scalar code so results are bit-identical (built with -ffp-contract=off).
#pragma once

#include "common.h"
#include "libm_f32.h"

// jxl-render/src/util.rs:376-386
__device__ __forceinline__ int mirror_idx(int offset, int len) {
    for (;;) {
        if (offset < 0) offset = -(offset + 1);
        else if (offset >= len) offset = -(offset + 1) + len * 2;
        else return offset;
    }
}

// ---------------------------------------------------------------- Gabor-like (F1)
// One output sample.  `at(dx, dy)` reads the input at (x+dx, y+dy) for in-image coordinates only;
// the four border regimes use the reference's own expressions (they differ in summation order
// from the interior one): run_gabor_row_generic (filter/impls/generic/gabor.rs:89-168) for
// interior rows, gabor_row_edge (:3-86) for the first/last row and the single-row image.
template <typename At>
__device__ __forceinline__ float gabor_sample(At at, int x, int y, int width, int height, float w0, float w1) {
    const float global_weight = 1.0f / (1.0f + w0 * 4.0f + w1 * 4.0f);
    if (height == 1) {
        if (width == 1) return at(0, 0);
        float merged_w0 = 1.0f + 2.0f + w0;
        float merged_w1 = w0 + 2.0f * w1;
        if (x == 0) return (at(0, 0) * (merged_w0 + merged_w1) + at(1, 0) * merged_w1) * global_weight;
        if (x == width - 1) return (at(0, 0) * (merged_w0 + merged_w1) + at(-1, 0) * merged_w1) * global_weight;
        return (at(0, 0) * merged_w0 + (at(-1, 0) + at(1, 0)) * merged_w1) * global_weight;
    }
    if (y == 0 || y == height - 1) {
        const int ay = y == 0 ? 1 : -1;  // the adjacent row
        if (width == 1) {
            float u = at(0, ay), c = at(0, 0);
            return (c * (1.0f + 3.0f * w0 + 2.0f * w1) + u * (w0 + 2.0f * w1)) * global_weight;
        }
        if (x == 0) {
            float a1 = at(0, ay), a0 = at(1, ay), c1 = at(0, 0), c0 = at(1, 0);
            return (c1 * (1.0f + 2.0f * w0 + w1) + (a1 + c0) * (w0 + w1) + a0 * w1) * global_weight;
        }
        if (x == width - 1) {
            float a0 = at(-1, ay), a1 = at(0, ay), c0 = at(-1, 0), c1 = at(0, 0);
            return (c1 * (1.0f + 2.0f * w0 + w1) + (a1 + c0) * (w0 + w1) + a0 * w1) * global_weight;
        }
        float a0 = at(-1, ay), a1 = at(0, ay), a2 = at(1, ay);
        float c0 = at(-1, 0), c1 = at(0, 0), c2 = at(1, 0);
        return (c1 + (a1 + c0 + c1 + c2) * w0 + (a0 + a2 + c0 + c2) * w1) * global_weight;
    }
    if (width == 1) {
        float t = at(0, -1), c = at(0, 0), b = at(0, 1);
        float sum_side = t + 2.0f * c + b;
        float sum_diag = 2.0f * (t + b);
        return (c + sum_side * w0 + sum_diag * w1) * global_weight;
    }
    if (x == 0) {
        float t1 = at(0, -1), c1 = at(0, 0), b1 = at(0, 1), t0 = at(1, -1), c0 = at(1, 0), b0 = at(1, 1);
        float sum_side = t1 + c0 + c1 + b1;
        float sum_diag = t0 + t1 + b0 + b1;
        return (c1 + sum_side * w0 + sum_diag * w1) * global_weight;
    }
    if (x == width - 1) {
        float t1 = at(0, -1), c1 = at(0, 0), b1 = at(0, 1), t0 = at(-1, -1), c0 = at(-1, 0), b0 = at(-1, 1);
        float sum_side = t1 + c0 + c1 + b1;
        float sum_diag = t0 + t1 + b0 + b1;
        return (c1 + sum_side * w0 + sum_diag * w1) * global_weight;
    }
    float sum_side = at(0, -1) + at(-1, 0) + at(1, 0) + at(0, 1);
    float sum_diag = at(-1, -1) + at(1, -1) + at(-1, 1) + at(1, 1);
    return (at(0, 0) + sum_side * w0 + sum_diag * w1) * global_weight;
}

// ---------------------------------------------------------------- EPF (F2)
// filter/impls/generic/epf.rs:207-210
__device__ __forceinline__ float epf_weight(float scaled_distance, float sigma, float step_multiplier) {
    const float FRAC_1_SQRT_2 = 0.70710678118654752440f;
    float neg_inv_sigma = 6.6f * (FRAC_1_SQRT_2 - 1.0f) / sigma * step_multiplier;
    return fmaxf(1.0f + scaled_distance * neg_inv_sigma, 0.0f);
}

// (y + 1) & 0b110 == 0, x & 7 in {0, 7}: filter/impls/generic/epf.rs:29-37
__device__ __forceinline__ float epf_step_mul(int x, int y, float step_multiplier, float border_sad_mul) {
    bool is_y_border = ((y + 1) & 6) == 0;
    int xm = x & 7;
    if (is_y_border || xm == 0 || xm == 7) return step_multiplier * border_sad_mul;
    return step_multiplier;
}

// One output pixel of EPF step STEP.  `at(c, dx, dy)` reads channel c at (x+dx, y+dy) with the
// mirroring of run_epf_rows / epf_row (filter/epf.rs:212-216, generic/epf.rs:73-77) applied by
// the caller.  Kernel / distance offsets: filter/epf.rs:263-291, in the reference's order.
template <int STEP, typename At>
__device__ __forceinline__ void epf_pixel(At at, float sigma_val, float sm, const float (&channel_scale)[3],
                                          float (&out)[3]) {
    constexpr int NK = STEP == 0 ? 12 : 4;
    constexpr int ND = STEP == 2 ? 1 : 5;
    constexpr int K1[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
    constexpr int K2[12][2] = {{0, -2}, {-1, -1}, {0, -1}, {1, -1}, {-2, 0}, {-1, 0},
                               {1, 0},  {2, 0},   {-1, 1}, {0, 1},  {1, 1},  {0, 2}};
    constexpr int D0[5][2] = {{0, -1}, {1, 0}, {0, 0}, {-1, 0}, {0, 1}};
    constexpr int D1[5][2] = {{0, -1}, {0, 0}, {0, 1}, {-1, 0}, {1, 0}};
    float sum_weights = 1.0f;
    float sum_channels[3] = {at(0, 0, 0), at(1, 0, 0), at(2, 0, 0)};
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int kx = STEP == 0 ? K2[k][0] : K1[k][0];
        const int ky = STEP == 0 ? K2[k][1] : K1[k][1];
        float dist = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int ix = STEP == 0 ? D0[i][0] : STEP == 1 ? D1[i][0] : 0;
                const int iy = STEP == 0 ? D0[i][1] : STEP == 1 ? D1[i][1] : 0;
                acc += fabsf(at(c, kx + ix, ky + iy) - at(c, ix, iy));
            }
            dist += channel_scale[c] * acc;
        }
        float w = epf_weight(dist, sigma_val, sm);
        sum_weights += w;
#pragma unroll
        for (int c = 0; c < 3; ++c) sum_channels[c] += w * at(c, kx, ky);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = sum_channels[c] / sum_weights;
}

// ---------------------------------------------------------------- colour (C1-C4)
__device__ __forceinline__ float linear_to_srgb_dev(float s) {
    // jxl-color/src/tf/srgb.rs:33-50; tables :4-9 packed into two 64-bit immediates
    const uint64_t UP_LO = 0x5c4d413226190a00ull, UP_HI = 0xc6b9aaa08f837568ull;
    const uint64_t LO_LO = 0x6841e7cb0d04b700ull, LO_HI = 0x0d04b700f2ebd151ull;
    uint32_t v = __float_as_uint(s) & 0x7fffffffu;
    float v_adj = __uint_as_float((v | 0x3e800000u) & 0x3effffffu);
    float pow = 0.059914046f;
    pow = pow * v_adj - 0.10889456f;
    pow = pow * v_adj + 0.107963754f;
    pow = pow * v_adj + 0.018092343f;
    uint32_t idx = ((v >> 23) - 118u) & 0xfu;
    uint32_t sh = (idx & 7u) * 8u;
    uint32_t upper = (uint32_t)(((idx & 8u) ? UP_HI : UP_LO) >> sh) & 0xffu;
    uint32_t lower = (uint32_t)(((idx & 8u) ? LO_HI : LO_LO) >> sh) & 0xffu;
    uint32_t mul = 0x40000000u | (upper << 18) | (lower << 10);
    float vf = __uint_as_float(v);
    float small = vf * 12.92f;
    float acc = pow * __uint_as_float(mul) - 0.055f;
    return copysignf(vf <= 0.0031308f ? small : acc, s);
}

__host__ __device__ __forceinline__ float rational_poly5_dev(float x, const float (&p)[5], const float (&q)[5]) {
    // jxl-color/src/fastmath/rational_poly.rs:2-6
    float yp = p[4], yq = q[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) yp = yp * x + p[i];
#pragma unroll
    for (int i = 3; i >= 0; --i) yq = yq * x + q[i];
    return yp / yq;
}

__host__ __device__ __forceinline__ float linear_to_pq_dev(float s, float intensity_target) {
    // jxl-color/src/tf/pq.rs:127-142, tables :26-35
    const float P[5] = {1.351392e-2f, -1.095778f, 5.522776e1f, 1.492516e2f, 4.838434e1f};
    const float Q[5] = {1.012416f, 2.016708e1f, 9.26371e1f, 1.120607e2f, 2.590418e1f};
    const float PS[5] = {9.863406e-6f, 3.881234e-1f, 1.352821e2f, 6.889862e4f, -2.864824e5f};
    const float QS[5] = {3.371868e1f, 1.477719e3f, 1.608477e4f, -4.389884e4f, -2.072546e5f};
    float y_mult = intensity_target / 10000.0f;
    float a = fabsf(s);
    float a_scaled = a * y_mult;
    float a_1_4 = sqrtf(sqrtf(a_scaled));
    float y = a < 1e-4f ? rational_poly5_dev(a_1_4, PS, QS) : rational_poly5_dev(a_1_4, P, Q);
    return copysignf(y, s);
}

__device__ __forceinline__ float pq_to_linear_dev(float s, float intensity_target) {
    // jxl-color/src/tf/pq.rs:336-343, tables :10-23
    const float P[5] = {2.6297566e-4f, -6.235531e-3f, 7.386023e-1f, 2.6455317f, 5.500349e-1f};
    const float Q[5] = {4.213501e2f, -4.2873682e2f, 1.7436467e2f, -3.3907887e1f, 2.6771877f};
    float y_mult = 10000.0f / intensity_target;
    float a = fabsf(s);
    float x = __builtin_fmaf(a, a, a);
    float y = rational_poly5_dev(x, P, Q);
    return copysignf(y * y_mult, s);
}

// fast_pow2f_generic, jxl-color/src/fastmath/powf.rs:6-24.  `x_floor as i32` saturates in Rust;
// v_cvt_i32_f32 saturates too (and maps NaN to 0).
__device__ __forceinline__ float fast_pow2f_dev(float x) {
    float x_floor = floorf(x);
    float e = __uint_as_float(((uint32_t)(int32_t)x_floor + 127u) << 23);
    float frac = x - x_floor;
    float num = frac + 1.01749063e1f;
    num = num * frac + 4.88687798e1f;
    num = num * frac + 9.85506591e1f;
    num = num * e;
    float den = 2.10242958e-1f * frac + -2.22328856e-2f;
    den = den * frac + -1.94414990e1f;
    den = den * frac + 9.85506633e1f;
    return num / den;
}

// fast_log2f_generic, powf.rs:146-156 (tables :134-144)
__device__ __forceinline__ float fast_log2f_dev(float x) {
    uint32_t x_bits = __float_as_uint(x);
    int32_t exp_bits = (int32_t)(x_bits - 0x3f2aaaabu);
    int32_t exp_shifted = exp_bits >> 23;
    float mantissa = __uint_as_float(x_bits - ((uint32_t)exp_shifted << 23));
    float exp_val = (float)exp_shifted;
    float m = mantissa - 1.0f;
    float yp = 7.4245873327820566e-1f;
    yp = yp * m + 1.4287160470083755f;
    yp = yp * m + -1.8503833400518310e-6f;
    float yq = 1.7409343003366853e-1f;
    yq = yq * m + 1.0096718572241148f;
    yq = yq * m + 9.9032814277590719e-1f;
    return yp / yq + exp_val;
}

__device__ __forceinline__ float fast_powf_dev(float base, float e) {  // powf.rs:242-244
    return fast_pow2f_dev(fast_log2f_dev(base) * e);
}

__device__ __forceinline__ float linear_to_bt709_dev(float a) {  // tf/bt709.rs:60-68
    return a <= 0.018f ? 4.5f * a : __builtin_fmaf(fast_powf_dev(a, 0.45f), 1.099f, -0.099f);
}

__device__ __forceinline__ float apply_gamma_dev(float a, float gamma) {  // tf.rs:60-68
    return a <= 1e-7f ? 0.0f : fast_powf_dev(a, gamma);
}

__device__ __forceinline__ float clamp01_dev(float v) {  // f32::clamp(0,1), convert.rs:951
    v = v < 0.0f ? 0.0f : v;
    return v > 1.0f ? 1.0f : v;
}

// tone_map_generic (convert/tone_map.rs:179-211) with rec2408_eetf_generic (tf/rec2408.rs:4-56);
// the per-frame constants of the EETF (steps 1-2) come precomputed from the host
// (fill_color_args), evaluated with the same f32 operations in the same order.
__device__ __forceinline__ void tone_map_dev(const ColorArgs& cp, float (&v)[3]) {
    float y = v[0] * cp.tm_lum[0] + v[1] * cp.tm_lum[1] + v[2] * cp.tm_lum[2];
    float y_pq = linear_to_pq_dev(y, cp.intensity_target);
    float normalized = (y_pq - cp.tm_lum0_pq) / cp.tm_source_pq_diff;
    float compressed;
    if (normalized < cp.tm_ks) {
        compressed = normalized;
    } else {
        float t = (normalized - cp.tm_ks) / cp.tm_one_sub_ks;
        float t_p2 = t * t;
        float t_p3 = t_p2 * t;
        compressed = (2.0f * t_p3 - 3.0f * t_p2 + 1.0f) * cp.tm_ks + (t_p3 - 2.0f * t_p2 + t) * cp.tm_one_sub_ks +
                     (-2.0f * t_p3 + 3.0f * t_p2) * cp.tm_max_luminance;
    }
    float x = 1.0f - compressed;
    float p4 = x * x * x * x;
    float normalized_target = p4 * cp.tm_min_luminance + compressed;
    float y_mapped = normalized_target * cp.tm_source_pq_diff + cp.tm_lum0_pq;
    y_mapped = pq_to_linear_dev(y_mapped, cp.intensity_target);
    float ratio = fabsf(y) <= 1e-7f ? y_mapped * cp.tm_scale : y_mapped / y * cp.tm_scale;
    v[0] *= ratio;
    v[1] *= ratio;
    v[2] *= ratio;
}

__device__ __forceinline__ void map_gamut_dev(float (&rgb)[3], const float (&lum)[3], float saturation_factor) {
    // jxl-color/src/gamut.rs:4-46
    float y = rgb[0] * lum[0] + rgb[1] * lum[1] + rgb[2] * lum[2];
    float gray_saturation = 0.0f, gray_luminance = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = rgb[i];
        float v_sub_y = v - y;
        float inv_v_sub_y = 1.0f / (v_sub_y == 0.0f ? 1.0f : v_sub_y);
        float v_over_v_sub_y = v * inv_v_sub_y;
        float new_sat = v_sub_y >= 0.0f ? gray_saturation : fmaxf(gray_saturation, v_over_v_sub_y);
        float lum_cand = v_sub_y <= 0.0f ? new_sat : v_over_v_sub_y - inv_v_sub_y;
        gray_luminance = fmaxf(lum_cand, gray_luminance);
        gray_saturation = new_sat;
    }
    float gray_mix = saturation_factor * (gray_saturation - gray_luminance) + gray_luminance;
    if (gray_mix < 0.0f) gray_mix = 0.0f;
    if (gray_mix > 1.0f) gray_mix = 1.0f;
    float mixed[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) mixed[i] = gray_mix * (y - rgb[i]) + rgb[i];
    float max_color_val = 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) max_color_val = fmaxf(rgb[i], max_color_val);
#pragma unroll
    for (int i = 0; i < 3; ++i) rgb[i] = mixed[i] / max_color_val;
}

__device__ __forceinline__ void matmul3vec_dev(const float (&a)[9], float (&v)[3]) {
    // jxl-color/src/ciexyz.rs:81-87
    float b0 = v[0], b1 = v[1], b2 = v[2];
    v[0] = a[0] * b0 + a[1] * b1 + a[2] * b2;
    v[1] = a[3] * b0 + a[4] * b1 + a[5] * b2;
    v[2] = a[6] * b0 + a[7] * b1 + a[8] * b2;
}

// hlg_inverse_oo (tf.rs:118-143) for one pixel; `mixed.powf(exp)` as glibc's powf computes it (libm_f32.h).  A negative
// or zero luminance mix gives what it gives in the reference: NaN (x86's default NaN pattern) or +inf times the sample.
__device__ __forceinline__ void hlg_inverse_oo_dev(const ColorArgs& cp, float (&v)[3]) {
    const float mixed = __builtin_fmaf(v[0], cp.hlg_lum[0], __builtin_fmaf(v[1], cp.hlg_lum[1], v[2] * cp.hlg_lum[2]));
    const float mult = libm_f32::powf(mixed, cp.hlg_exp);
    v[0] *= mult;
    v[1] *= mult;
    v[2] *= mult;
}

// linear_to_hlg (tf.rs:145-160); `ln` as glibc's logf computes it
__device__ __forceinline__ float linear_to_hlg_dev(float s) {
    const float a = fabsf(s);
    const float v = a <= 1.0f / 12.0f ? sqrtf(3.0f * a)
                                      : 0.17883277f * libm_f32::logf(__builtin_fmaf(a, 12.0f, -0.28466892f)) + 0.5599107f;
    return copysignf(v, s);
}

// XybToMixedLms -> Matrix -> [GamutMap -> Matrix] -> [ToneMap] -> [HlgInverseOotf] -> [GamutMap] -> TransferFunction for one pixel
// (op list built at jxl-color/src/convert.rs:208-549; xyb.rs:44-58 for the first op).
// FULL = false: the op lists the fused kernels evaluate (everything but the HLG ones); FULL = true adds the HLG inverse OOTF, the
// GamutMap without a tone map in front of it and linear_to_hlg — the double-precision libm restatements cost registers, so only
// the staged colour kernel (color_kernel, filter_kernels.hip) is built with them and ColorArgs::staged_only sends such frames there.
template <bool FULL>
__device__ __forceinline__ void color_pixel_t(const ColorArgs& cp, float (&v)[3]) {
    if (cp.ycbcr) {
        // ycbcr_to_rgb run_generic, jxl-color/src/ycbcr.rs:40-56 (planes are Cb, Y, Cr)
        const float cb = v[0], yy = v[1] + 128.0f / 255.0f, cr = v[2];
        v[0] = __builtin_fmaf(cr, 1.402f, yy);
        v[1] = __builtin_fmaf(cb, -0.114f * 1.772f / 0.587f, __builtin_fmaf(cr, -0.299f * 1.402f / 0.587f, yy));
        v[2] = __builtin_fmaf(cb, 1.772f, yy);
        return;
    }
    float x = v[0], y = v[1], b = v[2];
    float g_l = y + x, g_m = y - x, g_s = b;
    g_l = g_l - cp.cbrt_opsin_bias[0];
    g_m = g_m - cp.cbrt_opsin_bias[1];
    g_s = g_s - cp.cbrt_opsin_bias[2];
    v[0] = __builtin_fmaf(g_l * g_l, g_l, cp.opsin_bias[0]) * cp.itscale;
    v[1] = __builtin_fmaf(g_m * g_m, g_m, cp.opsin_bias[1]) * cp.itscale;
    v[2] = __builtin_fmaf(g_s * g_s, g_s, cp.opsin_bias[2]) * cp.itscale;
    matmul3vec_dev(cp.matrix, v);
    if (cp.gamut_map == JXLGPU_GAMUT_MAP) {
        map_gamut_dev(v, cp.gamut_lum, cp.gamut_sat);
    } else if (cp.gamut_map == JXLGPU_GAMUT_CLIP) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = clamp01_dev(v[c]);
    }
    if (cp.has_matrix2) matmul3vec_dev(cp.matrix2, v);
    if constexpr (FULL) {
        if (cp.tone_map) tone_map_dev(cp, v);
        if (cp.hlg_ootf) hlg_inverse_oo_dev(cp, v);    // convert.rs:501-536 / :1021-1032
        if (cp.tm_gamut_map) map_gamut_dev(v, cp.tm_lum, cp.tm_gamut_sat);
    } else if (cp.tone_map) {
        tone_map_dev(cp, v);
        if (cp.tm_gamut_map) map_gamut_dev(v, cp.tm_lum, cp.tm_gamut_sat);
    }
    if (cp.tf == JXLGPU_TF_BT709) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = linear_to_bt709_dev(v[c]);
    } else if (cp.tf == JXLGPU_TF_GAMMA) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = apply_gamma_dev(v[c], cp.gamma);
    } else if (cp.tf == JXLGPU_TF_SRGB) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = linear_to_srgb_dev(v[c]);
    } else if (cp.tf == JXLGPU_TF_PQ) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = linear_to_pq_dev(v[c], cp.intensity_target);
    } else if (FULL && cp.tf == JXLGPU_TF_HLG) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = linear_to_hlg_dev(v[c]);
    }
}

__device__ __forceinline__ void color_pixel(const ColorArgs& cp, float (&v)[3]) { color_pixel_t<false>(cp, v); }

// Same, specialised for the common chain XybToMixedLms -> Matrix -> sRGB (no branches).
__device__ __forceinline__ void color_pixel_srgb(const ColorArgs& cp, float (&v)[3]) {
    float x = v[0], y = v[1], b = v[2];
    float g_l = y + x, g_m = y - x, g_s = b;
    g_l = g_l - cp.cbrt_opsin_bias[0];
    g_m = g_m - cp.cbrt_opsin_bias[1];
    g_s = g_s - cp.cbrt_opsin_bias[2];
    v[0] = __builtin_fmaf(g_l * g_l, g_l, cp.opsin_bias[0]) * cp.itscale;
    v[1] = __builtin_fmaf(g_m * g_m, g_m, cp.opsin_bias[1]) * cp.itscale;
    v[2] = __builtin_fmaf(g_s * g_s, g_s, cp.opsin_bias[2]) * cp.itscale;
    matmul3vec_dev(cp.matrix, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = linear_to_srgb_dev(v[c]);
}

// 2^(e/2.4)-style multipliers of linear_to_srgb (srgb.rs:4-9 tables expanded): 0x40000000 |
// upper[idx] << 18 | lower[idx] << 10, looked up from a 16-word LDS table instead of two 64-bit
// shift/select sequences.
__device__ constexpr uint32_t kSrgbMulBits[16] = {
    0x40000000u | (0x00u << 18) | (0x00u << 10), 0x40000000u | (0x0au << 18) | (0xb7u << 10),
    0x40000000u | (0x19u << 18) | (0x04u << 10), 0x40000000u | (0x26u << 18) | (0x0du << 10),
    0x40000000u | (0x32u << 18) | (0xcbu << 10), 0x40000000u | (0x41u << 18) | (0xe7u << 10),
    0x40000000u | (0x4du << 18) | (0x41u << 10), 0x40000000u | (0x5cu << 18) | (0x68u << 10),
    0x40000000u | (0x68u << 18) | (0x51u << 10), 0x40000000u | (0x75u << 18) | (0xd1u << 10),
    0x40000000u | (0x83u << 18) | (0xebu << 10), 0x40000000u | (0x8fu << 18) | (0xf2u << 10),
    0x40000000u | (0xa0u << 18) | (0x00u << 10), 0x40000000u | (0xaau << 18) | (0xb7u << 10),
    0x40000000u | (0xb9u << 18) | (0x04u << 10), 0x40000000u | (0xc6u << 18) | (0x0du << 10)};

__device__ __forceinline__ float linear_to_srgb_lut(float s, const uint32_t* lut) {
    uint32_t v = __float_as_uint(s) & 0x7fffffffu;
    float v_adj = __uint_as_float((v | 0x3e800000u) & 0x3effffffu);
    float pow = 0.059914046f;
    pow = pow * v_adj - 0.10889456f;
    pow = pow * v_adj + 0.107963754f;
    pow = pow * v_adj + 0.018092343f;
    uint32_t idx = ((v >> 23) - 118u) & 0xfu;
    float vf = __uint_as_float(v);
    float small = vf * 12.92f;
    float acc = pow * __uint_as_float(lut[idx]) - 0.055f;
    return copysignf(vf <= 0.0031308f ? small : acc, s);
}

__device__ __forceinline__ void color_pixel_srgb_lut(const ColorArgs& cp, float (&v)[3], const uint32_t* lut) {
    float x = v[0], y = v[1], b = v[2];
    float g_l = y + x, g_m = y - x, g_s = b;
    g_l = g_l - cp.cbrt_opsin_bias[0];
    g_m = g_m - cp.cbrt_opsin_bias[1];
    g_s = g_s - cp.cbrt_opsin_bias[2];
    v[0] = __builtin_fmaf(g_l * g_l, g_l, cp.opsin_bias[0]) * cp.itscale;
    v[1] = __builtin_fmaf(g_m * g_m, g_m, cp.opsin_bias[1]) * cp.itscale;
    v[2] = __builtin_fmaf(g_s * g_s, g_s, cp.opsin_bias[2]) * cp.itscale;
    matmul3vec_dev(cp.matrix, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = linear_to_srgb_lut(v[c], lut);
}
