/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors);
 * checked against f64 formulas (cube + opsin inverse, IEC 61966-2-1 sRGB, SMPTE ST 2084 PQ) in
 * tests/test_oracle_color.py.
 *
 * Follows:
 *   xyb::run_generic            jxl-color/src/xyb.rs:35-60
 *   matmul3vec                  jxl-color/src/ciexyz.rs:81-87   (op Matrix, convert.rs:874-885)
 *   linear_to_srgb (scalar)     jxl-color/src/tf/srgb.rs:33-50
 *   linear_to_pq_generic        jxl-color/src/tf/pq.rs:127-142, tables :26-35
 *   rational_poly::eval_generic jxl-color/src/fastmath/rational_poly.rs:2-6
 *   map_gamut_generic           jxl-color/src/gamut.rs:4-46
 *   pq_to_linear_generic        jxl-color/src/tf/pq.rs:336-343, tables :10-23
 *   rec2408_eetf_generic        jxl-color/src/tf/rec2408.rs:4-56
 *   tone_map_generic            jxl-color/src/convert/tone_map.rs:179-211 (detect_peak = false)
 *   fast_pow2f/log2f/powf_generic  jxl-color/src/fastmath/powf.rs:6-24, :134-156, :242-244
 *   linear_to_bt709 (scalar)    jxl-color/src/tf/bt709.rs:60-68
 *   apply_gamma (scalar)        jxl-color/src/tf.rs:60-68
 *   Clip                        jxl-color/src/convert.rs:948-955
 *   hlg_inverse_oo              jxl-color/src/tf.rs:118-143   (op HlgInverseOotf, convert.rs:906-916; TransferFunction
 *   linear_to_hlg               jxl-color/src/tf.rs:145-160    {Hlg}, convert.rs:1021-1032)
 *     Both go through the platform libm (f32::powf / ln / log2 -> powf / logf / log2f): the calls below are the
 *     SAME libm calls, so on this box the oracle computes what the reference computes on this box.  (The product
 *     restates glibc's powf / logf for the device, csrc/libm_f32.h; tests/test_libm_f32.py compares that
 *     restatement with the installed libm on every float.)
 * Not restated: peak detection (a whole-image reduction ahead of the per-sample
 * pass; ColorTransformBuilder's default is detect_peak = false, convert.rs:141).
 * Op order of the pipeline: jxl-color/src/convert.rs:208-549.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* tf/srgb.rs:4-9 (data tables) */
static const uint8_t SRGB_POWTABLE_UPPER[16] = {0x00, 0x0a, 0x19, 0x26, 0x32, 0x41, 0x4d, 0x5c,
                                                0x68, 0x75, 0x83, 0x8f, 0xa0, 0xaa, 0xb9, 0xc6};
static const uint8_t SRGB_POWTABLE_LOWER[16] = {0x00, 0xb7, 0x04, 0x0d, 0xcb, 0xe7, 0x41, 0x68,
                                                0x51, 0xd1, 0xeb, 0xf2, 0x00, 0xb7, 0x04, 0x0d};

/* tf/srgb.rs:33-50 */
static float linear_to_srgb(float s) {
    uint32_t v = f2u(s) & 0x7fffffffu;
    float v_adj = u2f((v | 0x3e800000u) & 0x3effffffu);
    float pow = 0.059914046f;
    pow = pow * v_adj - 0.10889456f;
    pow = pow * v_adj + 0.107963754f;
    pow = pow * v_adj + 0.018092343f;
    uint32_t idx = ((v >> 23) - 118u) & 0xfu;
    uint32_t mul = 0x40000000u | ((uint32_t)SRGB_POWTABLE_UPPER[idx] << 18) |
                   ((uint32_t)SRGB_POWTABLE_LOWER[idx] << 10);
    float vf = u2f(v);
    float small = vf * 12.92f;
    float acc = pow * u2f(mul) - 0.055f;
    return copysignf(vf <= 0.0031308f ? small : acc, s);
}

/* fastmath/rational_poly.rs:2-6 */
static float rational_poly5(float x, const float p[5], const float q[5]) {
    float yp = p[4];
    for (int i = 3; i >= 0; --i) yp = yp * x + p[i];
    float yq = q[4];
    for (int i = 3; i >= 0; --i) yq = yq * x + q[i];
    return yp / yq;
}

/* tf/pq.rs:26-35 (data tables) */
static const float INV_EOTF_P[5] = {1.351392e-2f, -1.095778f, 5.522776e1f, 1.492516e2f, 4.838434e1f};
static const float INV_EOTF_Q[5] = {1.012416f, 2.016708e1f, 9.26371e1f, 1.120607e2f, 2.590418e1f};
static const float INV_EOTF_P_SMALL[5] = {9.863406e-6f, 3.881234e-1f, 1.352821e2f, 6.889862e4f,
                                          -2.864824e5f};
static const float INV_EOTF_Q_SMALL[5] = {3.371868e1f, 1.477719e3f, 1.608477e4f, -4.389884e4f,
                                          -2.072546e5f};

/* tf/pq.rs:127-142 */
static float linear_to_pq(float s, float intensity_target) {
    float y_mult = intensity_target / 10000.0f;
    float a = fabsf(s);
    float a_scaled = a * y_mult;
    float a_1_4 = sqrtf(sqrtf(a_scaled));
    float y = a < 1e-4f ? rational_poly5(a_1_4, INV_EOTF_P_SMALL, INV_EOTF_Q_SMALL)
                        : rational_poly5(a_1_4, INV_EOTF_P, INV_EOTF_Q);
    return copysignf(y, s);
}

/* tf/pq.rs:10-23 (data tables) */
static const float EOTF_P[5] = {2.6297566e-4f, -6.235531e-3f, 7.386023e-1f, 2.6455317f, 5.500349e-1f};
static const float EOTF_Q[5] = {4.213501e2f, -4.2873682e2f, 1.7436467e2f, -3.3907887e1f, 2.6771877f};

/* tf/pq.rs:336-343 */
static float pq_to_linear(float s, float intensity_target) {
    float y_mult = 10000.0f / intensity_target;
    float a = fabsf(s);
    float x = fmaf(a, a, a);
    float y = rational_poly5(x, EOTF_P, EOTF_Q);
    return copysignf(y * y_mult, s);
}

/* tf/rec2408.rs:4-56 */
static float rec2408_eetf(float from_pq_sample, float intensity_target, const float from_range[2],
                          const float to_range[2]) {
    float lum[4] = {from_range[0] / intensity_target, from_range[1] / intensity_target,
                    to_range[0] / intensity_target, to_range[1] / intensity_target};
    for (int i = 0; i < 4; ++i) lum[i] = linear_to_pq(lum[i], intensity_target);
    /* Step 1 */
    float source_pq_diff = lum[1] - lum[0];
    float normalized = (from_pq_sample - lum[0]) / source_pq_diff;
    float min_luminance = (lum[2] - lum[0]) / source_pq_diff;
    float max_luminance = (lum[3] - lum[0]) / source_pq_diff;
    /* Step 2 */
    float ks = 1.5f * max_luminance - 0.5f;
    float b = min_luminance;
    /* Step 3, 4 */
    float compressed;
    if (normalized < ks) {
        compressed = normalized;
    } else {
        float one_sub_ks = 1.0f - ks;
        float t = (normalized - ks) / one_sub_ks;
        float t_p2 = t * t;
        float t_p3 = t_p2 * t;
        compressed = (2.0f * t_p3 - 3.0f * t_p2 + 1.0f) * ks + (t_p3 - 2.0f * t_p2 + t) * one_sub_ks +
                     (-2.0f * t_p3 + 3.0f * t_p2) * max_luminance;
    }
    float x = 1.0f - compressed;
    float one_sub_compressed_p4 = x * x * x * x;
    float normalized_target = one_sub_compressed_p4 * b + compressed;
    /* Step 5 */
    return normalized_target * source_pq_diff + lum[0];
}

/* convert/tone_map.rs:8-31 (detect_peak = false: peak = intensity_target) and :179-211 */
static void tone_map(float rgb[3], const float lum[3], float intensity_target, float min_nits,
                     float target_display_luminance) {
    float peak_luminance = fminf(intensity_target, intensity_target);
    float from_range[2] = {min_nits, peak_luminance};
    float to_range[2] = {0.0f, target_display_luminance};
    float scale = intensity_target / to_range[1];
    float y = rgb[0] * lum[0] + rgb[1] * lum[1] + rgb[2] * lum[2];
    float y_pq = linear_to_pq(y, intensity_target);
    float y_mapped = rec2408_eetf(y_pq, intensity_target, from_range, to_range);
    y_mapped = pq_to_linear(y_mapped, intensity_target);
    float ratio = fabsf(y) <= 1e-7f ? y_mapped * scale : y_mapped / y * scale;
    rgb[0] *= ratio;
    rgb[1] *= ratio;
    rgb[2] *= ratio;
}

/* Hooks for tests/test_oracle_reference_vectors.py: the reference's own unit tests of these three
 * functions (tf/pq.rs:455-537, convert/tone_map.rs:760-815) are replayed against the restatement. */
float orc_test_linear_to_pq(float s, float intensity_target) { return linear_to_pq(s, intensity_target); }
float orc_test_pq_to_linear(float s, float intensity_target) { return pq_to_linear(s, intensity_target); }
void orc_test_tone_map(float rgb[3], const float lum[3], float intensity_target, float min_nits,
                       float target_display_luminance) {
    tone_map(rgb, lum, intensity_target, min_nits, target_display_luminance);
}

/* fastmath/powf.rs:3-4, :134-144 (data tables) */
static const float POW2F_NUMER[3] = {1.01749063e1f, 4.88687798e1f, 9.85506591e1f};
static const float POW2F_DENOM[4] = {2.10242958e-1f, -2.22328856e-2f, -1.94414990e1f, 9.85506633e1f};
static const float LOG2F_P[3] = {-1.8503833400518310e-6f, 1.4287160470083755f, 7.4245873327820566e-1f};
static const float LOG2F_Q[3] = {9.9032814277590719e-1f, 1.0096718572241148f, 1.7409343003366853e-1f};

/* Rust `f32 as i32`: saturating, NaN -> 0 */
static int32_t f32_as_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

/* fastmath/powf.rs:6-24 */
static float fast_pow2f(float x) {
    float x_floor = floorf(x);
    float exp = u2f(((uint32_t)f32_as_i32(x_floor) + 127u) << 23);
    float frac = x - x_floor;
    float num = frac + POW2F_NUMER[0];
    num = num * frac + POW2F_NUMER[1];
    num = num * frac + POW2F_NUMER[2];
    num = num * exp;
    float den = POW2F_DENOM[0] * frac + POW2F_DENOM[1];
    den = den * frac + POW2F_DENOM[2];
    den = den * frac + POW2F_DENOM[3];
    return num / den;
}

/* fastmath/powf.rs:146-156 (+ rational_poly.rs:2-6 with P = Q = 3) */
static float fast_log2f(float x) {
    uint32_t x_bits = f2u(x);
    int32_t exp_bits = (int32_t)(x_bits - 0x3f2aaaabu);
    int32_t exp_shifted = exp_bits >> 23;
    float mantissa = u2f(x_bits - ((uint32_t)exp_shifted << 23));
    float exp_val = (float)exp_shifted;
    float m = mantissa - 1.0f;
    float yp = LOG2F_P[2];
    yp = yp * m + LOG2F_P[1];
    yp = yp * m + LOG2F_P[0];
    float yq = LOG2F_Q[2];
    yq = yq * m + LOG2F_Q[1];
    yq = yq * m + LOG2F_Q[0];
    return yp / yq + exp_val;
}

/* fastmath/powf.rs:242-244 */
static float fast_powf(float base, float exp) { return fast_pow2f(fast_log2f(base) * exp); }

/* tf/bt709.rs:60-68 */
static float linear_to_bt709(float a) {
    return a <= 0.018f ? 4.5f * a : fmaf(fast_powf(a, 0.45f), 1.099f, -0.099f);
}

/* tf.rs:60-68 */
static float apply_gamma(float a, float gamma) { return a <= 1e-7f ? 0.0f : fast_powf(a, gamma); }

/* f32::clamp(0.0, 1.0) (convert.rs:951): NaN stays NaN */
static float clamp01(float v) {
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}

/* f32::max / f32::min semantics (IEEE maxNum: NaN loses) == fmaxf/fminf */

/* gamut.rs:4-46 */
static void map_gamut(float rgb[3], const float lum[3], float saturation_factor) {
    float r = rgb[0], g = rgb[1], b = rgb[2];
    float y = r * lum[0] + g * lum[1] + b * lum[2];
    float gray_saturation = 0.0f, gray_luminance = 0.0f;
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
    /* f32::clamp(0,1) */
    if (gray_mix < 0.0f) gray_mix = 0.0f;
    if (gray_mix > 1.0f) gray_mix = 1.0f;
    float mixed[3];
    for (int i = 0; i < 3; ++i) mixed[i] = gray_mix * (y - rgb[i]) + rgb[i];
    float max_color_val = 1.0f;
    for (int i = 0; i < 3; ++i) max_color_val = fmaxf(rgb[i], max_color_val);
    for (int i = 0; i < 3; ++i) rgb[i] = mixed[i] / max_color_val;
}

/* libm through pointers the compiler cannot see through: no constant folding (GCC folds with MPFR, correctly
 * rounded, which is not always what the library returns), no builtin expansion */
static float (*volatile libm_powf)(float, float) = powf;
static float (*volatile libm_logf)(float) = logf;
static float (*volatile libm_log2f)(float) = log2f;

/* tf.rs:118-143: hlg_inverse_oo.  Returns without touching the samples for 295 <= intensity_target <= 305. */
static void hlg_inverse_oo(float rgb[3], const float lum[3], float intensity_target) {
    if (intensity_target >= 295.0f && intensity_target <= 305.0f) return;
    float gamma = 1.2f * libm_powf(1.111f, libm_log2f(intensity_target / 1e3f));
    float exp = (1.0f - gamma) / gamma;
    float mixed = fmaf(rgb[0], lum[0], fmaf(rgb[1], lum[1], rgb[2] * lum[2]));
    float mult = libm_powf(mixed, exp);
    rgb[0] *= mult;
    rgb[1] *= mult;
    rgb[2] *= mult;
}

/* tf.rs:145-160: linear_to_hlg */
static float linear_to_hlg(float s) {
    const float HLG_A = 0.17883277f, HLG_B = 0.28466892f, HLG_C = 0.5599107f;
    float a = fabsf(s);
    float v = a <= 1.0f / 12.0f ? sqrtf(3.0f * a) : HLG_A * libm_logf(fmaf(a, 12.0f, -HLG_B)) + HLG_C;
    return copysignf(v, s);
}

/* hooks for tests/test_oracle_color.py */
float orc_test_linear_to_hlg(float s) { return linear_to_hlg(s); }
void orc_test_hlg_inverse_oo(float rgb[3], const float lum[3], float intensity_target) { hlg_inverse_oo(rgb, lum, intensity_target); }

static void matmul3vec(const float a[9], float v[3]) {
    float b0 = v[0], b1 = v[1], b2 = v[2];
    v[0] = a[0] * b0 + a[1] * b1 + a[2] * b2;
    v[1] = a[3] * b0 + a[4] * b1 + a[5] * b2;
    v[2] = a[6] * b0 + a[7] * b1 + a[8] * b2;
}

/* convert.rs:619-659 runs the op list per 65 536-sample chunk; every op is per-sample, so the
 * per-sample composition below is arithmetically identical. */
void orc_color_transform(float* const ch[3], size_t n, const JxlGpuColorParams* cp) {
    if (cp->ycbcr) {  /* lib.rs:950-954: do_ycbcr frames are not XYB */
        orc_ycbcr_to_rgb(ch[0], ch[1], ch[2], n);
        return;
    }
    if (!cp->enabled) return;
    float itscale = 255.0f / cp->intensity_target;
    float cbrt_ob[3];
    for (int c = 0; c < 3; ++c) cbrt_ob[c] = cbrtf(cp->opsin_bias[c]);
#pragma omp parallel for schedule(static)
    for (long chunk = 0; chunk < (long)((n + 65535) / 65536); ++chunk)
    for (size_t i = (size_t)chunk * 65536; i < (size_t)(chunk + 1) * 65536 && i < n; ++i) {
        float x = ch[0][i], y = ch[1][i], b = ch[2][i];
        /* xyb.rs:44-58 */
        float g_l = y + x, g_m = y - x, g_s = b;
        g_l = g_l - cbrt_ob[0];
        g_m = g_m - cbrt_ob[1];
        g_s = g_s - cbrt_ob[2];
        float v[3];
        v[0] = fmaf(g_l * g_l, g_l, cp->opsin_bias[0]) * itscale;
        v[1] = fmaf(g_m * g_m, g_m, cp->opsin_bias[1]) * itscale;
        v[2] = fmaf(g_s * g_s, g_s, cp->opsin_bias[2]) * itscale;
        matmul3vec(cp->matrix, v);
        if (cp->gamut_map == JXLGPU_GAMUT_MAP) map_gamut(v, cp->gamut_luminances, cp->gamut_saturation_factor);
        else if (cp->gamut_map == JXLGPU_GAMUT_CLIP) for (int c = 0; c < 3; ++c) v[c] = clamp01(v[c]);
        if (cp->has_matrix2) matmul3vec(cp->matrix2, v);
        if (cp->tone_map)
            tone_map(v, cp->tm_luminances, cp->intensity_target, cp->tm_min_nits, cp->tm_target_display_luminance);
        /* convert.rs:501-536 (PQ image, HLG target): ToneMapRec2408 -> HlgInverseOotf -> GamutMap; :1021-1032 (the
         * transfer function's own inverse OOTF): nothing sits between it and linear_to_hlg, so one position serves both */
        if (cp->hlg_ootf_intensity_target != 0.0f) hlg_inverse_oo(v, cp->hlg_luminances, cp->hlg_ootf_intensity_target);
        if (cp->tm_gamut_map) map_gamut(v, cp->tm_luminances, cp->tm_gamut_saturation_factor);
        for (int c = 0; c < 3; ++c) {
            switch (cp->transfer_function) {
                case JXLGPU_TF_SRGB: v[c] = linear_to_srgb(v[c]); break;
                case JXLGPU_TF_PQ: v[c] = linear_to_pq(v[c], cp->intensity_target); break;
                case JXLGPU_TF_BT709: v[c] = linear_to_bt709(v[c]); break;
                case JXLGPU_TF_GAMMA: v[c] = apply_gamma(v[c], cp->gamma); break;
                case JXLGPU_TF_HLG: v[c] = linear_to_hlg(v[c]); break;
                default: break;
            }
        }
        ch[0][i] = v[0]; ch[1][i] = v[1]; ch[2][i] = v[2];
    }
}
