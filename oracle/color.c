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

static void matmul3vec(const float a[9], float v[3]) {
    float b0 = v[0], b1 = v[1], b2 = v[2];
    v[0] = a[0] * b0 + a[1] * b1 + a[2] * b2;
    v[1] = a[3] * b0 + a[4] * b1 + a[5] * b2;
    v[2] = a[6] * b0 + a[7] * b1 + a[8] * b2;
}

/* convert.rs:619-659 runs the op list per 65 536-sample chunk; every op is per-sample, so the
 * per-sample composition below is arithmetically identical. */
void orc_color_transform(float* const ch[3], size_t n, const JxlGpuColorParams* cp) {
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
        if (cp->gamut_map) map_gamut(v, cp->gamut_luminances, cp->gamut_saturation_factor);
        if (cp->has_matrix2) matmul3vec(cp->matrix2, v);
        for (int c = 0; c < 3; ++c) {
            switch (cp->transfer_function) {
                case JXLGPU_TF_SRGB: v[c] = linear_to_srgb(v[c]); break;
                case JXLGPU_TF_PQ: v[c] = linear_to_pq(v[c], cp->intensity_target); break;
                default: break;
            }
        }
        ch[0][i] = v[0]; ch[1][i] = v[1]; ch[2][i] = v[2];
    }
}
