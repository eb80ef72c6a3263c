// F3 + C1-C4 in one pass for 2x upsampling (BASELINE config 5): non-separable upsampling
// (jxl-render/src/features/upsampling.rs:45-132) with the colour transform as its epilogue.
//
// The stage-at-a-time kernel (filter_kernels.hip) spends a thread per OUTPUT sample: 25 loads with
// mirror arithmetic and 25 weight loads each, then a second full 8K pass for the colour transform.
// Here a lane owns one INPUT column of all three channels and walks down the rows; the 5x5 window
// lives in a register ring (the x-2..x+2 neighbours of a row are fetched once, with DPP lane
// shifts, when the row enters the ring), the per-row min / max are cached, the 25 weights are
// scalar registers, and the four output phases of an input sample are produced together and
// converted to the output colour space before the only store.  The image border needs no special
// case: lanes and rows outside the image load the mirrored sample (util.rs:423-454 == mirror() for
// dimensions >= 2), exactly what the reference's padded copy holds.
//
// Arithmetic is the reference's: sum = 0; for iy, for ix: sum += w[ky][kx] * sample (mul, then add;
// no contraction), clamp to [min, max] of the 25 samples.
#include "common.h"
#include "pixel_device.h"
#include "fast_math_device.h"

namespace {

constexpr int UW = 60;   // output (input-resolution) columns per wave: 64 lanes - 2 halo lanes per side

struct UpStreamArgs {
    const float* in[3];
    float* out[3];
    uint32_t in_stride, out_stride;
    int w, h;                 // input size
    float wq[25];             // the one 5x5 kernel of K = 2 (weights_quarter[0])
    ColorArgs color;
    uint32_t do_color;
    int rows_per_seg, strips, segs;
    int wx0, wy0, wx1, wy1;   // input columns / rows whose 2x2 outputs are produced (the whole plane, or a region's)
};

__device__ __forceinline__ float lane_m1(float v) {   // value held by lane - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_p1(float v) {   // value held by lane + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

struct UpState {
    float R[5][3][5];     // rows r-2..r+2 (slot 4 = the newest), channel, column x-2..x+2
    float mn[5][3], mx[5][3];
};

// One code body per kernel (the colour transform of the general op list is ~1200 instructions: the
// row loop is NOT unrolled over ring phases and the two output rows of an input row share one copy
// through a rolled loop, or the kernel would not fit the instruction cache): the ring rotates with
// register moves, ~100 per row against the ~3000 of the sums and the colour transform.
template <bool COLOR>
__global__ __launch_bounds__(256) void upsample2_stream_kernel(UpStreamArgs a) {
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int strip = wave % a.strips, seg = wave / a.strips;
    if (seg >= a.segs) return;
    const int x = a.wx0 + strip * UW - 2 + lane;
    const int xl = mirror_idx(min(max(x, -a.w), 2 * a.w - 1), a.w);
    const bool store_lane = lane >= 2 && lane < 2 + UW && x < a.wx1;
    const uint32_t x_out_off = (uint32_t)(2 * max(x, 0)) * 4u;
    const int y0 = a.wy0 + seg * a.rows_per_seg, y1 = min(y0 + a.rows_per_seg, a.wy1);
    UpState st;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            st.mn[s][c] = st.mx[s][c] = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; ++i) st.R[s][c][i] = 0.0f;
        }
    // rows y0-2 .. y1+1 enter the ring; output starts once row y0+2 is in (centre y0)
#pragma unroll 1
    for (int j = y0 - 2; j < y1 + 2; ++j) {
        // ---- rotate, load row j (mirrored), fetch the x neighbours once
        const int jm = mirror_idx(j, a.h);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st.mn[s][c] = st.mn[s + 1][c];
                st.mx[s][c] = st.mx[s + 1][c];
#pragma unroll
                for (int i = 0; i < 5; ++i) st.R[s][c][i] = st.R[s + 1][c][i];
            }
            const float v = (a.in[c] + (size_t)(uint32_t)jm * a.in_stride)[xl];
            const float l1 = lane_m1(v), r1 = lane_p1(v);
            const float l2 = lane_m1(l1), r2 = lane_p1(r1);
            st.R[4][c][0] = l2; st.R[4][c][1] = l1; st.R[4][c][2] = v; st.R[4][c][3] = r1; st.R[4][c][4] = r2;
            st.mn[4][c] = fminf(fminf(fminf(l2, l1), fminf(v, r1)), r2);
            st.mx[4][c] = fmaxf(fmaxf(fmaxf(l2, l1), fmaxf(v, r1)), r2);
        }
        const int r = j - 2;
        if (r < y0) continue;  // wave-uniform
        float mn[3], mx[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mn[c] = st.mn[0][c]; mx[c] = st.mx[0][c];
#pragma unroll
            for (int s = 1; s < 5; ++s) { mn[c] = fminf(mn[c], st.mn[s][c]); mx[c] = fmaxf(mx[c], st.mx[s][c]); }
        }
#pragma unroll 1
        for (int ym = 0; ym < 2; ++ym) {
            // this output row's weights: rows flipped for ym = 1 (flip_v = ym >= MAT_N); uniform loads
            float wr[5][5];
#pragma unroll
            for (int iy = 0; iy < 5; ++iy)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) wr[iy][kx] = a.wq[(ym ? 4 - iy : iy) * 5 + kx];
            float o[2][3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int xm = 0; xm < 2; ++xm) {
                    float sum = 0.0f;
#pragma unroll
                    for (int iy = 0; iy < 5; ++iy)
#pragma unroll
                        for (int ix = 0; ix < 5; ++ix) sum += wr[iy][xm ? 4 - ix : ix] * st.R[iy][c][ix];
                    float v;
                    if (!isfinite(mn[c])) v = __builtin_nanf("");
                    else {
                        v = sum;
                        if (v < mn[c]) v = mn[c];
                        if (v > mx[c]) v = mx[c];
                    }
                    o[xm][c] = v;
                }
            if constexpr (COLOR) {
                color_pixel(a.color, o[0]);
                color_pixel(a.color, o[1]);
            }
            if (store_lane) {
                const size_t orow = (size_t)(uint32_t)(2 * r + ym) * a.out_stride;  // uniform
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(a.out[c] + orow) + x_out_off);
                    *reinterpret_cast<float2*>(p) = make_float2(o[0][c], o[1][c]);
                }
            }
        }
    }
}


// ---- packed colour chain for the HDR op list of config 5: XybToMixedLms -> Matrix -> GamutMap -> Matrix -> PQ
// (color_pixel with gamut_map == JXLGPU_GAMUT_MAP, has_matrix2, no tone map, tf == PQ, not YCbCr).  The two output
// pixels xm = 0, 1 of an input sample are the halves of every value: a multiplication, addition or mul_add of the
// scalar chain becomes ONE v_pk_{mul,add,fma}_f32 on the pair — the same operation per half, each rounded once —
// while divisions, square roots, comparisons and selects stay per element (the compiler's IEEE expansions), so the
// result is color_pixel's, bit for bit.  The ~50 constants of the chain are packed operands: they sit as {c, c}
// pairs in LDS (the compiler's own one-SGPR `op_sel_hi` broadcast is the form that read a stale half, DESIGN §2;
// 100 SGPRs of real pairs do not fit) and are fetched next to their use.
typedef float cf2 __attribute__((ext_vector_type(2)));
enum {
    HC_CBRT = 0,      // 3: cbrt_opsin_bias
    HC_BIAS = 3,      // 3: opsin_bias
    HC_ITS = 6,       // itscale
    HC_M = 7,         // 9: matrix
    HC_LUM = 16,      // 3: gamut luminances
    HC_SAT = 19,      // gamut saturation factor
    HC_M2 = 20,       // 9: matrix2
    HC_YMULT = 29,    // intensity_target / 10000
    HC_P = 30, HC_Q = 35, HC_PS = 40, HC_QS = 45,   // 5 each: linear_to_pq's rational polynomials
    HC_COUNT = 50
};

__device__ __forceinline__ void hdr_consts_fill(cf2* tab, const ColorArgs& cp, int lane) {
    // jxl-color/src/tf/pq.rs:26-35 (the tables of linear_to_pq_dev)
    const float P[5] = {1.351392e-2f, -1.095778f, 5.522776e1f, 1.492516e2f, 4.838434e1f};
    const float Q[5] = {1.012416f, 2.016708e1f, 9.26371e1f, 1.120607e2f, 2.590418e1f};
    const float PS[5] = {9.863406e-6f, 3.881234e-1f, 1.352821e2f, 6.889862e4f, -2.864824e5f};
    const float QS[5] = {3.371868e1f, 1.477719e3f, 1.608477e4f, -4.389884e4f, -2.072546e5f};
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < HC_COUNT; ++i) {
        float c;
        if (i < 3) c = cp.cbrt_opsin_bias[i];
        else if (i < 6) c = cp.opsin_bias[i - 3];
        else if (i == 6) c = cp.itscale;
        else if (i < 16) c = cp.matrix[i - 7];
        else if (i < 19) c = cp.gamut_lum[i - 16];
        else if (i == 19) c = cp.gamut_sat;
        else if (i < 29) c = cp.matrix2[i - 20];
        else if (i == 29) c = cp.intensity_target / 10000.0f;
        else if (i < 35) c = P[i - 30];
        else if (i < 40) c = Q[i - 35];
        else if (i < 45) c = PS[i - 40];
        else c = QS[i - 45];
        v = lane == i ? c : v;
    }
    if (lane < HC_COUNT) tab[lane] = cf2{v, v};
}

__device__ __forceinline__ cf2 pk_fma(cf2 a, cf2 b, cf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cf2 sel2(bool cx, bool cy, cf2 a, cf2 b) { return cf2{cx ? a.x : b.x, cy ? a.y : b.y}; }

// matmul3vec_dev on pairs (ciexyz.rs:81-87): (a0 b0 + a1 b1) + a2 b2, no contraction
__device__ __forceinline__ void matmul3_pair(const cf2* m, cf2 (&v)[3]) {
    const cf2 b0 = v[0], b1 = v[1], b2 = v[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = m[3 * r] * b0 + m[3 * r + 1] * b1 + m[3 * r + 2] * b2;
}

// map_gamut_dev on pairs (gamut.rs:4-46)
__device__ __forceinline__ void map_gamut_pair(const cf2* k, cf2 (&rgb)[3]) {
    const cf2 y = rgb[0] * k[HC_LUM] + rgb[1] * k[HC_LUM + 1] + rgb[2] * k[HC_LUM + 2];
    cf2 gray_saturation = {0.0f, 0.0f}, gray_luminance = {0.0f, 0.0f};
    // 1 / (v - y) for the three channels: the residual-chain reciprocal of fast_math_device.h when every denominator of the wave
    // lies in 2^-60 <= |d| <= 2^60 (an exact zero is replaced by 1, as in the reference), the ordinary divisions otherwise
    cf2 dsub[3];
    bool rcp_ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const cf2 t = rgb[i] - y;
        dsub[i] = cf2{t.x == 0.0f ? 1.0f : t.x, t.y == 0.0f ? 1.0f : t.y};
        rcp_ok = rcp_ok && fm_in_range_bits(fabsf(dsub[i].x), kFmBits2m60, kFmBits2p60) && fm_in_range_bits(fabsf(dsub[i].y), kFmBits2m60, kFmBits2p60);
    }
    const bool rcp_fast = __builtin_amdgcn_ballot_w64(!rcp_ok) == 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const cf2 v = rgb[i];
        const cf2 v_sub_y = v - y;
        cf2 inv;
        if (rcp_fast) {
            inv = rcp_cr_pair(dsub[i]);
        } else {
            asm volatile("; ordinary reciprocals" ::: "memory");
            inv = cf2{1.0f / dsub[i].x, 1.0f / dsub[i].y};
        }
        const cf2 v_over = v * inv;
        const cf2 new_sat = sel2(v_sub_y.x >= 0.0f, v_sub_y.y >= 0.0f, gray_saturation,
                                 cf2{fmaxf(gray_saturation.x, v_over.x), fmaxf(gray_saturation.y, v_over.y)});
        const cf2 lum_cand = sel2(v_sub_y.x <= 0.0f, v_sub_y.y <= 0.0f, new_sat, v_over - inv);
        gray_luminance = cf2{fmaxf(lum_cand.x, gray_luminance.x), fmaxf(lum_cand.y, gray_luminance.y)};
        gray_saturation = new_sat;
    }
    cf2 gray_mix = k[HC_SAT] * (gray_saturation - gray_luminance) + gray_luminance;
    gray_mix = cf2{gray_mix.x < 0.0f ? 0.0f : gray_mix.x, gray_mix.y < 0.0f ? 0.0f : gray_mix.y};
    gray_mix = cf2{gray_mix.x > 1.0f ? 1.0f : gray_mix.x, gray_mix.y > 1.0f ? 1.0f : gray_mix.y};
    cf2 mixed[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) mixed[i] = gray_mix * (y - rgb[i]) + rgb[i];
    cf2 max_color_val = {1.0f, 1.0f};
#pragma unroll
    for (int i = 0; i < 3; ++i) max_color_val = cf2{fmaxf(rgb[i].x, max_color_val.x), fmaxf(rgb[i].y, max_color_val.y)};
    // three quotients by one denominator >= 1 (gamut.rs:40-45).  Shared refined reciprocal + fma chains where that is provably the
    // correctly rounded quotient (fast_math_device.h: 1 <= d <= 2^20, 2^-100 <= |n| <= 2^20 or n = +-0), tested wave by wave;
    // the ordinary divisions otherwise (a tiny non-zero numerator, huge values, NaN)
    {
        // (max_color_val >= 1 by construction; the smallest and the largest magnitude of the six numerators with three-operand
        //  min / max and |.| modifiers: four instructions; a wave that fails looks again with exact zeros allowed)
        const float lo = fminf(fm_min3_abs(mixed[0].x, mixed[1].x, mixed[2].x), fm_min3_abs(mixed[0].y, mixed[1].y, mixed[2].y));
        const float hi = fm_max3(fm_max3_abs(mixed[0].x, mixed[1].x, mixed[2].x), fm_max3_abs(mixed[0].y, mixed[1].y, mixed[2].y),
                                 fmaxf(max_color_val.x, max_color_val.y));
        const cf2 nsum = mixed[0] + mixed[1] + mixed[2];                      // (v_min3 / v_max3 skip a NaN operand: a NaN numerator shows here)
        const bool no_nan = nsum.x == nsum.x && nsum.y == nsum.y;
        bool ok = lo >= 0x1p-100f && hi <= 0x1p20f && no_nan;
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
            ok = hi <= 0x1p20f && no_nan;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float ax = fabsf(mixed[i].x), ay = fabsf(mixed[i].y);
                ok = ok && (ax == 0.0f || ax >= 0x1p-100f) && (ay == 0.0f || ay >= 0x1p-100f);
            }
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
            const cf2 r = rcp_refined_pair(max_color_val);
#pragma unroll
            for (int i = 0; i < 3; ++i) rgb[i] = div_cr_pair_with(mixed[i], max_color_val, r);
        } else {
            asm volatile("; ordinary divisions" ::: "memory");
#pragma unroll
            for (int i = 0; i < 3; ++i) rgb[i] = cf2{mixed[i].x / max_color_val.x, mixed[i].y / max_color_val.y};
        }
    }
}

// rational_poly5_dev's numerator / denominator on pairs (fastmath/rational_poly.rs:2-6)
__device__ __forceinline__ cf2 horner5_pair(cf2 x, const cf2* p) {
    cf2 yv = p[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) yv = yv * x + p[i];
    return yv;
}

// linear_to_pq_dev on a pair (tf/pq.rs:127-142)
// `y_mult_le_1` (uniform): intensity_target <= 10000, the range of PQ itself.  Then, for 2^-60 <= a_scaled <= 2^12 (tested wave by
// wave), the fourth root is two correctly rounded square roots by the residual chain of fast_math_device.h (checked on the
// device against sqrtf(sqrtf(x)) for EVERY float of that range), x = a_scaled^(1/4) lies in [2^-15, 8], and the selected
// polynomials are inside the quotient chain's proven range: P(x) in [0.008, 2.8e5], Q(x) in [1.01, 1.7e5] on [0, 8]; the dark
// pair is only selected for a < 1e-4, i.e. x <= 0.1, where PS(x) in (0, 42] and QS(x) in [33, 280].  Everything else — zeros
// (a black image), denormals, huge or non-finite samples, larger intensity targets — keeps sqrtf and the ordinary division.
__device__ __forceinline__ cf2 linear_to_pq_pair(cf2 s, const cf2* k, bool y_mult_le_1) {
    const cf2 a = {fabsf(s.x), fabsf(s.y)};
    const cf2 a_scaled = a * k[HC_YMULT];
    const bool in_range = fm_in_range_bits(a_scaled.x, kFmBits2m60, kFmBits2p12) && fm_in_range_bits(a_scaled.y, kFmBits2m60, kFmBits2p12);
    const bool fast = y_mult_le_1 && __builtin_amdgcn_ballot_w64(!in_range) == 0;
    cf2 a_1_4;
    if (fast) {
        a_1_4 = sqrt_cr_pair(sqrt_cr_pair(a_scaled));
    } else {
        asm volatile("; ordinary square roots" ::: "memory");
        a_1_4 = cf2{sqrtf(sqrtf(a_scaled.x)), sqrtf(sqrtf(a_scaled.y))};
    }
    const bool sx = a.x < 1e-4f, sy = a.y < 1e-4f;
    cf2 yp = horner5_pair(a_1_4, k + HC_P), yq = horner5_pair(a_1_4, k + HC_Q);
    if (__builtin_amdgcn_ballot_w64(sx || sy) != 0) {   // dark samples: the other pair of polynomials (wave-uniform test)
        const cf2 yps = horner5_pair(a_1_4, k + HC_PS), yqs = horner5_pair(a_1_4, k + HC_QS);
        yp = sel2(sx, sy, yps, yp);
        yq = sel2(sx, sy, yqs, yq);
    }
    cf2 q;
    if (fast) {
        q = div_cr_pair(yp, yq);
    } else {
        asm volatile("; ordinary division" ::: "memory");
        q = cf2{yp.x / yq.x, yp.y / yq.y};
    }
    return cf2{copysignf(q.x, s.x), copysignf(q.y, s.y)};
}

__device__ __forceinline__ void color_pair_hdr(const cf2* k, cf2 (&v)[3]) {
    const cf2 x = v[0], y = v[1], b = v[2];
    cf2 g_l = y + x, g_m = y - x, g_s = b;
    g_l = g_l - k[HC_CBRT];
    g_m = g_m - k[HC_CBRT + 1];
    g_s = g_s - k[HC_CBRT + 2];
    v[0] = pk_fma(g_l * g_l, g_l, k[HC_BIAS]) * k[HC_ITS];
    v[1] = pk_fma(g_m * g_m, g_m, k[HC_BIAS + 1]) * k[HC_ITS];
    v[2] = pk_fma(g_s * g_s, g_s, k[HC_BIAS + 2]) * k[HC_ITS];
    asm volatile("" ::: "memory");   // (phase boundaries: keeps the compiler from fetching every constant pair up front)
    matmul3_pair(k + HC_M, v);
    asm volatile("" ::: "memory");
    map_gamut_pair(k, v);
    asm volatile("" ::: "memory");
    matmul3_pair(k + HC_M2, v);
    const bool y_mult_le_1 = __builtin_amdgcn_readfirstlane(__float_as_int(k[HC_YMULT].x)) <= __float_as_int(1.0f) &&
                             __builtin_amdgcn_readfirstlane(__float_as_int(k[HC_YMULT].x)) >= 0;   // 0 <= y_mult <= 1 (bit patterns of non-negative floats order like the values)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        asm volatile("" ::: "memory");
        v[c] = linear_to_pq_pair(v[c], k, y_mult_le_1);
    }
}

// Is the colour op list the one color_pair_hdr evaluates?
static bool is_hdr_pq_list(const ColorArgs& c) {
    return !c.ycbcr && c.gamut_map == JXLGPU_GAMUT_MAP && c.has_matrix2 && !c.tone_map && c.tf == JXLGPU_TF_PQ;
}

// ---- LDS-ring form (the default).  The register ring above holds 75 window samples + 30 cached extrema
// per lane (183 VGPRs: two waves per SIMD), and a dependent VALU instruction of one wave issues only every
// ~6th slot on gfx950 (tools/pk_probe.hip), so two resident waves of mostly serial arithmetic (25-term sums,
// IEEE division / square-root expansions of the colour transform) leave the SIMD idle more than half of the
// time.  Here the five window rows live in a wave-private LDS ring (row-major, one sample per lane + two
// pad samples per side; the x-2..x+2 neighbours are plain LDS reads at lane offsets: no DPP, no 5x
// replication), a lane keeps only the 25 samples of the channel it is summing, and the kernel fits four to
// five waves per SIMD.  The four phase sums of a sample run as two packed chains: phase xm = 1 uses the
// horizontally flipped kernel, so (w[iy][ix], w[iy][4 - ix]) * (s, s) accumulates (xm = 0, xm = 1) with one
// v_pk_mul_f32 + one v_pk_add_f32 per tap — the reference's mul-then-add, each rounded once, per half.
#ifndef UP2_YM_UNROLL
#define UP2_YM_UNROLL 1
#endif
typedef float uf2 __attribute__((ext_vector_type(2)));
constexpr int RING_STRIDE = 68;  // floats per (slot, channel) row: 2 pad + 64 lanes + 2 pad

template <int COLOR, int WAVES_PER_SIMD>   // COLOR: 0 none, 1 the general op list (color_pixel), 2 the packed HDR chain
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void upsample2_lds_kernel(UpStreamArgs a) {
    __shared__ float ring_all[4][5][3][RING_STRIDE];
    __shared__ cf2 hdr_tab[COLOR == 2 ? HC_COUNT : 1];
    if constexpr (COLOR == 2) {
        if (threadIdx.x < 64) hdr_consts_fill(hdr_tab, a.color, (int)threadIdx.x);
        __syncthreads();
    }
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = blockIdx.x * 4 + wib;
    const int lane = threadIdx.x & 63;
    const int strip = wave % a.strips, seg = wave / a.strips;
    if (seg >= a.segs) return;
    float* const ring = &ring_all[wib][0][0][0];
    const int x = a.wx0 + strip * UW - 2 + lane;
    const int xl = mirror_idx(min(max(x, -a.w), 2 * a.w - 1), a.w);
    const bool store_lane = lane >= 2 && lane < 2 + UW && x < a.wx1;
    const uint32_t x_out_off = (uint32_t)(2 * max(x, 0)) * 4u;
    const int y0 = a.wy0 + seg * a.rows_per_seg, y1 = min(y0 + a.rows_per_seg, a.wy1);
    // weight pairs (xm = 0, xm = 1), uniform: 64-bit scalar register pairs with two different halves
    uf2 wp[5][5];
#pragma unroll
    for (int iy = 0; iy < 5; ++iy)
#pragma unroll
        for (int ix = 0; ix < 5; ++ix) {
            wp[iy][ix] = uf2{a.wq[iy * 5 + ix], a.wq[iy * 5 + 4 - ix]};
            asm volatile("" : "+s"(wp[iy][ix]));
        }
    int slot = 0;   // ring slot that receives row j (wave-uniform)
#pragma unroll 1
    for (int j = y0 - 2; j < y1 + 2; ++j) {
        const int jm = mirror_idx(j, a.h);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            ring[(slot * 3 + c) * RING_STRIDE + 2 + lane] = (a.in[c] + (size_t)(uint32_t)jm * a.in_stride)[xl];
        // the ring is private to the wave and LDS operations of one wave execute in order: compiler fences only
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int newest = slot;
        slot = slot == 4 ? 0 : slot + 1;   // now the oldest row's slot (window row iy = 0) == next row's target
        const int r = j - 2;
        if (r < y0) continue;  // wave-uniform
        float o[2][2][3];      // [ym][xm][c]
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float V[5][5];
            int sl = newest;
#pragma unroll
            for (int iy = 4; iy >= 0; --iy) {   // window row iy = input row r - 2 + iy; the newest slot holds iy = 4
                const float* row = ring + (sl * 3 + c) * RING_STRIDE + lane;
#pragma unroll
                for (int ix = 0; ix < 5; ++ix) V[iy][ix] = row[ix];
                sl = sl == 0 ? 4 : sl - 1;
            }
            // minimum / maximum of the 25 window samples (upsampling.rs:100-113; exact operations: the order is free), three
            // operands per instruction: 12 + 12 instead of 24 + 24 + a canonicalising v_max per sample read from LDS
            float mn, mx;
            {
                float rmn[5], rmx[5];
#pragma unroll
                for (int iy = 0; iy < 5; ++iy) {
                    rmn[iy] = fm_min3(fm_min3(V[iy][0], V[iy][1], V[iy][2]), V[iy][3], V[iy][4]);
                    rmx[iy] = fm_max3(fm_max3(V[iy][0], V[iy][1], V[iy][2]), V[iy][3], V[iy][4]);
                }
                mn = fm_min3(fm_min3(rmn[0], rmn[1], rmn[2]), rmn[3], rmn[4]);
                mx = fm_max3(fm_max3(rmx[0], rmx[1], rmx[2]), rmx[3], rmx[4]);
            }
            uf2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};   // output rows ym = 0, 1 (flip_v: kernel rows reversed)
#pragma unroll
            for (int iy = 0; iy < 5; ++iy)
#pragma unroll
                for (int ix = 0; ix < 5; ++ix) {
                    const uf2 sv = {V[iy][ix], V[iy][ix]};
                    acc0 = acc0 + wp[iy][ix] * sv;
                    acc1 = acc1 + wp[4 - iy][ix] * sv;
                }
            // clamp to [mn, mx].  With every window sample below 2^100 in magnitude (tested wave by wave: then no product or
            // partial sum overflows, the sums are finite and mn <= mx are finite) the clamp is one v_med3_f32 per output; the
            // general form — infinities, NaN, the non-finite-minimum rule — otherwise
            const bool tame = fabsf(mn) < 0x1p100f && fabsf(mx) < 0x1p100f;
            if (__builtin_amdgcn_ballot_w64(!tame) == 0) {
                o[0][0][c] = fm_med3(acc0.x, mn, mx); o[0][1][c] = fm_med3(acc0.y, mn, mx);
                o[1][0][c] = fm_med3(acc1.x, mn, mx); o[1][1][c] = fm_med3(acc1.y, mn, mx);
            } else {
                asm volatile("; general clamp" ::: "memory");
                const bool bad = !isfinite(mn);
                const float nanv = __builtin_nanf("");
#pragma unroll
                for (int ym = 0; ym < 2; ++ym)
#pragma unroll
                    for (int xm = 0; xm < 2; ++xm) {
                        float v = ym ? (xm ? acc1.y : acc1.x) : (xm ? acc0.y : acc0.x);
                        v = v < mn ? mn : v;
                        v = v > mx ? mx : v;
                        o[ym][xm][c] = bad ? nanv : v;
                    }
            }
        }
#pragma unroll UP2_YM_UNROLL
        for (int ym = 0; ym < 2; ++ym) {
            float p0[3], p1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { p0[c] = ym ? o[1][0][c] : o[0][0][c]; p1[c] = ym ? o[1][1][c] : o[0][1][c]; }
            if constexpr (COLOR == 1) {
                color_pixel(a.color, p0);
                color_pixel(a.color, p1);
            } else if constexpr (COLOR == 2) {
                cf2 pv[3] = {cf2{p0[0], p1[0]}, cf2{p0[1], p1[1]}, cf2{p0[2], p1[2]}};
                color_pair_hdr(hdr_tab, pv);
#pragma unroll
                for (int c = 0; c < 3; ++c) { p0[c] = pv[c].x; p1[c] = pv[c].y; }
            }
            if (store_lane) {
                const size_t orow = (size_t)(uint32_t)(2 * r + ym) * a.out_stride;  // uniform
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(a.out[c] + orow) + x_out_off);
                    *reinterpret_cast<float2*>(p) = make_float2(p0[c], p1[c]);
                }
            }
        }
    }
}

}  // namespace

// 2x upsampling of three planes (+ the colour transform when `color` is non-null) in one launch.
// Returns false when the streaming form does not apply (tiny frames: the reference's padding has
// its own behaviour below 2 samples, kept by the stage-at-a-time kernel).
// `variant`: 0 = LDS-ring kernel (packed colour chain for the HDR PQ op list), 1 = register-ring kernel, 2 = LDS-ring kernel
// with the general colour code only; `rows`: rows per wave segment, 0 = sized from the
// chip's resident wave slots (a fixed 64 rows made a 4K input 2176 waves on the 2048 slots of the register-ring
// kernel: a second, nearly empty round doubled the launch time).
bool launch_upsample2_stream(hipStream_t s, const float* const in[3], uint32_t in_stride, uint32_t w, uint32_t h,
                             float* const out[3], uint32_t out_stride, const float* weights_quarter_host,
                             const ColorArgs* color, const PixRect* window, uint32_t num_cus, int variant, int rows) {
    if (w < 8 || h < 8) return false;
    UpStreamArgs a;
    memset(&a, 0, sizeof(a));
    for (int c = 0; c < 3; ++c) { a.in[c] = in[c]; a.out[c] = out[c]; }
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.w = (int)w; a.h = (int)h;
    memcpy(a.wq, weights_quarter_host, sizeof(a.wq));
    if (color) { a.color = *color; a.do_color = 1; }
    a.wx0 = window ? window->x0 : 0; a.wy0 = window ? window->y0 : 0;
    a.wx1 = window ? window->x1 : (int)w; a.wy1 = window ? window->y1 : (int)h;
    if (a.wx1 <= a.wx0 || a.wy1 <= a.wy0) return true;
    a.strips = (int)ceil_div((uint32_t)(a.wx1 - a.wx0), UW);
    const uint32_t nrows = (uint32_t)(a.wy1 - a.wy0);
    const uint32_t waves_per_simd = variant == 1 ? 2u : (color ? 4u : 5u);
    if (rows <= 0) {
        const uint32_t slots = num_cus * 4u * waves_per_simd;
        if (variant == 1) {
            // register-ring kernel: ONE resident round (every wave runs equally long; its run-in rows are full-price)
            const uint32_t segs_fit = std::max(1u, slots / (uint32_t)a.strips);
            rows = (int)std::max(16u, ceil_div(nrows, segs_fit));
        } else {
            // LDS-ring kernel: a run-in row is three loads and three LDS writes, so short segments are cheap and about
            // four rounds of waves keep every SIMD at its resident limit to the end (measured on a 4K input, rows per
            // segment 34 / 24 / 17 / 12 / 8: 0.78 / 0.78 / 0.74 / 0.75 / 0.73 ms for the whole post stage)
            const uint32_t segs_want = std::max(1u, 4u * slots / (uint32_t)a.strips);
            rows = (int)std::max(8u, ceil_div(nrows, segs_want));
        }
    }
    a.rows_per_seg = rows;
    a.segs = (int)ceil_div(nrows, (uint32_t)a.rows_per_seg);
    const int waves = a.strips * a.segs;
    const dim3 grid((waves + 3) / 4);
    if (variant == 1) {
        if (color) upsample2_stream_kernel<true><<<grid, 256, 0, s>>>(a);
        else upsample2_stream_kernel<false><<<grid, 256, 0, s>>>(a);
    } else {
        if (color && variant != 2 && is_hdr_pq_list(a.color)) upsample2_lds_kernel<2, 4><<<grid, 256, 0, s>>>(a);
        else if (color) upsample2_lds_kernel<1, 4><<<grid, 256, 0, s>>>(a);
        else upsample2_lds_kernel<0, 5><<<grid, 256, 0, s>>>(a);
    }
    return true;
}
