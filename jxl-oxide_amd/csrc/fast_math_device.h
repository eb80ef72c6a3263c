// Correctly rounded f32 square root and quotient without the compiler's IEEE expansions (v_div_scale / v_div_fmas /
// v_div_fixup, the scaled two-sided v_sqrt correction: ~15 instructions each, most of them of the 4-cycle class,
// profiles/r06_valu_cost_probe.txt), valid on the stated ranges only — the callers test the range wave by wave and keep
// the ordinary `sqrtf` / `/` for everything else.  Pairs (two f32 per lane, v_pk_*_f32) because the HDR colour chain of
// upsample_kernels.hip works on pairs.  Round 6; the reference arithmetic is jxl-color/src/tf/pq.rs:127-142 (`sqrt().sqrt()`,
// `yp / yq`) and gamut.rs:40-45 (three quotients by one denominator).
//
//   sqrt_cr_pair   2^-60 <= x <= 2^60: y = rsq(x); g = x y; h = y / 2; r = fma(-h, g, 1/2); g = fma(g, r, g); h = fma(h, r, h);
//                  d = fma(-g, g, x); result = fma(d, h, g) — the residual step on a root good to ~2^-44 relative, the
//                  classic correctly rounded finish (a square root is never a rounding midpoint).  Checked against the
//                  compiler's sqrtf ON THE DEVICE for every float of the range (v_rsq_f32 cannot be emulated on the host):
//                  tests/c/fast_math_check.hip, tests/test_gpu_fast_math.py.
//   rcp_cr_pair    2^-60 <= |d| <= 2^60: see below; every float of the range checked on the device.
//   div_cr_pair    1 <= d <= 2^20 and (2^-100 <= |n| <= 2^20 or n = +-0): the chain of div3_shared (post_pk.inc), whose
//                  proof of correct rounding for ANY 1-ulp reciprocal is checked on the host by tests/c/sdiv_check.c; the
//                  device check above runs it on 2^31 random pairs as well.
#pragma once
#include <hip/hip_runtime.h>

typedef float fm2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ fm2 fm_fma(fm2 a, fm2 b, fm2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ fm2 sqrt_cr_pair(fm2 x) {
    const fm2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const fm2 half = {0.5f, 0.5f};
    fm2 g = x * y;
    fm2 h = y * half;
    const fm2 r = fm_fma(-h, g, half);
    g = fm_fma(g, r, g);
    h = fm_fma(h, r, h);
    const fm2 d = fm_fma(-g, g, x);
    return fm_fma(d, h, g);
}
__device__ __forceinline__ float sqrt_cr(float x) {   // the same chain on one value (tests)
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = y * 0.5f;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// the refined reciprocal the quotient chains share
__device__ __forceinline__ fm2 rcp_refined_pair(fm2 d) {
    const fm2 one = {1.0f, 1.0f};
    fm2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const fm2 e = fm_fma(-d, r, one);
    return fm_fma(e, r, r);
}
__device__ __forceinline__ fm2 div_cr_pair_with(fm2 n, fm2 d, fm2 r) {
    const fm2 q0 = n * r;
    const fm2 t1 = fm_fma(d, q0, -n);
    const fm2 q1 = fm_fma(-t1, r, q0);
    const fm2 t2 = fm_fma(d, q1, -n);
    return fm_fma(-t2, r, q1);
}
__device__ __forceinline__ fm2 div_cr_pair(fm2 n, fm2 d) { return div_cr_pair_with(n, d, rcp_refined_pair(d)); }
// correctly rounded 1 / d for 2^-60 <= |d| <= 2^60 (either sign): r = rcp(d); r = fma(fma(-d, r, 1), r, r); result = fma(fma(-d, r, 1), r, r)
// — the second step is the residual correction of a reciprocal already good to ~2^-45 (checked on the device for every float of
// the range, both signs: tests/c/fast_math_check.hip)
__device__ __forceinline__ fm2 rcp_cr_pair(fm2 d) {
    const fm2 one = {1.0f, 1.0f};
    const fm2 r = rcp_refined_pair(d);
    return fm_fma(fm_fma(-d, r, one), r, r);
}
__device__ __forceinline__ float rcp_cr(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
// three-operand minimum / maximum / median: one instruction each (the compiler's fminf / fmaxf chains on values read from
// memory carry a canonicalising v_max_f32 x, x per operand in IEEE mode; like f32::min / f32::max they return the other
// operand when one is a NaN)
__device__ __forceinline__ float fm_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fm_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fm_min3_abs(float a, float b, float c) { float r; asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fm_max3_abs(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fm_med3(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float div_cr(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float q0 = n * r;
    const float t1 = __builtin_fmaf(d, q0, -n);
    const float q1 = __builtin_fmaf(-t1, r, q0);
    const float t2 = __builtin_fmaf(d, q1, -n);
    return __builtin_fmaf(-t2, r, q1);
}

// range tests on the bit patterns (one subtraction + one unsigned compare per value; a NaN fails both)
__device__ __forceinline__ bool fm_in_range_bits(float v, uint32_t lo_bits, uint32_t hi_bits) {   // lo <= v <= hi for positive finite lo, hi
    return __float_as_uint(v) - lo_bits <= hi_bits - lo_bits;
}
constexpr uint32_t kFmBits2m100 = (127u - 100u) << 23, kFmBits2m60 = (127u - 60u) << 23, kFmBits1 = 127u << 23,
                   kFmBits2p12 = (127u + 12u) << 23, kFmBits2p20 = (127u + 20u) << 23, kFmBits2p60 = (127u + 60u) << 23;
