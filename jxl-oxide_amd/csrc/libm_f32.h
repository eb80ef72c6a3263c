// logf / powf exactly as the reference's libm computes them.
//
// jxl-oxide's HLG path is the one place on the render path that calls the platform libm per sample
// (`mixed.powf(exp)` in hlg_inverse_oo, `(..).ln()` in linear_to_hlg: jxl-color/src/tf.rs:118-160).  Rust's
// f32::powf / f32::ln lower to the libm symbols powf / logf; on the reference's platform (x86_64-unknown-linux-gnu)
// that is glibc, whose single-precision functions have been the ARM "optimized-routines" algorithms since
// glibc 2.28 (sysdeps/ieee754/flt-32/e_logf.c, e_powf.c, e_exp2f_data.c, e_logf_data.c, e_powf_log2_data.c; the
// container and the GPU box run glibc 2.35): table-driven, evaluated in double precision, one rounding to float at
// the end.  libm is a dependency that is not under /root/reference, so this file restates the PUBLISHED algorithm
// (tables and polynomials are the published constants; tools/libm_tables.py re-reads them from the installed
// libm.so.6 and checks this file) — and tests/test_libm_f32.py compiles this very header with g++ and compares it
// with the installed libm: logf on EVERY float (all 2^32 bit patterns), powf on every float as the base for the
// exponents the HLG system gamma takes (188 more exponents x 2^32 bases were run once: no mismatch).  The arithmetic
// is IEEE double add / mul / fma and exact integer steps only, every fused multiply-add written out
// (-ffp-contract=off on both compilers), so the device evaluates bit for bit what the host test evaluates.
//
// Scope: what the HLG path can hand in.  logf: any float (the caller hands 12a - b with a > 1/12, NaN or inf).
// powf: any x, FINITE y (the exponent is a frame constant the host derives from intensity_target).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LIBM_F32_FN __host__ __device__ __forceinline__
#else
#define LIBM_F32_FN static inline
#endif

namespace libm_f32 {

LIBM_F32_FN uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
LIBM_F32_FN float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
LIBM_F32_FN uint64_t d2u(double f) { return __builtin_bit_cast(uint64_t, f); }
LIBM_F32_FN double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// x86 default NaN (what (x - x) / (x - x) and 0 * inf produce there); the device's own default NaN has the sign
// clear.  Parity tests compare NaN positions, not NaN payloads; producing the x86 pattern costs nothing.
LIBM_F32_FN float invalid_nan() { return u2f(0xffc00000u); }

// a * b + c with ONE rounding.  glibc selects its -mfma -mavx2 builds of these functions (sysdeps/x86_64/fpu/multiarch/
// e_logf-fma.c, e_powf-fma.c) on every x86 CPU with FMA3, i.e. on any host of the last decade, and GCC contracts every
// multiply-add of the source there.  For logf the float result is the same with or without contraction (checked on every
// float); for powf the contracted  r = z * invc - 1  decides a handful of results per 2^32 (8 of 11 x 2^32 checked).
LIBM_F32_FN double madd(double a, double b, double c) { return __builtin_fma(a, b, c); }

// e_logf_data.c: {1/c, log(c)} for the 16 subintervals of [OFF, 2 OFF), OFF = 0x3f330000
LIBM_F32_FN void logf_entry(int i, double& invc, double& logc) {
    static const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    invc = T[i][0];
    logc = T[i][1];
}

// e_powf_log2_data.c: {1/c, log2(c)}
LIBM_F32_FN void powf_log2_entry(int i, double& invc, double& logc) {
    static const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
        {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
        {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
        {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
        {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    invc = T[i][0];
    logc = T[i][1];
}

// e_exp2f_data.c: tab[i] = asuint64(2^(i/32)) - (i << 47)
LIBM_F32_FN uint64_t exp2f_entry(int i) {
    static const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    return T[i];
}

// e_logf.c
LIBM_F32_FN float logf(float x) {
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000u) return 0.0f;                      // log(1) = +0 in every rounding mode
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {     // x < 0x1p-126, inf or nan
        if (ix * 2u == 0u) return -__builtin_inff();         // __math_divzerof (1)
        if (ix == 0x7f800000u) return x;                     // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return x != x ? x : invalid_nan();
        ix = f2u(x * 0x1p23f);                               // subnormal: normalise
        ix -= 23u << 23;
    }
    // x = 2^k z, z in [OFF, 2 OFF), exact; i = the subinterval of z
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    double invc, logc;
    logf_entry(i, invc, logc);
    const double z = (double)u2f(iz);
    // log(x) = log1p(z/c - 1) + log(c) + k ln2
    const double r = madd(z, invc, -1.0);
    const double y0 = madd((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = madd(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = madd(-0x1.00ea348b88334p-2, r2, y);
    y = madd(y, r2, y0 + r);
    return (float)y;
}

// e_powf.c: checkint — 0: y is not an integer, 1: odd, 2: even
LIBM_F32_FN int powf_checkint(uint32_t iy) {
    const int e = (int)(iy >> 23 & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}

// e_powf.c, y finite
LIBM_F32_FN float powf(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = f2u(x);
    const uint32_t iy = f2u(y);
    if (iy * 2u == 0u) return 1.0f;                          // pow(x, +-0) = 1 (x is never signalling here)
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {     // x < 0x1p-126, inf or nan
        if (ix * 2u - 1u >= 2u * 0x7f800000u - 1u) {         // zeroinfnan(ix)
            float x2 = x * x;                                // nan stays nan
            if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
            if (ix * 2u == 0u && (iy & 0x80000000u))         // __math_divzerof: +-1 / 0
                return ((ix & 0x80000000u) && powf_checkint(iy) == 1) ? -__builtin_inff() : __builtin_inff();
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {                              // finite x < 0
            const int yint = powf_checkint(iy);
            if (yint == 0) return invalid_nan();             // __math_invalidf
            if (yint == 1) sign_bias = 1u << 16;             // SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11): bit 63 after << 47
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                              // subnormal: normalise
            ix = f2u(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    double invc, logc;
    powf_log2_entry(i, invc, logc);
    const double z = (double)u2f(iz);
    // log2(x) = log1p(z/c - 1) / ln2 + log2(c) + k
    const double r = madd(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double p0 = madd(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p1 = madd(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = madd(0x1.71547652ab82bp+0, r, y0);
    q = madd(p1, r2, q);
    p0 = madd(p0, r4, q);
    const double logx = p0;
    const double ylogx = (double)y * logx;                   // cannot overflow: y is single precision
    if ((d2u(ylogx) >> 47 & 0xffffu) >= (d2u(126.0) >> 47)) {  // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();  // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                                        // __math_uflowf
    }
    // exp2_inline: x = k/N + r, r in [-1/(2N), 1/(2N)], N = 32
    double kd = ylogx + 0x1.8p+47;                           // SHIFT = 0x1.8p+52 / N
    const uint64_t ki = d2u(kd);
    kd -= 0x1.8p+47;
    const double rr = ylogx - kd;
    uint64_t t = exp2f_entry((int)(ki & 31u));
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = u2d(t);
    const double zz = madd(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double yy = madd(0x1.62e42ff0c52d6p-1, rr, 1.0);
    yy = madd(zz, rr2, yy);
    yy = yy * s;
    return (float)yy;
}

}  // namespace libm_f32
