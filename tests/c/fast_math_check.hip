// Device check of jxl-oxide_amd/csrc/fast_math_device.h (round 6): v_rsq_f32 / v_rcp_f32 cannot be emulated on the host, so the
// claim "bit-identical to the compiler's correctly rounded sqrtf and division on the stated ranges" is checked where it
// runs.  sqrt_cr: EVERY float in [2^-60, 2^60]; sqrt_cr(sqrt_cr(x)) against sqrtf(sqrtf(x)) on [2^-60, 2^12] (what
// linear_to_pq evaluates); div_cr: 2^31 pseudo-random pairs with d in [1, 2^20], |n| in [2^-100, 2^20] (half of the
// denominators in [1, 256)), the edges of both ranges, exact zeros of either sign; rcp_cr: every float with |d| in [2^-60, 2^60].  Prints the mismatch counts; exit 1 on any.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I jxl-oxide_amd/csrc tests/c/fast_math_check.hip -o tools/_bin/fast_math_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "fast_math_device.h"

__global__ void check_sqrt(uint32_t lo_bits, uint32_t count, unsigned long long* bad, uint32_t* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        const float want = sqrtf(x);                  // -fhip-fp32-correctly-rounded-divide-sqrt: the IEEE expansion
        const float got = sqrt_cr(x);
        const fm2 gp = sqrt_cr_pair(fm2{x, x});
        if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(gp.x) != __float_as_uint(want) || __float_as_uint(gp.y) != __float_as_uint(want)) {
            if (atomicAdd(bad, 1ull) == 0) *first_bad = __float_as_uint(x);
        }
    }
}
__global__ void check_sqrt_sqrt(uint32_t lo_bits, uint32_t count, unsigned long long* bad, uint32_t* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        const float want = sqrtf(sqrtf(x));
        const fm2 gp = sqrt_cr_pair(sqrt_cr_pair(fm2{x, x}));
        if (__float_as_uint(gp.x) != __float_as_uint(want) || __float_as_uint(gp.y) != __float_as_uint(want)) {
            if (atomicAdd(bad, 1ull) == 0) *first_bad = __float_as_uint(x);
        }
    }
}
__global__ void check_rcp(uint32_t lo_bits, uint32_t count, unsigned long long* bad, uint32_t* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2ull * count; i += stride) {
        const float d = __uint_as_float((lo_bits + (uint32_t)(i >> 1)) | ((uint32_t)(i & 1) << 31));
        const float want = 1.0f / d;
        const fm2 gp = rcp_cr_pair(fm2{d, d});
        if (__float_as_uint(want) != __float_as_uint(rcp_cr(d)) || __float_as_uint(gp.x) != __float_as_uint(want) || __float_as_uint(gp.y) != __float_as_uint(want)) {
            if (atomicAdd(bad, 1ull) == 0) *first_bad = __float_as_uint(d);
        }
    }
}
__device__ __forceinline__ uint64_t xs(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__global__ void check_div(uint64_t per_thread, unsigned long long* bad, uint32_t* first_bad) {
    uint64_t s = 88172645463325252ull ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1) * 0x9E3779B97F4A7C15ull);
    for (uint64_t it = 0; it < per_thread; ++it) {
        const uint64_t a = xs(s), b = xs(s);
        const uint32_t ed = 127u + ((it & 1) ? (uint32_t)(a % 8) : (uint32_t)(a % 21));
        float d = __uint_as_float((ed << 23) | (uint32_t)((a >> 8) & 0x7fffffu));
        if (ed == 147u) d = 0x1p20f;                               // the upper edge itself
        const uint32_t en = 27u + (uint32_t)(b % 121);             // 2^-100 .. 2^20
        float n = __uint_as_float(((uint32_t)(b >> 63) << 31) | (en << 23) | (uint32_t)((b >> 8) & 0x7fffffu));
        if (fabsf(n) > 0x1p20f) n = copysignf(0x1p20f, n);
        if ((it & 1023) == 5) n = copysignf(0.0f, n);              // exact zeros of either sign
        if ((it & 1023) == 6) n = copysignf(0x1p-100f, n);
        if ((it & 1023) == 7) d = 1.0f;
        const float want = n / d;
        const float got = div_cr(n, d);
        const fm2 gp = div_cr_pair(fm2{n, -n}, fm2{d, d});
        const float wneg = (-n) / d;
        if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(gp.x) != __float_as_uint(want) || __float_as_uint(gp.y) != __float_as_uint(wneg)) {
            if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = __float_as_uint(n); first_bad[1] = __float_as_uint(d); }
        }
    }
}

int main() {
    unsigned long long* bad;
    uint32_t* first;
    if (hipMalloc(&bad, 4 * sizeof(*bad)) != hipSuccess || hipMalloc(&first, 5 * sizeof(*first)) != hipSuccess) { fprintf(stderr, "no device\n"); return 3; }
    (void)hipMemset(bad, 0, 4 * sizeof(*bad));
    (void)hipMemset(first, 0, 5 * sizeof(*first));
    const uint32_t lo = kFmBits2m60, hi = kFmBits2p60, hi12 = kFmBits2p12;
    check_sqrt<<<4096, 256>>>(lo, hi - lo + 1u, bad, first);
    check_sqrt_sqrt<<<4096, 256>>>(lo, hi12 - lo + 1u, bad + 1, first + 1);
    const uint64_t per_thread = (1ull << 31) / (4096ull * 256ull);
    check_div<<<4096, 256>>>(per_thread, bad + 2, first + 2);
    check_rcp<<<4096, 256>>>(lo, hi - lo + 1u, bad + 3, first + 4);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    unsigned long long hb[4];
    uint32_t hf[5];
    (void)hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost);
    printf("sqrt_cr: %llu values in [2^-60, 2^60], mismatches %llu (first x bits 0x%08x)\n", (unsigned long long)(hi - lo) + 1, hb[0], hf[0]);
    printf("sqrt_cr(sqrt_cr): %llu values in [2^-60, 2^12], mismatches %llu (first x bits 0x%08x)\n", (unsigned long long)(hi12 - lo) + 1, hb[1], hf[1]);
    printf("div_cr: %llu pairs, mismatches %llu (first n bits 0x%08x d bits 0x%08x)\n", per_thread * 4096ull * 256ull, hb[2], hf[2], hf[3]);
    printf("rcp_cr: %llu values, |d| in [2^-60, 2^60], both signs, mismatches %llu (first d bits 0x%08x)\n", 2ull * ((unsigned long long)(hi - lo) + 1), hb[3], hf[4]);
    return (hb[0] || hb[1] || hb[2] || hb[3]) ? 1 : 0;
}
