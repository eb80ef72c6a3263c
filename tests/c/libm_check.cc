// CPU check of jxl-oxide_amd/csrc/libm_f32.h (the header the device code includes) against the installed libm.
//   libm_check logf              every float (all 2^32 bit patterns); NaNs compared as NaNs
//   libm_check powf <y> [...]    every float x for each exponent y
// Prints "<what> mismatches: <n>" per run; exit status 1 if any.  Test infrastructure (tests/test_libm_f32.py).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../jxl-oxide_amd/csrc/libm_f32.h"

static bool same(float a, float b) {
    if (a != a || b != b) return (a != a) && (b != b);
    return libm_f32::f2u(a) == libm_f32::f2u(b);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    long total = 0;
    if (!strcmp(argv[1], "logf")) {
        long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
        for (int64_t u = 0; u < (1ll << 32); ++u) {
            float x = libm_f32::u2f((uint32_t)u);
            float a = ::logf(x), b = libm_f32::logf(x);
            if (!same(a, b)) {
                if (bad < 3) fprintf(stderr, "logf(%a): libm %a, restated %a\n", x, a, b);
                ++bad;
            }
        }
        printf("logf mismatches: %ld\n", bad);
        total += bad;
    } else if (!strcmp(argv[1], "powf")) {
        for (int j = 2; j < argc; ++j) {
            volatile float yv = strtof(argv[j], nullptr);
            const float y = yv;
            long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
            for (int64_t u = 0; u < (1ll << 32); ++u) {
                float x = libm_f32::u2f((uint32_t)u);
                float a = ::powf(x, y), b = libm_f32::powf(x, y);
                if (!same(a, b)) {
                    if (bad < 3) fprintf(stderr, "powf(%a, %a): libm %a, restated %a\n", x, y, a, b);
                    ++bad;
                }
            }
            printf("powf y=%a mismatches: %ld\n", y, bad);
            total += bad;
        }
    }
    return total ? 1 : 0;
}
