/* Test-input generation only (used by synth_modular.py through ctypes; built on demand with gcc).
 *
 * Encoder-side weighted ("self-correcting") predictor of ONE tile: residual(x, y) = v - ((pred + 3) >> 3).
 * This is synth_modular.weighted_residuals statement by statement — JPEG XL 18181-1 H.5 written with full
 * 2-D error arrays and explicit neighbour fall-back rules on the TRUE samples, not the decoder's running
 * state (jxl-modular/src/predictor.rs:312-441), so that decode(encode(x)) == x pins the decoder against an
 * independent formulation.  tests/test_oracle_modular.py checks this file against the Python function.
 */
#include <stdint.h>
#include <stdlib.h>

static int bit_length_u64(uint64_t v) {
    int n = 0;
    while (v) { ++n; v >>= 1; }
    return n;
}

void synth_wp_residuals_tile(const int64_t* img, size_t stride, int w, int h, int64_t* out, size_t out_stride,
                             const int32_t wp[11]) {
    const int64_t p1 = wp[0], p2 = wp[1], p3a = wp[2], p3b = wp[3], p3c = wp[4], p3d = wp[5], p3e = wp[6];
    int64_t div[65];
    div[0] = 0;
    for (int i = 1; i <= 64; ++i) div[i] = (1 << 24) / i;
    int64_t* terr = (int64_t*)calloc((size_t)w * h, sizeof(int64_t));
    int64_t* serr = (int64_t*)calloc((size_t)w * h * 4, sizeof(int64_t));
#define S(xx, yy) img[(size_t)(yy) * stride + (xx)]
#define TE(xx, yy) terr[(size_t)(yy) * w + (xx)]
#define SE(xx, yy, i) serr[((size_t)(yy) * w + (xx)) * 4 + (i)]
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int64_t Wv = x > 0 ? S(x - 1, y) : (y > 0 ? S(x, y - 1) : 0);
            const int64_t N = y > 0 ? S(x, y - 1) : Wv;
            const int64_t NW = (x > 0 && y > 0) ? S(x - 1, y - 1) : Wv;
            const int64_t NE = (x + 1 < w && y > 0) ? S(x + 1, y - 1) : N;
            const int64_t NN = y > 1 ? S(x, y - 2) : N;
            /* error neighbours: zero outside the tile, except that NW / NE fall back onto N */
            const int64_t te_w = x > 0 ? TE(x - 1, y) : 0;
            const int64_t te_n = y > 0 ? TE(x, y - 1) : 0;
            const int64_t te_nw = (x > 0 && y > 0) ? TE(x - 1, y - 1) : te_n;
            const int64_t te_ne = (x + 1 < w && y > 0) ? TE(x + 1, y - 1) : te_n;
            const int64_t n3 = N * 8, nw3 = NW * 8, ne3 = NE * 8, w3 = Wv * 8, nn3 = NN * 8;
            int64_t sub[4];
            sub[0] = w3 + ne3 - n3;
            sub[1] = n3 - (((te_w + te_n + te_ne) * p1) >> 5);
            sub[2] = w3 - (((te_w + te_n + te_nw) * p2) >> 5);
            sub[3] = n3 - ((te_nw * p3a + te_n * p3b + te_ne * p3c + (nn3 - n3) * p3d + (nw3 - w3) * p3e) >> 5);
            int64_t weight[4], wsum = 0;
            for (int i = 0; i < 4; ++i) {
                const int64_t e_n = y > 0 ? SE(x, y - 1, i) : 0;
                const int64_t e_w = x > 0 ? SE(x - 1, y, i) : 0;
                const int64_t e_ww = x > 1 ? SE(x - 2, y, i) : 0;
                const int64_t e_nw = (x > 0 && y > 0) ? SE(x - 1, y - 1, i) : e_n;
                const int64_t e_ne = (x + 1 < w && y > 0) ? SE(x + 1, y - 1, i) : e_n;
                uint64_t es = (uint64_t)(e_n + e_w + e_ww + e_nw + e_ne) & 0xFFFFFFFFull;
                if (x + 1 == w && x > 0) es = (es + (uint64_t)e_w) & 0xFFFFFFFFull; /* last column: NE folds onto N, which carries W */
                int shift = bit_length_u64((es + 1) >> 5) - 1;
                if (shift < 0) shift = 0;
                weight[i] = 4 + (((int64_t)wp[7 + i] * div[(es >> shift) + 1]) >> shift);
                wsum += weight[i];
            }
            const int lw = bit_length_u64((uint64_t)(wsum >> 4)) - 1;
            int64_t sw = 0, acc;
            for (int i = 0; i < 4; ++i) { weight[i] >>= lw; sw += weight[i]; }
            acc = (sw >> 1) - 1;
            for (int i = 0; i < 4; ++i) acc += sub[i] * weight[i];
            int64_t pred = (acc * div[sw]) >> 24;
            if (((te_n ^ te_w) | (te_n ^ te_nw)) <= 0) {
                int64_t mn = n3 < w3 ? n3 : w3, mx = n3 > w3 ? n3 : w3;
                if (ne3 < mn) mn = ne3;
                if (ne3 > mx) mx = ne3;
                pred = pred < mn ? mn : (pred > mx ? mx : pred);
            }
            const int64_t v = S(x, y);
            out[(size_t)y * out_stride + x] = v - ((pred + 3) >> 3);
            TE(x, y) = pred - (v * 8);
            for (int i = 0; i < 4; ++i) {
                int64_t d = sub[i] - (v * 8);
                if (d < 0) d = -d;
                SE(x, y, i) = (d + 3) >> 3;
            }
        }
#undef S
#undef TE
#undef SE
    free(terr);
    free(serr);
}
