/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors);
 * tests/test_oracle_modular.py inverts an independent numpy forward predictor for every kind
 * (residual = sample - prediction computed from the finished image) and checks the
 * self-correcting predictor against a straight Python transcription on small tiles.
 *
 * M4 where it is separable from the entropy decode: a single-leaf MA tree, so the decoder's
 * `decode_single_node` paths (jxl-modular/src/image.rs:716-777, :880-949) reduce to
 *     sample = residual * multiplier + offset + predict(neighbours)        (decode_one, :878-890)
 * with the neighbour state machine of PredictorState / Properties::record.  Kept in the reference's
 * structure (prev_row / curr_row vectors of i32, w / n / nw registers, the self-correcting
 * predictor's error rows), one tile = one Modular group channel.
 *   Predictor::predict          predictor.rs:79-125
 *   PredictorState nn/ne/nee/ww predictor.rs:226-273
 *   Properties::record          predictor.rs:540-577
 *   SelfCorrectingPredictor     predictor.rs:275-442 (predict :312-390, record :394-441)
 *   DIV_LOOKUP                  predictor.rs:150-160
 *   Sample::wrapping_muladd_i32 / add   sample.rs:119-126 (i32), :169-176 (i16)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct {
    uint32_t width, x, y;
    int32_t* true_err_row;
    uint32_t (*subpred_err_row)[4];
    int32_t p1, p2, p3[5];
    uint32_t wn[4];
    int32_t true_err_w, true_err_nw, true_err_n, true_err_ne;
    uint32_t subpred_err_nw_ww[4], subpred_err_n_w[4], subpred_err_ne[4];
} ScPred;

typedef struct {
    int64_t prediction;
    int64_t subpred[4];
} ScResult;

static uint32_t div_lookup(uint32_t i) { return i == 0 ? 0u : (uint32_t)((1u << 24) / i); }

static uint32_t ilog2_u64(uint64_t v) { return 63u - (uint32_t)__builtin_clzll(v); }

/* predictor.rs:312-390 */
static ScResult sc_predict(const ScPred* sc, int32_t n, int32_t nw, int32_t ne, int32_t w, int32_t nn) {
    int64_t true_err_w = sc->true_err_w, true_err_nw = sc->true_err_nw;
    int64_t true_err_n = sc->true_err_n, true_err_ne = sc->true_err_ne;
    int64_t n3 = (int64_t)n * 8, nw3 = (int64_t)nw * 8, ne3 = (int64_t)ne * 8, w3 = (int64_t)w * 8,
            nn3 = (int64_t)nn * 8;
    ScResult r;
    r.subpred[0] = w3 + ne3 - n3;
    r.subpred[1] = n3 - (((true_err_w + true_err_n + true_err_ne) * (int64_t)sc->p1) >> 5);
    r.subpred[2] = w3 - (((true_err_w + true_err_n + true_err_nw) * (int64_t)sc->p2) >> 5);
    r.subpred[3] = n3 - ((true_err_nw * (int64_t)sc->p3[0] + true_err_n * (int64_t)sc->p3[1] +
                          true_err_ne * (int64_t)sc->p3[2] + (nn3 - n3) * (int64_t)sc->p3[3] +
                          (nw3 - w3) * (int64_t)sc->p3[4]) >> 5);
    uint32_t weight[4];
    for (int i = 0; i < 4; ++i) {
        uint32_t err_sum = sc->subpred_err_nw_ww[i] + sc->subpred_err_n_w[i] + sc->subpred_err_ne[i];
        uint64_t t = ((uint64_t)err_sum + 1) >> 5;
        uint32_t shift = t ? ilog2_u64(t) : 0;
        weight[i] = 4 + ((sc->wn[i] * div_lookup((err_sum >> shift) + 1)) >> shift);
    }
    uint32_t sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
    uint32_t log_weight = ilog2_u64((uint64_t)sum_weights >> 4);
    for (int i = 0; i < 4; ++i) weight[i] >>= log_weight;
    sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
    int64_t s = ((int64_t)sum_weights >> 1) - 1;
    for (int i = 0; i < 4; ++i) s += r.subpred[i] * (int64_t)weight[i];
    int64_t prediction = (s * (int64_t)div_lookup(sum_weights)) >> 24;
    if (((true_err_n ^ true_err_w) | (true_err_n ^ true_err_nw)) <= 0) {
        int64_t mn = n3 < w3 ? n3 : w3; if (ne3 < mn) mn = ne3;
        int64_t mx = n3 > w3 ? n3 : w3; if (ne3 > mx) mx = ne3;
        if (prediction < mn) prediction = mn;
        if (prediction > mx) prediction = mx;
    }
    r.prediction = prediction;
    return r;
}

/* predictor.rs:394-441 */
static void sc_record(ScPred* sc, const ScResult* pred, int32_t sample_i32) {
    int64_t sample = sample_i32;
    int64_t true_err = pred->prediction - sample * 8;
    uint32_t subpred_err[4];
    for (int i = 0; i < 4; ++i) {
        int64_t d = pred->subpred[i] - sample * 8;
        uint64_t ad = d < 0 ? (uint64_t)(-d) : (uint64_t)d;
        subpred_err[i] = (uint32_t)((ad + 3) >> 3);
    }
    sc->true_err_row[sc->x] = (int32_t)true_err;
    memcpy(sc->subpred_err_row[sc->x], subpred_err, sizeof(subpred_err));
    sc->x += 1;
    if (sc->x >= sc->width) {
        sc->y += 1;
        sc->x = 0;
        sc->true_err_w = 0;
        sc->true_err_n = sc->true_err_row[0];
        sc->true_err_nw = sc->true_err_n;
        memcpy(sc->subpred_err_n_w, sc->subpred_err_row[0], 16);
        memcpy(sc->subpred_err_nw_ww, sc->subpred_err_n_w, 16);
        if (sc->width <= 1) {
            sc->true_err_ne = sc->true_err_n;
            memcpy(sc->subpred_err_ne, sc->subpred_err_n_w, 16);
        } else {
            sc->true_err_ne = sc->true_err_row[1];
            memcpy(sc->subpred_err_ne, sc->subpred_err_row[1], 16);
        }
    } else {
        sc->true_err_w = (int32_t)true_err;
        sc->true_err_nw = sc->true_err_n;
        sc->true_err_n = sc->true_err_ne;
        memcpy(sc->subpred_err_nw_ww, sc->subpred_err_n_w, 16);
        memcpy(sc->subpred_err_n_w, sc->subpred_err_ne, 16);
        for (int i = 0; i < 4; ++i) sc->subpred_err_n_w[i] += subpred_err[i];
        if (sc->x + 1 >= sc->width) {
            sc->true_err_ne = sc->true_err_n;
            memcpy(sc->subpred_err_ne, sc->subpred_err_n_w, 16);
        } else if (sc->y != 0) {
            sc->true_err_ne = sc->true_err_row[sc->x + 1];
            memcpy(sc->subpred_err_ne, sc->subpred_err_row[sc->x + 1], 16);
        }
    }
}

typedef struct {
    uint32_t width, x, y;
    int32_t *prev_row, *curr_row;
    size_t prev_len, curr_len;   /* Vec lengths: rows fill up as samples are recorded */
    int32_t w, n, nw;
} PState;

static int32_t ps_nn(const PState* p) { return p->x < p->curr_len ? p->curr_row[p->x] : p->n; }
static int32_t ps_ne(const PState* p) {
    return (p->prev_len == 0 || p->x + 1 >= p->width) ? p->n : p->prev_row[p->x + 1];
}
static int32_t ps_nee(const PState* p) {
    return (p->prev_len == 0 || p->x + 2 >= p->width) ? ps_ne(p) : p->prev_row[p->x + 2];
}
static int32_t ps_ww(const PState* p) { return p->x >= 2 ? p->curr_row[p->x - 2] : p->w; }

/* predictor.rs:79-125 */
static int32_t predict(const PState* p, uint32_t predictor, const ScResult* sc) {
    int64_t n = p->n, w = p->w, nw = p->nw;
    switch (predictor) {
        case 0: return 0;
        case 1: return p->w;
        case 2: return p->n;
        case 3: return (int32_t)((w + n) / 2);
        case 4: {
            uint64_t dn = (uint64_t)(n > nw ? n - nw : nw - n), dw = (uint64_t)(w > nw ? w - nw : nw - w);
            return dn < dw ? p->w : p->n;
        }
        case 5: {
            int64_t g = n + w - nw, lo = w < n ? w : n, hi = w > n ? w : n;
            return (int32_t)(g < lo ? lo : (g > hi ? hi : g));
        }
        case 6: return (int32_t)((sc->prediction + 3) >> 3);
        case 7: return ps_ne(p);
        case 8: return p->nw;
        case 9: return ps_ww(p);
        case 10: return (int32_t)((w + nw) / 2);
        case 11: return (int32_t)((n + nw) / 2);
        case 12: return (int32_t)((n + (int64_t)ps_ne(p)) / 2);
        default: {
            int64_t nn = ps_nn(p), ww = ps_ww(p), nee = ps_nee(p), ne = ps_ne(p);
            return (int32_t)((6 * n - 2 * nn + 7 * w + ww + nee + 3 * ne + 8) / 16);
        }
    }
}

/* Properties::record, predictor.rs:540-577 */
static void ps_record(PState* p, int32_t sample) {
    p->curr_row[p->x] = sample;
    if (p->x >= p->curr_len) p->curr_len = p->x + 1;
    p->x += 1;
    if (p->x >= p->width) {
        p->y += 1;
        p->x = 0;
        int32_t* t = p->prev_row; p->prev_row = p->curr_row; p->curr_row = t;
        size_t tl = p->prev_len; p->prev_len = p->curr_len; p->curr_len = tl;
        int32_t n = p->prev_row[0];
        p->n = n; p->w = n; p->nw = n;
    } else {
        p->w = sample;
        if (p->prev_len == 0) {
            p->nw = sample;
            p->n = sample;
        } else {
            p->nw = p->n;
            p->n = p->prev_row[p->x];
        }
    }
}

/* One tile (a Modular group channel): residuals -> samples in place.  `esz` 2 = i16, 4 = i32.
 * wp = {p1, p2, p3a..p3e, w0..w3} (WpHeader) for predictor 6.
 * `axis`: 0 = one leaf for the tile (leaves[0]); 1 = the leaf of a sample is leaves[y] (a tree that splits on property 2);
 * 2 = leaves[x] (property 3) — decode_slow, image.rs:1169-1228, with get_leaf a function of the row / column alone.
 * One PredictorState per tile either way; the self-correcting predictor's state is kept for EVERY sample as soon as one
 * leaf uses it (PredictorState::reset with the WpHeader when FlatMaTree::need_self_correcting, ma.rs:275-285,
 * image.rs:556-560; properties() then runs it per sample, predictor.rs:206-212). */
void orc_predict_apply_leaves(void* tile, size_t stride, size_t width, size_t height, int esz, int axis,
                              const JxlGpuMaLeaf* leaves, const int32_t wp[11]) {
    if (width == 0 || height == 0) return;
    PState ps;
    memset(&ps, 0, sizeof(ps));
    ps.width = (uint32_t)width;
    ps.prev_row = (int32_t*)calloc(width, 4);
    ps.curr_row = (int32_t*)calloc(width, 4);
    ScPred sc;
    memset(&sc, 0, sizeof(sc));
    const size_t nleaves = axis == 0 ? 1 : (axis == 1 ? height : width);
    int use_sc = 0;
    for (size_t i = 0; i < nleaves; ++i) use_sc |= leaves[i].predictor == 6;
    if (use_sc) {
        sc.width = (uint32_t)width;
        sc.true_err_row = (int32_t*)calloc(width, 4);
        sc.subpred_err_row = (uint32_t(*)[4])calloc(width, 16);
        sc.p1 = wp[0]; sc.p2 = wp[1];
        for (int i = 0; i < 5; ++i) sc.p3[i] = wp[2 + i];
        for (int i = 0; i < 4; ++i) sc.wn[i] = (uint32_t)wp[7 + i];
    }
    for (size_t y = 0; y < height; ++y) {
        for (size_t x = 0; x < width; ++x) {
            const JxlGpuMaLeaf* lf = &leaves[axis == 0 ? 0 : (axis == 1 ? y : x)];
            const uint32_t predictor = lf->predictor;
            const int32_t multiplier = lf->multiplier, offset = lf->offset;
            ScResult r;
            memset(&r, 0, sizeof(r));
            if (use_sc) r = sc_predict(&sc, ps.n, ps.nw, ps_ne(&ps), ps.w, ps_nn(&ps));
            int32_t pred = predict(&ps, predictor, &r);
            int32_t value;
            if (esz == 2) {
                int16_t* px = (int16_t*)tile + y * stride + x;
                int16_t diff = (int16_t)((int16_t)(*px * (int16_t)multiplier) + (int16_t)offset);
                int16_t v = (int16_t)(diff + (int16_t)pred);
                *px = v;
                value = v;
            } else {
                int32_t* px = (int32_t*)tile + y * stride + x;
                int32_t diff = (int32_t)((uint32_t)*px * (uint32_t)multiplier + (uint32_t)offset);
                int32_t v = (int32_t)((uint32_t)diff + (uint32_t)pred);
                *px = v;
                value = v;
            }
            if (use_sc) sc_record(&sc, &r, value);
            ps_record(&ps, value);
        }
    }
    free(ps.prev_row); free(ps.curr_row);
    free(sc.true_err_row); free(sc.subpred_err_row);
}

void orc_predict_apply(void* tile, size_t stride, size_t width, size_t height, int esz, uint32_t predictor,
                       int32_t multiplier, int32_t offset, const int32_t wp[11]) {
    const JxlGpuMaLeaf lf = {predictor, multiplier, offset};
    orc_predict_apply_leaves(tile, stride, width, height, esz, 0, &lf, wp);
}

/* The predictor pass of Palette::inverse_inner's slow path (transform/palette.rs:112-142): one
 * PredictorState over the whole channel; samples flagged in `need_delta` (index < nb_deltas) get
 * the prediction added.  The state records the i32 `sample_value` (palette.rs:130-139), which for
 * i16 buffers can differ from the stored, truncated sample. */
void orc_palette_delta_pass(void* grid, size_t stride, size_t width, size_t height, int esz,
                            const uint8_t* need_delta, uint32_t d_pred, const int32_t wp[11]) {
    if (width == 0 || height == 0) return;
    PState ps;
    memset(&ps, 0, sizeof(ps));
    ps.width = (uint32_t)width;
    ps.prev_row = (int32_t*)calloc(width, 4);
    ps.curr_row = (int32_t*)calloc(width, 4);
    ScPred sc;
    memset(&sc, 0, sizeof(sc));
    if (d_pred == 6) {
        sc.width = (uint32_t)width;
        sc.true_err_row = (int32_t*)calloc(width, 4);
        sc.subpred_err_row = (uint32_t(*)[4])calloc(width, 16);
        sc.p1 = wp[0]; sc.p2 = wp[1];
        for (int i = 0; i < 5; ++i) sc.p3[i] = wp[2 + i];
        for (int i = 0; i < 4; ++i) sc.wn[i] = (uint32_t)wp[7 + i];
    }
    for (size_t y = 0; y < height; ++y) {
        for (size_t x = 0; x < width; ++x) {
            ScResult r;
            if (d_pred == 6) r = sc_predict(&sc, ps.n, ps.nw, ps_ne(&ps), ps.w, ps_nn(&ps));
            int32_t sample_value = esz == 2 ? (int32_t)((int16_t*)grid)[y * stride + x] : ((int32_t*)grid)[y * stride + x];
            if (need_delta[y * width + x]) {
                int32_t diff = predict(&ps, d_pred, &r);
                sample_value = (int32_t)((uint32_t)sample_value + (uint32_t)diff);
                if (esz == 2) ((int16_t*)grid)[y * stride + x] = (int16_t)sample_value;
                else ((int32_t*)grid)[y * stride + x] = sample_value;
            }
            if (d_pred == 6) sc_record(&sc, &r, sample_value);
            ps_record(&ps, sample_value);
        }
    }
    free(ps.prev_row); free(ps.curr_row);
    free(sc.true_err_row); free(sc.subpred_err_row);
}
