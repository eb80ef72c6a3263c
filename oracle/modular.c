/*
 * ORACLE — test infrastructure only (see oracle.h).  CPU restatement of jxl-oxide's Modular
 * inverse transforms and the int -> float tail, generic scalar flavour.  PARITY UNPINNED upstream
 * (no vectors in this container); pinned here by exact round trips: forward Squeeze / RCT /
 * gradient prediction written independently in numpy (tests/modular_forward.py) must invert to
 * the original integers bit-for-bit.
 *
 * Follows:
 *   inverse_h/v_i32_base, _i16_base   jxl-modular/src/transform/squeeze.rs:59-120, 803-862
 *   tendency_i32 / tendency_i16       jxl-modular/src/transform/squeeze.rs:1104-1172
 *   Squeeze::set_default_params       jxl-modular/src/transform.rs:285-341
 *   Squeeze::transform_channel_info   jxl-modular/src/transform.rs:343-437 (sub-rectangle carving)
 *   Squeeze::inverse, SqueezeParams   jxl-modular/src/transform.rs:439-493
 *   inverse_row_*_base, inverse_permute   jxl-modular/src/transform/rct.rs:154-256
 *   Palette::inverse / inverse_simple jxl-modular/src/transform.rs:260-281, transform/palette.rs:146-173
 *   Palette::inverse_inner slow path  transform/palette.rs:27-142 (DELTA_PALETTE :11-24)
 *   decode_simple_grad (arithmetic)   jxl-modular/src/image.rs:821-872, sample.rs:129-136,179-186
 *   convert_to_float_modular(_xyb)    jxl-render/src/image.rs:93-189
 *   BitDepth::parse_integer_sample    jxl-image/src/lib.rs:458-494
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---------------------------------------------------------------- tendency, per sample type */
#define DEFINE_SQUEEZE(S, SUFFIX)                                                                  \
    static S tendency_##SUFFIX(S a, S b, S c) {                                                    \
        if (a >= b && b >= c) {                                                                    \
            S x = (S)((S)((S)((S)((S)(4 * a) - (S)(3 * c)) - b) + 6) / 12);                        \
            if ((S)(x - (x & 1)) > (S)(2 * (S)(a - b))) x = (S)((S)(2 * (S)(a - b)) + 1);          \
            if ((S)(x + (x & 1)) > (S)(2 * (S)(b - c))) x = (S)(2 * (S)(b - c));                   \
            return x;                                                                              \
        } else if (a <= b && b <= c) {                                                             \
            S x = (S)((S)((S)((S)((S)(4 * a) - (S)(3 * c)) - b) - 6) / 12);                        \
            if ((S)(x + (x & 1)) < (S)(2 * (S)(a - b))) x = (S)((S)(2 * (S)(a - b)) - 1);          \
            if ((S)(x - (x & 1)) < (S)(2 * (S)(b - c))) x = (S)(2 * (S)(b - c));                   \
            return x;                                                                              \
        }                                                                                          \
        return 0;                                                                                  \
    }                                                                                              \
    /* squeeze.rs:59-88 / 91-120: `merged` = width x height window, row stride `stride` */         \
    static void inverse_h_##SUFFIX(S* merged, size_t stride, size_t width, size_t height) {        \
        size_t avg_width = (width + 1) / 2;                                                        \
        _Pragma("omp parallel for schedule(static)")                                               \
        for (long yy = 0; yy < (long)height; ++yy) {                                               \
            S* row_out = merged + (size_t)yy * stride;                                             \
            S* scratch = (S*)malloc(sizeof(S) * width);                                            \
            memcpy(scratch, row_out, sizeof(S) * width);                                           \
            const S* avg_row = scratch;                                                            \
            const S* residu_row = scratch + avg_width;                                             \
            S avg = avg_row[0];                                                                    \
            S left = avg;                                                                          \
            for (size_t x = 0; x < width / 2; ++x) {                                               \
                S residu = residu_row[x];                                                          \
                S next_avg = x + 1 < avg_width ? avg_row[x + 1] : avg;                             \
                S diff = (S)(residu + tendency_##SUFFIX(left, avg, next_avg));                     \
                S first = (S)(avg + (S)(diff / 2));                                                \
                S second = (S)(first - diff);                                                      \
                row_out[2 * x] = first;                                                            \
                row_out[2 * x + 1] = second;                                                       \
                avg = next_avg;                                                                    \
                left = second;                                                                     \
            }                                                                                      \
            if (width & 1) row_out[width - 1] = avg_row[avg_width - 1];                            \
            free(scratch);                                                                         \
        }                                                                                          \
    }                                                                                              \
    /* squeeze.rs:803-831 / 834-862 */                                                             \
    static void inverse_v_##SUFFIX(S* merged, size_t stride, size_t width, size_t height) {        \
        size_t avg_height = (height + 1) / 2;                                                      \
        _Pragma("omp parallel for schedule(static)")                                               \
        for (long xx = 0; xx < (long)width; ++xx) {                                                \
            size_t x = (size_t)xx;                                                                 \
            S* scratch = (S*)malloc(sizeof(S) * height);                                           \
            for (size_t y = 0; y < height; ++y) scratch[y] = merged[y * stride + x];               \
            const S* avg_col = scratch;                                                            \
            const S* residu_col = scratch + avg_height;                                            \
            S avg = avg_col[0];                                                                    \
            S top = avg;                                                                           \
            for (size_t y = 0; y < height / 2; ++y) {                                              \
                S residu = residu_col[y];                                                          \
                S next_avg = y + 1 < avg_height ? avg_col[y + 1] : avg;                            \
                S diff = (S)(residu + tendency_##SUFFIX(top, avg, next_avg));                      \
                S first = (S)(avg + (S)(diff / 2));                                                \
                S second = (S)(first - diff);                                                      \
                merged[(2 * y) * stride + x] = first;                                              \
                merged[(2 * y + 1) * stride + x] = second;                                         \
                avg = next_avg;                                                                    \
                top = second;                                                                      \
            }                                                                                      \
            if (height & 1) merged[(height - 1) * stride + x] = avg_col[avg_height - 1];           \
            free(scratch);                                                                         \
        }                                                                                          \
    }                                                                                              \
    /* rct.rs:154-184 / 201-231 + inverse_permute :234-256, one W x H plane triple */              \
    static void inverse_rct_##SUFFIX(S* pa, S* pb, S* pc, size_t sa, size_t sb, size_t sc,         \
                                     size_t width, size_t height, uint32_t rct_type) {             \
        uint32_t permutation = rct_type / 7, type = rct_type % 7;                                  \
        _Pragma("omp parallel for schedule(static)")                                               \
        for (long yy = 0; yy < (long)height; ++yy) {                                               \
            S* ra = pa + (size_t)yy * sa;                                                          \
            S* rb = pb + (size_t)yy * sb;                                                          \
            S* rc = pc + (size_t)yy * sc;                                                          \
            for (size_t x = 0; x < width; ++x) {                                                   \
                S a = ra[x], b = rb[x], c = rc[x], d, e, f;                                        \
                if (type == 6) {                                                                   \
                    S tmp = (S)(a - (S)(c >> 1));                                                  \
                    e = (S)(c + tmp);                                                              \
                    f = (S)(tmp - (S)(b >> 1));                                                    \
                    d = (S)(f + b);                                                                \
                } else {                                                                           \
                    d = a;                                                                         \
                    f = (type & 1) ? (S)(c + a) : c;                                               \
                    e = (type >> 1) == 1 ? (S)(b + a)                                              \
                      : (type >> 1) == 2 ? (S)(b + (S)((S)(a + f) >> 1)) : b;                      \
                }                                                                                  \
                /* inverse_permute: sequence of row swaps applied to (a,b,c) = (d,e,f) */          \
                S t;                                                                               \
                switch (permutation) {                                                             \
                    case 1: t = d; d = e; e = t; t = d; d = f; f = t; break;                       \
                    case 2: t = d; d = e; e = t; t = e; e = f; f = t; break;                       \
                    case 3: t = e; e = f; f = t; break;                                            \
                    case 4: t = d; d = e; e = t; break;                                            \
                    case 5: t = d; d = f; f = t; break;                                            \
                    default: break;                                                                \
                }                                                                                  \
                ra[x] = d; rb[x] = e; rc[x] = f;                                                   \
            }                                                                                      \
        }                                                                                          \
    }                                                                                              \
    /* decode_simple_grad arithmetic (image.rs:821-872): residuals -> samples, in place */          \
    static void gradient_apply_##SUFFIX(S* g, size_t stride, size_t width, size_t height) {        \
        S w = 0;                                                                                   \
        for (size_t x = 0; x < width; ++x) { w = (S)(g[x] + w); g[x] = w; }                        \
        for (size_t y = 1; y < height; ++y) {                                                      \
            const S* prev = g + (y - 1) * stride;                                                  \
            S* out = g + y * stride;                                                               \
            w = (S)(out[0] + prev[0]);                                                             \
            out[0] = w;                                                                            \
            for (size_t x = 1; x < width; ++x) {                                                   \
                int64_t n = prev[x], nw = prev[x - 1], ww = w;                                     \
                int64_t hi = ww > n ? ww : n, lo = ww > n ? n : ww;                                \
                int64_t p = lo + hi - nw;                                                          \
                if (p < lo) p = lo;                                                                \
                if (p > hi) p = hi;                                                                \
                S value = (S)(out[x] + (S)p);                                                      \
                out[x] = value;                                                                    \
                w = value;                                                                         \
            }                                                                                      \
        }                                                                                          \
    }

/* transform/palette.rs:11-24 (data table) */
static const int16_t DELTA_PALETTE[72][3] = {
    {0, 0, 0}, {4, 4, 4}, {11, 0, 0}, {0, 0, -13}, {0, -12, 0}, {-10, -10, -10},
    {-18, -18, -18}, {-27, -27, -27}, {-18, -18, 0}, {0, 0, -32}, {-32, 0, 0}, {-37, -37, -37},
    {0, -32, -32}, {24, 24, 45}, {50, 50, 50}, {-45, -24, -24}, {-24, -45, -45}, {0, -24, -24},
    {-34, -34, 0}, {-24, 0, -24}, {-45, -45, -24}, {64, 64, 64}, {-32, 0, -32}, {0, -32, 0},
    {-32, 0, 32}, {-24, -45, -24}, {45, 24, 45}, {24, -24, -45}, {-45, -24, 24}, {80, 80, 80},
    {64, 0, 0}, {0, 0, -64}, {0, -64, -64}, {-24, -24, 45}, {96, 96, 96}, {64, 64, 0},
    {45, -24, -24}, {34, -34, 0}, {112, 112, 112}, {24, -45, -45}, {45, 45, -24}, {0, -32, 32},
    {24, -24, 45}, {0, 96, 96}, {45, -24, 24}, {24, -45, -24}, {-24, -45, 24}, {0, -64, 0},
    {96, 0, 0}, {128, 128, 128}, {64, 0, 64}, {144, 144, 144}, {96, 96, 0}, {-36, -36, 36},
    {45, -24, -45}, {45, -45, -24}, {0, 0, -96}, {0, 128, 128}, {0, 96, 0}, {45, 24, -45},
    {-128, 0, 0}, {24, -45, 24}, {-45, 24, -45}, {64, 0, -64}, {64, -64, -64}, {96, 0, 96},
    {45, -45, 24}, {24, 45, -45}, {64, 64, -64}, {128, 128, 0}, {0, 0, -128}, {-24, 45, -45},
};

DEFINE_SQUEEZE(int32_t, i32)
DEFINE_SQUEEZE(int16_t, i16)

/* Wrapping arithmetic note: the casts above truncate to S after every operation, which is what
 * Wrapping<S> does; for S = int32_t the intermediate products are evaluated in unsigned-safe
 * ranges by the callers' data (|sample| < 2^28 in the tests) and -fwrapv in the Makefile. */

/* ---------------------------------------------------------------- channel bookkeeping */
typedef struct {
    int buf;                 /* index into channel buffers (>= 0) or ~meta index (< 0)      */
    uint32_t x0, y0, w, h;   /* sub-rectangle inside the buffer                              */
    int32_t hshift, vshift;
    int nmembers;            /* palette: merged member grids                                 */
    int members[8];
    uint32_t orig_w, orig_h; /* ModularChannelInfo.original_width / height (lib.rs:157-191)  */
} Grid;

typedef struct {
    Grid* g;
    int n, cap;
    int nb_meta;
} GridList;

static void gl_insert(GridList* l, int at, Grid g) {
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->g = (Grid*)realloc(l->g, sizeof(Grid) * l->cap); }
    memmove(l->g + at + 1, l->g + at, sizeof(Grid) * (l->n - at));
    l->g[at] = g;
    l->n++;
}
static Grid gl_remove(GridList* l, int at) {
    Grid g = l->g[at];
    memmove(l->g + at, l->g + at + 1, sizeof(Grid) * (l->n - at - 1));
    l->n--;
    return g;
}

/* transform.rs:285-341 */
static int default_squeeze(const GridList* l, JxlGpuSqueezeStep* sp) {
    int n = 0;
    uint32_t first = (uint32_t)l->nb_meta;
    uint32_t w = l->g[first].w, h = l->g[first].h;
    if ((uint32_t)l->n - first >= 3) {
        const Grid* nx = &l->g[first + 1];
        if (nx->w == w && nx->h == h) {
            sp[n++] = (JxlGpuSqueezeStep){1, 0, first + 1, 2};
            sp[n++] = (JxlGpuSqueezeStep){0, 0, first + 1, 2};
        }
    }
    uint32_t num_c = (uint32_t)l->n - first;
    if (h >= w && h > 8) { sp[n++] = (JxlGpuSqueezeStep){0, 1, first, num_c}; h = (h + 1) / 2; }
    while (w > 8 || h > 8) {
        if (w > 8) { sp[n++] = (JxlGpuSqueezeStep){1, 1, first, num_c}; w = (w + 1) / 2; }
        if (h > 8) { sp[n++] = (JxlGpuSqueezeStep){0, 1, first, num_c}; h = (h + 1) / 2; }
    }
    return n;
}

/* transform.rs:343-437, one step, info + sub-rectangle carving */
static int squeeze_forward_step(GridList* l, const JxlGpuSqueezeStep* sp) {
    uint32_t begin = sp->begin_c, end = sp->begin_c + sp->num_c;
    if (end > (uint32_t)l->n) return -1;
    if (begin < (uint32_t)l->nb_meta) {
        if (!sp->in_place || end > (uint32_t)l->nb_meta) return -1;
        l->nb_meta += (int)sp->num_c;
    }
    Grid* res = (Grid*)malloc(sizeof(Grid) * sp->num_c);
    for (uint32_t i = 0; i < sp->num_c; ++i) {
        Grid* ch = &l->g[begin + i];
        if (ch->w == 0 || ch->h == 0) { free(res); return -1; }
        Grid r = *ch;
        if (sp->horizontal) {
            uint32_t len = ch->w;
            ch->w = (len + 1) / 2;
            r.w = len / 2;
            r.x0 = ch->x0 + ch->w;
            if (ch->hshift >= 0) { ch->hshift++; r.hshift++; }
        } else {
            uint32_t len = ch->h;
            ch->h = (len + 1) / 2;
            r.h = len / 2;
            r.y0 = ch->y0 + ch->h;
            if (ch->vshift >= 0) { ch->vshift++; r.vshift++; }
        }
        res[i] = r;
    }
    int at = sp->in_place ? (int)end : l->n;
    for (uint32_t i = 0; i < sp->num_c; ++i) gl_insert(l, at + (int)i, res[i]);
    free(res);
    return 0;
}

typedef struct {
    const JxlGpuModularDesc* d;
    void** bufs;        /* working copies of the channel buffers */
    void** meta;        /* working copies of the meta buffers    */
} Work;

static void* grid_ptr(const Work* w, const Grid* g, size_t* stride, size_t esz) {
    if (g->buf >= 0) {
        *stride = w->d->channels[g->buf].width;
        return (char*)w->bufs[g->buf] + ((size_t)g->y0 * *stride + g->x0) * esz;
    }
    int m = ~g->buf;
    *stride = w->d->meta_channels[m].width;
    return (char*)w->meta[m] + ((size_t)g->y0 * *stride + g->x0) * esz;
}

/* SqueezeParams::inverse (transform.rs:458-493): merge avg + residual rectangles, then undo */
static void squeeze_inverse_pair(const Work* w, Grid* avg, const Grid* residu, int horizontal) {
    size_t esz = w->d->sample_type == JXLGPU_SAMPLE_I16 ? 2 : 4, stride;
    if (horizontal) avg->w += residu->w; else avg->h += residu->h;
    if (horizontal) { if (avg->hshift > 0) avg->hshift--; } else { if (avg->vshift > 0) avg->vshift--; }
    void* p = grid_ptr(w, avg, &stride, esz);
    if (esz == 2) {
        if (horizontal) inverse_h_i16((int16_t*)p, stride, avg->w, avg->h);
        else inverse_v_i16((int16_t*)p, stride, avg->w, avg->h);
    } else {
        if (horizontal) inverse_h_i32((int32_t*)p, stride, avg->w, avg->h);
        else inverse_v_i32((int32_t*)p, stride, avg->w, avg->h);
    }
}

/* Inverse transforms of a whole Modular image (transform.rs:75-86 in reverse order), in place on
 * copies of the channel buffers; out[c] receives channel c (channels[c].width x height). */
int jxl_oracle_modular_inverse(const JxlGpuModularDesc* d, void* const* out) {
    if (d->abi != JXLGPU_ABI_VERSION) return JXLGPU_ERR_ABI;
    size_t esz = d->sample_type == JXLGPU_SAMPLE_I16 ? 2 : 4;
    Work w = {d, NULL, NULL};
    w.bufs = (void**)calloc(d->num_channels, sizeof(void*));
    w.meta = (void**)calloc(d->num_meta_channels + 1, sizeof(void*));
    for (uint32_t c = 0; c < d->num_channels; ++c) {
        size_t n = (size_t)d->channels[c].width * d->channels[c].height * esz;
        w.bufs[c] = malloc(n);
        memcpy(w.bufs[c], d->channels[c].data, n);
    }
    for (uint32_t c = 0; c < d->num_meta_channels; ++c) {
        size_t n = (size_t)d->meta_channels[c].width * d->meta_channels[c].height * esz;
        w.meta[c] = malloc(n);
        memcpy(w.meta[c], d->meta_channels[c].data, n);
    }
    if (d->residual_predictor > 13 && d->residual_predictor != 0xFFFFFFFFu) return JXLGPU_ERR_INVALID_ARG;

    /* forward bookkeeping: which sub-rectangle is which transformed channel */
    GridList l = {NULL, 0, 0, 0};
    for (uint32_t c = 0; c < d->num_channels; ++c) {
        Grid g = {(int)c, 0, 0, d->channels[c].width, d->channels[c].height, 0, 0, 0, {0}};
        g.orig_w = g.w; g.orig_h = g.h;
        gl_insert(&l, l.n, g);
    }
    JxlGpuSqueezeStep** steps = (JxlGpuSqueezeStep**)calloc(d->num_transforms + 1, sizeof(void*));
    int* nsteps = (int*)calloc(d->num_transforms + 1, sizeof(int));
    int meta_next = 0, rc = 0;
    for (uint32_t t = 0; t < d->num_transforms && rc == 0; ++t) {
        const JxlGpuTransform* tr = &d->transforms[t];
        if (tr->kind == JXLGPU_TR_RCT) {
            if (tr->begin_c + 3 > (uint32_t)l.n) rc = JXLGPU_ERR_INVALID_ARG;
        } else if (tr->kind == JXLGPU_TR_PALETTE) {
            /* transform.rs:214-258 */
            uint32_t begin = tr->begin_c, end = tr->begin_c + tr->num_c;
            if (end > (uint32_t)l.n || tr->num_c > 8 || meta_next >= (int)d->num_meta_channels) { rc = JXLGPU_ERR_INVALID_ARG; break; }
            if (begin < (uint32_t)l.nb_meta) l.nb_meta = l.nb_meta + 2 - (int)tr->num_c; else l.nb_meta += 1;
            Grid* leader = &l.g[begin];
            leader->nmembers = 0;
            Grid members[8];
            for (uint32_t i = begin + 1; i < end; ++i) members[leader->nmembers++] = gl_remove(&l, (int)begin + 1);
            /* keep member buffers addressable: store their buffer ids */
            leader = &l.g[begin];
            for (int i = 0; i < leader->nmembers; ++i) leader->members[i] = members[i].buf;
            Grid pal = {~meta_next, 0, 0, tr->nb_colours, tr->num_c, -1, -1, 0, {0}};  /* new_unshiftable */
            pal.orig_w = pal.w; pal.orig_h = pal.h;
            meta_next++;
            gl_insert(&l, 0, pal);
        } else if (tr->kind == JXLGPU_TR_SQUEEZE) {
            JxlGpuSqueezeStep* sp = (JxlGpuSqueezeStep*)malloc(sizeof(JxlGpuSqueezeStep) * 128);
            int n;
            if (tr->num_sq) { n = (int)tr->num_sq; memcpy(sp, tr->sq, sizeof(*sp) * n); }
            else n = default_squeeze(&l, sp);
            steps[t] = sp;
            nsteps[t] = n;
            for (int i = 0; i < n && rc == 0; ++i)
                if (squeeze_forward_step(&l, &sp[i])) rc = JXLGPU_ERR_INVALID_ARG;
        } else rc = JXLGPU_ERR_INVALID_ARG;
    }

    /* M4: separable predictor application (single-leaf MA tree): residuals -> samples, one (group, channel)
     * subgrid at a time with a fresh PredictorState, on the TRANSFORMED channel list, before the inverse
     * transforms.  prepare_groups (image.rs:209-340): the leading meta channels and the leading channels
     * that fit one group are decoded whole by GlobalModular (skip_while, :224-228); every later channel is
     * cut with into_groups_with_fixed_count (jxl-grid mutable_subgrid.rs:480-515) into pass-group tiles
     * (group_dim >> hshift) x (group_dim >> vshift) when hshift < 3 or vshift < 3 (:258-285), else into
     * LF-group tiles (group_dim >> (shift - 3)) (:286-306); the tile count comes from original_width /
     * original_height.  decode_single_node's dispatch (image.rs:733-777): Gradient with offset 0 and
     * multiplier 1 takes decode_simple_grad, everything else decode_one (predict.c).                    */
    /* Per-unit leaves (JxlGpuModularDesc::unit_leaves): make_flat_tree(channel, stream_index, ..) resolves the decisions on the
     * static properties 0 and 1 (ma.rs:38-41, image.rs:477-490), so every decode unit may come with its own single node
     * (decode_single_node per unit, image.rs:553-562).  One entry per unit: channels in list order, ncols x nrows subgrids each
     * (raster order, the ones outside the transformed channel included), 1 for a channel decoded whole.                    */
    if ((d->residual_predictor <= 13 || d->num_unit_leaves) && rc == 0) {
        const uint32_t gd = d->group_dim ? d->group_dim : 256;
        int global_phase = 1;
        size_t unit_base = 0;
        for (int i = 0; i < l.n && rc == 0; ++i) {
            const Grid* g = &l.g[i];
            if (g->w == 0 || g->h == 0) continue;
            uint32_t tw, th, ncols, nrows;
            if (global_phase && (i < l.nb_meta || (g->w <= gd && g->h <= gd))) {
                tw = g->w; th = g->h; ncols = nrows = 1;
            } else {
                global_phase = 0;
                if (g->hshift < 0 || g->vshift < 0) { rc = JXLGPU_ERR_INVALID_ARG; break; }  /* assert!, image.rs:243 */
                if (g->hshift < 3 || g->vshift < 3) {
                    tw = gd >> g->hshift; th = gd >> g->vshift;
                    ncols = (g->orig_w + gd - 1) / gd; nrows = (g->orig_h + gd - 1) / gd;
                } else {
                    tw = gd >> (g->hshift - 3); th = gd >> (g->vshift - 3);
                    ncols = (g->orig_w + gd * 8 - 1) / (gd * 8); nrows = (g->orig_h + gd * 8 - 1) / (gd * 8);
                }
                if (g->hshift > 31 || g->vshift > 31 || tw == 0 || th == 0) { rc = JXLGPU_ERR_INVALID_ARG; break; }  /* InvalidSqueezeParams */
            }
            size_t stride;
            char* base = (char*)grid_ptr(&w, g, &stride, esz);
            const uint32_t W = g->w, H = g->h;
            if (d->num_unit_leaves && unit_base + (size_t)ncols * nrows > d->num_unit_leaves) { rc = JXLGPU_ERR_INVALID_ARG; break; }
            const JxlGpuMaLeaf* leaves = d->num_unit_leaves ? d->unit_leaves + unit_base : NULL;
            unit_base += (size_t)ncols * nrows;
            int bad_leaf = 0;
#pragma omp parallel for schedule(dynamic)
            for (long t = 0; t < (long)ncols * nrows; ++t) {
                uint32_t x0 = (uint32_t)(t % ncols) * tw, y0 = (uint32_t)(t / ncols) * th;
                if (x0 > W) x0 = W;
                if (y0 > H) y0 = H;
                const uint32_t gw = W - x0 < tw ? W - x0 : tw, gh = H - y0 < th ? H - y0 : th;
                if (gw == 0 || gh == 0) continue;
                char* p = base + ((size_t)y0 * stride + x0) * esz;
                const uint32_t predictor = leaves ? leaves[t].predictor : d->residual_predictor;
                const int32_t multiplier = leaves ? leaves[t].multiplier : d->residual_multiplier;
                const int32_t offset = leaves ? leaves[t].offset : d->residual_offset;
                if (leaves && (predictor == JXLGPU_LEAF_BY_ROW || predictor == JXLGPU_LEAF_BY_COLUMN)) {
                    /* a tree that splits on y / x inside the unit: one leaf per row / column from axis_leaves */
                    const size_t need = predictor == JXLGPU_LEAF_BY_ROW ? gh : gw;
                    if (multiplier < 0 || offset != 0 || !d->axis_leaves || (size_t)multiplier + need > d->num_axis_leaves) { bad_leaf = 1; continue; }
                    const JxlGpuMaLeaf* al = d->axis_leaves + multiplier;
                    int ok = 1;
                    for (size_t k = 0; k < need; ++k) ok &= al[k].predictor <= 13;
                    if (!ok) { bad_leaf = 1; continue; }
                    orc_predict_apply_leaves(p, stride, gw, gh, (int)esz, predictor == JXLGPU_LEAF_BY_ROW ? 1 : 2, al, d->wp_params);
                    continue;
                }
                if (predictor > 13) { bad_leaf = 1; continue; }
                const int simple_grad = predictor == 5 && offset == 0 && multiplier == 1;   /* image.rs:762 */
                if (!simple_grad) orc_predict_apply(p, stride, gw, gh, (int)esz, predictor, multiplier, offset, d->wp_params);
                else if (esz == 2) gradient_apply_i16((int16_t*)p, stride, gw, gh);
                else gradient_apply_i32((int32_t*)p, stride, gw, gh);
            }
            if (bad_leaf) rc = JXLGPU_ERR_INVALID_ARG;
        }
        if (rc == 0 && d->num_unit_leaves && unit_base != d->num_unit_leaves) rc = JXLGPU_ERR_INVALID_ARG;
    }

    /* inverse, last transform first */
    for (int t = (int)d->num_transforms - 1; t >= 0 && rc == 0; --t) {
        const JxlGpuTransform* tr = &d->transforms[t];
        if (tr->kind == JXLGPU_TR_SQUEEZE) {
            for (int i = nsteps[t] - 1; i >= 0; --i) {
                const JxlGpuSqueezeStep* sp = &steps[t][i];
                int begin = (int)sp->begin_c, count = (int)sp->num_c, end = begin + count;
                Grid res[64];
                if (count > 64) { rc = JXLGPU_ERR_UNSUPPORTED; break; }
                int from = sp->in_place ? end : l.n - count;
                for (int k = 0; k < count; ++k) res[k] = gl_remove(&l, from);
                for (int k = 0; k < count; ++k) squeeze_inverse_pair(&w, &l.g[begin + k], &res[k], (int)sp->horizontal);
            }
        } else if (tr->kind == JXLGPU_TR_RCT) {
            Grid* a = &l.g[tr->begin_c];
            size_t sa, sb, sc;
            void* pa = grid_ptr(&w, a, &sa, esz);
            void* pb = grid_ptr(&w, a + 1, &sb, esz);
            void* pc = grid_ptr(&w, a + 2, &sc, esz);
            if (esz == 2) inverse_rct_i16((int16_t*)pa, (int16_t*)pb, (int16_t*)pc, sa, sb, sc, a->w, a->h, tr->rct_type);
            else inverse_rct_i32((int32_t*)pa, (int32_t*)pb, (int32_t*)pc, sa, sb, sc, a->w, a->h, tr->rct_type);
        } else {
            /* Palette::inverse + inverse_simple (palette.rs:146-173); the delta/implicit path with
             * its serial predictor pass (palette.rs:52-142) stays on the host */
            Grid pal = gl_remove(&l, 0);
            Grid* leader = &l.g[tr->begin_c];
            size_t ps, ls;
            const void* pp = grid_ptr(&w, &pal, &ps, esz);
            void* lp = grid_ptr(&w, leader, &ls, esz);
            int simple = 1;
            for (uint32_t y = 0; y < leader->h && simple; ++y)
                for (uint32_t x = 0; x < leader->w; ++x) {
                    int32_t idx = esz == 2 ? ((int16_t*)lp)[y * ls + x] : ((int32_t*)lp)[y * ls + x];
                    if (idx < 0 || idx >= (int32_t)tr->nb_colours) { simple = 0; break; }
                }
            if (!simple) {
                /* palette.rs:47-142: implicit colours, delta entries, then the predictor pass */
                const int32_t nb_colors = (int32_t)tr->nb_colours, nb_deltas = (int32_t)tr->nb_deltas;
                const uint32_t bit_depth = d->bit_depth;
                const size_t W = leader->w, H = leader->h;
                uint8_t* need_delta = (uint8_t*)calloc(W * H, 1);
                size_t n_delta = 0;
                void* dsts[16]; size_t dstr[16];
                int channels = (int)tr->num_c;
                if (channels > 16) { free(need_delta); rc = JXLGPU_ERR_UNSUPPORTED; break; }
                for (int c = 0; c < channels; ++c) {
                    Grid mg = *leader;
                    if (c > 0) { mg.buf = leader->members[c - 1]; mg.nmembers = 0; }
                    dsts[c] = grid_ptr(&w, &mg, &dstr[c], esz);
                }
                for (size_t y = 0; y < H; ++y)
                    for (size_t x = 0; x < W; ++x) {
                        int32_t index = esz == 2 ? ((int16_t*)lp)[y * ls + x] : ((int32_t*)lp)[y * ls + x];
                        if (index < nb_deltas) { need_delta[y * W + x] = 1; ++n_delta; }
                        for (int c = 0; c < channels; ++c) {
                            int32_t v;
                            if (index >= 0 && index < nb_colors) {
                                v = esz == 2 ? ((const int16_t*)pp)[(size_t)c * ps + index] : ((const int32_t*)pp)[(size_t)c * ps + index];
                            } else if (index >= nb_colors) {
                                int32_t i2 = index - nb_colors;
                                if (i2 < 64) {
                                    v = ((i2 >> (2 * c)) % 4) * ((1 << bit_depth) - 1) / 4 + (1 << (bit_depth > 3 ? bit_depth - 3 : 0));
                                } else {
                                    int32_t i3 = i2 - 64;
                                    for (int k = 0; k < c; ++k) i3 /= 5;
                                    v = (i3 % 5) * ((1 << bit_depth) - 1) / 4;
                                }
                            } else if (c >= 3) {
                                v = 0;
                            } else {
                                int32_t i2 = -(index + 1);
                                i2 = i2 % 143;
                                int32_t t = DELTA_PALETTE[(i2 + 1) >> 1][c];
                                if ((i2 & 1) == 0) t = -t;
                                if (bit_depth > 8) t <<= (bit_depth < 24 ? bit_depth : 24) - 8;
                                v = t;
                            }
                            if (esz == 2) ((int16_t*)dsts[c])[y * dstr[c] + x] = (int16_t)v;
                            else ((int32_t*)dsts[c])[y * dstr[c] + x] = v;
                        }
                    }
                if (n_delta)
                    for (int c = 0; c < channels; ++c)
                        orc_palette_delta_pass(dsts[c], dstr[c], W, H, (int)esz, need_delta, tr->d_pred, tr->wp_params);
                free(need_delta);
            } else {
                /* inverse_simple, palette.rs:146-173 */
                for (int c = (int)tr->num_c - 1; c >= 0; --c) {
                    /* members first (they read the index grid), the leader itself last */
                    void* dst;
                    size_t ds;
                    Grid mg = *leader;
                    if (c > 0) { mg.buf = leader->members[c - 1]; mg.nmembers = 0; }
                    dst = grid_ptr(&w, &mg, &ds, esz);
                    for (uint32_t y = 0; y < leader->h; ++y)
                        for (uint32_t x = 0; x < leader->w; ++x) {
                            if (esz == 2) {
                                int32_t idx = ((int16_t*)lp)[y * ls + x];
                                ((int16_t*)dst)[y * ds + x] = ((const int16_t*)pp)[(size_t)c * ps + idx];
                            } else {
                                int32_t idx = ((int32_t*)lp)[y * ls + x];
                                ((int32_t*)dst)[y * ds + x] = ((const int32_t*)pp)[(size_t)c * ps + idx];
                            }
                        }
                }
            }
            /* un-merge: member grids come back right after the leader */
            int nm = leader->nmembers;
            int begin = (int)tr->begin_c;
            for (int i = 0; i < nm; ++i) {
                Grid mg = l.g[begin];
                mg.buf = l.g[begin].members[i];
                mg.nmembers = 0;
                gl_insert(&l, begin + 1 + i, mg);
            }
            l.g[begin].nmembers = 0;
        }
    }

    if (rc == 0)
        for (uint32_t c = 0; c < d->num_channels; ++c)
            if (out && out[c]) memcpy(out[c], w.bufs[c], (size_t)d->channels[c].width * d->channels[c].height * esz);
    for (uint32_t c = 0; c < d->num_channels; ++c) free(w.bufs[c]);
    for (uint32_t c = 0; c < d->num_meta_channels; ++c) free(w.meta[c]);
    for (uint32_t t = 0; t < d->num_transforms; ++t) free(steps[t]);
    free(steps); free(nsteps); free(w.bufs); free(w.meta); free(l.g);
    return rc;
}

/* ---------------------------------------------------------------- int -> float (M5, C5) */
/* jxl-image/src/lib.rs:458-494 */
static float parse_integer_sample(const JxlGpuModularDesc* d, int32_t sample) {
    if (!d->float_sample) {
        int32_t div = (int32_t)((1u << d->bit_depth) - 1);
        return (float)sample / (float)div;
    }
    uint32_t bits_per_sample = d->bit_depth, exp_bits = d->exp_bits;
    uint32_t s = (uint32_t)sample;
    uint32_t mantissa_bits = bits_per_sample - exp_bits - 1;
    uint32_t mantissa_mask = (1u << mantissa_bits) - 1;
    uint32_t exp_mask = ((1u << (bits_per_sample - 1)) - 1) ^ mantissa_mask;
    uint32_t is_signed = (s & (1u << (bits_per_sample - 1))) != 0;
    uint32_t mantissa = s & mantissa_mask;
    int32_t exp = (int32_t)((s & exp_mask) >> mantissa_bits);
    exp = exp - ((1 << (exp_bits - 1)) - 1);
    if (mantissa_bits < 23) mantissa <<= (23 - mantissa_bits);
    else if (mantissa_bits > 23) mantissa >>= (mantissa_bits - 23);
    uint32_t e = (uint32_t)(exp + 127);
    uint32_t bits = (is_signed << 31) | (e << 23) | mantissa;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

void orc_upsample_jpeg(const float* in, size_t in_stride, size_t in_w, size_t in_h, int hshift, int vshift,
                       float* out, size_t target_width, size_t target_height);

/* Whole Modular render: inverse transforms, int -> float (image.rs:93-189), then the shared
 * Gabor / EPF / upsampling / colour tail.  The first three channels are the colour channels
 * (Modular order: for XYB that is Y, X, B; image.rs:148-189 swaps Y/X into framebuffer order). */
int jxl_oracle_modular_render(const JxlGpuModularDesc* d, uint32_t stages, float* const out[3],
                              uint32_t out_stride) {
    /* grayscale (jxl-render/src/render.rs:74-134): `clone_gray` before the Gabor-like filter / the EPF, the clones
     * dropped afterwards (:133-134) — the three filter inputs are the one channel, plane 0 is the image; without
     * filters the channel passes through; noise is skipped on a grayscale buffer (:208-221) */
    const int gray = d->num_color_channels == 1;
    if (gray && d->xyb_encoded) return JXLGPU_ERR_INVALID_ARG;
    if (d->num_channels < (gray ? 1u : 3u)) return JXLGPU_ERR_INVALID_ARG;
    if (gray) stages &= ~(uint32_t)JXLGPU_STAGE_NOISE;
    size_t esz = d->sample_type == JXLGPU_SAMPLE_I16 ? 2 : 4;
    uint32_t W = d->channels[0].width, H = d->channels[0].height;
    /* chroma-subsampled colour channels (do_ycbcr + jpeg_upsampling): the frame is the largest channel; the others are
     * converted at their own size and go through upsample_jpeg (jxl-render/src/image.rs:448-485, render.rs:70-72) */
    int hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, subsampled = 0;
    for (int c = 1; c < 3 && !gray; ++c) {
        if (d->channels[c].width > W) W = d->channels[c].width;
        if (d->channels[c].height > H) H = d->channels[c].height;
    }
    for (int c = 0; c < 3 && !gray; ++c) {
        hs[c] = d->channels[c].width != W; vs[c] = d->channels[c].height != H;
        if ((hs[c] && d->channels[c].width != (W + 1) / 2) || (vs[c] && d->channels[c].height != (H + 1) / 2)) return JXLGPU_ERR_INVALID_ARG;
        subsampled |= hs[c] | vs[c];
    }
    if (subsampled && (d->xyb_encoded || !d->color.ycbcr)) return JXLGPU_ERR_INVALID_ARG;
    void** planes = (void**)calloc(d->num_channels, sizeof(void*));
    for (uint32_t c = 0; c < d->num_channels; ++c)
        planes[c] = malloc((size_t)d->channels[c].width * d->channels[c].height * esz);
    int rc = jxl_oracle_modular_inverse(d, planes);
    float* pix[3] = {NULL, NULL, NULL};
    if (rc == 0) {
        size_t n = (size_t)W * H;
        for (int c = 0; c < 3; ++c) pix[c] = (float*)malloc(sizeof(float) * n);
#define SAMPLE(c, i) (esz == 2 ? (int32_t)((int16_t*)planes[c])[i] : ((int32_t*)planes[c])[i])
        if (d->xyb_encoded) {
            /* image.rs:153-186: b += y (saturating), cast, y<-x*m_x, x<-y*m_y, b*=m_b;
             * result order [y, x, b] = framebuffer (X, Y, B) */
            for (size_t i = 0; i < n; ++i) {
                int32_t y = SAMPLE(0, i), x = SAMPLE(1, i), b = SAMPLE(2, i);
                int32_t bs;
                if (esz == 2) { int32_t t = b + y; bs = t > 32767 ? 32767 : t < -32768 ? -32768 : t; }
                else { int64_t t = (int64_t)b + y; bs = t > INT32_MAX ? INT32_MAX : t < INT32_MIN ? INT32_MIN : (int32_t)t; }
                float py = (float)y, px = (float)x, pb = (float)bs;
                pix[0][i] = px * d->m_lf_unscaled[0];
                pix[1][i] = py * d->m_lf_unscaled[1];
                pix[2][i] = pb * d->m_lf_unscaled[2];
            }
        } else {
            for (int c = 0; c < 3; ++c) {
                const int sc = gray ? 0 : c;
                const size_t cw = d->channels[sc].width, chh = d->channels[sc].height;
                if (!(hs[c] || vs[c])) {
                    for (size_t i = 0; i < n; ++i) pix[c][i] = parse_integer_sample(d, SAMPLE(sc, i));
                } else {
                    float* small = (float*)malloc(sizeof(float) * cw * chh);
                    for (size_t i = 0; i < cw * chh; ++i) small[i] = parse_integer_sample(d, SAMPLE(sc, i));
                    orc_upsample_jpeg(small, cw, cw, chh, hs[c], vs[c], pix[c], W, H);
                    free(small);
                }
            }
        }
#undef SAMPLE
        if (!(stages & JXLGPU_STAGE_MODULAR_TO_FLOAT)) rc = JXLGPU_ERR_INVALID_ARG;
    }
    if (rc == 0) {
        size_t w8 = (W + 7) / 8, h8 = (H + 7) / 8;
        float* sigma = (float*)malloc(sizeof(float) * w8 * h8);
        for (size_t i = 0; i < w8 * h8; ++i) sigma[i] = d->filter.epf_sigma_for_modular;
        JxlGpuColorParams cp = d->color;
        if (!d->xyb_encoded) cp.enabled = 0;  /* already in the display colour space */
        /* render.rs:175-180: no VarDCT LfGlobal -> base_correlations_xb = None -> (0.0, 1.0) (noise.rs:35) */
        rc = orc_post_stages(pix, W, W, H, sigma, w8, &d->filter, &d->upsampling, &d->noise, d->group_dim, 0.0f,
                             1.0f, &cp, stages, out, out_stride);
        free(sigma);
    }
    for (int c = 0; c < 3; ++c) free(pix[c]);
    for (uint32_t c = 0; c < d->num_channels; ++c) free(planes[c]);
    free(planes);
    return rc;
}
