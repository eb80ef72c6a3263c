/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED (no upstream vectors);
 * tests/test_oracle_noise.py checks the generator against an independent numpy xorshift128+ /
 * splitmix64, the convolution against an f64 5x5 Laplacian-like kernel on the mirrored noise
 * image, and the statistics of the result.
 *
 * Noise synthesis, jxl-render/src/features/noise.rs, kept in the reference's own structure
 * (per-group noise buffers, 9-neighbour padding, 5-row ring buffer) so that the device kernels,
 * which work on one global noise image with a mirrored border, are checked against the literal
 * adjacency logic:
 *   render_noise        noise.rs:12-90
 *   init_noise          noise.rs:92-164
 *   rng_seed0/1         noise.rs:168-177
 *   NoiseGroup::new     noise.rs:199-231
 *   convolve_fill       noise.rs:240-318
 *   fill_once           noise.rs:320-363
 *   fill_padded_row     noise.rs:365-393
 *   XorShift128Plus     noise.rs:397-448, split_mix_64 :451-456
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define NB 8 /* noise.rs:395 `const N: usize = 8` */

static uint64_t split_mix_64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct {
    uint64_t s0[NB], s1[NB];
} XorShift128Plus;

static void xs_new(XorShift128Plus* r, uint64_t seed0, uint64_t seed1) {
    r->s0[0] = split_mix_64(seed0 + 0x9E3779B97F4A7C15ull);
    r->s1[0] = split_mix_64(seed1 + 0x9E3779B97F4A7C15ull);
    for (int i = 1; i < NB; ++i) {
        r->s0[i] = split_mix_64(r->s0[i - 1]);
        r->s1[i] = split_mix_64(r->s1[i - 1]);
    }
}

/* fill_batch + get_u32_bits (little endian: low word first) */
static void xs_get_u32_bits(XorShift128Plus* r, uint32_t out[NB * 2]) {
    for (int i = 0; i < NB; ++i) {
        uint64_t s1 = r->s0[i];
        uint64_t s0 = r->s1[i];
        uint64_t ret = s1 + s0;
        r->s0[i] = s0;
        s1 ^= s1 << 23;
        r->s1[i] = s1 ^ (s0 ^ (s1 >> 18) ^ (s0 >> 5));
        out[2 * i] = (uint32_t)ret;
        out[2 * i + 1] = (uint32_t)(ret >> 32);
    }
}

typedef struct {
    float* buf[3];
    size_t width, height, stride;
} NoiseGroup;

typedef struct {
    const float* p;
    size_t width, height, stride;
    int present;
} Sub;

static const float* sub_row(const Sub* s, size_t y) { return s->p + y * s->stride; }

static void noise_group_new(NoiseGroup* g, size_t width, size_t height, uint64_t seed0, uint64_t seed1) {
    size_t width_n2 = (width + NB * 2 - 1) / (NB * 2);
    g->width = width; g->height = height; g->stride = width_n2 * NB * 2;
    XorShift128Plus rng;
    xs_new(&rng, seed0, seed1);
    for (int c = 0; c < 3; ++c) {
        size_t num_iters = width_n2 * height;
        g->buf[c] = (float*)malloc(sizeof(float) * num_iters * NB * 2);
        for (size_t it = 0; it < num_iters; ++it) {
            uint32_t bits[NB * 2];
            xs_get_u32_bits(&rng, bits);
            for (int k = 0; k < NB * 2; ++k) {
                uint32_t u = (bits[k] >> 9) | 0x3f800000u;
                memcpy(&g->buf[c][it * NB * 2 + k], &u, 4);
            }
        }
    }
}

/* noise.rs:365-393; out has this_len + 4 entries */
static void fill_padded_row(float* out, size_t out_len, const float* this_, size_t this_len, const float* left,
                            size_t left_len, const float* right, size_t right_len) {
    if (left) {
        out[0] = left[left_len - 2];
        out[1] = left[left_len - 1];
    } else if (this_len >= 2) {
        out[0] = this_[1];
        out[1] = this_[0];
    } else {
        out[0] = this_[0];
        out[1] = this_[0];
    }
    memcpy(out + 2, this_, sizeof(float) * this_len);
    if (right) {
        if (right_len >= 2) {
            out[out_len - 2] = right[0];
            out[out_len - 1] = right[1];
        } else {
            out[out_len - 2] = right[0];
            out[out_len - 1] = right[0];
        }
    } else {
        out[out_len - 2] = out[out_len - 3];
        out[out_len - 1] = out[out_len - 4];
    }
}

static void fill_from(float* out, size_t out_len, const Sub* c, const Sub* l, const Sub* r, size_t y) {
    fill_padded_row(out, out_len, sub_row(c, y), c->width, l->present ? sub_row(l, y) : NULL, l->width,
                    r->present ? sub_row(r, y) : NULL, r->width);
}

/* noise.rs:320-363.  Returns -1 where the reference would panic (row index out of range). */
static int fill_once(float* out, size_t out_len, size_t fill_y, const Sub adj[9]) {
    const Sub* this_ = &adj[4];
    size_t height = this_->height;
    size_t source_y;
    const Sub *c, *l, *r;
    if (fill_y >= height) {
        source_y = fill_y - height;
        c = &adj[7]; l = &adj[6]; r = &adj[8];
    } else {
        source_y = fill_y;
        c = &adj[4]; l = &adj[3]; r = &adj[5];
    }
    if (!c->present) {
        if (height - 1 >= source_y) {
            source_y = height - 1 - source_y;
            c = this_; l = &adj[3]; r = &adj[5];
        } else {
            size_t dy = source_y - height + 1;
            if (adj[1].present) {
                c = &adj[1]; l = &adj[0]; r = &adj[2];
                source_y = c->height - dy;
            } else {
                c = this_; l = &adj[3]; r = &adj[5];
                source_y = 0;
            }
        }
    }
    if (source_y >= c->height) return -1; /* get_row panics, shared_subgrid.rs:117-124 */
    fill_from(out, out_len, c, l, r, source_y);
    return 0;
}

/* noise.rs:240-318 */
static int convolve_fill(float* out, size_t out_stride, size_t width, size_t height, const Sub adj[9]) {
    const Sub* this_ = &adj[4];
    size_t input_width = width + 4;
    float* rows = (float*)calloc(input_width * 5, sizeof(float));
    int rc = 0;
    if (adj[1].present) {
        const Sub* c = &adj[1];
        for (int offset_y = -2; offset_y < 0; ++offset_y)
            fill_from(rows + (size_t)(2 + offset_y) * input_width, input_width, c, &adj[0], &adj[2],
                      c->height + offset_y);
    } else if (height >= 2) {
        for (int offset_y = -2; offset_y < 0; ++offset_y) {
            size_t y = (size_t)(-(offset_y + 1));
            fill_from(rows + (size_t)(2 + offset_y) * input_width, input_width, this_, &adj[3], &adj[5], y);
        }
    } else {
        for (int y = 0; y < 2; ++y) fill_from(rows + (size_t)y * input_width, input_width, this_, &adj[3], &adj[5], 0);
    }
    for (size_t y = 0; y < 3 && !rc; ++y) rc = fill_once(rows + (2 + y) * input_width, input_width, y, adj);

    for (size_t y = 0; y < height && !rc; ++y) {
        size_t center_y = (y + 2) % 5;
        float* out_buf = out + y * out_stride;
        for (size_t x = 0; x < width; ++x) {
            float sum = 0.0f;
            for (int dy = 0; dy < 5; ++dy) {
                const float* input_row = rows + (size_t)dy * input_width;
                for (int dx = 0; dx < 5; ++dx) sum += input_row[x + dx] * 0.16f;
            }
            out_buf[x] = sum - rows[center_y * input_width + x + 2] * 4.0f;
        }
        if (y != height - 1) {
            size_t next_y = y + 3;
            size_t fill_y = (next_y + 2) % 5;
            rc = fill_once(rows + fill_y * input_width, input_width, next_y, adj);
        }
    }
    free(rows);
    return rc;
}

/* init_noise (noise.rs:92-164) + render_noise (noise.rs:12-90) on full planes (region = frame) */
int orc_render_noise(float* const ch[3], size_t stride, size_t width, size_t height, size_t group_dim,
                     const JxlGpuNoiseParams* np, float corr_x, float corr_b) {
    uint64_t seed0 = ((uint64_t)np->visible_frames << 32) + (uint64_t)np->invisible_frames;
    size_t groups_per_row = (width + group_dim - 1) / group_dim;
    size_t group_rows = (height + group_dim - 1) / group_dim;
    size_t num_groups = groups_per_row * group_rows;
    NoiseGroup* groups = (NoiseGroup*)calloc(num_groups, sizeof(NoiseGroup));
#pragma omp parallel for schedule(dynamic)
    for (long gi = 0; gi < (long)num_groups; ++gi) {
        size_t x0 = ((size_t)gi % groups_per_row) * group_dim, y0 = ((size_t)gi / groups_per_row) * group_dim;
        uint64_t seed1 = ((uint64_t)x0 << 32) + (uint64_t)y0;
        size_t gw = group_dim < width - x0 ? group_dim : width - x0;
        size_t gh = group_dim < height - y0 ? group_dim : height - y0;
        noise_group_new(&groups[gi], gw, gh, seed0, seed1);
    }
    float* conv[3];
    for (int c = 0; c < 3; ++c) conv[c] = (float*)malloc(sizeof(float) * width * height);
    int rc = 0;
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int c = 0; c < 3; ++c) {
        for (long gi = 0; gi < (long)num_groups; ++gi) {
            long gx = gi % (long)groups_per_row, gy = gi / (long)groups_per_row;
            Sub adj[9];
            for (int idx = 0; idx < 9; ++idx) {
                long x = gx + idx % 3 - 1, y = gy + idx / 3 - 1;
                adj[idx].present = 0; adj[idx].p = NULL; adj[idx].width = adj[idx].height = adj[idx].stride = 0;
                if (x < 0 || y < 0 || x >= (long)groups_per_row || y >= (long)group_rows) continue;
                const NoiseGroup* g = &groups[y * (long)groups_per_row + x];
                adj[idx].present = 1; adj[idx].p = g->buf[c];
                adj[idx].width = g->width; adj[idx].height = g->height; adj[idx].stride = g->stride;
            }
            const NoiseGroup* g = &groups[gi];
            int r = convolve_fill(conv[c] + (size_t)gy * group_dim * width + (size_t)gx * group_dim, width, g->width,
                                  g->height, adj);
            if (r) {
#pragma omp atomic write
                rc = r;
            }
        }
    }
    if (!rc) {
        float lut[9];
        memcpy(lut, np->lut, sizeof(float) * 8);
        lut[8] = np->lut[7];
#pragma omp parallel for schedule(static)
        for (long y = 0; y < (long)height; ++y) {
            float* row_x = ch[0] + (size_t)y * stride;
            float* row_y = ch[1] + (size_t)y * stride;
            float* row_b = ch[2] + (size_t)y * stride;
            const float* nrx = conv[0] + (size_t)y * width;
            const float* nry = conv[1] + (size_t)y * width;
            const float* nrb = conv[2] + (size_t)y * width;
            for (size_t x = 0; x < width; ++x) {
                float grid_x = row_x[x], grid_y = row_y[x];
                float noise_x = nrx[x], noise_y = nry[x], noise_b = nrb[x];
                float in_x = grid_x + grid_y;
                float in_y = grid_y - grid_x;
                float in_scaled_x = fmaxf(0.0f, in_x * 3.0f);
                float in_scaled_y = fmaxf(0.0f, in_y * 3.0f);
                /* `as usize` saturates; values are >= 0 here */
                size_t in_x_int = in_scaled_x >= 8.0f ? 7 : (size_t)in_scaled_x;
                size_t in_y_int = in_scaled_y >= 8.0f ? 7 : (size_t)in_scaled_y;
                float in_x_frac = in_scaled_x - (float)in_x_int;
                float in_y_frac = in_scaled_y - (float)in_y_int;
                float sx = (lut[in_x_int + 1] - lut[in_x_int]) * in_x_frac + lut[in_x_int];
                float sy = (lut[in_y_int + 1] - lut[in_y_int]) * in_y_frac + lut[in_y_int];
                float nx = 0.22f * sx * (0.0078125f * noise_x + 0.9921875f * noise_b);
                float ny = 0.22f * sy * (0.0078125f * noise_y + 0.9921875f * noise_b);
                row_x[x] += corr_x * (nx + ny) + nx - ny;
                row_y[x] += nx + ny;
                row_b[x] += corr_b * (nx + ny);
            }
        }
    }
    for (int c = 0; c < 3; ++c) free(conv[c]);
    for (size_t gi = 0; gi < num_groups; ++gi)
        for (int c = 0; c < 3; ++c) free(groups[gi].buf[c]);
    free(groups);
    return rc ? JXLGPU_ERR_UNSUPPORTED : 0;
}

/* Raw (pre-convolution) noise of one group, for the generator test. */
void orc_noise_group(uint32_t width, uint32_t height, uint64_t seed0, uint64_t seed1, float* out, uint32_t* stride_out) {
    NoiseGroup g;
    noise_group_new(&g, width, height, seed0, seed1);
    *stride_out = (uint32_t)g.stride;
    for (int c = 0; c < 3; ++c) {
        memcpy(out + (size_t)c * g.stride * height, g.buf[c], sizeof(float) * g.stride * height);
        free(g.buf[c]);
    }
}
