/* ORACLE — test infrastructure only.  See dct.h for provenance. */
#include "dct.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* dct_common.rs:10-50 (data table) */
static const float SEC_HALF_4[2] = {0.541196100146197f, 1.3065629648763764f};
static const float SEC_HALF_8[4] = {0.5097955791041592f, 0.6013448869350453f, 0.8999762231364156f,
                                    2.5629154477415055f};
static const float SEC_HALF_16[8] = {0.5024192861881557f, 0.5224986149396889f, 0.5669440348163577f,
                                     0.6468217833599901f, 0.7881546234512502f, 1.060677685990347f,
                                     1.7224470982383342f, 5.101148618689155f};
static const float SEC_HALF_32[16] = {
    0.5006029982351963f, 0.5054709598975436f, 0.5154473099226246f, 0.5310425910897841f,
    0.5531038960344445f, 0.5829349682061339f, 0.6225041230356648f, 0.6748083414550057f,
    0.7445362710022984f, 0.8393496454155268f, 0.9725682378619608f, 1.1694399334328847f,
    1.4841646163141662f, 2.057781009953411f,  3.407608418468719f,  10.190008123548033f};

static float g_sec_large[3][128];
static int g_sec_large_ready[3];

/* dct_common.rs:52-70 */
const float* orc_sec_half(size_t n) {
    switch (n) {
        case 4: return SEC_HALF_4;
        case 8: return SEC_HALF_8;
        case 16: return SEC_HALF_16;
        case 32: return SEC_HALF_32;
        default: break;
    }
    int idx = n == 64 ? 0 : n == 128 ? 1 : 2;
    if (!g_sec_large_ready[idx]) {
        for (size_t k = 0; k < n / 2; ++k) {
            float theta = (float)(2 * k + 1) / (float)(2 * n) * 3.14159265358979323846f;
            g_sec_large[idx][k] = (1.0f / cosf(theta)) / 2.0f;
        }
        g_sec_large_ready[idx] = 1;
    }
    return g_sec_large[idx];
}

void orc_set_sec_half_large(size_t n, const float* table) {
    int idx = n == 64 ? 0 : n == 128 ? 1 : 2;
    memcpy(g_sec_large[idx], table, sizeof(float) * (n / 2));
    g_sec_large_ready[idx] = 1;
}

/* dct_common.rs:77-115 (data table) */
float orc_scale_f(size_t c, size_t logb) {
    static const float SCALE_F[32] = {
        1.0000000000000000f, 0.9996047255830407f, 0.9984194528776054f, 0.9964458326264695f,
        0.9936866130906366f, 0.9901456355893141f, 0.9858278282666936f, 0.9807391980963174f,
        0.9748868211368796f, 0.9682788310563117f, 0.9609244059440204f, 0.9528337534340876f,
        0.9440180941651672f, 0.9344896436056892f, 0.9242615922757944f, 0.9133480844001980f,
        0.9017641950288744f, 0.8895259056651056f, 0.8766500784429904f, 0.8631544288990163f,
        0.8490574973847023f, 0.8343786191696513f, 0.8191378932865928f, 0.8033561501721485f,
        0.7870549181591013f, 0.7702563888779096f, 0.7529833816270532f, 0.7352593067735488f,
        0.7171081282466044f, 0.6985543251889097f, 0.6796228528314652f, 0.6603391026591464f};
    return SCALE_F[c << logb];
}

#define SQRT2F 1.41421356237309504880f

/* generic/dct.rs:144-172 */
static void dct4(const float in[4], float out[4], int forward) {
    const float sec0 = 0.5411961f;
    const float sec1 = 1.306563f;
    if (forward) {
        float sum03 = in[0] + in[3];
        float sum12 = in[1] + in[2];
        float tmp0 = (in[0] - in[3]) * sec0;
        float tmp1 = (in[1] - in[2]) * sec1;
        float out0 = (tmp0 + tmp1) / 4.0f;
        float out1 = (tmp0 - tmp1) / 4.0f;
        out[0] = (sum03 + sum12) / 4.0f;
        out[1] = out0 * SQRT2F + out1;
        out[2] = (sum03 - sum12) / 4.0f;
        out[3] = out1;
    } else {
        float tmp0 = in[1] * SQRT2F;
        float tmp1 = in[1] + in[3];
        float out0 = (tmp0 + tmp1) * sec0;
        float out1 = (tmp0 - tmp1) * sec1;
        float sum02 = in[0] + in[2];
        float sub02 = in[0] - in[2];
        out[0] = sum02 + out0;
        out[1] = sub02 + out1;
        out[2] = sub02 - out1;
        out[3] = sum02 - out0;
    }
}

/* generic/dct.rs:174-293 */
void orc_dct_1d(float* io, float* scratch, size_t n, int forward) {
    if (n == 0 || n == 1) return;
    if (n == 2) {
        float tmp0 = io[0] + io[1];
        float tmp1 = io[0] - io[1];
        if (forward) {
            io[0] = tmp0 / 2.0f;
            io[1] = tmp1 / 2.0f;
        } else {
            io[0] = tmp0;
            io[1] = tmp1;
        }
        return;
    }
    if (n == 4) {
        float in[4] = {io[0], io[1], io[2], io[3]};
        dct4(in, io, forward);
        return;
    }
    if (n == 8) {
        const float* sec = SEC_HALF_8;
        if (forward) {
            float input0[4] = {(io[0] + io[7]) / 2.0f, (io[1] + io[6]) / 2.0f,
                               (io[2] + io[5]) / 2.0f, (io[3] + io[4]) / 2.0f};
            float input1[4] = {(io[0] - io[7]) * sec[0] / 2.0f, (io[1] - io[6]) * sec[1] / 2.0f,
                               (io[2] - io[5]) * sec[2] / 2.0f, (io[3] - io[4]) * sec[3] / 2.0f};
            float output0[4], output1[4];
            dct4(input0, output0, 1);
            for (int i = 0; i < 4; ++i) io[i * 2] = output0[i];
            dct4(input1, output1, 1);
            output1[0] *= SQRT2F;
            for (int i = 0; i < 3; ++i) io[i * 2 + 1] = output1[i] + output1[i + 1];
            io[7] = output1[3];
        } else {
            float input0[4] = {io[0], io[2], io[4], io[6]};
            float input1[4] = {io[1] * SQRT2F, io[3] + io[1], io[5] + io[3], io[7] + io[5]};
            float output0[4], output1[4];
            dct4(input0, output0, 0);
            dct4(input1, output1, 0);
            for (int i = 0; i < 4; ++i) {
                float r = output1[i] * sec[i];
                io[i] = output0[i] + r;
                io[7 - i] = output0[i] - r;
            }
        }
        return;
    }

    const size_t h = n / 2;
    const float* sec = orc_sec_half(n);
    float* input0 = scratch;
    float* input1 = scratch + h;
    float* output0 = io;
    float* output1 = io + h;
    if (forward) {
        for (size_t i = 0; i < h; ++i) {
            input0[i] = (io[i] + io[n - i - 1]) / 2.0f;
            input1[i] = (io[i] - io[n - i - 1]) / 2.0f;
        }
        for (size_t i = 0; i < h; ++i) input1[i] *= sec[i];
        orc_dct_1d(input0, output0, h, 1);
        orc_dct_1d(input1, output1, h, 1);
        input1[0] *= SQRT2F;
        for (size_t i = 0; i + 1 < h; ++i) input1[i] += input1[i + 1];
        for (size_t i = 0; i < h; ++i) io[i * 2] = input0[i];
        for (size_t i = 0; i < h; ++i) io[i * 2 + 1] = input1[i];
    } else {
        for (size_t i = 0; i < h; ++i) {
            input0[i] = io[i * 2];
            input1[i] = io[i * 2 + 1];
        }
        for (size_t i = 1; i < h; ++i) input1[h - i] += input1[h - i - 1];
        input1[0] *= SQRT2F;
        orc_dct_1d(input0, output0, h, 0);
        orc_dct_1d(input1, output1, h, 0);
        for (size_t i = 0; i < h; ++i) input1[i] *= sec[i];
        for (size_t i = 0; i < h; ++i) {
            output0[i] = scratch[i] + scratch[i + h];
            output1[h - i - 1] = scratch[i] - scratch[i + h];
        }
    }
}

#define AT(x, y) io[(size_t)(y) * stride + (size_t)(x)]

/* generic/dct.rs:5-141.  The reference's general case is: 1-D DCT of every row, transpose in
 * square tiles, 1-D DCT of every (former) column, transpose back — i.e. rows then columns; the
 * tile transposes only move data, so a strided column gather is arithmetically identical.      */
void orc_dct_2d(float* io, size_t stride, size_t width, size_t height, int forward) {
    if (width * height <= 1) return;
    const float mul = forward ? 0.5f : 1.0f;
    if (width == 2 && height == 1) {
        float v0 = AT(0, 0), v1 = AT(1, 0);
        AT(0, 0) = (v0 + v1) * mul;
        AT(1, 0) = (v0 - v1) * mul;
        return;
    }
    if (width == 1 && height == 2) {
        float v0 = AT(0, 0), v1 = AT(0, 1);
        AT(0, 0) = (v0 + v1) * mul;
        AT(0, 1) = (v0 - v1) * mul;
        return;
    }
    if (width == 2 && height == 2) {
        float v00 = AT(0, 0), v01 = AT(1, 0), v10 = AT(0, 1), v11 = AT(1, 1);
        AT(0, 0) = (v00 + v01 + v10 + v11) * mul * mul;
        AT(1, 0) = (v00 - v01 + v10 - v11) * mul * mul;
        AT(0, 1) = (v00 + v01 - v10 - v11) * mul * mul;
        AT(1, 1) = (v00 - v01 - v10 + v11) * mul * mul;
        return;
    }

    size_t maxdim = width > height ? width : height;
    float* buf = (float*)malloc(sizeof(float) * maxdim * 3);
    float* scratch = buf;         /* `buf` in the reference                */
    float* col = buf + maxdim;    /* gathered column / `row` temporaries   */
    float* col1 = buf + 2 * maxdim;

    if (height == 1) {
        orc_dct_1d(&AT(0, 0), scratch, width, forward);
        free(buf);
        return;
    }
    if (width == 1) {
        for (size_t y = 0; y < height; ++y) col[y] = AT(0, y);
        orc_dct_1d(col, scratch, height, forward);
        for (size_t y = 0; y < height; ++y) AT(0, y) = col[y];
        free(buf);
        return;
    }
    if (height == 2) {
        for (size_t x = 0; x < width; ++x) {
            float tv0 = AT(x, 0), tv1 = AT(x, 1);
            AT(x, 0) = (tv0 + tv1) * mul;
            AT(x, 1) = (tv0 - tv1) * mul;
        }
        orc_dct_1d(&AT(0, 0), scratch, width, forward);
        orc_dct_1d(&AT(0, 1), scratch, width, forward);
        free(buf);
        return;
    }
    if (width == 2) {
        for (size_t y = 0; y < height; ++y) {
            float v0 = AT(0, y), v1 = AT(1, y);
            col[y] = (v0 + v1) * mul;
            col1[y] = (v0 - v1) * mul;
        }
        orc_dct_1d(col, scratch, height, forward);
        orc_dct_1d(col1, scratch, height, forward);
        for (size_t y = 0; y < height; ++y) {
            AT(0, y) = col[y];
            AT(1, y) = col1[y];
        }
        free(buf);
        return;
    }

    for (size_t y = 0; y < height; ++y) orc_dct_1d(&AT(0, y), scratch, width, forward);
    for (size_t x = 0; x < width; ++x) {
        for (size_t y = 0; y < height; ++y) col[y] = AT(x, y);
        orc_dct_1d(col, scratch, height, forward);
        for (size_t y = 0; y < height; ++y) AT(x, y) = col[y];
    }
    free(buf);
}
