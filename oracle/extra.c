/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED against a run of the reference; the bit layout
 * of parse_integer_sample is pinned by tests/test_oracle_extra.py against numpy's own f16 / f32 decoding.
 * CPU restatement of what jxl-oxide does to an extra channel at the end of a frame's render.
 *
 * Follows:
 *   ImageWithRegion::upsample_nonseparable   jxl-render/src/image.rs:487-557  (own bit depth, own shift)
 *   ImageBuffer::convert_to_float_modular    jxl-render/src/image.rs:93-120
 *   BitDepth::parse_integer_sample           jxl-image/src/lib.rs:458-494
 *   upsample (8x passes, then 2x / 4x)       jxl-render/src/features/upsampling.rs:6-41
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static float ec_parse_integer_sample(const JxlGpuExtraChannel* d, int32_t sample) {
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
    uint32_t bits = (is_signed << 31) | ((uint32_t)(exp + 127) << 23) | mantissa;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* out: (width << L) x (height << L) floats, row stride out_stride */
int orc_extra_channel(const JxlGpuExtraChannel* ec, float* out, size_t out_stride) {
    const size_t n = (size_t)ec->width * ec->height;
    float* cur = (float*)malloc(n * sizeof(float));
    if (!cur) return JXLGPU_ERR_OOM;
    for (size_t i = 0; i < n; ++i) {
        int32_t v = ec->sample_type == JXLGPU_SAMPLE_I16 ? (int32_t)((const int16_t*)ec->data)[i] : ((const int32_t*)ec->data)[i];
        cur[i] = ec_parse_integer_sample(ec, v);
    }
    size_t w = ec->width, h = ec->height;
    const uint32_t up8 = ec->upsampling_log2 / 3, last = ec->upsampling_log2 % 3;
    for (uint32_t pass = 0; pass < up8 + (last ? 1 : 0); ++pass) {
        const int k = pass < up8 ? 8 : (last == 1 ? 2 : 4);
        const float* weights = k == 8 ? ec->weights.up8_weight : (k == 2 ? ec->weights.up2_weight : ec->weights.up4_weight);
        if (!weights) { free(cur); return JXLGPU_ERR_INVALID_ARG; }
        float* nxt = (float*)malloc(w * k * h * k * sizeof(float));
        if (!nxt) { free(cur); return JXLGPU_ERR_OOM; }
        orc_upsample_inner(cur, w, w, h, nxt, w * k, k, weights);
        free(cur);
        cur = nxt; w *= k; h *= k;
    }
    for (size_t y = 0; y < h; ++y) memcpy(out + y * out_stride, cur + y * w, w * sizeof(float));
    free(cur);
    return 0;
}
