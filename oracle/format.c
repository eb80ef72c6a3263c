/*
 * ORACLE — test infrastructure only (see oracle.h).  PARITY UNPINNED upstream; checked against
 * numpy flips/transposes in tests/test_oracle_format.py.
 * Follows ImageStream::write_to_buffer / to_original_coord (jxl-oxide/src/fb.rs:309-397) and the
 * sample conversions Sealed::copy_from_grid / copy_from_f32 for f32 grids (fb.rs:436-527).
 */
#include <math.h>
#include <stdint.h>

#include "oracle.h"

static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* `nch` planes (3 colour + extra channels), each with its own row stride; interleaved output. */
int orc_format_output_n(const float* const* planes, const size_t* strides, uint32_t nch, uint32_t width, uint32_t height,
                        uint32_t sample_format, uint32_t orientation, void* out) {
    if (orientation < 1 || orientation > 8) return JXLGPU_ERR_INVALID_ARG;
    uint32_t ow = orientation <= 4 ? width : height, oh = orientation <= 4 ? height : width;
    for (uint32_t y = 0; y < oh; ++y)
        for (uint32_t x = 0; x < ow; ++x) {
            uint32_t ox, oy;  /* to_original_coord, `width`/`height` there are the output's */
            switch (orientation) {
                case 1: ox = x; oy = y; break;
                case 2: ox = ow - x - 1; oy = y; break;
                case 3: ox = ow - x - 1; oy = oh - y - 1; break;
                case 4: ox = x; oy = oh - y - 1; break;
                case 5: ox = y; oy = x; break;
                case 6: ox = y; oy = ow - x - 1; break;
                case 7: ox = oh - y - 1; oy = ow - x - 1; break;
                default: ox = oh - y - 1; oy = x; break;
            }
            for (uint32_t c = 0; c < nch; ++c) {
                float v = planes[c][(size_t)oy * strides[c] + ox];
                size_t i = ((size_t)y * ow + x) * nch + c;
                if (sample_format == JXLGPU_FMT_F32) ((float*)out)[i] = v;
                else if (sample_format == JXLGPU_FMT_U16) {
                    float t = clampf(v * 65535.0f + 0.5f, 0.0f, 65535.0f);  /* NaN stays NaN -> `as u16` = 0 */
                    ((uint16_t*)out)[i] = t != t ? 0 : (uint16_t)t;
                } else {
                    float t = clampf(v * 255.0f + 0.5f, 0.0f, 255.0f);
                    ((uint8_t*)out)[i] = t != t ? 0 : (uint8_t)t;
                }
            }
        }
    return 0;
}

int orc_format_output(const float* const planes[3], size_t stride, uint32_t width, uint32_t height,
                      uint32_t sample_format, uint32_t orientation, void* out) {
    if (orientation < 1 || orientation > 8) return JXLGPU_ERR_INVALID_ARG;
    uint32_t ow = orientation <= 4 ? width : height, oh = orientation <= 4 ? height : width;
    for (uint32_t y = 0; y < oh; ++y)
        for (uint32_t x = 0; x < ow; ++x) {
            uint32_t ox, oy;  /* to_original_coord, `width`/`height` there are the output's */
            switch (orientation) {
                case 1: ox = x; oy = y; break;
                case 2: ox = ow - x - 1; oy = y; break;
                case 3: ox = ow - x - 1; oy = oh - y - 1; break;
                case 4: ox = x; oy = oh - y - 1; break;
                case 5: ox = y; oy = x; break;
                case 6: ox = y; oy = ow - x - 1; break;
                case 7: ox = oh - y - 1; oy = ow - x - 1; break;
                default: ox = oh - y - 1; oy = x; break;
            }
            for (int c = 0; c < 3; ++c) {
                float v = planes[c][(size_t)oy * stride + ox];
                size_t i = ((size_t)y * ow + x) * 3 + c;
                if (sample_format == JXLGPU_FMT_F32) ((float*)out)[i] = v;
                else if (sample_format == JXLGPU_FMT_U16) {
                    float t = clampf(v * 65535.0f + 0.5f, 0.0f, 65535.0f);  /* NaN stays NaN -> `as u16` = 0 */
                    ((uint16_t*)out)[i] = t != t ? 0 : (uint16_t)t;
                } else {
                    float t = clampf(v * 255.0f + 0.5f, 0.0f, 255.0f);
                    ((uint8_t*)out)[i] = t != t ? 0 : (uint8_t)t;
                }
            }
        }
    return 0;
}
