/* A C caller of the C ABI (include/jxlgpu.h): no Python, no torch.  Builds the smallest valid
 * VarDCT frame (16x8 samples = two DCT8 varblocks, HF coefficients zero), renders V1-V8 through
 * jxlgpu_vardct_render_host and checks the result against what the format prescribes for a
 * DC-only block: every sample of a block equals its dequantised LF value
 *     lf = q * (m_lf * 2^(9 - extra_precision) / (global_scale * quant_lf))    (vardct/mod.rs:387-412)
 * Exit codes: 0 ok, 3 no usable device (what a GPU-less host must get), 1 anything else. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jxlgpu.h"

int main(void) {
    if (jxlgpu_abi_version() != JXLGPU_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    jxlgpu_ctx* ctx = NULL;
    int rc = jxlgpu_create(0, &ctx);
    if (rc == JXLGPU_ERR_DEVICE) { fprintf(stderr, "no device: jxlgpu_create -> %d (no CPU fallback)\n", rc); return 3; }
    if (rc != JXLGPU_OK) { fprintf(stderr, "jxlgpu_create -> %d\n", rc); return 1; }

    enum { W = 16, H = 8 };
    static int32_t coeff[3][H * W];          /* all zero: DC-only blocks */
    static float ones[64];
    for (int i = 0; i < 64; ++i) ones[i] = 1.0f;
    int16_t lfq[3][2] = {{40, -24}, {3, -5}, {17, 9}}; /* lf_quant channels: [0]=Y, [1]=X, [2]=B */
    uint8_t kind[2] = {JXLGPU_DCT8, JXLGPU_DCT8};
    int32_t hf_mul[2] = {5, 7};
    float sigma[2] = {1.0f, 1.0f};
    int32_t zero_tile[1] = {0};

    JxlGpuLfGroup g;
    memset(&g, 0, sizeof(g));
    g.width_px = W; g.height_px = H;
    for (int k = 0; k < 3; ++k) g.lf_quant[k] = lfq[k];
    g.extra_precision = 1;
    g.has_hf_meta = 1;
    g.block_kind = kind; g.hf_mul = hf_mul; g.epf_sigma = sigma;
    g.x_from_y = zero_tile; g.b_from_y = zero_tile;

    JxlGpuVardctDesc d;
    memset(&d, 0, sizeof(d));
    d.abi = JXLGPU_ABI_VERSION;
    d.width = W; d.height = H; d.group_dim = 256;
    d.lf_sample_type = JXLGPU_SAMPLE_I16;
    for (int c = 0; c < 3; ++c) d.coeff[c] = coeff[c];
    d.coeff_stride = W;
    d.coeff_format = JXLGPU_COEFF_DENSE; d.coeff_sample_type = JXLGPU_SAMPLE_I32;
    d.num_lf_groups = 1; d.lf_groups = &g;
    d.global_scale = 4096; d.quant_lf = 16;
    d.m_lf[0] = 1.0f / 32.0f; d.m_lf[1] = 1.0f / 4.0f; d.m_lf[2] = 1.0f / 2.0f;
    d.colour_factor = 84; d.base_correlation_x = 0.0f; d.base_correlation_b = 0.0f;
    d.x_factor_lf = 128; d.b_factor_lf = 128;   /* CfL-LF factors exactly zero */
    d.x_qm_scale = 2; d.b_qm_scale = 2;
    d.quant_bias[0] = d.quant_bias[1] = d.quant_bias[2] = 0.5f;
    d.quant_bias_numerator = 0.145f;
    d.skip_adaptive_lf_smoothing = 1;
    for (int c = 0; c < 3; ++c) d.dequant[JXLGPU_DCT8][c] = ones;
    d.upsampling.factor = 1;

    static float out[3][H * W];
    JxlGpuOut o;
    for (int c = 0; c < 3; ++c) o.planes[c] = out[c];
    o.stride = W; o.mem = JXLGPU_MEM_HOST;
    rc = jxlgpu_vardct_render_host(ctx, &d, JXLGPU_STAGE_LF | JXLGPU_STAGE_TRANSFORM, &o);
    if (rc != JXLGPU_OK) { fprintf(stderr, "render_host -> %d: %s\n", rc, jxlgpu_last_error(ctx)); jxlgpu_destroy(ctx); return 1; }

    /* framebuffer order X, Y, B <- lf_quant channels 1, 0, 2 (util.rs:275-298) */
    static const int src[3] = {1, 0, 2};
    int bad = 0;
    for (int c = 0; c < 3; ++c) {
        float scale = (float)((double)d.m_lf[c] * 256.0 / ((double)d.global_scale * (double)d.quant_lf));
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float want = (float)lfq[src[c]][x / 8] * scale;
                float got = out[c][y * W + x];
                float diff = got - want;
                if (diff < 0) diff = -diff;
                /* the inverse DCT of a DC-only block returns the DC value up to rounding */
                if (diff > 1e-6f * (want < 0 ? -want : want) + 1e-9f) {
                    if (bad++ < 5) fprintf(stderr, "c=%d (%d,%d): got %.9g want %.9g\n", c, x, y, got, want);
                }
            }
    }
    /* an ABI-version mismatch must be refused, not crash */
    d.abi = 1;
    jxlgpu_frame* f = NULL;
    if (jxlgpu_vardct_upload(ctx, &d, &f) != JXLGPU_ERR_ABI) { fprintf(stderr, "stale ABI accepted\n"); bad++; }
    jxlgpu_destroy(ctx);
    if (bad) return 1;
    printf("ok\n");
    return 0;
}
