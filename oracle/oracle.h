/*
 * ORACLE — test infrastructure only.  CPU restatement (plain C) of the jxl-oxide hot path
 * (generic scalar flavour).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use anything in oracle/; the product (jxl-oxide_amd/, libjxlgpu.so) never does.
 *
 * PARITY STATUS: pinned only for the 1-D DCT (the reference's own six unit tests,
 * jxl-render/src/vardct/generic/dct.rs:299-435).  Every other function is "parity unpinned":
 * the upstream conformance vectors are not available in this container, so
 * these are checked against f64 analytic formulas and round-trips instead.
 *
 * The oracle shares only the POD descriptor structs of include/jxlgpu.h with the product so the
 * parity tests can hand both sides the very same inputs.
 */
#ifndef JXL_ORACLE_H_
#define JXL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#include "../include/jxlgpu.h"

/* ---- vardct.c ---- */
void orc_dct_select_size(int t, int* bw, int* bh);
void orc_copy_lf_dequant(float* out, size_t out_stride, const void* quant, uint32_t sample_type,
                         size_t width, size_t height, float m_lf, uint32_t global_scale,
                         uint32_t quant_lf, uint32_t extra_precision);
void orc_chroma_from_luma_lf(float* x, const float* y, float* b, size_t n, uint32_t colour_factor,
                             float base_x, float base_b, uint32_t x_factor_lf, uint32_t b_factor_lf);
void orc_adaptive_lf_smoothing(size_t width, size_t height, float* in_x, float* in_y, float* in_b,
                               const float m_lf[3], uint32_t global_scale, uint32_t quant_lf);
void orc_transform_block(float* coeff, size_t stride, int dct_select);
void orc_inject_llf(float* coeff, size_t stride, const float* lf, size_t lf_stride, int dct_select);
void orc_vardct_lf(const JxlGpuVardctDesc* d, float* const lf[3]);
int jxl_oracle_vardct_render(const JxlGpuVardctDesc* d, uint32_t stages, float* const out[3],
                             uint32_t out_stride, float* const lf_out[3]);

/* ---- modular.c ---- */
int jxl_oracle_modular_inverse(const JxlGpuModularDesc* d, void* const* out);
int jxl_oracle_modular_render(const JxlGpuModularDesc* d, uint32_t stages, float* const out[3],
                              uint32_t out_stride);

/* ---- filters.c ---- */
/* apply_gabor_like on one width x height plane (in -> out, both tight stride `stride_*`). */
void orc_gabor_plane(const float* in, size_t in_stride, float* out, size_t out_stride, size_t width,
                     size_t height, const float weights[2]);
/* One EPF step (0, 1 or 2) over three planes.  sigma: per-8x8-cell plane, stride sigma_stride. */
void orc_epf_step(int step, const float* const in[3], size_t in_stride, float* const out[3],
                  size_t out_stride, size_t width, size_t height, const float* sigma,
                  size_t sigma_stride, const JxlGpuFilterParams* fp);

/* ---- upsample.c ---- */
/* upsample_inner<K>: in (w x h) -> out (w*K x h*K). weights: 15/55/210 floats. */
void orc_upsample_inner(const float* in, size_t in_stride, size_t w, size_t h, float* out,
                        size_t out_stride, int k, const float* weights);

/* ---- format.c ---- */
int orc_format_output(const float* const planes[3], size_t stride, uint32_t width, uint32_t height,
                      uint32_t sample_format, uint32_t orientation, void* out);

int orc_format_output_n(const float* const* planes, const size_t* strides, uint32_t nch, uint32_t width, uint32_t height,
                        uint32_t sample_format, uint32_t orientation, void* out);

/* ---- extra.c ---- */
int orc_extra_channel(const JxlGpuExtraChannel* ec, float* out, size_t out_stride);

/* ---- color.c ---- */
void orc_color_transform(float* const ch[3], size_t n, const JxlGpuColorParams* cp);

/* ---- post.c: gabor -> epf -> upsample -> colour on planes `pix` (stride `stride`) ---- */
int orc_post_stages(float* const pix[3], size_t stride, size_t width, size_t height,
                    const float* sigma, size_t sigma_stride, const JxlGpuFilterParams* fp,
                    const JxlGpuUpsampling* up, const JxlGpuNoiseParams* np, size_t group_dim, float corr_x,
                    float corr_b, const JxlGpuColorParams* cp, uint32_t stages, float* const out[3],
                    uint32_t out_stride);
/* noise.c: features/noise.rs on full planes (the frame after upsampling); JXLGPU_ERR_UNSUPPORTED
 * for the geometry on which the reference itself panics (see noise.c fill_once). */
int orc_render_noise(float* const ch[3], size_t stride, size_t width, size_t height, size_t group_dim,
                     const JxlGpuNoiseParams* np, float corr_x, float corr_b);
void orc_noise_group(uint32_t width, uint32_t height, uint64_t seed0, uint64_t seed1, float* out, uint32_t* stride_out);

/* predict.c: single-leaf predictor application on one tile (residuals -> samples, in place) */
void orc_predict_apply_leaves(void* tile, size_t stride, size_t width, size_t height, int esz, int axis,
                              const JxlGpuMaLeaf* leaves, const int32_t wp[11]);
void orc_predict_apply(void* tile, size_t stride, size_t width, size_t height, int esz, uint32_t predictor,
                       int32_t multiplier, int32_t offset, const int32_t wp[11]);

void orc_palette_delta_pass(void* grid, size_t stride, size_t width, size_t height, int esz,
                            const uint8_t* need_delta, uint32_t d_pred, const int32_t wp[11]);

/* jpeg.c: chroma-subsampled VarDCT (JPEG transcodes), chroma upsampling, YCbCr -> RGB */
void orc_jpeg_shift(const uint32_t jpeg_upsampling[3], int idx, int* hshift, int* vshift, int* has_h, int* has_v);
void orc_upsample_jpeg(const float* in, size_t in_stride, size_t in_w, size_t in_h, int hshift, int vshift,
                       float* out, size_t target_width, size_t target_height);
void orc_ycbcr_to_rgb(float* cb_r, float* y_g, float* cr_b, size_t n);
int orc_vardct_subsampled(const JxlGpuVardctDesc* d, float* const full[3]);

/* blend.c: blend_single on one channel rectangle (alpha planes are host pointers) */
void orc_blend_rect(float* base, size_t base_stride, const float* new_grid, size_t new_stride, const JxlGpuBlendRect* r);

/* OpenMP thread count of the oracle's parallel loops (returns the value in effect). */
int jxl_oracle_set_threads(int n);

#endif
