/*
 * jxlgpu.h — C ABI of libjxlgpu.so, the MI355X (gfx950) implementation of jxl-oxide's
 * per-group transform-and-render hot path.
 *
 * The reference (tirr-c/jxl-oxide, pure Rust) has no FFI of its own.  This header declares the
 * entry points a Rust `extern "C"` block would bind at the two seams where the reference hands
 * entropy-decoded state to its per-group CPU workers:
 *
 *   - VarDCT : jxl-render/src/vardct/mod.rs:316  ("Dequant and transform", the
 *              `pool.for_each_vec(groups)` loop) plus the LF prologue at :164-204, the restoration
 *              filters at jxl-render/src/render.rs:76-131 and the colour transform at
 *              jxl-render/src/lib.rs:925-998.
 *   - Modular: jxl-render/src/modular.rs:134 (`modular_image.prepare_subimage().finish(pool)`),
 *              jxl-render/src/image.rs:148-189 (XYB dequant) and the same filters / colour tail.
 *
 * Everything is POD: plain pointers, sizes and scalars; no C++/torch types.  All functions return
 * JXLGPU_OK (0) or a negative error code and never throw or abort across the boundary
 * (the reference's convention is `Result<T, jxl_render::Error>`, jxl-render/src/error.rs).
 *
 * Threading: a `jxlgpu_ctx` owns its HIP streams (render, upload, download) and may be used from one
 * thread at a time; create one ctx per rendering thread (the reference renders keyframes from arbitrary
 * rayon workers, jxl-oxide-cli/src/decode.rs:293-304).  There is no global state.  A ctx keeps a few host
 * worker threads of its own for the per-frame work-list build of an upload (JXLGPU_HOST_THREADS=n, 0 = none).
 *
 * Asynchrony: jxlgpu_vardct_upload (grouped transport), a render with out == NULL and
 * jxlgpu_frame_format_output into JXLGPU_MEM_HOST_PINNED memory return without waiting for the device; the
 * operations of one frame run in the order they were called, transfers of one frame overlap the kernels of
 * another.  jxlgpu_frame_wait / jxlgpu_synchronize wait.  jxlgpu_frame_free never blocks.
 */
#ifndef JXLGPU_H_
#define JXLGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JXLGPU_ABI_VERSION 24u

/* ---- error codes (map to jxl_render::Error in the Rust shim) ---- */
#define JXLGPU_OK 0
#define JXLGPU_ERR_INVALID_ARG (-1) /* malformed descriptor (maps to a panic/assert in the reference) */
#define JXLGPU_ERR_OOM (-2)         /* device or pinned allocation failed (jxl_grid::OutOfMemory)     */
#define JXLGPU_ERR_DEVICE (-3)      /* HIP runtime error; jxlgpu_last_error() has the text             */
#define JXLGPU_ERR_UNSUPPORTED (-4) /* valid JPEG XL, but outside this library's scope (caller falls   */
                                    /* back to the CPU path): frames above the size limits, ...         */
#define JXLGPU_ERR_ABI (-5)         /* desc->abi != JXLGPU_ABI_VERSION                                  */

typedef struct jxlgpu_ctx jxlgpu_ctx;     /* device + stream + scratch arena          */
typedef struct jxlgpu_frame jxlgpu_frame; /* one frame's device-resident decode state */

/* ---- TransformType, numbering of jxl-vardct/src/dct_select.rs:4-32 ---- */
enum {
    JXLGPU_DCT8 = 0, JXLGPU_HORNUSS, JXLGPU_DCT2, JXLGPU_DCT4, JXLGPU_DCT16, JXLGPU_DCT32,
    JXLGPU_DCT16X8, JXLGPU_DCT8X16, JXLGPU_DCT32X8, JXLGPU_DCT8X32, JXLGPU_DCT32X16,
    JXLGPU_DCT16X32, JXLGPU_DCT4X8, JXLGPU_DCT8X4, JXLGPU_AFV0, JXLGPU_AFV1, JXLGPU_AFV2,
    JXLGPU_AFV3, JXLGPU_DCT64, JXLGPU_DCT64X32, JXLGPU_DCT32X64, JXLGPU_DCT128,
    JXLGPU_DCT128X64, JXLGPU_DCT64X128, JXLGPU_DCT256, JXLGPU_DCT256X128, JXLGPU_DCT128X256,
    JXLGPU_NUM_TRANSFORMS = 27
};
/* BlockInfo (jxl-vardct/src/hf_metadata.rs:39-50) flattened to one byte per 8x8 cell:
 * 0..26 = Data{dct_select}, with hf_mul in the parallel int32 array.                      */
#define JXLGPU_BLOCK_OCCUPIED 0xFEu
#define JXLGPU_BLOCK_UNINIT 0xFFu

/* Sample type of integer planes (`S: Sample`, jxl-modular/src/sample.rs). */
#define JXLGPU_SAMPLE_I32 0u
#define JXLGPU_SAMPLE_I16 1u

/* ---- restoration filters (jxl-frame/src/filter.rs:6-202) ---- */
typedef struct {
    uint32_t gab_enabled;      /* Gabor::Enabled                                              */
    float gab_weights[3][2];   /* per channel [w0 (side), w1 (diag)], filter.rs:12-16          */
    uint32_t epf_iters;        /* 0 = EdgePreservingFilter::Disabled, else 1..3                */
    float epf_channel_scale[3];
    float epf_pass0_sigma_scale, epf_pass2_sigma_scale, epf_border_sad_mul; /* EpfSigma        */
    float epf_sigma_for_modular; /* sigma used where no HfMetadata sigma grid exists           */
} JxlGpuFilterParams;

/* ---- colour transform XYB -> display (jxl-color/src/convert.rs:208-549 op list) ---- */
#define JXLGPU_TF_LINEAR 0u
#define JXLGPU_TF_SRGB 1u
#define JXLGPU_TF_PQ 2u
#define JXLGPU_TF_BT709 3u
#define JXLGPU_TF_GAMMA 4u
#define JXLGPU_TF_HLG 5u   /* linear_to_hlg (tf.rs:148-160): sqrt / libm logf, the latter as glibc computes  */
                           /* it (csrc/libm_f32.h); see hlg_ootf_intensity_target for the inverse OOTF       */
/* PLATFORM CONDITION of JXLGPU_TF_HLG (and of hlg_ootf_intensity_target below): the reference hands these samples to the
 * platform libm (powf / ln); the device evaluates glibc's published algorithms.  Bit-identity with the caller's own CPU
 * path therefore holds where that libm is glibc >= 2.28 on x86_64 with FMA (its `-mfma` variants); on musl, macOS,
 * Windows or x86 without FMA the local CPU path itself computes other bits.  A caller that needs CPU / GPU agreement
 * there runs jxlgpu_selftest_libm once against its own libm and keeps HLG op lists on the CPU when the results differ
 * (the library does not refuse them: it cannot know the caller's libm).                                              */
/* JxlGpuColorParams.gamut_map: what sits between the two Matrix ops (convert.rs:398-414) */
#define JXLGPU_GAMUT_NONE 0u
#define JXLGPU_GAMUT_MAP 1u  /* ColorTransformOp::GamutMap (perceptual intent)                     */
#define JXLGPU_GAMUT_CLIP 2u /* ColorTransformOp::Clip (other intents)                             */
typedef struct {
    uint32_t enabled;          /* 0: leave XYB (save_before_ct / stage tests)                   */
    float opsin_bias[3];       /* OpsinInverseMatrix.opsin_bias, jxl-image/src/color.rs:620      */
    float intensity_target;    /* ToneMapping.intensity_target                                  */
    float matrix[9];           /* merged Matrix op after XybToMixedLms (convert.rs:661-690):     */
                               /* opsin inv_mat, or  target_primaries * inv_mat                  */
    uint32_t gamut_map;        /* JXLGPU_GAMUT_*: GamutMap{luminances, saturation_factor} or Clip   */
                               /* before matrix2                                                 */
    float gamut_luminances[3];
    float gamut_saturation_factor;
    uint32_t has_matrix2;      /* second Matrix op (after GamutMap)                              */
    float matrix2[9];
    uint32_t transfer_function;/* JXLGPU_TF_*                                                    */
    float gamma;               /* for JXLGPU_TF_GAMMA: the exponent apply_gamma receives         */
                               /* (convert.rs:979-1020: 1e7/g, g/1e7, or 1/2.6 for DCI)          */
    float hlg_luminances[3];   /* HdrParams.luminances of the HLG inverse OOTF (see the last field) */
    /* ToneMapRec2408 { hdr_params, target_display_luminance, detect_peak: false } and the
     * GamutMap that follows it for perceptual intent (convert.rs:478-500), between matrix2 and the
     * transfer function.  detect_peak = true (a whole-image reduction) is not offered.            */
    uint32_t tone_map;
    float tm_luminances[3];    /* HdrParams.luminances (row Y of primaries_to_xyz_mat)            */
    float tm_min_nits;         /* HdrParams.min_nits                                              */
    float tm_target_display_luminance; /* 255.0 (SDR target) or 1000.0 (PQ -> HLG)                */
    uint32_t tm_gamut_map;     /* GamutMap{luminances = tm_luminances, saturation_factor}         */
    float tm_gamut_saturation_factor;
    /* frame_header.do_ycbcr (JPEG-recompressed frames): the planes are Cb, Y, Cr and
     * `jxl_color::ycbcr_to_rgb` (jxl-color/src/ycbcr.rs:40-56, jxl-render/src/lib.rs:950-954) runs
     * instead of the XYB op list (`enabled` is ignored).                                          */
    uint32_t ycbcr;
    /* HLG targets (ABI 23).  0: no inverse OOTF.  Otherwise `tf::hlg_inverse_oo(rgb, hlg_luminances, this)`
     * (tf.rs:118-143; a no-op for 295..=305 as there) runs after the tone map and before the tone map's GamutMap —
     * the one place the reference's op lists put it:
     *   XYB image, HLG target (convert.rs:1021-1032): TransferFunction{Hlg} = inverse OOTF with the image's
     *     intensity_target, then linear_to_hlg           -> this = intensity_target, no tone map;
     *   PQ image, HLG target (`from_pq`, convert.rs:501-536): ToneMapRec2408{target 1000, detect_peak: false},
     *     HlgInverseOotf{intensity_target: 1000}, GamutMap{0.1} for perceptual intent, then the transfer function
     *     with intensity_target 300 (its own OOTF skipped)
     *                                                    -> tone_map = 1, tm_target_display_luminance = 1000,
     *                                                       this = 1000, tm_gamut_map / tm_gamut_saturation_factor 0.1;
     *   the same with 999 <= intensity_target <= 1001: only the GamutMap and the transfer function
     *                                                    -> tone_map = 0, this = 0, tm_gamut_map = 1 (since ABI 23
     *                                                       tm_gamut_map is honoured without tone_map, with tm_luminances).
     * The system gamma `1.2 * 1.111.powf(log2(it / 1000))` is evaluated once per frame on the host with the platform libm,
     * as the reference does; the per-sample `mixed.powf(exp)` and `ln` on the device as glibc's powf / logf compute them. */
    float hlg_ootf_intensity_target;
} JxlGpuColorParams;

/* ---- non-separable upsampling (jxl-render/src/features/upsampling.rs) ---- */
typedef struct {
    uint32_t factor;           /* 1, 2, 4 or 8 (frame_header.upsampling)                         */
    const float* up2_weight;   /* 15 floats,  ImageMetadata.up2_weight (jxl-image/src/lib.rs:164) */
    const float* up4_weight;   /* 55 floats                                                       */
    const float* up8_weight;   /* 210 floats                                                      */
} JxlGpuUpsampling;

/* ---- noise synthesis (jxl-render/src/features/noise.rs, LfGlobal.noise) ---- */
typedef struct {
    uint32_t enabled;          /* lf_global.noise.is_some() (render.rs:207)                        */
    float lut[8];              /* NoiseParameters.lut (jxl-frame/src/data/noise.rs:2-4)            */
    uint32_t visible_frames;   /* rng_seed0 = visible << 32 + invisible (noise.rs:168-170)         */
    uint32_t invisible_frames;
} JxlGpuNoiseParams;

/* ---- one LF group's decoded state (jxl-frame LfGroup + jxl-vardct HfMetadata) ---- */
typedef struct {
    uint32_t width_px, height_px; /* LF group size in colour samples (<= group_dim*8)            */
    /* LfCoeff.lf_quant image channels in the reference's channel order [0]=Y,[1]=X,[2]=B
     * (jxl-render/src/util.rs:275-298); ceil(width_px/8) x ceil(height_px/8), tight stride.     */
    const void* lf_quant[3];
    uint32_t extra_precision;     /* LfCoeff.extra_precision (0..3)                               */
    uint32_t has_hf_meta;         /* LfGroup.hf_meta.is_some()                                    */
    const uint8_t* block_kind;    /* bw*bh cells, see JXLGPU_BLOCK_*                              */
    const int32_t* hf_mul;        /* bw*bh, valid where block_kind <= 26                          */
    const float* epf_sigma;       /* bw*bh (HfMetadata.epf_sigma)                                 */
    const int32_t* x_from_y;      /* ceil(width_px/64) x ceil(height_px/64)                       */
    const int32_t* b_from_y;
} JxlGpuLfGroup;

/* Stage mask for jxlgpu_*_render: lets the parity tests stop after any stage. */
#define JXLGPU_STAGE_LF 0x01u        /* V1-V3: LF dequant, CfL-LF, adaptive smoothing              */
#define JXLGPU_STAGE_TRANSFORM 0x02u /* V4-V8: dequant, CfL-HF, LLF injection, inverse transforms   */
#define JXLGPU_STAGE_GABOR 0x04u
#define JXLGPU_STAGE_EPF 0x08u
#define JXLGPU_STAGE_UPSAMPLE 0x10u
#define JXLGPU_STAGE_COLOR 0x20u
#define JXLGPU_STAGE_NOISE 0x80u     /* between UPSAMPLE and COLOR (render.rs:207-222); no-op unless  */
                                     /* noise.enabled                                               */
#define JXLGPU_STAGE_ALL 0xBFu

#define JXLGPU_COEFF_DENSE 0u
#define JXLGPU_COEFF_SPARSE 1u
#define JXLGPU_COEFF_GROUPED 2u

/* One pass group's HF coefficients exactly as the decode loop of `write_hf_coeff` produces them
 * (jxl-vardct/src/hf_coeff.rs:97-254), before they would be stored into the coefficient grid:
 * for every `BlockInfo::Data` cell of the group in raster order (:97-106), for c in [Y, X, B]
 * (:138-140): the number of coefficients actually decoded for it — the `non_zeros` value read at :188
 * MINUS what was left of it when the loop ended early (the coefficient order ran out, or with `allow_partial`
 * the bitstream ended inside the varblock: whatever was stored before that is kept, :207-244) — and then that
 * many (dx, dy, coeff) triples, where
 * (dx, dy) is the coefficient's position inside the varblock after the need_transpose swap
 * (:236-241) and coeff = unpack_signed(ucoeff) << coeff_shift (:235).  The device transform
 * consumes these lists directly (no dense coefficient plane is ever built): the shim replaces the
 * store at hf_coeff.rs:243 by a push.  Every triple needs |coeff| < 32768 (anything else uses
 * JXLGPU_COEFF_SPARSE / _DENSE).  Progressive frames (frame_header.passes.num_passes > 1) hand over one
 * JxlGpuHfGroup per (pass, group): pass p adds its `unpack_signed(ucoeff) << coeff_shift` values to what the
 * earlier passes left at the same positions (`+=`, hf_coeff.rs:234); see JxlGpuVardctDesc.num_passes.     */
typedef struct {
    uint32_t num_varblocks;   /* BlockInfo::Data cells of the group                               */
    uint32_t num_nz;          /* entries in `nz` = sum of nz_count                                */
    const uint16_t* nz_count; /* 3 per varblock, in the order decoded: [Y, X, B]                  */
    const uint32_t* nz;       /* dx | dy << 8 | (uint32_t)(uint16_t)coeff << 16, varblock after    */
                              /* varblock, Y then X then B, in coefficient (decode) order         */
} JxlGpuHfGroup;

typedef struct {
    uint32_t abi;                 /* = JXLGPU_ABI_VERSION                                         */
    uint32_t width, height;       /* frame_header.color_sample_width()/height()                   */
    uint32_t group_dim;           /* frame_header.group_dim(); only 256 is supported              */
    uint32_t lf_sample_type;      /* JXLGPU_SAMPLE_* of lf_quant                                   */
    /* frame_header.jpeg_upsampling, per framebuffer channel (Cb, Y, Cr): ChannelShift::
     * from_jpeg_upsampling (jxl-modular/src/param.rs:105-122).  Non-zero (chroma-subsampled) frames
     * are supported when every varblock is DCT8 and skip_adaptive_lf_smoothing is set (what JPEG
     * transcodes are); then the block grids of the LF groups are rounded up to even cell counts
     * (hf_metadata.rs:70-81), channel c's coefficient plane is shift_size(w8, h8) * 8 samples with
     * row stride `coeff_stride >> hshift(c)`, and lf_quant[k] has the shifted size of its channel. */
    uint32_t jpeg_upsampling[3];
    /* HF coefficients as `write_hf_coeff` leaves them (jxl-vardct/src/hf_coeff.rs:207-244):
     * planes in framebuffer order [0]=X,[1]=Y,[2]=B, width_rounded x height_rounded (ceil to 8),
     * row stride `coeff_stride` elements (jxl-render/src/vardct/mod.rs:206-222, 262-265).
     *   JXLGPU_COEFF_DENSE : coeff[c] = the plane, elements of `coeff_sample_type` (i32 is the
     *                        reference's own framebuffer; i16 halves the H2D volume and is valid
     *                        whenever every |coefficient| < 32768).
     *   JXLGPU_COEFF_GROUPED: see JxlGpuHfGroup above — the preferred transport (4 bytes per
 *                        non-zero coefficient across PCIe, no device-side layout pass).
 *   JXLGPU_COEFF_SPARSE: the non-zero stores of hf_coeff.rs:234 as lists.  coeff[c] = values
     *                        (`coeff_sample_type`), sparse_pos[c][i] = y * coeff_stride + x,
     *                        sparse_count[c] entries; entries accumulate (`+=`, as the passes of a
     *                        progressive frame do), everything not listed is 0.                  */
    const void* coeff[3];
    uint32_t coeff_stride;
    uint32_t coeff_format;        /* JXLGPU_COEFF_*                                                */
    uint32_t coeff_sample_type;   /* JXLGPU_SAMPLE_I32 / JXLGPU_SAMPLE_I16                         */
    const uint32_t* sparse_pos[3];
    uint64_t sparse_count[3];
    /* JXLGPU_COEFF_GROUPED: one entry per pass group, raster order over the frame's
     * ceil(width / group_dim) x ceil(height / group_dim) groups; `coeff`, `coeff_stride`,
     * `coeff_sample_type` and the sparse fields are ignored.  Not offered for chroma-subsampled
     * frames (jpeg_upsampling != 0).                                                             */
    uint32_t num_hf_groups;
    const JxlGpuHfGroup* hf_groups;
    /* JXLGPU_COEFF_GROUPED, progressive frames: `hf_groups` holds num_passes x num_hf_groups entries, pass after
     * pass (hf_groups[p * num_hf_groups + g]); every pass lists every varblock of the block map (with zero counts
     * where it has nothing).  The device sums the passes into its coefficient cells (an integer accumulation, then
     * the dense transform kernels: dequantisation is not linear in the coefficient).  0 or 1: a single pass, the
     * lists feed the transform kernels directly.                                                          */
    uint32_t num_passes;
    /* A truncated stream rendered as far as it goes (`allow_partial`, jxl-render/src/vardct/mod.rs:275-305:
     * a pass group whose decode failed part-way keeps what was decoded): with JXLGPU_COEFF_GROUPED a group
     * may then list FEWER varblocks than its block map holds — the rest have no HF coefficients.  With several
     * passes every (pass, group) list may stop on its own (a progressive stream cut short: later passes
     * typically stop earlier); the device sums what each pass did deliver.                              */
    uint32_t allow_partial;
    /* frame_header.flags.use_lf_frame(): the LF image is the blended render of a previously decoded LF
     * frame (lf_level = 1), used as it is — no LF dequantisation, CfL-LF or adaptive smoothing
     * (jxl-render/src/vardct/mod.rs:175-179).  Three f32 planes X, Y, B of ceil(width / 8) x ceil(height / 8)
     * samples, row stride `lf_frame_stride` elements; NULL = decode the LF from lf_quant (V1-V3).  With an LF
     * frame the lf_quant pointers of the LF groups may be NULL.                                           */
    const float* lf_frame[3];
    uint32_t lf_frame_stride;
    uint32_t num_lf_groups;       /* frame_header.num_lf_groups(), raster order                    */
    const JxlGpuLfGroup* lf_groups;
    /* Quantizer / LfChannelDequantization / LfChannelCorrelation (jxl-vardct/src/lf.rs:11-34) */
    uint32_t global_scale, quant_lf;
    float m_lf[3];                /* m_x_lf, m_y_lf, m_b_lf                                        */
    uint32_t colour_factor;
    float base_correlation_x, base_correlation_b;
    uint32_t x_factor_lf, b_factor_lf;
    uint32_t x_qm_scale, b_qm_scale; /* frame_header, jxl-frame/src/header.rs:32-39                */
    float quant_bias[3];          /* OpsinInverseMatrix, jxl-image/src/color.rs:621-626            */
    float quant_bias_numerator;
    uint32_t skip_adaptive_lf_smoothing; /* frame_header.flags                                      */
    /* DequantMatrixSet (jxl-vardct/src/dequant.rs:661-716): for every TransformType and channel
     * the matrix *as applied* at jxl-render/src/vardct/mod.rs:516-520 — i.e. `get_transposed()`
     * when `need_transpose()`, else `get()` — raster order, (8*bw) x (8*bh) floats.               */
    const float* dequant[JXLGPU_NUM_TRANSFORMS][3];
    /* sec_half(n) tables for n = 64,128,256 (jxl-render/src/vardct/dct_common.rs:52-70): the
     * reference computes them at run time with f32 cos(); pass them so both sides agree
     * bit-for-bit.  NULL = library computes them with cosf().                                      */
    const float* sec_half_large[3];
    JxlGpuFilterParams filter;
    JxlGpuUpsampling upsampling;
    JxlGpuNoiseParams noise;
    JxlGpuColorParams color;
} JxlGpuVardctDesc;

/* Output planes.  `planes[c]` are caller-owned f32 buffers of `height_out` rows with row stride
 * `stride` elements (an `AlignedGrid<f32>` backing store: tight stride = width).  `mem` says
 * whether the pointers are host or device memory.                                               */
#define JXLGPU_MEM_HOST 0u
#define JXLGPU_MEM_DEVICE 1u
#define JXLGPU_MEM_HOST_PINNED 2u /* host memory from jxlgpu_host_alloc: the DMA engine writes it directly */
typedef struct {
    float* planes[3];
    uint32_t stride;
    uint32_t mem;
} JxlGpuOut;

/* ---- context ---- */
int jxlgpu_create(int device, jxlgpu_ctx** out_ctx);
void jxlgpu_destroy(jxlgpu_ctx* ctx);
const char* jxlgpu_last_error(const jxlgpu_ctx* ctx);
uint32_t jxlgpu_abi_version(void);
/* Block until everything queued on the ctx's stream has finished. */
int jxlgpu_synchronize(jxlgpu_ctx* ctx);
/* The ctx's render hipStream_t (so callers can record HIP events around launches). */
void* jxlgpu_stream(jxlgpu_ctx* ctx);
/* Block until everything queued for `frame` (upload, renders, asynchronous output copies) has finished. */
int jxlgpu_frame_wait(jxlgpu_ctx* ctx, jxlgpu_frame* frame);
/* Pinned (page-locked) host memory, for output buffers the DMA engine writes directly (JXLGPU_MEM_HOST_PINNED):
 * what a `Vec<u8>` output buffer of `ImageStream::write_to_buffer` (jxl-oxide/src/fb.rs:309-397) becomes when
 * the caller wants the copy off its thread.  Free with jxlgpu_host_free.                                      */
int jxlgpu_host_alloc(jxlgpu_ctx* ctx, size_t bytes, void** out);
void jxlgpu_host_free(jxlgpu_ctx* ctx, void* p);
/* Memory budget (the reference's `AllocTracker`, jxl-grid/src/alloc_tracker.rs:17-75: every grid allocation is charged
 * to a byte budget and fails with OutOfMemory beyond it).  `limit_bytes` bounds the device memory the ctx's FRAMES may
 * hold at one time (live buffers; recycled buffers waiting in the ctx's pool are given back first and do not count);
 * an upload or render that would exceed it fails with JXLGPU_ERR_OOM and leaves the ctx usable.  0 = no limit (the
 * default).  jxlgpu_memory_usage reports the bytes held by live frames and by the pool.  Outside the budget: buffers the
 * caller owns (jxlgpu_device_alloc, jxlgpu_host_alloc) and the ctx's pinned staging arenas (host memory).            */
int jxlgpu_set_memory_limit(jxlgpu_ctx* ctx, uint64_t limit_bytes);
int jxlgpu_memory_usage(const jxlgpu_ctx* ctx, uint64_t* live_bytes, uint64_t* pooled_bytes);
/* Measurement hook: where the host time of the ctx's last jxlgpu_vardct_upload went, in milliseconds —
 * ms[0] work-list / side-plane build (worker threads), ms[1] reserved (0), ms[2] device allocations + enqueueing
 * the copies, ms[3] the whole call, ms[4] the arena's H2D copy on the device (HIP events; waits for it).      */
int jxlgpu_upload_split(jxlgpu_ctx* ctx, double ms[5]);
/* Diagnostic (ABI 23): evaluates on the device, for `n` host floats, the restatements of the platform libm that the HLG
 * colour ops use (csrc/libm_f32.h; the reference calls libm there: jxl-color/src/tf.rs:118-160) — which = 0: logf(x[i]),
 * which = 1: powf(x[i], y) with a finite y — and copies the results to `out`.  Synchronous.  A caller (or a test) compares
 * them with its own libm: equal bits on glibc >= 2.28 / x86_64, which is what makes HLG frames bit-identical there.     */
int jxlgpu_selftest_libm(jxlgpu_ctx* ctx, int which, const float* x, size_t n, float y, float* out);

/* Measurement hook (no reference counterpart: the reference only has wall-clock MP/s in its CLI,
 * jxl-oxide-cli/src/decode.rs:164-209).  Brackets every launch group of the selected kind with
 * a HIP event pair on the ctx stream; `jxlgpu_profile_read` synchronises, sums the elapsed times
 * of the brackets recorded since the last read and resets.  group: 0 = LF prologue (V1-V3),
 * 1 = varblock transforms (V4-V8), 2 = restoration filters + upsampling + colour, 3 = Modular
 * inverse transforms; -1 = off.                                                                   */
int jxlgpu_profile_select(jxlgpu_ctx* ctx, int group);

/* Tracing hook (ABI 24): the reference wraps the work this library replaces in `tracing` spans ("Load LF groups",
 * "Dequant and transform", "Edge-preserving filter", "Inverse Modular transform": jxl-render/src/vardct/mod.rs:164, :316,
 * filter/epf.rs:21, modular.rs:134).  With a callback set, the library calls it on the CALLING thread when it starts
 * (begin = 1) and when it has finished ENQUEUEING (begin = 0) the launch group that stands for such a span, with the
 * reference's span name — what a Rust shim forwards to `tracing::trace_span!(..).entered()` / the guard's drop, so that a
 * subscriber sees the same span tree with the device path as without it.  Device durations are not what the callback
 * reports (the calls are asynchronous): jxlgpu_profile_select / rocprofv3 measure those.  callback = NULL switches it off. */
typedef void (*jxlgpu_trace_fn)(void* user, const char* span, int begin);
int jxlgpu_set_trace(jxlgpu_ctx* ctx, jxlgpu_trace_fn callback, void* user);
int jxlgpu_profile_read(jxlgpu_ctx* ctx, double* total_ms, uint64_t* brackets);

/* ---- VarDCT ---- */
/* Copy one frame's decoded state to the device and build the device-side tables: the host builds every
 * table into one pinned staging arena (worker threads), one asynchronous H2D copy on the ctx's upload stream
 * moves it, the render stream waits for that copy by event.  The descriptor and everything it points to may
 * be released as soon as the call returns (it has been copied); the call does not wait for the device, except
 * with the dense / sparse coefficient transports and an LF frame, whose planes are copied synchronously.   */
int jxlgpu_vardct_upload(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* desc, jxlgpu_frame** out_frame);
/* Run the selected stages on the device.  `out` may be NULL (results stay in the frame's device
 * buffers, e.g. for benchmarking); otherwise the result of the last selected stage is written to
 * `out` (D2H copy for JXLGPU_MEM_HOST, followed by a stream synchronisation).  A `stages` mask
 * without JXLGPU_STAGE_TRANSFORM produces the LF planes only (jxlgpu_frame_download_lf reads
 * them): passing `out` with it is JXLGPU_ERR_INVALID_ARG.  Frames (after upsampling) taller than
 * 65535 rows are rejected at upload with JXLGPU_ERR_UNSUPPORTED.                                    */
int jxlgpu_vardct_render(jxlgpu_ctx* ctx, jxlgpu_frame* frame, uint32_t stages, const JxlGpuOut* out);
/* Region (cropped) render: `RenderContext::request_image_region` / `render_frame_cropped`
 * (jxl-render/src/lib.rs:232, jxl-oxide-tests/tests/crop/mod.rs:8-222).  `region` is a rectangle of the
 * frame's OUTPUT (after upsampling), `Region { left, top, width, height }` of jxl-render/src/region.rs:4-10;
 * it is intersected with the frame.  The result is the INTERSECTION's width x height samples (what
 * jxlgpu_frame_result_size reports afterwards; region.width x region.height when the region lies inside the
 * frame), written at the origin of `out`, bit-identical to that rectangle of a whole-frame render: the device transforms only the varblocks the padded colour region
 * touches (the padding rules of jxl-render/src/util.rs:51-120: upsampling support, EPF / Gabor reach) and
 * runs the filters, upsampling and colour transform on the rectangle.  The LF image (V1-V3) is always
 * whole: it is 1/64 of the frame.  `stages` must include JXLGPU_STAGE_TRANSFORM.  A frame with noise
 * synthesis (seeded per absolute group) is rendered whole and the region cropped from it.               */
typedef struct {
    int32_t left, top;
    uint32_t width, height;
} JxlGpuRegion;
int jxlgpu_vardct_render_region(jxlgpu_ctx* ctx, jxlgpu_frame* frame, uint32_t stages, const JxlGpuRegion* region,
                                const JxlGpuOut* out);
/* Batch variant: `n` uploaded frames, one launch per stage for all of them — what the
 * reference's callers do with a parallel loop over keyframes (jxl-oxide-cli/src/decode.rs:293-304).
 * Asynchronous, like a render with out == NULL: after jxlgpu_synchronize the results are on the
 * device (jxlgpu_frame_result_plane / jxlgpu_frame_download_result / jxlgpu_frame_format_output).
 * Frames of different sizes may be mixed.  V1-V8 share launches unless a frame has a varblock
 * >= 128 px, chroma subsampling or LF-only groups; the post stage shares launches for the default
 * pipeline (all stages; Gabor + EPF iters 2; no upsampling / noise; plain XYB -> sRGB) and follows
 * frame by frame for anything else.  Whatever does not qualify is rendered one by one by the same
 * call: same results.                                                                              */
int jxlgpu_vardct_render_batch(jxlgpu_ctx* ctx, jxlgpu_frame* const* frames, uint32_t n, uint32_t stages);
/* Copy the result of the frame's last render to `out` (planar f32; host or device memory). */
int jxlgpu_frame_download_result(jxlgpu_ctx* ctx, jxlgpu_frame* frame, const JxlGpuOut* out);
/* Size of the result of the frame's last render (jxlgpu_frame_out_size answers for a stage mask; this
 * answers for what was actually rendered).  JXLGPU_ERR_INVALID_ARG before the first render.           */
int jxlgpu_frame_result_size(const jxlgpu_frame* frame, uint32_t* width, uint32_t* height);
/* upload + render(stages) + free in one call: the drop-in for `render_vardct` + filters + colour. */
int jxlgpu_vardct_render_host(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* desc, uint32_t stages,
                              const JxlGpuOut* out);
/* Never blocks: the frame's device memory is recycled once the work queued so far has finished. */
void jxlgpu_frame_free(jxlgpu_ctx* ctx, jxlgpu_frame* frame);
/* Dimensions of the output of `stages` for this frame (upsampling changes them). */
int jxlgpu_frame_out_size(const jxlgpu_frame* frame, uint32_t stages, uint32_t* width, uint32_t* height);
/* Device pointer to the result of the last render (plane c), for device-side consumers
 * (RCCL gather in the multi-GPU path).  Valid until the next render/free on this frame.            */
const float* jxlgpu_frame_result_plane(const jxlgpu_frame* frame, uint32_t c);
/* Intermediate buffers for stage-level parity tests: the LF image after V1-V3 (3 planes,
 * ceil(width/8) x ceil(height/8)).  Copies to host.                                                */
int jxlgpu_frame_download_lf(jxlgpu_ctx* ctx, const jxlgpu_frame* frame, float* const planes[3]);

/* ---- frame compositing primitive: `blend_single`, jxl-render/src/blend.rs:550-728 ----
 * The per-sample arithmetic of frame blending (`blend`, blend.rs:179-416) and of patches (`patch`,
 * :418-548) on device planes; choosing channels, alpha planes, reference frames and regions stays
 * with the caller, as in the reference.  Rectangles are applied in list order (patches overlap).   */
#define JXLGPU_BLEND_REPLACE 0u
#define JXLGPU_BLEND_ADD 1u
#define JXLGPU_BLEND_MUL 2u       /* BlendMode::Mul(clamp)                                          */
#define JXLGPU_BLEND_BLEND 3u     /* BlendMode::Blend(BlendAlpha); new_alpha == NULL -> Replace      */
#define JXLGPU_BLEND_MULADD 4u    /* BlendMode::MulAdd(BlendAlpha); new_alpha == NULL -> Add         */
#define JXLGPU_BLEND_MIXALPHA 5u  /* BlendMode::MixAlpha { clamp, swapped } (the alpha channel itself) */
#define JXLGPU_BLEND_SKIP 6u
typedef struct {
    uint32_t mode;                 /* JXLGPU_BLEND_*                                                  */
    uint32_t clamp, swapped, premultiplied;
    const float* base_alpha;       /* device plane addressed like `base`, or NULL (alpha = 0)          */
    uint32_t base_alpha_stride;
    const float* new_alpha;        /* device plane addressed like `new_plane`, or NULL                 */
    uint32_t new_alpha_stride;
    uint32_t base_x, base_y;       /* BlendParams.base_topleft                                        */
    uint32_t new_x, new_y;         /* BlendParams.new_topleft                                         */
    uint32_t width, height;
} JxlGpuBlendRect;
/* `base` (base_w x base_h, stride base_stride) and `new_plane` are device pointers, e.g. from
 * jxlgpu_frame_result_plane.  Runs on the ctx stream and returns after it has finished (the copy
 * of the rectangle list on the device is released before the call returns).  A rectangle taller
 * than 65535 rows is rejected (JXLGPU_ERR_UNSUPPORTED).                                             */
int jxlgpu_blend_rects(jxlgpu_ctx* ctx, float* base, uint32_t base_stride, uint32_t base_w, uint32_t base_h,
                       const float* new_plane, uint32_t new_stride, uint32_t new_w, uint32_t new_h,
                       const JxlGpuBlendRect* rects, uint32_t num_rects);

/* ---- output formatting on the device (the step right after the path) ----
 * `ImageStream::write_to_buffer` (jxl-oxide/src/fb.rs:309-397, sample conversion :436-527):
 * planar f32 result of the last render -> interleaved, oriented, f32 / u16 / u8 samples
 * (`(v * max + 0.5).clamp(0, max) as uN`).  Shrinks the D2H copy / the multi-GPU gather 4x for u8. */
#define JXLGPU_FMT_F32 0u
#define JXLGPU_FMT_U16 1u
#define JXLGPU_FMT_U8 2u
#define JXLGPU_MAX_EXTRA 8u
typedef struct {
    uint32_t sample_format;   /* JXLGPU_FMT_*                                                     */
    uint32_t orientation;     /* ImageMetadata.orientation, 1..8 (EXIF numbering)                 */
    /* extra channels interleaved behind the three colour samples of every pixel (RGBA: the alpha channel;
     * `ImageStream` with alpha, jxl-oxide/src/fb.rs:40-118): indices of planes rendered with
     * jxlgpu_frame_render_extra, each of the size of the colour result.  0 = colour only.              */
    uint32_t num_extra;       /* 0..4 */
    uint32_t extra[4];
} JxlGpuFormatDesc;
/* Writes out_w*out_h*(3 + num_extra) samples (out_w/out_h swap for orientations 5..8) to `out` (host memory when
 * out_mem == JXLGPU_MEM_HOST, a device pointer for JXLGPU_MEM_DEVICE).  Needs a render queued on `frame`.
 * JXLGPU_MEM_HOST_PINNED (`out` from jxlgpu_host_alloc): asynchronous — the call returns once the kernel and the
 * copy are queued, `out` is complete after jxlgpu_frame_wait(frame).                                          */
int jxlgpu_frame_format_output(jxlgpu_ctx* ctx, jxlgpu_frame* frame, const JxlGpuFormatDesc* fmt,
                               void* out, uint32_t out_mem, uint32_t* out_w, uint32_t* out_h);

/* ---- extra channels (alpha, depth, spot colours, ...) ----
 * The reference carries every extra channel of a frame through the tail of the render next to the colour
 * channels: `prepare_color_upsampling` adds the frame's upsampling shift to the channel's own (ec_upsampling /
 * dim_shift), and `upsample_nonseparable` (jxl-render/src/image.rs:487-557, called at render.rs:149) converts
 * the integer grid with the channel's OWN bit depth (`convert_to_float_modular` -> BitDepth::parse_integer_sample,
 * jxl-image/src/lib.rs:458-494) and upsamples it with the image's 5x5 kernels (features/upsampling.rs:6-41: 8x
 * passes first, then the 2x / 4x remainder).  The restoration filters and the colour transform never touch extra
 * channels.  One call per channel: upload + int -> float + upsampling; the result stays on the device with the frame
 * (slot `index` < JXLGPU_MAX_EXTRA; jxlgpu_frame_format_output interleaves it) and is copied to `out` if that is not
 * NULL (f32, `out_stride` elements per row; JXLGPU_MEM_HOST_PINNED: asynchronous, see jxlgpu_frame_wait).          */
typedef struct {
    const void* data;          /* width x height integer samples, tight rows: the channel as the Modular decode left it */
    uint32_t width, height;    /* the channel's own (possibly downsampled) size                          */
    uint32_t sample_type;      /* JXLGPU_SAMPLE_I16 / JXLGPU_SAMPLE_I32                                   */
    uint32_t bit_depth;        /* ec_info[i].bit_depth.bits_per_sample                                   */
    uint32_t float_sample;     /* BitDepth::FloatSample                                                   */
    uint32_t exp_bits;
    uint32_t upsampling_log2;  /* the channel's ChannelShift when upsample_nonseparable runs: log2(ec upsampling) +
                                  log2(frame upsampling), 0..6 (0: conversion only)                       */
    JxlGpuUpsampling weights;  /* ImageMetadata.up2/up4/up8_weight (`factor` is ignored); needed when
                                  upsampling_log2 != 0                                                    */
} JxlGpuExtraChannel;
int jxlgpu_frame_render_extra(jxlgpu_ctx* ctx, jxlgpu_frame* frame, uint32_t index, const JxlGpuExtraChannel* ec,
                              float* out, uint32_t out_stride, uint32_t out_mem);
/* Device plane of extra channel `index` (tight rows) and its size; NULL before jxlgpu_frame_render_extra. */
const float* jxlgpu_frame_extra_plane(const jxlgpu_frame* frame, uint32_t index, uint32_t* width, uint32_t* height);

/* ---- multi-GPU: the stitched output without a collective (BASELINE configs 4 / 5) ----
 * One process per GPU.  The rank that owns the stitched output allocates it with jxlgpu_device_alloc and exports
 * it (jxlgpu_ipc_export); every other rank opens the handle (jxlgpu_ipc_open: a peer mapping over xGMI) and passes
 * `base + its offset` as the JXLGPU_MEM_DEVICE destination of jxlgpu_frame_format_output (or of a render's
 * JxlGpuOut): the formatting kernel's stores go straight to the owner's HBM, all ranks at once, each over its own
 * xGMI link to the owner — no rooted gather, no staging copy, nothing for the host to wait for but the end of its
 * own stream.  The 64 handle bytes travel by whatever the host program already has (a pipe, MPI, a torch store).
 * Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver only offers dmabuf IPC.                              */
#define JXLGPU_IPC_HANDLE_BYTES 64
int jxlgpu_device_alloc(jxlgpu_ctx* ctx, size_t bytes, void** out);   /* plain device memory (exportable), zero-filled */
void jxlgpu_device_free(jxlgpu_ctx* ctx, void* p);
int jxlgpu_ipc_export(jxlgpu_ctx* ctx, void* dev_ptr, uint8_t handle[JXLGPU_IPC_HANDLE_BYTES]);
int jxlgpu_ipc_open(jxlgpu_ctx* ctx, const uint8_t handle[JXLGPU_IPC_HANDLE_BYTES], void** dev_ptr);
int jxlgpu_ipc_close(jxlgpu_ctx* ctx, void* dev_ptr);
/* Device-to-host copy of `bytes` from a device pointer (e.g. the stitched output on its owner), after everything
 * queued on the ctx's streams: for hosts without another GPU runtime binding.                                   */
int jxlgpu_device_download(jxlgpu_ctx* ctx, const void* dev_ptr, void* host, size_t bytes);
/* Host-to-device copy into a device pointer (own memory or a peer mapping), blocking: e.g. a probe pattern written
 * through a fresh jxlgpu_ipc_open mapping before kernels are allowed to store through it.                        */
int jxlgpu_device_upload(jxlgpu_ctx* ctx, void* dev_ptr, const void* host, size_t bytes);

/* Bytes the algorithm must move per render for the given stages (compulsory HBM traffic:
 * coefficient read + final write + side data), used by bench.py for the roofline.                 */
uint64_t jxlgpu_frame_algorithmic_bytes(const jxlgpu_frame* frame, uint32_t stages);

/* ---- Modular (inverse transforms after the per-group entropy decode) ---- */
#define JXLGPU_TR_RCT 0u
#define JXLGPU_TR_PALETTE 1u
#define JXLGPU_TR_SQUEEZE 2u
/* One squeeze step (jxl-modular/src/transform.rs:125-137 SqueezeParams). */
typedef struct {
    uint32_t horizontal, in_place, begin_c, num_c;
} JxlGpuSqueezeStep;
/* TransformInfo (jxl-modular/src/transform.rs:19-23). */
typedef struct {
    uint32_t kind;               /* JXLGPU_TR_*                                                    */
    /* Rct */
    uint32_t begin_c, rct_type;
    /* Palette (simple gather, implicit colours above nb_colours, delta entries below nb_deltas
     * with the whole-channel d_pred predictor pass: transform/palette.rs:27-173) */
    uint32_t num_c, nb_colours, nb_deltas, d_pred;
    int32_t wp_params[11];       /* WpHeader p1, p2, p3a..p3e, w0..w3 when d_pred == 6                  */
    /* Squeeze: explicit steps (after set_default_params, transform.rs:285-341) */
    uint32_t num_sq;
    const JxlGpuSqueezeStep* sq;
} JxlGpuTransform;

/* One channel of the *untransformed* image: a full-resolution buffer that holds, after the
 * entropy decode, the transformed sub-channels carved as sub-rectangles exactly as
 * `transform_channel_info` carves them (jxl-modular/src/transform.rs:343-437,
 * jxl-modular/src/image.rs:209-371).                                                               */
typedef struct {
    const void* data;            /* width*height samples, tight stride                              */
    uint32_t width, height;
} JxlGpuModularChannel;

/* The MA-tree leaf of ONE decode unit (a (group, channel) subgrid), without its entropy-coding cluster:
 * MaTreeLeafClustered { predictor, offset, multiplier } (jxl-modular/src/ma.rs).                     */
typedef struct {
    uint32_t predictor;          /* Predictor id 0..13 (jxl-modular/src/predictor.rs:26-41); in `unit_leaves` also
                                  * JXLGPU_LEAF_BY_ROW / JXLGPU_LEAF_BY_COLUMN (below)                 */
    int32_t multiplier, offset;
} JxlGpuMaLeaf;
/* A unit whose (flattened) tree still splits on property 2 (y) or on property 3 (x) — the static properties of
 * `Properties::get`, predictor.rs:458-478: nothing in them depends on decoded samples, so the host can still read
 * every token (decode_slow, image.rs:1169-1228, with `get_leaf` a function of the row / the column alone).  Its entry
 * in `unit_leaves` carries one of these two values as `predictor`, and `multiplier` = the index of its first entry
 * in `axis_leaves`: one leaf per ROW of the unit's subgrid (BY_ROW: gh entries) or per COLUMN (BY_COLUMN: gw
 * entries), each a plain leaf (predictor 0..13).  The unit keeps ONE PredictorState; the self-correcting predictor's
 * state is kept for every sample of the unit as soon as one of its leaves uses it (FlatMaTree::need_self_correcting,
 * ma.rs:275-285).  A tree that splits on y AND x inside one unit is not served (JXLGPU_ERR_UNSUPPORTED is the
 * caller's to raise: there is no encoding for it here).  `offset` of such an entry must be 0.          */
#define JXLGPU_LEAF_BY_ROW 14u
#define JXLGPU_LEAF_BY_COLUMN 15u

typedef struct {
    uint32_t abi;
    uint32_t sample_type;        /* JXLGPU_SAMPLE_I16 / I32 (modular_16bit_buffers)                  */
    uint32_t bit_depth;          /* bits_per_sample (palette delta scaling, int->float)              */
    uint32_t num_channels;       /* colour channels first (1 or 3), then extra channels              */
    /* frame_header.encoded_color_channels(): 3, or 1 for a grayscale image (then never XYB).  0 = 3.  The
     * render of a grayscale frame follows jxl-render/src/render.rs:74-134: the gray channel is cloned into
     * three for the Gabor-like filter and the EPF (which sums its distances over three channels) and the
     * clones are dropped again — result plane 0 is the image, planes 1 and 2 are scratch; noise is not
     * rendered on grayscale (render.rs:208-221).                                                        */
    uint32_t num_color_channels;
    const JxlGpuModularChannel* channels;
    uint32_t num_meta_channels;  /* palette meta channels, in the order the inverse pops them         */
    const JxlGpuModularChannel* meta_channels;
    uint32_t num_transforms;     /* in bitstream order; the inverse runs them in reverse             */
    const JxlGpuTransform* transforms;
    /* M4, predictor application where it is separable from the entropy decode: a single-leaf MA
     * tree (`decode_single_node`, jxl-modular/src/image.rs:716-777), so that
     *     sample = residual * multiplier + offset + predict(neighbours)      (decode_one, :878-890)
     * 0xFFFFFFFF = the channel buffers already hold reconstructed samples; 0..13 = they hold the
     * residuals (`unpack_signed` tokens) of that Predictor (jxl-modular/src/predictor.rs:26-41),
     * applied with a fresh PredictorState per (group, channel) subgrid exactly as the reference
     * carves them (prepare_groups, jxl-modular/src/image.rs:209-340): every TRANSFORMED channel —
     * each Squeeze sub-channel, each palette table — on its own tile grid: the leading meta / small
     * channels whole (GlobalModular), the others in (group_dim >> hshift) x (group_dim >> vshift)
     * pass-group tiles, or LF-group tiles once both shifts reach 3.  The channel buffers and the meta
     * channels hold the residuals of those transformed channels, carved as described above.  6 =
     * SelfCorrecting with `wp_params`.  MA trees with more than one leaf choose the entropy-coding
     * context from the neighbours, so they stay with the entropy decoder on the host (a `SimpleMaTable`
     * tree, image.rs:951-1166, whose one decision property is static — channel, stream, y, x — has one
     * predictor / multiplier / offset for all leaves and IS a single leaf here).  All channels
     * have ChannelShift 0 (no extra-channel upsampling shifts).                                     */
    uint32_t residual_predictor;
    int32_t residual_multiplier; /* MaTreeLeafClustered.multiplier (1 for a default leaf)            */
    int32_t residual_offset;     /* MaTreeLeafClustered.offset                                       */
    int32_t wp_params[11];       /* WpHeader: p1, p2, p3a..p3e, w0..w3 (predictor.rs:8-21)           */
    uint32_t group_dim;          /* frame_header.group_dim(): 128, 256, 512 or 1024 (0 = 256)         */
    /* what happens after the inverse transforms (jxl-render/src/image.rs:93-189) */
    uint32_t xyb_encoded;        /* 1: convert_modular_xyb (M5); 0: int -> float by bit depth (C5)    */
    float m_lf_unscaled[3];      /* m_x_lf/128, m_y_lf/128, m_b_lf/128 (lf.rs:37-50)                  */
    uint32_t float_sample;       /* BitDepth::FloatSample (exp_bits), 0 = integer                     */
    uint32_t exp_bits;
    JxlGpuFilterParams filter;
    JxlGpuUpsampling upsampling;
    JxlGpuNoiseParams noise;
    JxlGpuColorParams color;
    /* Per-unit leaves (optional).  The reference specialises the MA tree for every decode unit before it decodes it:
     * `make_flat_tree(channel, stream_index, prev_channels)` (jxl-modular/src/ma.rs:38-41, image.rs:477-490) resolves the
     * decisions on the static properties 0 (channel index) and 1 (stream index), so a tree that splits on those two only is
     * a SINGLE NODE for every unit — with that unit's own predictor / multiplier / offset — and takes decode_single_node
     * (image.rs:553-562, 716-777).  Nothing in such a tree depends on decoded samples, so the entropy decode is still
     * separable.  num_unit_leaves = 0: every unit uses residual_predictor / residual_multiplier / residual_offset above.
     * Otherwise residuals are present whatever residual_predictor says, and `unit_leaves` has one entry per unit, in this
     * order: the TRANSFORMED channels in list order (meta channels first), channels with a zero dimension skipped; a
     * channel decoded whole (GlobalModular) is one unit; a grouped channel contributes ncols x nrows units in raster order
     * (gy * ncols + gx) where ncols / nrows = ceil(original size / group_dim), or / (8 group_dim) once both shifts reach 3
     * (image.rs:258-306) — subgrids that fall outside the (smaller) transformed channel are counted too and ignored.  A
     * wrong count is JXLGPU_ERR_INVALID_ARG at the first inverse.  The array is copied by jxlgpu_modular_upload.       */
    const JxlGpuMaLeaf* unit_leaves;
    uint32_t num_unit_leaves;
    /* ABI 24: the per-row / per-column leaves of the units marked JXLGPU_LEAF_BY_ROW / _BY_COLUMN in unit_leaves
     * (null / 0 when there are none).  Copied by jxlgpu_modular_upload.                                        */
    const JxlGpuMaLeaf* axis_leaves;
    uint32_t num_axis_leaves;
} JxlGpuModularDesc;

#define JXLGPU_STAGE_MODULAR_INVERSE 0x02u /* same bit as TRANSFORM: inverse Squeeze/RCT/Palette    */
#define JXLGPU_STAGE_MODULAR_TO_FLOAT 0x40u /* M5 / C5                                              */

int jxlgpu_modular_upload(jxlgpu_ctx* ctx, const JxlGpuModularDesc* desc, jxlgpu_frame** out_frame);
/* Integer result of the inverse transforms (bit-exact contract), copied to host buffers of the
 * frame's sample type: `planes[c]` has channels[c].width*height samples.                            */
int jxlgpu_modular_inverse(jxlgpu_ctx* ctx, jxlgpu_frame* frame, void* const* planes_or_null);
/* Inverse transforms + int->float + filters + upsampling + colour, like jxlgpu_vardct_render.       */
int jxlgpu_modular_render(jxlgpu_ctx* ctx, jxlgpu_frame* frame, uint32_t stages, const JxlGpuOut* out);
/* The same for a region of the output (see jxlgpu_vardct_render_region): the inverse transforms are
 * whole-image (Squeeze has no locality), everything after them runs on the rectangle.               */
int jxlgpu_modular_render_region(jxlgpu_ctx* ctx, jxlgpu_frame* frame, uint32_t stages, const JxlGpuRegion* region,
                                 const JxlGpuOut* out);

#ifdef __cplusplus
}
#endif
#endif /* JXLGPU_H_ */
