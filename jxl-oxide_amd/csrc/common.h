// Internal definitions shared by the host API (api.hip) and the kernels.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <deque>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/jxlgpu.h"

#define JXL_SQRT2F 1.41421356237309504880f

// Varblock classes: one kernel instantiation per pixel shape (W x H) of the transform.
// TransformType -> (bw, bh) follows jxl-vardct/src/dct_select.rs:52-76.
enum VbClass : int {
    CLS_DCT8 = 0,    // 8x8 DCT
    CLS_SPECIAL8,    // Hornuss, Dct2, Dct4, Dct4x8, Dct8x4, Afv0-3
    CLS_16x16,
    CLS_8x16,        // W=8,  H=16  (Dct16x8)
    CLS_16x8,        // W=16, H=8   (Dct8x16)
    CLS_32x32,
    CLS_8x32,        // Dct32x8
    CLS_32x8,        // Dct8x32
    CLS_16x32,       // Dct32x16
    CLS_32x16,       // Dct16x32
    CLS_64x64,
    CLS_32x64,       // Dct64x32
    CLS_64x32,       // Dct32x64
    CLS_BIG,         // >= 128 in either dimension: global-memory two-pass path
    CLS_COUNT
};

// HF coefficients AND the transform output live in HBM as 8x8 cells, channel-interleaved: cell
// (cx, cy) holds 3 x 64 words {X, Y, B}, each 8 rows of 8.  Every varblock, whatever
// its shape or alignment, then reads and writes whole 256-byte runs: a row-major plane would take
// 32-byte pieces of 128-byte lines from varblocks of different shape classes at different times
// (measured with tools/mem_probe.hip: 1.8 TB/s against 4.0 TB/s for scattered 8x8 varblocks).
// Word index of sample (px, py) of channel c:
__host__ __device__ inline size_t coeff_tiled_index(uint32_t px, uint32_t py, uint32_t c, uint32_t w8) {
    return ((((size_t)(py >> 3) * w8 + (px >> 3)) * 3 + c) << 6) + ((py & 7u) << 3) + (px & 7u);
}

struct TransformArgs {
    const int32_t* coeff;      // i32 coefficients, cell-tiled (coeff_tiled_index)
    float* pix;                // transform output, cell-tiled like `coeff` (coeff_tiled_index)
    const float* lf[3];        // LF planes after V1-V3, stride = w8
    const uint8_t* kind;       // frame-level BlockInfo plane, stride w8
    const int32_t* hf_mul;
    const float* kx_map;       // base_correlation_x + x_from_y/colour_factor, per 64x64 tile
    const float* kb_map;
    const float* dequant;      // all matrices, flat
    const uint32_t* deq_off;   // [27*3] offsets into `dequant` (device copy, per-lane lookups)
    uint32_t deq_off_v[27 * 3]; // the same table by value (kernarg: compile-time type -> scalar load)
    const float* sec64;        // sec_half(64/128/256)
    const float* sec128;
    const float* sec256;
    uint32_t pstride, w8, h8, w64;  // pstride: row stride of the big_tmp scratch planes
    float global_scale;        // as f32
    float qm_scale[3];
    float quant_bias[3];
    float quant_bias_numerator;
    float* big_tmp;            // 6 row-major planes of pstride x (h8*8): working storage of the >=128 path
    const float* deq_lut;      // 256 x quant_bias_numerator / k (k >= 2), or nullptr: divide
    const uint32_t* nz;        // JXLGPU_COEFF_GROUPED: every group's (dx | dy << 8 | coeff << 16) words; `coeff` is null then
    uint32_t rect[4];          // cells [x0, x1) x [y0, y1) = rect[0..3]: only varblocks touching it are transformed
                               // (region renders; {0, 0, 65535, 65535} otherwise)
#ifdef JXL_TR_PROFILE
    unsigned long long* prof;  // tools only (make PROF=1): per-phase s_memtime sums, 4 families x 16 slots
#endif
};

struct LfArgs {
    const void* lfq[3];        // frame-level quantised LF planes, [0]=X,[1]=Y,[2]=B (already reordered)
    uint32_t is_i16;
    const float* scale;        // per LF group x 3 channels
    float* out[3];
    uint32_t w8, h8, lf_groups_per_row, group_cells_x, group_cells_y;
    float kx, kb;              // CfL-LF factors
};

struct SmoothArgs {
    const float* in[3];
    float* out[3];
    uint32_t w8, h8;
    float lf_div[3];           // lf_x, lf_y, lf_b (vardct/mod.rs:420-422)
};

// Half-open pixel rectangle (region renders: jxlgpu_*_render_region).
struct PixRect {
    int x0, y0, x1, y1;
    bool empty() const { return x1 <= x0 || y1 <= y0; }
};

struct PlaneSet {
    float* p[3];
    uint32_t stride;
};

struct FilterArgs {
    const float* in[3];
    float* out[3];
    uint32_t in_stride, out_stride;
    uint32_t width, height;
    const float* sigma;        // per 8x8 cell, stride sigma_stride
    uint32_t sigma_stride;
    JxlGpuFilterParams fp;
};

struct ColorArgs {
    float opsin_bias[3];
    float cbrt_opsin_bias[3];
    float itscale;
    float intensity_target;
    float matrix[9];
    uint32_t gamut_map;
    float gamut_lum[3];
    float gamut_sat;
    uint32_t has_matrix2;
    float matrix2[9];
    uint32_t tf;
    uint32_t ycbcr;            // do_ycbcr frames: ycbcr_to_rgb instead of the XYB op list
    float gamma;
    // ToneMapRec2408 (detect_peak = false): constants of rec2408_eetf_generic steps 1-2
    uint32_t tone_map;
    float tm_lum[3];
    float tm_lum0_pq, tm_source_pq_diff, tm_min_luminance, tm_max_luminance, tm_ks, tm_one_sub_ks, tm_scale;
    uint32_t tm_gamut_map;     // honoured with or without tone_map (PQ image, HLG target, intensity_target ~ 1000)
    float tm_gamut_sat;
    // HLG targets: inverse OOTF (tf.rs:118-143) between the tone map and its GamutMap; hlg_exp = (1 - gamma) / gamma with the
    // system gamma evaluated once per frame on the host (platform libm, as the reference does)
    uint32_t hlg_ootf;
    float hlg_exp;
    float hlg_lum[3];
    // the op list has something only the staged colour kernel evaluates (color_pixel_t<true>): the fused post / upsampling
    // kernels run without their colour epilogue and launch_color follows (run_post_stages)
    uint32_t staged_only;
};

// Fused post stage (fused_kernels.hip): Gabor -> EPF -> colour in one pass.
struct FusedArgs {
    const float* in[3];      // row-major planes, or in[0] = the cell-tiled transform output (in_w8 != 0)
    float* out[3];
    uint32_t in_stride, out_stride;
    uint32_t in_w8;          // cells per row of the tiled input (coeff_tiled_index)
    int width, height;
    const float* sigma;
    uint32_t sigma_stride;
    JxlGpuFilterParams fp;
    ColorArgs color;
    uint32_t do_color;
    const uint32_t* tiles;   // optional list of tiles (tx | ty << 16); null = full 2-D grid
    // streaming kernel geometry: it covers [sx0, sx1) x [sy0, sy1), strictly inside the image
    int sx0, sx1, sy0, sy1, rows_per_seg, strips, segs;
    uint32_t n_ring_h;       // ring tile list: the first n_ring_h tiles are 32 x 16, the rest 16 x 32
    uint32_t pk;             // the geometry is the packed kernel's (strips of 120 columns, two per lane)
    uint32_t tb;             // packed kernel, round 6: it takes the TOP and BOTTOM image borders itself (sy0 = 0, sy1 = height:
                             // every stage's mirrored rows are patched into the row rings, post_pk.inc); the tile list then holds
                             // the left / right 16-px columns only
    // Modular XYB frames: in[] = the INTEGER planes of the inverse transforms in channel order (Y, X, B), in_stride in samples;
    // the loaders convert on the fly (convert_to_float_modular_xyb, jxl-render/src/image.rs:148-189) — no float copy of the
    // frame is made.  0: f32 input; 1: int16 planes; 2: int32 planes.  in_m = m_lf_unscaled (X, Y, B).
    uint32_t in_int;
    float in_m[3];
};

// Workgroups of one transform launch: class k owns workgroups [wg_begin[k], wg_begin[k + 1]).
struct ClassTable {
    uint32_t n_classes;
    uint32_t wg_begin[6];    // n_classes + 1 entries used
    uint32_t cls[5];
    uint32_t first_entry[5]; // into `entries`
    uint32_t count[5];       // varblocks of the class
};

// Batched launches (jxlgpu_vardct_render_batch): every frame keeps a device-resident copy of the
// arguments of its default pipeline (all stages, Gabor + EPF iters 2 through the streaming kernel,
// colour fused); a launch takes up to JXLGPU_MAX_BATCH pointers to those blocks by value and picks
// its frame with blockIdx.y / .z, so N frames cost one launch per stage instead of N.
struct FrameDev {
    LfArgs lf;
    SmoothArgs smooth;
    uint32_t skip_smooth;
    TransformArgs tr;
    ClassTable ct[4];
    const uint4* entries;
    const uint32_t* nzc;     // list-fed frames: per entry, non-zero counts Y | X << 16 (B: entry.y >> 16; first word: entry.w)
    uint32_t special_first, special_count;
    FusedArgs post;
    uint32_t n_ring_tiles;
};
#ifndef JXLGPU_MAX_BATCH_N
#define JXLGPU_MAX_BATCH_N 32
#endif
constexpr int JXLGPU_MAX_BATCH = JXLGPU_MAX_BATCH_N;
struct FrameBatch {
    const FrameDev* f[JXLGPU_MAX_BATCH];
};
// The blocks are read-only for the kernels: constant address space makes every field a scalar load.
typedef const FrameDev __attribute__((address_space(4))) * FrameDevC;
// by-value copy of one member of a block (the compiler keeps only the fields a kernel uses, as SGPRs)
template <typename T>
__device__ __forceinline__ T load_const(const T __attribute__((address_space(4))) * p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}

// Kernel groups that can be bracketed with HIP events (jxlgpu_profile_*).
enum ProfGroup : int { PROF_LF = 0, PROF_TRANSFORM = 1, PROF_POST = 2, PROF_MODULAR = 3, PROF_COUNT = 4 };

// Tuning / debug switches.  Read from the environment ONCE, at jxlgpu_create, into the context:
// the library keeps no process-global mutable state (include/jxlgpu.h "Threading").
// Issue priority of the waves of the batched V1-V8 launches (s_setprio 0..3): they share their SIMDs with the waves of the
// post kernel of the previous chunk, which always have a VALU instruction ready.  Measured (round 5): 109.2 us per frame with
// priority 3, 110.7 with 1, 108.0 with 0 — no effect, so it stays off; the switch remains for experiments.
#ifndef JXL_TR_PRIO
#define JXL_TR_PRIO 0
#endif
#define JXL_SET_TR_PRIO() do { if (JXL_TR_PRIO > 0) __builtin_amdgcn_s_setprio(JXL_TR_PRIO); } while (0)

struct Tuning {
    int stream_rows = 48;        // JXLGPU_STREAM_ROWS: rows per wave segment of post_stream_kernel
    int batch_stream_rows = 0;   // JXLGPU_BATCH_STREAM_ROWS: the same for batched launches (0: one resident round of waves per launch)
    int batch_chunk = 0;         // JXLGPU_BATCH_CHUNK: > 0: frames per launch of a batch (default: JXLGPU_MAX_BATCH)
    bool int_post = false;       // JXLGPU_INT_POST=1: the post stage of Modular XYB frames reads the integer planes (no float copy by to_float_kernel);
                                 // measured on config 3: 14.7 vs 15.2-15.4 GP/s (the VALU-bound post kernel pays more for the conversion than the copy costs): off
    int batch_lf_mode = 0;       // JXLGPU_BATCH_LF_MODE (experiment, round 6): where the LF launches (V1-V3) of a chunk of a batched render sit: 0 = in front of
                                 // its own transform launches (the default), 1 = in front of the previous chunk's, 2 = behind the previous chunk's
                                 // 8- / 16-px launches.  Measured (64 frames, one box): 104.3-104.4 / 106.0 / 109.3-113.9 us per frame — moving the
                                 // two tiny launches out from behind the border-ring launch only moves the wait to the launches that follow
    int post_lds_pad = 0;        // JXLGPU_POST_LDS_PAD=bytes (experiment, round 6): dynamic LDS reserved per workgroup of the batched packed post
                                 // launch — 81920 leaves ONE workgroup = one post wave per SIMD on a CU: what a one-pass kernel that keeps a
                                 // 64-row window of the transform output in LDS would have to live with
    int batch_tr_mult = 1;       // JXLGPU_BATCH_TR_MULT: chunks per LF / transform launch of a batched render (post launches: one chunk)
    int tr_side_max = 16;        // JXLGPU_TR_SIDE_MAX: launches of <= this many frames run the big-shape transforms on the side stream
                                 // (short launches: their tails overlap; +2.7 % at 8 frames per launch, nothing at 32)
    bool no_pk = false;          // JXLGPU_NO_PK: scalar streaming kernel (one column per lane)
    bool pk_tb = false;          // JXLGPU_PK_TB=1 (round 6, measured, not the default): the packed streaming kernel takes the top / bottom image rows
                                 // itself (mirrored rows patched into its row rings, post_pk.inc) and the ring-tile kernel only the left / right
                                 // columns (-64 % of its tiles).  Bit-identical (tests/test_gpu_schedules.py runs both forms); 115.1-115.2 against
                                 // 114.1-114.2 us per frame on one box, 112.5-112.9 against 111.5-112.0 on another: the 32 extra rows sit in the
                                 // launch that sets the period, the ring tiles they replace ran beside it
    bool no_stream = false;      // JXLGPU_NO_STREAM: LDS tile kernel for the whole frame
    bool no_fused = false;       // JXLGPU_NO_FUSED: one kernel per post stage
    bool no_sparse_tr = false;   // JXLGPU_NO_SPARSE_TR: grouped lists are expanded to dense cells first (dense kernels)
    bool debug_sync = false;     // JXLGPU_DEBUG_SYNC: synchronise + report after every launch group
    bool no_batch_overlap = false; // JXLGPU_NO_BATCH_OVERLAP: batched renders on one stream, stage after stage (round-3 form)
    int ring_mode = 0;           // JXLGPU_RING_MODE: border-ring launch of a batched chunk: 0 = on the side stream beside the streaming post launch,
                                 // 1 = on the render stream in front of it, 2 = behind it.  (Round 5 kernel trace: beside it the ring launch, an
                                 // LDS tile kernel of 56 VGPRs, takes the registers two post waves per SIMD leave over for ~0.65 ms and the LF
                                 // launch of the next chunk waits behind it; in front of it the transform launches start earlier but the post
                                 // launch they then share the chip with runs 20 % longer: 111 / 114-125 us per frame, mode 0 stays.)
    bool post_fast = false;      // JXLGPU_POST_FAST: batched default pipeline through post_pk_fast_batch_kernel (NOT bit-exact: a measured option)
    int tr_streams = 2;          // JXLGPU_TR_STREAMS: streams the transform families of a batched chunk are spread over (2: the 8 / 16-px families
                                 // on the transform stream, the rest on its side stream; up to 5: one stream per family — an experiment)
    uint32_t batch_heavy = 0;    // JXLGPU_BATCH_HEAVY: mask of transform families (bit F = family F, bit 4 = special 8x8) of chunk k that
                                 // run on the RENDER stream between post(k-1) and post(k) instead of on the transform stream beside
                                 // post(k-1): families whose waves do not fit beside two post waves per SIMD displace them
    int tr_wgs_per_cu[4] = {0, 0, 0, 0};  // JXLGPU_TR_WGS_PER_CU="a,b,c,d": persistent transform workgroups per CU
                                 // for the 8-, 16-, 32- and 64-px launch (0: one workgroup per item, no run-ahead)
    int sqz_seg = 64;            // JXLGPU_SQZ_SEG: pairs per inverse-Squeeze segment
    uint32_t sqz_runin = 1;      // JXLGPU_SQZ_RUNIN: 0 forces the Squeeze fix-up path (tests)
    bool sqz_h_rows = false;     // JXLGPU_SQZ_H_ROWS: horizontal Squeeze steps through the lane-per-row segment kernel
    bool pred_wide = false;      // JXLGPU_PRED_WIDE: the self-correcting predictor in 64-bit arithmetic only
    bool pred_step_v1 = false;   // JXLGPU_PRED_STEP_V1=1: the round-3 step (x-indexed error rows, position-indexed rings) also where every wave has D = 4
    int pred_late_steps = 3;     // JXLGPU_PRED_LATE_STEPS: residuals of the first so many (forward) Squeeze steps get their predictor waves on a
                                 // side stream, beside the deep Squeeze levels; the inverse step that reads them waits (0: everything in front)
    bool pred_prio = false;      // JXLGPU_PRED_PRIO=1 (round 6, measured, not adopted): issue priority by chain length in the narrow predictor kernel
                                 // (s_setprio 3 .. 0 for the longest quarter .. the short waves).  Config 3: 14.1-14.2 against 15.2-15.3 GP/s — the waves
                                 // of the misaligned subgrids on the side stream (no priority of their own) take 0.48 instead of 0.30 ms
    bool pred_wg = false;        // JXLGPU_PRED_WG: predictor subgrids through the workgroup-per-subgrid kernel only
    int up2_variant = 0;         // JXLGPU_UP2_VARIANT: 1 = register-ring form of the 2x upsampling kernel, 2 = LDS ring with the
                                 // general colour code only (0: LDS ring, packed colour chain for the HDR PQ op list)
    int up2_rows = 0;            // JXLGPU_UP2_ROWS: rows per wave segment of that kernel (0: one resident round)
};

// One guarded device buffer (JXLGPU_GUARD, api.hip guard_malloc)
struct GuardRec {
    void* base = nullptr;       // reserved range: [guard granule][mapping][guard granule]
    size_t reserved = 0, mapped = 0;
    hipMemGenericAllocationHandle_t handle = {};
};

// Host worker threads of a context (api.hip): the per-frame work-list build of an upload is split over them.
struct WorkerPool;
// Pinned staging buffer of the upload arena: the host build writes every descriptor array of a frame into one of
// these, ONE hipMemcpyAsync moves it; `ev` (recorded behind that copy) guards its reuse.
struct StageBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool busy = false;
};
// Device buffers (and the Modular host state) of a freed frame, waiting for the work that was queued when the
// frame was freed: jxlgpu_frame_free never blocks, the buffers go back to the pool once `ev` have all fired.
struct Deferred {
    std::vector<void*> ptrs;
    hipEvent_t ev[6] = {};
    void* modular = nullptr;
    void (*modular_free)(void*) = nullptr;
};

struct jxlgpu_ctx {
    int device = 0;
    uint32_t num_cus = 256;     // hipDeviceProp_t::multiProcessorCount (persistent grids are sized from it)
    Tuning tune;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // side stream: 64-pixel varblock kernels overlap the <=32 kernel
    hipStream_t stream_tr = nullptr;    // batched renders: V1-V8 of chunk k+1 beside the post stage of chunk k
    hipStream_t stream_tr_extra[3] = {};   // JXLGPU_TR_STREAMS > 2 (created on first use; forked from / joined into stream_tr)
    hipEvent_t ev_tr_join[5] = {};
    hipStream_t stream_tr2 = nullptr;   // ... its 32 / 64-px and special launches (forked from and joined into stream_tr: never stream2, where
                                        // the border-ring launch of chunk k would sit in front of the heavy transforms of chunk k+1)
    hipEvent_t ev_tr[8] = {};
    uint32_t ev_tr_next = 0;
    hipStream_t stream_up = nullptr;    // H2D of the upload arenas: overlaps the kernels of earlier frames
    hipStream_t stream_down = nullptr;  // asynchronous D2H of formatted output (JXLGPU_MEM_HOST_PINNED)
    StageBuf stage[3];
    uint32_t stage_next = 0;
    StageBuf rt_stage[4];               // region renders: tile lists on their way to the device
    uint32_t rt_stage_next = 0;
    WorkerPool* workers = nullptr;      // created by the first upload that is worth splitting
    int host_threads = -1;              // JXLGPU_HOST_THREADS (-1: min(8, cores) - 1 workers + the calling thread)
    std::deque<Deferred> deferred;
    std::vector<hipEvent_t> ev_spare;   // recycled events of reaped Deferred entries
    double up_split[4] = {};            // last upload: host build, staging fill, alloc + enqueue, whole call (ms)
    hipEvent_t ev_h2d[2] = {};          // around the last upload's arena copy (jxlgpu_upload_split reads the elapsed time)
    bool h2d_timed = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_d2h[3] = {};      // one per output plane: host copy-out overlaps the next plane's D2H
    hipEvent_t ev_slice[4] = {};    // formatted output: one per D2H slice
    std::string last_error;
    // Device-buffer pool: frames of one stream of images have the same sizes, so the ~25 buffers of
    // a freed frame are handed to the next upload instead of going through hipFree/hipMalloc
    // (each an implicit device-wide synchronisation).  Exact-size reuse; capped (JXLGPU_POOL_MB).
    std::unordered_map<void*, size_t> live;
    std::multimap<size_t, void*> pool;
    size_t pool_bytes = 0, pool_cap = (size_t)8 << 30;
    size_t live_bytes = 0, mem_limit = 0;   // jxlgpu_set_memory_limit: bound on live_bytes (0: none)
    int guard_mode = 0;         // JXLGPU_GUARD: 1 = buffers end at an unmapped page, 2 = start after one
    size_t guard_gran = 0;
    int guard_seq = 0, guard_zero = -1;   // JXLGPU_GUARD_ZERO=k|all: allocation #k (all) is zero-filled instead of poisoned
    bool guard_log = false;               // JXLGPU_GUARD_LOG: one stderr line per allocation
    std::unordered_map<void*, GuardRec> guard_live;
#ifdef JXL_TR_PROFILE
    unsigned long long* tr_prof = nullptr;
#endif
    void* noise_jump = nullptr; // device copy of the xorshift128+ jump matrices (noise_kernels.hip)
    void* pinned = nullptr;     // pinned staging buffer (grown on demand)
    size_t pinned_size = 0;
    // event-pair profiling of one kernel group on the ctx stream
    int prof_group = -1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    // jxlgpu_set_trace: the reference's span of every launch group, on the calling thread (begin / end of the ENQUEUE)
    void (*trace_fn)(void* user, const char* span, int begin) = nullptr;
    void* trace_user = nullptr;
    static const char* span_name(int g) {
        // jxl-render/src/vardct/mod.rs:164 ("Load LF groups": LF dequant + CfL + adaptive smoothing sit inside it), :316,
        // filter/epf.rs:21 (the fused launch group also holds the Gabor-like stage and the colour transform), modular.rs:134
        static const char* const kNames[PROF_COUNT] = {"Load LF groups", "Dequant and transform", "Edge-preserving filter", "Inverse Modular transform"};
        return g >= 0 && g < PROF_COUNT ? kNames[g] : "?";
    }
    void prof_begin(int g, hipStream_t on = nullptr) {
        if (trace_fn) trace_fn(trace_user, span_name(g), 1);
        if (g != prof_group) return;
        if (!on) on = stream;
        if (prof_used == prof_events.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            prof_events.emplace_back(a, b);
        }
        (void)hipEventRecord(prof_events[prof_used].first, on);
    }
    void prof_end(int g, hipStream_t on = nullptr) {
        if (trace_fn) trace_fn(trace_user, span_name(g), 0);
        if (g != prof_group) return;
        if (!on) on = stream;
        (void)hipEventRecord(prof_events[prof_used].second, on);
        ++prof_used;
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct jxlgpu_frame {
    // set around run_post_stages by the Modular render: the post stage's input planes are integers (FusedArgs::in_int / in_m)
    uint32_t post_in_int = 0;
    float post_in_m[3] = {0.0f, 0.0f, 0.0f};
    int kind_of_frame = 0;      // 0 = VarDCT, 1 = Modular
    // geometry
    uint32_t width = 0, height = 0, w8 = 0, h8 = 0, wr = 0, hr = 0, w64 = 0, h64 = 0;
    uint32_t group_dim = 256, lf_groups_per_row = 1, num_lf_groups = 1;
    std::vector<void*> allocs;  // everything hipMalloc'ed for this frame
    // VarDCT device state
    int32_t* coeff = nullptr;   // 3 * wr * hr words, cell-tiled (coeff_tiled_index)
    void* lfq[3] = {};
    uint32_t lf_is_i16 = 0;
    float* lf_scale = nullptr;
    uint8_t* kind = nullptr;
    int32_t* hf_mul = nullptr;
    float* sigma = nullptr;
    float* kx_map = nullptr;
    float* kb_map = nullptr;
    float* dequant = nullptr;
    uint32_t* deq_off = nullptr;
    uint32_t deq_off_host[JXLGPU_NUM_TRANSFORMS][3] = {};
    float* sec[3] = {};
    float* lf_a[3] = {};        // after V1+V2
    float* lf[3] = {};          // after V3 (== lf_a when smoothing is skipped)
    float* pix_t = nullptr;     // transform output, 3 * wr * hr words, cell-tiled (coeff_tiled_index)
    float* pix[3] = {};         // chroma-subsampled parent only: row-major planes after upsample_jpeg
    float* buf_a[3] = {};       // filter ping
    float* buf_b[3] = {};       // filter pong
    float* big_tmp = nullptr;  // scratch for the >=128 transform path (aliases buf_a[0])
    float* up[3] = {};          // upsampled planes
    float* up_tmp[3] = {};
    uint4* entries = nullptr;            // all varblocks, classes concatenated
    // JXLGPU_COEFF_GROUPED: the decoder's non-zero lists, consumed by the list-fed transform kernels
    // (transform_sparse.hip); `coeff` stays null unless the frame falls back to the dense kernels
    uint32_t* nz = nullptr;              // all groups' list words
    uint32_t* nzc = nullptr;             // per entry: count Y | count X << 16
    uint64_t nz_total = 0;
    bool sparse_tr = false;              // V4-V8 run the list-fed kernels
    bool lf_from_frame = false;          // the LF image came from an LF frame (descriptor lf_frame): V1-V3 are skipped
    uint32_t class_first[CLS_COUNT] = {};
    uint32_t list_count[CLS_COUNT] = {};
    bool has_no_meta_groups = false;
    uint32_t* nometa_groups = nullptr;  // groups whose LF group has no HfMetadata
    uint32_t nometa_count = 0;
    // params
    JxlGpuVardctDesc desc = {}; // scalar copy (pointers invalid after upload)
    float qm_scale[3] = {1, 1, 1};
    float kx_lf = 0, kb_lf = 0;
    float lf_div[3] = {};
    ColorArgs color = {};
    void* fmt_buf = nullptr;             // device staging for jxlgpu_frame_format_output
    size_t fmt_bytes = 0;
    // chroma-subsampled parent: one single-geometry child per distinct (hshift, vshift)
    struct Sub {
        jxlgpu_frame* child = nullptr;
        int hshift = 0, vshift = 0;
        bool member[3] = {false, false, false};
    };
    std::vector<Sub> subs;
    uint32_t lfg_cells_x = 256, lfg_cells_y = 256;  // LF group size in cells
    float* noise_raw[3] = {};            // raw noise planes (allocated on the first noise render)
    uint32_t noise_w = 0, noise_h = 0;
    uint32_t noise_group_dim = 256;
    float noise_corr_x = 0.0f, noise_corr_b = 1.0f;  // base_correlations_xb (render.rs:175-180)
    float* deq_lut = nullptr;            // quant_bias_numerator / k, k < 256 (dequant_one_lut)
    hipEvent_t ev_last = nullptr;        // behind the last asynchronous operation queued for this frame (jxlgpu_frame_wait)
    bool ev_last_set = false;
    hipEvent_t ev_fmt = nullptr;         // behind the last download-stream copy that reads fmt_buf (JXLGPU_MEM_HOST_PINNED): the next
    bool ev_fmt_set = false;             // formatting kernel into fmt_buf waits for it
    FrameDev* dev_args = nullptr;        // device copy of the default pipeline's arguments (batched launches)
    bool dev_args_ready = false;
    bool batch_ok = false;               // the frame qualifies for the batched default pipeline (V1-V8 and post)
    bool batch_tr_ok = false;            // V1-V8 of the frame can share launches (any post pipeline)
    uint32_t batch_wgs[4] = {}, batch_stream_wgs = 0;
    bool batch_pk = false;               // the batched post launch of this frame is the packed kernel
    uint32_t* ring_tiles = nullptr;      // border ring of the streaming post path: tile origins x0 | y0 << 16
    uint32_t n_ring_tiles = 0, n_ring_h = 0;  // the first n_ring_h are 32 x 16 (top / bottom), the rest 16 x 32
    std::vector<uint32_t> ring_host;     // the same list on the host (region renders launch the tiles they touch)
    uint32_t ring_tb = 0;                // the list was built for a streaming kernel that takes the top / bottom borders itself (FusedArgs::tb)
    uint32_t* region_tiles[2] = {};      // region renders: tile lists of the launches in flight
    size_t region_tiles_cap[2] = {};
    float* up_weights[3] = {};  // expanded 5x5 kernels per phase for 2x/4x/8x
    float up2_wq[25] = {};      // host copy of the 2x kernel (kernel argument of the streaming form)
    bool have_up2 = false;
    // result of the last render
    const float* result[3] = {};
    uint32_t result_stride = 0, result_w = 0, result_h = 0;
    // extra channels (jxlgpu_frame_render_extra): device planes, tight rows
    float* extra[JXLGPU_MAX_EXTRA] = {};
    uint32_t extra_w[JXLGPU_MAX_EXTRA] = {}, extra_h[JXLGPU_MAX_EXTRA] = {};
    // modular state lives in modular.hip's own struct
    void* modular = nullptr;
    void (*modular_free)(void*) = nullptr;
};

#define HIP_TRY(ctx, expr)                                                           \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);  \
            return e_ == hipErrorOutOfMemory ? JXLGPU_ERR_OOM : JXLGPU_ERR_DEVICE;   \
        }                                                                            \
    } while (0)

static inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// kernel launchers implemented in the .hip files
void launch_lf_dequant_cfl(hipStream_t s, const LfArgs& a);
void launch_lf_smooth(hipStream_t s, const SmoothArgs& a);
void launch_transform_class(hipStream_t s, int cls, const TransformArgs& a, const uint4* entries,
                            uint32_t count);
// batched launches over FrameDev blocks (n <= JXLGPU_MAX_BATCH)
void build_class_table(int family, const uint32_t class_first[CLS_COUNT], const uint32_t list_count[CLS_COUNT],
                       uint32_t num_cus, int wgs_per_cu, ClassTable* ct);
hipError_t launch_lf_batch(hipStream_t s, const FrameBatch& b, uint32_t n, uint32_t max_w8, uint32_t max_h8, bool any_smooth);
// `mask`: bit F = launch family F (0: 8-px ... 3: 64-px), bit 4 = the special 8x8 family
hipError_t launch_transform_batch(hipStream_t s, hipStream_t side, const FrameBatch& b, uint32_t n, const uint32_t max_wgs[4],
                                  uint32_t max_special, uint32_t mask = 31u);
hipError_t launch_post_batch(hipStream_t s, hipStream_t side, const FrameBatch& b, uint32_t n, uint32_t max_stream_wgs,
                             uint32_t max_ring, bool pk, bool fast = false, int lds_pad = 0);
hipError_t launch_transform_items(hipStream_t s, int family, const TransformArgs& a, const uint4* entries,
                                  const uint32_t class_first[CLS_COUNT], const uint32_t list_count[CLS_COUNT],
                                  uint32_t num_cus, int wgs_per_cu);
// transform_sparse.hip: the same launches for list-fed frames (JXLGPU_COEFF_GROUPED)
hipError_t launch_transform_items_sparse(hipStream_t s, int family, const TransformArgs& a, const uint4* entries,
                                         const uint32_t* nzc, const uint32_t class_first[CLS_COUNT],
                                         const uint32_t list_count[CLS_COUNT], uint32_t num_cus);
void launch_transform_special_sparse(hipStream_t s, const TransformArgs& a, const uint4* entries, const uint32_t* nzc,
                                     uint32_t count);
hipError_t launch_transform_batch_sparse(hipStream_t s, hipStream_t side, const FrameBatch& b, uint32_t n,
                                         const uint32_t max_wgs[4], uint32_t max_special, uint32_t mask = 31u);
// grouped lists -> dense cell-tiled coefficients (fallback for frames with >= 128-px varblocks; `coeff` zeroed first)
void launch_grouped_to_dense(hipStream_t s, const uint4* entries, const uint32_t* nzc, uint32_t n_entries,
                             const uint32_t* nz, uint32_t w8, int32_t* coeff, bool accumulate);
int transform_items_nbi(int cls);
void launch_nometa_groups(hipStream_t s, const TransformArgs& a, const uint32_t* groups,
                          uint32_t count, uint32_t group_dim, uint32_t groups_per_row);
void launch_gabor(hipStream_t s, const FilterArgs& a);
void launch_epf(hipStream_t s, int step, const FilterArgs& a);
// api.hip: pooled device memory (see jxlgpu_ctx::pool)
hipError_t ctx_dev_malloc(jxlgpu_ctx* ctx, void** out, size_t bytes);
void ctx_dev_release(jxlgpu_ctx* ctx, void* p);
// release `ptrs` once everything queued so far on the ctx's streams has finished (never blocks)
void ctx_defer_release(jxlgpu_ctx* ctx, std::vector<void*>&& ptrs, void* modular = nullptr, void (*modular_free)(void*) = nullptr);
void ctx_reap(jxlgpu_ctx* ctx, bool wait);
// upsample_inner's weights_quarter (features/upsampling.rs:77-93) from the 15 / 55 / 210 coded weights
std::vector<float> expand_up_weights_public(const float* weights, int k);
// f(0) ... f(n - 1) on the ctx's host worker threads + the caller; returns when all are done
void ctx_host_parallel(jxlgpu_ctx* ctx, uint32_t n, const std::function<void(uint32_t)>& f);
// mark "the frame's last queued operation is here" on stream `s` (jxlgpu_frame_wait)
void frame_mark(jxlgpu_ctx* ctx, jxlgpu_frame* f, hipStream_t s);
size_t noise_jump_table_bytes();
const void* noise_jump_table_host();
bool noise_geometry_unsupported(uint32_t height, uint32_t group_dim);
void launch_noise(hipStream_t s, const JxlGpuNoiseParams& np, const void* jump_dev, float* const raw[3],
                  float* const ch[3], uint32_t stride, uint32_t width, uint32_t height, uint32_t group_dim,
                  float corr_x, float corr_b);
void launch_upsample_jpeg(hipStream_t s, const float* in_tiled, uint32_t in_w8, uint32_t c, uint32_t in_w, uint32_t in_h,
                          int hshift, int vshift, float* out, uint32_t out_stride, uint32_t width, uint32_t height);
void launch_untile(hipStream_t s, const float* tiled, uint32_t w8, float* const out[3], uint32_t out_stride,
                   uint32_t width, uint32_t height);
void launch_coeff_retile(hipStream_t s, const void* src, bool src_i16, uint32_t wr, uint32_t hr, uint32_t c,
                         int32_t* dst);
void launch_coeff_scatter(hipStream_t s, const uint32_t* pos, const void* val, bool val_i16, size_t count,
                          uint32_t src_stride, uint32_t wr, uint32_t hr, uint32_t c, int32_t* dst, uint32_t* bad);
void launch_color(hipStream_t s, const ColorArgs& c, float* const planes[3], uint32_t stride,
                  uint32_t width, uint32_t height);
bool launch_upsample2_stream(hipStream_t s, const float* const in[3], uint32_t in_stride, uint32_t w, uint32_t h,
                             float* const out[3], uint32_t out_stride, const float* weights_quarter_host,
                             const ColorArgs* color, const PixRect* window, uint32_t num_cus, int variant, int rows);
void launch_upsample(hipStream_t s, const float* in, uint32_t in_stride, uint32_t w, uint32_t h,
                     float* out, uint32_t out_stride, int k, const float* kernels, const PixRect* window = nullptr);
