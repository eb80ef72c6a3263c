// Host side of libjxlgpu.so: the C ABI of include/jxlgpu.h.  Uploads one frame's decoded state,
// builds the per-shape varblock work lists (the device-side replacement for the reference's
// serial `for_each_varblocks` scan, jxl-render/src/vardct/mod.rs:693-730) and sequences the
// kernels on the context's HIP stream.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

#include "common.h"
#include "pixel_device.h"  // linear_to_pq_dev (host+device) for the tone-map constants

// JXLGPU_GUARD (debug): every device buffer gets its own virtual-memory mapping with an unmapped
// granule on each side, so that an access outside the buffer is a GPU page fault with the kernel's
// name on it (AMD_LOG_LEVEL=3) instead of a silent read of a neighbour.  Mode 1: the buffer ENDS at
// the end of the mapping (bytes rounded up to 16: the kernels' widest access) — overruns fault;
// mode 2: it STARTS at the start of the mapping — underruns fault; mode 3: mode 1 with the end rounded
// up to 4 bytes only.  No pooling: a freed buffer is
// unmapped at once, so a use-after-free faults too.
static hipError_t guard_malloc(jxlgpu_ctx* ctx, void** out, size_t bytes) {
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    if (!ctx->guard_gran) {
        size_t g = 0;
        hipError_t e = hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess) return e;
        ctx->guard_gran = g ? g : 4096;
    }
    const size_t gran = ctx->guard_gran;
    // mode 3: as mode 1 with the end rounded up to 4 bytes only — an access of 4 or 8 bytes past the end of a buffer whose
    // size is not a multiple of 16 faults too (the buffer then starts 4-byte aligned only)
    const size_t user = ctx->guard_mode == 3 ? (bytes + 3) & ~(size_t)3 : (bytes + 15) & ~(size_t)15;
    GuardRec r;
    r.mapped = (user + gran - 1) / gran * gran;
    r.reserved = r.mapped + 2 * gran;
    hipError_t e = hipMemAddressReserve(&r.base, r.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&r.handle, r.mapped, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(r.base, r.reserved); return e; }
    char* at = static_cast<char*>(r.base) + gran;
    e = hipMemMap(at, r.mapped, 0, r.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc ad;
        memset(&ad, 0, sizeof(ad));
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(at, r.mapped, &ad, 1);
        if (e != hipSuccess) (void)hipMemUnmap(at, r.mapped);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(r.handle);
        (void)hipMemAddressFree(r.base, r.reserved);
        return e;
    }
    // poison: a kernel that reads slack inside the mapping (mode 2: behind the buffer) sees NaNs, not zeros
    // (device-wide sync: hipMemset on device memory may return before the fill has run, and the ctx streams
    //  are non-blocking — they do not order behind the null stream)
    const int seq = ctx->guard_seq++;
    const bool zero = ctx->guard_zero == -2 || ctx->guard_zero == seq;   // bisecting reads of uninitialised memory
    (void)hipMemset(at, zero ? 0x00 : 0xff, r.mapped);
    (void)hipDeviceSynchronize();
    if (ctx->guard_log) fprintf(stderr, "[jxlgpu guard] alloc #%d: %zu bytes%s\n", seq, bytes, zero ? " (zero-filled)" : "");
    *out = ctx->guard_mode == 2 ? at : at + (r.mapped - user);
    ctx->guard_live[*out] = r;
    return hipSuccess;
}

static void guard_free(jxlgpu_ctx* ctx, void* p) {
    auto it = ctx->guard_live.find(p);
    if (it == ctx->guard_live.end()) return;
    const GuardRec r = it->second;
    ctx->guard_live.erase(it);
    char* at = static_cast<char*>(r.base) + ctx->guard_gran;
    (void)hipMemUnmap(at, r.mapped);
    (void)hipMemRelease(r.handle);
    // The address range is NOT given back: on this runtime (ROCm 7.2) a range that is freed, reserved again and
    // mapped to new memory serves stale translations — tools/vmm_churn.hip, a pure-HIP program, gets hundreds of
    // millions of wrong words with hipMemAddressFree in its loop and none without it (profiles/r04_fault_hunt.md).
    // Never recycling a range also keeps every freed buffer's addresses unmapped for good: a use-after-free faults.
}

hipError_t ctx_dev_malloc(jxlgpu_ctx* ctx, void** out, size_t bytes) {
    bytes = std::max<size_t>(bytes, 16);
    if (ctx->mem_limit && ctx->live_bytes + bytes > ctx->mem_limit) {
        ctx_reap(ctx, true);  // frames being freed may still hold part of the budget
        if (ctx->live_bytes + bytes > ctx->mem_limit) return hipErrorOutOfMemory;   // AllocTracker::alloc -> OutOfMemory
    }
    if (ctx->guard_mode) {
        hipError_t e = guard_malloc(ctx, out, bytes);
        if (e == hipSuccess) { ctx->live[*out] = bytes; ctx->live_bytes += bytes; }
        return e;
    }
    auto it = ctx->pool.find(bytes);
    if (it != ctx->pool.end()) {
        *out = it->second;
        ctx->pool.erase(it);
        ctx->pool_bytes -= bytes;
        ctx->live[*out] = bytes;
        ctx->live_bytes += bytes;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {  // wait for the frames being freed, give the pooled memory back and retry
        (void)hipGetLastError();
        ctx_reap(ctx, true);
        for (auto& kv : ctx->pool) (void)hipFree(kv.second);
        ctx->pool.clear();
        ctx->pool_bytes = 0;
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) { ctx->live[*out] = bytes; ctx->live_bytes += bytes; }
    return e;
}

// The caller guarantees that no queued work still touches `p` (frame_free synchronises first).
void ctx_dev_release(jxlgpu_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->live.find(p);
    if (ctx->guard_mode) {
        if (it != ctx->live.end()) { ctx->live_bytes -= it->second; ctx->live.erase(it); }
        guard_free(ctx, p);
        return;
    }
    if (it == ctx->live.end()) { (void)hipFree(p); return; }
    const size_t bytes = it->second;
    ctx->live.erase(it);
    ctx->live_bytes -= bytes;
    if (ctx->pool_bytes + bytes <= ctx->pool_cap) {
        ctx->pool.emplace(bytes, p);
        ctx->pool_bytes += bytes;
    } else {
        (void)hipFree(p);
    }
}

// ---- host worker threads (one pool per context; the calling thread takes part)
struct WorkerPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(uint32_t)>* fn = nullptr;
    uint32_t n_tasks = 0, active = 0;
    std::atomic<uint32_t> next{0};
    uint64_t gen = 0;
    bool stop = false;
    explicit WorkerPool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) threads.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_work.notify_all();
        for (auto& t : threads) t.join();
    }
    void drain(const std::function<void(uint32_t)>& f) {
        for (;;) {
            const uint32_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_tasks) break;
            f(i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv_work.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            const std::function<void(uint32_t)>* f = fn;
            lk.unlock();
            drain(*f);
            lk.lock();
            if (--active == 0) cv_done.notify_one();
        }
    }
    // f(0) ... f(n - 1), each exactly once, on the pool + the caller; returns when all are done
    void run(uint32_t n, const std::function<void(uint32_t)>& f) {
        if (threads.empty() || n <= 1) {
            for (uint32_t i = 0; i < n; ++i) f(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            fn = &f; n_tasks = n; next.store(0); active = (uint32_t)threads.size(); ++gen;
        }
        cv_work.notify_all();
        drain(f);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

void ctx_host_parallel(jxlgpu_ctx* ctx, uint32_t n, const std::function<void(uint32_t)>& f) {
    if (!ctx->workers && ctx->host_threads != 0 && n > 1) {
        unsigned want = ctx->host_threads > 0 ? (unsigned)ctx->host_threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        ctx->workers = new (std::nothrow) WorkerPool(want > 1 ? want - 1 : 0);
    }
    if (ctx->workers) ctx->workers->run(n, f);
    else for (uint32_t i = 0; i < n; ++i) f(i);
}

static hipEvent_t ctx_event(jxlgpu_ctx* ctx) {
    if (!ctx->ev_spare.empty()) { hipEvent_t e = ctx->ev_spare.back(); ctx->ev_spare.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    return e;
}

void ctx_reap(jxlgpu_ctx* ctx, bool wait) {
    while (!ctx->deferred.empty()) {
        Deferred& d = ctx->deferred.front();
        bool done = true;
        for (hipEvent_t e : d.ev) {
            if (!e) continue;
            if (wait) (void)hipEventSynchronize(e);
            else if (hipEventQuery(e) != hipSuccess) { done = false; break; }
        }
        if (!done) { (void)hipGetLastError(); break; }  // hipErrorNotReady is not an error of ours; entries complete in order
        for (void* p : d.ptrs) ctx_dev_release(ctx, p);
        if (d.modular && d.modular_free) d.modular_free(d.modular);
        for (hipEvent_t e : d.ev) if (e) ctx->ev_spare.push_back(e);
        ctx->deferred.pop_front();
    }
}

void ctx_defer_release(jxlgpu_ctx* ctx, std::vector<void*>&& ptrs, void* modular, void (*modular_free)(void*)) {
    if (ptrs.empty() && !modular) return;
    Deferred d;
    d.ptrs = std::move(ptrs);
    d.modular = modular; d.modular_free = modular_free;
    hipStream_t st[6] = {ctx->stream, ctx->stream2, ctx->stream_up, ctx->stream_down, ctx->stream_tr, ctx->stream_tr2};
    bool ok = true;
    for (int i = 0; i < 6; ++i) {
        d.ev[i] = ctx_event(ctx);
        ok = ok && d.ev[i] && hipEventRecord(d.ev[i], st[i]) == hipSuccess;
    }
    if (!ok) {  // cannot track: fall back to draining the device
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        for (hipEvent_t& e : d.ev) { if (e) ctx->ev_spare.push_back(e); e = nullptr; }
    }
    ctx->deferred.push_back(std::move(d));
    ctx_reap(ctx, false);
}

void frame_mark(jxlgpu_ctx* ctx, jxlgpu_frame* f, hipStream_t s) {
    if (!f->ev_last && hipEventCreateWithFlags(&f->ev_last, hipEventDisableTiming) != hipSuccess) { f->ev_last = nullptr; return; }
    f->ev_last_set = hipEventRecord(f->ev_last, s) == hipSuccess;
}

hipError_t launch_fused_post(hipStream_t s, jxlgpu_frame* f, const float* const in[3], uint32_t in_stride,
                             uint32_t in_tiled_w8, float* const out[3], uint32_t out_stride, bool gabor, int epf_iters,
                             bool color, jxlgpu_ctx* ctx, const PixRect* rc = nullptr);
bool fused_post_supported(const jxlgpu_ctx* ctx, const jxlgpu_frame* f, bool gabor, int epf_iters);
hipError_t fused_prepare(jxlgpu_ctx* ctx, jxlgpu_frame* f, const float* const in[3], uint32_t in_stride,
                         uint32_t in_tiled_w8, float* const out[3], uint32_t out_stride, bool gabor, int epf_iters,
                         bool color, FusedArgs* pa, bool* stream_out, bool* plain_srgb, int rows_per_seg);

struct UploadOpts {
    uint32_t lfg_cells_x = 0, lfg_cells_y = 0;  // LF group size in cells (0: group_dim)
    bool no_cfl = false;                        // chroma-subsampled frames skip both CfL steps
    bool no_post = false;                       // no filter buffers: the frame only runs V1-V8
};
extern "C" int vardct_upload_impl(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* d, const UploadOpts& o, jxlgpu_frame** out_frame);
int upload_subsampled(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* d, jxlgpu_frame** out_frame);
int render_subsampled(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuOut* out);

namespace {

// TransformType -> (bw, bh), jxl-vardct/src/dct_select.rs:52-76
const uint8_t kSize[27][2] = {{1, 1}, {1, 1}, {1, 1}, {1, 1}, {2, 2}, {4, 4}, {1, 2}, {2, 1}, {1, 4},
                              {4, 1}, {2, 4}, {4, 2}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {1, 1},
                              {8, 8}, {4, 8}, {8, 4}, {16, 16}, {8, 16}, {16, 8}, {32, 32}, {16, 32}, {32, 16}};

int class_of(int t) {
    switch (t) {
        case JXLGPU_DCT8: return CLS_DCT8;
        case JXLGPU_HORNUSS: case JXLGPU_DCT2: case JXLGPU_DCT4: case JXLGPU_DCT4X8: case JXLGPU_DCT8X4:
        case JXLGPU_AFV0: case JXLGPU_AFV1: case JXLGPU_AFV2: case JXLGPU_AFV3: return CLS_SPECIAL8;
        case JXLGPU_DCT16: return CLS_16x16;
        case JXLGPU_DCT16X8: return CLS_8x16;
        case JXLGPU_DCT8X16: return CLS_16x8;
        case JXLGPU_DCT32: return CLS_32x32;
        case JXLGPU_DCT32X8: return CLS_8x32;
        case JXLGPU_DCT8X32: return CLS_32x8;
        case JXLGPU_DCT32X16: return CLS_16x32;
        case JXLGPU_DCT16X32: return CLS_32x16;
        case JXLGPU_DCT64: return CLS_64x64;
        case JXLGPU_DCT64X32: return CLS_32x64;
        case JXLGPU_DCT32X64: return CLS_64x32;
        default: return CLS_BIG;
    }
}

// compiler-rt __powisf2: what Rust's f32::powi lowers to (vardct/mod.rs:458-462)
float powi_f32(float a, int b) {
    const bool recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

template <typename T>
int dev_alloc(jxlgpu_ctx* ctx, jxlgpu_frame* f, T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    HIP_TRY(ctx, ctx_dev_malloc(ctx, &p, bytes));
    f->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return JXLGPU_OK;
}

template <typename T>
int dev_upload(jxlgpu_ctx* ctx, jxlgpu_frame* f, T** out, const std::vector<T>& host) {
    int rc = dev_alloc(ctx, f, out, host.size());
    if (rc) return rc;
    if (!host.empty())
        HIP_TRY(ctx, hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    // blocking copy: `host` (pageable) may be released as soon as this returns
    return JXLGPU_OK;
}

// temporaries of the upload: released once the work queued by the upload has finished (never blocks)
struct Scratch {
    jxlgpu_ctx* ctx = nullptr;
    std::vector<void*> ptrs;
    ~Scratch() {
        if (!ptrs.empty()) ctx_defer_release(ctx, std::move(ptrs));
    }
    int alloc(jxlgpu_ctx* c, void** out, size_t bytes) {
        ctx = c;
        HIP_TRY(c, ctx_dev_malloc(c, out, bytes));
        ptrs.push_back(*out);
        return JXLGPU_OK;
    }
};

#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != JXLGPU_OK) return rc_; \
    } while (0)

int fail(jxlgpu_ctx* ctx, int code, const char* msg) {
    ctx->last_error = msg;
    return code;
}

// HF coefficients of channel c into the cell-tiled device layout f->coeff (coeff_kernels.hip).
// Dense planes cross PCIe row-major as the caller holds them (a staging buffer, freed after the
// upload) and are re-laid-out on the device; sparse lists scatter straight into the tiled layout
// (the caller zero-fills f->coeff once before the first channel).
int upload_coeff_plane(jxlgpu_ctx* ctx, jxlgpu_frame* f, const JxlGpuVardctDesc* d, int c, Scratch& tmp,
                       uint32_t* d_bad) {
    const size_t npix = (size_t)f->wr * f->hr;
    const bool v16 = d->coeff_sample_type == JXLGPU_SAMPLE_I16;
    const size_t vsz = v16 ? 2 : 4;
    if (d->coeff_format == JXLGPU_COEFF_DENSE) {
        void* t = nullptr;
        TRY(tmp.alloc(ctx, &t, npix * vsz));
        HIP_TRY(ctx, hipMemcpy2D(t, (size_t)f->wr * vsz, d->coeff[c], (size_t)d->coeff_stride * vsz, (size_t)f->wr * vsz,
                                 f->hr, hipMemcpyHostToDevice));
        launch_coeff_retile(ctx->stream, t, v16, f->wr, f->hr, (uint32_t)c, f->coeff);
        HIP_TRY(ctx, hipGetLastError());
        return JXLGPU_OK;
    }
    const size_t n = (size_t)d->sparse_count[c];
    if (n == 0) return JXLGPU_OK;
    void *dp = nullptr, *dv = nullptr;
    TRY(tmp.alloc(ctx, &dp, n * 4));
    TRY(tmp.alloc(ctx, &dv, n * vsz));
    HIP_TRY(ctx, hipMemcpy(dp, d->sparse_pos[c], n * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dv, d->coeff[c], n * vsz, hipMemcpyHostToDevice));
    launch_coeff_scatter(ctx->stream, static_cast<const uint32_t*>(dp), dv, v16, n, d->coeff_stride, f->wr, f->hr,
                         (uint32_t)c, f->coeff, d_bad);
    HIP_TRY(ctx, hipGetLastError());
    return JXLGPU_OK;
}

void fill_color_args(const JxlGpuColorParams& cp, ColorArgs* c) {
    memset(c, 0, sizeof(*c));
    for (int i = 0; i < 3; ++i) {
        c->opsin_bias[i] = cp.opsin_bias[i];
        c->cbrt_opsin_bias[i] = cbrtf(cp.opsin_bias[i]);  // xyb.rs:42, once per frame on the host
        c->gamut_lum[i] = cp.gamut_luminances[i];
    }
    c->itscale = 255.0f / cp.intensity_target;
    c->intensity_target = cp.intensity_target;
    memcpy(c->matrix, cp.matrix, sizeof(c->matrix));
    memcpy(c->matrix2, cp.matrix2, sizeof(c->matrix2));
    c->gamut_map = cp.gamut_map;
    c->gamut_sat = cp.gamut_saturation_factor;
    c->has_matrix2 = cp.has_matrix2;
    c->tf = cp.transfer_function;
    c->ycbcr = cp.ycbcr;
    c->gamma = cp.gamma;
    c->tone_map = cp.tone_map;
    if (cp.tone_map) {
        // convert/tone_map.rs:19-31 (detect_peak = false) and tf/rec2408.rs:10-29: evaluated once
        // per frame on the host with the same f32 operations (linear_to_pq_dev is host+device).
        const float it = cp.intensity_target;
        const float peak = fminf(it, it);
        float lum[4] = {cp.tm_min_nits / it, peak / it, 0.0f / it, cp.tm_target_display_luminance / it};
        for (float& y : lum) y = linear_to_pq_dev(y, it);
        c->tm_lum0_pq = lum[0];
        c->tm_source_pq_diff = lum[1] - lum[0];
        c->tm_min_luminance = (lum[2] - lum[0]) / c->tm_source_pq_diff;
        c->tm_max_luminance = (lum[3] - lum[0]) / c->tm_source_pq_diff;
        c->tm_ks = 1.5f * c->tm_max_luminance - 0.5f;
        c->tm_one_sub_ks = 1.0f - c->tm_ks;
        c->tm_scale = it / cp.tm_target_display_luminance;
    }
    // the GamutMap behind the tone map; on its own in the PQ -> HLG list of a 1000-nit image (convert.rs:521-528)
    for (int i = 0; i < 3; ++i) c->tm_lum[i] = cp.tm_luminances[i];
    c->tm_gamut_map = cp.tm_gamut_map;
    c->tm_gamut_sat = cp.tm_gamut_saturation_factor;
    // HlgInverseOotf / the inverse OOTF of TransferFunction{Hlg}: tf.rs:118-143.  The system gamma is a frame constant; the
    // reference evaluates it with the platform libm (f32::log2 / powf), and so does this line — the same two libm calls.
    const float hit = cp.hlg_ootf_intensity_target;
    if (hit != 0.0f && !(hit >= 295.0f && hit <= 305.0f)) {
        const float gamma = 1.2f * ::powf(1.111f, ::log2f(hit / 1e3f));
        c->hlg_ootf = 1;
        c->hlg_exp = (1.0f - gamma) / gamma;
        for (int i = 0; i < 3; ++i) c->hlg_lum[i] = cp.hlg_luminances[i];
    }
    c->staged_only = (c->hlg_ootf || c->tf == JXLGPU_TF_HLG || (c->tm_gamut_map && !c->tone_map)) && !c->ycbcr;
}

// upsample_inner's weights_quarter (features/upsampling.rs:77-93)
std::vector<float> expand_up_weights(const float* weights, int k) {
    int mat_n = k / 2;
    std::vector<float> wq((size_t)mat_n * mat_n * 25, 0.0f);
    size_t idx = 0;
    for (int y = 0; y < 5 * mat_n; ++y) {
        int mat_y = y / 5, ky = y % 5;
        for (int x = y; x < 5 * mat_n; ++x) {
            int mat_x = x / 5, kx = x % 5;
            float w = weights[idx++];
            wq[(size_t)(mat_y * mat_n + mat_x) * 25 + ky * 5 + kx] = w;
            wq[(size_t)(mat_x * mat_n + mat_y) * 25 + kx * 5 + ky] = w;
        }
    }
    return wq;
}

}  // namespace

void fill_color_args_public(const JxlGpuColorParams& cp, ColorArgs* c) { fill_color_args(cp, c); }
std::vector<float> expand_up_weights_public(const float* weights, int k) { return expand_up_weights(weights, k); }

// nullptr if the colour op list can run on the device, else why not
const char* color_params_unsupported(const JxlGpuColorParams& cp) {
    if (!cp.enabled || cp.ycbcr) return nullptr;
    if (cp.transfer_function > JXLGPU_TF_HLG) return "unknown transfer function";
    if (!(cp.hlg_ootf_intensity_target >= 0.0f) || std::isinf(cp.hlg_ootf_intensity_target))
        return "hlg_ootf_intensity_target is 0 (no inverse OOTF) or a finite positive intensity target";
    if (cp.gamut_map > JXLGPU_GAMUT_CLIP) return "unknown gamut_map mode";
    // a GamutMap behind the tone map WITHOUT a tone map exists in one op list of the reference only: PQ -> HLG of a 1000-nit image
    // (convert.rs:521-528).  Anything else with tm_gamut_map set and tone_map clear is a stale field of the caller:
    // refuse it instead of applying an extra GamutMap
    if (cp.tm_gamut_map && !cp.tone_map && cp.transfer_function != JXLGPU_TF_HLG)
        return "tm_gamut_map without tone_map is the PQ -> HLG op list only (transfer_function = JXLGPU_TF_HLG)";
    return nullptr;
}

int upload_post_params(jxlgpu_ctx* ctx, jxlgpu_frame* f, const JxlGpuUpsampling& up) {
    const float* src[3] = {up.up2_weight, up.up4_weight, up.up8_weight};
    const int ks[3] = {2, 4, 8};
    for (int i = 0; i < 3; ++i) {
        if (!src[i]) continue;
        std::vector<float> wq = expand_up_weights(src[i], ks[i]);
        TRY(dev_upload(ctx, f, &f->up_weights[i], wq));
        if (i == 0) {
            memcpy(f->up2_wq, wq.data(), sizeof(f->up2_wq));
            f->have_up2 = true;
        }
    }
    return JXLGPU_OK;
}

extern "C" {

uint32_t jxlgpu_abi_version(void) { return JXLGPU_ABI_VERSION; }

// The transform streams of batched renders: JXLGPU_STREAM_PRIO > 0: lowest priority (the post launches on the render stream get
// the wave slots first, the latency-bound transform launches fill what is left); < 0: highest (round 6: the transform chain is
// the one that sets the period of a chunk — profiles/r06_kernel_timeline.txt).
static hipError_t create_tr_stream(jxlgpu_ctx* ctx, hipStream_t* out) {
    const char* v = getenv("JXLGPU_STREAM_PRIO");
    if (v && atoi(v) != 0) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            return hipStreamCreateWithPriority(out, hipStreamNonBlocking, atoi(v) > 0 ? lo : hi);
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

int jxlgpu_create(int device, jxlgpu_ctx** out_ctx) {
    if (!out_ctx) return JXLGPU_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return JXLGPU_ERR_DEVICE;
    jxlgpu_ctx* ctx = new (std::nothrow) jxlgpu_ctx();
    if (!ctx) return JXLGPU_ERR_OOM;
    ctx->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->num_cus = (uint32_t)prop.multiProcessorCount;
    }
#ifdef JXL_TR_PROFILE
    if (hipMalloc(reinterpret_cast<void**>(&ctx->tr_prof), 64 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(ctx->tr_prof, 0, 64 * sizeof(unsigned long long));
    else
        ctx->tr_prof = nullptr;
#endif
    // environment -> per-context tuning, read once here (no process-global state afterwards)
    if (const char* mb = getenv("JXLGPU_POOL_MB")) ctx->pool_cap = (size_t)strtoull(mb, nullptr, 10) << 20;
    if (const char* v = getenv("JXLGPU_STREAM_ROWS")) {
        const int r = atoi(v);
        if (r >= 8 && r <= 1024 && r % 4 == 0) ctx->tune.stream_rows = r;
    }
    if (const char* v = getenv("JXLGPU_BATCH_STREAM_ROWS")) {
        const int r = atoi(v);
        if (r >= 8 && r <= 4096 && r % 4 == 0) ctx->tune.batch_stream_rows = r;
    }
    if (const char* v = getenv("JXLGPU_BATCH_CHUNK")) {
        const int r = atoi(v);
        if (r >= 0 && r <= JXLGPU_MAX_BATCH) ctx->tune.batch_chunk = r;
    }
    ctx->tune.no_pk = getenv("JXLGPU_NO_PK") != nullptr;
    ctx->tune.pk_tb = getenv("JXLGPU_PK_TB") != nullptr && atoi(getenv("JXLGPU_PK_TB")) != 0;
    if (const char* e = getenv("JXLGPU_TR_SIDE_MAX")) ctx->tune.tr_side_max = atoi(e);
    ctx->tune.no_stream = getenv("JXLGPU_NO_STREAM") != nullptr;
    ctx->tune.no_fused = getenv("JXLGPU_NO_FUSED") != nullptr;
    ctx->tune.no_sparse_tr = getenv("JXLGPU_NO_SPARSE_TR") != nullptr;
    ctx->tune.debug_sync = getenv("JXLGPU_DEBUG_SYNC") != nullptr;
    ctx->tune.no_batch_overlap = getenv("JXLGPU_NO_BATCH_OVERLAP") != nullptr;
#ifdef JXL_ENABLE_POST_FAST   // the non-bit-exact post kernel exists in experiment builds only (tools/build_variant.sh ... -DJXL_ENABLE_POST_FAST)
    ctx->tune.post_fast = getenv("JXLGPU_POST_FAST") != nullptr && atoi(getenv("JXLGPU_POST_FAST")) != 0;
#endif
    if (const char* v = getenv("JXLGPU_TR_STREAMS")) ctx->tune.tr_streams = std::min(5, std::max(2, atoi(v)));
    if (const char* v = getenv("JXLGPU_RING_MODE")) ctx->tune.ring_mode = std::min(2, std::max(0, atoi(v)));
    ctx->tune.int_post = getenv("JXLGPU_INT_POST") != nullptr && atoi(getenv("JXLGPU_INT_POST")) != 0;
    if (const char* v = getenv("JXLGPU_BATCH_TR_MULT")) ctx->tune.batch_tr_mult = std::min(4, std::max(1, atoi(v)));
    if (const char* v = getenv("JXLGPU_BATCH_LF_MODE")) ctx->tune.batch_lf_mode = std::min(2, std::max(0, atoi(v)));
    if (const char* v = getenv("JXLGPU_POST_LDS_PAD")) ctx->tune.post_lds_pad = std::min(150 * 1024, std::max(0, atoi(v)));
    if (const char* v = getenv("JXLGPU_BATCH_HEAVY")) ctx->tune.batch_heavy = (uint32_t)strtoul(v, nullptr, 0) & 31u;
    if (const char* v = getenv("JXLGPU_GUARD")) {
        const int m = atoi(v);
        if (m >= 1 && m <= 3) ctx->guard_mode = m;
    }
    if (const char* v = getenv("JXLGPU_GUARD_ZERO")) ctx->guard_zero = strcmp(v, "all") == 0 ? -2 : atoi(v);
    ctx->guard_log = getenv("JXLGPU_GUARD_LOG") != nullptr;
    if (const char* v = getenv("JXLGPU_TR_WGS_PER_CU")) {
        int t[4];
        if (sscanf(v, "%d,%d,%d,%d", &t[0], &t[1], &t[2], &t[3]) == 4)
            for (int i = 0; i < 4; ++i)
                if (t[i] >= 0 && t[i] <= 16) ctx->tune.tr_wgs_per_cu[i] = t[i];
    }
    if (const char* v = getenv("JXLGPU_SQZ_SEG")) {
        const int r = atoi(v);
        if (r >= 8 && r <= 4096) ctx->tune.sqz_seg = r;
    }
    if (const char* v = getenv("JXLGPU_SQZ_RUNIN")) ctx->tune.sqz_runin = (uint32_t)atoi(v);
    ctx->tune.pred_wg = getenv("JXLGPU_PRED_WG") != nullptr;
    if (const char* v = getenv("JXLGPU_PRED_PRIO")) ctx->tune.pred_prio = atoi(v) != 0;
    ctx->tune.pred_wide = getenv("JXLGPU_PRED_WIDE") != nullptr;
    if (const char* v = getenv("JXLGPU_PRED_STEP_V1")) ctx->tune.pred_step_v1 = atoi(v) != 0;
    if (const char* v = getenv("JXLGPU_PRED_LATE_STEPS")) ctx->tune.pred_late_steps = std::max(0, atoi(v));
    ctx->tune.sqz_h_rows = getenv("JXLGPU_SQZ_H_ROWS") != nullptr;
    if (const char* v = getenv("JXLGPU_UP2_VARIANT")) ctx->tune.up2_variant = atoi(v);
    if (const char* v = getenv("JXLGPU_UP2_ROWS")) ctx->tune.up2_rows = atoi(v);
    if (const char* v = getenv("JXLGPU_HOST_THREADS")) ctx->host_threads = std::max(0, atoi(v));
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        create_tr_stream(ctx, &ctx->stream_tr) != hipSuccess ||
        create_tr_stream(ctx, &ctx->stream_tr2) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_up, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_down, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return JXLGPU_ERR_DEVICE;
    }
    *out_ctx = ctx;
    return JXLGPU_OK;
}

void jxlgpu_destroy(jxlgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
#ifdef JXL_TR_PROFILE
    if (ctx->tr_prof) {
        unsigned long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, ctx->tr_prof, sizeof(h), hipMemcpyDeviceToHost);
        static const char* kPh[11] = {"n", "entries", "coef+lut", "barrier0", "zero", "scatter", "llf+dequant", "ydq+barrier1", "cfl+rows", "cols+stores", "drain"};
        for (int fam = 0; fam < 4; ++fam) {
            if (!h[fam * 16]) continue;
            fprintf(stderr, "[tr_prof] family %d: %llu wave-items;", fam, h[fam * 16]);
            for (int i = 1; i < 11; ++i) fprintf(stderr, " %s %.0f", kPh[i], (double)h[fam * 16 + i] / (double)h[fam * 16]);
            fprintf(stderr, " (cycles / wave-item)\n");
        }
        (void)hipFree(ctx->tr_prof);
    }
#endif
    for (hipStream_t st : {ctx->stream_tr2, ctx->stream_tr, ctx->stream, ctx->stream2, ctx->stream_up, ctx->stream_down})
        if (st) (void)hipStreamSynchronize(st);
    ctx_reap(ctx, true);
    delete ctx->workers;
    ctx->workers = nullptr;
    for (StageBuf& sb : ctx->stage) {
        if (sb.p) (void)hipHostFree(sb.p);
        if (sb.ev) (void)hipEventDestroy(sb.ev);
    }
    for (StageBuf& sb : ctx->rt_stage) {
        if (sb.p) (void)hipHostFree(sb.p);
        if (sb.ev) (void)hipEventDestroy(sb.ev);
    }
    for (hipEvent_t e : ctx->ev_spare) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_h2d) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream_tr2) (void)hipStreamDestroy(ctx->stream_tr2);
    for (hipStream_t q : ctx->stream_tr_extra) if (q) (void)hipStreamDestroy(q);
    for (hipEvent_t e : ctx->ev_tr_join) if (e) (void)hipEventDestroy(e);
    if (ctx->stream_tr) (void)hipStreamDestroy(ctx->stream_tr);
    for (hipEvent_t e : ctx->ev_tr) if (e) (void)hipEventDestroy(e);
    if (ctx->stream_up) (void)hipStreamDestroy(ctx->stream_up);
    if (ctx->stream_down) (void)hipStreamDestroy(ctx->stream_down);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (hipEvent_t e : ctx->ev_d2h) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_slice) if (e) (void)hipEventDestroy(e);
    if (ctx->noise_jump) ctx_dev_release(ctx, ctx->noise_jump);
    for (auto& kv : ctx->pool) (void)hipFree(kv.second);
    while (!ctx->guard_live.empty()) guard_free(ctx, ctx->guard_live.begin()->first);
    for (auto& e : ctx->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete ctx;
}

const char* jxlgpu_last_error(const jxlgpu_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null ctx"; }

int jxlgpu_synchronize(jxlgpu_ctx* ctx) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_up));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_tr2));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_tr));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_down));
    ctx_reap(ctx, false);
    return JXLGPU_OK;
}

int jxlgpu_frame_wait(jxlgpu_ctx* ctx, jxlgpu_frame* f) {
    if (!ctx || !f) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (f->ev_last && f->ev_last_set) {
        HIP_TRY(ctx, hipEventSynchronize(f->ev_last));
        return JXLGPU_OK;
    }
    return jxlgpu_synchronize(ctx);
}

int jxlgpu_host_alloc(jxlgpu_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out || !bytes) return JXLGPU_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return JXLGPU_OK;
}

void jxlgpu_host_free(jxlgpu_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipHostFree(p);
}

int jxlgpu_set_memory_limit(jxlgpu_ctx* ctx, uint64_t limit_bytes) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    ctx->mem_limit = (size_t)limit_bytes;
    return JXLGPU_OK;
}

int jxlgpu_memory_usage(const jxlgpu_ctx* ctx, uint64_t* live_bytes, uint64_t* pooled_bytes) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    if (live_bytes) *live_bytes = ctx->live_bytes;
    if (pooled_bytes) *pooled_bytes = ctx->pool_bytes;
    return JXLGPU_OK;
}

int jxlgpu_upload_split(jxlgpu_ctx* ctx, double ms[5]) {
    if (!ctx || !ms) return JXLGPU_ERR_INVALID_ARG;
    for (int i = 0; i < 4; ++i) ms[i] = ctx->up_split[i];
    ms[4] = 0.0;
    if (ctx->h2d_timed && ctx->ev_h2d[0] && ctx->ev_h2d[1]) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_h2d[1]));
        float t = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev_h2d[0], ctx->ev_h2d[1]));
        ms[4] = t;
    }
    return JXLGPU_OK;
}

void* jxlgpu_stream(jxlgpu_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int jxlgpu_set_trace(jxlgpu_ctx* ctx, jxlgpu_trace_fn fn, void* user) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    ctx->trace_fn = fn;
    ctx->trace_user = user;
    return JXLGPU_OK;
}

int jxlgpu_profile_select(jxlgpu_ctx* ctx, int group) {
    if (!ctx || group >= PROF_COUNT) return JXLGPU_ERR_INVALID_ARG;
    ctx->prof_group = group;
    ctx->prof_used = 0;
    return JXLGPU_OK;
}

int jxlgpu_profile_read(jxlgpu_ctx* ctx, double* total_ms, uint64_t* brackets) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0;
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i].first, ctx->prof_events[i].second));
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (brackets) *brackets = ctx->prof_used;
    ctx->prof_used = 0;
    return JXLGPU_OK;
}

// Never blocks: the frame's device buffers go back to the pool once everything that was queued on the ctx's
// streams at this moment has finished (ctx_defer_release); the handle is dead when the call returns.
static void frame_collect(jxlgpu_frame* f, std::vector<void*>* ptrs, std::vector<std::pair<void*, void (*)(void*)>>* mods) {
    for (auto& sub : f->subs)
        if (sub.child) frame_collect(sub.child, ptrs, mods);
    ptrs->insert(ptrs->end(), f->allocs.begin(), f->allocs.end());
    if (f->modular && f->modular_free) mods->emplace_back(f->modular, f->modular_free);
    if (f->ev_last) (void)hipEventDestroy(f->ev_last);
    if (f->ev_fmt) (void)hipEventDestroy(f->ev_fmt);
    delete f;
}

void jxlgpu_frame_free(jxlgpu_ctx* ctx, jxlgpu_frame* f) {
    if (!f) return;
    std::vector<void*> ptrs;
    std::vector<std::pair<void*, void (*)(void*)>> mods;
    if (!ctx) {
        (void)hipDeviceSynchronize();
        frame_collect(f, &ptrs, &mods);
        for (void* p : ptrs) (void)hipFree(p);
        for (auto& m : mods) m.second(m.first);
        return;
    }
    (void)hipSetDevice(ctx->device);
    frame_collect(f, &ptrs, &mods);
    if (mods.empty()) {
        ctx_defer_release(ctx, std::move(ptrs));
    } else {
        ctx_defer_release(ctx, std::move(ptrs), mods[0].first, mods[0].second);
        for (size_t i = 1; i < mods.size(); ++i) ctx_defer_release(ctx, {}, mods[i].first, mods[i].second);
    }
}

int jxlgpu_vardct_upload(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* d, jxlgpu_frame** out_frame) {
    if (!ctx || !d || !out_frame) return JXLGPU_ERR_INVALID_ARG;
    *out_frame = nullptr;
    if (d->abi != JXLGPU_ABI_VERSION) return fail(ctx, JXLGPU_ERR_ABI, "descriptor ABI version mismatch");
    if (d->jpeg_upsampling[0] | d->jpeg_upsampling[1] | d->jpeg_upsampling[2]) return upload_subsampled(ctx, d, out_frame);
    return vardct_upload_impl(ctx, d, UploadOpts{}, out_frame);
}

}  // extern "C" (reopened below)

// One frame geometry: every channel has the same size.  `o` lets upload_subsampled build the
// per-geometry children of a chroma-subsampled frame from derived descriptors.
//
// Host side of an upload, per frame (the reference's equivalent is the serial `for_each_varblocks` scan,
// jxl-render/src/vardct/mod.rs:693-730, once per group inside the rayon loop):
//   phase 1 (worker threads): the LF groups' grids are copied into frame-level planes; every pass group's
//            varblocks are validated and counted per work-list slot; the groups' non-zero lists are copied;
//   prefix : where each (group, slot) run starts in the entry array;
//   phase 2 (worker threads): the entries are written.
// Everything lands in ONE pinned staging arena, crosses PCIe as ONE hipMemcpyAsync on the ctx's upload stream
// into ONE device allocation, and the render stream waits for that copy by event: the call returns without
// waiting for the device (except for the dense / sparse transports, whose caller-owned planes are copied
// with blocking hipMemcpy's, and the LF-frame planes).
namespace {

// Work-list slots: the shape classes in launch order, the special 8x8 family split by transform type (its kernel
// wants runs of one type).  Slot order == order of the entry array.
constexpr int kNumSlots = CLS_COUNT + 8;
int slot_of(int t) {
    switch (t) {
        case JXLGPU_DCT8: return 0;
        case JXLGPU_HORNUSS: return 1;
        case JXLGPU_DCT2: return 2;
        case JXLGPU_DCT4: return 3;
        case JXLGPU_DCT4X8: return 4;
        case JXLGPU_DCT8X4: return 5;
        case JXLGPU_AFV0: return 6;
        case JXLGPU_AFV1: return 7;
        case JXLGPU_AFV2: return 8;
        case JXLGPU_AFV3: return 9;
        default: return class_of(t) + 8;  // CLS_16x16 (2) -> 10 ... CLS_BIG (13) -> 21
    }
}
int class_of_slot(int s) { return s == 0 ? CLS_DCT8 : (s <= 9 ? CLS_SPECIAL8 : s - 8); }

struct ArenaItem { size_t off, bytes; void** dev; };
struct Arena {
    std::vector<ArenaItem> items;
    size_t total = 0;
    // reserves `bytes` (256-byte aligned); *dev receives the device address once the arena is placed
    size_t add(size_t bytes, void** dev) {
        const size_t off = total;
        items.push_back({off, bytes, dev});
        total += (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
        return off;
    }
};

size_t pool_bucket(size_t bytes) {  // sizes that differ by a few per cent share a pool bucket
    if (bytes <= 4096) return 4096;
    size_t step = (size_t)1 << 12;
    while (step * 16 < bytes) step <<= 1;
    return (bytes + step - 1) / step * step;
}

}  // namespace

int vardct_upload_impl(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* d, const UploadOpts& o, jxlgpu_frame** out_frame) {
    *out_frame = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    if (d->jpeg_upsampling[0] | d->jpeg_upsampling[1] | d->jpeg_upsampling[2])
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "internal: subsampled descriptor in the single-geometry upload");
    if (d->group_dim != 256) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "only group_dim == 256 is supported");
    if (const char* why = color_params_unsupported(d->color)) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, why);
    if (d->width == 0 || d->height == 0 || d->width > (1u << 18) || d->height > (1u << 18))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "bad frame size");
    if (d->global_scale == 0 || d->quant_lf == 0 || d->colour_factor == 0)
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "zero quantizer / colour_factor");
    if (d->filter.epf_iters > 3) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "epf_iters > 3");
    const uint32_t upf = d->upsampling.factor ? d->upsampling.factor : 1;
    if (upf != 1 && upf != 2 && upf != 4 && upf != 8) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "bad upsampling factor");
    // several per-row kernels launch one grid row per image row (HIP: grid.y <= 65535)
    if ((uint64_t)d->height * upf > 65535u) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "output taller than 65535 rows");
    if (d->coeff_format > JXLGPU_COEFF_GROUPED ||
        (d->coeff_format != JXLGPU_COEFF_GROUPED && d->coeff_sample_type > JXLGPU_SAMPLE_I16))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "bad coeff_format / coeff_sample_type");
    const bool grouped = d->coeff_format == JXLGPU_COEFF_GROUPED;
    for (int c = 0; c < 3; ++c) {
        if (d->coeff_format == JXLGPU_COEFF_DENSE && !d->coeff[c])
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null coefficient plane");
        if (d->coeff_format == JXLGPU_COEFF_SPARSE && d->sparse_count[c] && (!d->coeff[c] || !d->sparse_pos[c]))
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null sparse coefficient list");
    }

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx_reap(ctx, false);
    jxlgpu_frame* f = new (std::nothrow) jxlgpu_frame();
    if (!f) return JXLGPU_ERR_OOM;
    struct Guard {
        jxlgpu_ctx* c; jxlgpu_frame* f; bool armed = true;
        ~Guard() { if (armed) jxlgpu_frame_free(c, f); }
    } guard{ctx, f};

    f->desc = *d;
    f->width = d->width; f->height = d->height;
    f->w8 = ceil_div(d->width, 8); f->h8 = ceil_div(d->height, 8);
    f->wr = f->w8 * 8; f->hr = f->h8 * 8;
    f->w64 = ceil_div(d->width, 64); f->h64 = ceil_div(d->height, 64);
    f->group_dim = d->group_dim;
    // LF group size in cells; children of a subsampled frame keep the parent's LF-group tiling
    f->lfg_cells_x = o.lfg_cells_x ? o.lfg_cells_x : d->group_dim;
    f->lfg_cells_y = o.lfg_cells_y ? o.lfg_cells_y : d->group_dim;
    const uint32_t lf_px_x = f->lfg_cells_x * 8, lf_px_y = f->lfg_cells_y * 8;
    f->lf_groups_per_row = ceil_div(d->width, lf_px_x);
    const uint32_t lf_rows = ceil_div(d->height, lf_px_y);
    f->num_lf_groups = f->lf_groups_per_row * lf_rows;
    if (d->num_lf_groups != f->num_lf_groups || !d->lf_groups)
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "num_lf_groups does not match the frame size");
    if (!grouped && d->coeff_stride < f->wr) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "coeff_stride < width_rounded");
    if ((uint64_t)f->wr * f->hr * 3 >= (1ull << 30))  // kernels address the tiled planes with 32-bit word offsets
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "frame larger than 357 megapixels");

    const bool use_lf_frame = d->lf_frame[0] || d->lf_frame[1] || d->lf_frame[2];
    if (use_lf_frame && (!d->lf_frame[0] || !d->lf_frame[1] || !d->lf_frame[2] || d->lf_frame_stride < f->w8))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "lf_frame needs three planes with lf_frame_stride >= ceil(width / 8)");
    f->lf_from_frame = use_lf_frame;
    const size_t ncell = (size_t)f->w8 * f->h8, ntile = (size_t)f->w64 * f->h64;
    const size_t npix = (size_t)f->wr * f->hr;
    const bool i16 = d->lf_sample_type == JXLGPU_SAMPLE_I16;
    f->lf_is_i16 = i16;
    const size_t lf_elem = i16 ? 2 : 4;
    const uint32_t w8 = f->w8, w64 = f->w64;

    // ---- serial checks of the LF groups (cheap: a handful per frame); everything per cell runs on the workers
    const uint64_t scale_inv = (uint64_t)d->global_scale * (uint64_t)d->quant_lf;
    std::vector<uint8_t> has_meta(f->num_lf_groups, 0);
    for (uint32_t g = 0; g < f->num_lf_groups; ++g) {
        const JxlGpuLfGroup& lg = d->lf_groups[g];
        const uint32_t gx = g % f->lf_groups_per_row, gy = g / f->lf_groups_per_row;
        const uint32_t exp_w = std::min(lf_px_x, d->width - gx * lf_px_x), exp_h = std::min(lf_px_y, d->height - gy * lf_px_y);
        if (lg.width_px != exp_w || lg.height_px != exp_h)
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "LF group size does not match the frame geometry");
        if (lg.extra_precision > 3) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra_precision > 3");
        if (!use_lf_frame && (!lg.lf_quant[0] || !lg.lf_quant[1] || !lg.lf_quant[2]))
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null lf_quant");
        if (!lg.has_hf_meta) continue;
        if (!lg.block_kind || !lg.hf_mul || !lg.x_from_y || !lg.b_from_y)
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "HfMetadata pointers missing");
        has_meta[g] = 1;
    }
    const uint32_t gcells = d->group_dim / 8;
    const uint32_t groups_x = ceil_div(d->width, d->group_dim), groups_y = ceil_div(d->height, d->group_dim);
    const uint32_t n_groups = groups_x * groups_y;
    if (grouped && (d->num_hf_groups != n_groups || !d->hf_groups))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "num_hf_groups does not match the frame size");
    // progressive frames: one list set per pass, pass-major (hf_groups[p * n_groups + g])
    const uint32_t n_passes = grouped ? std::max(1u, d->num_passes) : 1u;
    if (n_passes > 11) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "more than 11 passes");  // jxl-frame/src/header.rs Passes: num_passes <= 11
    // (allow_partial with several passes: every (pass, group) list may stop before its block map does, each on its own —
    //  a truncated progressive stream, vardct/mod.rs:275-305; scan_group treats the missing varblocks of a list as empty)
    const uint32_t n_lists = n_groups * n_passes;
    std::vector<uint64_t> nz_base((size_t)n_lists + 1, 0);  // list words in front of each (pass, group)
    if (grouped) {
        for (uint32_t g = 0; g < n_lists; ++g) {
            const JxlGpuHfGroup& hg = d->hf_groups[g];
            if ((hg.num_varblocks && !hg.nz_count) || (hg.num_nz && !hg.nz))
                return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null list pointers in an HF group");
            nz_base[g + 1] = nz_base[g] + hg.num_nz;
        }
        if (nz_base[n_lists] >= (1ull << 32)) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "more than 2^32 non-zero coefficients");
    }
    const uint64_t nz_total = nz_base[n_lists];

    // ---- the arena: fixed-size items first, the lists (upper bounds) behind them
    Arena ar;
    void* lfq_dev[3] = {};
    const size_t off_kind = ar.add(ncell, reinterpret_cast<void**>(&f->kind));
    const size_t off_mul = ar.add(ncell * 4, reinterpret_cast<void**>(&f->hf_mul));
    const size_t off_sigma = ar.add(ncell * 4, reinterpret_cast<void**>(&f->sigma));
    const size_t off_kx = ar.add(ntile * 4, reinterpret_cast<void**>(&f->kx_map));
    const size_t off_kb = ar.add(ntile * 4, reinterpret_cast<void**>(&f->kb_map));
    const size_t off_scale = ar.add((size_t)f->num_lf_groups * 12, reinterpret_cast<void**>(&f->lf_scale));
    size_t off_lfq[3];
    for (int c = 0; c < 3; ++c) off_lfq[c] = ar.add(ncell * lf_elem, &lfq_dev[c]);
    const size_t off_deq_off = ar.add(27 * 3 * 4, reinterpret_cast<void**>(&f->deq_off));
    const size_t off_lut = ar.add(256 * 4, reinterpret_cast<void**>(&f->deq_lut));
    size_t off_sec[3];
    for (int i = 0, n = 64; i < 3; ++i, n *= 2) off_sec[i] = ar.add((size_t)(n / 2) * 4, reinterpret_cast<void**>(&f->sec[i]));
    std::vector<float> upw[3];
    size_t off_upw[3] = {};
    if (upf > 1) {
        const float* src[3] = {d->upsampling.up2_weight, d->upsampling.up4_weight, d->upsampling.up8_weight};
        const int ks[3] = {2, 4, 8};
        for (int i = 0; i < 3; ++i) {
            if (!src[i]) continue;
            upw[i] = expand_up_weights(src[i], ks[i]);
            off_upw[i] = ar.add(upw[i].size() * 4, reinterpret_cast<void**>(&f->up_weights[i]));
            if (i == 0) {
                memcpy(f->up2_wq, upw[i].data(), sizeof(f->up2_wq));
                f->have_up2 = true;
            }
        }
        const int need = upf == 2 ? 0 : upf == 4 ? 1 : 2;
        if (upw[need].empty()) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "upsampling weights missing");
    }
    const size_t off_nz = grouped ? ar.add((size_t)nz_total * 4, reinterpret_cast<void**>(&f->nz)) : 0;
    // upper bounds: one entry per cell, every dequantisation matrix, every group without HfMetadata
    size_t deq_max = 0;
    for (int t = 0; t < 27; ++t) deq_max += (size_t)kSize[t][0] * kSize[t][1] * 64 * 3;
    const size_t off_entries = ar.add(ncell * sizeof(uint4), reinterpret_cast<void**>(&f->entries));
    const size_t off_nzc = grouped ? ar.add(ncell * 4, reinterpret_cast<void**>(&f->nzc)) : 0;
    // passes 1 .. n_passes - 1: their own entry / count arrays (same varblocks, other list offsets and counts)
    std::vector<uint4*> pass_entries_dev(n_passes, nullptr);
    std::vector<uint32_t*> pass_nzc_dev(n_passes, nullptr);
    std::vector<size_t> off_pass_entries(n_passes, 0), off_pass_nzc(n_passes, 0);
    for (uint32_t p = 1; p < n_passes; ++p) {
        off_pass_entries[p] = ar.add(ncell * sizeof(uint4), reinterpret_cast<void**>(&pass_entries_dev[p]));
        off_pass_nzc[p] = ar.add(ncell * 4, reinterpret_cast<void**>(&pass_nzc_dev[p]));
    }
    const size_t off_nometa = ar.add((size_t)n_groups * 4, reinterpret_cast<void**>(&f->nometa_groups));
    const size_t off_deq = ar.add(deq_max * 4, reinterpret_cast<void**>(&f->dequant));

    // ---- staging buffer (pinned; reused round-robin, guarded by the event behind its last copy)
    StageBuf& sb = ctx->stage[ctx->stage_next++ % 3];
    if (sb.busy) {
        HIP_TRY(ctx, hipEventSynchronize(sb.ev));
        sb.busy = false;
    }
    if (sb.cap < ar.total) {
        if (sb.p) (void)hipHostFree(sb.p);
        sb.p = nullptr; sb.cap = 0;
        const size_t cap = pool_bucket(ar.total + ar.total / 8);
        HIP_TRY(ctx, hipHostMalloc(&sb.p, cap, hipHostMallocDefault));
        sb.cap = cap;
    }
    if (!sb.ev) HIP_TRY(ctx, hipEventCreateWithFlags(&sb.ev, hipEventDisableTiming));
    char* const H = static_cast<char*>(sb.p);
    uint8_t* const kind = reinterpret_cast<uint8_t*>(H + off_kind);
    int32_t* const hf_mul = reinterpret_cast<int32_t*>(H + off_mul);
    float* const sigma = reinterpret_cast<float*>(H + off_sigma);
    float* const kx_map = reinterpret_cast<float*>(H + off_kx);
    float* const kb_map = reinterpret_cast<float*>(H + off_kb);
    float* const lf_scale = reinterpret_cast<float*>(H + off_scale);
    uint4* const entries = reinterpret_cast<uint4*>(H + off_entries);
    uint32_t* const nzc = grouped ? reinterpret_cast<uint32_t*>(H + off_nzc) : nullptr;
    uint32_t* const nzw = grouped ? reinterpret_cast<uint32_t*>(H + off_nz) : nullptr;

    // ---- phase 1
    struct GroupScan {
        uint32_t count[kNumSlots];
        uint32_t used_mask;      // transform types present (bit t)
        uint32_t nometa;
        int err;
        const char* msg;
    };
    std::vector<GroupScan> gs(n_groups);
    // one pass group's varblocks in decode order (raster inside the group): `emit(slot, entry, cyx)`; returns an error text or null
    auto scan_group = [&](uint32_t g, uint32_t pass, auto&& emit, GroupScan* st) -> const char* {
        const uint32_t gx = g % groups_x, gy = g / groups_x;
        const uint32_t lfx = gx * gcells / f->lfg_cells_x, lfy = gy * gcells / f->lfg_cells_y;
        const uint32_t lfg = lfy * f->lf_groups_per_row + lfx;
        const JxlGpuHfGroup* hg = grouped ? &d->hf_groups[(size_t)pass * n_groups + g] : nullptr;
        if (!has_meta[lfg]) {
            if (hg && (hg->num_varblocks || hg->num_nz)) return "HF lists for a group without HfMetadata";
            if (st) st->nometa = 1;
            return nullptr;
        }
        const JxlGpuLfGroup& lg = d->lf_groups[lfg];
        const uint32_t lbw = ceil_div(lg.width_px, 8);                      // row stride of the LF group's grids
        const uint32_t cx0 = lfx * f->lfg_cells_x, cy0 = lfy * f->lfg_cells_y;  // the LF group's first cell
        const uint32_t x1 = std::min(w8, (gx + 1) * gcells), y1 = std::min(f->h8, (gy + 1) * gcells);
        uint32_t vb_k = 0;   // varblocks of this group seen so far (decode order)
        uint64_t nz_k = 0;   // their list words
        static const uint16_t kNone[3] = {0, 0, 0};
        for (uint32_t y = gy * gcells; y < y1; ++y) {
            const uint8_t* krow = lg.block_kind + (size_t)(y - cy0) * lbw;   // in-bounds row pointers, indexed with x - cx0
            const int32_t* mrow = lg.hf_mul + (size_t)(y - cy0) * lbw;
            for (uint32_t x = gx * gcells; x < x1; ++x) {
                const uint8_t t = krow[x - cx0];
                if (t > 26) continue;
                const uint32_t bw = kSize[t][0], bh = kSize[t][1];
                // hf_metadata.rs:144-158: a varblock never crosses a group; keep the device safe
                if (x + bw > x1 || y + bh > y1) return "varblock crosses a group border";
                const int32_t mul = mrow[x - cx0];
                if (mul <= 0) return "non-positive HfMul";
                if (!d->dequant[t][0] || !d->dequant[t][1] || !d->dequant[t][2]) return "missing dequant matrix for a used transform";
                uint4 e = make_uint4(x | (y << 16), t, (uint32_t)mul, 0);
                uint32_t cyx = 0;
                if (hg) {
                    if (vb_k >= hg->num_varblocks && !d->allow_partial) return "fewer nz_count entries than varblocks in a group";
                    // allow_partial: the group's decode stopped before this varblock — no HF coefficients
                    const uint16_t* cnt = vb_k < hg->num_varblocks ? hg->nz_count + 3 * (size_t)vb_k : kNone;  // decode order: Y, X, B
                    const uint32_t max_nz = 63u * bw * bh;                 // hf_coeff.rs:193
                    if (cnt[0] > max_nz || cnt[1] > max_nz || cnt[2] > max_nz) return "non_zeros too large";
                    e.w = (uint32_t)(nz_base[(size_t)pass * n_groups + g] + nz_k);
                    e.y |= (uint32_t)cnt[2] << 16;
                    cyx = (uint32_t)cnt[0] | (uint32_t)cnt[1] << 16;
                    nz_k += (uint64_t)cnt[0] + cnt[1] + cnt[2];
                    ++vb_k;
                }
                emit(slot_of(t), e, cyx);
                if (st) st->used_mask |= 1u << t;
            }
        }
        if (hg && ((vb_k != hg->num_varblocks && !(d->allow_partial && vb_k > hg->num_varblocks)) || nz_k != hg->num_nz))
            return "HF group lists do not match the block map";
        return nullptr;
    };

    // tasks: [0, n_rows) plane assembly by cell-row chunk; then the groups' scans; then the list copies
    const uint32_t row_chunk = 32, n_row_tasks = ceil_div(f->h8, row_chunk);
    const uint32_t nz_chunks = grouped ? std::max<uint32_t>(1, std::min<uint32_t>(16, (uint32_t)(nz_total >> 16))) : 0;
    const auto t_phase1 = std::chrono::steady_clock::now();
    std::vector<const char*> pass_err((size_t)n_lists, nullptr);   // passes >= 1: validated in phase 1, written in phase 2
    ctx_host_parallel(ctx, n_row_tasks + n_lists + nz_chunks, [&](uint32_t task) {
        if (task < n_row_tasks) {
            // frame-level side planes from the per-LF-group grids, cell rows [r0, r1)
            const uint32_t r0 = task * row_chunk, r1 = std::min(f->h8, r0 + row_chunk);
            static const int SRC[3] = {1, 0, 2};  // util.rs:275-298: lf_x <- channel 1, lf_y <- channel 0, lf_b <- channel 2
            for (uint32_t y = r0; y < r1; ++y) {
                const uint32_t lfy = y / f->lfg_cells_y, ly = y - lfy * f->lfg_cells_y;
                for (uint32_t lfx = 0; lfx < f->lf_groups_per_row; ++lfx) {
                    const JxlGpuLfGroup& lg = d->lf_groups[lfy * f->lf_groups_per_row + lfx];
                    const uint32_t bw = ceil_div(lg.width_px, 8), cx0 = lfx * f->lfg_cells_x;
                    const size_t dst = (size_t)y * w8 + cx0;
                    for (int c = 0; c < 3; ++c) {
                        char* q = H + off_lfq[c] + dst * lf_elem;
                        if (use_lf_frame) memset(q, 0, bw * lf_elem);
                        else memcpy(q, static_cast<const char*>(lg.lf_quant[SRC[c]]) + (size_t)ly * bw * lf_elem, bw * lf_elem);
                    }
                    if (lg.has_hf_meta) {
                        memcpy(kind + dst, lg.block_kind + (size_t)ly * bw, bw);
                        memcpy(hf_mul + dst, lg.hf_mul + (size_t)ly * bw, bw * 4);
                        if (lg.epf_sigma) memcpy(sigma + dst, lg.epf_sigma + (size_t)ly * bw, bw * 4);
                        else for (uint32_t x = 0; x < bw; ++x) sigma[dst + x] = d->filter.epf_sigma_for_modular;
                    } else {
                        memset(kind + dst, JXLGPU_BLOCK_UNINIT, bw);
                        memset(hf_mul + dst, 0, bw * 4);
                        for (uint32_t x = 0; x < bw; ++x) sigma[dst + x] = d->filter.epf_sigma_for_modular;
                    }
                }
            }
            // chroma-from-luma maps (64 x 64 tiles): tile rows [r0 / 8, ...) of this chunk
            for (uint32_t ty = r0 / 8; ty < std::min(f->h64, ceil_div(r1, 8u)); ++ty) {
                if (ty * 8 < r0) continue;  // (chunks are multiples of 8 cell rows: never true)
                const uint32_t lfy = ty * 8 / f->lfg_cells_y, lty = ty - lfy * (f->lfg_cells_y / 8);
                for (uint32_t lfx = 0; lfx < f->lf_groups_per_row; ++lfx) {
                    const JxlGpuLfGroup& lg = d->lf_groups[lfy * f->lf_groups_per_row + lfx];
                    const uint32_t cw = ceil_div(lg.width_px, 64), tx0 = lfx * (f->lfg_cells_x / 8);
                    for (uint32_t x = 0; x < cw; ++x) {
                        float kx = 0.0f, kb = 0.0f;
                        if (lg.has_hf_meta && !o.no_cfl) {  // vardct/mod.rs:355: no chroma-from-luma on subsampled frames
                            // chroma_from_luma_hf_grouped, vardct/mod.rs:590-593
                            kx = d->base_correlation_x + ((float)lg.x_from_y[(size_t)lty * cw + x] / (float)d->colour_factor);
                            kb = d->base_correlation_b + ((float)lg.b_from_y[(size_t)lty * cw + x] / (float)d->colour_factor);
                        }
                        kx_map[(size_t)ty * w64 + tx0 + x] = kx;
                        kb_map[(size_t)ty * w64 + tx0 + x] = kb;
                    }
                }
            }
        } else if (task < n_row_tasks + n_groups) {
            const uint32_t g = task - n_row_tasks;
            GroupScan& st = gs[g];
            memset(&st, 0, sizeof(st));
            st.msg = scan_group(g, 0, [&](int slot, const uint4&, uint32_t) { ++st.count[slot]; }, &st);
            st.err = st.msg ? JXLGPU_ERR_INVALID_ARG : JXLGPU_OK;
        } else if (task < n_row_tasks + n_lists) {
            const uint32_t l = task - n_row_tasks;   // a later pass of a group: validation only
            pass_err[l] = scan_group(l % n_groups, l / n_groups, [](int, const uint4&, uint32_t) {}, nullptr);
        } else {
            // the groups' non-zero lists, concatenated: an even share of the words per task
            const uint32_t k = task - n_row_tasks - n_lists;
            const uint64_t w0 = nz_total * k / nz_chunks, w1 = nz_total * (k + 1) / nz_chunks;
            uint32_t g = (uint32_t)(std::upper_bound(nz_base.begin(), nz_base.end(), w0) - nz_base.begin()) - 1;
            for (uint64_t w = w0; w < w1 && g < n_lists; ++g) {
                const uint64_t ge = std::min<uint64_t>(nz_base[g + 1], w1);
                if (ge > w) memcpy(nzw + w, d->hf_groups[g].nz + (w - nz_base[g]), (size_t)(ge - w) * 4);
                w = std::max(w, ge);
            }
        }
    });
    for (uint32_t g = 0; g < n_groups; ++g)
        if (gs[g].err) return fail(ctx, gs[g].err, gs[g].msg);
    for (uint32_t l = n_groups; l < n_lists; ++l)
        if (pass_err[l]) return fail(ctx, JXLGPU_ERR_INVALID_ARG, pass_err[l]);
    // copy_lf_dequant scale (vardct/mod.rs:398-400), f64 on the host exactly as the reference
    for (uint32_t g = 0; g < f->num_lf_groups; ++g) {
        const int32_t precision_scale = 1 << (9 - d->lf_groups[g].extra_precision);
        for (int c = 0; c < 3; ++c)
            lf_scale[(size_t)g * 3 + c] = (float)((double)d->m_lf[c] * (double)precision_scale / (double)scale_inv);
    }

    // ---- prefix: where every (slot, group) run starts; the class tables
    std::vector<uint32_t> start((size_t)n_groups * kNumSlots);
    uint32_t used_mask = 0, n_entries = 0, n_nometa = 0;
    memset(f->list_count, 0, sizeof(f->list_count));
    for (int cls = 0; cls < CLS_COUNT; ++cls) f->class_first[cls] = 0xffffffffu;
    for (int s = 0; s < kNumSlots; ++s) {
        const int cls = class_of_slot(s);
        if (f->class_first[cls] == 0xffffffffu) f->class_first[cls] = n_entries;
        for (uint32_t g = 0; g < n_groups; ++g) {
            start[(size_t)g * kNumSlots + s] = n_entries;
            n_entries += gs[g].count[s];
            f->list_count[cls] += gs[g].count[s];
        }
    }
    uint32_t* const nometa = reinterpret_cast<uint32_t*>(H + off_nometa);
    for (uint32_t g = 0; g < n_groups; ++g) {
        used_mask |= gs[g].used_mask;
        if (gs[g].nometa) nometa[n_nometa++] = g;
    }
    f->nometa_count = n_nometa;

    // ---- phase 2: the entries (+ the dequantisation matrices of the types that appear, as one more task)
    uint32_t* const deq_off = reinterpret_cast<uint32_t*>(H + off_deq_off);
    float* const deq = reinterpret_cast<float*>(H + off_deq);
    size_t deq_words = 0;
    bool zero_stays_zero = true;
    ctx_host_parallel(ctx, n_lists + 1, [&](uint32_t task) {
        if (task < n_lists) {
            const uint32_t g = task % n_groups, pass = task / n_groups;
            uint4* const ent = pass == 0 ? entries : reinterpret_cast<uint4*>(H + off_pass_entries[pass]);
            uint32_t* const cnt = pass == 0 ? nzc : reinterpret_cast<uint32_t*>(H + off_pass_nzc[pass]);
            uint32_t pos[kNumSlots];
            for (int s = 0; s < kNumSlots; ++s) pos[s] = start[(size_t)g * kNumSlots + s];
            (void)scan_group(g, pass, [&](int slot, const uint4& e, uint32_t cyx) {
                const uint32_t i = pos[slot]++;
                ent[i] = e;
                if (cnt) cnt[i] = cyx;
            }, nullptr);
            return;
        }
        // dequant matrices (only the types that appear), flat, 16-byte aligned offsets
        memset(deq_off, 0, 27 * 3 * 4);
        auto nonneg_finite = [](float v) { uint32_t u; memcpy(&u, &v, 4); return (u >> 31) == 0 && (u & 0x7f800000u) != 0x7f800000u; };
        for (int t = 0; t < 27; ++t) {
            if (!(used_mask >> t & 1u)) continue;
            const size_t n = (size_t)kSize[t][0] * 8 * kSize[t][1] * 8;
            for (int c = 0; c < 3; ++c) {
                deq_off[t * 3 + c] = (uint32_t)deq_words;
                f->deq_off_host[t][c] = (uint32_t)deq_words;
                memcpy(deq + deq_words, d->dequant[t][c], n * 4);
                // the list-fed kernels never touch the zeros: 0 * quant_bias * matrix * mul must be +0.0, true when the
                // factors are finite and not negative — every conforming stream (the reference rejects weights <= 0 or
                // >= 1e8, jxl-vardct/src/dequant.rs:191-196, 391-396)
                if (grouped)
                    for (size_t i = 0; i < n; ++i) zero_stays_zero &= nonneg_finite(d->dequant[t][c][i]);
                deq_words += n;
            }
        }
        for (int c = 0; c < 3; ++c) zero_stays_zero &= nonneg_finite(d->quant_bias[c]);
        // quant_bias_numerator / k with the host's IEEE f32 division == the device's correctly
        // rounded one; entries 0 and 1 are never selected (|q| <= 1 takes the quant_bias branch)
        float* lut = reinterpret_cast<float*>(H + off_lut);
        lut[0] = lut[1] = 0.0f;
        for (int k = 2; k < 256; ++k) lut[k] = d->quant_bias_numerator / (float)k;
        // sec_half(64/128/256): dct_common.rs:56-66
        for (int i = 0, n = 64; i < 3; ++i, n *= 2) {
            float* tbl = reinterpret_cast<float*>(H + off_sec[i]);
            if (d->sec_half_large[i]) {
                memcpy(tbl, d->sec_half_large[i], sizeof(float) * (n / 2));
            } else {
                for (int k = 0; k < n / 2; ++k) {
                    float theta = (float)(2 * k + 1) / (float)(2 * n) * 3.14159265358979323846f;
                    tbl[k] = (1.0f / cosf(theta)) / 2.0f;
                }
            }
        }
        for (int i = 0; i < 3; ++i)
            if (!upw[i].empty()) memcpy(H + off_upw[i], upw[i].data(), upw[i].size() * 4);
    });
    ctx->up_split[0] = ms_since(t_phase1);
    f->nz_total = nz_total;
    // grouped lists feed the transform kernels directly; dense cells are built from them only for frames
    // with >= 128-px varblocks (global-memory path) or on request (JXLGPU_NO_SPARSE_TR)
    // ... and for progressive frames: the passes are summed into dense cells first (integer accumulation)
    f->sparse_tr = grouped && n_passes == 1 && !ctx->tune.no_sparse_tr && f->list_count[CLS_BIG] == 0 && zero_stays_zero;

    // ---- trim the arena to what was used: entries, nzc, nometa and the matrices are the tail items
    // (the items keep their offsets; the copy stops after the last used byte of each, the device allocation keeps the bound)
    const auto t_dev = std::chrono::steady_clock::now();
    char* dev_arena = nullptr;
    if (!ctx->guard_mode) {
        void* p = nullptr;
        HIP_TRY(ctx, ctx_dev_malloc(ctx, &p, pool_bucket(ar.total)));
        f->allocs.push_back(p);
        dev_arena = static_cast<char*>(p);
        for (const ArenaItem& it : ar.items) *it.dev = dev_arena + it.off;
        // one copy for the fixed part + the lists up to the entries; the used parts of the tail items after it
        struct Piece { size_t off, bytes; };
        std::vector<Piece> pieces = {{0, off_entries}, {off_entries, (size_t)n_entries * sizeof(uint4)},
                                     {off_nzc, grouped ? (size_t)n_entries * 4 : 0}, {off_nometa, (size_t)n_nometa * 4},
                                     {off_deq, deq_words * 4}};
        for (uint32_t p = 1; p < n_passes; ++p) {
            pieces.push_back({off_pass_entries[p], (size_t)n_entries * sizeof(uint4)});
            pieces.push_back({off_pass_nzc[p], (size_t)n_entries * 4});
        }
        if (!ctx->ev_h2d[0]) {
            HIP_TRY(ctx, hipEventCreate(&ctx->ev_h2d[0]));
            HIP_TRY(ctx, hipEventCreate(&ctx->ev_h2d[1]));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_h2d[0], ctx->stream_up));
        for (const Piece& pc : pieces)
            if (pc.bytes) HIP_TRY(ctx, hipMemcpyAsync(dev_arena + pc.off, H + pc.off, pc.bytes, hipMemcpyHostToDevice, ctx->stream_up));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_h2d[1], ctx->stream_up));
        ctx->h2d_timed = true;
    } else {
        // JXLGPU_GUARD: every array in its own guarded mapping, exact size (out-of-bounds accesses must fault)
        for (const ArenaItem& it : ar.items) {
            size_t bytes = it.bytes;
            if (it.off == off_entries) bytes = (size_t)n_entries * sizeof(uint4);
            else if (grouped && it.off == off_nzc) bytes = (size_t)n_entries * 4;
            else if (it.off == off_nometa) bytes = (size_t)n_nometa * 4;
            else if (it.off == off_deq) bytes = deq_words * 4;
            for (uint32_t p = 1; p < n_passes; ++p) {
                if (it.off == off_pass_entries[p]) bytes = (size_t)n_entries * sizeof(uint4);
                if (it.off == off_pass_nzc[p]) bytes = (size_t)n_entries * 4;
            }
            void* p = nullptr;
            HIP_TRY(ctx, ctx_dev_malloc(ctx, &p, std::max<size_t>(bytes, 16)));
            f->allocs.push_back(p);
            *it.dev = p;
            if (bytes) HIP_TRY(ctx, hipMemcpyAsync(p, H + it.off, bytes, hipMemcpyHostToDevice, ctx->stream_up));
        }
    }
    HIP_TRY(ctx, hipEventRecord(sb.ev, ctx->stream_up));
    sb.busy = true;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, sb.ev, 0));  // everything rendered from this frame comes after the arena
    for (int c = 0; c < 3; ++c) f->lfq[c] = lfq_dev[c];
    if (!n_nometa) f->nometa_groups = nullptr;

    // ---- the frame's working buffers (pooled, exact sizes) and the coefficient planes of the other transports
    Scratch tmp;
    uint32_t* d_bad = nullptr;
    if (d->coeff_format == JXLGPU_COEFF_SPARSE) {
        void* d_bad_v = nullptr;
        TRY(tmp.alloc(ctx, &d_bad_v, 4));
        d_bad = static_cast<uint32_t*>(d_bad_v);
        HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    }
    if (!f->sparse_tr) TRY(dev_alloc(ctx, f, &f->coeff, npix * 3));
    TRY(dev_alloc(ctx, f, &f->pix_t, npix * 3));
    if (d->coeff_format != JXLGPU_COEFF_DENSE && f->coeff) HIP_TRY(ctx, hipMemsetAsync(f->coeff, 0, npix * 12, ctx->stream));
    for (int c = 0; c < 3; ++c) {
        if (!grouped) TRY(upload_coeff_plane(ctx, f, d, c, tmp, d_bad));
        TRY(dev_alloc(ctx, f, &f->lf_a[c], ncell));
        TRY(dev_alloc(ctx, f, &f->lf[c], ncell));
        if (use_lf_frame) {
            // the LF frame's samples ARE the LF image (vardct/mod.rs:175-179): both LF plane sets hold them
            for (float* dst : {f->lf_a[c], f->lf[c]})
                HIP_TRY(ctx, hipMemcpy2D(dst, (size_t)f->w8 * 4, d->lf_frame[c], (size_t)d->lf_frame_stride * 4, (size_t)f->w8 * 4,
                                         f->h8, hipMemcpyHostToDevice));
        }
        if (o.no_post) continue;
        TRY(dev_alloc(ctx, f, &f->buf_a[c], npix));
        TRY(dev_alloc(ctx, f, &f->buf_b[c], npix));
    }
    if (grouped && !f->sparse_tr) {
        launch_grouped_to_dense(ctx->stream, f->entries, f->nzc, n_entries, f->nz, f->w8, f->coeff, false);
        for (uint32_t p = 1; p < n_passes; ++p)  // `+=`, hf_coeff.rs:234: stream order makes the passes' sums exact
            launch_grouped_to_dense(ctx->stream, pass_entries_dev[p], pass_nzc_dev[p], n_entries, f->nz, f->w8, f->coeff, true);
    }
    if (f->list_count[CLS_BIG]) TRY(dev_alloc(ctx, f, &f->big_tmp, npix * 6));
    if (upf > 1) {
        const uint32_t ow = d->width * upf, oh = d->height * upf;
        for (int c = 0; c < 3; ++c) TRY(dev_alloc(ctx, f, &f->up[c], (size_t)ow * oh));
    }

    // ---- per-frame scalars
    f->qm_scale[0] = powi_f32(0.8f, (int)d->x_qm_scale - 2);
    f->qm_scale[1] = 1.0f;
    f->qm_scale[2] = powi_f32(0.8f, (int)d->b_qm_scale - 2);
    {   // chroma_from_luma_lf, vardct/mod.rs:557-560
        int32_t x_factor = (int32_t)d->x_factor_lf - 128, b_factor = (int32_t)d->b_factor_lf - 128;
        f->kx_lf = d->base_correlation_x + ((float)x_factor / (float)d->colour_factor);
        f->kb_lf = d->base_correlation_b + ((float)b_factor / (float)d->colour_factor);
        if (o.no_cfl) f->kx_lf = f->kb_lf = 0.0f;  // vardct/mod.rs:184: skipped when subsampled; x + 0*y == x
    }
    for (int c = 0; c < 3; ++c) f->lf_div[c] = (float)(512.0 * (double)d->m_lf[c] / (double)scale_inv);
    fill_color_args(d->color, &f->color);
    f->noise_group_dim = d->group_dim;
    f->noise_corr_x = d->base_correlation_x;  // render.rs:175-180
    f->noise_corr_b = d->base_correlation_b;

    HIP_TRY(ctx, hipGetLastError());
    if (d->coeff_format == JXLGPU_COEFF_SPARSE) {
        // the scatter kernels validate the positions on the device: the one transport that waits for it
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t bad = 0;
        HIP_TRY(ctx, hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        if (bad) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "sparse coefficient position outside the frame");
    }
    // pointers inside the descriptor copy are dead from here on (everything they pointed to has been copied:
    // into the pinned staging arena, or by blocking copies)
    for (int c = 0; c < 3; ++c) f->desc.coeff[c] = f->desc.lf_frame[c] = nullptr;
    f->desc.lf_groups = nullptr;
    f->desc.hf_groups = nullptr;
    memset(f->desc.dequant, 0, sizeof(f->desc.dequant));
    frame_mark(ctx, f, ctx->stream);
    guard.armed = false;
    *out_frame = f;
    ctx->up_split[1] = 0.0;
    ctx->up_split[2] = ms_since(t_dev);
    ctx->up_split[3] = ms_since(t_begin);
    return JXLGPU_OK;
}

extern "C" {

int jxlgpu_frame_out_size(const jxlgpu_frame* f, uint32_t stages, uint32_t* width, uint32_t* height) {
    if (!f) return JXLGPU_ERR_INVALID_ARG;
    uint32_t k = 1;
    k = (stages & JXLGPU_STAGE_UPSAMPLE) && f->desc.upsampling.factor > 1 ? f->desc.upsampling.factor : 1;
    if (width) *width = f->width * k;
    if (height) *height = f->height * k;
    return JXLGPU_OK;
}

int jxlgpu_frame_result_size(const jxlgpu_frame* f, uint32_t* width, uint32_t* height) {
    if (!f || !f->result[0]) return JXLGPU_ERR_INVALID_ARG;
    if (width) *width = f->result_w;
    if (height) *height = f->result_h;
    return JXLGPU_OK;
}

const float* jxlgpu_frame_result_plane(const jxlgpu_frame* f, uint32_t c) {
    return (f && c < 3) ? f->result[c] : nullptr;
}

uint64_t jxlgpu_frame_algorithmic_bytes(const jxlgpu_frame* f, uint32_t stages) {
    if (!f) return 0;
    uint32_t ow = 0, oh = 0;
    jxlgpu_frame_out_size(f, stages, &ow, &oh);
    const uint64_t ncell = (uint64_t)f->w8 * f->h8;
    uint64_t bytes = 0;
    if (stages & JXLGPU_STAGE_TRANSFORM) bytes += (uint64_t)f->wr * f->hr * 12;   // 3 x i32 coefficients
    bytes += (uint64_t)ow * oh * 12;                                              // 3 x f32 result
    bytes += ncell * (3 * (f->lf_is_i16 ? 2 : 4) + 1 + 4);                        // LF quant, BlockInfo, hf_mul
    if (stages & JXLGPU_STAGE_EPF) bytes += ncell * 4;                            // sigma
    bytes += (uint64_t)f->w64 * f->h64 * 8;                                       // CfL maps
    return bytes;
}

int jxlgpu_frame_download_lf(jxlgpu_ctx* ctx, const jxlgpu_frame* f, float* const planes[3]) {
    if (!ctx || !f || !planes) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!f->subs.empty()) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "LF planes of a chroma-subsampled frame have per-channel sizes");
    const float* const* src = f->desc.skip_adaptive_lf_smoothing ? f->lf_a : f->lf;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < 3; ++c)
        if (planes[c])
            HIP_TRY(ctx, hipMemcpy(planes[c], src[c], (size_t)f->w8 * f->h8 * 4, hipMemcpyDeviceToHost));
    return JXLGPU_OK;
}

}  // extern "C"

// Gabor -> EPF -> upsample -> colour on device planes; shared by the VarDCT and Modular paths.
// `cur` holds W x H samples with stride `*cur_stride`; on return it points at the result.
// `tiled_in` (VarDCT): the input is the cell-tiled transform output f->pix_t instead of `cur`; the
// fused post kernels read it as it is, every other consumer gets row-major planes first.
// `region` (null: everything): the rectangle of the OUTPUT (after upsampling) that is wanted.  The
// stages are cut to it where the kernels can be (fused filters, upsampling, colour); the planes keep
// the whole frame's addressing, so the caller finds the result at (region->x0, region->y0).
int run_post_stages(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuFilterParams& fp,
                    uint32_t up_factor, float* cur[3], uint32_t* cur_stride, uint32_t* ow, uint32_t* oh,
                    bool tiled_in, const PixRect* region) {
    hipStream_t s = ctx->stream;
    const uint32_t W = f->width, H = f->height;
    const bool do_gab = (stages & JXLGPU_STAGE_GABOR) && fp.gab_enabled;
    const int epf_iters = (stages & JXLGPU_STAGE_EPF) ? (int)fp.epf_iters : 0;
    const bool do_up = (stages & JXLGPU_STAGE_UPSAMPLE) && up_factor > 1;
    const bool do_color = (stages & JXLGPU_STAGE_COLOR) && (f->desc.color.enabled || f->desc.color.ycbcr);
    const bool do_noise = (stages & JXLGPU_STAGE_NOISE) && f->desc.noise.enabled;
    // noise sits between upsampling and colour; HLG op lists are evaluated by the staged colour kernel only
    const bool fuse_color = do_color && !do_up && !do_noise && !f->color.staged_only;
    if (region && do_noise)
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "noise synthesis with a region (the generator is seeded per absolute group)");
    // the rectangle of coded samples the output region needs: its own, or with 2x / 4x / 8x upsampling the
    // samples under it plus the two of the 5x5 kernel's reach (upsampling.rs:45-132; util.rs:60-83)
    PixRect crect{0, 0, (int)W, (int)H};
    if (region) {
        const int k = do_up ? (int)up_factor : 1;
        auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        crect = PixRect{fdiv(region->x0, k) - (do_up ? 2 : 0), fdiv(region->y0, k) - (do_up ? 2 : 0),
                        fdiv(region->x1 + k - 1, k) + (do_up ? 2 : 0), fdiv(region->y1 + k - 1, k) + (do_up ? 2 : 0)};
        crect.x0 = std::max(crect.x0, 0); crect.y0 = std::max(crect.y0, 0);
        crect.x1 = std::min(crect.x1, (int)W); crect.y1 = std::min(crect.y1, (int)H);
    }
    const PixRect* crp = region ? &crect : nullptr;

    // Fast path: everything after the transform in one tile kernel (fused_kernels.hip)
    const bool fused = (do_gab || epf_iters) && fused_post_supported(ctx, f, do_gab, epf_iters);
    if (tiled_in && !fused) {
        // no fused kernel in front: planes for whoever reads next (staged filters, upsampling,
        // colour, the download of a transform-only render)
        launch_untile(s, f->pix_t, f->w8, f->buf_a, f->wr, f->wr, f->hr);
        for (int c = 0; c < 3; ++c) cur[c] = f->buf_a[c];
        *cur_stride = f->wr;
    }
    if (fused) {
        const float* in[3] = {tiled_in ? f->pix_t : cur[0], cur[1], cur[2]};
        float** dst = (!tiled_in && cur[0] == f->buf_a[0]) ? f->buf_b : f->buf_a;
        HIP_TRY(ctx, launch_fused_post(s, f, in, *cur_stride, tiled_in ? f->w8 : 0u, dst, f->wr, do_gab, epf_iters,
                                       fuse_color, ctx, crp));
        for (int c = 0; c < 3; ++c) cur[c] = dst[c];
        *cur_stride = f->wr;
        if (fuse_color) {
            *ow = W; *oh = H;
            return JXLGPU_OK;
        }
    } else {
        auto other = [&](float* const* now) { return now[0] == f->buf_a[0] ? f->buf_b : f->buf_a; };
        FilterArgs fa;
        fa.width = W; fa.height = H;
        fa.sigma = f->sigma; fa.sigma_stride = f->w8;
        fa.fp = fp;
        auto run = [&](int what, int step) {
            float** dst = other(cur);
            for (int c = 0; c < 3; ++c) { fa.in[c] = cur[c]; fa.out[c] = dst[c]; }
            fa.in_stride = *cur_stride; fa.out_stride = f->wr;
            if (what == 0) launch_gabor(s, fa); else launch_epf(s, step, fa);
            for (int c = 0; c < 3; ++c) cur[c] = dst[c];
            *cur_stride = f->wr;
        };
        if (do_gab) run(0, 0);
        if (epf_iters == 3) run(1, 0);   // filter/epf.rs:44-93: step 0 only for iters == 3
        if (epf_iters >= 1) run(1, 1);
        if (epf_iters >= 2) run(1, 2);
    }
    *ow = W; *oh = H;
    if (do_up) {
        // features/upsampling.rs:18-41: 8x passes first, then the 2x / 4x remainder
        const int log2f = up_factor == 2 ? 1 : up_factor == 4 ? 2 : 3;
        const int k = log2f == 3 ? 8 : (log2f == 1 ? 2 : 4);
        const float* kern = f->up_weights[k == 2 ? 0 : k == 4 ? 1 : 2];
        // 2x (BASELINE config 5): one streaming pass for the three planes, colour transform fused
        // when nothing (noise) sits between the two
        const bool fuse = do_color && !do_noise && !f->color.staged_only;
        // output window: the region; for the 2x streaming kernel, the input samples under it
        PixRect win2{0, 0, (int)W, (int)H};
        if (region) win2 = PixRect{std::max(region->x0, 0) / 2, std::max(region->y0, 0) / 2,
                                   std::min((region->x1 + 1) / 2, (int)W), std::min((region->y1 + 1) / 2, (int)H)};
        if (k == 2 && f->have_up2 && !ctx->tune.no_fused &&
            launch_upsample2_stream(s, cur, *cur_stride, W, H, f->up, W * k, f->up2_wq, fuse ? &f->color : nullptr,
                                    region ? &win2 : nullptr, ctx->num_cus, ctx->tune.up2_variant, ctx->tune.up2_rows)) {
            for (int c = 0; c < 3; ++c) cur[c] = f->up[c];
            *ow = W * k; *oh = H * k; *cur_stride = W * k;
            if (fuse) return JXLGPU_OK;
        } else {
            for (int c = 0; c < 3; ++c) launch_upsample(s, cur[c], *cur_stride, W, H, f->up[c], W * k, k, kern, region);
            for (int c = 0; c < 3; ++c) cur[c] = f->up[c];
            *ow = W * k; *oh = H * k; *cur_stride = W * k;
        }
    }
    if (do_noise) {
        // render.rs:207-222: after upsampling, on the frame's final size
        if (noise_geometry_unsupported(*oh, f->noise_group_dim))
            return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "noise on a frame whose last group row is one sample high (the reference panics there)");
        if (!ctx->noise_jump) {
            HIP_TRY(ctx, ctx_dev_malloc(ctx, &ctx->noise_jump, noise_jump_table_bytes()));
            HIP_TRY(ctx, hipMemcpy(ctx->noise_jump, noise_jump_table_host(), noise_jump_table_bytes(), hipMemcpyHostToDevice));
        }
        if (!f->noise_raw[0] || f->noise_w != *ow || f->noise_h != *oh) {
            for (int c = 0; c < 3; ++c) TRY(dev_alloc(ctx, f, &f->noise_raw[c], (size_t)*ow * *oh));
            f->noise_w = *ow; f->noise_h = *oh;
        }
        if (cur[0] != f->buf_a[0] && cur[0] != f->buf_b[0] && cur[0] != f->up[0]) {
            // the noise is added in place: never into the transform output, which a later render of
            // the same frame with other stages would read again
            float** dst = f->buf_a;
            for (int c = 0; c < 3; ++c)
                HIP_TRY(ctx, hipMemcpy2DAsync(dst[c], (size_t)f->wr * 4, cur[c], (size_t)*cur_stride * 4, (size_t)*ow * 4,
                                              *oh, hipMemcpyDeviceToDevice, s));
            for (int c = 0; c < 3; ++c) cur[c] = dst[c];
            *cur_stride = f->wr;
        }
        launch_noise(s, f->desc.noise, ctx->noise_jump, f->noise_raw, cur, *cur_stride, *ow, *oh, f->noise_group_dim,
                     f->noise_corr_x, f->noise_corr_b);
    }
    if (ctx->tune.debug_sync) HIP_TRY(ctx, hipStreamSynchronize(s));
    if (do_color && region) {
        // in place on the rectangle (whole-frame addressing kept)
        float* sub[3];
        for (int c = 0; c < 3; ++c) sub[c] = cur[c] + (size_t)region->y0 * *cur_stride + region->x0;
        launch_color(s, f->color, sub, *cur_stride, (uint32_t)(region->x1 - region->x0), (uint32_t)(region->y1 - region->y0));
    } else if (do_color) {
        launch_color(s, f->color, cur, *cur_stride, *ow, *oh);
    }
    if (ctx->tune.debug_sync) HIP_TRY(ctx, hipStreamSynchronize(s));
    return JXLGPU_OK;
}

// Clips a caller's region to the w x h output; false if nothing is left.
bool clip_region(const JxlGpuRegion* r, uint32_t w, uint32_t h, PixRect* out) {
    if (!r) return false;
    const int64_t x0 = std::max<int64_t>(r->left, 0), y0 = std::max<int64_t>(r->top, 0);
    const int64_t x1 = std::min<int64_t>((int64_t)r->left + r->width, w), y1 = std::min<int64_t>((int64_t)r->top + r->height, h);
    if (x1 <= x0 || y1 <= y0) return false;
    *out = PixRect{(int)x0, (int)y0, (int)x1, (int)y1};
    return true;
}

int finish_render(jxlgpu_ctx* ctx, jxlgpu_frame* f, float* cur[3], uint32_t stride, uint32_t ow, uint32_t oh,
                  const JxlGpuOut* out);
// finish_render for a region: the planes keep the whole frame's addressing, the result is the rectangle
int finish_render_region(jxlgpu_ctx* ctx, jxlgpu_frame* f, float* cur[3], uint32_t stride, const PixRect& r, const JxlGpuOut* out) {
    float* sub[3];
    for (int c = 0; c < 3; ++c) sub[c] = cur[c] + (size_t)r.y0 * stride + r.x0;
    return finish_render(ctx, f, sub, stride, (uint32_t)(r.x1 - r.x0), (uint32_t)(r.y1 - r.y0), out);
}

// Host-side copy out of the pinned staging buffer, split over a few threads (one memcpy stream
// moves ~10 GB/s; the PCIe link delivers ~50).
void parallel_copy_rows(char* dst, size_t dst_stride, const char* src, size_t src_stride, size_t row_bytes,
                        uint32_t rows) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::min<unsigned>(8, hw ? hw : 1);
    if ((size_t)rows * row_bytes < ((size_t)4 << 20) || rows < nt) nt = 1;
    auto work = [=](uint32_t y0, uint32_t y1) {
        if (dst_stride == row_bytes && src_stride == row_bytes) {
            memcpy(dst + (size_t)y0 * row_bytes, src + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes);
        } else {
            for (uint32_t y = y0; y < y1; ++y) memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride, row_bytes);
        }
    };
    if (nt == 1) { work(0, rows); return; }
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; ++i)
        th.emplace_back(work, (uint32_t)((uint64_t)rows * i / nt), (uint32_t)((uint64_t)rows * (i + 1) / nt));
    work(0, (uint32_t)((uint64_t)rows / nt));
    for (auto& t : th) t.join();
}

int finish_render(jxlgpu_ctx* ctx, jxlgpu_frame* f, float* cur[3], uint32_t stride, uint32_t ow, uint32_t oh,
                  const JxlGpuOut* out) {
    for (int c = 0; c < 3; ++c) f->result[c] = cur[c];
    f->result_stride = stride; f->result_w = ow; f->result_h = oh;
    HIP_TRY(ctx, hipGetLastError());
    frame_mark(ctx, f, ctx->stream);
    if (out) {
        if (out->stride < ow) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "output stride < output width");
        if (out->mem > JXLGPU_MEM_HOST_PINNED) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "unknown JxlGpuOut.mem");
        if (out->mem == JXLGPU_MEM_HOST_PINNED) {
            // the caller's planes are pinned (jxlgpu_host_alloc): the DMA engine writes them directly
            for (int c = 0; c < 3; ++c)
                if (out->planes[c])
                    HIP_TRY(ctx, hipMemcpy2DAsync(out->planes[c], (size_t)out->stride * 4, cur[c], (size_t)stride * 4,
                                                  (size_t)ow * 4, oh, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        } else if (out->mem == JXLGPU_MEM_DEVICE) {
            for (int c = 0; c < 3; ++c)
                if (out->planes[c])
                    HIP_TRY(ctx, hipMemcpy2DAsync(out->planes[c], (size_t)out->stride * 4, cur[c], (size_t)stride * 4,
                                                  (size_t)ow * 4, oh, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            // D2H into the ctx's pinned staging buffer (tight rows), then into the caller's grids
            const size_t plane = (size_t)ow * oh * 4;
            if (ctx->pinned_size < plane * 3) {
                if (ctx->pinned) (void)hipHostFree(ctx->pinned);
                ctx->pinned = nullptr; ctx->pinned_size = 0;
                HIP_TRY(ctx, hipHostMalloc(&ctx->pinned, plane * 3, hipHostMallocDefault));
                ctx->pinned_size = plane * 3;
            }
            // Plane c lands in its third of the staging buffer; while plane c+1 is still crossing
            // PCIe, worker threads move plane c into the caller's grid.
            for (int c = 0; c < 3; ++c) {
                if (!ctx->ev_d2h[c]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_d2h[c], hipEventDisableTiming));
                HIP_TRY(ctx, hipMemcpy2DAsync((char*)ctx->pinned + plane * c, (size_t)ow * 4, cur[c], (size_t)stride * 4,
                                              (size_t)ow * 4, oh, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_d2h[c], ctx->stream));
            }
            for (int c = 0; c < 3; ++c) {
                HIP_TRY(ctx, hipEventSynchronize(ctx->ev_d2h[c]));
                if (out->planes[c])
                    parallel_copy_rows((char*)out->planes[c], (size_t)out->stride * 4, (const char*)ctx->pinned + plane * c,
                                       (size_t)ow * 4, (size_t)ow * 4, oh);
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return JXLGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Chroma-subsampled frames (JPEG transcodes).  Channels of different sizes are
// different *geometries*: the frame becomes one child frame per distinct (hshift, vshift), each a
// plain single-geometry frame whose three channel slots keep their own parameters (quant bias,
// qm scale, dequant matrices, LF scale); slots that do not belong to the geometry are pointed at a
// member's data and their output is ignored.  Children keep the parent's LF-group tiling
// (256 >> shift cells), so the per-group `lf_quant` arrays and `extra_precision` pass through
// unchanged.  After the children's V1-V8, upsample_jpeg brings every channel to full resolution
// (image.rs:448-485) and the common post stages run on the parent.
namespace {

uint32_t ssize1(uint32_t n, bool has, bool sub) {  // ChannelShift::shift_size, param.rs:142-165
    if (!has) return n;
    const uint32_t size = (n + 1) / 2;
    return sub ? size : size * 2;
}

}  // namespace

int upload_subsampled(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* d, jxlgpu_frame** out_frame) {
    if (!d->skip_adaptive_lf_smoothing)
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "chroma-subsampled frame with adaptive LF smoothing stays on the CPU path");
    if (d->group_dim != 256) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "only group_dim == 256 is supported");
    if ((d->upsampling.factor ? d->upsampling.factor : 1) != 1)
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "chroma-subsampled frame with non-separable upsampling");
    if (d->coeff_format != JXLGPU_COEFF_DENSE)
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "sparse / grouped coefficient transport on a chroma-subsampled frame");
    if (d->lf_frame[0] || d->lf_frame[1] || d->lf_frame[2])
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "chroma-subsampled frame with an LF frame stays on the CPU path");
    if (d->width == 0 || d->height == 0 || d->width > (1u << 18) || d->height > (1u << 18))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "bad frame size");
    if (d->height > 65535u) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "output taller than 65535 rows");
    if (!d->lf_groups) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "no LF groups");
    // ChannelShift::from_jpeg_upsampling, param.rs:105-122
    bool has_h = false, has_v = false;
    for (int i = 0; i < 3; ++i) {
        if (d->jpeg_upsampling[i] > 3) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "jpeg_upsampling > 3");
        has_h |= d->jpeg_upsampling[i] == 1 || d->jpeg_upsampling[i] == 2;
        has_v |= d->jpeg_upsampling[i] == 1 || d->jpeg_upsampling[i] == 3;
    }
    int hs[3], vs[3];
    for (int c = 0; c < 3; ++c) {
        switch (d->jpeg_upsampling[c]) {
            case 0: hs[c] = has_h; vs[c] = has_v; break;
            case 1: hs[c] = 0; vs[c] = 0; break;
            case 2: hs[c] = 0; vs[c] = has_v; break;
            default: hs[c] = has_h; vs[c] = 0; break;
        }
    }
    const uint32_t W = d->width, H = d->height, W8 = ceil_div(W, 8), H8 = ceil_div(H, 8);
    const uint32_t lf_dim = d->group_dim * 8, per_row = ceil_div(W, lf_dim), lf_rows = ceil_div(H, lf_dim);
    if (d->num_lf_groups != per_row * lf_rows) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "num_lf_groups does not match the frame size");
    if (d->coeff_stride < ssize1(W8, has_h, false) * 8 || (d->coeff_stride & 1))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "coeff_stride < width_rounded (or odd)");

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    jxlgpu_frame* f = new (std::nothrow) jxlgpu_frame();
    if (!f) return JXLGPU_ERR_OOM;
    struct Guard {
        jxlgpu_ctx* c; jxlgpu_frame* f; bool armed = true;
        ~Guard() { if (armed) jxlgpu_frame_free(c, f); }
    } guard{ctx, f};
    f->desc = *d;
    f->width = W; f->height = H;
    f->w8 = W8; f->h8 = H8; f->wr = W8 * 8; f->hr = H8 * 8;
    f->w64 = ceil_div(W, 64); f->h64 = ceil_div(H, 64);
    f->group_dim = d->group_dim;
    f->lfg_cells_x = f->lfg_cells_y = d->group_dim;
    f->lf_groups_per_row = per_row; f->num_lf_groups = d->num_lf_groups;

    // frame-level sigma for the post stages; every varblock must be DCT8 (what JPEG transcodes are:
    // for_each_varblocks, vardct/mod.rs:693-730, maps other shapes onto overlapping chroma blocks)
    std::vector<float> sigma((size_t)W8 * H8, d->filter.epf_sigma_for_modular);
    for (uint32_t g = 0; g < d->num_lf_groups; ++g) {
        const JxlGpuLfGroup& lg = d->lf_groups[g];
        const uint32_t gx = g % per_row, gy = g / per_row;
        if (lg.width_px != std::min(lf_dim, W - gx * lf_dim) || lg.height_px != std::min(lf_dim, H - gy * lf_dim))
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "LF group size does not match the frame geometry");
        if (!lg.has_hf_meta) continue;
        if (!lg.block_kind || !lg.hf_mul) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "HfMetadata pointers missing");
        const uint32_t gbw = ceil_div(lg.width_px, 8), gbh = ceil_div(lg.height_px, 8);
        const uint32_t bw = ssize1(gbw, has_h, false), bh = ssize1(gbh, has_v, false);  // hf_metadata.rs:70-81
        for (uint32_t y = 0; y < bh; ++y)
            for (uint32_t x = 0; x < bw; ++x) {
                if (lg.block_kind[(size_t)y * bw + x] != JXLGPU_DCT8)
                    return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "chroma-subsampled frame with varblocks other than DCT8 stays on the CPU path");
                if (lg.epf_sigma && x < gbw && y < gbh)
                    sigma[(size_t)(gy * d->group_dim + y) * W8 + gx * d->group_dim + x] = lg.epf_sigma[(size_t)y * bw + x];
            }
    }

    // ---- one child per distinct geometry
    bool done[3] = {false, false, false};
    for (int lead = 0; lead < 3; ++lead) {
        if (done[lead]) continue;
        jxlgpu_frame::Sub sub;
        sub.hshift = hs[lead]; sub.vshift = vs[lead];
        for (int c = 0; c < 3; ++c) {
            sub.member[c] = hs[c] == hs[lead] && vs[c] == vs[lead];
            done[c] |= sub.member[c];
        }
        const bool sh = sub.hshift != 0, sv = sub.vshift != 0;
        JxlGpuVardctDesc cd = *d;
        for (int i = 0; i < 3; ++i) cd.jpeg_upsampling[i] = 0;
        cd.width = ssize1(W8, has_h, sh) * 8;
        cd.height = ssize1(H8, has_v, sv) * 8;
        cd.coeff_stride = d->coeff_stride >> sub.hshift;
        for (int c = 0; c < 3; ++c) cd.coeff[c] = d->coeff[sub.member[c] ? c : lead];
        cd.skip_adaptive_lf_smoothing = 1;
        memset(&cd.filter, 0, sizeof(cd.filter));
        memset(&cd.noise, 0, sizeof(cd.noise));
        memset(&cd.color, 0, sizeof(cd.color));
        cd.upsampling.factor = 1;
        // lf_quant channel k holds framebuffer slot kSlot[k] (util.rs:275-298)
        static const int kSlot[3] = {1, 0, 2};
        int k_of_lead = 0;
        for (int k = 0; k < 3; ++k) if (kSlot[k] == lead) k_of_lead = k;
        std::vector<JxlGpuLfGroup> groups(d->num_lf_groups);
        std::vector<std::vector<uint8_t>> kinds(d->num_lf_groups);
        std::vector<std::vector<int32_t>> muls(d->num_lf_groups), zeros(d->num_lf_groups);
        for (uint32_t g = 0; g < d->num_lf_groups; ++g) {
            const JxlGpuLfGroup& lg = d->lf_groups[g];
            JxlGpuLfGroup& cg = groups[g];
            cg = lg;
            const uint32_t gbw = ceil_div(lg.width_px, 8), gbh = ceil_div(lg.height_px, 8);
            const uint32_t pbw = ssize1(gbw, has_h, false);  // parent grid stride (rounded)
            const uint32_t lw = ssize1(gbw, has_h, sh), lh = ssize1(gbh, has_v, sv);
            cg.width_px = lw * 8; cg.height_px = lh * 8;
            for (int k = 0; k < 3; ++k) cg.lf_quant[k] = lg.lf_quant[sub.member[kSlot[k]] ? k : k_of_lead];
            cg.epf_sigma = nullptr;
            zeros[g].assign((size_t)ceil_div(cg.width_px, 64) * ceil_div(cg.height_px, 64), 0);
            cg.x_from_y = zeros[g].data(); cg.b_from_y = zeros[g].data();
            if (!lg.has_hf_meta) continue;
            kinds[g].assign((size_t)lw * lh, (uint8_t)JXLGPU_DCT8);
            muls[g].resize((size_t)lw * lh);
            for (uint32_t y = 0; y < lh; ++y)
                for (uint32_t x = 0; x < lw; ++x)
                    muls[g][(size_t)y * lw + x] = lg.hf_mul[(size_t)(y << sub.vshift) * pbw + (x << sub.hshift)];
            cg.block_kind = kinds[g].data();
            cg.hf_mul = muls[g].data();
        }
        cd.lf_groups = groups.data();
        UploadOpts o;
        o.lfg_cells_x = d->group_dim >> sub.hshift;
        o.lfg_cells_y = d->group_dim >> sub.vshift;
        o.no_cfl = true;
        o.no_post = true;
        TRY(vardct_upload_impl(ctx, &cd, o, &sub.child));
        f->subs.push_back(sub);
    }

    const size_t npix = (size_t)f->wr * f->hr;
    for (int c = 0; c < 3; ++c) {
        TRY(dev_alloc(ctx, f, &f->pix[c], npix));
        TRY(dev_alloc(ctx, f, &f->buf_a[c], npix));
        TRY(dev_alloc(ctx, f, &f->buf_b[c], npix));
    }
    TRY(dev_upload(ctx, f, &f->sigma, sigma));
    fill_color_args(d->color, &f->color);
    f->noise_group_dim = d->group_dim;
    f->noise_corr_x = d->base_correlation_x;
    f->noise_corr_b = d->base_correlation_b;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < 3; ++c) f->desc.coeff[c] = nullptr;
    f->desc.lf_groups = nullptr;
    memset(f->desc.dequant, 0, sizeof(f->desc.dequant));
    guard.armed = false;
    *out_frame = f;
    return JXLGPU_OK;
}

// Kernel arguments of one frame's V1-V3 / V4-V8 (shared by the single-frame and the batched path).
static void fill_lf_args(const jxlgpu_frame* f, LfArgs* la, SmoothArgs* sa) {
    for (int c = 0; c < 3; ++c) { la->lfq[c] = f->lfq[c]; la->out[c] = f->lf_a[c]; }
    la->is_i16 = f->lf_is_i16; la->scale = f->lf_scale;
    la->w8 = f->w8; la->h8 = f->h8; la->lf_groups_per_row = f->lf_groups_per_row;
    la->group_cells_x = f->lfg_cells_x; la->group_cells_y = f->lfg_cells_y;
    la->kx = f->kx_lf; la->kb = f->kb_lf;
    for (int c = 0; c < 3; ++c) { sa->in[c] = f->lf_a[c]; sa->out[c] = f->lf[c]; sa->lf_div[c] = f->lf_div[c]; }
    sa->w8 = f->w8; sa->h8 = f->h8;
}

static void fill_transform_args(jxlgpu_ctx* ctx, const jxlgpu_frame* f, TransformArgs* pta) {
    TransformArgs& ta = *pta;
    memset(&ta, 0, sizeof(ta));
    const JxlGpuVardctDesc& d = f->desc;
    float* const* lf = d.skip_adaptive_lf_smoothing ? f->lf_a : f->lf;
    ta.coeff = f->coeff;
    ta.pix = f->pix_t;
    for (int c = 0; c < 3; ++c) {
        ta.lf[c] = lf[c];
        ta.qm_scale[c] = f->qm_scale[c]; ta.quant_bias[c] = d.quant_bias[c];
    }
    ta.kind = f->kind; ta.hf_mul = f->hf_mul; ta.kx_map = f->kx_map; ta.kb_map = f->kb_map;
    ta.dequant = f->dequant; ta.deq_off = f->deq_off;
    memcpy(ta.deq_off_v, f->deq_off_host, sizeof(ta.deq_off_v));
    ta.sec64 = f->sec[0]; ta.sec128 = f->sec[1]; ta.sec256 = f->sec[2];
    ta.pstride = f->wr; ta.w8 = f->w8; ta.h8 = f->h8; ta.w64 = f->w64;
    ta.global_scale = (float)d.global_scale;
    ta.quant_bias_numerator = d.quant_bias_numerator;
    ta.big_tmp = f->big_tmp;
    ta.deq_lut = f->deq_lut;
    ta.nz = f->nz;
    ta.rect[0] = ta.rect[1] = 0; ta.rect[2] = ta.rect[3] = 65535;
#ifdef JXL_TR_PROFILE
    ta.prof = ctx->tr_prof;
#else
    (void)ctx;
#endif
}

extern "C" int jxlgpu_vardct_render(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuOut* out);

int render_subsampled(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuOut* out) {
    bool has_h = false, has_v = false;
    for (const auto& sub : f->subs) { has_h |= sub.hshift != 0; has_v |= sub.vshift != 0; }
    for (auto& sub : f->subs) TRY(jxlgpu_vardct_render(ctx, sub.child, stages & (JXLGPU_STAGE_LF | JXLGPU_STAGE_TRANSFORM), nullptr));
    if (!(stages & JXLGPU_STAGE_TRANSFORM)) return JXLGPU_OK;
    // upsample_jpeg, image.rs:448-485 / filter/ycbcr.rs:6-89
    ctx->prof_begin(PROF_POST);
    for (auto& sub : f->subs)
        for (int c = 0; c < 3; ++c) {
            if (!sub.member[c]) continue;
            const uint32_t in_w = ssize1(f->width, has_h, sub.hshift != 0), in_h = ssize1(f->height, has_v, sub.vshift != 0);
            launch_upsample_jpeg(ctx->stream, sub.child->pix_t, sub.child->w8, (uint32_t)c, in_w, in_h, sub.hshift,
                                 sub.vshift, f->pix[c], f->wr, f->width, f->height);
        }
    float* cur[3] = {f->pix[0], f->pix[1], f->pix[2]};
    uint32_t stride = f->wr, ow = f->width, oh = f->height;
    TRY(run_post_stages(ctx, f, stages, f->desc.filter, 1, cur, &stride, &ow, &oh, false, nullptr));
    ctx->prof_end(PROF_POST);
    return finish_render(ctx, f, cur, stride, ow, oh, out);
}

extern "C" {

static int vardct_render_impl(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuRegion* region_in,
                              const JxlGpuOut* out);

int jxlgpu_vardct_render(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuOut* out) {
    return vardct_render_impl(ctx, f, stages, nullptr, out);
}

int jxlgpu_vardct_render_region(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuRegion* region,
                                const JxlGpuOut* out) {
    if (!region) return JXLGPU_ERR_INVALID_ARG;
    return vardct_render_impl(ctx, f, stages, region, out);
}

static int vardct_render_impl(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuRegion* region_in,
                              const JxlGpuOut* out) {
    if (!ctx || !f || f->kind_of_frame != 0) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const JxlGpuVardctDesc& d = f->desc;
    // ---- region renders: the output rectangle, and from it the cells V4-V8 have to produce
    PixRect region{0, 0, 0, 0};
    bool cut = false;   // V4-V8 and the post stages are cut to the region (else: whole frame, cropped at the end)
    if (region_in) {
        if (!(stages & JXLGPU_STAGE_TRANSFORM)) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "a region render needs JXLGPU_STAGE_TRANSFORM");
        uint32_t fw = 0, fh = 0;
        jxlgpu_frame_out_size(f, stages, &fw, &fh);
        if (!clip_region(region_in, fw, fh, &region)) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "the region does not intersect the frame");
        const bool any_filter = ((stages & JXLGPU_STAGE_GABOR) && d.filter.gab_enabled) || ((stages & JXLGPU_STAGE_EPF) && d.filter.epf_iters);
        // cut when every consumer of the transform output can be: the fused filters, or no filters at all
        // (a frame with noise renders whole: the generator is seeded per absolute group; the region is cropped at the end)
        const bool noisy = (stages & JXLGPU_STAGE_NOISE) && d.noise.enabled;
        cut = f->subs.empty() && !noisy && (!any_filter || fused_post_supported(ctx, f, true, 2));
    }
    if (!f->subs.empty()) {
        // chroma-subsampled frames (JPEG transcodes): whole frame, the region is cropped from it
        if (!region_in) return render_subsampled(ctx, f, stages, out);
        TRY(render_subsampled(ctx, f, stages, nullptr));
        float* cur[3] = {const_cast<float*>(f->result[0]), const_cast<float*>(f->result[1]), const_cast<float*>(f->result[2])};
        return finish_render_region(ctx, f, cur, f->result_stride, region, out);
    }

    // ---- V1-V3
    LfArgs la;
    SmoothArgs sa;
    fill_lf_args(f, &la, &sa);
    ctx->prof_begin(PROF_LF);
    if (!f->lf_from_frame) {
        launch_lf_dequant_cfl(s, la);
        if (!d.skip_adaptive_lf_smoothing) launch_lf_smooth(s, sa);
    }
    ctx->prof_end(PROF_LF);
    if (!(stages & JXLGPU_STAGE_TRANSFORM)) {
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(s));
        if (out) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "an LF-only render has no pixel output: use jxlgpu_frame_download_lf");
        return JXLGPU_OK;
    }

    // ---- V4-V8
    TransformArgs ta;
    fill_transform_args(ctx, f, &ta);
    if (cut) {
        // coded samples the post stages may read: the colour rectangle of the region (upsampling support
        // included) + the reach of the cut launches (tiles of 32 + a 7-sample halo, twice for EPF iters 3): 96
        const int k = ((stages & JXLGPU_STAGE_UPSAMPLE) && d.upsampling.factor > 1) ? (int)d.upsampling.factor : 1;
        const int pad = 96 + (k > 1 ? 2 : 0);
        const int x0 = region.x0 / k - pad, y0 = region.y0 / k - pad, x1 = (region.x1 + k - 1) / k + pad, y1 = (region.y1 + k - 1) / k + pad;
        ta.rect[0] = (uint32_t)(std::max(x0, 0) / 8); ta.rect[1] = (uint32_t)(std::max(y0, 0) / 8);
        ta.rect[2] = (uint32_t)std::min<int>((x1 + 7) / 8, (int)f->w8); ta.rect[3] = (uint32_t)std::min<int>((y1 + 7) / 8, (int)f->h8);
    }
    ctx->prof_begin(PROF_TRANSFORM);
    // few, long work items first (64-px, 32-px shapes, the special 8x8 family), the bulk last
    if (f->sparse_tr) {
        for (int fam : {3, 2}) HIP_TRY(ctx, launch_transform_items_sparse(s, fam, ta, f->entries, f->nzc, f->class_first, f->list_count, ctx->num_cus));
        launch_transform_special_sparse(s, ta, f->entries + f->class_first[CLS_SPECIAL8], f->nzc + f->class_first[CLS_SPECIAL8],
                                        f->list_count[CLS_SPECIAL8]);
        for (int fam : {1, 0}) HIP_TRY(ctx, launch_transform_items_sparse(s, fam, ta, f->entries, f->nzc, f->class_first, f->list_count, ctx->num_cus));
    } else {
        for (int fam : {3, 2}) HIP_TRY(ctx, launch_transform_items(s, fam, ta, f->entries, f->class_first, f->list_count, ctx->num_cus, ctx->tune.tr_wgs_per_cu[fam]));
        launch_transform_class(s, CLS_SPECIAL8, ta, f->entries + f->class_first[CLS_SPECIAL8], f->list_count[CLS_SPECIAL8]);
        for (int fam : {1, 0}) HIP_TRY(ctx, launch_transform_items(s, fam, ta, f->entries, f->class_first, f->list_count, ctx->num_cus, ctx->tune.tr_wgs_per_cu[fam]));
        launch_transform_class(s, CLS_BIG, ta, f->entries + f->class_first[CLS_BIG], f->list_count[CLS_BIG]);
    }
    launch_nometa_groups(s, ta, f->nometa_groups, f->nometa_count, f->group_dim, ceil_div(f->width, f->group_dim));
    ctx->prof_end(PROF_TRANSFORM);

    if (!f->buf_a[0]) {  // single-geometry child of a chroma-subsampled frame: V1-V8 only, the parent reads pix_t
        HIP_TRY(ctx, hipGetLastError());
        return JXLGPU_OK;
    }
    float* cur[3] = {nullptr, nullptr, nullptr};  // the input of the post stages is the tiled f->pix_t
    uint32_t stride = f->wr, ow = f->width, oh = f->height;
    ctx->prof_begin(PROF_POST);
    TRY(run_post_stages(ctx, f, stages, d.filter, d.upsampling.factor ? d.upsampling.factor : 1, cur, &stride, &ow, &oh, true,
                        cut ? &region : nullptr));
    ctx->prof_end(PROF_POST);
    if (region_in) return finish_render_region(ctx, f, cur, stride, region, out);
    return finish_render(ctx, f, cur, stride, ow, oh, out);
}

// Device block of the default pipeline's arguments (FrameDev) for the batched launches; decides once
// per frame whether it qualifies.
static int ensure_dev_args(jxlgpu_ctx* ctx, jxlgpu_frame* f) {
    if (f->dev_args_ready) return JXLGPU_OK;
    f->batch_ok = false;
    const JxlGpuVardctDesc& d = f->desc;
    const uint32_t upf = d.upsampling.factor ? d.upsampling.factor : 1;
    f->batch_tr_ok = false;
    // V1-V8 of any single-geometry frame made of <= 64-px varblocks can share launches ...
    if (f->kind_of_frame != 0 || !f->subs.empty() || !f->buf_a[0] || f->list_count[CLS_BIG] || f->nometa_count || f->lf_from_frame) {
        f->dev_args_ready = true;
        return JXLGPU_OK;
    }
    // ... the post stage only for the default pipeline (Gabor + EPF iters 2 + plain XYB -> sRGB, no upsampling / noise)
    const bool post_default = d.filter.gab_enabled && d.filter.epf_iters == 2 && upf == 1 && !d.noise.enabled &&
                              d.color.enabled && !d.color.ycbcr && fused_post_supported(ctx, f, true, 2);
    FrameDev h;
    memset(&h, 0, sizeof(h));
    fill_lf_args(f, &h.lf, &h.smooth);
    h.skip_smooth = d.skip_adaptive_lf_smoothing ? 1u : 0u;
    fill_transform_args(ctx, f, &h.tr);
    for (int fam = 0; fam < 4; ++fam) {
        build_class_table(fam, f->class_first, f->list_count, ctx->num_cus, 0, &h.ct[fam]);
        f->batch_wgs[fam] = h.ct[fam].wg_begin[h.ct[fam].n_classes];
    }
    h.entries = f->entries;
    h.nzc = f->nzc;
    h.special_first = f->class_first[CLS_SPECIAL8];
    h.special_count = f->list_count[CLS_SPECIAL8];
    bool post_ok = false;
    if (post_default) {
        const float* in[3] = {f->pix_t, nullptr, nullptr};
        bool stream = false, plain_srgb = false;
        // batched launches have waves to spare: taller wave segments (less halo-row recompute)
        const int launch_frames = ctx->tune.batch_chunk > 0 ? std::min<int>(ctx->tune.batch_chunk, JXLGPU_MAX_BATCH)
                                                             : (ctx->tune.no_batch_overlap ? (int)JXLGPU_MAX_BATCH : 16);
        HIP_TRY(ctx, fused_prepare(ctx, f, in, f->wr, f->w8, f->buf_a, f->wr, true, 2, true, &h.post, &stream, &plain_srgb,
                                   ctx->tune.batch_stream_rows > 0 ? ctx->tune.batch_stream_rows : -launch_frames));
        f->batch_pk = h.post.pk != 0;
        // one kernel per batched launch: a frame that needs the other streaming kernel renders its post stage alone
        post_ok = stream && plain_srgb && f->batch_pk != ctx->tune.no_pk;
        if (post_ok) {
            h.post.tiles = f->ring_tiles;
            h.n_ring_tiles = f->n_ring_tiles;
            f->batch_stream_wgs = (uint32_t)(h.post.strips * h.post.segs + 3) / 4;
        }
    }
    if (!f->dev_args) TRY(dev_alloc(ctx, f, &f->dev_args, 1));
    HIP_TRY(ctx, hipMemcpy(f->dev_args, &h, sizeof(h), hipMemcpyHostToDevice));
    // only now: a failure above (allocation, copy) is reported and retried by the next call instead of
    // silently demoting the frame to one-by-one rendering for good
    f->dev_args_ready = true;
    f->batch_tr_ok = true;
    f->batch_ok = post_ok;
    return JXLGPU_OK;
}

// N frames, one launch per stage (SURVEY §8b "batch variants taking N descs"; the caller pattern is
// jxl-oxide-cli/src/decode.rs:293-304, keyframes rendered in a parallel loop).  Asynchronous like a
// render without an output descriptor: results stay on the device (jxlgpu_frame_result_plane,
// jxlgpu_frame_format_output) after jxlgpu_synchronize.  V1-V8 share launches unless a frame has
// >= 128-px varblocks, chroma subsampling or LF-only groups; the post stage shares launches for the
// default pipeline (Gabor + EPF iters 2 + plain XYB -> sRGB) and follows frame by frame otherwise
// (other filter settings, upsampling, noise, other colour chains).
int jxlgpu_vardct_render_batch(jxlgpu_ctx* ctx, jxlgpu_frame* const* frames, uint32_t n, uint32_t stages) {
    if (!ctx || (!frames && n)) return JXLGPU_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i)
        if (!frames[i] || frames[i]->kind_of_frame != 0) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t need_tr = JXLGPU_STAGE_LF | JXLGPU_STAGE_TRANSFORM;
    const uint32_t need = need_tr | JXLGPU_STAGE_GABOR | JXLGPU_STAGE_EPF | JXLGPU_STAGE_COLOR;
    // V1-V8 share launches whenever every frame's transform qualifies; the post stage as well when every
    // frame runs the default pipeline, otherwise it follows frame by frame on the transformed batch
    bool batched_tr = (stages & need_tr) == need_tr && n > 0;
    bool batched = (stages & need) == need;
    for (uint32_t i = 0; i < n && batched_tr; ++i) {
        TRY(ensure_dev_args(ctx, frames[i]));
        // one kernel per launch: list-fed and dense frames do not share one
        batched_tr = frames[i]->batch_tr_ok && frames[i]->sparse_tr == frames[0]->sparse_tr;
        batched = batched && frames[i]->batch_ok;
    }
    const bool sparse_tr = n > 0 && frames[0]->sparse_tr;
    if (!batched_tr) {
        for (uint32_t i = 0; i < n; ++i) TRY(jxlgpu_vardct_render(ctx, frames[i], stages, nullptr));
        return JXLGPU_OK;
    }
    // <= `chunk` frames per launch, stage after stage.  V1-V8 run on their own stream: the transform launches of
    // chunk k+1 (latency-bound: list-fed, a third of the VALU issue slots) share the CUs with the post launch of
    // chunk k (VALU-bound) instead of queueing behind it.  Order per frame: its transform waits for whatever was
    // queued last for that frame (upload, an earlier render that still reads the transform output), its post
    // stage waits for its transform.
    const bool overlap = !ctx->tune.no_batch_overlap;
    const uint32_t chunk = ctx->tune.batch_chunk > 0 ? std::min<uint32_t>((uint32_t)ctx->tune.batch_chunk, JXLGPU_MAX_BATCH)
                                                     : (overlap ? std::min<uint32_t>(16u, JXLGPU_MAX_BATCH) : JXLGPU_MAX_BATCH);
    hipStream_t st = overlap ? ctx->stream_tr : ctx->stream, sp = ctx->stream;
    // JXLGPU_BATCH_TR_MULT = k (experiment): the LF / transform launches take k chunks at once (their tails and the serial
    // head of their chain amortised over k times the frames), the post launches stay one chunk = one resident round of waves
    const uint32_t tchunk = (overlap && batched) ? std::min<uint32_t>(chunk * (uint32_t)std::max(1, ctx->tune.batch_tr_mult), JXLGPU_MAX_BATCH) : chunk;
    for (uint32_t i0 = 0; i0 < n; i0 += tchunk) {
        const uint32_t m = std::min<uint32_t>(tchunk, n - i0);
        FrameBatch b;
        memset(&b, 0, sizeof(b));
        uint32_t max_w8 = 0, max_h8 = 0, max_wgs[4] = {}, max_special = 0;
        bool no_event = false;
        for (uint32_t i = 0; i < m; ++i) {
            jxlgpu_frame* f = frames[i0 + i];
            b.f[i] = f->dev_args;
            max_w8 = std::max(max_w8, f->w8); max_h8 = std::max(max_h8, f->h8);
            for (int fam = 0; fam < 4; ++fam) max_wgs[fam] = std::max(max_wgs[fam], f->batch_wgs[fam]);
            max_special = std::max(max_special, f->list_count[CLS_SPECIAL8]);
            if (overlap && f->ev_last && f->ev_last_set) HIP_TRY(ctx, hipStreamWaitEvent(st, f->ev_last, 0));
            else if (overlap) no_event = true;   // (event creation / record failed earlier: order behind the render stream instead)
        }
        if (no_event) {
            // a frame without a usable "last operation" event: everything queued on the render and upload streams so far
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, sp));
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_fork, 0));
            if (ctx->stream_up) {
                HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream_up));
                HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
            }
        }
        ctx->prof_begin(PROF_LF, st);
        // V1-V3 of chunk `k0` (its frames' LF planes are their own); `wait`: behind the frames' last operations
        auto launch_lf_of = [&](uint32_t k0, bool wait) -> int {
            const uint32_t mk = std::min<uint32_t>(tchunk, n - k0);
            FrameBatch bl;
            memset(&bl, 0, sizeof(bl));
            uint32_t lw8 = 0, lh8 = 0;
            bool sm = false;
            for (uint32_t i = 0; i < mk; ++i) {
                jxlgpu_frame* f = frames[k0 + i];
                bl.f[i] = f->dev_args;
                lw8 = std::max(lw8, f->w8); lh8 = std::max(lh8, f->h8);
                sm |= !f->desc.skip_adaptive_lf_smoothing;
                if (wait && f->ev_last && f->ev_last_set) HIP_TRY(ctx, hipStreamWaitEvent(st, f->ev_last, 0));
            }
            HIP_TRY(ctx, launch_lf_batch(st, bl, mk, lw8, lh8, sm));
            return JXLGPU_OK;
        };
        // Where the LF launches of a chunk sit (JXLGPU_BATCH_LF_MODE, an experiment of round 6): 0 = in front of the chunk's own
        // transform launches (the default); 1 = in front of the PREVIOUS chunk's; 2 = BEHIND the previous chunk's 8- / 16-px launches
        // on the transform stream.  The kernel trace (profiles/r06_kernel_timeline.txt) shows the transform chain of chunk k + 1
        // starting together with the post + border-ring launches of chunk k and its first two (tiny) launches sitting 0.35-0.5 ms
        // behind the ring launch, on the chain that sets the period (1.7 ms against 1.13 ms of post).  Moving them does not help:
        // the launches that follow then wait instead (104.3 / 106.0 / 109-114 us per frame for modes 0 / 1 / 2) — whichever group
        // gets the registers first, the other one loses what it gains.
        int lf_mode = overlap ? ctx->tune.batch_lf_mode : 0;
        for (uint32_t i = 0; i < n && lf_mode; ++i)   // a frame without a usable "last operation" event is ordered by the no_event path above, chunk by chunk
            if (!(frames[i]->ev_last && frames[i]->ev_last_set)) lf_mode = 0;
        if (lf_mode == 0) TRY(launch_lf_of(i0, false));
        else if (i0 == 0) TRY(launch_lf_of(0, false));
        if (lf_mode == 1 && i0 + tchunk < n) TRY(launch_lf_of(i0 + tchunk, true));
        ctx->prof_end(PROF_LF, st);
        ctx->prof_begin(PROF_TRANSFORM, st);
        bool lf_done_ahead = false;
        const uint32_t heavy = overlap ? ctx->tune.batch_heavy : 0u;
        if (heavy) {
            // the families of `heavy` on the render stream, behind post(k-1) and in front of post(k) (they need the LF
            // stage of this chunk: an event); the others on the transform stream, beside post(k-1)
            hipEvent_t& evl = ctx->ev_tr[ctx->ev_tr_next++ % 8];
            if (!evl) HIP_TRY(ctx, hipEventCreateWithFlags(&evl, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(evl, st));
            HIP_TRY(ctx, hipStreamWaitEvent(sp, evl, 0));
            HIP_TRY(ctx, sparse_tr ? launch_transform_batch_sparse(sp, nullptr, b, m, max_wgs, max_special, heavy)
                                   : launch_transform_batch(sp, nullptr, b, m, max_wgs, max_special, heavy));
            HIP_TRY(ctx, sparse_tr ? launch_transform_batch_sparse(st, nullptr, b, m, max_wgs, max_special, 31u & ~heavy)
                                   : launch_transform_batch(st, nullptr, b, m, max_wgs, max_special, 31u & ~heavy));
        } else if ((int)m <= ctx->tune.tr_side_max) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
            hipStream_t side = overlap ? ctx->stream_tr2 : ctx->stream2;
            if (overlap && ctx->tune.tr_streams > 2) {
                // experiment (JXLGPU_TR_STREAMS = 3..5): every family on a stream of its own behind the LF stage, joined into `st`
                static const uint32_t kOrder[5] = {8u, 4u, 16u, 2u, 1u};   // 64-px, 32-px, special, 16-px, 8-px
                const int ns = std::min(5, ctx->tune.tr_streams);
                hipStream_t pool5[5] = {st, side, nullptr, nullptr, nullptr};
                for (int k = 2; k < ns; ++k) {
                    if (!ctx->stream_tr_extra[k - 2]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream_tr_extra[k - 2], hipStreamNonBlocking));
                    pool5[k] = ctx->stream_tr_extra[k - 2];
                }
                for (int k = 1; k < ns; ++k) HIP_TRY(ctx, hipStreamWaitEvent(pool5[k], ctx->ev_fork, 0));
                for (int i = 0; i < 5; ++i) {
                    hipStream_t q = pool5[(4 - i) % ns];   // the 8-px family (last) stays on `st`
                    HIP_TRY(ctx, sparse_tr ? launch_transform_batch_sparse(q, nullptr, b, m, max_wgs, max_special, kOrder[i])
                                           : launch_transform_batch(q, nullptr, b, m, max_wgs, max_special, kOrder[i]));
                }
                for (int k = 1; k < ns; ++k) {
                    if (!ctx->ev_tr_join[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_tr_join[k], hipEventDisableTiming));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_tr_join[k], pool5[k]));
                    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_tr_join[k], 0));
                }
            } else {
            HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_fork, 0));
            HIP_TRY(ctx, sparse_tr ? launch_transform_batch_sparse(st, side, b, m, max_wgs, max_special)
                                   : launch_transform_batch(st, side, b, m, max_wgs, max_special));
            if (lf_mode == 2 && i0 + tchunk < n) { TRY(launch_lf_of(i0 + tchunk, true)); lf_done_ahead = true; }
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join, side));
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
            }
        } else {
            HIP_TRY(ctx, sparse_tr ? launch_transform_batch_sparse(st, nullptr, b, m, max_wgs, max_special)
                                   : launch_transform_batch(st, nullptr, b, m, max_wgs, max_special));
        }
        if (lf_mode == 2 && !lf_done_ahead && i0 + tchunk < n) TRY(launch_lf_of(i0 + tchunk, true));   // (the other launch orders: behind the chunk's transforms)
        ctx->prof_end(PROF_TRANSFORM, st);
        if (overlap) {
            hipEvent_t& ev = ctx->ev_tr[ctx->ev_tr_next++ % 8];
            if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ev, st));
            HIP_TRY(ctx, hipStreamWaitEvent(sp, ev, 0));
        }
        if (!batched) {
            for (uint32_t i = 0; i < m; ++i) {
                jxlgpu_frame* f = frames[i0 + i];
                float* cur[3] = {nullptr, nullptr, nullptr};  // the input of the post stages is the tiled f->pix_t
                uint32_t stride = f->wr, ow = f->width, oh = f->height;
                ctx->prof_begin(PROF_POST);
                TRY(run_post_stages(ctx, f, stages, f->desc.filter, f->desc.upsampling.factor ? f->desc.upsampling.factor : 1,
                                    cur, &stride, &ow, &oh, true, nullptr));
                ctx->prof_end(PROF_POST);
                TRY(finish_render(ctx, f, cur, stride, ow, oh, nullptr));
            }
            continue;
        }
        for (uint32_t j0 = 0; j0 < m; j0 += chunk) {
            const uint32_t mp = std::min<uint32_t>(chunk, m - j0);
            FrameBatch bp;
            memset(&bp, 0, sizeof(bp));
            uint32_t p_stream = 0, p_ring = 0;
            for (uint32_t i = 0; i < mp; ++i) {
                const jxlgpu_frame* f = frames[i0 + j0 + i];
                bp.f[i] = f->dev_args;
                p_stream = std::max(p_stream, f->batch_stream_wgs);
                p_ring = std::max(p_ring, f->n_ring_tiles);
            }
            ctx->prof_begin(PROF_POST, sp);
            if (overlap && ctx->tune.ring_mode != 0) {
                // the border rings on the render stream itself, at full occupancy, in front of / behind the streaming kernel
                if (ctx->tune.ring_mode == 1) HIP_TRY(ctx, launch_post_batch(sp, nullptr, bp, mp, 0, p_ring, !ctx->tune.no_pk));
                HIP_TRY(ctx, launch_post_batch(sp, nullptr, bp, mp, p_stream, 0, !ctx->tune.no_pk, ctx->tune.post_fast, ctx->tune.post_lds_pad));
                if (ctx->tune.ring_mode == 2) HIP_TRY(ctx, launch_post_batch(sp, nullptr, bp, mp, 0, p_ring, !ctx->tune.no_pk));
            } else {
                // one fork / join per launch: the border rings run beside the streaming kernel
                HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, sp));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                HIP_TRY(ctx, launch_post_batch(sp, ctx->stream2, bp, mp, p_stream, p_ring, !ctx->tune.no_pk, ctx->tune.post_fast, ctx->tune.post_lds_pad));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
                HIP_TRY(ctx, hipStreamWaitEvent(sp, ctx->ev_join, 0));
            }
            ctx->prof_end(PROF_POST, sp);
            for (uint32_t i = 0; i < mp; ++i) {
                jxlgpu_frame* f = frames[i0 + j0 + i];
                for (int c = 0; c < 3; ++c) f->result[c] = f->buf_a[c];
                f->result_stride = f->wr; f->result_w = f->width; f->result_h = f->height;
                frame_mark(ctx, f, sp);
            }
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return JXLGPU_OK;
}

int jxlgpu_frame_download_result(jxlgpu_ctx* ctx, jxlgpu_frame* f, const JxlGpuOut* out) {
    if (!ctx || !f || !out) return JXLGPU_ERR_INVALID_ARG;
    if (!f->result[0]) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "the frame has not been rendered");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* cur[3] = {const_cast<float*>(f->result[0]), const_cast<float*>(f->result[1]), const_cast<float*>(f->result[2])};
    return finish_render(ctx, f, cur, f->result_stride, f->result_w, f->result_h, out);
}

int jxlgpu_vardct_render_host(jxlgpu_ctx* ctx, const JxlGpuVardctDesc* desc, uint32_t stages, const JxlGpuOut* out) {
    jxlgpu_frame* f = nullptr;
    int rc = jxlgpu_vardct_upload(ctx, desc, &f);
    if (rc != JXLGPU_OK) return rc;
    rc = jxlgpu_vardct_render(ctx, f, stages, out);
    jxlgpu_frame_free(ctx, f);
    return rc;
}

}  // extern "C"
