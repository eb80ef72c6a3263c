// Extra channels (alpha, depth, spot colours): int -> float with the channel's own bit depth, then the image's
// non-separable upsampling — what `ImageWithRegion::upsample_nonseparable` (jxl-render/src/image.rs:487-557) does to
// every channel that is not a colour channel.  The upsampling kernels are the colour path's (upsample_kernels.hip).
#include <algorithm>

#include "common.h"

namespace {

// BitDepth::parse_integer_sample, jxl-image/src/lib.rs:458-494
__global__ __launch_bounds__(256) void ec_to_float_kernel(const void* __restrict__ in, uint32_t is_i16, size_t n, uint32_t bit_depth,
                                                          uint32_t float_sample, uint32_t exp_bits, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t v = is_i16 ? (int32_t)static_cast<const int16_t*>(in)[i] : static_cast<const int32_t*>(in)[i];
    float r;
    if (!float_sample) {
        const int32_t div = (int32_t)((1u << bit_depth) - 1);
        r = (float)v / (float)div;
    } else {
        const uint32_t s = (uint32_t)v;
        const uint32_t mantissa_bits = bit_depth - exp_bits - 1;
        const uint32_t mantissa_mask = (1u << mantissa_bits) - 1;
        const uint32_t exp_mask = ((1u << (bit_depth - 1)) - 1) ^ mantissa_mask;
        const uint32_t is_signed = (s & (1u << (bit_depth - 1))) != 0;
        uint32_t mantissa = s & mantissa_mask;
        const int32_t exp = (int32_t)((s & exp_mask) >> mantissa_bits) - ((1 << (exp_bits - 1)) - 1);
        if (mantissa_bits < 23) mantissa <<= (23 - mantissa_bits);
        else if (mantissa_bits > 23) mantissa >>= (mantissa_bits - 23);
        r = __uint_as_float((is_signed << 31) | ((uint32_t)(exp + 127) << 23) | mantissa);
    }
    out[i] = r;
}

int fail(jxlgpu_ctx* ctx, int code, const char* msg) {
    ctx->last_error = msg;
    return code;
}

// Buffers of one jxlgpu_frame_render_extra call: everything but the final plane goes back to the pool (deferred: behind
// the work queued so far) when the call ends, on success and on failure alike.
struct ExtraScratch {
    jxlgpu_ctx* ctx;
    std::vector<void*> bufs;
    void* keep = nullptr;
    ~ExtraScratch() {
        std::vector<void*> rel;
        for (void* p : bufs)
            if (p != keep) rel.push_back(p);
        if (!rel.empty()) ctx_defer_release(ctx, std::move(rel));
    }
};
template <typename T>
int alloc_scratch(jxlgpu_ctx* ctx, ExtraScratch* sc, T** out, size_t bytes) {
    void* p = nullptr;
    HIP_TRY(ctx, ctx_dev_malloc(ctx, &p, std::max<size_t>(bytes, 16)));
    sc->bufs.push_back(p);
    *out = static_cast<T*>(p);
    return JXLGPU_OK;
}

}  // namespace

extern "C" int jxlgpu_frame_render_extra(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t index, const JxlGpuExtraChannel* ec,
                                         float* out, uint32_t out_stride, uint32_t out_mem) {
    if (!ctx || !f || !ec || index >= JXLGPU_MAX_EXTRA || out_mem > JXLGPU_MEM_HOST_PINNED) return JXLGPU_ERR_INVALID_ARG;
    if (!ec->data || ec->width == 0 || ec->height == 0 || ec->width > (1u << 18) || ec->height > (1u << 18))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: bad plane");
    if (ec->sample_type > JXLGPU_SAMPLE_I16) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: bad sample_type");
    if (ec->upsampling_log2 > 6) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: upsampling_log2 > 6");
    if (!ec->float_sample && (ec->bit_depth == 0 || ec->bit_depth > 31))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: bits_per_sample outside 1..31");
    if (ec->float_sample && (ec->bit_depth > 32 || ec->exp_bits == 0 || ec->exp_bits + 1 >= ec->bit_depth))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: bad float sample layout");
    const uint32_t L = ec->upsampling_log2, up8 = L / 3, last = L % 3;
    const uint64_t ow = (uint64_t)ec->width << L, oh = (uint64_t)ec->height << L;
    if (oh > 65535u || ow > (1u << 18)) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "extra channel: output taller than 65535 rows");
    if (L && (ec->width < 2 || ec->height < 2))  // the reference's padded copy and mirror() differ below two samples (util.rs:423-454)
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "extra channel: upsampling a plane less than two samples wide or high");
    if (out && out_stride < ow) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: output stride < output width");
    if ((up8 && !ec->weights.up8_weight) || (last == 1 && !ec->weights.up2_weight) || (last == 2 && !ec->weights.up4_weight))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "extra channel: upsampling weights missing");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    // ---- the integer plane, and the f32 plane of the channel's own size
    const size_t n = (size_t)ec->width * ec->height, esz = ec->sample_type == JXLGPU_SAMPLE_I16 ? 2 : 4;
    void* d_int = nullptr;
    float* cur = nullptr;
    ExtraScratch sc{ctx};
    int rc = alloc_scratch(ctx, &sc, &d_int, n * esz);
    if (rc) return rc;
    if ((rc = alloc_scratch(ctx, &sc, &cur, n * 4))) return rc;
    HIP_TRY(ctx, hipMemcpy(d_int, ec->data, n * esz, hipMemcpyHostToDevice));  // pageable source: returns when it has been read
    ec_to_float_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_int, ec->sample_type == JXLGPU_SAMPLE_I16, n, ec->bit_depth,
                                                                   ec->float_sample, ec->exp_bits, cur);

    // ---- upsampling: 8x passes first, then the 2x / 4x remainder (features/upsampling.rs:18-41)
    uint32_t w = ec->width, h = ec->height;
    auto pass = [&](int k, const float* coded) -> int {
        std::vector<float> wq = expand_up_weights_public(coded, k);
        float* d_w = nullptr;
        float* nxt = nullptr;
        int r = alloc_scratch(ctx, &sc, &d_w, wq.size() * 4);
        if (r) return r;
        HIP_TRY(ctx, hipMemcpy(d_w, wq.data(), wq.size() * 4, hipMemcpyHostToDevice));
        if ((r = alloc_scratch(ctx, &sc, &nxt, (size_t)w * k * h * k * 4))) return r;
        launch_upsample(s, cur, w, w, h, nxt, w * k, k, d_w);
        cur = nxt; w *= k; h *= k;
        return JXLGPU_OK;
    };
    for (uint32_t i = 0; i < up8; ++i)
        if ((rc = pass(8, ec->weights.up8_weight))) return rc;
    if (last == 1 && (rc = pass(2, ec->weights.up2_weight))) return rc;
    if (last == 2 && (rc = pass(4, ec->weights.up4_weight))) return rc;
    HIP_TRY(ctx, hipGetLastError());
    // the final plane stays with the frame; the plane of an earlier render of this index is released (deferred)
    if (f->extra[index]) {
        void* old = f->extra[index];
        auto it = std::find(f->allocs.begin(), f->allocs.end(), old);
        if (it != f->allocs.end()) {
            f->allocs.erase(it);
            ctx_defer_release(ctx, std::vector<void*>{old});
        }
    }
    sc.keep = cur;
    f->allocs.push_back(cur);
    f->extra[index] = cur; f->extra_w[index] = w; f->extra_h[index] = h;
    frame_mark(ctx, f, s);
    if (!out) return JXLGPU_OK;
    if (out_mem == JXLGPU_MEM_HOST_PINNED) {
        if (!f->ev_last) return JXLGPU_ERR_DEVICE;
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_down, f->ev_last, 0));
        HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)out_stride * 4, cur, (size_t)w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, ctx->stream_down));
        frame_mark(ctx, f, ctx->stream_down);
        return JXLGPU_OK;
    }
    HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)out_stride * 4, cur, (size_t)w * 4, (size_t)w * 4, h,
                                  out_mem == JXLGPU_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return JXLGPU_OK;
}

extern "C" const float* jxlgpu_frame_extra_plane(const jxlgpu_frame* f, uint32_t index, uint32_t* width, uint32_t* height) {
    if (!f || index >= JXLGPU_MAX_EXTRA || !f->extra[index]) return nullptr;
    if (width) *width = f->extra_w[index];
    if (height) *height = f->extra_h[index];
    return f->extra[index];
}

// ---- multi-GPU plumbing: exportable device memory and IPC mappings (include/jxlgpu.h "multi-GPU")
static_assert(sizeof(hipIpcMemHandle_t) <= JXLGPU_IPC_HANDLE_BYTES, "IPC handle larger than the ABI's byte array");

extern "C" int jxlgpu_device_alloc(jxlgpu_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out || !bytes) return JXLGPU_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(out, bytes));
    HIP_TRY(ctx, hipMemsetAsync(*out, 0, bytes, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JXLGPU_OK;
}

extern "C" void jxlgpu_device_free(jxlgpu_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)jxlgpu_synchronize(ctx);
    }
    (void)hipFree(p);
}

extern "C" int jxlgpu_ipc_export(jxlgpu_ctx* ctx, void* dev_ptr, uint8_t handle[JXLGPU_IPC_HANDLE_BYTES]) {
    if (!ctx || !dev_ptr || !handle) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    HIP_TRY(ctx, hipIpcGetMemHandle(&h, dev_ptr));
    memset(handle, 0, JXLGPU_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return JXLGPU_OK;
}

extern "C" int jxlgpu_ipc_open(jxlgpu_ctx* ctx, const uint8_t handle[JXLGPU_IPC_HANDLE_BYTES], void** dev_ptr) {
    if (!ctx || !handle || !dev_ptr) return JXLGPU_ERR_INVALID_ARG;
    *dev_ptr = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    HIP_TRY(ctx, hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return JXLGPU_OK;
}

extern "C" int jxlgpu_ipc_close(jxlgpu_ctx* ctx, void* dev_ptr) {
    if (!ctx || !dev_ptr) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // stores to the peer must have left before the mapping goes
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_down));
    HIP_TRY(ctx, hipIpcCloseMemHandle(dev_ptr));
    return JXLGPU_OK;
}

extern "C" int jxlgpu_device_download(jxlgpu_ctx* ctx, const void* dev_ptr, void* host, size_t bytes) {
    if (!ctx || !dev_ptr || !host) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = jxlgpu_synchronize(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(host, dev_ptr, bytes, hipMemcpyDeviceToHost));
    return JXLGPU_OK;
}

extern "C" int jxlgpu_device_upload(jxlgpu_ctx* ctx, void* dev_ptr, const void* host, size_t bytes) {
    if (!ctx || !dev_ptr || !host) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(dev_ptr, host, bytes, hipMemcpyHostToDevice));
    return JXLGPU_OK;
}
