// Output formatting on the device: planar f32 render result -> interleaved, oriented
// f32 / u16 / u8 samples (ImageStream::write_to_buffer, jxl-oxide/src/fb.rs:309-397; sample
// conversion fb.rs:487-490, 524-527: `(v * max + 0.5).clamp(0, max) as uN`).  One lane per output
// pixel (3 samples); the oriented read goes through L2, the interleaved write is contiguous.
#include "common.h"

namespace {

struct FormatArgs {
    const float* in[3 + 4];   // colour planes, then the selected extra channels
    void* out;
    uint32_t in_stride, ow, oh, orientation;
    uint32_t nch;             // 3 + number of extra channels
    uint32_t ex_stride[4];    // row strides of the extra planes (tight: their width)
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int FMT>
__global__ __launch_bounds__(256) void format_kernel(FormatArgs a) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.ow) return;
    uint32_t ox, oy;  // to_original_coord, fb.rs:383-397
    switch (a.orientation) {
        case 1: ox = x; oy = y; break;
        case 2: ox = a.ow - x - 1; oy = y; break;
        case 3: ox = a.ow - x - 1; oy = a.oh - y - 1; break;
        case 4: ox = x; oy = a.oh - y - 1; break;
        case 5: ox = y; oy = x; break;
        case 6: ox = y; oy = a.ow - x - 1; break;
        case 7: ox = a.oh - y - 1; oy = a.ow - x - 1; break;
        default: ox = a.oh - y - 1; oy = x; break;
    }
    const size_t gi = (size_t)oy * a.in_stride + ox;
    const size_t o = ((size_t)y * a.ow + x) * a.nch;
    for (uint32_t c = 0; c < a.nch; ++c) {
        float v = c < 3 ? a.in[c][gi] : a.in[c][(size_t)oy * a.ex_stride[c - 3] + ox];
        if (FMT == JXLGPU_FMT_F32) {
            ((float*)a.out)[o + c] = v;
        } else if (FMT == JXLGPU_FMT_U16) {
            float t = clampf(v * 65535.0f + 0.5f, 0.0f, 65535.0f);
            ((uint16_t*)a.out)[o + c] = t != t ? (uint16_t)0 : (uint16_t)t;   // Rust `as u16`: NaN -> 0, truncation
        } else {
            float t = clampf(v * 255.0f + 0.5f, 0.0f, 255.0f);
            ((uint8_t*)a.out)[o + c] = t != t ? (uint8_t)0 : (uint8_t)t;
        }
    }
}

// u8, orientation 1, width a multiple of 4, 16-byte aligned rows: four pixels per lane — three 16-byte
// loads, 12 bytes out as three dwords (the one-pixel form writes single bytes: a quarter of a dword
// per store).  Same conversion per sample.
__global__ __launch_bounds__(256) void format_u8_x4_kernel(FormatArgs a) {
    const uint32_t x4 = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x4 * 4 >= a.ow) return;
    const size_t gi = (size_t)y * a.in_stride + x4 * 4;
    float v[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(a.in[c] + gi);
        v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
    }
    uint32_t b[12];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = clampf(v[c][p] * 255.0f + 0.5f, 0.0f, 255.0f);
            b[p * 3 + c] = t != t ? 0u : (uint32_t)(uint8_t)t;
        }
    uint32_t* o = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(a.out) + ((size_t)y * a.ow + x4 * 4) * 3);
#pragma unroll
    for (int w = 0; w < 3; ++w) o[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
}

}  // namespace

extern "C" int jxlgpu_frame_format_output(jxlgpu_ctx* ctx, jxlgpu_frame* f, const JxlGpuFormatDesc* fmt, void* out,
                                          uint32_t out_mem, uint32_t* out_w, uint32_t* out_h) {
    if (!ctx || !f || !fmt || !out || out_mem > JXLGPU_MEM_HOST_PINNED) return JXLGPU_ERR_INVALID_ARG;
    if (fmt->orientation < 1 || fmt->orientation > 8 || fmt->sample_format > JXLGPU_FMT_U8 || fmt->num_extra > 4) {
        ctx->last_error = "bad orientation / sample format / more than 4 extra channels";
        return JXLGPU_ERR_INVALID_ARG;
    }
    if (!f->result[0]) {
        ctx->last_error = "format_output needs a completed render on this frame";
        return JXLGPU_ERR_INVALID_ARG;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t w = f->result_w, h = f->result_h;
    const uint32_t ow = fmt->orientation <= 4 ? w : h, oh = fmt->orientation <= 4 ? h : w;
    const size_t esz = fmt->sample_format == JXLGPU_FMT_F32 ? 4 : fmt->sample_format == JXLGPU_FMT_U16 ? 2 : 1;
    const size_t bytes = (size_t)ow * oh * (3 + fmt->num_extra) * esz;
    void* dst = out;
    if (out_mem != JXLGPU_MEM_DEVICE) {
        if (f->fmt_bytes < bytes) {
            void* p = nullptr;
            HIP_TRY(ctx, ctx_dev_malloc(ctx, &p, bytes));
            f->allocs.push_back(p);
            f->fmt_buf = p;
            f->fmt_bytes = bytes;
        }
        dst = f->fmt_buf;
        // an earlier JXLGPU_MEM_HOST_PINNED output of this frame may still be reading fmt_buf on the download stream
        if (f->ev_fmt_set) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, f->ev_fmt, 0));
    }
    FormatArgs a;
    memset(&a, 0, sizeof(a));
    for (int c = 0; c < 3; ++c) a.in[c] = f->result[c];
    a.nch = 3 + fmt->num_extra;
    for (uint32_t i = 0; i < fmt->num_extra; ++i) {
        const uint32_t e = fmt->extra[i];
        if (e >= JXLGPU_MAX_EXTRA || !f->extra[e] || f->extra_w[e] != f->result_w || f->extra_h[e] != f->result_h) {
            ctx->last_error = "format_output: extra channel not rendered, or not of the size of the colour result";
            return JXLGPU_ERR_INVALID_ARG;
        }
        a.in[3 + i] = f->extra[e];
        a.ex_stride[i] = f->extra_w[e];
    }
    a.out = dst; a.in_stride = f->result_stride; a.ow = ow; a.oh = oh; a.orientation = fmt->orientation;
    dim3 grid(ceil_div(ow, 256), oh);
    const bool x4 = fmt->num_extra == 0 && fmt->sample_format == JXLGPU_FMT_U8 && fmt->orientation == 1 && (ow & 3) == 0 && (a.in_stride & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(dst) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in[0]) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.in[1]) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.in[2]) & 15) == 0;
    if (x4) format_u8_x4_kernel<<<dim3(ceil_div(ow / 4, 256), oh), 256, 0, ctx->stream>>>(a);
    else if (fmt->sample_format == JXLGPU_FMT_F32) format_kernel<JXLGPU_FMT_F32><<<grid, 256, 0, ctx->stream>>>(a);
    else if (fmt->sample_format == JXLGPU_FMT_U16) format_kernel<JXLGPU_FMT_U16><<<grid, 256, 0, ctx->stream>>>(a);
    else format_kernel<JXLGPU_FMT_U8><<<grid, 256, 0, ctx->stream>>>(a);
    HIP_TRY(ctx, hipGetLastError());
    if (out_mem == JXLGPU_MEM_HOST_PINNED) {
        // asynchronous: the copy runs on the download stream behind the formatting kernel and the call returns;
        // jxlgpu_frame_wait (or jxlgpu_synchronize) tells when `out` is complete.  The kernels of the next frames
        // do not wait for it.
        hipEvent_t ev = nullptr;
        if (!f->ev_last) HIP_TRY(ctx, hipEventCreateWithFlags(&f->ev_last, hipEventDisableTiming));
        ev = f->ev_last;
        HIP_TRY(ctx, hipEventRecord(ev, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_down, ev, 0));
        HIP_TRY(ctx, hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, ctx->stream_down));
        if (!f->ev_fmt) HIP_TRY(ctx, hipEventCreateWithFlags(&f->ev_fmt, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(f->ev_fmt, ctx->stream_down));
        f->ev_fmt_set = true;
        frame_mark(ctx, f, ctx->stream_down);
    } else if (out_mem != JXLGPU_MEM_DEVICE) {
        if (ctx->pinned_size < bytes) {
            if (ctx->pinned) (void)hipHostFree(ctx->pinned);
            ctx->pinned = nullptr; ctx->pinned_size = 0;
            HIP_TRY(ctx, hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
            ctx->pinned_size = bytes;
        }
        // D2H in slices: the ctx's worker threads move slice k into the caller's (pageable) buffer while slice
        // k + 1 crosses PCIe
        const size_t n_slices = bytes >= ((size_t)8 << 20) ? 4 : 1;
        auto cut = [&](size_t k) { return k >= n_slices ? bytes : bytes * k / n_slices / 4096 * 4096; };
        for (size_t k = 0; k < n_slices; ++k) {
            if (!ctx->ev_slice[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_slice[k], hipEventDisableTiming));
            HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->pinned + cut(k), (const char*)dst + cut(k), cut(k + 1) - cut(k),
                                        hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_slice[k], ctx->stream));
        }
        for (size_t k = 0; k < n_slices; ++k) {
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev_slice[k]));
            const size_t b0 = cut(k), nb = cut(k + 1) - b0;
            const uint32_t parts = nb >= ((size_t)1 << 20) ? 8 : 1;
            ctx_host_parallel(ctx, parts, [&](uint32_t i) {
                const size_t p0 = nb * i / parts, p1 = nb * (i + 1) / parts;
                memcpy((char*)out + b0 + p0, (const char*)ctx->pinned + b0 + p0, p1 - p0);
            });
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (out_w) *out_w = ow;
    if (out_h) *out_h = oh;
    return JXLGPU_OK;
}
