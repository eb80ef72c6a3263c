// Stage-at-a-time restoration / colour / upsampling kernels (one launch per stage, planes in
// HBM).  These are the simple forms used for stage-level parity tests and for configurations the
// fused tile kernel (fused_kernels.hip) does not cover; the fused kernel is the fast path.
#include <cmath>

#include "common.h"
#include "pixel_device.h"

// ---------------------------------------------------------------- F1 Gabor-like
__global__ __launch_bounds__(256) void gabor_kernel(FilterArgs a) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int c = blockIdx.z;
    if (x >= (int)a.width || y >= (int)a.height) return;
    const float* in = a.in[c];
    auto at = [&](int dx, int dy) { return in[(size_t)(y + dy) * a.in_stride + (x + dx)]; };
    a.out[c][(size_t)y * a.out_stride + x] =
        gabor_sample(at, x, y, (int)a.width, (int)a.height, a.fp.gab_weights[c][0], a.fp.gab_weights[c][1]);
}

void launch_gabor(hipStream_t s, const FilterArgs& a) {
    dim3 grid(ceil_div(a.width, 64), ceil_div(a.height, 4), 3);
    gabor_kernel<<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------- F2 EPF, one step
template <int STEP>
__global__ __launch_bounds__(256) void epf_kernel(FilterArgs a) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= (int)a.width || y >= (int)a.height) return;
    const int W = (int)a.width, H = (int)a.height;
    float sigma_val = a.sigma[(size_t)(y >> 3) * a.sigma_stride + (x >> 3)];
    size_t o = (size_t)y * a.out_stride + x;
    if (sigma_val < 0.3f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[c][o] = a.in[c][(size_t)y * a.in_stride + x];
        return;
    }
    float step_multiplier = STEP == 0 ? a.fp.epf_pass0_sigma_scale : STEP == 2 ? a.fp.epf_pass2_sigma_scale : 1.0f;
    float sm = epf_step_mul(x, y, step_multiplier, a.fp.epf_border_sad_mul);
    auto at = [&](int c, int dx, int dy) {
        int xx = mirror_idx(x + dx, W), yy = mirror_idx(y + dy, H);
        return a.in[c][(size_t)yy * a.in_stride + xx];
    };
    float out[3];
    epf_pixel<STEP>(at, sigma_val, sm, a.fp.epf_channel_scale, out);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[c][o] = out[c];
}

void launch_epf(hipStream_t s, int step, const FilterArgs& a) {
    dim3 grid(ceil_div(a.width, 64), ceil_div(a.height, 4));
    if (step == 0) epf_kernel<0><<<grid, 256, 0, s>>>(a);
    else if (step == 1) epf_kernel<1><<<grid, 256, 0, s>>>(a);
    else epf_kernel<2><<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------- C1-C4 colour, in place
// FULL: with the HLG ops (color_pixel_t<true>; ColorArgs::staged_only frames), the only kernel built with them
template <bool FULL>
__global__ __launch_bounds__(256) void color_kernel(ColorArgs cp, float* p0, float* p1, float* p2,
                                                    uint32_t stride, uint32_t width, uint32_t height) {
    uint32_t x = blockIdx.x * 256 + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= width) return;
    size_t i = (size_t)y * stride + x;
    float v[3] = {p0[i], p1[i], p2[i]};
    color_pixel_t<FULL>(cp, v);
    p0[i] = v[0];
    p1[i] = v[1];
    p2[i] = v[2];
}

void launch_color(hipStream_t s, const ColorArgs& c, float* const planes[3], uint32_t stride,
                  uint32_t width, uint32_t height) {
    dim3 grid(ceil_div(width, 256), height);
    if (c.staged_only) color_kernel<true><<<grid, 256, 0, s>>>(c, planes[0], planes[1], planes[2], stride, width, height);
    else color_kernel<false><<<grid, 256, 0, s>>>(c, planes[0], planes[1], planes[2], stride, width, height);
}

// ---------------------------------------------------------------- device self-test of libm_f32.h
namespace {
__global__ __launch_bounds__(256) void libm_selftest_kernel(int which, const float* __restrict__ x, size_t n, float y,
                                                            float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = which == 0 ? libm_f32::logf(x[i]) : libm_f32::powf(x[i], y);
}
}  // namespace

extern "C" int jxlgpu_selftest_libm(jxlgpu_ctx* ctx, int which, const float* x, size_t n, float y, float* out) {
    if (!ctx) return JXLGPU_ERR_INVALID_ARG;
    if ((which != 0 && which != 1) || (n && (!x || !out)) || !std::isfinite(y) || n > ((size_t)1 << 30)) {
        ctx->last_error = "jxlgpu_selftest_libm: which is 0 (logf) or 1 (powf), y finite, n <= 2^30";
        return JXLGPU_ERR_INVALID_ARG;
    }
    if (n == 0) return JXLGPU_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *dx = nullptr, *dout = nullptr;   // a diagnostic: plain allocations, outside the pool and the memory budget
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dx), n * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        libm_selftest_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(which, dx, n, y, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (dx) (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
    HIP_TRY(ctx, e);
    return JXLGPU_OK;
}

// ---------------------------------------------------------------- F3 non-separable upsampling
// upsample_inner<K> (jxl-render/src/features/upsampling.rs:45-132): 5x5 kernel per output phase,
// mirrored 2-px border (util.rs:423-454 == mirror() for dimensions >= 2), clamp to the min/max
// of the 25 inputs.  `kernels` = weights_quarter expanded on the host: (K/2)^2 x 25 floats.
template <int K>
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ in, uint32_t in_stride,
                                                       uint32_t w, uint32_t h, float* __restrict__ out,
                                                       uint32_t out_stride, const float* __restrict__ kernels,
                                                       uint32_t ox0, uint32_t oy0, uint32_t ox1, uint32_t oy1) {
    constexpr int MAT_N = K / 2;
    // output window [ox0, ox1) x [oy0, oy1) (region renders; the whole w*K x h*K plane otherwise)
    uint32_t x = ox0 + blockIdx.x * 64 + (threadIdx.x & 63);
    uint32_t y = oy0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ox1 || y >= oy1) return;
    int ref_x = x / K, ref_y = y / K;
    int xm = x % K, ym = y % K;
    int mat_x = min(xm, K - xm - 1), mat_y = min(ym, K - ym - 1);
    bool flip_h = xm >= MAT_N, flip_v = ym >= MAT_N;
    const float* kernel = kernels + (mat_y * MAT_N + mat_x) * 25;
    float sum = 0.0f, mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int iy = 0; iy < 5; ++iy) {
        int ky = flip_v ? 4 - iy : iy;
        int sy = mirror_idx(ref_y + iy - 2, (int)h);
#pragma unroll
        for (int ix = 0; ix < 5; ++ix) {
            int kx = flip_h ? 4 - ix : ix;
            int sx = mirror_idx(ref_x + ix - 2, (int)w);
            float sample = in[(size_t)sy * in_stride + sx];
            sum += kernel[ky * 5 + kx] * sample;
            mn = fminf(mn, sample);
            mx = fmaxf(mx, sample);
        }
    }
    float r;
    if (!isfinite(mn)) r = __builtin_nanf("");
    else {
        r = sum;
        if (r < mn) r = mn;
        if (r > mx) r = mx;
    }
    out[(size_t)y * out_stride + x] = r;
}

void launch_upsample(hipStream_t s, const float* in, uint32_t in_stride, uint32_t w, uint32_t h,
                     float* out, uint32_t out_stride, int k, const float* kernels, const PixRect* window) {
    // `window`: output samples to produce (null: all of them)
    const uint32_t ox0 = window ? (uint32_t)window->x0 : 0u, oy0 = window ? (uint32_t)window->y0 : 0u;
    const uint32_t ox1 = window ? (uint32_t)window->x1 : w * k, oy1 = window ? (uint32_t)window->y1 : h * k;
    if (ox1 <= ox0 || oy1 <= oy0) return;
    dim3 grid(ceil_div(ox1 - ox0, 64), ceil_div(oy1 - oy0, 4));
    if (k == 2) upsample_kernel<2><<<grid, 256, 0, s>>>(in, in_stride, w, h, out, out_stride, kernels, ox0, oy0, ox1, oy1);
    else if (k == 4) upsample_kernel<4><<<grid, 256, 0, s>>>(in, in_stride, w, h, out, out_stride, kernels, ox0, oy0, ox1, oy1);
    else upsample_kernel<8><<<grid, 256, 0, s>>>(in, in_stride, w, h, out, out_stride, kernels, ox0, oy0, ox1, oy1);
}
