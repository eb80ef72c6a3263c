// Chroma upsampling of JPEG-transcoded frames: ImageWithRegion::upsample_jpeg
// (jxl-render/src/image.rs:448-485) -> apply_jpeg_upsampling_single (filter/ycbcr.rs:6-89).
// The reference upsamples horizontally into the output buffer, then vertically in place from the
// bottom row up; every output sample is a function of at most four input samples, evaluated here
// per output pixel with the same two-step f32 arithmetic:
//   h:  out[2i]   = 0.25 * in[i-1] + 0.75 * in[i]      out[2i+1] = 0.75 * in[i] + 0.25 * in[i+1]
//   v:  out[2y]   = 0.75 * t[y] + 0.25 * t[y-1]        out[2y+1] = 0.25 * t[y+1] + 0.75 * t[y]
// with the edge sample replicated.  Built with -ffp-contract=off (no fused multiply-add).
#include "common.h"

namespace {

struct UpJpegArgs {
    const float* in;   // the child's transform output: cell-tiled (coeff_tiled_index), channel `c`
    float* out;
    uint32_t in_w8, c, in_w, in_h, out_stride, width, height;
    int hshift, vshift;
    uint32_t in_row_stride;   // > 0: `in` is a row-major plane of this stride (Modular frames) instead of the tiled transform output
};

__device__ __forceinline__ float h_value(const UpJpegArgs& a, uint32_t x, uint32_t row) {
    auto r = [&](uint32_t xi) {
        return a.in_row_stride ? a.in[(size_t)row * a.in_row_stride + xi] : a.in[coeff_tiled_index(xi, row, a.c, a.in_w8)];
    };
    if (!a.hshift) return r(x);
    const uint32_t i = x >> 1;
    const float curr = r(i);
    if ((x & 1) == 0) {
        const float prev = i > 0 ? r(i - 1) : r(0);
        return 0.25f * prev + 0.75f * curr;
    }
    const float next = i + 1 < a.in_w ? r(i + 1) : r(a.in_w - 1);
    return 0.75f * curr + 0.25f * next;
}

__global__ __launch_bounds__(256) void upsample_jpeg_kernel(UpJpegArgs a) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.width) return;
    float v;
    if (!a.vshift) {
        v = h_value(a, x, y);
    } else {
        const uint32_t yy = y >> 1;
        const float curr = h_value(a, x, yy);
        if ((y & 1) == 0) {
            const float above = yy > 0 ? h_value(a, x, yy - 1) : curr;
            v = 0.75f * curr + 0.25f * above;
        } else {
            const float below = yy + 1 < a.in_h ? h_value(a, x, yy + 1) : curr;
            v = 0.25f * below + 0.75f * curr;
        }
    }
    a.out[(size_t)y * a.out_stride + x] = v;
}

}  // namespace

void launch_upsample_jpeg(hipStream_t s, const float* in_tiled, uint32_t in_w8, uint32_t c, uint32_t in_w, uint32_t in_h,
                          int hshift, int vshift, float* out, uint32_t out_stride, uint32_t width, uint32_t height) {
    UpJpegArgs a{in_tiled, out, in_w8, c, in_w, in_h, out_stride, width, height, hshift, vshift, 0u};
    hipLaunchKernelGGL(upsample_jpeg_kernel, dim3((width + 255) / 256, height), dim3(256), 0, s, a);
}

// The same for a row-major input plane (chroma-subsampled Modular frames).
void launch_upsample_jpeg_rows(hipStream_t s, const float* in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, int hshift, int vshift,
                               float* out, uint32_t out_stride, uint32_t width, uint32_t height) {
    UpJpegArgs a{in, out, 0u, 0u, in_w, in_h, out_stride, width, height, hshift, vshift, in_stride};
    hipLaunchKernelGGL(upsample_jpeg_kernel, dim3((width + 255) / 256, height), dim3(256), 0, s, a);
}
