// Compact HF-coefficient transport: the host side of `write_hf_coeff`
// (jxl-vardct/src/hf_coeff.rs:207-244) may hand the decoded coefficients over as 16-bit planes or
// as (position, value) lists instead of dense i32 planes; these kernels build the device layout the
// transform kernels read (8x8 cells, channel-interleaved: coeff_tiled_index) from any of them.
// Integer only, so the values are identical to what `*coeff_grid.get_mut(x, y) += coeff`
// (hf_coeff.rs:234) leaves in the reference's framebuffer.
#include "common.h"

namespace {

// Row-major staging plane (i32 or i16, wr x hr, tight) of channel c -> the cell-tiled, channel-
// interleaved device layout (coeff_tiled_index).  One thread moves one row of one 8x8 cell:
// 32 (16) contiguous bytes in, 32 contiguous bytes out.
template <typename V>
__global__ __launch_bounds__(256) void retile_kernel(const V* __restrict__ src, int32_t* __restrict__ dst, uint32_t w8,
                                                     uint32_t hr, uint32_t c) {
    const size_t n = (size_t)w8 * hr;  // cell rows
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        const uint32_t py = (uint32_t)(i / w8), cx = (uint32_t)(i - (size_t)py * w8);
        int32_t v[8];
        if constexpr (sizeof(V) == 4) {
            const int4 lo = reinterpret_cast<const int4*>(src)[i * 2], hi = reinterpret_cast<const int4*>(src)[i * 2 + 1];
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        } else {
            const int4 p = reinterpret_cast<const int4*>(src)[i];
            v[0] = (int16_t)(p.x & 0xffff); v[1] = p.x >> 16; v[2] = (int16_t)(p.y & 0xffff); v[3] = p.y >> 16;
            v[4] = (int16_t)(p.z & 0xffff); v[5] = p.z >> 16; v[6] = (int16_t)(p.w & 0xffff); v[7] = p.w >> 16;
        }
        int32_t* out = dst + coeff_tiled_index(cx * 8, py, c, w8);
        reinterpret_cast<int4*>(out)[0] = make_int4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<int4*>(out)[1] = make_int4(v[4], v[5], v[6], v[7]);
    }
}

// Sparse lists accumulate (`+=`, hf_coeff.rs:234: later passes add shifted refinements at the
// same position); integer atomics make the sum independent of the order entries arrive in.
template <typename V>
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t* __restrict__ pos, const V* __restrict__ val,
                                                      size_t count, uint32_t src_stride, uint32_t wr, uint32_t hr,
                                                      uint32_t c, int32_t* __restrict__ dst, uint32_t* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < count; i += step) {
        const uint32_t p = pos[i];
        const uint32_t y = p / src_stride, x = p - y * src_stride;
        if (x >= wr || y >= hr) {
            atomicAdd(bad, 1u);
            continue;
        }
        atomicAdd(&dst[coeff_tiled_index(x, y, c, wr / 8)], (int32_t)val[i]);
    }
}

uint32_t grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    const size_t cap = 256 * 32;  // grid-stride beyond 32 workgroups per CU
    return (uint32_t)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

void launch_coeff_retile(hipStream_t s, const void* src, bool src_i16, uint32_t wr, uint32_t hr, uint32_t c,
                         int32_t* dst) {
    const size_t n = (size_t)(wr / 8) * hr;
    if (src_i16)
        hipLaunchKernelGGL(retile_kernel<int16_t>, dim3(grid_for(n)), dim3(256), 0, s, static_cast<const int16_t*>(src),
                           dst, wr / 8, hr, c);
    else
        hipLaunchKernelGGL(retile_kernel<int32_t>, dim3(grid_for(n)), dim3(256), 0, s, static_cast<const int32_t*>(src),
                           dst, wr / 8, hr, c);
}

void launch_coeff_scatter(hipStream_t s, const uint32_t* pos, const void* val, bool val_i16, size_t count,
                          uint32_t src_stride, uint32_t wr, uint32_t hr, uint32_t c, int32_t* dst, uint32_t* bad) {
    if (count == 0) return;
    if (val_i16)
        hipLaunchKernelGGL(scatter_kernel<int16_t>, dim3(grid_for(count)), dim3(256), 0, s, pos,
                           static_cast<const int16_t*>(val), count, src_stride, wr, hr, c, dst, bad);
    else
        hipLaunchKernelGGL(scatter_kernel<int32_t>, dim3(grid_for(count)), dim3(256), 0, s, pos,
                           static_cast<const int32_t*>(val), count, src_stride, wr, hr, c, dst, bad);
}
