// Compact HF-coefficient transport: the host side of `write_hf_coeff`
// (jxl-vardct/src/hf_coeff.rs:207-244) may hand the decoded coefficients over as 16-bit planes or
// as (position, value) lists instead of dense i32 planes; these kernels rebuild the dense i32
// planes the transform kernels read.  Integer only, so the rebuilt planes are identical to what
// `*coeff_grid.get_mut(x, y) += coeff` (hf_coeff.rs:234) leaves in the reference's framebuffer.
#include "common.h"

namespace {

// One thread widens 8 samples (one 16-byte load, two 16-byte stores).  Planes are wr x hr with wr
// a multiple of 8 and hipMalloc alignment, so the vector accesses are aligned.
__global__ __launch_bounds__(256) void widen_i16_kernel(const int16_t* __restrict__ src, int32_t* __restrict__ dst,
                                                        size_t n8) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n8; i += step) {
        const int4 v = reinterpret_cast<const int4*>(src)[i];
        int4 lo, hi;
        lo.x = (int16_t)(v.x & 0xffff); lo.y = v.x >> 16;
        lo.z = (int16_t)(v.y & 0xffff); lo.w = v.y >> 16;
        hi.x = (int16_t)(v.z & 0xffff); hi.y = v.z >> 16;
        hi.z = (int16_t)(v.w & 0xffff); hi.w = v.w >> 16;
        reinterpret_cast<int4*>(dst)[2 * i] = lo;
        reinterpret_cast<int4*>(dst)[2 * i + 1] = hi;
    }
}

// Sparse lists accumulate (`+=`, hf_coeff.rs:234: later passes add shifted refinements at the
// same position); integer atomics make the sum independent of the order entries arrive in.
template <typename V>
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t* __restrict__ pos, const V* __restrict__ val,
                                                      size_t count, uint32_t src_stride, uint32_t wr, uint32_t hr,
                                                      int32_t* __restrict__ dst, uint32_t* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < count; i += step) {
        const uint32_t p = pos[i];
        const uint32_t y = p / src_stride, x = p - y * src_stride;
        if (x >= wr || y >= hr) {
            atomicAdd(bad, 1u);
            continue;
        }
        atomicAdd(&dst[(size_t)y * wr + x], (int32_t)val[i]);
    }
}

uint32_t grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    const size_t cap = 256 * 32;  // grid-stride beyond 32 workgroups per CU
    return (uint32_t)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

void launch_widen_i16(hipStream_t s, const int16_t* src, int32_t* dst, size_t count) {
    const size_t n8 = count / 8;  // count = wr * hr, a multiple of 64
    hipLaunchKernelGGL(widen_i16_kernel, dim3(grid_for(n8)), dim3(256), 0, s, src, dst, n8);
}

void launch_coeff_scatter(hipStream_t s, const uint32_t* pos, const void* val, bool val_i16, size_t count,
                          uint32_t src_stride, uint32_t wr, uint32_t hr, int32_t* dst, uint32_t* bad) {
    if (count == 0) return;
    if (val_i16)
        hipLaunchKernelGGL(scatter_kernel<int16_t>, dim3(grid_for(count)), dim3(256), 0, s, pos,
                           static_cast<const int16_t*>(val), count, src_stride, wr, hr, dst, bad);
    else
        hipLaunchKernelGGL(scatter_kernel<int32_t>, dim3(grid_for(count)), dim3(256), 0, s, pos,
                           static_cast<const int32_t*>(val), count, src_stride, wr, hr, dst, bad);
}
