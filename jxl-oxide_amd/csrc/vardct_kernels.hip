// VarDCT LF-stage kernels for gfx950: LF dequant + CfL-LF (V1, V2), adaptive LF smoothing (V3),
// and the fill of groups that carry no HfMetadata.  The varblock transform (V4-V8) is in
// transform_kernels.hip.
#include "common.h"

// ---------------------------------------------------------------- V1 + V2
// copy_lf_dequant (jxl-render/src/vardct/mod.rs:387-412) + chroma_from_luma_lf (:544-568).
__device__ __forceinline__ void lf_dequant_cfl_body(const LfArgs& a) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= a.w8 || y >= a.h8) return;
    uint32_t g = (y / a.group_cells_y) * a.lf_groups_per_row + x / a.group_cells_x;
    size_t i = (size_t)y * a.w8 + x;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int32_t q = a.is_i16 ? (int32_t)((const int16_t*)a.lfq[c])[i] : ((const int32_t*)a.lfq[c])[i];
        v[c] = (float)q * a.scale[g * 3 + c];
    }
    float yy = v[1];
    v[0] += a.kx * yy;
    v[2] += a.kb * yy;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[c][i] = v[c];
}

__global__ __launch_bounds__(256) void lf_dequant_cfl_kernel(LfArgs a) { lf_dequant_cfl_body(a); }
__global__ __launch_bounds__(256) void lf_dequant_cfl_batch_kernel(FrameBatch b) {
    JXL_SET_TR_PRIO();
    const LfArgs a = load_const(&((FrameDevC)b.f[blockIdx.z])->lf);
    lf_dequant_cfl_body(a);
}

void launch_lf_dequant_cfl(hipStream_t s, const LfArgs& a) {
    dim3 grid(ceil_div(a.w8, 256), a.h8);
    lf_dequant_cfl_kernel<<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------- V3
// adaptive_lf_smoothing_impl (jxl-render/src/vardct/generic/mod.rs:11-103).  The CPU code runs
// in place but only ever reads unsmoothed neighbours (udsum scratch + `prev` carry), so it is an
// out-of-place 3x3 stencil; border samples are copied.
__device__ __forceinline__ void lf_smooth_body(const SmoothArgs& a) {
    const float SCALE_SELF = 0.052262735f, SCALE_SIDE = 0.2034514f, SCALE_DIAG = 0.03348292f;
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= a.w8 || y >= a.h8) return;
    size_t w = a.w8;
    size_t i = (size_t)y * w + x;
    bool interior = a.w8 > 2 && a.h8 > 2 && x >= 1 && x + 1 < a.w8 && y >= 1 && y + 1 < a.h8;
    if (!interior) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[c][i] = a.in[c][i];
        return;
    }
    float self[3], wa[3], gap = 0.5f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = a.in[c];
        self[c] = p[i];
        float ud_c = p[i - w] + p[i + w];             // udsum[x]
        float ud_l = p[i - w - 1] + p[i + w - 1];     // udsum[x-1]
        float ud_r = p[i - w + 1] + p[i + w + 1];     // udsum[x+1]
        float side = p[i - 1] + p[i + 1] + ud_c;
        float diag = ud_l + ud_r;
        wa[c] = self[c] * SCALE_SELF + side * SCALE_SIDE + diag * SCALE_DIAG;
        float gap_t = fabsf(wa[c] - self[c]) / a.lf_div[c];
        gap = fmaxf(gap, gap_t);
    }
    float gap_scale = fmaxf(3.0f - 4.0f * gap, 0.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[c][i] = (wa[c] - self[c]) * gap_scale + self[c];
}

__global__ __launch_bounds__(256) void lf_smooth_kernel(SmoothArgs a) { lf_smooth_body(a); }
__global__ __launch_bounds__(256) void lf_smooth_batch_kernel(FrameBatch b) {
    JXL_SET_TR_PRIO();
    const FrameDevC fd = (FrameDevC)b.f[blockIdx.z];
    if (fd->skip_smooth) return;
    const SmoothArgs a = load_const(&fd->smooth);
    lf_smooth_body(a);
}

hipError_t launch_lf_batch(hipStream_t s, const FrameBatch& b, uint32_t n, uint32_t max_w8, uint32_t max_h8, bool any_smooth) {
    const dim3 grid(ceil_div(max_w8, 256), max_h8, n);
    lf_dequant_cfl_batch_kernel<<<grid, 256, 0, s>>>(b);
    if (any_smooth) lf_smooth_batch_kernel<<<grid, 256, 0, s>>>(b);
    return hipGetLastError();
}

void launch_lf_smooth(hipStream_t s, const SmoothArgs& a) {
    dim3 grid(ceil_div(a.w8, 256), a.h8);
    lf_smooth_kernel<<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------- groups without HfMetadata
// transform_with_lf_grouped, mod.rs:655-665: replicate each LF sample over its 8x8 cell.
__global__ __launch_bounds__(256) void nometa_kernel(TransformArgs a, const uint32_t* groups,
                                                     uint32_t group_dim, uint32_t groups_per_row) {
    uint32_t g = groups[blockIdx.y];
    uint32_t gx = g % groups_per_row, gy = g / groups_per_row;
    uint32_t row = blockIdx.x;  // row inside the group
    uint32_t py = gy * group_dim + row;
    if (py >= a.h8 * 8) return;
    for (uint32_t x = threadIdx.x; x < group_dim; x += 256) {
        uint32_t px = gx * group_dim + x;
        if (px >= a.w8 * 8) break;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            a.pix[coeff_tiled_index(px, py, c, a.w8)] = a.lf[c][(size_t)(py / 8) * a.w8 + px / 8];
    }
}

void launch_nometa_groups(hipStream_t s, const TransformArgs& a, const uint32_t* groups,
                          uint32_t count, uint32_t group_dim, uint32_t groups_per_row) {
    if (!count) return;
    nometa_kernel<<<dim3(group_dim, count), 256, 0, s>>>(a, groups, group_dim, groups_per_row);
}

// ---------------------------------------------------------------- tiled -> row-major planes
// Only for renders that stop after the transform (tests, `stages` without a post stage) and for
// the one-kernel-per-stage fallback: the fused post kernels read the tiled layout directly.
__global__ __launch_bounds__(256) void untile_kernel(const float* __restrict__ tiled, uint32_t w8, float* o0, float* o1,
                                                     float* o2, uint32_t out_stride, uint32_t width, uint32_t height) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    float* const out[3] = {o0, o1, o2};
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) out[c][(size_t)y * out_stride + x] = tiled[coeff_tiled_index(x, y, c, w8)];
}

void launch_untile(hipStream_t s, const float* tiled, uint32_t w8, float* const out[3], uint32_t out_stride,
                   uint32_t width, uint32_t height) {
    untile_kernel<<<dim3(ceil_div(width, 256), height), 256, 0, s>>>(tiled, w8, out[0], out[1], out[2], out_stride, width, height);
}
