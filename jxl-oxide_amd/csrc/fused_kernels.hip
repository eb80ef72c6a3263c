// Fused restoration + colour kernel: Gabor-like -> EPF (1-3 steps) -> XYB->display in ONE pass
// over HBM.  The CPU reference makes a full-image pass per stage (render.rs:76-131 +
// lib.rs:925-998: 5 read+write sweeps at iters=2); here a 256-thread workgroup owns a 32x32
// output tile, stages the tile plus its halo (1 px Gabor + 2/1 px per EPF step, 3 for step 0) for
// all three channels in LDS, runs every stage tile-locally (recomputing the halo), and only the
// finished colour samples go back to HBM: 12 B/px in, 12 B/px out.
//
// Image borders: each stage of the reference mirrors ITS OWN input (util.rs:376-386), so after
// every stage the out-of-image cells of the stage's region are refilled from the mirrored
// in-image cells before the next stage reads them.
#include "common.h"
#include "pixel_device.h"

namespace {

constexpr int T = 32;  // output tile

template <bool GAB, int ITERS>
struct PostCfg {
    static constexpr int R_GAB = GAB ? 1 : 0;
    static constexpr int R_E0 = ITERS == 3 ? 3 : 0;
    static constexpr int R_E1 = ITERS >= 1 ? 2 : 0;
    static constexpr int R_E2 = ITERS >= 2 ? 1 : 0;
    static constexpr int HALO = R_GAB + R_E0 + R_E1 + R_E2;
    static constexpr int LW = T + 2 * HALO;       // LDS plane width/height
    static constexpr int PLANE = LW * LW;
};

struct FusedArgs {
    const float* in[3];
    float* out[3];
    uint32_t in_stride, out_stride;
    int width, height;
    const float* sigma;
    uint32_t sigma_stride;
    JxlGpuFilterParams fp;
    ColorArgs color;
    uint32_t do_color;
};

// Refill the out-of-image cells of the square region [lo, LW-lo) of `buf` (3 planes) from their
// mirrored in-image cells.  (ox, oy) = image coordinate of LDS cell (0, 0).
template <int LW>
__device__ __forceinline__ void mirror_fill(float* buf, int lo, int ox, int oy, int width, int height, int t) {
    const int n = LW - 2 * lo;
    for (int i = t; i < n * n; i += 256) {
        int ly = lo + i / n, lx = lo + i % n;
        int x = ox + lx, y = oy + ly;
        if (x >= 0 && x < width && y >= 0 && y < height) continue;
        int sx = mirror_idx(x, width) - ox, sy = mirror_idx(y, height) - oy;
        // cells whose mirror source lies outside the region are beyond the reach of later stages
        if (sx < lo || sx >= LW - lo || sy < lo || sy >= LW - lo) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c * LW * LW + ly * LW + lx] = buf[c * LW * LW + sy * LW + sx];
    }
}

template <int STEP, int LW>
__device__ __forceinline__ void epf_stage(const float* src, float* dst, int lo, int ox, int oy, int width,
                                          int height, const FusedArgs& a, int t, bool last, float (&res)[4][3]) {
    const int n = LW - 2 * lo;
    const float step_multiplier = STEP == 0 ? a.fp.epf_pass0_sigma_scale
                                : STEP == 2 ? a.fp.epf_pass2_sigma_scale : 1.0f;
    int k = 0;
    for (int i = t; i < n * n; i += 256, ++k) {
        int ly = lo + i / n, lx = lo + i % n;
        int x = ox + lx, y = oy + ly;
        if (x < 0 || x >= width || y < 0 || y >= height) continue;
        const float* p = src + ly * LW + lx;
        float o[3];
        float sigma_val = a.sigma[(size_t)(y >> 3) * a.sigma_stride + (x >> 3)];
        if (sigma_val < 0.3f) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = p[c * LW * LW];
        } else {
            float sm = epf_step_mul(x, y, step_multiplier, a.fp.epf_border_sad_mul);
            auto at = [&](int c, int dx, int dy) { return p[c * LW * LW + dy * LW + dx]; };
            epf_pixel<STEP>(at, sigma_val, sm, a.fp.epf_channel_scale, o);
        }
        if (last) {
#pragma unroll
            for (int c = 0; c < 3; ++c) res[k][c] = o[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c * LW * LW + ly * LW + lx] = o[c];
        }
    }
}

template <bool GAB, int ITERS>
__global__ __launch_bounds__(256) void fused_post_kernel(FusedArgs a) {
    using Cfg = PostCfg<GAB, ITERS>;
    constexpr int LW = Cfg::LW, PLANE = Cfg::PLANE, HALO = Cfg::HALO;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = lds + 3 * PLANE;
    const int t = threadIdx.x;
    const int tx0 = blockIdx.x * T, ty0 = blockIdx.y * T;
    const int ox = tx0 - HALO, oy = ty0 - HALO;  // image coordinate of LDS cell (0,0)
    const int W = a.width, H = a.height;
    const bool border = ox < 0 || oy < 0 || ox + LW > W || oy + LW > H;

    // ---- load tile + halo (mirrored at the image border), 3 channels
    for (int i = t; i < PLANE; i += 256) {
        int ly = i / LW, lx = i % LW;
        int x = mirror_idx(ox + lx, W), y = mirror_idx(oy + ly, H);
        size_t g = (size_t)y * a.in_stride + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) bufA[c * PLANE + i] = a.in[c][g];
    }
    __syncthreads();

    float* src = bufA;
    float* dst = bufB;
    int lo = 0;  // the current stage's output region is [lo, LW - lo)^2
    float res[4][3];

    if constexpr (GAB) {
        lo += 1;
        const int n = LW - 2 * lo;
        constexpr bool last = ITERS == 0;
        int k = 0;
        for (int i = t; i < n * n; i += 256, ++k) {
            int ly = lo + i / n, lx = lo + i % n;
            int x = ox + lx, y = oy + ly;
            if (x < 0 || x >= W || y < 0 || y >= H) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* p = src + c * PLANE + ly * LW + lx;
                auto at = [&](int dx, int dy) { return p[dy * LW + dx]; };
                float v = gabor_sample(at, x, y, W, H, a.fp.gab_weights[c][0], a.fp.gab_weights[c][1]);
                if (last) res[k][c] = v;
                else dst[c * PLANE + ly * LW + lx] = v;
            }
        }
        if (!last) {
            __syncthreads();
            if (border) {
                mirror_fill<LW>(dst, lo, ox, oy, W, H, t);
                __syncthreads();
            }
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    if constexpr (ITERS == 3) {
        lo += 3;
        epf_stage<0, LW>(src, dst, lo, ox, oy, W, H, a, t, false, res);
        __syncthreads();
        if (border) {
            mirror_fill<LW>(dst, lo, ox, oy, W, H, t);
            __syncthreads();
        }
        float* tmp = src; src = dst; dst = tmp;
    }
    if constexpr (ITERS >= 1) {
        lo += 2;
        constexpr bool last = ITERS == 1;
        epf_stage<1, LW>(src, dst, lo, ox, oy, W, H, a, t, last, res);
        if (!last) {
            __syncthreads();
            if (border) {
                mirror_fill<LW>(dst, lo, ox, oy, W, H, t);
                __syncthreads();
            }
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    if constexpr (ITERS >= 2) {
        lo += 1;
        epf_stage<2, LW>(src, dst, lo, ox, oy, W, H, a, t, true, res);
    }
    if constexpr (!GAB && ITERS == 0) {
        // colour only: straight from the staged tile
        int k = 0;
        for (int i = t; i < T * T; i += 256, ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) res[k][c] = src[c * PLANE + (i / T) * LW + (i % T)];
    }

    // ---- final region is the T x T tile: colour + store
    static_assert(T * T == 4 * 256, "4 output samples per lane");
    int k = 0;
    for (int i = t; i < T * T; i += 256, ++k) {
        int x = tx0 + i % T, y = ty0 + i / T;
        if (x >= W || y >= H) continue;
        float v[3] = {res[k][0], res[k][1], res[k][2]};
        if (a.do_color) color_pixel(a.color, v);
        size_t g = (size_t)y * a.out_stride + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[c][g] = v[c];
    }
}

template <bool GAB, int ITERS>
void launch_cfg(hipStream_t s, const FusedArgs& a) {
    dim3 grid(ceil_div(a.width, T), ceil_div(a.height, T));
    constexpr size_t lds_bytes = 2 * 3 * PostCfg<GAB, ITERS>::PLANE * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_post_kernel<GAB, ITERS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    fused_post_kernel<GAB, ITERS><<<grid, 256, lds_bytes, s>>>(a);
}

}  // namespace

bool fused_post_supported(const jxlgpu_frame* f, bool gabor, int epf_iters) {
    if (getenv("JXLGPU_NO_FUSED")) return false;
    return true;
}

void launch_fused_post(hipStream_t s, jxlgpu_frame* f, const float* const in[3], uint32_t in_stride,
                       float* const out[3], uint32_t out_stride, bool gabor, int epf_iters, bool color) {
    FusedArgs a;
    for (int c = 0; c < 3; ++c) { a.in[c] = in[c]; a.out[c] = out[c]; }
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.width = (int)f->width; a.height = (int)f->height;
    a.sigma = f->sigma; a.sigma_stride = f->w8;
    a.fp = f->kind_of_frame == 0 ? f->desc.filter : f->desc.filter;
    a.color = f->color;
    a.do_color = color ? 1u : 0u;
    switch ((gabor ? 4 : 0) + epf_iters) {
        case 0: launch_cfg<false, 0>(s, a); break;
        case 1: launch_cfg<false, 1>(s, a); break;
        case 2: launch_cfg<false, 2>(s, a); break;
        case 3: launch_cfg<false, 3>(s, a); break;
        case 4: launch_cfg<true, 0>(s, a); break;
        case 5: launch_cfg<true, 1>(s, a); break;
        case 6: launch_cfg<true, 2>(s, a); break;
        default: launch_cfg<true, 3>(s, a); break;
    }
}
