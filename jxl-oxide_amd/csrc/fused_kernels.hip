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
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "pixel_device.h"

namespace {

constexpr int T = 32;  // output tile

template <bool GAB, int ITERS, int TW = T, int TH = T>
struct PostCfg {
    static constexpr int R_GAB = GAB ? 1 : 0;
    // ITERS: 0-3 = the frame's epf_iters (3: steps 0, 1, 2; 2: steps 1, 2; 1: step 1);
    //        4 = step 0 alone (first pass of the split iters-3 path, launch_fused_post)
    static constexpr bool HAS_E0 = ITERS == 3 || ITERS == 4, HAS_E1 = ITERS >= 1 && ITERS <= 3, HAS_E2 = ITERS == 2 || ITERS == 3;
    static constexpr int R_E0 = HAS_E0 ? 3 : 0;
    static constexpr int R_E1 = HAS_E1 ? 2 : 0;
    static constexpr int R_E2 = HAS_E2 ? 1 : 0;
    static constexpr int HALO = R_GAB + R_E0 + R_E1 + R_E2;
    static constexpr int LWX = TW + 2 * HALO, LWY = TH + 2 * HALO;  // LDS plane width / height
    static constexpr int PLANE = LWX * LWY;
};

// Border ring of the streaming path: strips of kRingT pixels, tiles of 32 x 16 along the top / bottom
// and 16 x 32 along the sides (same LDS plane size); the streaming kernels take everything inside.
constexpr int kRingT = 16, kRingL = 32;

// Modular XYB planes as the inverse transforms leave them (integers, channel order Y, X, B) -> the float sample of
// output channel c (X, Y, B): convert_to_float_modular_xyb, jxl-render/src/image.rs:148-189 — B is stored as B - Y,
// the sum saturates in the sample type, every channel is scaled by m_lf_unscaled.  `i`: element index of the sample.
__device__ __forceinline__ int32_t int_plane_at(const FusedArgs& a, int plane, size_t i) {
    return a.in_int == 1 ? (int32_t)reinterpret_cast<const int16_t*>(a.in[plane])[i] : reinterpret_cast<const int32_t*>(a.in[plane])[i];
}
__device__ __forceinline__ int32_t xyb_b_plus_y(uint32_t in_int, int32_t bv, int32_t yv) {
    if (in_int == 1) {
        const int32_t t = bv + yv;
        return t > 32767 ? 32767 : (t < -32768 ? -32768 : t);
    }
    const int64_t t = (int64_t)bv + yv;
    return t > 2147483647ll ? 2147483647 : (t < -2147483648ll ? (int32_t)-2147483648ll : (int32_t)t);
}
__device__ __forceinline__ float load_in_int(const FusedArgs& a, int c, size_t i) {
    if (c == 0) return (float)int_plane_at(a, 1, i) * a.in_m[0];
    if (c == 1) return (float)int_plane_at(a, 0, i) * a.in_m[1];
    return (float)xyb_b_plus_y(a.in_int, int_plane_at(a, 2, i), int_plane_at(a, 0, i)) * a.in_m[2];
}

template <bool TILED>
__device__ __forceinline__ float load_in(const FusedArgs& a, int c, int x, int y) {
    if constexpr (TILED) return a.in[0][coeff_tiled_index((uint32_t)x, (uint32_t)y, (uint32_t)c, a.in_w8)];
    else {
        if (a.in_int) return load_in_int(a, c, (size_t)y * a.in_stride + x);
        return a.in[c][(size_t)y * a.in_stride + x];
    }
}

// Refill the out-of-image cells of the region [lo, LWX-lo) x [lo, LWY-lo) of `buf` (3 planes) from their
// mirrored in-image cells.  (ox, oy) = image coordinate of LDS cell (0, 0).
template <int LWX, int LWY>
__device__ __forceinline__ void mirror_fill(float* buf, int lo, int ox, int oy, int width, int height, int t) {
    const int nx = LWX - 2 * lo, ny = LWY - 2 * lo;
    for (int i = t; i < nx * ny; i += 256) {
        int ly = lo + i / nx, lx = lo + i % nx;
        int x = ox + lx, y = oy + ly;
        if (x >= 0 && x < width && y >= 0 && y < height) continue;
        int sx = mirror_idx(x, width) - ox, sy = mirror_idx(y, height) - oy;
        // cells whose mirror source lies outside the region are beyond the reach of later stages
        if (sx < lo || sx >= LWX - lo || sy < lo || sy >= LWY - lo) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c * LWX * LWY + ly * LWX + lx] = buf[c * LWX * LWY + sy * LWX + sx];
    }
}

template <int STEP, int LWX, int LWY, int K>
__device__ __forceinline__ void epf_stage(const float* src, float* dst, int lo, int ox, int oy, int width,
                                          int height, const FusedArgs& a, int t, bool last, float (&res)[K][3], bool border = true) {
    constexpr int PLANE = LWX * LWY;
    const int nx = LWX - 2 * lo, ny = LWY - 2 * lo;
    const float step_multiplier = STEP == 0 ? a.fp.epf_pass0_sigma_scale
                                : STEP == 2 ? a.fp.epf_pass2_sigma_scale : 1.0f;
    int k = 0;
    for (int i = t; i < nx * ny; i += 256, ++k) {
        int ly = lo + i / nx, lx = lo + i % nx;
        int x = ox + lx, y = oy + ly;
        if (border && (x < 0 || x >= width || y < 0 || y >= height)) continue;   // (workgroup-uniform `border`: false = the whole halo lies inside the image)
        const float* p = src + ly * LWX + lx;
        float o[3];
        float sigma_val = a.sigma[(size_t)(y >> 3) * a.sigma_stride + (x >> 3)];
        if (sigma_val < 0.3f) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = p[c * PLANE];
        } else {
            float sm = epf_step_mul(x, y, step_multiplier, a.fp.epf_border_sad_mul);
            auto at = [&](int c, int dx, int dy) { return p[c * PLANE + dy * LWX + dx]; };
            epf_pixel<STEP>(at, sigma_val, sm, a.fp.epf_channel_scale, o);
        }
        if (last) {
#pragma unroll
            for (int c = 0; c < 3; ++c) res[k][c] = o[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c * PLANE + ly * LWX + lx] = o[c];
        }
    }
}

// TW x TH output tile; with a tile list (`a.tiles`) an entry is the tile's pixel origin x0 | y0 << 16.
template <bool GAB, int ITERS, bool TILED, int TW = T, int TH = T>
__device__ __forceinline__ void fused_post_body(const FusedArgs& a, float* lds, uint32_t tile_index) {
    using Cfg = PostCfg<GAB, ITERS, TW, TH>;
    constexpr int LWX = Cfg::LWX, LWY = Cfg::LWY, PLANE = Cfg::PLANE, HALO = Cfg::HALO;
    constexpr int K = (TW * TH + 255) / 256;  // output samples per lane
    float* bufA = lds;
    float* bufB = lds + 3 * PLANE;
    const int t = threadIdx.x;
    int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    if (a.tiles) {
        const uint32_t e = a.tiles[tile_index];
        tx0 = e & 0xffffu;
        ty0 = e >> 16;
    }
    const int ox = tx0 - HALO, oy = ty0 - HALO;  // image coordinate of LDS cell (0,0)
    const int W = a.width, H = a.height;
    const bool border = ox < 0 || oy < 0 || ox + LWX > W || oy + LWY > H;

    // ---- load tile + halo (mirrored at the image border), 3 channels.  A tile whose halo lies inside the image (workgroup-uniform:
    //      nearly every tile of a frame) needs neither the mirror nor, below, the per-sample edge tests of the stages.
    if (!border) {
        for (int i = t; i < PLANE; i += 256) {
            const int ly = i / LWX, lx = i % LWX;
            const int x = ox + lx, y = oy + ly;
            if constexpr (TILED) {
                const size_t base = coeff_tiled_index((uint32_t)x, (uint32_t)y, 0u, a.in_w8);   // channels are 64 words apart inside a cell
#pragma unroll
                for (int c = 0; c < 3; ++c) bufA[c * PLANE + i] = a.in[0][base + 64u * c];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) bufA[c * PLANE + i] = load_in<TILED>(a, c, x, y);
            }
        }
    } else {
        for (int i = t; i < PLANE; i += 256) {
            int ly = i / LWX, lx = i % LWX;
            int x = mirror_idx(ox + lx, W), y = mirror_idx(oy + ly, H);
#pragma unroll
            for (int c = 0; c < 3; ++c) bufA[c * PLANE + i] = load_in<TILED>(a, c, x, y);
        }
    }
    __syncthreads();

    float* src = bufA;
    float* dst = bufB;
    int lo = 0;  // the current stage's output region is [lo, LWX - lo) x [lo, LWY - lo)
    float res[K][3];

    if constexpr (GAB) {
        lo += 1;
        const int nx = LWX - 2 * lo, ny = LWY - 2 * lo;
        constexpr bool last = ITERS == 0;
        int k = 0;
        for (int i = t; i < nx * ny; i += 256, ++k) {
            int ly = lo + i / nx, lx = lo + i % nx;
            int x = ox + lx, y = oy + ly;
            if (border && (x < 0 || x >= W || y < 0 || y >= H)) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* p = src + c * PLANE + ly * LWX + lx;
                auto at = [&](int dx, int dy) { return p[dy * LWX + dx]; };
                float v;
                if (border) {
                    v = gabor_sample(at, x, y, W, H, a.fp.gab_weights[c][0], a.fp.gab_weights[c][1]);
                } else {   // the interior expression of gabor_sample (run_gabor_row_generic, gabor.rs:135-147): no edge row / column in reach
                    const float w0 = a.fp.gab_weights[c][0], w1 = a.fp.gab_weights[c][1];
                    const float sum_side = at(0, -1) + at(-1, 0) + at(1, 0) + at(0, 1);
                    const float sum_diag = at(-1, -1) + at(1, -1) + at(-1, 1) + at(1, 1);
                    v = (at(0, 0) + sum_side * w0 + sum_diag * w1) * (1.0f / (1.0f + w0 * 4.0f + w1 * 4.0f));
                }
                if (last) res[k][c] = v;
                else dst[c * PLANE + ly * LWX + lx] = v;
            }
        }
        if (!last) {
            __syncthreads();
            if (border) {
                mirror_fill<LWX, LWY>(dst, lo, ox, oy, W, H, t);
                __syncthreads();
            }
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    if constexpr (Cfg::HAS_E0) {
        lo += 3;
        constexpr bool last = ITERS == 4;
        epf_stage<0, LWX, LWY, K>(src, dst, lo, ox, oy, W, H, a, t, last, res, border);
        if (!last) {
            __syncthreads();
            if (border) {
                mirror_fill<LWX, LWY>(dst, lo, ox, oy, W, H, t);
                __syncthreads();
            }
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    if constexpr (Cfg::HAS_E1) {
        lo += 2;
        constexpr bool last = ITERS == 1;
        epf_stage<1, LWX, LWY, K>(src, dst, lo, ox, oy, W, H, a, t, last, res, border);
        if (!last) {
            __syncthreads();
            if (border) {
                mirror_fill<LWX, LWY>(dst, lo, ox, oy, W, H, t);
                __syncthreads();
            }
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    if constexpr (Cfg::HAS_E2) {
        lo += 1;
        epf_stage<2, LWX, LWY, K>(src, dst, lo, ox, oy, W, H, a, t, true, res, border);
    }
    if constexpr (!GAB && ITERS == 0) {
        // colour only: straight from the staged tile
        int k = 0;
        for (int i = t; i < TW * TH; i += 256, ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) res[k][c] = src[c * PLANE + (i / TW) * LWX + (i % TW)];
    }

    // ---- final region is the TW x TH tile: colour + store
    int k = 0;
    for (int i = t; i < TW * TH; i += 256, ++k) {
        int x = tx0 + i % TW, y = ty0 + i / TW;
        if (x >= W || y >= H) continue;
        float v[3] = {res[k][0], res[k][1], res[k][2]};
        if (a.do_color) color_pixel(a.color, v);
        size_t g = (size_t)y * a.out_stride + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[c][g] = v[c];
    }
}


template <bool GAB, int ITERS, bool TILED>
__global__ __launch_bounds__(256) void fused_post_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fused_post_body<GAB, ITERS, TILED>(a, lds, blockIdx.x);
}

// The border ring of the streaming path (EPF steps 1, 2): the first a.n_ring_h tiles of the list are
// 32 x 16 (top / bottom strips), the rest 16 x 32 (left / right strips).
template <bool GAB, bool TILED>
__device__ __forceinline__ void ring_body(const FusedArgs& a, float* lds, uint32_t tile) {
    if (tile < a.n_ring_h) fused_post_body<GAB, 2, TILED, kRingL, kRingT>(a, lds, tile);
    else fused_post_body<GAB, 2, TILED, kRingT, kRingL>(a, lds, tile);
}
template <bool GAB, bool TILED>
__global__ __launch_bounds__(256) void post_ring_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ring_body<GAB, TILED>(a, lds, blockIdx.x);
}

// the border ring of n frames in one launch (blockIdx.y = frame; default pipeline only)
__global__ __launch_bounds__(256) void post_ring_batch_kernel(FrameBatch b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FrameDevC fd = (FrameDevC)b.f[blockIdx.y];
    if (blockIdx.x >= fd->n_ring_tiles) return;
    const FusedArgs a = load_const(&fd->post);
    ring_body<true, true>(a, lds, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the same pipeline for the interior of the image (Gabor + EPF steps 1,2 +
// colour, the default `iters = 2` configuration): no LDS at all.  One wave owns a strip of 64
// columns (56 outputs + 4 halo lanes each side) and walks down `rows_per_seg` rows; every lane
// keeps the sliding row windows of its column in registers and reaches x-1 / x+1 with DPP
// wave shifts.  Absolute differences are shared between taps and between neighbouring pixels:
// with V(x,y) = |G(x,y) - G(x,y-1)| and H(x,y) = |G(x,y) - G(x-1,y)| every term
// |G(p+k+i) - G(p+i)| of epf_row<1> (filter/impls/generic/epf.rs:118-145) is a V or an H, summed
// in the reference's order, so results stay bit-identical while the subtractions drop 10x.
// The frame's outer ring of 32-px tiles (where the reference mirrors) goes through the tile
// kernel above; this kernel never sees a border.
constexpr int SW = 56;  // outputs per strip
constexpr int SH = 4;   // halo lanes / rows (1 Gabor + 2 EPF step 1 + 1 EPF step 2)

// bound_ctrl = 1: the lane without a source (0 resp. 63, always a halo lane) reads 0, which is what
// `old = 0` gave it before, but without the v_mov that materialised `old` in front of every shift.
__device__ __forceinline__ float from_left(float v) {   // value held by lane - 1 (pixel x - 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v) {  // value held by lane + 1 (pixel x + 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

struct StreamState {
    // row rings, slot = (row phase) & 3: I rows j-2..j; G, V, H rows g-3..g (g = j-1);
    // E rows e-2..e (e = j-3); Wd = |E(r) - E(r-1)| rows e-1..e
    float I[4][3], G[4][3], V[4][3], H[4][3], E[4][3], Wd[4][3];
    float nxt[3];
    float sig_nxt, sig_e, sig_f;  // sigma for rows e(next), e, f = e-1 (one load per row, prefetched)
    float inv_nxt, inv_e, inv_f;  // K / sigma for the same rows (recomputed when the 8x8 cell row changes)
    float d1_dn, d2_dn;           // scaled distance of the (0,+1) tap of the previous row == (0,-1) tap of this one
};

struct StreamConst {
    float gw0[3], gw1[3], ggw[3];
    float cs0, cs1, cs2, K;
    float sm1_in, sm1_bd, sm2_in, sm2_bd;
    bool x_bd, store_lane;
    int x, xl, yb;
    // lane parts of the addresses, in BYTES (a 32-bit byte offset from a uniform base is the
    // `global_load saddr + voffset` form: no vector address arithmetic per row)
    uint32_t in_off;   // input: (tiled: cell column * 192 + x & 7; planes: x) * 4
    uint32_t sig_off;  // sigma: (x >> 3) * 4
    uint32_t out_off;  // output: x * 4
};

__device__ __forceinline__ float ld_off(const float* uniform_base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(uniform_base) + byte_off);
}
__device__ __forceinline__ void st_off(float* uniform_base, uint32_t byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(uniform_base) + byte_off) = v;
}

// Input sample (x = this lane's column, row y) of channel c: uniform row base (scalar unit) + the
// lane's constant 32-bit offset, so a row costs no vector address arithmetic.
template <bool TILED>
__device__ __forceinline__ float load_row_in(const FusedArgs& a, const StreamConst& k, int c, int y) {
    if constexpr (TILED) {
        const float* rowbase = a.in[0] + ((size_t)(uint32_t)(y >> 3) * a.in_w8 * 192u + (uint32_t)((y & 7) << 3) + (uint32_t)c * 64u);
        return ld_off(rowbase, k.in_off);
    } else {
        const float* rowbase = a.in[c] + (size_t)y * a.in_stride;
        return ld_off(rowbase, k.in_off);
    }
}

// One row step; P = (j - j_start) & 3 is a compile-time phase so every ring slot below is a
// fixed register (no rotation moves).
template <int P, int TF, bool TILED, bool GAB>
__device__ __forceinline__ void stream_row(const FusedArgs& a, const StreamConst& k, StreamState& st, int j,
                                           const uint32_t* srgb_lut) {
    constexpr int s0 = P & 3, sm1 = (P + 3) & 3, sm2 = (P + 2) & 3, sm3 = (P + 1) & 3;  // rows j, j-1, j-2, j-3 (== j-4 -> s0)
    // ---- take the prefetched input row j, prefetch row j+1
#pragma unroll
    for (int c = 0; c < 3; ++c) st.I[s0][c] = st.nxt[c];
    {
        int jn = min(j + 1, a.height - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) st.nxt[c] = load_row_in<TILED>(a, k, c, jn);
        // sigma of row e+1 = j-2 for the next step; this step's f = e-1 reuses the previous e
        st.sig_f = st.sig_e; st.inv_f = st.inv_e;
        st.sig_e = st.sig_nxt; st.inv_e = st.inv_nxt;
        if (((j - 2) & 7) == 0) {  // wave-uniform: a new row of 8x8 cells starts
            const float* srow = a.sigma + (size_t)(uint32_t)((j - 2) >> 3) * a.sigma_stride;  // uniform
            st.sig_nxt = ld_off(srow, k.sig_off);
            st.inv_nxt = k.K / st.sig_nxt;
        }
    }
    // ---- Gabor row g = j-1 (run_gabor_row_generic interior expression, gabor.rs:135-147):
    //      t = I(j-2), c = I(j-1), b = I(j).  G(g) -> slot sm1; G(g-1) is slot sm2.
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float It = st.I[sm2][c], Ic = st.I[sm1][c], Ib = st.I[s0][c];
        float g;
        if constexpr (GAB) {
            float sum_side = It + from_left(Ic) + from_right(Ic) + Ib;
            float sum_diag = from_left(It) + from_right(It) + from_left(Ib) + from_right(Ib);
            g = (Ic + sum_side * k.gw0[c] + sum_diag * k.gw1[c]) * k.ggw[c];
        } else {
            g = Ic;  // no Gabor-like stage (Modular frames): the EPF reads the input rows directly
        }
        st.V[sm1][c] = fabsf(g - st.G[sm2][c]);
        st.G[sm1][c] = g;
        st.H[sm1][c] = fabsf(g - from_left(g));
    }
    // ---- EPF step 1, row e = j-3 = g-2: V(e-1..e+2) = slots s0(g-3) sm3(g-2) sm2(g-1) sm1(g);
    //      H, G (e-1..e+1) = slots s0, sm3, sm2
    const int e = j - 3;
    {
        float sigma_val = st.sig_e;
        bool y_bd = ((e + 1) & 6) == 0;
        float sm = (y_bd || k.x_bd) ? k.sm1_bd : k.sm1_in;
        float neg_inv_sigma = st.inv_e * sm;
        // The five |differences| of tap (0,+1) at row e are exactly those of tap (0,-1) at row e+1
        // (same values, same order), and tap (+1,0) at x is tap (-1,0) at x+1: only the "down"
        // and "left" distances are computed; "up" comes from the previous step, "right" from lane+1.
        float acc_dn[3], acc_lf[3], tapv[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float Vb = st.V[sm3][c], Vc = st.V[sm2][c], Vd = st.V[sm1][c];
            float Ha = st.H[s0][c], Hb = st.H[sm3][c], Hc = st.H[sm2][c];
            // tap (0,1):  V(x,y) V(x,y+1) V(x,y+2) V(x-1,y+1) V(x+1,y+1)
            acc_dn[c] = Vb + Vc + Vd + from_left(Vc) + from_right(Vc);
            // tap (-1,0): H(x,y-1) H(x,y) H(x,y+1) H(x-1,y) H(x+1,y)
            acc_lf[c] = Ha + Hb + Hc + from_left(Hb) + from_right(Hb);
            float Gc = st.G[sm3][c];
            tapv[0][c] = st.G[s0][c];
            tapv[1][c] = st.G[sm2][c];
            tapv[2][c] = from_left(Gc);
            tapv[3][c] = from_right(Gc);
        }
        float dist[4];
        dist[1] = k.cs0 * acc_dn[0];
        dist[1] += k.cs1 * acc_dn[1];
        dist[1] += k.cs2 * acc_dn[2];
        dist[2] = k.cs0 * acc_lf[0];
        dist[2] += k.cs1 * acc_lf[1];
        dist[2] += k.cs2 * acc_lf[2];
        dist[0] = st.d1_dn;
        st.d1_dn = dist[1];
        dist[3] = from_right(dist[2]);
        float sum_w = 1.0f;
        float sum_c[3] = {st.G[sm3][0], st.G[sm3][1], st.G[sm3][2]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float w = fmaxf(1.0f + dist[t] * neg_inv_sigma, 0.0f);
            sum_w += w;
#pragma unroll
            for (int c = 0; c < 3; ++c) sum_c[c] += w * tapv[t][c];
        }
        const bool copy = sigma_val < 0.3f;
        // E(e) -> slot sm3; E(e-1) is slot s0 (row j-4)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float r = copy ? st.G[sm3][c] : sum_c[c] / sum_w;
            st.Wd[sm3][c] = fabsf(r - st.E[s0][c]);
            st.E[sm3][c] = r;
        }
    }
    // ---- EPF step 2, row f = j-4 = e-1: E(f-1), E(f), E(f+1) = slots sm1 (row j-5), s0, sm3
    const int f = j - 4;
    float o[3];
    {
        float sigma_val = st.sig_f;
        bool y_bd = ((f + 1) & 6) == 0;
        float sm = (y_bd || k.x_bd) ? k.sm2_bd : k.sm2_in;
        float neg_inv_sigma = st.inv_f * sm;
        float d_dn[3], d_lf[3], tapv[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float Ec = st.E[s0][c];
            float el = from_left(Ec);
            d_dn[c] = st.Wd[sm3][c];              // |E(f+1) - E(f)|
            d_lf[c] = fabsf(el - Ec);             // |E(x-1,f) - E(x,f)|
            tapv[0][c] = st.E[sm1][c];
            tapv[1][c] = st.E[sm3][c];
            tapv[2][c] = el;
            tapv[3][c] = from_right(Ec);
        }
        float dist[4];
        dist[1] = k.cs0 * d_dn[0];
        dist[1] += k.cs1 * d_dn[1];
        dist[1] += k.cs2 * d_dn[2];
        dist[2] = k.cs0 * d_lf[0];
        dist[2] += k.cs1 * d_lf[1];
        dist[2] += k.cs2 * d_lf[2];
        dist[0] = st.d2_dn;                       // |E(f) - E(f-1)| terms: the previous row's "down"
        st.d2_dn = dist[1];
        dist[3] = from_right(dist[2]);            // |E(x+1,f) - E(x,f)| terms: lane+1's "left"
        float sum_w = 1.0f;
        float sum_c[3] = {st.E[s0][0], st.E[s0][1], st.E[s0][2]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float w = fmaxf(1.0f + dist[t] * neg_inv_sigma, 0.0f);
            sum_w += w;
#pragma unroll
            for (int c = 0; c < 3; ++c) sum_c[c] += w * tapv[t][c];
        }
        const bool copy = sigma_val < 0.3f;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = copy ? st.E[s0][c] : sum_c[c] / sum_w;
    }
    if (f >= k.yb && k.store_lane) {
        if (a.do_color) {
            if constexpr (TF == JXLGPU_TF_SRGB) color_pixel_srgb_lut(a.color, o, srgb_lut);
            else color_pixel(a.color, o);
        }
        const size_t orow = (size_t)(uint32_t)f * a.out_stride;  // uniform
#pragma unroll
        for (int c = 0; c < 3; ++c) st_off(a.out[c] + orow, k.out_off, o[c]);
    }
}

template <int TF, bool TILED, bool GAB>
__device__ __forceinline__ void post_stream_body(const FusedArgs& a, const uint32_t* srgb_lut) {
    // readfirstlane: the wave index is uniform, and telling the compiler so turns the whole row
    // bookkeeping (row index, 8x8 border tests, row base addresses) into scalar-unit work
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int strip = wave % a.strips, seg = wave / a.strips;
    if (seg >= a.segs) return;
    StreamConst k;
    k.x = a.sx0 + strip * SW - SH + lane;
    k.xl = min(k.x, a.width - 1);
    k.yb = a.sy0 + seg * a.rows_per_seg;
    const int ye = min(k.yb + a.rows_per_seg, a.sy1);
    k.store_lane = lane >= SH && lane < SH + SW && k.x < a.sx1;
    k.in_off = (TILED ? (uint32_t)(k.xl >> 3) * 192u + (uint32_t)(k.xl & 7) : (uint32_t)k.xl) * 4u;
    k.sig_off = (uint32_t)(k.xl >> 3) * 4u;
    k.out_off = (uint32_t)k.x * 4u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        k.gw0[c] = a.fp.gab_weights[c][0];
        k.gw1[c] = a.fp.gab_weights[c][1];
        k.ggw[c] = 1.0f / (1.0f + k.gw0[c] * 4.0f + k.gw1[c] * 4.0f);
    }
    k.cs0 = a.fp.epf_channel_scale[0]; k.cs1 = a.fp.epf_channel_scale[1]; k.cs2 = a.fp.epf_channel_scale[2];
    const float FRAC_1_SQRT_2 = 0.70710678118654752440f;
    k.K = 6.6f * (FRAC_1_SQRT_2 - 1.0f);
    const float border = a.fp.epf_border_sad_mul;
    k.sm1_in = 1.0f; k.sm1_bd = 1.0f * border;
    k.sm2_in = a.fp.epf_pass2_sigma_scale; k.sm2_bd = a.fp.epf_pass2_sigma_scale * border;
    k.x_bd = (k.x & 7) == 0 || (k.x & 7) == 7;

    StreamState st;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) st.I[i][c] = st.G[i][c] = st.V[i][c] = st.H[i][c] = st.E[i][c] = st.Wd[i][c] = 0.0f;
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) st.nxt[c] = load_row_in<TILED>(a, k, c, k.yb - SH);
        // first step is j = yb-SH with e = j-3: sig_nxt must hold sigma(row j-3) when it rotates in
        st.sig_e = st.sig_f = 1.0f;
        st.inv_e = st.inv_f = 0.0f;
        st.d1_dn = st.d2_dn = 0.0f;
        st.sig_nxt = ld_off(a.sigma + (size_t)(uint32_t)((k.yb - SH - 3) >> 3) * a.sigma_stride, k.sig_off);
        st.inv_nxt = k.K / st.sig_nxt;
    }
    // (ye - yb) and 2*SH are multiples of 4: whole groups of four phases
    for (int j = k.yb - SH; j < ye + SH; j += 4) {
        stream_row<0, TF, TILED, GAB>(a, k, st, j, srgb_lut);
        stream_row<1, TF, TILED, GAB>(a, k, st, j + 1, srgb_lut);
        stream_row<2, TF, TILED, GAB>(a, k, st, j + 2, srgb_lut);
        stream_row<3, TF, TILED, GAB>(a, k, st, j + 3, srgb_lut);
    }
}

template <int TF, bool TILED, bool GAB>
__global__ __launch_bounds__(256) void post_stream_kernel(FusedArgs a) {
    __shared__ uint32_t srgb_lut[16];
    if (threadIdx.x < 16) srgb_lut[threadIdx.x] = kSrgbMulBits[threadIdx.x];
    __syncthreads();
    post_stream_body<TF, TILED, GAB>(a, srgb_lut);
}

// n frames in one launch (blockIdx.y = frame): plain XYB -> sRGB from the cell-tiled transform output
__global__ __launch_bounds__(256) void post_stream_batch_kernel(FrameBatch b) {
    __shared__ uint32_t srgb_lut[16];
    if (threadIdx.x < 16) srgb_lut[threadIdx.x] = kSrgbMulBits[threadIdx.x];
    __syncthreads();
    const FusedArgs a = load_const(&((FrameDevC)b.f[blockIdx.y])->post);
    post_stream_body<JXLGPU_TF_SRGB, true, true>(a, srgb_lut);
}

#include "post_pk.inc"

template <bool GAB, int ITERS>
hipError_t launch_cfg(hipStream_t s, const FusedArgs& a, dim3 grid) {
    constexpr size_t lds_bytes = 2 * 3 * PostCfg<GAB, ITERS>::PLANE * sizeof(float);
    // > 64 KiB of dynamic LDS needs the attribute; setting it is idempotent and thread-safe
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_post_kernel<GAB, ITERS, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_post_kernel<GAB, ITERS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    if (a.in_w8) fused_post_kernel<GAB, ITERS, true><<<grid, 256, lds_bytes, s>>>(a);
    else fused_post_kernel<GAB, ITERS, false><<<grid, 256, lds_bytes, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_tile_kernel(hipStream_t s, const FusedArgs& a, bool gabor, int epf_iters, dim3 grid) {
    switch ((gabor ? 5 : 0) + epf_iters) {
        case 0: return launch_cfg<false, 0>(s, a, grid);
        case 1: return launch_cfg<false, 1>(s, a, grid);
        case 2: return launch_cfg<false, 2>(s, a, grid);
        case 3: return launch_cfg<false, 3>(s, a, grid);
        case 4: return launch_cfg<false, 4>(s, a, grid);
        case 5: return launch_cfg<true, 0>(s, a, grid);
        case 6: return launch_cfg<true, 1>(s, a, grid);
        case 7: return launch_cfg<true, 2>(s, a, grid);
        case 8: return launch_cfg<true, 3>(s, a, grid);
        default: return launch_cfg<true, 4>(s, a, grid);
    }
}

}  // namespace

bool fused_post_supported(const jxlgpu_ctx* ctx, const jxlgpu_frame* f, bool gabor, int epf_iters) {
    return !(ctx && ctx->tune.no_fused);
}

// Can the post stage of this frame read INTEGER planes (FusedArgs::in_int)?  Exactly when launch_fused_post will take the
// packed streaming kernel + the ring kernel: the conditions of fused_prepare's `stream` and `pk`, on the planes given.
bool fused_int_input_supported(const jxlgpu_ctx* ctx, const jxlgpu_frame* f, int epf_iters, const void* const in[3],
                               uint32_t in_stride, size_t elem) {
    if (!ctx || ctx->tune.no_fused || ctx->tune.no_stream || ctx->tune.no_pk || !ctx->tune.int_post) return false;
    if (epf_iters != 2 || f->width < 64 || f->height < 64 || f->width >= 65536 || f->height >= 65536) return false;
    const JxlGpuFilterParams& fp = f->desc.filter;
    if (!(fp.epf_channel_scale[0] >= 0 && fp.epf_channel_scale[1] >= 0 && fp.epf_channel_scale[2] >= 0 &&
          fp.epf_border_sad_mul >= 0 && fp.epf_pass2_sigma_scale >= 0)) return false;
    if ((in_stride & 1) || (f->wr & 1)) return false;
    for (int c = 0; c < 3; ++c)
        if ((reinterpret_cast<uintptr_t>(in[c]) & 7) || (elem != 2 && elem != 4) || !f->buf_a[c] || (reinterpret_cast<uintptr_t>(f->buf_a[c]) & 7)) return false;
    return true;
}

// Fills the arguments of the fused post stage of one frame and decides whether the streaming kernel
// applies (Gabor + 2 EPF steps on a frame with an interior; the ring-tile list is created on first
// use).  *plain_srgb: the colour tail is the plain XYB -> sRGB list (branch-free epilogue).
// Region renders: the origins (x0 | y0 << 16) in `tiles` go to scratch slot `slot` of the frame, in
// stream order.
static hipError_t upload_region_tiles(jxlgpu_ctx* ctx, jxlgpu_frame* f, int slot, const std::vector<uint32_t>& tiles,
                                      const uint32_t** dev) {
    const size_t n = std::max<size_t>(tiles.size(), 1);
    if (f->region_tiles_cap[slot] < n) {
        void* p = nullptr;
        hipError_t e = ctx_dev_malloc(ctx, &p, n * 2 * 4);
        if (e != hipSuccess) return e;
        f->allocs.push_back(p);  // the old one (if any) stays with the frame until it is freed: launches may still read it
        f->region_tiles[slot] = static_cast<uint32_t*>(p);
        f->region_tiles_cap[slot] = n * 2;
    }
    if (!tiles.empty()) {
        // through a pinned staging buffer of the context (a ring of four, each guarded by the event behind its last
        // copy): the copy is asynchronous for real and stream-ordered — a pageable source would make the runtime
        // block the host until every earlier kernel on the stream has finished
        StageBuf& sb = ctx->rt_stage[ctx->rt_stage_next++ % 4];
        if (sb.busy) {
            hipError_t e = hipEventSynchronize(sb.ev);
            if (e != hipSuccess) return e;
            sb.busy = false;
        }
        const size_t bytes = tiles.size() * 4;
        if (sb.cap < bytes) {
            if (sb.p) (void)hipHostFree(sb.p);
            sb.p = nullptr; sb.cap = 0;
            hipError_t e = hipHostMalloc(&sb.p, std::max<size_t>(bytes * 2, 4096), hipHostMallocDefault);
            if (e != hipSuccess) return e;
            sb.cap = std::max<size_t>(bytes * 2, 4096);
        }
        if (!sb.ev) {
            hipError_t e = hipEventCreateWithFlags(&sb.ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        memcpy(sb.p, tiles.data(), bytes);
        hipError_t e = hipMemcpyAsync(f->region_tiles[slot], sb.p, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return e;
        e = hipEventRecord(sb.ev, ctx->stream);
        if (e != hipSuccess) return e;
        sb.busy = true;
    }
    *dev = f->region_tiles[slot];
    return hipSuccess;
}

hipError_t fused_prepare(jxlgpu_ctx* ctx, jxlgpu_frame* f, const float* const in[3], uint32_t in_stride,
                         uint32_t in_tiled_w8, float* const out[3], uint32_t out_stride, bool gabor, int epf_iters,
                         bool color, FusedArgs* pa, bool* stream_out, bool* plain_srgb, int rows_per_seg) {
    FusedArgs& a = *pa;
    memset(&a, 0, sizeof(a));
    for (int c = 0; c < 3; ++c) { a.in[c] = in[c]; a.out[c] = out[c]; }
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.in_w8 = in_tiled_w8;  // != 0: in[0] is the cell-tiled transform output
    a.in_int = in_tiled_w8 ? 0u : f->post_in_int;   // integer planes of a Modular XYB frame (only with the packed streaming kernel, below)
    for (int c = 0; c < 3; ++c) a.in_m[c] = f->post_in_m[c];
    a.width = (int)f->width; a.height = (int)f->height;
    a.sigma = f->sigma; a.sigma_stride = f->w8;
    a.fp = f->desc.filter;
    a.color = f->color;
    a.do_color = color ? 1u : 0u;
    a.tiles = nullptr;
    // EPF steps 1, 2 (with or without the Gabor-like stage in front) on a frame with an interior: the
    // streaming kernel takes [sx0, sx1) x [sy0, sy1) — everything at least SH samples away from the
    // border, cut to whole 8x8 cells — and the tile kernel the ring of 16-px strips around it.
    const int W = (int)f->width, H = (int)f->height;
    const bool no_stream = ctx && ctx->tune.no_stream;
    bool stream = epf_iters == 2 && W >= 64 && H >= 64 && W < 65536 && H < 65536 && !no_stream;
    // Packed kernel (two columns per lane): 8-byte aligned input / output rows, and the sign conditions under which 1 + d * s <= 1 (clamp == max(., 0)).
    bool pk = false;
    {
        const JxlGpuFilterParams& fp = a.fp;
        pk = !(ctx && ctx->tune.no_pk) && (out_stride & 1) == 0 && (in_tiled_w8 || (in_stride & 1) == 0) &&
             fp.epf_channel_scale[0] >= 0 && fp.epf_channel_scale[1] >= 0 && fp.epf_channel_scale[2] >= 0 &&
             fp.epf_border_sad_mul >= 0 && fp.epf_pass2_sigma_scale >= 0;
        for (int c = 0; c < 3; ++c)
            pk = pk && (reinterpret_cast<uintptr_t>(out[c]) & 7) == 0 && (in_tiled_w8 || (reinterpret_cast<uintptr_t>(in[c]) & 7) == 0);
    }
    // JXLGPU_PK_TB=1: the packed kernel takes the image's top and bottom rows itself (round 6: measured, not the default — Tuning::pk_tb);
    // the tile kernel then serves the left / right columns only
    const bool tb = stream && pk && ctx && ctx->tune.pk_tb;
    const int sx0 = kRingT, sx1 = (W - SH) / 8 * 8;  // W - sx1 <= SH + 7 < kRingT
    const int sy0 = tb ? 0 : kRingT, sy1 = tb ? H : (H - SH) / 8 * 8;
    if (stream && (!f->ring_tiles || f->ring_tb != (tb ? 1u : 0u))) {
        std::vector<uint32_t> ring;  // pixel origins x0 | y0 << 16
        if (!tb)
            for (int x0 = 0; x0 < W; x0 += kRingL) {  // top, bottom: 32 x 16 tiles (corners included)
                ring.push_back((uint32_t)x0);
                ring.push_back((uint32_t)x0 | ((uint32_t)sy1 << 16));
            }
        const uint32_t n_h = (uint32_t)ring.size();
        for (int y0 = sy0; y0 < sy1; y0 += kRingL) {  // left, right: 16 x 32 tiles (the last pair may reach into the bottom
            ring.push_back((uint32_t)y0 << 16);        // strip — or past the image: the tile kernel clips — the same samples, written twice)
            ring.push_back((uint32_t)sx1 | ((uint32_t)y0 << 16));
        }
        void* p = nullptr;
        if (ctx_dev_malloc(ctx, &p, ring.size() * 4) != hipSuccess) {
            (void)hipGetLastError();
            stream = false;  // tile kernel over the whole frame
        } else {
            f->allocs.push_back(p);
            hipError_t e = hipMemcpy(p, ring.data(), ring.size() * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
            f->ring_tiles = static_cast<uint32_t*>(p);   // (a list built for the other mode stays with the frame until it is freed: launches may still read it)
            f->n_ring_tiles = (uint32_t)ring.size();
            f->n_ring_h = n_h;
            f->ring_host = ring;
            f->ring_tb = tb ? 1u : 0u;
        }
    }
    *stream_out = stream;
    if (stream) {
        a.sx0 = sx0; a.sx1 = sx1; a.sy0 = sy0; a.sy1 = sy1;
        a.n_ring_h = f->n_ring_h;
        a.rows_per_seg = rows_per_seg > 0 ? rows_per_seg : (ctx ? ctx->tune.stream_rows : 48);
        if (rows_per_seg < 0 && ctx) {
            // batched launches of -rows_per_seg frames: ONE resident round of waves (JXL_POST_WAVES per SIMD) per launch — the
            // tallest segments (8 run-in rows each) that still fill the chip, and no half-empty last round
            // (measured, 16 frames of 4K per launch: 5.5 rounds of 100-row segments 118 us per frame, one round of
            // 536-row segments 107)
            const int fpl = -rows_per_seg;
            const int strips_pk = (sx1 - sx0 + 119) / 120;   // packed kernel's strips (PW, defined below)
            const int slots = (int)ctx->num_cus * 4 * JXL_POST_WAVES;
            const int segs = std::max(1, (slots + fpl * strips_pk / 2) / (fpl * strips_pk));
            a.rows_per_seg = std::max(32, (sy1 - sy0 + segs - 1) / segs);
        }
        {   // equal segments (a short last one would pay the 8 run-in rows for little output)
            const int total = sy1 - sy0;
            const int n = std::max(1, (total + a.rows_per_seg / 2) / a.rows_per_seg);
            a.rows_per_seg = ((total + n - 1) / n + 3) / 4 * 4;
        }
        a.pk = pk ? 1u : 0u;
        a.tb = (tb && f->ring_tb) ? 1u : 0u;
        const int sw = pk ? PW : SW;
        a.strips = (a.sx1 - a.sx0 + sw - 1) / sw;
        a.segs = (a.sy1 - a.sy0 + a.rows_per_seg - 1) / a.rows_per_seg;
    }
    // plain XYB -> sRGB (no gamut map / second matrix) gets a branch-free colour epilogue
    *plain_srgb = a.color.tf == JXLGPU_TF_SRGB && !a.color.gamut_map && !a.color.has_matrix2 &&
                  !a.color.tone_map && !a.color.ycbcr && !a.color.staged_only;
    return hipSuccess;
}

// Returns a HIP error instead of silently skipping the launch (the caller switches `cur` to `out`
// only on success).  If the ring-tile list cannot be allocated the whole frame runs through the
// tile kernel, which needs no list.
// `rc` (region renders; null = the whole frame): only the output samples of that rectangle are needed.
// The launches are cut to it — the streaming rectangle to the whole cells it touches, the ring / tile
// launches to the tiles it touches — so what is written is a superset of `rc` made of the very same
// values a whole-frame render writes there; the inputs have to be valid 48 samples around `rc`.
hipError_t launch_fused_post(hipStream_t s, jxlgpu_frame* f, const float* const in[3], uint32_t in_stride,
                             uint32_t in_tiled_w8, float* const out[3], uint32_t out_stride, bool gabor, int epf_iters,
                             bool color, jxlgpu_ctx* ctx, const PixRect* rc) {
    FusedArgs a;
    bool stream = false, plain_srgb = false;
    hipError_t e;
    // tiles of tw x th at the origins of a regular grid that touch `r`
    auto grid_tiles = [&](const PixRect& r, int tw, int th) {
        std::vector<uint32_t> v;
        for (int y0 = std::max(0, r.y0) / th * th; y0 < std::min((int)f->height, r.y1); y0 += th)
            for (int x0 = std::max(0, r.x0) / tw * tw; x0 < std::min((int)f->width, r.x1); x0 += tw)
                v.push_back((uint32_t)x0 | ((uint32_t)y0 << 16));
        return v;
    };
    // epf_iters == 3 on a frame the streaming kernels can take: step 0 (+ the Gabor-like stage) through
    // the tile kernel into the frame's spare plane set, then steps 1, 2 (+ colour) through the streaming
    // path — every stage still mirrors its own input at the border.  The all-in-one tile kernel pays
    // a 7-sample halo (46 x 46 cells per 32 x 32 outputs) for three stages.
    if (epf_iters == 3 && in_tiled_w8 && f->buf_b[0] && f->width >= 64 && f->height >= 64 && f->width < 65536 &&
        f->height < 65536 && !(ctx && ctx->tune.no_stream)) {
        float* const* tmp = out[0] == f->buf_a[0] ? f->buf_b : f->buf_a;
        e = fused_prepare(ctx, f, in, in_stride, in_tiled_w8, tmp, f->wr, gabor, 4, false, &a, &stream, &plain_srgb, 0);
        if (e != hipSuccess) return e;
        if (rc) {
            // step 0 has to cover what steps 1, 2 read around the launches cut to `rc`: 40 samples around it
            const PixRect r0{rc->x0 - 40, rc->y0 - 40, rc->x1 + 40, rc->y1 + 40};
            const std::vector<uint32_t> tl = grid_tiles(r0, T, T);
            if ((e = upload_region_tiles(ctx, f, 0, tl, &a.tiles)) != hipSuccess) return e;
            if (!tl.empty()) e = launch_tile_kernel(s, a, gabor, 4, dim3((uint32_t)tl.size()));
        } else {
            e = launch_tile_kernel(s, a, gabor, 4, dim3(ceil_div(f->width, T), ceil_div(f->height, T)));
        }
        if (e != hipSuccess) return e;
        const float* const in2[3] = {tmp[0], tmp[1], tmp[2]};
        return launch_fused_post(s, f, in2, f->wr, 0u, out, out_stride, false, 2, color, ctx, rc);
    }
    e = fused_prepare(ctx, f, in, in_stride, in_tiled_w8, out, out_stride, gabor, epf_iters, color, &a, &stream,
                      &plain_srgb, 0);
    if (e != hipSuccess) return e;
    // integer input is read by the packed streaming kernel and the ring kernel only (fused_int_input_supported said so)
    if (a.in_int && !(stream && a.pk)) return hipErrorInvalidValue;
    if (!stream) {
        if (!rc) return launch_tile_kernel(s, a, gabor, epf_iters, dim3(ceil_div(f->width, T), ceil_div(f->height, T)));
        const std::vector<uint32_t> tl = grid_tiles(*rc, T, T);
        if ((e = upload_region_tiles(ctx, f, 1, tl, &a.tiles)) != hipSuccess) return e;
        return tl.empty() ? hipSuccess : launch_tile_kernel(s, a, gabor, epf_iters, dim3((uint32_t)tl.size()));
    }
    std::vector<uint32_t> ring_sel;   // region renders: the ring tiles `rc` touches (horizontal ones first)
    uint32_t ring_sel_h = 0;
    const uint32_t* ring_dev = nullptr;
    if (rc) {
        // the streaming rectangle cut to the whole cells `rc` touches (it stays a multiple of 8 away from sx0 / sy0)
        a.sx0 = std::max(a.sx0, rc->x0 / 8 * 8); a.sy0 = std::max(a.sy0, rc->y0 / 8 * 8);
        a.sx1 = std::min(a.sx1, (rc->x1 + 7) / 8 * 8); a.sy1 = std::min(a.sy1, (rc->y1 + 7) / 8 * 8);
        if (a.sx1 > a.sx0 && a.sy1 > a.sy0) {
            const int total = a.sy1 - a.sy0;
            const int base_rows = ctx ? ctx->tune.stream_rows : 48;
            const int n = std::max(1, (total + base_rows / 2) / base_rows);
            a.rows_per_seg = ((total + n - 1) / n + 3) / 4 * 4;
            a.strips = (a.sx1 - a.sx0 + (a.pk ? PW : SW) - 1) / (a.pk ? PW : SW);
            a.segs = (total + a.rows_per_seg - 1) / a.rows_per_seg;
        } else {
            a.strips = a.segs = 0;
        }
        for (uint32_t i = 0; i < (uint32_t)f->ring_host.size(); ++i) {
            const int x0 = (int)(f->ring_host[i] & 0xffffu), y0 = (int)(f->ring_host[i] >> 16);
            const int tw = i < f->n_ring_h ? kRingL : kRingT, th = i < f->n_ring_h ? kRingT : kRingL;
            if (x0 < rc->x1 && x0 + tw > rc->x0 && y0 < rc->y1 && y0 + th > rc->y0) {
                ring_sel.push_back(f->ring_host[i]);
                if (i < f->n_ring_h) ++ring_sel_h;
            }
        }
        // ordered on `s` in front of the fork event: the side stream's ring launch sees the list
        if ((e = upload_region_tiles(ctx, f, 1, ring_sel, &ring_dev)) != hipSuccess) return e;
    }
    const int waves = a.strips * a.segs;
    const bool side = ctx && ctx->stream2;
    if (side && (e = hipEventRecord(ctx->ev_fork, s)) != hipSuccess) return e;  // inputs are ready here
    const dim3 sgrid((waves + 3) / 4);
#define STREAM(TF, TILED)                                                            \
    do {                                                                              \
        if (gabor) post_stream_kernel<TF, TILED, true><<<sgrid, 256, 0, s>>>(a);      \
        else post_stream_kernel<TF, TILED, false><<<sgrid, 256, 0, s>>>(a);           \
    } while (0)
#define STREAM_PK(TF, TILED)                                                        \
    do {                                                                              \
        if (gabor) post_pk_kernel<TF, TILED, true><<<sgrid, 256, 0, s>>>(a);          \
        else post_pk_kernel<TF, TILED, false><<<sgrid, 256, 0, s>>>(a);               \
    } while (0)
    if (waves == 0) {
        // region inside the border ring: no streaming launch
    } else if (a.pk && a.in_w8) {
        if (plain_srgb) STREAM_PK(JXLGPU_TF_SRGB, true); else STREAM_PK(-1, true);
    } else if (a.pk) {
        if (plain_srgb) STREAM_PK(JXLGPU_TF_SRGB, false); else STREAM_PK(-1, false);
    } else if (a.in_w8) {
        if (plain_srgb) STREAM(JXLGPU_TF_SRGB, true); else STREAM(-1, true);
    } else {
        if (plain_srgb) STREAM(JXLGPU_TF_SRGB, false); else STREAM(-1, false);
    }
#undef STREAM
#undef STREAM_PK
    if ((e = hipGetLastError()) != hipSuccess) return e;
    a.tiles = f->ring_tiles;
    uint32_t n_ring = f->n_ring_tiles;
    if (rc) {
        a.tiles = ring_dev;
        n_ring = (uint32_t)ring_sel.size();
        a.n_ring_h = ring_sel_h;
    }
    // the border ring (a few hundred long-latency tiles) runs beside the streaming kernel
    constexpr size_t ring_lds = 2 * 3 * PostCfg<true, 2, kRingL, kRingT>::PLANE * sizeof(float);
    hipStream_t rs = side ? ctx->stream2 : s;
    if (side && (e = hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0)) != hipSuccess) return e;
    if (n_ring == 0) {
        // region inside the streaming rectangle: no ring launch
    } else if (gabor) {
        if (a.in_w8) post_ring_kernel<true, true><<<n_ring, 256, ring_lds, rs>>>(a);
        else post_ring_kernel<true, false><<<n_ring, 256, ring_lds, rs>>>(a);
    } else {
        if (a.in_w8) post_ring_kernel<false, true><<<n_ring, 256, ring_lds, rs>>>(a);
        else post_ring_kernel<false, false><<<n_ring, 256, ring_lds, rs>>>(a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (side) {
        if ((e = hipEventRecord(ctx->ev_join, ctx->stream2)) != hipSuccess) return e;
        return hipStreamWaitEvent(s, ctx->ev_join, 0);
    }
    return hipSuccess;
}

// Default pipeline of n frames: streaming kernel on `s`, border rings beside it on `side` (may be
// null: same stream).  The caller forks / joins the two streams once per batch.
hipError_t launch_post_batch(hipStream_t s, hipStream_t side, const FrameBatch& b, uint32_t n, uint32_t max_stream_wgs,
                             uint32_t max_ring, bool pk, bool fast, int lds_pad) {
    constexpr size_t lds_bytes = 2 * 3 * PostCfg<true, 2, kRingL, kRingT>::PLANE * sizeof(float);  // 23 KB
    if (max_ring) post_ring_batch_kernel<<<dim3(max_ring, n), 256, lds_bytes, side ? side : s>>>(b);
    if (lds_pad > 0) {   // JXLGPU_POST_LDS_PAD (experiment): reserved, never touched
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&post_pk_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        });
    }
#ifdef JXL_ENABLE_POST_FAST
    if (max_stream_wgs && pk && fast) { post_pk_fast_batch_kernel<<<dim3(max_stream_wgs, n), 256, 0, s>>>(b); return hipGetLastError(); }
#else
    (void)fast;
#endif
    if (max_stream_wgs && pk) post_pk_batch_kernel<<<dim3(max_stream_wgs, n), 256, (size_t)std::max(0, lds_pad), s>>>(b);
    else if (max_stream_wgs) post_stream_batch_kernel<<<dim3(max_stream_wgs, n), 256, 0, s>>>(b);
    return hipGetLastError();
}
