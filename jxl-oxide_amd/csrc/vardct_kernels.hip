// VarDCT stage kernels for gfx950: LF dequant + CfL-LF, adaptive LF smoothing, and the varblock
// transform (HF dequant, CfL-HF, LF->LLF injection, inverse variable-size DCT).
//
// What they compute follows the reference's generic CPU path (cited per function); how they are
// organised is MI355X-first: the host sorts varblocks into per-shape work lists at upload time,
// one 256-thread workgroup stages NB same-shape varblocks x 3 channels in LDS (rows padded by one
// word so both the row pass and the column pass are bank-conflict free), each lane runs whole
// 1-D butterflies in registers, and all HBM traffic is 16-byte vectors.
#include "common.h"
#include "dct_device.h"

#include "afv_basis.inc"

// ---------------------------------------------------------------- V1 + V2
// copy_lf_dequant (jxl-render/src/vardct/mod.rs:387-412) + chroma_from_luma_lf (:544-568).
__global__ __launch_bounds__(256) void lf_dequant_cfl_kernel(LfArgs a) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= a.w8) return;
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

void launch_lf_dequant_cfl(hipStream_t s, const LfArgs& a) {
    dim3 grid(ceil_div(a.w8, 256), a.h8);
    lf_dequant_cfl_kernel<<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------- V3
// adaptive_lf_smoothing_impl (jxl-render/src/vardct/generic/mod.rs:11-103).  The CPU code runs
// in place but only ever reads unsmoothed neighbours (udsum scratch + `prev` carry), so it is an
// out-of-place 3x3 stencil; border samples are copied.
__global__ __launch_bounds__(256) void lf_smooth_kernel(SmoothArgs a) {
    const float SCALE_SELF = 0.052262735f, SCALE_SIDE = 0.2034514f, SCALE_DIAG = 0.03348292f;
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= a.w8) return;
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

void launch_lf_smooth(hipStream_t s, const SmoothArgs& a) {
    dim3 grid(ceil_div(a.w8, 256), a.h8);
    lf_smooth_kernel<<<grid, 256, 0, s>>>(a);
}

// ---------------------------------------------------------------- V4: one coefficient
// dequant_hf_varblock_grouped inner loop, jxl-render/src/vardct/mod.rs:527-537
__device__ __forceinline__ float dequant_one(int32_t qn, float quant_bias, float qbn, float m, float mul) {
    float q = (float)qn;
    if (fabsf(q) <= 1.0f) q *= quant_bias;
    else q -= qbn / q;
    q *= m;
    q *= mul;
    return q;
}

// The same value through a table of quant_bias_numerator / k (k = |q| < 256, built on the host with
// the same correctly rounded f32 division): qbn / q == sign(q) * (qbn / |q|) exactly, so only the
// rare |q| >= 256 still divides.  Saves the 12-instruction division sequence and the per-coefficient
// exec-mask branches in the hot A1 loop.
__device__ __forceinline__ float dequant_one_lut(int32_t qn, float quant_bias, float qbn, const float* qlut, float m,
                                                 float mul) {
    float q = (float)qn;
    const uint32_t aq = qn < 0 ? 0u - (uint32_t)qn : (uint32_t)qn;
    float t = qlut[min(aq, 255u)];
    if (__builtin_expect(aq > 255u, 0)) t = qbn / fabsf(q);
    const float big = q - (qn < 0 ? -t : t);
    const float small = q * quant_bias;
    q = aq <= 1u ? small : big;
    q *= m;
    q *= mul;
    return q;
}

// ---------------------------------------------------------------- V8: special 8x8 transforms
// jxl-render/src/vardct/generic/transform.rs:14-219, operating on one 8x8 block in LDS
// (row stride S).  One lane per (block, channel); these types are ~10 % of blocks.
template <int S>
struct Blk {
    float* p;
    __device__ __forceinline__ float& operator()(int x, int y) const { return p[y * S + x]; }
};

template <int S, int SIZE>
__device__ __forceinline__ void aux_idct2_in_place(Blk<S> c) {
    constexpr int n = SIZE / 2;
    float s[SIZE][SIZE];
#pragma unroll
    for (int y = 0; y < n; ++y)
#pragma unroll
        for (int x = 0; x < n; ++x) {
            float c00 = c(x, y), c01 = c(x + n, y), c10 = c(x, y + n), c11 = c(x + n, y + n);
            s[2 * y][2 * x] = c00 + c01 + c10 + c11;
            s[2 * y][2 * x + 1] = c00 + c01 - c10 - c11;
            s[2 * y + 1][2 * x] = c00 - c01 + c10 - c11;
            s[2 * y + 1][2 * x + 1] = c00 - c01 - c10 + c11;
        }
#pragma unroll
    for (int y = 0; y < SIZE; ++y)
#pragma unroll
        for (int x = 0; x < SIZE; ++x) c(x, y) = s[y][x];
}

// inverse dct_2d of a 4x4 held as m[row][col]: rows first, then columns (dct.rs:93-140)
__device__ __forceinline__ void idct2d_4x4(float (&m)[4][4], const SecLarge& sl) {
#pragma unroll
    for (int y = 0; y < 4; ++y) idct<4>(m[y], sl);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        float col[4] = {m[0][x], m[1][x], m[2][x], m[3][x]};
        idct<4>(col, sl);
#pragma unroll
        for (int y = 0; y < 4; ++y) m[y][x] = col[y];
    }
}
// inverse dct_2d of 8 wide x 4 tall
__device__ __forceinline__ void idct2d_8x4(float (&m)[4][8], const SecLarge& sl) {
#pragma unroll
    for (int y = 0; y < 4; ++y) idct<8>(m[y], sl);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        float col[4] = {m[0][x], m[1][x], m[2][x], m[3][x]};
        idct<4>(col, sl);
#pragma unroll
        for (int y = 0; y < 4; ++y) m[y][x] = col[y];
    }
}

template <int S>
__device__ void transform_dct2(Blk<S> c) {
    aux_idct2_in_place<S, 2>(c);
    aux_idct2_in_place<S, 4>(c);
    aux_idct2_in_place<S, 8>(c);
}

template <int S>
__device__ void transform_dct4(Blk<S> c, const SecLarge& sl) {
    aux_idct2_in_place<S, 2>(c);
    float out[8][8];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float m[4][4];  // scratch.get_mut(iy, ix) = coeff(x + ix*2, y + iy*2): row ix, col iy
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) m[ix][iy] = c(x + ix * 2, y + iy * 2);
            idct2d_4x4(m, sl);
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) out[y * 4 + iy][x * 4 + ix] = m[iy][ix];
        }
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) c(x, y) = out[y][x];
}

template <int S>
__device__ void transform_hornuss(Blk<S> c) {
    aux_idct2_in_place<S, 2>(c);
    float out[8][8];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float s[16];
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) s[iy * 4 + ix] = c(x + ix * 2, y + iy * 2);
            float residual_sum = 0.0f;
#pragma unroll
            for (int i = 1; i < 16; ++i) residual_sum += s[i];
            float avg = s[0] - residual_sum / 16.0f;
            s[0] = s[5];
            s[5] = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] += avg;
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) out[y * 4 + iy][x * 4 + ix] = s[iy * 4 + ix];
        }
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) c(x, y) = out[y][x];
}

template <int S, bool TR>
__device__ void transform_dct4x8(Blk<S> c, const SecLarge& sl) {
    float coeff0 = c(0, 0), coeff1 = c(0, 1);
    c(0, 0) = coeff0 + coeff1;
    c(0, 1) = coeff0 - coeff1;
    float scratch[8][8];
#pragma unroll
    for (int idx = 0; idx < 2; ++idx) {
        float m[4][8];
#pragma unroll
        for (int iy = 0; iy < 4; ++iy)
#pragma unroll
            for (int ix = 0; ix < 8; ++ix) m[iy][ix] = c(ix, iy * 2 + idx);
        idct2d_8x4(m, sl);
#pragma unroll
        for (int iy = 0; iy < 4; ++iy)
#pragma unroll
            for (int ix = 0; ix < 8; ++ix) scratch[idx * 4 + iy][ix] = m[iy][ix];
    }
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            if (TR) c(y, x) = scratch[y][x];
            else c(x, y) = scratch[y][x];
        }
}

template <int S>
__device__ void transform_afv(Blk<S> c, int n, const SecLarge& sl) {
    int flip_x = n % 2, flip_y = n / 2;
    float coeff_afv[16];
    coeff_afv[0] = (c(0, 0) + c(1, 0) + c(0, 1)) * 4.0f;
#pragma unroll
    for (int idx = 1; idx < 16; ++idx) coeff_afv[idx] = c(2 * (idx % 4), 2 * (idx / 4));
    float samples_afv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) samples_afv[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) samples_afv[j] = __builtin_fmaf(coeff_afv[i], AFV_BASIS[i][j], samples_afv[j]);

    float m44[4][4];  // scratch_4x4[ix*4 + iy] = coeff(2ix+1, 2iy): row ix, col iy
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) m44[ix][iy] = c(2 * ix + 1, 2 * iy);
    m44[0][0] = c(0, 0) - c(1, 0) + c(0, 1);
    idct2d_4x4(m44, sl);

    float m48[4][8];
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int ix = 0; ix < 8; ++ix) m48[iy][ix] = c(ix, 2 * iy + 1);
    m48[0][0] = c(0, 0) - c(0, 1);
    idct2d_8x4(m48, sl);

    for (int iy = 0; iy < 4; ++iy) {
        int afv_y = flip_y == 0 ? iy : 3 - iy;
        for (int ix = 0; ix < 4; ++ix) {
            int afv_x = flip_x == 0 ? ix : 3 - ix;
            c(flip_x * 4 + ix, flip_y * 4 + iy) = samples_afv[afv_y * 4 + afv_x];
        }
    }
    for (int iy = 0; iy < 4; ++iy) {
        int y = flip_y * 4 + iy;
        for (int ix = 0; ix < 4; ++ix) c((1 - flip_x) * 4 + ix, y) = m44[iy][ix];
    }
    for (int iy = 0; iy < 4; ++iy) {
        int y = (1 - flip_y) * 4 + iy;
        for (int ix = 0; ix < 8; ++ix) c(ix, y) = m48[iy][ix];
    }
}

// ---------------------------------------------------------------- V4-V8: the varblock kernel
// W, H: pixel size of the varblock shape; SPECIAL: the 8x8 non-DCT8 family.
// entries[i] = {cell_x | cell_y << 16, TransformType, hf_mul, 0}: one 16-byte record per varblock,
// written by the host at upload in group-then-raster order so neighbouring lanes touch
// neighbouring cache lines and no lane has to chase BlockInfo -> hf_mul -> LF through HBM.
template <int W, int H>
struct VbCfg {
    static constexpr int BW = W / 8, BH = H / 8;
#ifndef JXL_VB_TILE
#define JXL_VB_TILE 2048
#endif
    static constexpr int NB = (W * H >= JXL_VB_TILE) ? 1 : JXL_VB_TILE / (W * H);
    static constexpr int S = W + 1;            // padded LDS row stride (words)
    static constexpr int BLK = H * S;          // words per block per channel
    static constexpr int CH = NB * BLK;        // words per channel
    static constexpr int LDS_WORDS = 3 * CH + NB;  // tiles + per-block cell position
};

template <int W, int H, bool SPECIAL, int NT = 256, int NBX = VbCfg<W, H>::NB, bool LUT = false>
__device__ __forceinline__ void run_class(const TransformArgs& a, const uint4* __restrict__ entries,
                                          int nvalid, float* lds, const float* qlut = nullptr) {
    using Cfg = VbCfg<W, H>;
    constexpr int NB = NBX, S = Cfg::S, BLK = Cfg::BLK, CH = NB * BLK, BW = Cfg::BW, BH = Cfg::BH;
    float* tile = lds;
    uint32_t* s_cell = reinterpret_cast<uint32_t*>(lds + 3 * CH);

    const SecLarge sl{a.sec64, a.sec128, a.sec256};
    const int t = threadIdx.x;

    // ---- A1: 4 coefficients x 3 channels per lane-iteration: dequantise (V4, mod.rs:513-537),
    //          chroma-from-luma (V5, mod.rs:589-600), stage in LDS.  The LLF corner is left to A2.
    constexpr int VEC_PER_BLK = W * H / 4;
    constexpr int VECS = NB * VEC_PER_BLK;
#pragma unroll 2
    for (int v4 = t; v4 < VECS; v4 += NT) {
        int blk = v4 / VEC_PER_BLK;
        if (blk >= nvalid) break;
        int r = v4 % VEC_PER_BLK;
        int y = r / (W / 4), x = (r % (W / 4)) * 4;
        const uint4 e = entries[blk];
        if (r == 0) s_cell[blk] = e.x;
        uint32_t px = (e.x & 0xffffu) * 8 + x, py = (e.x >> 16) * 8 + y;
        size_t goff = (size_t)py * a.cstride + px;
        int4 q[3];
        float4 m[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q[c] = *reinterpret_cast<const int4*>(a.coeff[c] + goff);
            uint32_t off = a.deq_off[e.y * 3 + c];
            m[c] = *reinterpret_cast<const float4*>(a.dequant + off + y * W + x);
        }
        float kx = a.kx_map[(py >> 6) * a.w64 + (px >> 6)];
        float kb = a.kb_map[(py >> 6) * a.w64 + (px >> 6)];
        const float mul_base = 65536.0f / (a.global_scale * (float)(int32_t)e.z);
        float d[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float mul = mul_base * a.qm_scale[c];
            if constexpr (LUT) {
                d[c][0] = dequant_one_lut(q[c].x, a.quant_bias[c], a.quant_bias_numerator, qlut, m[c].x, mul);
                d[c][1] = dequant_one_lut(q[c].y, a.quant_bias[c], a.quant_bias_numerator, qlut, m[c].y, mul);
                d[c][2] = dequant_one_lut(q[c].z, a.quant_bias[c], a.quant_bias_numerator, qlut, m[c].z, mul);
                d[c][3] = dequant_one_lut(q[c].w, a.quant_bias[c], a.quant_bias_numerator, qlut, m[c].w, mul);
            } else {
                d[c][0] = dequant_one(q[c].x, a.quant_bias[c], a.quant_bias_numerator, m[c].x, mul);
                d[c][1] = dequant_one(q[c].y, a.quant_bias[c], a.quant_bias_numerator, m[c].y, mul);
                d[c][2] = dequant_one(q[c].z, a.quant_bias[c], a.quant_bias_numerator, m[c].z, mul);
                d[c][3] = dequant_one(q[c].w, a.quant_bias[c], a.quant_bias_numerator, m[c].w, mul);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float cy_ = d[1][j];
            d[0][j] += kx * cy_;
            d[2][j] += kb * cy_;
        }
        const bool corner_row = y < BH && x < BW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float* dst = tile + c * CH + blk * BLK + y * S + x;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(corner_row && x + j < BW)) dst[j] = d[c][j];
        }
    }
    // ---- A2: one lane per (block, channel): LF -> lowest-frequency coefficients
    //          (transform_common.rs:40-66: copy LF, forward DCT, divide by scale_f products)
    if (t < NB * 3) {
        int blk = t / 3, c = t % 3;
        if (blk < nvalid) {
            const uint32_t pos = entries[blk].x;
            size_t cell = (size_t)(pos >> 16) * a.w8 + (pos & 0xffffu);
            float v[BH][BW];
#pragma unroll
            for (int y = 0; y < BH; ++y)
#pragma unroll
                for (int x = 0; x < BW; ++x) v[y][x] = a.lf[c][cell + (size_t)y * a.w8 + x];
            if constexpr (BW * BH > 1) {
                fdct2d_small<BW, BH>(v, sl);
                constexpr int sy = 5 - __builtin_ctz(BH), sx = 5 - __builtin_ctz(BW);
#pragma unroll
                for (int y = 0; y < BH; ++y)
#pragma unroll
                    for (int x = 0; x < BW; ++x) v[y][x] /= kScaleF[y << sy] * kScaleF[x << sx];
            }
            float* dst = tile + c * CH + blk * BLK;
#pragma unroll
            for (int y = 0; y < BH; ++y)
#pragma unroll
                for (int x = 0; x < BW; ++x) dst[y * S + x] = v[y][x];
        }
    }
    __syncthreads();

    if constexpr (SPECIAL) {
        // ---- one lane per (block, channel): transform.rs:225-240 dispatch
        if (t < NB * 3) {
            int blk = t / 3, c = t % 3;
            if (blk < nvalid) {
                Blk<S> b{tile + c * CH + blk * BLK};
                switch (entries[blk].y) {
                    case JXLGPU_DCT2: transform_dct2<S>(b); break;
                    case JXLGPU_DCT4: transform_dct4<S>(b, sl); break;
                    case JXLGPU_HORNUSS: transform_hornuss<S>(b); break;
                    case JXLGPU_DCT4X8: transform_dct4x8<S, false>(b, sl); break;
                    case JXLGPU_DCT8X4: transform_dct4x8<S, true>(b, sl); break;
                    case JXLGPU_AFV0: transform_afv<S>(b, 0, sl); break;
                    case JXLGPU_AFV1: transform_afv<S>(b, 1, sl); break;
                    case JXLGPU_AFV2: transform_afv<S>(b, 2, sl); break;
                    case JXLGPU_AFV3: transform_afv<S>(b, 3, sl); break;
                    default: break;
                }
            }
        }
    } else {
        // ---- P2: 1-D inverse DCT of every row (dct_2d, dct.rs:93-96), one row per lane
        constexpr int ROWS = 3 * NB * H;
        for (int r = t; r < ROWS; r += NT) {
            int c = r / (NB * H), rb = r % (NB * H);
            int blk = rb / H, y = rb % H;
            if (blk >= nvalid) continue;
            float* row = tile + c * CH + blk * BLK + y * S;
            float v[W];
#pragma unroll
            for (int x = 0; x < W; ++x) v[x] = row[x];
            idct<W>(v, sl);
#pragma unroll
            for (int x = 0; x < W; ++x) row[x] = v[x];
        }
        __syncthreads();
        // ---- P3: 1-D inverse DCT of every column (dct.rs:109-130), one column per lane
        constexpr int COLS = 3 * NB * W;
        for (int r = t; r < COLS; r += NT) {
            int c = r / (NB * W), rb = r % (NB * W);
            int blk = rb / W, x = rb % W;
            if (blk >= nvalid) continue;
            float* col = tile + c * CH + blk * BLK + x;
            float v[H];
#pragma unroll
            for (int y = 0; y < H; ++y) v[y] = col[y * S];
            idct<H>(v, sl);
#pragma unroll
            for (int y = 0; y < H; ++y) col[y * S] = v[y];
        }
    }
    __syncthreads();

    // ---- P4: 16-byte stores of the finished samples
    for (int v4 = t; v4 < VECS; v4 += NT) {
        int blk = v4 / VEC_PER_BLK;
        if (blk >= nvalid) break;
        int r = v4 % VEC_PER_BLK;
        int y = r / (W / 4), x = (r % (W / 4)) * 4;
        uint32_t e = s_cell[blk];
        uint32_t px = (e & 0xffffu) * 8 + x, py = (e >> 16) * 8 + y;
        size_t goff = (size_t)py * a.pstride + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* src = tile + c * CH + blk * BLK + y * S + x;
            float4 o = make_float4(src[0], src[1], src[2], src[3]);
            *reinterpret_cast<float4*>(a.pix[c] + goff) = o;
        }
    }
}

constexpr int kSmallLdsWords = VbCfg<8, 8>::LDS_WORDS > VbCfg<32, 32>::LDS_WORDS ? VbCfg<8, 8>::LDS_WORDS : VbCfg<32, 32>::LDS_WORDS;  // max over the classes above
static_assert(VbCfg<8, 8>::LDS_WORDS <= kSmallLdsWords && VbCfg<16, 16>::LDS_WORDS <= kSmallLdsWords &&
              VbCfg<8, 16>::LDS_WORDS <= kSmallLdsWords && VbCfg<16, 8>::LDS_WORDS <= kSmallLdsWords &&
              VbCfg<32, 32>::LDS_WORDS <= kSmallLdsWords && VbCfg<8, 32>::LDS_WORDS <= kSmallLdsWords &&
              VbCfg<32, 8>::LDS_WORDS <= kSmallLdsWords && VbCfg<16, 32>::LDS_WORDS <= kSmallLdsWords &&
              VbCfg<32, 16>::LDS_WORDS <= kSmallLdsWords, "LDS budget of transform_small_kernel");

// All varblock shapes up to 32x32 in ONE launch: every workgroup reads its descriptor
// {class, first entry, count} and branches (workgroup-uniformly) into the code for that shape, so
// the thin classes (a few hundred 32x8 blocks...) fill the machine together instead of each
// paying a launch and a tail.
template <bool LUT>
__global__ __launch_bounds__(256) void transform_small_kernel(TransformArgs a, const uint4* __restrict__ wgs,
                                                              const uint4* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint4 wg = wgs[blockIdx.x];
    const uint4* e = entries + wg.y;
    const int n = (int)wg.z;
    // quant_bias_numerator / k table behind the tiles (dequant_one_lut); nullptr keeps the division
    float* qlut = nullptr;
    if constexpr (LUT) {
        qlut = lds + kSmallLdsWords;
        qlut[threadIdx.x] = a.deq_lut[threadIdx.x];
        __syncthreads();
    }
    switch (wg.x) {
        case CLS_DCT8: run_class<8, 8, false, 256, VbCfg<8, 8>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_16x16: run_class<16, 16, false, 256, VbCfg<16, 16>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_8x16: run_class<8, 16, false, 256, VbCfg<8, 16>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_16x8: run_class<16, 8, false, 256, VbCfg<16, 8>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_32x32: run_class<32, 32, false, 256, VbCfg<32, 32>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_8x32: run_class<8, 32, false, 256, VbCfg<8, 32>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_32x8: run_class<32, 8, false, 256, VbCfg<32, 8>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_16x32: run_class<16, 32, false, 256, VbCfg<16, 32>::NB, LUT>(a, e, n, lds, qlut); break;
        case CLS_32x16: run_class<32, 16, false, 256, VbCfg<32, 16>::NB, LUT>(a, e, n, lds, qlut); break;
        default: break;
    }
}



void launch_transform_small(hipStream_t s, const TransformArgs& a, const uint4* wgs, uint32_t n_wgs,
                            const uint4* entries) {
    if (!n_wgs) return;
    if (a.deq_lut) transform_small_kernel<true><<<n_wgs, 256, (kSmallLdsWords + 256) * sizeof(float), s>>>(a, wgs, entries);
    else transform_small_kernel<false><<<n_wgs, 256, kSmallLdsWords * sizeof(float), s>>>(a, wgs, entries);
}

// The 8x8 non-DCT family (Hornuss, DCT2, DCT4, 4x8, 8x4, AFV): register-hungry serial code per
// block, kept out of the kernel above so it does not drag its occupancy down; runs on the side
// stream together with the 64-pixel shapes.
constexpr int kSpecialNB = 8;   // 8 blocks x 3 channels = 24 serial transforms per 64-lane workgroup
__global__ __launch_bounds__(64) void transform_special_kernel(TransformArgs a, const uint4* __restrict__ entries,
                                                               uint32_t count) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t first = blockIdx.x * kSpecialNB;
    run_class<8, 8, true, 64, kSpecialNB>(a, entries + first, (int)min((uint32_t)kSpecialNB, count - first), lds);
}

// 64-pixel shapes: one wave per (varblock, channel).  The tile of one channel (<= 16.6 KiB) is
// staged in LDS; X and B recompute the dequantised Y coefficient for chroma-from-luma instead of
// sharing it through LDS, which buys 3x more independent waves for these rare, long blocks.  The
// LF -> LLF forward DCT (up to 8x8) runs one row / one column per lane.
template <int W, int H>
__global__ __launch_bounds__(64) void transform_kernel64(TransformArgs a, const uint4* __restrict__ entries) {
    constexpr int S = W + 1, BW = W / 8, BH = H / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;                 // H x S
    float* llf = lds + H * S;          // BH x (BW + 1)
    constexpr int LS = BW + 1;
    const SecLarge sl{a.sec64, a.sec128, a.sec256};
    const int t = threadIdx.x;
    const int c = blockIdx.y;
    const uint4 e = entries[blockIdx.x];
    const uint32_t px0 = (e.x & 0xffffu) * 8, py0 = (e.x >> 16) * 8;
    const size_t cell = (size_t)(e.x >> 16) * a.w8 + (e.x & 0xffffu);
    const float mul_base = 65536.0f / (a.global_scale * (float)(int32_t)e.z);
    const float mul_c = mul_base * a.qm_scale[c], mul_y = mul_base * a.qm_scale[1];
    const float* mat_c = a.dequant + a.deq_off[e.y * 3 + c];
    const float* mat_y = a.dequant + a.deq_off[e.y * 3 + 1];

    // LF samples of this varblock -> LDS
    for (int i = t; i < BW * BH; i += 64) {
        int y = i / BW, x = i % BW;
        llf[y * LS + x] = a.lf[c][cell + (size_t)y * a.w8 + x];
    }
    // V4 + V5: dequantise + chroma-from-luma, 4 coefficients per lane-iteration
    constexpr int VECS = W * H / 4;
#pragma unroll 4
    for (int v4 = t; v4 < VECS; v4 += 64) {
        int y = v4 / (W / 4), x = (v4 % (W / 4)) * 4;
        uint32_t px = px0 + x, py = py0 + y;
        size_t goff = (size_t)py * a.cstride + px;
        int4 q = *reinterpret_cast<const int4*>(a.coeff[c] + goff);
        float4 m = *reinterpret_cast<const float4*>(mat_c + y * W + x);
        float d[4];
        d[0] = dequant_one(q.x, a.quant_bias[c], a.quant_bias_numerator, m.x, mul_c);
        d[1] = dequant_one(q.y, a.quant_bias[c], a.quant_bias_numerator, m.y, mul_c);
        d[2] = dequant_one(q.z, a.quant_bias[c], a.quant_bias_numerator, m.z, mul_c);
        d[3] = dequant_one(q.w, a.quant_bias[c], a.quant_bias_numerator, m.w, mul_c);
        if (c != 1) {
            int4 qy = *reinterpret_cast<const int4*>(a.coeff[1] + goff);
            float4 my = *reinterpret_cast<const float4*>(mat_y + y * W + x);
            uint32_t ti = (py >> 6) * a.w64 + (px >> 6);
            float k = c == 0 ? a.kx_map[ti] : a.kb_map[ti];
            d[0] += k * dequant_one(qy.x, a.quant_bias[1], a.quant_bias_numerator, my.x, mul_y);
            d[1] += k * dequant_one(qy.y, a.quant_bias[1], a.quant_bias_numerator, my.y, mul_y);
            d[2] += k * dequant_one(qy.z, a.quant_bias[1], a.quant_bias_numerator, my.z, mul_y);
            d[3] += k * dequant_one(qy.w, a.quant_bias[1], a.quant_bias_numerator, my.w, mul_y);
        }
        float* dst = tile + y * S + x;
        const bool corner_row = y < BH && x < BW;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(corner_row && x + j < BW)) dst[j] = d[j];
    }
    __syncthreads();
    // V6: forward DCT of the BW x BH LF block (dct_2d general case: rows, then columns), scale_f
    if (t < BH) {
        float v[BW];
#pragma unroll
        for (int x = 0; x < BW; ++x) v[x] = llf[t * LS + x];
        fdct<BW>(v, sl);
#pragma unroll
        for (int x = 0; x < BW; ++x) llf[t * LS + x] = v[x];
    }
    __syncthreads();
    if (t < BW) {
        float v[BH];
#pragma unroll
        for (int y = 0; y < BH; ++y) v[y] = llf[y * LS + t];
        fdct<BH>(v, sl);
        constexpr int sy = 5 - __builtin_ctz(BH), sx = 5 - __builtin_ctz(BW);
#pragma unroll
        for (int y = 0; y < BH; ++y) tile[y * S + t] = v[y] / (kScaleF[y << sy] * kScaleF[t << sx]);
    }
    __syncthreads();
    // V7: rows, then columns
    if (t < H) {
        float* row = tile + t * S;
        float v[W];
#pragma unroll
        for (int x = 0; x < W; ++x) v[x] = row[x];
        idct<W>(v, sl);
#pragma unroll
        for (int x = 0; x < W; ++x) row[x] = v[x];
    }
    __syncthreads();
    if (t < W) {
        float* col = tile + t;
        float v[H];
#pragma unroll
        for (int y = 0; y < H; ++y) v[y] = col[y * S];
        idct<H>(v, sl);
#pragma unroll
        for (int y = 0; y < H; ++y) col[y * S] = v[y];
    }
    __syncthreads();
    for (int v4 = t; v4 < VECS; v4 += 64) {
        int y = v4 / (W / 4), x = (v4 % (W / 4)) * 4;
        const float* src = tile + y * S + x;
        *reinterpret_cast<float4*>(a.pix[c] + (size_t)(py0 + y) * a.pstride + px0 + x) =
            make_float4(src[0], src[1], src[2], src[3]);
    }
}

template <int W, int H>
static void launch_tk64(hipStream_t s, const TransformArgs& a, const uint4* entries, uint32_t count) {
    constexpr int bytes = (H * (W + 1) + (H / 8) * (W / 8 + 1)) * sizeof(float);
    transform_kernel64<W, H><<<dim3(count, 3), 64, bytes, s>>>(a, entries);
}

void launch_big_blocks(hipStream_t s, const TransformArgs& a, const uint4* entries, uint32_t count);

void launch_transform_class(hipStream_t s, int cls, const TransformArgs& a, const uint4* entries,
                            uint32_t count) {
    if (count == 0) return;
    switch (cls) {
        case CLS_SPECIAL8:
            transform_special_kernel<<<ceil_div(count, kSpecialNB), 64,
                                       (3 * kSpecialNB * VbCfg<8, 8>::BLK + kSpecialNB) * sizeof(float), s>>>(a, entries, count);
            break;
        case CLS_64x64: launch_tk64<64, 64>(s, a, entries, count); break;
        case CLS_32x64: launch_tk64<32, 64>(s, a, entries, count); break;
        case CLS_64x32: launch_tk64<64, 32>(s, a, entries, count); break;
        case CLS_BIG: launch_big_blocks(s, a, entries, count); break;
        default: break;
    }
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
            a.pix[c][(size_t)py * a.pstride + px] = a.lf[c][(size_t)(py / 8) * a.w8 + px / 8];
    }
}

void launch_nometa_groups(hipStream_t s, const TransformArgs& a, const uint32_t* groups,
                          uint32_t count, uint32_t group_dim, uint32_t groups_per_row) {
    if (!count) return;
    nometa_kernel<<<dim3(group_dim, count), 256, 0, s>>>(a, groups, group_dim, groups_per_row);
}
