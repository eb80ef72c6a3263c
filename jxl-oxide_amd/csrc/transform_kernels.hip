// V4-V8 on gfx950: HF dequantisation, chroma-from-luma, LF -> LLF injection and the inverse
// variable-size DCT (jxl-render/src/vardct/mod.rs:442-682, transform_common.rs:11-75,
// generic/{dct,transform}.rs), one WAVE per work item, no workgroup barriers.
//
// Layout the kernels rely on: the i32 coefficients live in HBM as 8x8 cells,
// channel-interleaved — cell (cx, cy) is 3 x 64 words {X, Y, B}, each 8 rows of 8 — so any
// varblock, whatever its shape or alignment, reads whole 256-byte runs and never shares a cache
// line with a neighbour of another shape (the row-major planes of round 1 fetched every line
// once per shape class that touched it: 2.4x read amplification).
//
// A work item is NBI = 64 / min(W, H) varblocks of one shape.  The wave makes, per channel:
//   row pass   : lane = one row (W coefficients) -> 16-byte loads straight into registers ->
//                dequantise (+ CfL from the Y row kept in registers) -> 1-D IDCT in registers ->
//                one ds_write_b32 per sample into a padded LDS tile (bank-conflict free);
//   column pass: lane = one column: H ds_read_b32 -> 1-D IDCT -> H dword stores.
// The LDS tile belongs to the wave; LDS operations of one wave execute in order, so the passes
// are ordered by compiler fences only.  Operation order inside every butterfly is the reference's
// (dct_device.h), hence results are bit-identical to the CPU path.
#include "common.h"
#include "dct_device.h"

#include "afv_basis.inc"

namespace {

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dequant_hf_varblock_grouped inner loop (vardct/mod.rs:527-537) with `qbn / q` from a table of
// quant_bias_numerator / k (k = |q| < 256, built on the host with the same correctly rounded f32
// division): qbn / q == sign(q) * (qbn / |q|) exactly, so only |q| >= 256 still divides.
__device__ __forceinline__ float dequant_lut(int32_t qn, float quant_bias, float qbn, const float* qlut, float m,
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

template <int W, int H>
constexpr int type_of() {
    if (W == 8 && H == 8) return JXLGPU_DCT8;
    if (W == 16 && H == 16) return JXLGPU_DCT16;
    if (W == 8 && H == 16) return JXLGPU_DCT16X8;
    if (W == 16 && H == 8) return JXLGPU_DCT8X16;
    if (W == 32 && H == 32) return JXLGPU_DCT32;
    if (W == 8 && H == 32) return JXLGPU_DCT32X8;
    if (W == 32 && H == 8) return JXLGPU_DCT8X32;
    if (W == 16 && H == 32) return JXLGPU_DCT32X16;
    if (W == 32 && H == 16) return JXLGPU_DCT16X32;
    if (W == 64 && H == 64) return JXLGPU_DCT64;
    if (W == 32 && H == 64) return JXLGPU_DCT64X32;
    return JXLGPU_DCT32X64;
}

constexpr int block_stride(int W, int H) {
    // >= H * (W + 1); congruent to W mod 32 when W < 32 so the column lanes of successive blocks
    // (W lanes each) fall on distinct LDS banks
    int v = H * (W + 1);
    if (W >= 32) return v;
    while (v % 32 != W % 32) ++v;
    return v;
}

template <int W_, int H_>
struct RCfg {
    static constexpr int W = W_, H = H_, BW = W / 8, BH = H / 8;
    static constexpr int MINWH = W < H ? W : H;
    static constexpr int NBI = 64 / MINWH;        // varblocks per work item
    static constexpr int RP = NBI * H / 64;       // row passes per channel
    static constexpr int CP = NBI * W / 64;       // column passes per channel
    static constexpr int S = W + 1;               // padded row stride (words)
    static constexpr int BS = block_stride(W, H); // block stride (words)
    static constexpr int T_WORDS = NBI * BS;
    static constexpr int LLF_WORDS = NBI * 3 * BW * BH;
    static constexpr int WAVE_WORDS = T_WORDS + LLF_WORDS;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int kWaveWordsA = cmax(cmax(RCfg<8, 8>::WAVE_WORDS, RCfg<16, 16>::WAVE_WORDS),
                                 cmax(RCfg<8, 16>::WAVE_WORDS, RCfg<16, 8>::WAVE_WORDS));
constexpr int kWaveWordsB = cmax(cmax(cmax(RCfg<32, 32>::WAVE_WORDS, RCfg<8, 32>::WAVE_WORDS),
                                      cmax(RCfg<32, 8>::WAVE_WORDS, RCfg<16, 32>::WAVE_WORDS)),
                                 RCfg<32, 16>::WAVE_WORDS);
constexpr int kLutWords = 256;

// chroma-from-luma factor of a row that may straddle a 64-px tile column (varblocks need not be
// aligned): samples x < split take k0, the rest k1 (chroma_from_luma_hf_grouped, mod.rs:589-600)
struct CflRow {
    float kx0, kx1, kb0, kb1;
    int split;
};

template <int W>
__device__ __forceinline__ CflRow cfl_row(const TransformArgs& a, uint32_t px0, uint32_t py) {
    CflRow r;
    const uint32_t t0 = (py >> 6) * a.w64 + (px0 >> 6), t1 = (py >> 6) * a.w64 + ((px0 + W - 1) >> 6);
    r.kx0 = a.kx_map[t0]; r.kb0 = a.kb_map[t0];
    r.kx1 = a.kx_map[t1]; r.kb1 = a.kb_map[t1];
    r.split = 64 - (int)(px0 & 63u);
    return r;
}

// One row of W coefficients of channel c at cell row (cy + y / 8), in-cell row y % 8.
template <int W>
__device__ __forceinline__ void load_row(const TransformArgs& a, uint32_t cx, uint32_t cy, int y, int c,
                                         int4 (&raw)[W / 4]) {
    const int32_t* src = a.coeff + (((((size_t)(cy + (uint32_t)(y >> 3)) * a.w8 + cx) * 3 + (uint32_t)c) << 6) + ((y & 7) << 3));
#pragma unroll
    for (int i = 0; i < W / 8; ++i) {
        raw[2 * i] = *reinterpret_cast<const int4*>(src + i * 192);
        raw[2 * i + 1] = *reinterpret_cast<const int4*>(src + i * 192 + 4);
    }
}

template <int W>
__device__ __forceinline__ void load_mrow(const float* mrow, float4 (&m)[W / 4]) {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) m[i] = *reinterpret_cast<const float4*>(mrow + 4 * i);
}

template <int W>
__device__ __forceinline__ void dequant_row(const int4 (&raw)[W / 4], const float4 (&m)[W / 4], float bias, float qbn,
                                            const float* lut, float mul, float (&d)[W]) {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) {
        d[4 * i + 0] = dequant_lut(raw[i].x, bias, qbn, lut, m[i].x, mul);
        d[4 * i + 1] = dequant_lut(raw[i].y, bias, qbn, lut, m[i].y, mul);
        d[4 * i + 2] = dequant_lut(raw[i].z, bias, qbn, lut, m[i].z, mul);
        d[4 * i + 3] = dequant_lut(raw[i].w, bias, qbn, lut, m[i].w, mul);
    }
}

template <int W>
__device__ __forceinline__ void cfl_apply(float (&d)[W], const float (&y)[W], float k0, float k1, int split,
                                          bool any_straddle) {
    if (!any_straddle) {  // wave-uniform: every row of this pass lies inside one 64-px tile column
#pragma unroll
        for (int x = 0; x < W; ++x) d[x] += k0 * y[x];
    } else {
#pragma unroll
        for (int x = 0; x < W; ++x) d[x] += (x < split ? k0 : k1) * y[x];
    }
}

// ---------------------------------------------------------------------------------------------
// One work item: up to NBI varblocks of shape W x H, all three channels.
template <int W, int H, bool PREFETCH_ALL>
__device__ __forceinline__ void run_item(const TransformArgs& a, const uint4* __restrict__ ent, int nvalid,
                                         float* __restrict__ T, const float* __restrict__ lut, int lane) {
    using C = RCfg<W, H>;
    constexpr int BW = C::BW, BH = C::BH, NBI = C::NBI, RP = C::RP, CP = C::CP, S = C::S, BS = C::BS;
    constexpr int TYPE = type_of<W, H>();
    float* llf = T + C::T_WORDS;
    const SecLarge sl{a.sec64, a.sec128, a.sec256};

    // ---- V6 first half: LF -> lowest-frequency coefficients (transform_common.rs:40-66: copy the
    //      BW x BH LF samples, forward DCT, divide by the scale_f products); one lane per
    //      (varblock, channel), parked in LDS for the row lanes.
    if (lane < NBI * 3) {
        const int blk = lane / 3, c = lane - blk * 3;
        if (blk < nvalid) {
            const uint32_t pos = ent[blk].x;
            const size_t cell = (size_t)(pos >> 16) * a.w8 + (pos & 0xffffu);
            const float* lfp = c == 0 ? a.lf[0] : (c == 1 ? a.lf[1] : a.lf[2]);
            float v[BH][BW];
#pragma unroll
            for (int y = 0; y < BH; ++y)
#pragma unroll
                for (int x = 0; x < BW; ++x) v[y][x] = lfp[cell + (size_t)y * a.w8 + x];
            if constexpr (BW * BH > 1) {
                fdct2d_small<BW, BH>(v, sl);
                constexpr int sy = 5 - __builtin_ctz(BH), sx = 5 - __builtin_ctz(BW);
#pragma unroll
                for (int y = 0; y < BH; ++y)
#pragma unroll
                    for (int x = 0; x < BW; ++x) v[y][x] /= kScaleF[y << sy] * kScaleF[x << sx];
            }
            float* dst = llf + lane * (BW * BH);
#pragma unroll
            for (int y = 0; y < BH; ++y)
#pragma unroll
                for (int x = 0; x < BW; ++x) dst[y * BW + x] = v[y][x];
        }
    }

    // ---- geometry of this lane's rows (one per row pass) and columns (one per column pass)
    uint32_t rcx[RP], rcy[RP];
    int ry[RP], rblk[RP];
    bool rvalid[RP];
    float rmul[RP];
    CflRow cfl[RP];
    bool straddle = false;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        const int R = p * 64 + lane;
        rblk[p] = R / H;
        ry[p] = R % H;
        rvalid[p] = rblk[p] < nvalid;
        const uint4 e = ent[rvalid[p] ? rblk[p] : 0];
        rcx[p] = e.x & 0xffffu;
        rcy[p] = e.x >> 16;
        rmul[p] = 65536.0f / (a.global_scale * (float)(int32_t)e.z);
        cfl[p] = cfl_row<W>(a, rcx[p] * 8, rcy[p] * 8 + (uint32_t)ry[p]);
        straddle |= cfl[p].split < W;
    }
    const bool any_straddle = __builtin_amdgcn_ballot_w64(straddle) != 0;
    uint32_t ccx[CP], ccy[CP];
    int cxi[CP], cblk[CP];
    bool cvalid[CP];
#pragma unroll
    for (int q = 0; q < CP; ++q) {
        const int Cc = q * 64 + lane;
        cblk[q] = Cc / W;
        cxi[q] = Cc % W;
        cvalid[q] = cblk[q] < nvalid;
        const uint32_t pos = ent[cvalid[q] ? cblk[q] : 0].x;
        ccx[q] = pos & 0xffffu;
        ccy[q] = pos >> 16;
    }
    wave_lds_sync();

    // ---- optional: every channel's coefficients in flight at once (few, long items: one wave
    //      per SIMD, registers to spare, latency is what counts)
    int4 raw_all[PREFETCH_ALL ? 3 : 1][PREFETCH_ALL ? RP : 1][W / 4];
    if constexpr (PREFETCH_ALL) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < RP; ++p)
                if (rvalid[p]) load_row<W>(a, rcx[p], rcy[p], ry[p], c, raw_all[c][p]);
    }

    float ydq[RP][W];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);  // Y first: X and B need its dequantised row
        const float bias = a.quant_bias[c], qms = a.qm_scale[c];
        const float* mat = a.dequant + a.deq_off_v[TYPE * 3 + c];
        float* pixc = a.pix[c];
        // ---- row passes: V4 dequant, V5 chroma-from-luma, LLF patch, 1-D IDCT of the row (dct.rs:93-96)
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            if (!rvalid[p]) continue;
            const int y = ry[p];
            float4 m[W / 4];
            load_mrow<W>(mat + y * W, m);
            float d[W];
            if constexpr (PREFETCH_ALL) {
                dequant_row<W>(raw_all[c][p], m, bias, a.quant_bias_numerator, lut, rmul[p] * qms, d);
            } else {
                int4 raw[W / 4];
                load_row<W>(a, rcx[p], rcy[p], y, c, raw);
                dequant_row<W>(raw, m, bias, a.quant_bias_numerator, lut, rmul[p] * qms, d);
            }
            if (ci == 0) {
#pragma unroll
                for (int x = 0; x < W; ++x) ydq[p][x] = d[x];
            } else if (ci == 1) {
                cfl_apply<W>(d, ydq[p], cfl[p].kx0, cfl[p].kx1, cfl[p].split, any_straddle);
            } else {
                cfl_apply<W>(d, ydq[p], cfl[p].kb0, cfl[p].kb1, cfl[p].split, any_straddle);
            }
            if (y < BH) {
                const float* src = llf + (rblk[p] * 3 + c) * (BW * BH) + y * BW;
#pragma unroll
                for (int x = 0; x < BW; ++x) d[x] = src[x];
            }
            idct<W>(d, sl);
            float* row = T + rblk[p] * BS + y * S;
#pragma unroll
            for (int x = 0; x < W; ++x) row[x] = d[x];
        }
        wave_lds_sync();
        // ---- column passes: 1-D IDCT of the column (dct.rs:109-130), samples straight to HBM
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            if (!cvalid[q]) continue;
            const float* col = T + cblk[q] * BS + cxi[q];
            float v[H];
#pragma unroll
            for (int y = 0; y < H; ++y) v[y] = col[y * S];
            idct<H>(v, sl);
            float* dst = pixc + (size_t)(ccy[q] * 8) * a.pstride + ccx[q] * 8 + cxi[q];
#pragma unroll
            for (int y = 0; y < H; ++y) dst[(size_t)y * a.pstride] = v[y];
        }
        wave_lds_sync();
    }
}

// Work items of one launch: classes in launch order, `begin[k]` = first item of class k.
struct ItemTable {
    uint32_t n_classes;
    uint32_t begin[6];       // n_classes + 1 entries used
    uint32_t cls[5];
    uint32_t first_entry[5]; // into `entries`
    uint32_t count[5];       // varblocks of the class
};

// FAMILY 0: 8x8 and the 16-px shapes (the bulk: thousands of short items, occupancy hides latency).
// FAMILY 1: the 32-px shapes (about a thousand long items per 4K frame: roughly one wave per SIMD,
//           so every channel's loads are issued up front instead).
template <int FAMILY>
__global__ __launch_bounds__(256) void transform_rows_kernel(TransformArgs a, ItemTable it,
                                                             const uint4* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WAVE_WORDS = FAMILY == 0 ? kWaveWordsA : kWaveWordsB;
    float* lut = lds;
    lut[threadIdx.x] = a.deq_lut[threadIdx.x];
    __syncthreads();  // the only workgroup barrier: from here on the four waves are independent
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + wave;
    if (item >= it.begin[it.n_classes]) return;
    uint32_t k = 0;
    while (item >= it.begin[k + 1]) ++k;
    float* T = lds + kLutWords + wave * WAVE_WORDS;
    const uint32_t idx = item - it.begin[k];
#define RUN(W, H, PF)                                                                      \
    {                                                                                      \
        constexpr int NBI = RCfg<W, H>::NBI;                                               \
        const uint32_t first = idx * NBI;                                                  \
        run_item<W, H, PF>(a, entries + it.first_entry[k] + first,                         \
                           (int)min((uint32_t)NBI, it.count[k] - first), T, lut, lane);    \
    }
    if constexpr (FAMILY == 0) {
        switch (it.cls[k]) {
            case CLS_DCT8: RUN(8, 8, false) break;
            case CLS_16x16: RUN(16, 16, false) break;
            case CLS_8x16: RUN(8, 16, false) break;
            case CLS_16x8: RUN(16, 8, false) break;
            default: break;
        }
    } else {
        switch (it.cls[k]) {
            case CLS_32x32: RUN(32, 32, true) break;
            case CLS_8x32: RUN(8, 32, true) break;
            case CLS_32x8: RUN(32, 8, true) break;
            case CLS_16x32: RUN(16, 32, true) break;
            case CLS_32x16: RUN(32, 16, true) break;
            default: break;
        }
    }
#undef RUN
}

// ---------------------------------------------------------------------------------------------
// 64-px shapes (Dct64, Dct64x32, Dct32x64): one wave per (varblock, channel), row lane / column
// lane as above with the whole row (up to 64 coefficients) in registers.  X and B recompute the
// dequantised Y row for chroma-from-luma instead of sharing it, which makes the three channels of
// a block independent waves — there are only a few hundred such blocks in a 4K frame, latency per
// wave is what matters.  The LF -> LLF forward DCT (up to 8x8) runs one row / one column per lane
// (dct_2d general case: rows, then columns).
template <int W, int H>
__global__ __launch_bounds__(64) void transform_kernel64(TransformArgs a, const uint4* __restrict__ entries) {
    constexpr int S = W + 1, BW = W / 8, BH = H / 8, LS = BW + 1;
    constexpr int TYPE = type_of<W, H>();
    __shared__ float T[H * S + BH * LS + kLutWords];
    float* llf = T + H * S;
    float* lut = llf + BH * LS;
    const SecLarge sl{a.sec64, a.sec128, a.sec256};
    const int lane = threadIdx.x;
    const int c = blockIdx.y;
    const uint4 e = entries[blockIdx.x];
    const uint32_t cx = e.x & 0xffffu, cy = e.x >> 16;
    const size_t cell = (size_t)cy * a.w8 + cx;
    const float* lfp = c == 0 ? a.lf[0] : (c == 1 ? a.lf[1] : a.lf[2]);
    float* pixc = c == 0 ? a.pix[0] : (c == 1 ? a.pix[1] : a.pix[2]);

    // coefficient rows in flight first (lane = row)
    int4 raw[W / 4], rawy[W / 4];
    if (lane < H) {
        load_row<W>(a, cx, cy, lane, c, raw);
        if (c != 1) load_row<W>(a, cx, cy, lane, 1, rawy);
    }
#pragma unroll
    for (int i = 0; i < kLutWords / 64; ++i) lut[i * 64 + lane] = a.deq_lut[i * 64 + lane];
    for (int i = lane; i < BW * BH; i += 64) {
        const int y = i / BW, x = i % BW;
        llf[y * LS + x] = lfp[cell + (size_t)y * a.w8 + x];
    }
    wave_lds_sync();
    if (lane < BH) {
        float v[BW];
#pragma unroll
        for (int x = 0; x < BW; ++x) v[x] = llf[lane * LS + x];
        fdct<BW>(v, sl);
#pragma unroll
        for (int x = 0; x < BW; ++x) llf[lane * LS + x] = v[x];
    }
    wave_lds_sync();
    if (lane < BW) {
        float v[BH];
#pragma unroll
        for (int y = 0; y < BH; ++y) v[y] = llf[y * LS + lane];
        fdct<BH>(v, sl);
        constexpr int sy = 5 - __builtin_ctz(BH), sx = 5 - __builtin_ctz(BW);
#pragma unroll
        for (int y = 0; y < BH; ++y) llf[y * LS + lane] = v[y] / (kScaleF[y << sy] * kScaleF[lane << sx]);
    }
    wave_lds_sync();

    const float mul_base = 65536.0f / (a.global_scale * (float)(int32_t)e.z);
    if (lane < H) {
        const int y = lane;
        float d[W];
        {
            float4 m[W / 4];
            load_mrow<W>(a.dequant + a.deq_off_v[TYPE * 3 + c] + y * W, m);
            const float bias = c == 0 ? a.quant_bias[0] : (c == 1 ? a.quant_bias[1] : a.quant_bias[2]);
            const float qms = c == 0 ? a.qm_scale[0] : (c == 1 ? a.qm_scale[1] : a.qm_scale[2]);
            dequant_row<W>(raw, m, bias, a.quant_bias_numerator, lut, mul_base * qms, d);
        }
        if (c != 1) {
            float4 m[W / 4];
            load_mrow<W>(a.dequant + a.deq_off_v[TYPE * 3 + 1] + y * W, m);
            float yd[W];
            dequant_row<W>(rawy, m, a.quant_bias[1], a.quant_bias_numerator, lut, mul_base * a.qm_scale[1], yd);
            const CflRow k = cfl_row<W>(a, cx * 8, cy * 8 + (uint32_t)y);
            const bool any_straddle = __builtin_amdgcn_ballot_w64(k.split < W) != 0;
            cfl_apply<W>(d, yd, c == 0 ? k.kx0 : k.kb0, c == 0 ? k.kx1 : k.kb1, k.split, any_straddle);
        }
        if (y < BH) {
#pragma unroll
            for (int x = 0; x < BW; ++x) d[x] = llf[y * LS + x];
        }
        idct<W>(d, sl);
        float* row = T + y * S;
#pragma unroll
        for (int x = 0; x < W; ++x) row[x] = d[x];
    }
    wave_lds_sync();
    if (lane < W) {
        const float* col = T + lane;
        float v[H];
#pragma unroll
        for (int y = 0; y < H; ++y) v[y] = col[y * S];
        idct<H>(v, sl);
        float* dst = pixc + (size_t)(cy * 8) * a.pstride + cx * 8 + lane;
#pragma unroll
        for (int y = 0; y < H; ++y) dst[(size_t)y * a.pstride] = v[y];
    }
}

// ---------------------------------------------------------------------------------------------
// V8: the ten special 8x8 transforms (jxl-render/src/vardct/generic/transform.rs:14-219).  One
// lane per (varblock, channel) with the whole 8x8 block in registers: lanes 3b, 3b+1, 3b+2 hold
// X, Y, B of varblock b (21 varblocks per wave), so chroma-from-luma reaches the Y lane with one
// DPP wave shift per sample.  The host sorts this class by transform type, which makes the
// dispatch below (nearly) wave-uniform.
__device__ __forceinline__ float lane_shr1(float v) {   // value held by lane - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_shl1(float v) {   // value held by lane + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

typedef float Blk8[8][8];  // [y][x]

template <int SIZE>
__device__ __forceinline__ void aux_idct2(Blk8& b) {
    constexpr int n = SIZE / 2;
    float s[SIZE][SIZE];
#pragma unroll
    for (int y = 0; y < n; ++y)
#pragma unroll
        for (int x = 0; x < n; ++x) {
            float c00 = b[y][x], c01 = b[y][x + n], c10 = b[y + n][x], c11 = b[y + n][x + n];
            s[2 * y][2 * x] = c00 + c01 + c10 + c11;
            s[2 * y][2 * x + 1] = c00 + c01 - c10 - c11;
            s[2 * y + 1][2 * x] = c00 - c01 + c10 - c11;
            s[2 * y + 1][2 * x + 1] = c00 - c01 - c10 + c11;
        }
#pragma unroll
    for (int y = 0; y < SIZE; ++y)
#pragma unroll
        for (int x = 0; x < SIZE; ++x) b[y][x] = s[y][x];
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

__device__ __forceinline__ void transform_dct2(Blk8& b) {
    aux_idct2<2>(b);
    aux_idct2<4>(b);
    aux_idct2<8>(b);
}

__device__ __forceinline__ void transform_dct4(Blk8& b, const SecLarge& sl) {
    aux_idct2<2>(b);
    Blk8 out;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float m[4][4];  // scratch.get_mut(iy, ix) = coeff(x + ix*2, y + iy*2): row ix, col iy
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) m[ix][iy] = b[y + iy * 2][x + ix * 2];
            idct2d_4x4(m, sl);
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) out[y * 4 + iy][x * 4 + ix] = m[iy][ix];
        }
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) b[y][x] = out[y][x];
}

__device__ __forceinline__ void transform_hornuss(Blk8& b) {
    aux_idct2<2>(b);
    Blk8 out;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float s[16];
#pragma unroll
            for (int iy = 0; iy < 4; ++iy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) s[iy * 4 + ix] = b[y + iy * 2][x + ix * 2];
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
        for (int x = 0; x < 8; ++x) b[y][x] = out[y][x];
}

template <bool TR>
__device__ __forceinline__ void transform_dct4x8(Blk8& b, const SecLarge& sl) {
    float coeff0 = b[0][0], coeff1 = b[1][0];
    b[0][0] = coeff0 + coeff1;
    b[1][0] = coeff0 - coeff1;
    Blk8 scratch;
#pragma unroll
    for (int idx = 0; idx < 2; ++idx) {
        float m[4][8];
#pragma unroll
        for (int iy = 0; iy < 4; ++iy)
#pragma unroll
            for (int ix = 0; ix < 8; ++ix) m[iy][ix] = b[iy * 2 + idx][ix];
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
            if (TR) b[x][y] = scratch[y][x];   // c(y, x) = scratch[y][x]
            else b[y][x] = scratch[y][x];
        }
}

template <int N>
__device__ __forceinline__ void transform_afv(Blk8& b, const SecLarge& sl) {
    constexpr int flip_x = N % 2, flip_y = N / 2;
    float coeff_afv[16];
    coeff_afv[0] = (b[0][0] + b[0][1] + b[1][0]) * 4.0f;
#pragma unroll
    for (int idx = 1; idx < 16; ++idx) coeff_afv[idx] = b[2 * (idx / 4)][2 * (idx % 4)];
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
        for (int ix = 0; ix < 4; ++ix) m44[ix][iy] = b[2 * iy][2 * ix + 1];
    m44[0][0] = b[0][0] - b[0][1] + b[1][0];
    idct2d_4x4(m44, sl);

    float m48[4][8];
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int ix = 0; ix < 8; ++ix) m48[iy][ix] = b[2 * iy + 1][ix];
    m48[0][0] = b[0][0] - b[1][0];
    idct2d_8x4(m48, sl);

#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
        constexpr int dummy = 0; (void)dummy;
        const int afv_y = flip_y == 0 ? iy : 3 - iy;
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
            const int afv_x = flip_x == 0 ? ix : 3 - ix;
            b[flip_y * 4 + iy][flip_x * 4 + ix] = samples_afv[afv_y * 4 + afv_x];
        }
    }
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) b[flip_y * 4 + iy][(1 - flip_x) * 4 + ix] = m44[iy][ix];
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int ix = 0; ix < 8; ++ix) b[(1 - flip_y) * 4 + iy][ix] = m48[iy][ix];
}

constexpr int kSpecialPerWave = 21;  // varblocks per wave: 63 lanes

__global__ __launch_bounds__(64) void transform_special_kernel(TransformArgs a, const uint4* __restrict__ entries,
                                                               uint32_t count) {
    __shared__ float lut[kLutWords];
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < kLutWords / 64; ++i) lut[i * 64 + lane] = a.deq_lut[i * 64 + lane];
    wave_lds_sync();
    const SecLarge sl{a.sec64, a.sec128, a.sec256};
    const int bl = lane / 3, c = lane - bl * 3;
    const uint32_t bi = blockIdx.x * kSpecialPerWave + bl;
    const bool valid = lane < kSpecialPerWave * 3 && bi < count;
    const uint4 e = entries[valid ? bi : (count - 1)];
    const uint32_t cx = e.x & 0xffffu, cy = e.x >> 16;
    const uint32_t type = e.y;
    const size_t cell = (size_t)cy * a.w8 + cx;

    // ---- V4: the lane's 64 coefficients are one contiguous 256-byte run of the tiled layout
    Blk8 b;
    {
        const int32_t* src = a.coeff + ((cell * 3 + (uint32_t)c) << 6);
        const float* mat = a.dequant + a.deq_off[type * 3 + c];
        const float bias = c == 0 ? a.quant_bias[0] : (c == 1 ? a.quant_bias[1] : a.quant_bias[2]);
        const float qms = c == 0 ? a.qm_scale[0] : (c == 1 ? a.qm_scale[1] : a.qm_scale[2]);
        const float mul = 65536.0f / (a.global_scale * (float)(int32_t)e.z) * qms;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            int4 raw[2];
            float4 m[2];
            raw[0] = *reinterpret_cast<const int4*>(src + y * 8);
            raw[1] = *reinterpret_cast<const int4*>(src + y * 8 + 4);
            m[0] = *reinterpret_cast<const float4*>(mat + y * 8);
            m[1] = *reinterpret_cast<const float4*>(mat + y * 8 + 4);
            dequant_row<8>(raw, m, bias, a.quant_bias_numerator, lut, mul, b[y]);
        }
    }
    // ---- V5: an 8x8 varblock lies inside one 64x64 tile; lanes (X, Y, B) = (3b, 3b+1, 3b+2)
    {
        const uint32_t ti = (cy >> 3) * a.w64 + (cx >> 3);
        const float k = c == 0 ? a.kx_map[ti] : (c == 2 ? a.kb_map[ti] : 0.0f);
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float from_r = lane_shl1(b[y][x]);  // X lane: the Y lane is lane + 1
                const float from_l = lane_shr1(b[y][x]);  // B lane: the Y lane is lane - 1
                const float yv = c == 0 ? from_r : from_l;
                if (c != 1) b[y][x] += k * yv;
            }
    }
    // ---- V6: 1x1 LF block -> coefficient (0, 0) (transform_common.rs:40-48)
    {
        const float* lfp = c == 0 ? a.lf[0] : (c == 1 ? a.lf[1] : a.lf[2]);
        b[0][0] = lfp[cell];
    }
    // ---- V8 (transform.rs:225-240 dispatch)
    switch (type) {
        case JXLGPU_DCT2: transform_dct2(b); break;
        case JXLGPU_DCT4: transform_dct4(b, sl); break;
        case JXLGPU_HORNUSS: transform_hornuss(b); break;
        case JXLGPU_DCT4X8: transform_dct4x8<false>(b, sl); break;
        case JXLGPU_DCT8X4: transform_dct4x8<true>(b, sl); break;
        case JXLGPU_AFV0: transform_afv<0>(b, sl); break;
        case JXLGPU_AFV1: transform_afv<1>(b, sl); break;
        case JXLGPU_AFV2: transform_afv<2>(b, sl); break;
        case JXLGPU_AFV3: transform_afv<3>(b, sl); break;
        default: break;
    }
    if (valid) {
        float* pixc = c == 0 ? a.pix[0] : (c == 1 ? a.pix[1] : a.pix[2]);
        float* dst = pixc + (size_t)(cy * 8) * a.pstride + cx * 8;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            *reinterpret_cast<float4*>(dst + (size_t)y * a.pstride) = make_float4(b[y][0], b[y][1], b[y][2], b[y][3]);
            *reinterpret_cast<float4*>(dst + (size_t)y * a.pstride + 4) = make_float4(b[y][4], b[y][5], b[y][6], b[y][7]);
        }
    }
}

template <int W, int H>
void launch_tk64(hipStream_t s, const TransformArgs& a, const uint4* entries, uint32_t count) {
    transform_kernel64<W, H><<<dim3(count, 3), 64, 0, s>>>(a, entries);
}

}  // namespace

void launch_big_blocks(hipStream_t s, const TransformArgs& a, const uint4* entries, uint32_t count);

// Varblocks per work item of the row-lane kernels (host side of RCfg<W, H>::NBI)
int transform_items_nbi(int cls) {
    switch (cls) {
        case CLS_DCT8: return RCfg<8, 8>::NBI;
        case CLS_16x16: return RCfg<16, 16>::NBI;
        case CLS_8x16: return RCfg<8, 16>::NBI;
        case CLS_16x8: return RCfg<16, 8>::NBI;
        case CLS_32x32: return RCfg<32, 32>::NBI;
        case CLS_8x32: return RCfg<8, 32>::NBI;
        case CLS_32x8: return RCfg<32, 8>::NBI;
        case CLS_16x32: return RCfg<16, 32>::NBI;
        case CLS_32x16: return RCfg<32, 16>::NBI;
        default: return 1;
    }
}

// One launch per family: `classes` in launch order (long items first).
hipError_t launch_transform_rows(hipStream_t s, int family, const TransformArgs& a, const uint4* entries,
                                 const uint32_t class_first[CLS_COUNT], const uint32_t list_count[CLS_COUNT]) {
    static const int kFamA[] = {CLS_16x16, CLS_8x16, CLS_16x8, CLS_DCT8};
    static const int kFamB[] = {CLS_32x32, CLS_16x32, CLS_32x16, CLS_8x32, CLS_32x8};
    const int* classes = family == 0 ? kFamA : kFamB;
    const int n = family == 0 ? 4 : 5;
    ItemTable it;
    memset(&it, 0, sizeof(it));
    uint32_t items = 0;
    for (int i = 0; i < n; ++i) {
        const int cls = classes[i];
        if (!list_count[cls]) continue;
        const uint32_t k = it.n_classes++;
        it.begin[k] = items;
        it.cls[k] = (uint32_t)cls;
        it.first_entry[k] = class_first[cls];
        it.count[k] = list_count[cls];
        items += ceil_div(list_count[cls], (uint32_t)transform_items_nbi(cls));
    }
    it.begin[it.n_classes] = items;
    if (!items) return hipSuccess;
    const uint32_t wgs = ceil_div(items, 4);
    if (family == 0) {
        transform_rows_kernel<0><<<wgs, 256, (kLutWords + 4 * kWaveWordsA) * sizeof(float), s>>>(a, it, entries);
    } else {
        transform_rows_kernel<1><<<wgs, 256, (kLutWords + 4 * kWaveWordsB) * sizeof(float), s>>>(a, it, entries);
    }
    return hipGetLastError();
}

void launch_transform_class(hipStream_t s, int cls, const TransformArgs& a, const uint4* entries,
                            uint32_t count) {
    if (count == 0) return;
    switch (cls) {
        case CLS_SPECIAL8:
            transform_special_kernel<<<ceil_div(count, kSpecialPerWave), 64, 0, s>>>(a, entries, count);
            break;
        case CLS_64x64: launch_tk64<64, 64>(s, a, entries, count); break;
        case CLS_32x64: launch_tk64<32, 64>(s, a, entries, count); break;
        case CLS_64x32: launch_tk64<64, 32>(s, a, entries, count); break;
        case CLS_BIG: launch_big_blocks(s, a, entries, count); break;
        default: break;
    }
}
