// V4-V8 on gfx950: HF dequantisation, chroma-from-luma, LF -> LLF injection and the inverse
// variable-size DCT (jxl-render/src/vardct/mod.rs:442-682, transform_common.rs:11-75,
// generic/{dct,transform}.rs): one 192-thread workgroup (a wave per channel) per work item.
//
// Layout the kernels rely on: the i32 coefficients live in HBM as 8x8 cells,
// channel-interleaved — cell (cx, cy) is 3 x 64 words {X, Y, B}, each 8 rows of 8 — so any
// varblock, whatever its shape or alignment, reads whole 256-byte runs and never shares a cache
// line with a neighbour of another shape (the row-major planes of round 1 fetched every line
// once per shape class that touched it: 2.4x read amplification).
//
// A work item is NBI = 64 / min(W, H) varblocks of one shape, handled by one 192-thread workgroup:
// wave 0 takes channel Y, wave 1 X, wave 2 B.  Each wave makes
//   row pass   : lane = one row (W coefficients) -> 16-byte loads straight into registers ->
//                dequantise -> (X, B: chroma-from-luma from the Y wave's dequantised rows, handed
//                over through LDS at the one mid-item barrier) -> 1-D IDCT in registers ->
//                one ds_write_b32 per sample into the wave's padded LDS tile (bank-conflict free);
//   column pass: lane = one column: H ds_read_b32 -> 1-D IDCT -> H dword stores.
// The tile belongs to the wave; LDS operations of one wave execute in order, so the two passes are
// ordered by compiler fences only.  Operation order inside every butterfly is the reference's
// (dct_device.h), hence results are bit-identical to the CPU path.
#include "common.h"
#include "dct_device.h"

#include "afv_basis.inc"

#include <mutex>

namespace {

// Phase timing for tools/ (make PROF=1 -> libjxlgpu_prof.so): s_memtime stamps per wave, summed per
// phase into TransformArgs::prof.  Compiled out of the product library.
#ifdef JXL_TR_PROFILE
#define TR_STAMP_DECL unsigned long long tr_t[12]; int tr_n = 0;
#define TR_STAMP(drain)                                                   \
    do {                                                                  \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        tr_t[tr_n++] = __builtin_amdgcn_s_memtime();                      \
        asm volatile("" ::: "memory");                                    \
    } while (0)
#define TR_STAMP_FLUSH(slot)                                                                      \
    do {                                                                                          \
        if (a.prof && (threadIdx.x & 63) == 0 && blockIdx.x % 41 == 0) {                                                \
            for (int i_ = 1; i_ < tr_n; ++i_) atomicAdd(a.prof + (slot) * 16 + i_, tr_t[i_] - tr_t[i_ - 1]); \
            atomicAdd(a.prof + (slot) * 16, 1ull);                                                \
        }                                                                                         \
    } while (0)
#else
#define TR_STAMP_DECL
#define TR_STAMP(drain) do {} while (0)
#define TR_STAMP_FLUSH(slot) do {} while (0)
#endif

// value of channel c (wave-uniform) out of three: by VALUE on purpose — `c == 0 ? s.x[0] : s.x[1]`
// on struct members is an lvalue select, i.e. a dynamically indexed load, which forces a by-value
// argument struct into scratch memory
template <typename T>
__device__ __forceinline__ T pick3(int c, T v0, T v1, T v2) {
    return c == 0 ? v0 : (c == 1 ? v1 : v2);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dequant_hf_varblock_grouped inner loop (vardct/mod.rs:527-537) with `qbn / q` from a table of
// quant_bias_numerator / k (k = |q| < 256, built on the host with the same correctly rounded f32
// division): qbn / q == sign(q) * (qbn / |q|) exactly.  Branch-free; the caller tracks the largest
// |q| of its rows in `amax` and redoes a row with dequant_div when it reaches 256 (never, in
// practice: a d1 stream keeps |q| in the tens).
__device__ __forceinline__ float dequant_lut(int32_t qn, float quant_bias, const float* qlut, float m, float mul,
                                             uint32_t& amax) {
    float q = (float)qn;
    const uint32_t aq = (uint32_t)max(qn, -qn);
    amax = max(amax, aq);
    const float t = qlut[min(aq, 255u)];
    const float big = q - __builtin_copysignf(t, q);
    const float small = q * quant_bias;
    q = aq <= 1u ? small : big;
    q *= m;
    q *= mul;
    return q;
}
__device__ __forceinline__ float dequant_div(int32_t qn, float quant_bias, float qbn, float m, float mul) {
    float q = (float)qn;
    if (fabsf(q) <= 1.0f) q *= quant_bias;
    else q -= qbn / q;
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
    static constexpr int LLF_WORDS = NBI * BW * BH;              // one channel
    static constexpr int WAVE_WORDS = T_WORDS + LLF_WORDS;       // per wave (= per channel)
    // The dequantised Y rows, [pass][x][lane] = RP * W * 64 = NBI * H * W words, are handed to the X and
    // B waves INSIDE those waves' own (still unused) tiles: no extra LDS.
    static constexpr int WG_WORDS = 256 + 3 * WAVE_WORDS;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int kLutWords = 256;
// One row of W coefficients of channel c at cell row (cy + y / 8), in-cell row y % 8.
template <int W>
__device__ __forceinline__ void load_row(const TransformArgs& a, uint32_t cx, uint32_t cy, int y, int c,
                                         int4 (&raw)[W / 4]) {
    // 32-bit lane offset from the uniform base (frames are checked at upload to fit)
    const uint32_t off = ((((cy + (uint32_t)(y >> 3)) * a.w8 + cx) * 3 + (uint32_t)c) << 6) + (uint32_t)((y & 7) << 3);
#pragma unroll
    for (int i = 0; i < W / 8; ++i) {
        raw[2 * i] = *reinterpret_cast<const int4*>(a.coeff + off + i * 192);
        raw[2 * i + 1] = *reinterpret_cast<const int4*>(a.coeff + off + i * 192 + 4);
    }
}

template <int W>
__device__ __forceinline__ void load_mrow(const float* mrow, float4 (&m)[W / 4]) {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) m[i] = *reinterpret_cast<const float4*>(mrow + 4 * i);
}

template <int W>
__device__ __forceinline__ void dequant_row(const int4 (&raw)[W / 4], const float4 (&m)[W / 4], float bias,
                                            const float* lut, float mul, float (&d)[W], uint32_t& amax) {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) {
        d[4 * i + 0] = dequant_lut(raw[i].x, bias, lut, m[i].x, mul, amax);
        d[4 * i + 1] = dequant_lut(raw[i].y, bias, lut, m[i].y, mul, amax);
        d[4 * i + 2] = dequant_lut(raw[i].z, bias, lut, m[i].z, mul, amax);
        d[4 * i + 3] = dequant_lut(raw[i].w, bias, lut, m[i].w, mul, amax);
    }
}

// Cold path: the row again, from memory, with the division (some |q| >= 256).
template <int W>
__device__ __forceinline__ void dequant_row_div(const TransformArgs& a, uint32_t cx, uint32_t cy, int y, int c,
                                                const float* mrow, float bias, float mul, float (&d)[W]) {
    int4 raw[W / 4];
    float4 m[W / 4];
    load_row<W>(a, cx, cy, y, c, raw);
    load_mrow<W>(mrow, m);
#pragma unroll
    for (int i = 0; i < W / 4; ++i) {
        d[4 * i + 0] = dequant_div(raw[i].x, bias, a.quant_bias_numerator, m[i].x, mul);
        d[4 * i + 1] = dequant_div(raw[i].y, bias, a.quant_bias_numerator, m[i].y, mul);
        d[4 * i + 2] = dequant_div(raw[i].z, bias, a.quant_bias_numerator, m[i].z, mul);
        d[4 * i + 3] = dequant_div(raw[i].w, bias, a.quant_bias_numerator, m[i].w, mul);
    }
}

// Workgroup barrier that only drains LDS traffic: the HBM loads of the NEXT work item stay in
// flight across it (a __syncthreads() would make the compiler wait for vmcnt(0) first).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// What a lane needs to know about one work item: where its rows / columns / LF block live.
template <int RP, int CP>
struct Geo {
    uint32_t rcx[RP], rcy[RP];
    float rmul[RP];
    uint32_t ccx[CP], ccy[CP];
    uint32_t lpos;  // cell of varblock `lane` (lanes < NBI): its LF samples
};
// Everything a lane reads from HBM for one work item (issued one item ahead).
template <int RP, int W, int BW, int BH>
struct Pre {
    int4 raw[RP][W / 4];
    float k0[RP], k1[RP];
    int split[RP];
    float lfv[BH][BW];
};

// ---------------------------------------------------------------------------------------------
// A persistent workgroup walks the work items wg, wg + n_wgs, ... of ONE shape class; wave 0 takes
// channel Y, wave 1 X, wave 2 B.  The loop is software-pipelined by hand: at the top of iteration i
// the entries of item i+2 and the coefficients of item i+1 are requested, so both HBM round trips of
// an item hide behind a whole iteration of butterflies.
template <int W, int H, bool PIPE>
__device__ __forceinline__ void run_class(const TransformArgs& a, const uint4* __restrict__ ent_class,
                                          uint32_t count, uint32_t wg, uint32_t n_wgs, float* __restrict__ lds,
                                          int wave, int lane) {
    using C = RCfg<W, H>;
    constexpr int BW = C::BW, BH = C::BH, NBI = C::NBI, RP = C::RP, CP = C::CP, S = C::S, BS = C::BS;
    constexpr int TYPE = type_of<W, H>();
    constexpr bool PAR_LLF = BW >= 4 && BH >= 4;  // LF -> LLF forward DCT spread over lanes (32 / 64-px shapes)
    float* lut = lds;
    float* T = lds + kLutWords + wave * C::WAVE_WORDS;
    float* llf = T + C::T_WORDS;
    float* Tx = lds + kLutWords + 1 * C::WAVE_WORDS;   // tiles of the X and B waves: the Y wave parks its
    float* Tb = lds + kLutWords + 2 * C::WAVE_WORDS;   // dequantised rows there for chroma-from-luma
    const SecLarge sl{a.sec64, a.sec128, a.sec256};
    const int c = wave == 0 ? 1 : (wave == 1 ? 0 : 2);  // wave 0 = Y: the other two wait for its rows
    const float* lfp = pick3<const float*>(c, a.lf[0], a.lf[1], a.lf[2]);
    const float bias = pick3(c, a.quant_bias[0], a.quant_bias[1], a.quant_bias[2]);
    const float qms = pick3(c, a.qm_scale[0], a.qm_scale[1], a.qm_scale[2]);
    const float* mat = a.dequant + pick3(c, a.deq_off_v[TYPE * 3], a.deq_off_v[TYPE * 3 + 1], a.deq_off_v[TYPE * 3 + 2]);
    const float* kmap = pick3<const float*>(c, a.kx_map, a.kb_map, a.kb_map);
    const uint32_t n_items = (count + NBI - 1) / NBI;

    // ---- lane constants: which row / column of which varblock of an item this lane owns
    int ry[RP], rblk[RP], cxi[CP], cblk[CP];
#pragma unroll
    for (int p = 0; p < RP; ++p) { rblk[p] = (p * 64 + lane) / H; ry[p] = (p * 64 + lane) % H; }
#pragma unroll
    for (int q = 0; q < CP; ++q) { cblk[q] = (q * 64 + lane) / W; cxi[q] = (q * 64 + lane) % W; }
    // the dequantisation matrix rows of this lane never change (one shape class per workgroup):
    // kept in registers across items when they are short, re-read (L2) per item otherwise
    // Long rows (W >= 32): the matrix row is consumed 8 values at a time so the compiler is free to
    // keep as few or as many of those (L2-hit) loads in flight as the register budget allows.
    constexpr bool STREAM = W >= 32;
    constexpr bool HOIST_M = PIPE && !STREAM && RP * W <= 32;
    float4 mh[HOIST_M ? RP : 1][W / 4];
    if constexpr (HOIST_M) {
#pragma unroll
        for (int p = 0; p < RP; ++p) load_mrow<W>(mat + ry[p] * W, mh[p]);
    }

    auto load_geo = [&](uint32_t item, Geo<RP, CP>& g) __attribute__((always_inline)) {
        const uint4* ent = ent_class + (size_t)item * NBI;
        const int nv = (int)min((uint32_t)NBI, count - item * NBI);
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            const uint4 e = ent[min(rblk[p], nv - 1)];  // lanes past the last varblock shadow it (never stored)
            g.rcx[p] = e.x & 0xffffu;
            g.rcy[p] = e.x >> 16;
            g.rmul[p] = 65536.0f / (a.global_scale * (float)(int32_t)e.z);
        }
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            const uint32_t pos = ent[min(cblk[q], nv - 1)].x;
            g.ccx[q] = pos & 0xffffu;
            g.ccy[q] = pos >> 16;
        }
        g.lpos = ent[min(PAR_LLF ? lane / BH : lane, nv - 1)].x;  // the varblock whose LF samples this lane loads
    };
    auto issue_loads = [&](const Geo<RP, CP>& g, Pre<RP, W, BW, BH>& pr) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            load_row<W>(a, g.rcx[p], g.rcy[p], ry[p], c, pr.raw[p]);
            // chroma-from-luma factor of a row that may straddle a 64-px tile column (varblocks need
            // not be aligned): samples x < split take k0, the rest k1 (mod.rs:589-600)
            const uint32_t px0 = g.rcx[p] * 8, py = g.rcy[p] * 8 + (uint32_t)ry[p];
            pr.k0[p] = kmap[(py >> 6) * a.w64 + (px0 >> 6)];
            pr.k1[p] = kmap[(py >> 6) * a.w64 + ((px0 + W - 1) >> 6)];
            pr.split[p] = 64 - (int)(px0 & 63u);
        }
        if constexpr (PAR_LLF) {
            if (lane < NBI * BH) {  // lane = (varblock, LF row): one row of BW samples
                const size_t cell = (size_t)((g.lpos >> 16) + (uint32_t)(lane % BH)) * a.w8 + (g.lpos & 0xffffu);
#pragma unroll
                for (int x = 0; x < BW; ++x) pr.lfv[0][x] = lfp[cell + x];
            }
        } else if (lane < NBI) {
            const size_t cell = (size_t)(g.lpos >> 16) * a.w8 + (g.lpos & 0xffffu);
#pragma unroll
            for (int y = 0; y < BH; ++y)
#pragma unroll
                for (int x = 0; x < BW; ++x) pr.lfv[y][x] = lfp[cell + (size_t)y * a.w8 + x];
        }
    };

    // Pipeline: at the top of iteration i the entries of item i+2 and the coefficients of item i+1
    // are requested; item i's coefficients were requested one iteration ago.
    // (PIPE = false: no run-ahead, the occupancy of the launch hides the two round trips instead;
    //  one work item per workgroup then, and the registers of the look-ahead buffers are saved.)
    Geo<RP, CP> g, gn, gnn;
    Pre<RP, W, BW, BH> pr, prn;
    TR_STAMP_DECL
    TR_STAMP(false);
    load_geo(wg, g);
    TR_STAMP(true);   // 1: entries arrived
    issue_loads(g, pr);
    if constexpr (PIPE) {
        if (wg + n_wgs < n_items) load_geo(wg + n_wgs, gn);
    }
    for (int i = threadIdx.x; i < kLutWords; i += 192) lut[i] = a.deq_lut[i];
    TR_STAMP(true);   // 2: coefficients, LF, table arrived
    lds_barrier();  // the table of quant_bias_numerator / k is in place
    TR_STAMP(false);  // 3: barrier 0

    for (uint32_t item = wg; item < n_items; item += n_wgs) {
        const int nvalid = (int)min((uint32_t)NBI, count - item * NBI);
        const bool has_next = item + n_wgs < n_items;        // workgroup-uniform
        const bool has_next2 = item + 2 * n_wgs < n_items;
        if constexpr (PIPE) {
            if (has_next) issue_loads(gn, prn);
            if (has_next2) load_geo(item + 2 * n_wgs, gnn);
        }

        // ---- V6 first half: LF -> lowest-frequency coefficients of this channel
        //      (transform_common.rs:40-66: copy the BW x BH LF samples, forward DCT, divide by the
        //      scale_f products); one lane per varblock, parked in LDS for the row lanes.
        if constexpr (PAR_LLF) {
            // >= 4 x 4 LF samples: the generic branch of fdct2d_small (rows, then columns, independent 1-D
            // transforms) spread over lanes — BH row lanes, the wave's llf patch as the transposition
            // buffer, BW column lanes — instead of one lane doing BH + BW transforms and BW * BH divisions
            // (measured: ~7 k of the ~36 k cycles of a 64 x 64 item)
            if (lane < NBI * BH) {
                float r[BW];
#pragma unroll
                for (int x = 0; x < BW; ++x) r[x] = pr.lfv[0][x];
                fdct<BW>(r, sl);
                float* dst = llf + lane * BW;  // [varblock][y][x]
#pragma unroll
                for (int x = 0; x < BW; ++x) dst[x] = r[x];
            }
            wave_lds_sync();
            if (lane < NBI * BW) {
                const int blk = lane / BW, x = lane % BW;
                float* colp = llf + blk * (BW * BH) + x;
                float col[BH];
#pragma unroll
                for (int y = 0; y < BH; ++y) col[y] = colp[y * BW];
                fdct<BH>(col, sl);
                constexpr int sy = 5 - __builtin_ctz(BH), sx = 5 - __builtin_ctz(BW);
                float fx = kScaleF[0];  // kScaleF[x << sx] through selects on immediates: a table lookup here is a
#pragma unroll                  // global load whose latency nothing hides
                for (int i = 1; i < BW; ++i) fx = x == i ? kScaleF[i << sx] : fx;
#pragma unroll
                for (int y = 0; y < BH; ++y) colp[y * BW] = col[y] / (kScaleF[y << sy] * fx);
            }
        } else if (lane < NBI) {
            float v[BH][BW];
#pragma unroll
            for (int y = 0; y < BH; ++y)
#pragma unroll
                for (int x = 0; x < BW; ++x) v[y][x] = pr.lfv[y][x];
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
        // ---- V4: dequantise every row of this lane
        float d[RP][W];
        bool straddle = false;
        float k0[RP], k1[RP];
        int split[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            const float* mrow = mat + ry[p] * W;
            const float mul = g.rmul[p] * qms;
            uint32_t amax = 0;
            if constexpr (STREAM) {
#pragma unroll
                for (int i = 0; i < W / 8; ++i) {
                    const float4 m0 = *reinterpret_cast<const float4*>(mrow + 8 * i), m1 = *reinterpret_cast<const float4*>(mrow + 8 * i + 4);
                    const int4 r0 = pr.raw[p][2 * i], r1 = pr.raw[p][2 * i + 1];
                    d[p][8 * i + 0] = dequant_lut(r0.x, bias, lut, m0.x, mul, amax);
                    d[p][8 * i + 1] = dequant_lut(r0.y, bias, lut, m0.y, mul, amax);
                    d[p][8 * i + 2] = dequant_lut(r0.z, bias, lut, m0.z, mul, amax);
                    d[p][8 * i + 3] = dequant_lut(r0.w, bias, lut, m0.w, mul, amax);
                    d[p][8 * i + 4] = dequant_lut(r1.x, bias, lut, m1.x, mul, amax);
                    d[p][8 * i + 5] = dequant_lut(r1.y, bias, lut, m1.y, mul, amax);
                    d[p][8 * i + 6] = dequant_lut(r1.z, bias, lut, m1.z, mul, amax);
                    d[p][8 * i + 7] = dequant_lut(r1.w, bias, lut, m1.w, mul, amax);
                }
            } else if constexpr (HOIST_M) {
                dequant_row<W>(pr.raw[p], mh[p], bias, lut, mul, d[p], amax);
            } else {
                float4 m[W / 4];
                load_mrow<W>(mrow, m);
                dequant_row<W>(pr.raw[p], m, bias, lut, mul, d[p], amax);
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(amax > 255u) != 0, 0)) {
                if (amax > 255u) dequant_row_div<W>(a, g.rcx[p], g.rcy[p], ry[p], c, mrow, bias, mul, d[p]);
            }
            k0[p] = pr.k0[p]; k1[p] = pr.k1[p]; split[p] = pr.split[p];
            straddle |= split[p] < W;
        }
        const bool any_straddle = __builtin_amdgcn_ballot_w64(straddle) != 0;
        TR_STAMP(false);  // 4: LLF + dequant

        // ---- V5: the Y wave publishes its dequantised rows; X and B add k * Y (mod.rs:589-600)
        if (c == 1) {
#pragma unroll
            for (int p = 0; p < RP; ++p)
#pragma unroll
                for (int x = 0; x < W; ++x) {
                    Tx[(p * W + x) * 64 + lane] = d[p][x];
                    Tb[(p * W + x) * 64 + lane] = d[p][x];
                }
        }
        lds_barrier();
        TR_STAMP(false);  // 5: Y rows published + barrier 1
        if (c != 1) {
            // all of these reads precede (program order, one wave) the row writes into the same tile
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                if (!any_straddle) {  // wave-uniform: every row lies inside one 64-px tile column
#pragma unroll
                    for (int x = 0; x < W; ++x) d[p][x] += k0[p] * T[(p * W + x) * 64 + lane];
                } else {
#pragma unroll
                    for (int x = 0; x < W; ++x) d[p][x] += (x < split[p] ? k0[p] : k1[p]) * T[(p * W + x) * 64 + lane];
                }
            }
            wave_lds_sync();
        }
        // ---- LLF patch, 1-D IDCT of every row (dct.rs:93-96), rows into the wave's tile
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            if (rblk[p] >= nvalid) continue;
            const int y = ry[p];
            if (y < BH) {
                const float* src = llf + rblk[p] * (BW * BH) + y * BW;
#pragma unroll
                for (int x = 0; x < BW; ++x) d[p][x] = src[x];
            }
            idct<W>(d[p], sl);
            float* row = T + rblk[p] * BS + y * S;
#pragma unroll
            for (int x = 0; x < W; ++x) row[x] = d[p][x];
        }
        wave_lds_sync();
        TR_STAMP(false);  // 6: CfL + row IDCT + tile writes
        // ---- column passes: 1-D IDCT of the column (dct.rs:109-130), samples straight to HBM
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            if (cblk[q] >= nvalid) continue;
            const float* col = T + cblk[q] * BS + cxi[q];
            float v[H];
#pragma unroll
            for (int y = 0; y < H; ++y) v[y] = col[y * S];
            idct<H>(v, sl);
            // cell-tiled output: this lane's column runs down H / 8 cells, 8 words apart inside each.
            // 32-bit lane offset + uniform row base: one address register for all H stores.
            const uint32_t off = (((g.ccy[q] * a.w8 + g.ccx[q] + (uint32_t)(cxi[q] >> 3)) * 3 + (uint32_t)c) << 6) + (uint32_t)(cxi[q] & 7);
            const uint32_t cell_row = a.w8 * 192;
#pragma unroll
            for (int y = 0; y < H; ++y) {
                float* rowbase = a.pix + (size_t)((uint32_t)(y >> 3) * cell_row + (uint32_t)((y & 7) << 3));  // uniform
                rowbase[off] = v[y];
            }
        }
        TR_STAMP(false);  // 7: column IDCT + stores issued
        TR_STAMP(true);   // 8: stores drained
        TR_STAMP_FLUSH(C::MINWH >= 64 ? 3 : (C::MINWH >= 32 ? 2 : (C::MINWH >= 16 ? 1 : 0)));  // = the launch family
#ifdef JXL_TR_PROFILE
        tr_n = 0;
        TR_STAMP(false);
        TR_STAMP(false); TR_STAMP(false); TR_STAMP(false);
#endif
        if (has_next) {
            lds_barrier();  // X and B are done with this item's Y rows before the next ones land
            if constexpr (PIPE) {
                g = gn;
                gn = gnn;
                pr = prn;
            } else {
                load_geo(item + n_wgs, g);
                issue_loads(g, pr);
            }
        }
    }
}

// One launch per register class, so that the long-row shapes do not set the occupancy of the
// short ones: FAMILY 0 = 8x8 (<= 64 VGPRs: eight waves per SIMD keep ~64 KiB of coefficient
// loads in flight per CU, what 8 TB/s x ~2 us of loaded latency asks for), 1 = the 16-px shapes,
// 2 = the 32-px shapes, 3 = the 64-px shapes (whole rows of 64 in registers).
template <int FAMILY>
struct FamCfg;
template <> struct FamCfg<0> { static constexpr int WAVES = 8; static constexpr int WORDS = RCfg<8, 8>::WG_WORDS; };
template <> struct FamCfg<1> {
    static constexpr int WAVES = 4;
    static constexpr int WORDS = cmax(RCfg<16, 16>::WG_WORDS, cmax(RCfg<8, 16>::WG_WORDS, RCfg<16, 8>::WG_WORDS));
};
template <> struct FamCfg<2> {
    static constexpr int WAVES = 2;
    static constexpr int WORDS = cmax(cmax(cmax(RCfg<32, 32>::WG_WORDS, RCfg<8, 32>::WG_WORDS),
                                           cmax(RCfg<32, 8>::WG_WORDS, RCfg<16, 32>::WG_WORDS)), RCfg<32, 16>::WG_WORDS);
};
template <> struct FamCfg<3> {
    static constexpr int WAVES = 1;
    static constexpr int WORDS = cmax(cmax(RCfg<64, 64>::WG_WORDS, RCfg<32, 64>::WG_WORDS), RCfg<64, 32>::WG_WORDS);
};

// Field-by-field copy of a frame's TransformArgs out of its device block (a memcpy of the whole
// struct is not scalarised by the compiler and ends up in scratch; this form becomes SGPRs).
__device__ __forceinline__ TransformArgs load_transform_args(FrameDevC fd) {
    TransformArgs a;
    a.coeff = fd->tr.coeff; a.pix = fd->tr.pix;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.lf[c] = fd->tr.lf[c]; a.qm_scale[c] = fd->tr.qm_scale[c]; a.quant_bias[c] = fd->tr.quant_bias[c];
    }
    a.kind = fd->tr.kind; a.hf_mul = fd->tr.hf_mul; a.kx_map = fd->tr.kx_map; a.kb_map = fd->tr.kb_map;
    a.dequant = fd->tr.dequant; a.deq_off = fd->tr.deq_off;
#pragma unroll
    for (int i = 0; i < 27 * 3; ++i) a.deq_off_v[i] = fd->tr.deq_off_v[i];
    a.sec64 = fd->tr.sec64; a.sec128 = fd->tr.sec128; a.sec256 = fd->tr.sec256;
    a.pstride = fd->tr.pstride; a.w8 = fd->tr.w8; a.h8 = fd->tr.h8; a.w64 = fd->tr.w64;
    a.global_scale = fd->tr.global_scale; a.quant_bias_numerator = fd->tr.quant_bias_numerator;
    a.big_tmp = fd->tr.big_tmp; a.deq_lut = fd->tr.deq_lut;
#ifdef JXL_TR_PROFILE
    a.prof = fd->tr.prof;
#endif
    return a;
}

template <int FAMILY, bool PIPE>
__device__ __forceinline__ void items_body(const TransformArgs& a, uint32_t cls, const uint4* __restrict__ ent_class,
                                           uint32_t count, uint32_t wg, uint32_t n_wgs, float* lds) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
#define RUN(W, H) run_class<W, H, PIPE>(a, ent_class, count, wg, n_wgs, lds, wave, lane);
    if constexpr (FAMILY == 0) {
        RUN(8, 8)
    } else if constexpr (FAMILY == 1) {
        switch (cls) {
            case CLS_16x16: RUN(16, 16) break;
            case CLS_8x16: RUN(8, 16) break;
            case CLS_16x8: RUN(16, 8) break;
            default: break;
        }
    } else if constexpr (FAMILY == 2) {
        switch (cls) {
            case CLS_32x32: RUN(32, 32) break;
            case CLS_8x32: RUN(8, 32) break;
            case CLS_32x8: RUN(32, 8) break;
            case CLS_16x32: RUN(16, 32) break;
            case CLS_32x16: RUN(32, 16) break;
            default: break;
        }
    } else {
        switch (cls) {
            case CLS_64x64: RUN(64, 64) break;
            case CLS_32x64: RUN(32, 64) break;
            case CLS_64x32: RUN(64, 32) break;
            default: break;
        }
    }
#undef RUN
}

template <int FAMILY, bool PIPE>
__global__ __launch_bounds__(192, FamCfg<FAMILY>::WAVES) void transform_items_kernel(TransformArgs a, ClassTable ct,
                                                                                     const uint4* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint32_t k = 0;
    while (blockIdx.x >= ct.wg_begin[k + 1]) ++k;
    const uint32_t wg = blockIdx.x - ct.wg_begin[k], n_wgs = ct.wg_begin[k + 1] - ct.wg_begin[k];
    items_body<FAMILY, PIPE>(a, ct.cls[k], entries + ct.first_entry[k], ct.count[k], wg, n_wgs, lds);
}

// Batched form: blockIdx.y picks the frame; its arguments come from the frame's device block.
template <int FAMILY>
__global__ __launch_bounds__(192, FamCfg<FAMILY>::WAVES) void transform_items_batch_kernel(FrameBatch b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FrameDevC fd = (FrameDevC)b.f[blockIdx.y];
    const uint32_t ncls = fd->ct[FAMILY].n_classes;
    if (blockIdx.x >= fd->ct[FAMILY].wg_begin[ncls]) return;
    uint32_t k = 0;
    while (blockIdx.x >= fd->ct[FAMILY].wg_begin[k + 1]) ++k;
    const uint32_t wg = blockIdx.x - fd->ct[FAMILY].wg_begin[k];
    const uint32_t n_wgs = fd->ct[FAMILY].wg_begin[k + 1] - fd->ct[FAMILY].wg_begin[k];
    const TransformArgs a = load_transform_args(fd);
    items_body<FAMILY, false>(a, fd->ct[FAMILY].cls[k], fd->entries + fd->ct[FAMILY].first_entry[k],
                              fd->ct[FAMILY].count[k], wg, n_wgs, lds);
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

__device__ __forceinline__ void special_body(const TransformArgs& a, const uint4* __restrict__ entries, uint32_t count,
                                             float* lut) {
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
        const float bias = pick3(c, a.quant_bias[0], a.quant_bias[1], a.quant_bias[2]);
        const float qms = pick3(c, a.qm_scale[0], a.qm_scale[1], a.qm_scale[2]);
        const float mul = 65536.0f / (a.global_scale * (float)(int32_t)e.z) * qms;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            int4 raw[2];
            float4 m[2];
            raw[0] = *reinterpret_cast<const int4*>(src + y * 8);
            raw[1] = *reinterpret_cast<const int4*>(src + y * 8 + 4);
            m[0] = *reinterpret_cast<const float4*>(mat + y * 8);
            m[1] = *reinterpret_cast<const float4*>(mat + y * 8 + 4);
            uint32_t amax = 0;
            dequant_row<8>(raw, m, bias, lut, mul, b[y], amax);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(amax > 255u) != 0, 0)) {
                if (amax > 255u) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        b[y][4 * i + 0] = dequant_div(raw[i].x, bias, a.quant_bias_numerator, m[i].x, mul);
                        b[y][4 * i + 1] = dequant_div(raw[i].y, bias, a.quant_bias_numerator, m[i].y, mul);
                        b[y][4 * i + 2] = dequant_div(raw[i].z, bias, a.quant_bias_numerator, m[i].z, mul);
                        b[y][4 * i + 3] = dequant_div(raw[i].w, bias, a.quant_bias_numerator, m[i].w, mul);
                    }
                }
            }
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
        const float* lfp = pick3<const float*>(c, a.lf[0], a.lf[1], a.lf[2]);
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
        float* dst = a.pix + ((cell * 3 + (uint32_t)c) << 6);  // one contiguous 256-byte run of the tiled output
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            *reinterpret_cast<float4*>(dst + y * 8) = make_float4(b[y][0], b[y][1], b[y][2], b[y][3]);
            *reinterpret_cast<float4*>(dst + y * 8 + 4) = make_float4(b[y][4], b[y][5], b[y][6], b[y][7]);
        }
    }
}

__global__ __launch_bounds__(64) void transform_special_kernel(TransformArgs a, const uint4* __restrict__ entries,
                                                               uint32_t count) {
    __shared__ float lut[kLutWords];
    special_body(a, entries, count, lut);
}

__global__ __launch_bounds__(64) void transform_special_batch_kernel(FrameBatch b) {
    __shared__ float lut[kLutWords];
    const FrameDevC fd = (FrameDevC)b.f[blockIdx.y];
    const uint32_t count = fd->special_count;
    if (blockIdx.x * kSpecialPerWave >= count) return;
    const TransformArgs a = load_transform_args(fd);
    special_body(a, fd->entries + fd->special_first, count, lut);
}


}  // namespace

void launch_big_blocks(hipStream_t s, const TransformArgs& a, const uint4* entries, uint32_t count);

// Varblocks per work item (host side of RCfg<W, H>::NBI)
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
        case CLS_64x64: return RCfg<64, 64>::NBI;
        case CLS_32x64: return RCfg<32, 64>::NBI;
        case CLS_64x32: return RCfg<64, 32>::NBI;
        default: return 1;
    }
}

// Workgroups of one family launch.  wgs_per_cu = 0: one workgroup per work item (the default: the
// occupancy of the launch hides latency).  > 0: persistent workgroups with run-ahead; each shape
// class gets a share of num_cus * wgs_per_cu proportional to its work (pixels, weighted by the depth
// of its butterflies), never more than it has items.  Classes with long items come first.
void build_class_table(int family, const uint32_t class_first[CLS_COUNT], const uint32_t list_count[CLS_COUNT],
                       uint32_t num_cus, int wgs_per_cu, ClassTable* out) {
    static const int kFam0[] = {CLS_DCT8};
    static const int kFam1[] = {CLS_16x16, CLS_8x16, CLS_16x8};
    static const int kFam2[] = {CLS_32x32, CLS_16x32, CLS_32x16, CLS_8x32, CLS_32x8};
    static const int kFam3[] = {CLS_64x64, CLS_32x64, CLS_64x32};
    static const double kCost[CLS_COUNT] = {/*8x8*/ 1.0, 0, /*16x16*/ 1.25, /*8x16*/ 1.12, /*16x8*/ 1.12, /*32x32*/ 1.5,
                                            /*8x32*/ 1.25, /*32x8*/ 1.25, /*16x32*/ 1.38, /*32x16*/ 1.38,
                                            /*64x64*/ 1.75, /*32x64*/ 1.62, /*64x32*/ 1.62, 0};
    static const int kArea[CLS_COUNT] = {64, 0, 256, 128, 128, 1024, 256, 256, 512, 512, 4096, 2048, 2048, 0};
    static const int* const kFam[4] = {kFam0, kFam1, kFam2, kFam3};
    static const int kFamN[4] = {1, 3, 5, 3};
    ClassTable& ct = *out;
    memset(&ct, 0, sizeof(ct));
    const int* classes = kFam[family];
    const int n = kFamN[family];
    const uint32_t budget = wgs_per_cu > 0 ? std::max(1u, num_cus) * (uint32_t)wgs_per_cu : 0xffffffu;
    double work[5] = {}, total = 0;
    uint32_t items[5] = {};
    for (int i = 0; i < n; ++i) {
        const int cls = classes[i];
        items[i] = ceil_div(list_count[cls], (uint32_t)transform_items_nbi(cls));
        work[i] = (double)list_count[cls] * kArea[cls] * kCost[cls];
        total += work[i];
    }
    uint32_t wgs = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i]) continue;
        const int cls = classes[i];
        uint32_t share = (uint32_t)(budget * (work[i] / total) + 0.5);
        share = std::min(items[i], std::max(1u, share));
        const uint32_t k = ct.n_classes++;
        ct.wg_begin[k] = wgs;
        ct.cls[k] = (uint32_t)cls;
        ct.first_entry[k] = class_first[cls];
        ct.count[k] = list_count[cls];
        wgs += share;
    }
    ct.wg_begin[ct.n_classes] = wgs;
    for (uint32_t k = ct.n_classes + 1; k < 6; ++k) ct.wg_begin[k] = 0xffffffffu;
}

namespace {
template <int F>
void set_lds_attr() {
    constexpr size_t bytes = FamCfg<F>::WORDS * sizeof(float);
    if constexpr (bytes > 65536) {  // > 64 KiB of dynamic LDS needs the attribute; idempotent and thread-safe
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&transform_items_kernel<F, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&transform_items_kernel<F, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&transform_items_batch_kernel<F>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        });
    }
}
}  // namespace

hipError_t launch_transform_items(hipStream_t s, int family, const TransformArgs& a, const uint4* entries,
                                  const uint32_t class_first[CLS_COUNT], const uint32_t list_count[CLS_COUNT],
                                  uint32_t num_cus, int wgs_per_cu) {
    if (family < 0 || family > 3) return hipErrorInvalidValue;
    ClassTable ct;
    build_class_table(family, class_first, list_count, num_cus, wgs_per_cu, &ct);
    const uint32_t wgs = ct.wg_begin[ct.n_classes];
    if (!ct.n_classes || !wgs) return hipSuccess;
    const bool pipe = wgs_per_cu > 0;
#define LAUNCH(F)                                                                          \
    do {                                                                                   \
        constexpr size_t bytes = FamCfg<F>::WORDS * sizeof(float);                         \
        set_lds_attr<F>();                                                                 \
        if (pipe) transform_items_kernel<F, true><<<wgs, 192, bytes, s>>>(a, ct, entries); \
        else transform_items_kernel<F, false><<<wgs, 192, bytes, s>>>(a, ct, entries);     \
    } while (0)
    switch (family) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
#undef LAUNCH
    return hipGetLastError();
}

// All four families + the special 8x8 family of n frames: long work items first, the bulk last.
// `side` (optional): the 64 / 32-px families and the special one run there beside the 16 / 8-px families
// on `s` (disjoint varblocks; the caller forks / joins the streams) — for small batches, whose launches
// are short enough for their tails to show.
hipError_t launch_transform_batch(hipStream_t s, hipStream_t side, const FrameBatch& b, uint32_t n, const uint32_t max_wgs[4],
                                  uint32_t max_special) {
    if (!side) side = s;
#define LAUNCHB(F, ST)                                                                                       \
    if (max_wgs[F]) {                                                                                       \
        set_lds_attr<F>();                                                                                  \
        transform_items_batch_kernel<F><<<dim3(max_wgs[F], n), 192, FamCfg<F>::WORDS * sizeof(float), ST>>>(b); \
    }
    LAUNCHB(3, side)
    LAUNCHB(2, side)
    if (max_special)
        transform_special_batch_kernel<<<dim3(ceil_div(max_special, kSpecialPerWave), n), 64, 0, side>>>(b);
    LAUNCHB(1, s)
    LAUNCHB(0, s)
#undef LAUNCHB
    return hipGetLastError();
}

void launch_transform_class(hipStream_t s, int cls, const TransformArgs& a, const uint4* entries,
                            uint32_t count) {
    if (count == 0) return;
    switch (cls) {
        case CLS_SPECIAL8:
            transform_special_kernel<<<ceil_div(count, kSpecialPerWave), 64, 0, s>>>(a, entries, count);
            break;
        case CLS_BIG: launch_big_blocks(s, a, entries, count); break;
        default: break;
    }
}
