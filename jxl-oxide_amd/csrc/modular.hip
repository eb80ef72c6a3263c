// Modular stage for gfx950: predictor application where it is separable (Gradient, single leaf),
// inverse Squeeze / RCT / Palette on the integer channel buffers, and the int -> float tail.
//
// Integer, wrapping arithmetic in the sample type S (i16 or i32) exactly as the reference
// (jxl-modular/src/transform/{squeeze,rct,palette}.rs); results are bit-exact.
//
// Layout: every channel is a full-resolution buffer holding the squeezed pyramid as nested
// sub-rectangles (avg = left/top half, residual = right/bottom half, transform.rs:343-437).  The
// CPU code undoes a step in place through a per-row scratch copy; here each step reads the avg and
// residual rectangles from where they live and writes the merged rectangle into another of three
// working copies of the buffer (the merged rectangle overlaps both inputs, so in place would race
// across lanes).  Parallelism is what the transform allows: one lane per row (horizontal step) or
// per column (vertical step), each lane a serial chain because `tendency` needs the previous
// output sample.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <unordered_map>
#include <type_traits>
#include <vector>

#include "common.h"

bool clip_region(const JxlGpuRegion* r, uint32_t w, uint32_t h, PixRect* out);
int finish_render_region(jxlgpu_ctx* ctx, jxlgpu_frame* f, float* cur[3], uint32_t stride, const PixRect& r, const JxlGpuOut* out);
bool fused_post_supported(const jxlgpu_ctx* ctx, const jxlgpu_frame* f, bool gabor, int epf_iters);
bool fused_int_input_supported(const jxlgpu_ctx* ctx, const jxlgpu_frame* f, int epf_iters, const void* const in[3], uint32_t in_stride, size_t elem);
int run_post_stages(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuFilterParams& fp,
                    uint32_t up_factor, float* cur[3], uint32_t* cur_stride, uint32_t* ow, uint32_t* oh,
                    bool tiled_in, const PixRect* region);
int finish_render(jxlgpu_ctx* ctx, jxlgpu_frame* f, float* cur[3], uint32_t stride, uint32_t ow, uint32_t oh,
                  const JxlGpuOut* out);
int upload_post_params(jxlgpu_ctx* ctx, jxlgpu_frame* f, const JxlGpuUpsampling& up);
void fill_color_args_public(const JxlGpuColorParams& cp, ColorArgs* c);
const char* color_params_unsupported(const JxlGpuColorParams& cp);
void launch_upsample_jpeg_rows(hipStream_t s, const float* in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, int hshift, int vshift,
                               float* out, uint32_t out_stride, uint32_t width, uint32_t height);

namespace {

// ---------------------------------------------------------------- device: Squeeze
// tendency_i32 / tendency_i16, squeeze.rs:1104-1172 (Wrapping<S> == truncate to S after every op)
template <typename S>
__device__ __forceinline__ S tendency(S a, S b, S c) {
    if (a >= b && b >= c) {
        S x = (S)((S)((S)((S)((S)(4 * a) - (S)(3 * c)) - b) + 6) / 12);
        if ((S)(x - (x & 1)) > (S)(2 * (S)(a - b))) x = (S)((S)(2 * (S)(a - b)) + 1);
        if ((S)(x + (x & 1)) > (S)(2 * (S)(b - c))) x = (S)(2 * (S)(b - c));
        return x;
    } else if (a <= b && b <= c) {
        S x = (S)((S)((S)((S)((S)(4 * a) - (S)(3 * c)) - b) - 6) / 12);
        if ((S)(x + (x & 1)) < (S)(2 * (S)(a - b))) x = (S)((S)(2 * (S)(a - b)) - 1);
        if ((S)(x - (x & 1)) < (S)(2 * (S)(b - c))) x = (S)(2 * (S)(b - c));
        return x;
    }
    return 0;
}

// i32 needs 4*a etc. to wrap without UB: go through unsigned
template <>
__device__ __forceinline__ int32_t tendency<int32_t>(int32_t a, int32_t b, int32_t c) {
    auto W = [](uint32_t v) { return (int32_t)v; };
    const uint32_t ua = (uint32_t)a, ub = (uint32_t)b, uc = (uint32_t)c;
    if (a >= b && b >= c) {
        int32_t x = W(4u * ua - 3u * uc - ub + 6u) / 12;
        int32_t ab2 = W(2u * (ua - ub)), bc2 = W(2u * (ub - uc));
        if (W((uint32_t)x - (uint32_t)(x & 1)) > ab2) x = W((uint32_t)ab2 + 1u);
        if (W((uint32_t)x + (uint32_t)(x & 1)) > bc2) x = bc2;
        return x;
    } else if (a <= b && b <= c) {
        int32_t x = W(4u * ua - 3u * uc - ub - 6u) / 12;
        int32_t ab2 = W(2u * (ua - ub)), bc2 = W(2u * (ub - uc));
        if (W((uint32_t)x + (uint32_t)(x & 1)) < ab2) x = W((uint32_t)ab2 - 1u);
        if (W((uint32_t)x - (uint32_t)(x & 1)) < bc2) x = bc2;
        return x;
    }
    return 0;
}

template <typename S>
struct Wrap;
template <>
struct Wrap<int16_t> {
    static __device__ __forceinline__ int16_t add(int16_t a, int16_t b) { return (int16_t)(a + b); }
    static __device__ __forceinline__ int16_t sub(int16_t a, int16_t b) { return (int16_t)(a - b); }
    static __device__ __forceinline__ int16_t mul(int16_t a, int16_t b) { return (int16_t)(a * b); }
};
template <>
struct Wrap<int32_t> {
    static __device__ __forceinline__ int32_t add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
    static __device__ __forceinline__ int32_t sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
    static __device__ __forceinline__ int32_t mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
};

struct SqzArgs {
    const void* avg;   // top-left of the avg rectangle
    const void* res;   // top-left of the residual rectangle
    void* out;         // top-left of the merged rectangle (another working copy)
    uint32_t avg_stride, res_stride, out_stride;  // elements
    uint32_t width, height;                       // merged size
};

template <typename S>
__device__ __forceinline__ void squeeze_pair(S residu, S next_avg, S& avg, S& prev, S& first, S& second) {
    S diff = Wrap<S>::add(residu, tendency<S>(prev, avg, next_avg));
    first = Wrap<S>::add(avg, (S)(diff / 2));
    second = Wrap<S>::sub(first, diff);
    avg = next_avg;
    prev = second;
}

// inverse_v_*_base, squeeze.rs:803-862: one lane per column, coalesced row accesses, PF rows of
// residuals / averages in flight per lane (the chain itself is serial in y).
template <typename S>
__global__ __launch_bounds__(64) void squeeze_v_kernel(SqzArgs a) {
    constexpr int PF = 16;
    uint32_t x = blockIdx.x * 64 + threadIdx.x;
    if (x >= a.width) return;
    const S* avgp = (const S*)a.avg + x;
    const S* resp = (const S*)a.res + x;
    S* out = (S*)a.out + x;
    const uint32_t avg_h = (a.height + 1) / 2, pairs = a.height / 2;
    S avg = avgp[0];
    S top = avg;
    uint32_t y = 0;
    for (; y + PF <= pairs; y += PF) {
        S r[PF], n[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            r[k] = resp[(size_t)(y + k) * a.res_stride];
            uint32_t ny = y + k + 1;
            n[k] = ny < avg_h ? avgp[(size_t)ny * a.avg_stride] : (S)0;
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            S next_avg = (y + k + 1 < avg_h) ? n[k] : avg;
            S first, second;
            squeeze_pair<S>(r[k], next_avg, avg, top, first, second);
            out[(size_t)(2 * (y + k)) * a.out_stride] = first;
            out[(size_t)(2 * (y + k) + 1) * a.out_stride] = second;
        }
    }
    for (; y < pairs; ++y) {
        S r = resp[(size_t)y * a.res_stride];
        S next_avg = (y + 1 < avg_h) ? avgp[(size_t)(y + 1) * a.avg_stride] : avg;
        S first, second;
        squeeze_pair<S>(r, next_avg, avg, top, first, second);
        out[(size_t)(2 * y) * a.out_stride] = first;
        out[(size_t)(2 * y + 1) * a.out_stride] = second;
    }
    if (a.height & 1) out[(size_t)(a.height - 1) * a.out_stride] = avgp[(size_t)(avg_h - 1) * a.avg_stride];
}

// inverse_h_*_base, squeeze.rs:59-120: one lane per row.  Each lane streams its own row with
// 16-byte vector loads / stores when the rectangles are 16-byte aligned (every 128-byte line a
// lane touches is consumed over the next iterations out of L1/L2), scalar accesses otherwise.
template <typename S, bool VEC>
__global__ __launch_bounds__(64) void squeeze_h_kernel(SqzArgs a) {
    constexpr int N = 16 / sizeof(S);            // pairs per step on the vector path
    using V = int4;
    const uint32_t y = blockIdx.x * 64 + threadIdx.x;
    if (y >= a.height) return;
    const uint32_t avg_w = (a.width + 1) / 2, pairs = a.width / 2;
    const S* avgp = (const S*)a.avg + (size_t)y * a.avg_stride;
    const S* resp = (const S*)a.res + (size_t)y * a.res_stride;
    S* outp = (S*)a.out + (size_t)y * a.out_stride;
    S avg = avgp[0];
    S left = avg;
    uint32_t x = 0;
    if constexpr (VEC) {
        union Pack { V v; S s[N]; };
        // avg[x+1 .. x+N] is needed for the N pairs at x: keep the next vector of averages loaded
        Pack cur_a;
        if (avg_w >= (uint32_t)N) cur_a.v = *reinterpret_cast<const V*>(avgp);
        // the loop needs the next vector of averages fully inside the avg rectangle; the last
        // couple of vectors of a row fall through to the scalar loops below
        if (avg_w < (uint32_t)N) cur_a.v = V{0, 0, 0, 0};
        for (; x + N <= pairs && x + 2 * N <= avg_w; x += N) {
            Pack r, nxt_a, o0, o1;
            r.v = *reinterpret_cast<const V*>(resp + x);
            nxt_a.v = *reinterpret_cast<const V*>(avgp + x + N);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                uint32_t nx = x + k + 1;
                S cand = k + 1 < N ? cur_a.s[k + 1] : nxt_a.s[0];
                S next_avg = nx < avg_w ? cand : avg;
                S first, second;
                squeeze_pair<S>(r.s[k], next_avg, avg, left, first, second);
                if (2 * k < N) { o0.s[2 * k] = first; o0.s[2 * k + 1] = second; }
                else { o1.s[2 * k - N] = first; o1.s[2 * k + 1 - N] = second; }
            }
            *reinterpret_cast<V*>(outp + 2 * x) = o0.v;
            *reinterpret_cast<V*>(outp + 2 * x + N) = o1.v;
            cur_a.v = nxt_a.v;
        }
    }
    constexpr int PF = 8;
    for (; x + PF <= pairs; x += PF) {
        S r[PF], n[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            r[k] = resp[x + k];
            n[k] = (x + k + 1 < avg_w) ? avgp[x + k + 1] : (S)0;
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            S next_avg = (x + k + 1 < avg_w) ? n[k] : avg;
            S first, second;
            squeeze_pair<S>(r[k], next_avg, avg, left, first, second);
            outp[2 * (x + k)] = first;
            outp[2 * (x + k) + 1] = second;
        }
    }
    for (; x < pairs; ++x) {
        S next_avg = (x + 1 < avg_w) ? avgp[x + 1] : avg;
        S first, second;
        squeeze_pair<S>(resp[x], next_avg, avg, left, first, second);
        outp[2 * x] = first;
        outp[2 * x + 1] = second;
    }
    if (a.width & 1) outp[a.width - 1] = avgp[avg_w - 1];
}

// ---------------------------------------------------------------------------------------------
// Segment-parallel Squeeze.  The inverse is a serial chain along the scan direction only through
// `prev` (the previous output sample, fed to `tendency`), and a wrong `prev` is forgotten within a
// few pairs (the error shrinks ~6x per pair and `tendency` is 0 off monotone runs).  So a row /
// column is cut into segments: every segment starts OV pairs early from a guessed `prev`, and
// records (a) the `prev` it had reached at its real start and (b) the `prev` it ends with.  A
// check kernel then walks the chain of segments: segment 0 is exact by construction, segment s is
// exact iff its (a) equals segment s-1's (b); a segment whose link fails is redone serially from the
// true state (squeeze_check3_kernel).  Bit-exactness never depends on the guess — only speed does.  This turns ~13 k long
// chains (8K image) into ~400 k short ones, which is what fills 256 CUs.
struct SegArgs {
    SqzArgs a;
    uint32_t seg_pairs;  // pairs per segment (multiple of the chunk size)
    uint32_t nseg;
    void* chk;           // S[nseg][2][lines]: [.][0] = prev at the segment's start, [.][1] = prev at its end
    uint32_t runin;      // 1 = start OV pairs early (normal); 0 = no run-in (tests: forces the serial fix-up)
};

template <typename S>
__device__ __forceinline__ void squeeze_v_seg_body(const SegArgs& g, uint32_t x, uint32_t seg) {
    constexpr int PF = 16, OV = 16;
    const SqzArgs& a = g.a;
    const S* avgp = (const S*)a.avg + x;
    const S* resp = (const S*)a.res + x;
    S* out = (S*)a.out + x;
    S* chk = (S*)g.chk;
    const uint32_t avg_h = (a.height + 1) / 2, pairs = a.height / 2;
    const uint32_t y_s = seg * g.seg_pairs;
    const uint32_t y_e = (seg + 1 == g.nseg) ? pairs : y_s + g.seg_pairs;
    uint32_t y = (seg == 0 || !g.runin) ? y_s : y_s - OV;
    S avg = avgp[(size_t)y * a.avg_stride];
    S top = y == 0 ? avg : avgp[(size_t)(y - 1) * a.avg_stride];  // guess: previous pair's average (exact at y == 0)
    for (; y + PF <= y_e; y += PF) {
        if (y == y_s) chk[((size_t)seg * 2 + 0) * a.width + x] = top;
        S r[PF], n[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            r[k] = resp[(size_t)(y + k) * a.res_stride];
            uint32_t ny = y + k + 1;
            n[k] = ny < avg_h ? avgp[(size_t)ny * a.avg_stride] : (S)0;
        }
        const bool store = y >= y_s;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            S next_avg = (y + k + 1 < avg_h) ? n[k] : avg;
            S first, second;
            squeeze_pair<S>(r[k], next_avg, avg, top, first, second);
            if (store) {
                out[(size_t)(2 * (y + k)) * a.out_stride] = first;
                out[(size_t)(2 * (y + k) + 1) * a.out_stride] = second;
            }
        }
    }
    for (; y < y_e; ++y) {  // tail of the last segment
        S r = resp[(size_t)y * a.res_stride];
        S next_avg = (y + 1 < avg_h) ? avgp[(size_t)(y + 1) * a.avg_stride] : avg;
        S first, second;
        squeeze_pair<S>(r, next_avg, avg, top, first, second);
        out[(size_t)(2 * y) * a.out_stride] = first;
        out[(size_t)(2 * y + 1) * a.out_stride] = second;
    }
    chk[((size_t)seg * 2 + 1) * a.width + x] = top;
    if (seg + 1 == g.nseg && (a.height & 1))
        out[(size_t)(a.height - 1) * a.out_stride] = avgp[(size_t)(avg_h - 1) * a.avg_stride];
}

// whole column, serially (the fix-up path and the small-rectangle path share this)
template <typename S>
__device__ __forceinline__ void squeeze_v_line(const SqzArgs& a, uint32_t x) {
    constexpr int PF = 16;
    const S* avgp = (const S*)a.avg + x;
    const S* resp = (const S*)a.res + x;
    S* out = (S*)a.out + x;
    const uint32_t avg_h = (a.height + 1) / 2, pairs = a.height / 2;
    S avg = avgp[0];
    S top = avg;
    uint32_t y = 0;
    for (; y + PF <= pairs; y += PF) {
        S r[PF], n[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            r[k] = resp[(size_t)(y + k) * a.res_stride];
            uint32_t ny = y + k + 1;
            n[k] = ny < avg_h ? avgp[(size_t)ny * a.avg_stride] : (S)0;
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            S next_avg = (y + k + 1 < avg_h) ? n[k] : avg;
            S first, second;
            squeeze_pair<S>(r[k], next_avg, avg, top, first, second);
            out[(size_t)(2 * (y + k)) * a.out_stride] = first;
            out[(size_t)(2 * (y + k) + 1) * a.out_stride] = second;
        }
    }
    for (; y < pairs; ++y) {
        S r = resp[(size_t)y * a.res_stride];
        S next_avg = (y + 1 < avg_h) ? avgp[(size_t)(y + 1) * a.avg_stride] : avg;
        S first, second;
        squeeze_pair<S>(r, next_avg, avg, top, first, second);
        out[(size_t)(2 * y) * a.out_stride] = first;
        out[(size_t)(2 * y + 1) * a.out_stride] = second;
    }
    if (a.height & 1) out[(size_t)(a.height - 1) * a.out_stride] = avgp[(size_t)(avg_h - 1) * a.avg_stride];
}

// Horizontal, vector path only (16-byte aligned rectangles): lane = row, blockIdx.y = segment.
template <typename S>
__device__ __forceinline__ void squeeze_h_seg_body(const SegArgs& g, uint32_t y, uint32_t seg) {
    constexpr int N = 16 / sizeof(S);
    constexpr int OV = 8;  // pairs of run-in (multiple of N)
    using V = int4;
    union Pack { V v; S s[N]; };
    const SqzArgs& a = g.a;
    const uint32_t avg_w = (a.width + 1) / 2, pairs = a.width / 2;
    const S* avgp = (const S*)a.avg + (size_t)y * a.avg_stride;
    const S* resp = (const S*)a.res + (size_t)y * a.res_stride;
    S* outp = (S*)a.out + (size_t)y * a.out_stride;
    S* chk = (S*)g.chk;
    const uint32_t x_s = seg * g.seg_pairs;
    const uint32_t x_e = (seg + 1 == g.nseg) ? pairs : x_s + g.seg_pairs;
    uint32_t x = (seg == 0 || !g.runin) ? x_s : x_s - OV;
    S avg = avgp[x];
    S left = x == 0 ? avg : avgp[x - 1];  // guess: previous pair's average (exact at x == 0)
    Pack cur_a;
    cur_a.v = *reinterpret_cast<const V*>(avgp + x);  // segments are only used when avg_w >= 2N
    auto vec_step = [&](const Pack& r, const Pack& nxt_a, uint32_t xx) __attribute__((always_inline)) {
        Pack o0, o1;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            S next_avg = k + 1 < N ? cur_a.s[k + 1] : nxt_a.s[0];  // xx+k+1 < avg_w holds for every vector step
            S first, second;
            squeeze_pair<S>(r.s[k], next_avg, avg, left, first, second);
            if (2 * k < N) { o0.s[2 * k] = first; o0.s[2 * k + 1] = second; }
            else { o1.s[2 * k - N] = first; o1.s[2 * k + 1 - N] = second; }
        }
        if (xx >= x_s) {
            *reinterpret_cast<V*>(outp + 2 * xx) = o0.v;
            *reinterpret_cast<V*>(outp + 2 * xx + N) = o1.v;
        }
        cur_a.v = nxt_a.v;
    };
    // the run-in (one vector step), then whole 64-byte pieces of the row per step: a lane walks its
    // own row, so every line it touches must be used up while it is still in the caches — with one
    // 16-byte vector per step a line was fetched again for each of its 8 vectors on an 8K image
    for (; x < x_s && x + 2 * N <= avg_w; x += N) {
        Pack r, nxt_a;
        r.v = *reinterpret_cast<const V*>(resp + x);
        nxt_a.v = *reinterpret_cast<const V*>(avgp + x + N);
        vec_step(r, nxt_a, x);
    }
    constexpr int VC = 4;  // vectors per step
    for (; x + VC * N <= x_e && x + (VC + 1) * N <= avg_w; x += VC * N) {
        if (x == x_s) chk[((size_t)seg * 2 + 0) * a.height + y] = left;
        Pack r[VC], na[VC];
#pragma unroll
        for (int v = 0; v < VC; ++v) {
            r[v].v = *reinterpret_cast<const V*>(resp + x + v * N);
            na[v].v = *reinterpret_cast<const V*>(avgp + x + (v + 1) * N);
        }
#pragma unroll
        for (int v = 0; v < VC; ++v) vec_step(r[v], na[v], x + v * N);
    }
    for (; x + N <= x_e && x + 2 * N <= avg_w; x += N) {
        if (x == x_s) chk[((size_t)seg * 2 + 0) * a.height + y] = left;
        Pack r, nxt_a;
        r.v = *reinterpret_cast<const V*>(resp + x);
        nxt_a.v = *reinterpret_cast<const V*>(avgp + x + N);
        vec_step(r, nxt_a, x);
    }
    for (; x < x_e; ++x) {  // scalar tail (only the last segment gets here with x >= x_s)
        if (x == x_s) chk[((size_t)seg * 2 + 0) * a.height + y] = left;
        S next_avg = (x + 1 < avg_w) ? avgp[x + 1] : avg;
        S first, second;
        squeeze_pair<S>(resp[x], next_avg, avg, left, first, second);
        if (x >= x_s) {
            outp[2 * x] = first;
            outp[2 * x + 1] = second;
        }
    }
    chk[((size_t)seg * 2 + 1) * a.height + y] = left;
    if (seg + 1 == g.nseg && (a.width & 1)) outp[a.width - 1] = avgp[avg_w - 1];
}

template <typename S>
__device__ __forceinline__ void squeeze_h_line(const SqzArgs& a, uint32_t y) {
    const uint32_t avg_w = (a.width + 1) / 2, pairs = a.width / 2;
    const S* avgp = (const S*)a.avg + (size_t)y * a.avg_stride;
    const S* resp = (const S*)a.res + (size_t)y * a.res_stride;
    S* outp = (S*)a.out + (size_t)y * a.out_stride;
    S avg = avgp[0];
    S left = avg;
    for (uint32_t x = 0; x < pairs; ++x) {
        S next_avg = (x + 1 < avg_w) ? avgp[x + 1] : avg;
        S first, second;
        squeeze_pair<S>(resp[x], next_avg, avg, left, first, second);
        outp[2 * x] = first;
        outp[2 * x + 1] = second;
    }
    if (a.width & 1) outp[a.width - 1] = avgp[avg_w - 1];
}

// Walk the chain of segments of every line; redo the line serially if a link is broken.
template <typename S, bool HORIZONTAL>
__global__ __launch_bounds__(64) void squeeze_check_kernel(SegArgs g, int* redo_count) {
    const uint32_t lines = HORIZONTAL ? g.a.height : g.a.width;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= lines) return;
    const S* chk = (const S*)g.chk;
    bool ok = true;
    for (uint32_t s = 1; s < g.nseg; ++s)
        ok &= chk[((size_t)s * 2 + 0) * lines + i] == chk[((size_t)(s - 1) * 2 + 1) * lines + i];
    if (ok) return;
    if (redo_count) atomicAdd(redo_count, 1);
    if (HORIZONTAL) squeeze_h_line<S>(g.a, i);
    else squeeze_v_line<S>(g.a, i);
}

// ---------------------------------------------------------------------------------------------
// Round-2 forms of the segment kernels: up to three channels of one squeeze step in ONE launch
// (blockIdx.z picks the channel: the channels of a step are independent), and a vertical kernel in
// which a lane owns NC adjacent columns instead of one: the one-column form moves 2 bytes per lane
// per load with 16-bit samples (a 128-byte line per wave instruction).  NC is a trade: wider loads
// against fewer, longer lanes (NC = 8 left an 8K step with 240 waves per channel: measured 2x slower
// than NC = 1); two 16-bit columns per lane keep ~1000 waves per channel and halve the memory
// instructions.
struct SegArgs3 {
    SegArgs g[3];
};

template <int BYTES> struct VecOf;
template <> struct VecOf<4> { using type = uint32_t; };
template <> struct VecOf<8> { using type = uint2; };
template <> struct VecOf<16> { using type = int4; };

template <typename S, int NC>
__global__ __launch_bounds__(64) void squeeze_v_seg3_kernel(SegArgs3 g3) {
    constexpr int PF = 16 / NC < 4 ? 4 : 16 / NC, OV = 16;
    using V = typename VecOf<NC * sizeof(S)>::type;
    union Pack { V v; S s[NC]; };
    const SegArgs& g = g3.g[blockIdx.z];
    const SqzArgs& a = g.a;
    const uint32_t nvec = a.width / NC, rem = a.width - nvec * NC;
    const uint32_t lane_id = blockIdx.x * 64 + threadIdx.x;
    const uint32_t seg = blockIdx.y;
    if (seg >= g.nseg || lane_id >= nvec + rem) return;
    S* chk = (S*)g.chk;
    const uint32_t avg_h = (a.height + 1) / 2, pairs = a.height / 2;
    const uint32_t y_s = seg * g.seg_pairs;
    const uint32_t y_e = (seg + 1 == g.nseg) ? pairs : y_s + g.seg_pairs;
    uint32_t y = (seg == 0 || !g.runin) ? y_s : y_s - OV;
    if (lane_id >= nvec) {
        // the last width % NC columns: one column per lane
        const uint32_t x = nvec * NC + (lane_id - nvec);
        const S* avgp = (const S*)a.avg + x;
        const S* resp = (const S*)a.res + x;
        S* out = (S*)a.out + x;
        S avg = avgp[(size_t)y * a.avg_stride];
        S top = y == 0 ? avg : avgp[(size_t)(y - 1) * a.avg_stride];
        for (; y < y_e; ++y) {
            if (y == y_s) chk[((size_t)seg * 2 + 0) * a.width + x] = top;
            S r = resp[(size_t)y * a.res_stride];
            S next_avg = (y + 1 < avg_h) ? avgp[(size_t)(y + 1) * a.avg_stride] : avg;
            S first, second;
            squeeze_pair<S>(r, next_avg, avg, top, first, second);
            if (y >= y_s) {
                out[(size_t)(2 * y) * a.out_stride] = first;
                out[(size_t)(2 * y + 1) * a.out_stride] = second;
            }
        }
        chk[((size_t)seg * 2 + 1) * a.width + x] = top;
        if (seg + 1 == g.nseg && (a.height & 1))
            out[(size_t)(a.height - 1) * a.out_stride] = avgp[(size_t)(avg_h - 1) * a.avg_stride];
        return;
    }
    const uint32_t x0 = lane_id * NC;
    const S* avgp = (const S*)a.avg + x0;
    const S* resp = (const S*)a.res + x0;
    S* out = (S*)a.out + x0;
    Pack avg, top;
    avg.v = *reinterpret_cast<const V*>(avgp + (size_t)y * a.avg_stride);
    if (y == 0) top = avg;  // exact; otherwise the guess: the previous pair's average
    else top.v = *reinterpret_cast<const V*>(avgp + (size_t)(y - 1) * a.avg_stride);
    auto step = [&](const Pack& r, const Pack& n, uint32_t yy, bool has_next) __attribute__((always_inline)) {
        Pack o0, o1;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            S next_avg = has_next ? n.s[j] : avg.s[j];
            squeeze_pair<S>(r.s[j], next_avg, avg.s[j], top.s[j], o0.s[j], o1.s[j]);
        }
        if (yy >= y_s) {
            *reinterpret_cast<V*>(out + (size_t)(2 * yy) * a.out_stride) = o0.v;
            *reinterpret_cast<V*>(out + (size_t)(2 * yy + 1) * a.out_stride) = o1.v;
        }
    };
    auto put_chk = [&](uint32_t which) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NC; ++j) chk[((size_t)seg * 2 + which) * a.width + x0 + j] = top.s[j];
    };
    for (; y + PF <= y_e; y += PF) {
        if (y == y_s) put_chk(0);
        Pack r[PF], n[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            r[k].v = *reinterpret_cast<const V*>(resp + (size_t)(y + k) * a.res_stride);
            const uint32_t ny = min(y + k + 1, avg_h - 1);
            n[k].v = *reinterpret_cast<const V*>(avgp + (size_t)ny * a.avg_stride);
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) step(r[k], n[k], y + k, y + k + 1 < avg_h);
    }
    for (; y < y_e; ++y) {
        if (y == y_s) put_chk(0);
        Pack r, n;
        r.v = *reinterpret_cast<const V*>(resp + (size_t)y * a.res_stride);
        n.v = *reinterpret_cast<const V*>(avgp + (size_t)min(y + 1, avg_h - 1) * a.avg_stride);
        step(r, n, y, y + 1 < avg_h);
    }
    put_chk(1);
    if (seg + 1 == g.nseg && (a.height & 1))
        *reinterpret_cast<V*>(out + (size_t)(a.height - 1) * a.out_stride) =
            *reinterpret_cast<const V*>(avgp + (size_t)(avg_h - 1) * a.avg_stride);
}

template <typename S, bool HORIZONTAL>
__global__ __launch_bounds__(64) void squeeze_seg3_kernel(SegArgs3 g3) {
    const SegArgs& g = g3.g[blockIdx.z];
    const uint32_t lines = HORIZONTAL ? g.a.height : g.a.width;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= lines || blockIdx.y >= g.nseg) return;
    if (HORIZONTAL) squeeze_h_seg_body<S>(g, i, blockIdx.y);
    else squeeze_v_seg_body<S>(g, i, blockIdx.y);
}

// Horizontal step, lane = SEGMENT (the default for 16-byte aligned rectangles).  In squeeze_h_seg_body a lane
// walks its own row, so one wave instruction touches 64 different 128-byte lines and uses 16 bytes of each:
// the step fetched 2.7x what it wrote (profiles/r02_modular_cfg3_pmc.txt).  Here adjacent lanes own adjacent
// 2N-pair pieces of ONE row (N = samples per 16-byte vector: 16 pairs of i16), so every load and store of
// the wave is a contiguous kilobyte; a lane starts 8 pairs early from a guessed `prev` as before, and the
// links BETWEEN the lanes of a wave are settled in the kernel: lane l compares the `prev` it had reached at
// its first own pair with the `prev` lane l - 1 ended with (one DPP shift) and redoes its piece from the true
// value until no link of the wave is open (nearly always zero rounds; inputs and outputs stay in registers,
// nothing is stored before the wave agrees).  Only the links between WAVES go through `chk` and
// squeeze_check3_kernel: a "segment" there is a wave's 64 pieces.  Rows with fewer than 33 pieces share a
// wave (lanes per row = pow2ceil(pieces)).
template <typename S>
__global__ __launch_bounds__(64) void squeeze_h_lanes_kernel(SegArgs3 g3) {
    constexpr int N = 16 / sizeof(S);
    constexpr int SP = 2 * N;   // pairs per lane
    constexpr int OV = 8;       // run-in pairs (a multiple of N for both sample types)
    using V = int4;
    union Pack { V v; S s[N]; };
    const SegArgs& g = g3.g[blockIdx.z];
    const SqzArgs& a = g.a;
    const uint32_t avg_w = (a.width + 1) / 2, pairs = a.width / 2;
    const uint32_t nls = (pairs + SP - 1) / SP;                     // pieces per row
    uint32_t lpr = 64;                                              // lanes per row
    if (nls <= 32) { lpr = 1; while (lpr < nls) lpr <<= 1; }
    const uint32_t rows_per_wave = 64 / lpr;
    const uint32_t lane = threadIdx.x;
    const uint32_t y = blockIdx.y * rows_per_wave + lane / lpr;
    const uint32_t in_row = lane & (lpr - 1);
    const uint32_t ls = blockIdx.x * 64 + in_row;                   // blockIdx.x > 0 only when lpr == 64
    if (blockIdx.x >= g.nseg || blockIdx.y * rows_per_wave >= a.height) return;   // wave-uniform
    const bool active = y < a.height && ls < nls;
    const uint32_t x_s = ls * SP;
    const uint32_t npairs = active ? min((uint32_t)SP, pairs - x_s) : 0u;
    const bool exact_start = ls == 0;                               // the row's first piece: prev = avg, no run-in
    const bool runin = !exact_start && g.runin != 0;
    const S* avgp = (const S*)a.avg + (size_t)(active ? y : 0) * a.avg_stride;
    const S* resp = (const S*)a.res + (size_t)(active ? y : 0) * a.res_stride;
    S* outp = (S*)a.out + (size_t)(active ? y : 0) * a.out_stride;

    // inputs of pairs [x_s - OV, x_s + SP): av[i] = avg sample x_s - OV + i (one more for the last pair's next_avg)
    S av[OV + SP + 1], rs[OV + SP];
    const bool vec = active && x_s + SP + N <= avg_w;
    if (vec) {
#pragma unroll
        for (int v = 0; v < (OV + SP) / N + 1; ++v) {
            Pack pk;
            pk.v = make_int4(0, 0, 0, 0);
            if (v * N >= OV || runin) pk.v = *reinterpret_cast<const V*>(avgp + x_s - OV + v * N);
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (v * N + k < OV + SP + 1) av[v * N + k] = pk.s[k];
        }
#pragma unroll
        for (int v = 0; v < (OV + SP) / N; ++v) {
            Pack pk;
            pk.v = make_int4(0, 0, 0, 0);
            if (v * N >= OV || runin) pk.v = *reinterpret_cast<const V*>(resp + x_s - OV + v * N);
#pragma unroll
            for (int k = 0; k < N; ++k) rs[v * N + k] = pk.s[k];
        }
    } else {
#pragma unroll
        for (int i = 0; i < OV + SP + 1; ++i) {
            const uint32_t xi = x_s - OV + i;   // wraps below zero for the first piece: masked by `i >= OV || runin`
            av[i] = (active && (i >= OV || runin) && xi < avg_w) ? avgp[xi] : (S)0;
        }
#pragma unroll
        for (int i = 0; i < OV + SP; ++i) {
            const uint32_t xi = x_s - OV + i;
            rs[i] = (active && (i >= OV || runin) && xi < pairs) ? resp[xi] : (S)0;
        }
    }
    S o[2 * SP];
    // pairs [FROM, OV + npairs) of the piece from `prev`; returns the prev at the end, *at_start = the prev at pair OV
    auto chain = [&](int from, S prev, S* at_start) __attribute__((always_inline)) -> S {
        S avg = av[from];
#pragma unroll
        for (int p = 0; p < OV + SP; ++p) {
            if (p < from) continue;
            if (p == OV) *at_start = prev;
            if ((uint32_t)(p - OV) < npairs || p < OV) {
                const S next_avg = (x_s - OV + p + 1 < avg_w) ? av[p + 1] : avg;
                S first, second;
                squeeze_pair<S>(rs[p], next_avg, avg, prev, first, second);
                if (p >= OV) { o[2 * (p - OV)] = first; o[2 * (p - OV) + 1] = second; }
            }
        }
        return prev;
    };
    S start_prev = 0, end_prev = 0;
    if (active) {
        if (runin) {
            // the previous pair's average; a run-in that starts at pair 0 (i32: the row's second piece) starts exactly
            const S guess = x_s == (uint32_t)OV ? av[0] : (S)avgp[x_s - OV - 1];
            end_prev = chain(0, guess, &start_prev);
        } else {
            const S p0 = exact_start ? av[OV] : (S)avgp[x_s - 1];  // runin == 0 (tests): a bare guess, the links settle it
            end_prev = chain(OV, p0, &start_prev);
        }
    }
    // settle the links inside the wave
    for (;;) {
        const S before = (S)__shfl_up((int)end_prev, 1);
        const bool open = active && in_row > 0 && start_prev != before;
        if (__builtin_amdgcn_ballot_w64(open) == 0) break;
        if (open) {
            S dummy;
            end_prev = chain(OV, before, &dummy);
            start_prev = before;
        }
    }
    // links between waves: what this wave began with / ended with
    if (active && g.nseg > 1) {
        S* chk = (S*)g.chk;
        if (in_row == 0) chk[((size_t)blockIdx.x * 2 + 0) * a.height + y] = start_prev;
        if (in_row == 63 || ls + 1 == nls) chk[((size_t)blockIdx.x * 2 + 1) * a.height + y] = end_prev;
    }
    if (vec) {
#pragma unroll
        for (int v = 0; v < 2 * SP / N; ++v) {
            Pack pk;
#pragma unroll
            for (int k = 0; k < N; ++k) pk.s[k] = o[v * N + k];
            *reinterpret_cast<V*>(outp + 2 * x_s + v * N) = pk.v;
        }
    } else if (active) {
#pragma unroll
        for (int p = 0; p < SP; ++p)
            if ((uint32_t)p < npairs) { outp[2 * (x_s + p)] = o[2 * p]; outp[2 * (x_s + p) + 1] = o[2 * p + 1]; }
    }
    if (active && ls + 1 == nls && (a.width & 1)) outp[a.width - 1] = avgp[avg_w - 1];
}

// One segment again, serially, from its true starting state (`prev` = what the segment before it
// really ended with); returns the `prev` it ends with.  Pairs [p_s, p_e) of line i.
template <typename S, bool HORIZONTAL>
__device__ __forceinline__ S squeeze_redo_range(const SqzArgs& a, uint32_t i, uint32_t p_s, uint32_t p_e, S prev, S stored_end) {
    const uint32_t len = HORIZONTAL ? a.width : a.height;
    const uint32_t avg_n = (len + 1) / 2;
    const size_t as = HORIZONTAL ? 1 : a.avg_stride, rs = HORIZONTAL ? 1 : a.res_stride, os = HORIZONTAL ? 1 : a.out_stride;
    const S* avgp = (const S*)a.avg + (HORIZONTAL ? (size_t)i * a.avg_stride : (size_t)i);
    const S* resp = (const S*)a.res + (HORIZONTAL ? (size_t)i * a.res_stride : (size_t)i);
    S* outp = (S*)a.out + (HORIZONTAL ? (size_t)i * a.out_stride : (size_t)i);
    S avg = avgp[(size_t)p_s * as];
    for (uint32_t p = p_s; p < p_e; ++p) {
        const S next_avg = (p + 1 < avg_n) ? avgp[(size_t)(p + 1) * as] : avg;
        S first, second;
        squeeze_pair<S>(resp[(size_t)p * rs], next_avg, avg, prev, first, second);
        // the only state a pair hands on is `prev` = its second sample: once the redone chain reaches the sample
        // the first pass stored there, everything behind it (and the segment's recorded end) already stands
        const bool rejoined = outp[(size_t)(2 * p + 1) * os] == second;
        outp[(size_t)(2 * p) * os] = first;
        outp[(size_t)(2 * p + 1) * os] = second;
        if (rejoined) return stored_end;
    }
    return prev;
}

// Walk the chain of segments of every line.  Segment s is exact iff the `prev` it had reached at its
// real start equals the `prev` the segment before it really ended with; a segment that is not is
// redone alone from that true state (not the whole line: a redone segment nearly always ends with the
// `prev` it ended with before, so the chain behind it stands), and the walk goes on with its new end.
template <typename S, bool HORIZONTAL>
__global__ __launch_bounds__(64) void squeeze_check3_kernel(SegArgs3 g3, int* redo_count) {
    const SegArgs& g = g3.g[blockIdx.y];
    const uint32_t lines = HORIZONTAL ? g.a.height : g.a.width;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= lines) return;
    const S* chk = (const S*)g.chk;
    const uint32_t pairs = (HORIZONTAL ? g.a.width : g.a.height) / 2;
    S end_prev = chk[((size_t)0 * 2 + 1) * lines + i];
    for (uint32_t s = 1; s < g.nseg; ++s) {
        if (chk[((size_t)s * 2 + 0) * lines + i] == end_prev) {
            end_prev = chk[((size_t)s * 2 + 1) * lines + i];
            continue;
        }
        if (redo_count) atomicAdd(redo_count, 1);
        const uint32_t p_s = s * g.seg_pairs, p_e = (s + 1 == g.nseg) ? pairs : p_s + g.seg_pairs;
        end_prev = squeeze_redo_range<S, HORIZONTAL>(g.a, i, p_s, p_e, end_prev, chk[((size_t)s * 2 + 1) * lines + i]);
    }
}

// The smallest levels of the pyramid (chains of a few dozen pairs, a few hundred lines) are pure
// launch latency as kernels of their own: one workgroup per channel walks all of them, one lane
// per line, a workgroup barrier between levels (the data goes through L2; a level is a few KB).
constexpr int kChainMaxSteps = 16;
struct ChainArgs {
    SqzArgs a[3][kChainMaxSteps];
    uint32_t horizontal[3][kChainMaxSteps];
    uint32_t n[3];
};

template <typename S>
__global__ __launch_bounds__(1024) void squeeze_chain_kernel(ChainArgs c) {
    const uint32_t ch = blockIdx.x;
    const uint32_t n = c.n[ch];
    for (uint32_t i = 0; i < n; ++i) {
        const SqzArgs a = c.a[ch][i];
        const bool horizontal = c.horizontal[ch][i] != 0;
        const uint32_t lines = horizontal ? a.height : a.width;
        for (uint32_t l = threadIdx.x; l < lines; l += 1024) {
            if (horizontal) squeeze_h_line<S>(a, l);
            else squeeze_v_line<S>(a, l);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------- device: RCT, palette, gradient
struct RctArgs {
    void* p[3];
    uint32_t stride[3];
    uint32_t width, height, rct_type;
};

// inverse_row_*_base + inverse_permute, rct.rs:154-256
template <typename S>
__global__ __launch_bounds__(256) void rct_kernel(RctArgs g) {
    uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.width) return;
    S* pa = (S*)g.p[0] + (size_t)y * g.stride[0] + x;
    S* pb = (S*)g.p[1] + (size_t)y * g.stride[1] + x;
    S* pc = (S*)g.p[2] + (size_t)y * g.stride[2] + x;
    const uint32_t permutation = g.rct_type / 7, type = g.rct_type % 7;
    S a = *pa, b = *pb, c = *pc, d, e, f;
    if (type == 6) {
        S tmp = Wrap<S>::sub(a, (S)(c >> 1));
        e = Wrap<S>::add(c, tmp);
        f = Wrap<S>::sub(tmp, (S)(b >> 1));
        d = Wrap<S>::add(f, b);
    } else {
        d = a;
        f = (type & 1) ? Wrap<S>::add(c, a) : c;
        e = (type >> 1) == 1 ? Wrap<S>::add(b, a)
          : (type >> 1) == 2 ? Wrap<S>::add(b, (S)(Wrap<S>::add(a, f) >> 1)) : b;
    }
    S o0 = d, o1 = e, o2 = f;  // rows after inverse_permute's swap sequence
    switch (permutation) {
        case 1: o0 = f; o1 = d; o2 = e; break;
        case 2: o0 = e; o1 = f; o2 = d; break;
        case 3: o0 = d; o1 = f; o2 = e; break;
        case 4: o0 = e; o1 = d; o2 = f; break;
        case 5: o0 = f; o1 = e; o2 = d; break;
        default: break;
    }
    *pa = o0; *pb = o1; *pc = o2;
}

struct PalArgs {
    const void* palette;   // nb_colours x num_c, stride pal_stride
    void* idx;             // leader (index) grid
    void* dst[8];          // dst[0] = leader itself
    uint32_t pal_stride, idx_stride, dst_stride[8];
    uint32_t width, height, num_c, nb_colours;
    int32_t nb_deltas;
    uint32_t bit_depth;
    uint8_t* need_delta;   // width x height: index < nb_deltas (palette.rs:63-65)
    int* n_delta;          // how many such samples
};

// transform/palette.rs:11-24 (data table)
__device__ constexpr int16_t kDeltaPalette[72][3] = {
    {0, 0, 0}, {4, 4, 4}, {11, 0, 0}, {0, 0, -13}, {0, -12, 0}, {-10, -10, -10},
    {-18, -18, -18}, {-27, -27, -27}, {-18, -18, 0}, {0, 0, -32}, {-32, 0, 0}, {-37, -37, -37},
    {0, -32, -32}, {24, 24, 45}, {50, 50, 50}, {-45, -24, -24}, {-24, -45, -45}, {0, -24, -24},
    {-34, -34, 0}, {-24, 0, -24}, {-45, -45, -24}, {64, 64, 64}, {-32, 0, -32}, {0, -32, 0},
    {-32, 0, 32}, {-24, -45, -24}, {45, 24, 45}, {24, -24, -45}, {-45, -24, 24}, {80, 80, 80},
    {64, 0, 0}, {0, 0, -64}, {0, -64, -64}, {-24, -24, 45}, {96, 96, 96}, {64, 64, 0},
    {45, -24, -24}, {34, -34, 0}, {112, 112, 112}, {24, -45, -45}, {45, 45, -24}, {0, -32, 32},
    {24, -24, 45}, {0, 96, 96}, {45, -24, 24}, {24, -45, -24}, {-24, -45, 24}, {0, -64, 0},
    {96, 0, 0}, {128, 128, 128}, {64, 0, 64}, {144, 144, 144}, {96, 96, 0}, {-36, -36, 36},
    {45, -24, -45}, {45, -45, -24}, {0, 0, -96}, {0, 128, 128}, {0, 96, 0}, {45, 24, -45},
    {-128, 0, 0}, {24, -45, 24}, {-45, 24, -45}, {64, 0, -64}, {64, -64, -64}, {96, 0, 96},
    {45, -45, 24}, {24, 45, -45}, {64, 64, -64}, {128, 128, 0}, {0, 0, -128}, {-24, 45, -45},
};

// Palette::inverse_inner's per-sample part (palette.rs:58-108): palette rows, the implicit 4x4x4 and
// 5x5x5 colour cubes above nb_colours, the fixed delta palette below zero.  With only in-range
// indices this is exactly inverse_simple (:146-173).
template <typename S>
__global__ __launch_bounds__(256) void palette_kernel(PalArgs g) {
    uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.width) return;
    const int32_t index = ((const S*)g.idx)[(size_t)y * g.idx_stride + x];
    const int32_t nb_colors = (int32_t)g.nb_colours;
    const bool delta = index < g.nb_deltas;
    g.need_delta[(size_t)y * g.width + x] = delta ? 1 : 0;
    if (delta) atomicAdd(g.n_delta, 1);
    const int32_t maxv = (1 << g.bit_depth) - 1;
    for (uint32_t c = g.num_c; c-- > 0;) {
        int32_t v;
        if (index >= 0 && index < nb_colors) {
            v = ((const S*)g.palette)[(size_t)c * g.pal_stride + index];
        } else if (index >= nb_colors) {
            int32_t i2 = index - nb_colors;
            if (i2 < 64) {
                v = ((i2 >> (2 * c)) % 4) * maxv / 4 + (1 << (g.bit_depth > 3 ? g.bit_depth - 3 : 0));
            } else {
                int32_t i3 = i2 - 64;
                for (uint32_t k = 0; k < c; ++k) i3 /= 5;
                v = (i3 % 5) * maxv / 4;
            }
        } else if (c >= 3) {
            v = 0;
        } else {
            int32_t i2 = (-(index + 1)) % 143;
            int32_t t = kDeltaPalette[(i2 + 1) >> 1][c];
            if ((i2 & 1) == 0) t = -t;
            if (g.bit_depth > 8) t <<= (g.bit_depth < 24 ? g.bit_depth : 24) - 8;
            v = t;
        }
        ((S*)g.dst[c])[(size_t)y * g.dst_stride[c] + x] = (S)v;
    }
}

// ---------------------------------------------------------------- device: M4, any single-leaf predictor
// decode_single_node[_slow] / decode_one / decode_simple_grad (jxl-modular/src/image.rs:716-949) for ONE
// (group, channel) subgrid per workgroup: sample = residual * multiplier + offset + predict(neighbours),
// Wrapping<S>.  All subgrids of all transformed channels of a frame form ONE launch (a tile list built
// by the host, largest first): a subgrid is a serial chain, so what fills the GPU is the number of
// subgrids in flight, not the size of one.
//
// Lane r owns row r and trails row r-1 by three columns (NEE and the self-correcting predictor's NE
// error reach two columns ahead in the previous row); one workgroup barrier per step.  Nothing on the
// step-to-step critical path touches global memory:
//   * every lane publishes its finished samples in an LDS ring (16 columns per row); the lanes below
//     read N / NE / NEE / NN from the rings of rows r-1 and r-2 (the producer is 3 / 6 columns ahead);
//   * residuals are requested 16 columns ahead — eight loads in flight per lane, rotating through
//     eight registers of the 8x unrolled step loop — and parked in a second LDS ring;
//   * results leave with fire-and-forget stores.
// (The first form of this kernel read the previous rows back from global memory after every barrier:
//  1.9 us per step, 37 ms for the 67 sub-channels of an 8K Squeeze frame, one launch per channel.)
// The neighbour registers (w, n, nw) and the self-correcting predictor's error registers follow
// PredictorState / Properties::record (predictor.rs:540-577) and SelfCorrectingPredictor
// (predictor.rs:312-441) statement by statement; its two error rows live in LDS and are
// overwritten in place exactly like the reference's `true_err_row` / `subpred_err_row`.
constexpr uint32_t kPredMaxTileW = 1024;  // the self-correcting predictor's error rows live in LDS
constexpr uint32_t kPredLaneMaxW = 1024;  // widest subgrid of the lane-packed kernel: row r - 2 is 2 D ring columns ahead, D <= 8 with the
                                          // 16-column sample ring (subgrids up to 512 columns), D = 16 with the 64-column one (group_dim 1024)
constexpr int kRing = 16;                 // columns per row in the LDS rings (power of two, > 6 + look-ahead)
constexpr int kBigRing = 64;              // sample ring of the lane kernels when a subgrid is wider than 512 columns (2 D = 32 columns between rows r and r - 2)
struct PredTile {
    void* base;              // first sample of the subgrid: residuals in, samples out (in place)
    uint32_t stride, gw, gh; // elements
    uint32_t packed;         // 1: handled by the lane-packed kernel (P lanes of a wave), 0: a workgroup of its own
    // the unit's MA-tree leaf (MaTreeLeafClustered without its cluster): the frame's one leaf, or the unit's own
    // (JxlGpuModularDesc::unit_leaves: trees that split on the static properties channel / stream index, make_flat_tree)
    // ... or, predictor == JXLGPU_LEAF_BY_ROW / _BY_COLUMN: a tree that still splits on y / x inside the unit (decode_slow,
    // image.rs:1169-1228, with get_leaf a function of the row / column alone): the leaf of a sample is PredArgs::axis[mul + row]
    // resp. [mul + column]; `sc` = one of those leaves is the self-correcting predictor, whose state is then kept for every
    // sample of the unit (FlatMaTree::need_self_correcting, ma.rs:275-285).  Kernels compiled with MAPPED serve these.
    uint32_t predictor;
    int32_t mul, off;
    uint32_t sc;
};
struct PredArgs {
    const PredTile* tiles;
    const JxlGpuMaLeaf* axis;  // JxlGpuModularDesc::axis_leaves on the device (MAPPED kernels)
    void* sink;              // 64 samples nobody reads: where off-grid lanes of the one-wave kernels store
    uint32_t err_w;          // columns of the error rows in dynamic LDS (>= the widest subgrid; 1 when unused)
    int32_t wp[11];
};

__device__ __forceinline__ uint32_t div_lookup_dev(uint32_t i) { return i == 0 ? 0u : (1u << 24) / i; }  // predictor.rs:150-160

template <typename S, bool MAPPED = false>
__global__ __launch_bounds__(256) void predict_tiles_kernel(PredArgs a) {
    // the self-correcting predictor's error rows: 5 x err_w words of dynamic LDS (err_w = widest subgrid of the launch)
    extern __shared__ int32_t s_err[];
    int32_t* s_true_err = s_err;
    uint32_t* s_sub_err[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s_sub_err[k] = reinterpret_cast<uint32_t*>(s_err) + (size_t)(k + 1) * a.err_w;
    __shared__ uint32_t s_div[65];              // DIV_LOOKUP, predictor.rs:150-160 (a table as in the reference: the
                                                // five divisions per sample were a quarter of the step's instructions)
    if (threadIdx.x < 65) s_div[threadIdx.x] = div_lookup_dev(threadIdx.x);
    __shared__ int32_t s_out[256][kRing + 1];   // finished samples, row r, column x & 15 (+1: bank spread)
    __shared__ int32_t s_in[256][kRing + 1];    // residuals requested ahead
    const PredTile t = a.tiles[blockIdx.x];
    const uint32_t gw = t.gw, gh = t.gh;
    const uint32_t r = threadIdx.x;
    S* row = (S*)t.base + (size_t)r * t.stride;
    const bool have_row = r < gh;
    const uint32_t tile_pred = t.predictor;   // one subgrid per workgroup: uniform
    const bool sc_on = tile_pred == 6 || (MAPPED && tile_pred >= JXLGPU_LEAF_BY_ROW && t.sc);
    for (uint32_t i = r; i < 5 * a.err_w; i += 256) s_err[i] = 0;
    __syncthreads();
    const int32_t* prev = s_out[r > 0 ? r - 1 : 0];
    const int32_t* prev2 = s_out[r > 1 ? r - 2 : 0];

    // PredictorState registers
    int32_t w = 0, n = 0, nw = 0, ww1 = 0 /* sample at x-1 */, ww2 = 0 /* sample at x-2 */;
    // SelfCorrectingPredictor registers
    int32_t te_w = 0, te_nw = 0, te_n = 0, te_ne = 0;
    uint32_t se_nw_ww[4] = {0, 0, 0, 0}, se_n_w[4] = {0, 0, 0, 0}, se_ne[4] = {0, 0, 0, 0};

    int32_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // residuals in flight: pf[j] was requested 8 steps ago for column x + 8
    const int32_t steps = (int32_t)(gw + 3 * (gh - 1));
    for (int32_t s0 = -16; s0 < steps; s0 += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int32_t s = s0 + j;
            const int32_t x = s - 3 * (int32_t)r;
            // rows whose lanes are all idle (not within 16 columns of starting, or finished) skip the step
            const bool busy = have_row && x >= -16 && x < (int32_t)gw;
            if (__builtin_amdgcn_ballot_w64(busy) != 0) {
                // ---- residual pipeline: park what arrived, request 16 columns ahead
                if (busy) {
                    if (x + 8 >= 0 && x + 8 < (int32_t)gw) s_in[r][(x + 8) & (kRing - 1)] = pf[j];
                    if (x + 16 >= 0 && x + 16 < (int32_t)gw) pf[j] = (int32_t)row[x + 16];
                }
                if (busy && x >= 0) {
                    if (x == 0) {
                        // state at a row start: reset() for row 0, the row-end branch of record() otherwise
                        if (r == 0) {
                            w = n = nw = 0;
                        } else {
                            w = n = nw = prev[0];
                            if (sc_on) {
                                te_w = 0;
                                te_n = s_true_err[0];
                                te_nw = te_n;
#pragma unroll
                                for (int i = 0; i < 4; ++i) { se_n_w[i] = s_sub_err[i][0]; se_nw_ww[i] = se_n_w[i]; }
                                if (gw <= 1) {
                                    te_ne = te_n;
#pragma unroll
                                    for (int i = 0; i < 4; ++i) se_ne[i] = se_n_w[i];
                                } else {
                                    te_ne = s_true_err[1];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i][1];
                                }
                            }
                        }
                    }
                    // neighbours beyond w / n / nw (predictor.rs:226-273, EDGE = true everywhere)
                    const bool no_prev = r == 0;
                    const int32_t ne = (no_prev || x + 1 >= (int32_t)gw) ? n : prev[(x + 1) & (kRing - 1)];
                    const int32_t nee = (no_prev || x + 2 >= (int32_t)gw) ? ne : prev[(x + 2) & (kRing - 1)];
                    const int32_t nn = r >= 2 ? prev2[x & (kRing - 1)] : n;
                    const int32_t ww = x >= 2 ? ww2 : w;

                    int64_t sc_prediction = 0, subpred[4] = {0, 0, 0, 0};
                    if (sc_on) {
                        const int64_t tw = te_w, tnw = te_nw, tn = te_n, tne = te_ne;
                        const int64_t n3 = (int64_t)n * 8, nw3 = (int64_t)nw * 8, ne3 = (int64_t)ne * 8, w3 = (int64_t)w * 8,
                                      nn3 = (int64_t)nn * 8;
                        subpred[0] = w3 + ne3 - n3;
                        subpred[1] = n3 - (((tw + tn + tne) * (int64_t)a.wp[0]) >> 5);
                        subpred[2] = w3 - (((tw + tn + tnw) * (int64_t)a.wp[1]) >> 5);
                        subpred[3] = n3 - ((tnw * (int64_t)a.wp[2] + tn * (int64_t)a.wp[3] + tne * (int64_t)a.wp[4] +
                                            (nn3 - n3) * (int64_t)a.wp[5] + (nw3 - w3) * (int64_t)a.wp[6]) >> 5);
                        uint32_t weight[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t err_sum = se_nw_ww[i] + se_n_w[i] + se_ne[i];
                            const uint64_t tt = ((uint64_t)err_sum + 1) >> 5;
                            const uint32_t shift = tt ? 63u - (uint32_t)__builtin_clzll(tt) : 0u;
                            weight[i] = 4 + (((uint32_t)a.wp[7 + i] * s_div[(err_sum >> shift) + 1]) >> shift);
                        }
                        uint32_t sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                        const uint32_t log_weight = 31u - (uint32_t)__builtin_clz(sum_weights >> 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) weight[i] >>= log_weight;
                        sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                        int64_t acc = ((int64_t)sum_weights >> 1) - 1;
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc += subpred[i] * (int64_t)weight[i];
                        int64_t prediction = (acc * (int64_t)s_div[sum_weights]) >> 24;
                        if (((tn ^ tw) | (tn ^ tnw)) <= 0) {
                            const int64_t mn = min(min(n3, w3), ne3), mx = max(max(n3, w3), ne3);
                            prediction = prediction < mn ? mn : (prediction > mx ? mx : prediction);
                        }
                        sc_prediction = prediction;
                    }

                    // Predictor::predict, predictor.rs:79-125
                    uint32_t predictor = tile_pred;
                    int32_t lmul = t.mul, loff = t.off;
                    if constexpr (MAPPED) {
                        if (tile_pred >= JXLGPU_LEAF_BY_ROW) {   // the sample's own leaf: by row (property 2) or by column (property 3)
                            const JxlGpuMaLeaf lf = a.axis[(uint32_t)t.mul + (tile_pred == JXLGPU_LEAF_BY_ROW ? r : (uint32_t)x)];
                            predictor = lf.predictor; lmul = lf.multiplier; loff = lf.offset;
                        }
                    }
                    int32_t pred;
                    {
                        const int64_t N = n, W = w, NW = nw;
                        switch (predictor) {
                            case 0: pred = 0; break;
                            case 1: pred = w; break;
                            case 2: pred = n; break;
                            case 3: pred = (int32_t)((W + N) / 2); break;
                            case 4: {
                                const int64_t dn = N > NW ? N - NW : NW - N, dw = W > NW ? W - NW : NW - W;
                                pred = dn < dw ? w : n;
                                break;
                            }
                            case 5: {
                                const int64_t g = N + W - NW, lo = W < N ? W : N, hi = W > N ? W : N;
                                pred = (int32_t)(g < lo ? lo : (g > hi ? hi : g));
                                break;
                            }
                            case 6: pred = (int32_t)((sc_prediction + 3) >> 3); break;
                            case 7: pred = ne; break;
                            case 8: pred = nw; break;
                            case 9: pred = ww; break;
                            case 10: pred = (int32_t)((W + NW) / 2); break;
                            case 11: pred = (int32_t)((N + NW) / 2); break;
                            case 12: pred = (int32_t)((N + (int64_t)ne) / 2); break;
                            default:
                                pred = (int32_t)((6 * N - 2 * (int64_t)nn + 7 * W + (int64_t)ww + (int64_t)nee + 3 * (int64_t)ne + 8) / 16);
                                break;
                        }
                    }
                    // decode_one: diff = residual.wrapping_muladd_i32(multiplier, offset); diff.add(prediction)
                    const S res = (S)s_in[r][x & (kRing - 1)];
                    const S diff = Wrap<S>::add(Wrap<S>::mul(res, (S)lmul), (S)loff);
                    const S value = Wrap<S>::add(diff, (S)pred);
                    row[x] = value;
                    const int32_t sample = (int32_t)value;
                    s_out[r][x & (kRing - 1)] = sample;

                    if (sc_on) {
                        // SelfCorrectingPredictor::record, predictor.rs:394-441 (the row-end branch is the
                        // x == 0 block above, run by the next row's lane)
                        const int64_t s8 = (int64_t)sample * 8;
                        const int64_t true_err = sc_prediction - s8;
                        uint32_t sub_err[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int64_t d = subpred[i] - s8;
                            sub_err[i] = (uint32_t)(((uint64_t)(d < 0 ? -d : d) + 3) >> 3);
                        }
                        s_true_err[x] = (int32_t)true_err;
#pragma unroll
                        for (int i = 0; i < 4; ++i) s_sub_err[i][x] = sub_err[i];
                        if (x + 1 < (int32_t)gw) {
                            te_w = (int32_t)true_err;
                            te_nw = te_n;
                            te_n = te_ne;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                se_nw_ww[i] = se_n_w[i];
                                se_n_w[i] = se_ne[i] + sub_err[i];
                            }
                            if (x + 2 >= (int32_t)gw) {
                                te_ne = te_n;
#pragma unroll
                                for (int i = 0; i < 4; ++i) se_ne[i] = se_n_w[i];
                            } else if (r != 0) {
                                te_ne = s_true_err[x + 2];
#pragma unroll
                                for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i][x + 2];
                            }
                        }
                    }
                    // Properties::record, predictor.rs:552-576 (not at a row end)
                    if (x + 1 < (int32_t)gw) {
                        ww2 = ww1;
                        ww1 = sample;
                        w = sample;
                        if (r == 0) {
                            nw = sample;
                            n = sample;
                        } else {
                            nw = n;
                            n = prev[(x + 1) & (kRing - 1)];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- lane-packed form (the default for subgrids up to 512 columns).  In the kernel above a subgrid owns a
// whole 256-thread workgroup and only gw / 3 of its rows are inside the subgrid at any step: 11 lanes for the
// 32 x 64 tiles of a deep Squeeze level, 43 for a 128-column one — and every wave that holds one of them issues
// the full step (measured on the 67 sub-channels of an 8K Squeeze frame: 1.4 us per step, the VALU time of the
// 2.3 waves that are partly active, 4.8 ms per frame).  Here a subgrid gets P = pow2ceil(gw) / 4 lanes (1..64)
// of ONE wave and row r + 1 trails row r by D = pow2ceil(gw) / P >= 4 columns: a lane that finishes row r has
// exactly reached the start step of row r + P, so it walks rows k, k + P, k + 2P, ... back to back (its stream
// position u = step - D k splits into round u >> log2(DP) and column u & (DP - 1)), every lane is busy from its
// first row to its last, a wave carries 64 / P subgrids, and there is no workgroup barrier: the rings belong to
// the wave, whose LDS operations execute in order.  Same registers, same statements per sample as above
// (predictor.rs:26-442, image.rs:716-949); only the schedule differs, and any number of rows is taken.
// Step boundary of the one-wave kernels: the rings are private to the wave and LDS operations of one wave execute in
// program order, so all that is needed is that the COMPILER keeps LDS accesses on their side of the boundary.  (A
// wavefront-scope release / acquire fence pair does that too, but it is lowered to s_waitcnt vmcnt(0) lgkmcnt(0):
// every step then waited for the residual it had just requested 16 columns ahead and for its own store — one HBM
// round trip per step, 1.9 us of the 1.9 us a step took.)
// A pointer read from a table in memory is a generic (flat) pointer to the compiler; flat loads count in lgkmcnt as
// well as vmcnt, so every wait for an LDS read would also wait for the residual requested 16 columns ahead.
template <typename T>
using GlobalPtr = T __attribute__((address_space(1)))*;
template <typename T>
__device__ __forceinline__ GlobalPtr<T> as_global(T* p) { return (GlobalPtr<T>)p; }

__device__ __forceinline__ void lds_step_boundary() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

struct PredWave {
    uint32_t first, count;    // tiles[first .. first + count): the subgrids of this wave, P lanes each
    uint32_t log2p, log2dp;   // P lanes per subgrid; DP = D * P columns per round (>= every gw of the wave)
    uint32_t steps;           // max over the wave's subgrids of gw + D (gh - 1)
    uint32_t vec, pad[2];     // every subgrid of the wave takes four-sample accesses (alignment, gw % 4 == 0)
};

struct PredSrc {
    const void* src;   // residuals of the subgrid (same geometry as PredTile.base; == base when the pass runs in place)
};

// `srcs` / `wave_flags` non-null: the redo pass behind predict_lanes_narrow_kernel — only flagged waves run, and
// they read the residuals from `srcs` (the narrow pass has written over `base`).
// VEC: four-sample global accesses, as in predict_lanes_narrow_kernel below.
template <typename S, bool VEC, int RO, bool MAPPED = false>
__global__ __launch_bounds__(64) void predict_lanes_kernel(PredArgs a, const PredWave* waves, const PredSrc* srcs,
                                                           const uint32_t* wave_flags) {
    if (wave_flags && wave_flags[blockIdx.x] == 0) return;
    extern __shared__ int32_t s_err[];          // 5 x err_w words: true_err, sub_err[4]; a subgrid's columns start at slot * (err_w * P / 64)
    __shared__ uint32_t s_div[65];
    __shared__ int32_t s_out[64][RO + 1];    // finished samples of the lane's current / previous rows, stream position & 15
    __shared__ int32_t s_in[64][kRing + 1];     // residuals requested ahead, stream position & 15
    const PredWave wv = waves[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const uint32_t log2p = wv.log2p, log2dp = wv.log2dp, P = 1u << log2p, DPm1 = (1u << log2dp) - 1u;
    const int32_t D = (int32_t)(1u << (log2dp - log2p));
    const uint32_t slot = lane >> log2p, k = lane & (P - 1);
    const bool have_tile = slot < wv.count;
    PredTile t = a.tiles[wv.first + (have_tile ? slot : 0)];
    const GlobalPtr<const S> src = as_global(srcs ? (const S*)srcs[wv.first + (have_tile ? slot : 0)].src : (const S*)t.base);
    const uint32_t gw = have_tile ? t.gw : 0, gh = have_tile ? t.gh : 0;
    const uint32_t ecol = slot * ((a.err_w << log2p) >> 6);   // this subgrid's first column in the error rows
    int32_t* s_true_err = s_err + ecol;
    uint32_t* s_sub_err[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_sub_err[i] = reinterpret_cast<uint32_t*>(s_err) + (size_t)(i + 1) * a.err_w + ecol;
    for (uint32_t i = lane; i < 5 * a.err_w; i += 64) s_err[i] = 0;
    s_div[lane] = div_lookup_dev(lane);
    if (lane == 0) s_div[64] = div_lookup_dev(64);
    __syncthreads();
    // the subgrids of a wave share their predictor (the host forms waves by it); multiplier and offset are the subgrid's own
    // (MAPPED: a wave's subgrids share the marker BY_ROW / BY_COLUMN; the leaf is the sample's own, fetched per step — a
    //  load inside the step loop, which is why this is a separate instantiation: see the note on counted accesses below)
    const uint32_t wave_pred = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tiles[wv.first].predictor);
    const bool sc_on = wave_pred == 6 || (MAPPED && wave_pred >= JXLGPU_LEAF_BY_ROW && t.sc);
    // rows r - 1 and r - 2 live in the rings of lanes k - 1 and k - 2 (mod P), one round back where the index wrapped
    const uint32_t lane0 = lane & ~(P - 1);
    uint32_t wrap1 = 0, wrap2 = 0;
    int32_t q1 = (int32_t)k - 1, q2 = (int32_t)k - 2;
    while (q1 < 0) { q1 += (int32_t)P; ++wrap1; }
    while (q2 < 0) { q2 += (int32_t)P; ++wrap2; }
    const int32_t* prev = s_out[lane0 + (uint32_t)q1];
    const int32_t* prev2 = s_out[lane0 + (uint32_t)q2];

    int32_t w = 0, n = 0, nw = 0, ww1 = 0, ww2 = 0;
    int32_t te_w = 0, te_nw = 0, te_n = 0, te_ne = 0;
    uint32_t se_nw_ww[4] = {0, 0, 0, 0}, se_n_w[4] = {0, 0, 0, 0}, se_ne[4] = {0, 0, 0, 0};
    int32_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef uint32_t RawV2 __attribute__((ext_vector_type(2)));
    typedef uint32_t RawV4 __attribute__((ext_vector_type(4)));
    using V4 = typename std::conditional<sizeof(S) == 2, RawV2, RawV4>::type;
    union Pack4 { V4 v; S s[4]; };
    Pack4 pfv[2], sbuf;
    pfv[0].v = pfv[1].v = V4{};
    sbuf.v = V4{};
    const int32_t steps = (int32_t)wv.steps;
    const int32_t u0 = -D * (int32_t)k;
    // the element at stream position q: round q >> log2dp, column q & (DP - 1); inside the subgrid?
    auto where = [&](int32_t q, uint32_t* r, uint32_t* x) -> bool {
        *r = k + (((uint32_t)q >> log2dp) << log2p);
        *x = (uint32_t)q & DPm1;
        return q >= 0 && *r < gh && *x < gw;
    };
    for (int32_t s0 = -16; s0 < steps; s0 += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int32_t u = u0 + s0 + j;
            if constexpr (VEC) {
                if ((j & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) s_in[lane][(u + 8 + i) & (kRing - 1)] = (int32_t)pfv[j >> 2].s[i];
                    uint32_t rq, xq;
                    const bool ahead = where(u + 16, &rq, &xq);
                    pfv[j >> 2].v = *reinterpret_cast<GlobalPtr<const V4>>(src + (ahead ? (size_t)rq * t.stride + xq : (size_t)0));
                }
            } else {   // residual pipeline: park what arrived for position u + 8, request position u + 16.  Every global access
                // of the step is issued unconditionally (an off-grid lane reads the subgrid's first sample, parks it in a
                // slot nobody reads and stores to `sink`): with a load or store under a branch the compiler cannot count
                // the accesses in flight and waits for ALL of them (s_waitcnt vmcnt(0)) before it parks a residual
                uint32_t rq, xq;
                s_in[lane][(u + 8) & (kRing - 1)] = pf[j];
                const bool ahead = where(u + 16, &rq, &xq);
                pf[j] = (int32_t)src[ahead ? (size_t)rq * t.stride + xq : (size_t)0];
            }
            uint32_t r, ux;
            S st_value = 0;
            GlobalPtr<S> st_ptr = as_global((S*)a.sink + lane * 4 + 3);
            if (where(u, &r, &ux)) {
                const int32_t x = (int32_t)ux;
                const uint32_t round = (uint32_t)u >> log2dp;
                // ring positions of column 0 of rows r, r - 1, r - 2 (stream position of their lane & 15)
                const uint32_t ob = (round << log2dp) & (RO - 1);
                const uint32_t pb1 = ((round - wrap1) << log2dp) & (RO - 1), pb2 = ((round - wrap2) << log2dp) & (RO - 1);
                if (x == 0) {
                    if (r == 0) {
                        w = n = nw = 0;
                    } else {
                        w = n = nw = prev[pb1];
                        if (sc_on) {
                            te_w = 0;
                            te_n = s_true_err[0];
                            te_nw = te_n;
#pragma unroll
                            for (int i = 0; i < 4; ++i) { se_n_w[i] = s_sub_err[i][0]; se_nw_ww[i] = se_n_w[i]; }
                            if (gw <= 1) {
                                te_ne = te_n;
#pragma unroll
                                for (int i = 0; i < 4; ++i) se_ne[i] = se_n_w[i];
                            } else {
                                te_ne = s_true_err[1];
#pragma unroll
                                for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i][1];
                            }
                        }
                    }
                }
                const bool no_prev = r == 0;
                const int32_t ne = (no_prev || x + 1 >= (int32_t)gw) ? n : prev[(pb1 + x + 1) & (RO - 1)];
                const int32_t nee = (no_prev || x + 2 >= (int32_t)gw) ? ne : prev[(pb1 + x + 2) & (RO - 1)];
                const int32_t nn = r >= 2 ? prev2[(pb2 + x) & (RO - 1)] : n;
                const int32_t ww = x >= 2 ? ww2 : w;

                int64_t sc_prediction = 0, subpred[4] = {0, 0, 0, 0};
                if (sc_on) {
                    const int64_t tw = te_w, tnw = te_nw, tn = te_n, tne = te_ne;
                    const int64_t n3 = (int64_t)n * 8, nw3 = (int64_t)nw * 8, ne3 = (int64_t)ne * 8, w3 = (int64_t)w * 8,
                                  nn3 = (int64_t)nn * 8;
                    subpred[0] = w3 + ne3 - n3;
                    subpred[1] = n3 - (((tw + tn + tne) * (int64_t)a.wp[0]) >> 5);
                    subpred[2] = w3 - (((tw + tn + tnw) * (int64_t)a.wp[1]) >> 5);
                    subpred[3] = n3 - ((tnw * (int64_t)a.wp[2] + tn * (int64_t)a.wp[3] + tne * (int64_t)a.wp[4] +
                                        (nn3 - n3) * (int64_t)a.wp[5] + (nw3 - w3) * (int64_t)a.wp[6]) >> 5);
                    uint32_t weight[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t err_sum = se_nw_ww[i] + se_n_w[i] + se_ne[i];
                        const uint64_t tt = ((uint64_t)err_sum + 1) >> 5;
                        const uint32_t shift = tt ? 63u - (uint32_t)__builtin_clzll(tt) : 0u;
                        weight[i] = 4 + (((uint32_t)a.wp[7 + i] * s_div[(err_sum >> shift) + 1]) >> shift);
                    }
                    uint32_t sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                    const uint32_t log_weight = 31u - (uint32_t)__builtin_clz(sum_weights >> 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) weight[i] >>= log_weight;
                    sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                    int64_t acc = ((int64_t)sum_weights >> 1) - 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc += subpred[i] * (int64_t)weight[i];
                    int64_t prediction = (acc * (int64_t)s_div[sum_weights]) >> 24;
                    if (((tn ^ tw) | (tn ^ tnw)) <= 0) {
                        const int64_t mn = min(min(n3, w3), ne3), mx = max(max(n3, w3), ne3);
                        prediction = prediction < mn ? mn : (prediction > mx ? mx : prediction);
                    }
                    sc_prediction = prediction;
                }

                uint32_t predictor = wave_pred;
                int32_t lmul = t.mul, loff = t.off;
                if constexpr (MAPPED) {
                    if (wave_pred >= JXLGPU_LEAF_BY_ROW) {
                        const JxlGpuMaLeaf lf = a.axis[(uint32_t)t.mul + (wave_pred == JXLGPU_LEAF_BY_ROW ? r : ux)];
                        predictor = lf.predictor; lmul = lf.multiplier; loff = lf.offset;
                    }
                }
                int32_t pred;
                {
                    const int64_t N = n, W = w, NW = nw;
                    switch (predictor) {
                        case 0: pred = 0; break;
                        case 1: pred = w; break;
                        case 2: pred = n; break;
                        case 3: pred = (int32_t)((W + N) / 2); break;
                        case 4: {
                            const int64_t dn = N > NW ? N - NW : NW - N, dw = W > NW ? W - NW : NW - W;
                            pred = dn < dw ? w : n;
                            break;
                        }
                        case 5: {
                            const int64_t g = N + W - NW, lo = W < N ? W : N, hi = W > N ? W : N;
                            pred = (int32_t)(g < lo ? lo : (g > hi ? hi : g));
                            break;
                        }
                        case 6: pred = (int32_t)((sc_prediction + 3) >> 3); break;
                        case 7: pred = ne; break;
                        case 8: pred = nw; break;
                        case 9: pred = ww; break;
                        case 10: pred = (int32_t)((W + NW) / 2); break;
                        case 11: pred = (int32_t)((N + NW) / 2); break;
                        case 12: pred = (int32_t)((N + (int64_t)ne) / 2); break;
                        default:
                            pred = (int32_t)((6 * N - 2 * (int64_t)nn + 7 * W + (int64_t)ww + (int64_t)nee + 3 * (int64_t)ne + 8) / 16);
                            break;
                    }
                }
                const S res = (S)s_in[lane][u & (kRing - 1)];
                const S diff = Wrap<S>::add(Wrap<S>::mul(res, (S)lmul), (S)loff);
                const S value = Wrap<S>::add(diff, (S)pred);
                st_value = value;
                st_ptr = as_global((S*)t.base + (size_t)r * t.stride + ux);
                const int32_t sample = (int32_t)value;
                s_out[lane][(ob + ux) & (RO - 1)] = sample;

                if (sc_on) {
                    const int64_t s8 = (int64_t)sample * 8;
                    const int64_t true_err = sc_prediction - s8;
                    uint32_t sub_err[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t d = subpred[i] - s8;
                        sub_err[i] = (uint32_t)(((uint64_t)(d < 0 ? -d : d) + 3) >> 3);
                    }
                    s_true_err[x] = (int32_t)true_err;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s_sub_err[i][x] = sub_err[i];
                    if (x + 1 < (int32_t)gw) {
                        te_w = (int32_t)true_err;
                        te_nw = te_n;
                        te_n = te_ne;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            se_nw_ww[i] = se_n_w[i];
                            se_n_w[i] = se_ne[i] + sub_err[i];
                        }
                        if (x + 2 >= (int32_t)gw) {
                            te_ne = te_n;
#pragma unroll
                            for (int i = 0; i < 4; ++i) se_ne[i] = se_n_w[i];
                        } else if (r != 0) {
                            te_ne = s_true_err[x + 2];
#pragma unroll
                            for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i][x + 2];
                        }
                    }
                }
                if (x + 1 < (int32_t)gw) {
                    ww2 = ww1;
                    ww1 = sample;
                    w = sample;
                    if (r == 0) {
                        nw = sample;
                        n = sample;
                    } else {
                        nw = n;
                        n = prev[(pb1 + x + 1) & (RO - 1)];
                    }
                }
            }
            if constexpr (VEC) {
                sbuf.s[j & 3] = st_value;
                if ((j & 3) == 3) {
                    const GlobalPtr<S> g4 = st_ptr - 3;
                    *reinterpret_cast<GlobalPtr<V4>>(g4) = sbuf.v;
                }
            } else {
                *st_ptr = st_value;
            }
            lds_step_boundary();
        }
    }
}

// ---- the self-correcting predictor in 32-bit arithmetic (predictor 6 in the lane-packed form; since round 6 for launches with a wave of
// D > 4 only — subgrids wider than 256 columns, group_dim 512 / 1024 — and under JXLGPU_PRED_STEP_V1: predict_lanes_wp4_kernel below serves the rest).
// The reference computes in i64 (predictor.rs:312-441) and the kernels above follow it: 20 v_mad_u64_u32, 40
// carry pairs and a dozen 64-bit compares per sample, most of the ~320 instructions of a step.  With
// |sample| < 2^17, |true_err| < 2^19 and the WpHeader fields in their coded ranges (p1, p2, p3a-e < 32,
// w0-3 < 16; checked by the host) every intermediate fits 32 bits — |subpred| < 2^23, |sum| < 2^30; only
// acc * DIV_LOOKUP[sum_weights] needs the high half of one product — so int32 arithmetic gives the same
// numbers.  A lane that produces a sample or a true error outside those bounds raises its wave's flag and the
// wave stops (the offending value is itself still exact: its inputs were in range); predict_lanes_kernel then
// redoes exactly the flagged waves from the untouched residuals: the narrow kernel reads them from `src`
// (the read-only upload) and writes `base`.  Nothing an 8- to 16-bit image produces comes near the bounds.
// Besides the arithmetic: the step is straight-line code but for the row-start block (state that the next row
// start overwrites anyway is updated unconditionally; the error rows are zero where the first row reads them),
// and every LDS read that does not depend on this step's arithmetic is issued at its top.
// VEC: a lane moves its residuals and samples four at a time (8- or 16-byte accesses; the subgrids of the wave are
// aligned for it and their width is a multiple of four).  With one sample per lane per step every wave instruction
// touched 64 different cache lines for 128 useful bytes, twice per step: the address path of the CU, not the
// arithmetic, set the step time (same kernel without its global accesses: 0.8 instead of 2.0 ms per 8K frame).
// A lane's stream position is congruent to the step index modulo 4 (D is a multiple of 4), so the group boundaries
// are compile-time positions of the unrolled loop: request + park at steps 0 and 4, store at steps 3 and 7.
template <typename S, bool VEC, int RO>
__global__ __launch_bounds__(64) void predict_lanes_narrow_kernel(PredArgs a, const PredWave* waves, const PredSrc* srcs,
                                                                  uint32_t* wave_flags) {
    extern __shared__ int32_t s_err[];
    __shared__ uint32_t s_div[65];
    // the sample / residual rings hold values of the sample type: 16-bit rings for 16-bit buffers (BASELINE config 3) — the LDS
    // footprint is what bounds the waves per CU of this kernel (14 -> 9.6 KB per wave with the 5 x 256-word error rows: 11 -> 16
    // waves per CU), and a wave is one serial chain per subgrid
    using R = typename std::conditional<sizeof(S) == 2, int16_t, int32_t>::type;
    __shared__ R s_out[64][RO + 1];
    __shared__ R s_in[64][kRing + 1];
    const PredWave wv = waves[blockIdx.x];
    // A wave is one serial chain; the launch lasts as long as its longest wave (1276 steps for a 256 x 256 subgrid), and a step's
    // latency stretches with every other wave that shares the SIMD's issue slots.  The host ranks the waves by length
    // (PredWave::pad[1]): the long ones issue ahead of the short ones, which have slack — JXLGPU_PRED_PRIO=1, an experiment of round 6 that
    // LOST 7 % on config 3 (Tuning::pred_prio): off by default, pad[1] is 0.
    switch (__builtin_amdgcn_readfirstlane((int)(wv.pad[1] & 3u))) {
        case 3: __builtin_amdgcn_s_setprio(3); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        default: break;
    }
    const uint32_t lane = threadIdx.x;
    const uint32_t log2p = wv.log2p, log2dp = wv.log2dp, P = 1u << log2p, DPm1 = (1u << log2dp) - 1u;
    const int32_t D = (int32_t)(1u << (log2dp - log2p));
    const uint32_t slot = lane >> log2p, k = lane & (P - 1);
    const bool have_tile = slot < wv.count;
    const PredTile t = a.tiles[wv.first + (have_tile ? slot : 0)];
    const GlobalPtr<const S> src = as_global((const S*)srcs[wv.first + (have_tile ? slot : 0)].src);
    const uint32_t gw = have_tile ? t.gw : 0, gh = have_tile ? t.gh : 0;
    const uint32_t ecol = slot * ((a.err_w << log2p) >> 6);
    int32_t* s_true_err = s_err + ecol;
    uint32_t* s_sub_err[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_sub_err[i] = reinterpret_cast<uint32_t*>(s_err) + (size_t)(i + 1) * a.err_w + ecol;
    for (uint32_t i = lane; i < 5 * a.err_w; i += 64) s_err[i] = 0;
    s_div[lane] = div_lookup_dev(lane);
    if (lane == 0) { s_div[64] = div_lookup_dev(64); wave_flags[blockIdx.x] = 0; }
    __syncthreads();
    const uint32_t lane0 = lane & ~(P - 1);
    uint32_t wrap1 = 0, wrap2 = 0;
    int32_t q1 = (int32_t)k - 1, q2 = (int32_t)k - 2;
    while (q1 < 0) { q1 += (int32_t)P; ++wrap1; }
    while (q2 < 0) { q2 += (int32_t)P; ++wrap2; }
    const R* prev = s_out[lane0 + (uint32_t)q1];
    const R* prev2 = s_out[lane0 + (uint32_t)q2];
    const int32_t wp0 = a.wp[0], wp1 = a.wp[1], wp2 = a.wp[2], wp3 = a.wp[3], wp4 = a.wp[4], wp5 = a.wp[5], wp6 = a.wp[6];
    const uint32_t ww[4] = {(uint32_t)a.wp[7], (uint32_t)a.wp[8], (uint32_t)a.wp[9], (uint32_t)a.wp[10]};

    int32_t w = 0, n = 0, nw = 0;
    int32_t te_w = 0, te_nw = 0, te_n = 0, te_ne = 0;
    uint32_t se_nw_ww[4] = {0, 0, 0, 0}, se_n_w[4] = {0, 0, 0, 0}, se_ne[4] = {0, 0, 0, 0};
    uint32_t pb1 = 0, pb2 = 0;   // ring positions of column 0 of rows r - 1, r - 2 (set at the row start)
    int32_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef uint32_t RawV2 __attribute__((ext_vector_type(2)));
    typedef uint32_t RawV4 __attribute__((ext_vector_type(4)));
    using V4 = typename std::conditional<sizeof(S) == 2, RawV2, RawV4>::type;   // four samples (a plain vector type: usable through address-space pointers)
    union Pack4 { V4 v; S s[4]; };
    Pack4 pfv[2], sbuf;
    pfv[0].v = pfv[1].v = V4{};
    sbuf.v = V4{};
    const int32_t steps = (int32_t)wv.steps;
    const int32_t u0 = -D * (int32_t)k;
    auto where = [&](int32_t q, uint32_t* r, uint32_t* x) -> bool {
        *r = k + (((uint32_t)q >> log2dp) << log2p);
        *x = (uint32_t)q & DPm1;
        return q >= 0 && *r < gh && *x < gw;
    };
    for (int32_t s0 = -16; s0 < steps; s0 += 8) {
        bool out_of_range = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int32_t u = u0 + s0 + j;
            if constexpr (VEC) {
                if ((j & 3) == 0) {
                    // park the four residuals requested 8 steps ago (positions u + 8 .. u + 11: one row, contiguous ring
                    // slots), request positions u + 16 .. u + 19; always issued (see the one-sample form below)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s_in[lane][(u + 8 + i) & (kRing - 1)] = (R)pfv[j >> 2].s[i];
                    uint32_t rq, xq;
                    const bool ahead = where(u + 16, &rq, &xq);
                    pfv[j >> 2].v = *reinterpret_cast<GlobalPtr<const V4>>(src + (ahead ? (size_t)rq * t.stride + xq : (size_t)0));
                }
            } else {   // residual pipeline: park what arrived for position u + 8, request position u + 16.  Every global access
                // of the step is issued unconditionally (an off-grid lane reads the subgrid's first sample, parks it in a
                // slot nobody reads and stores to `sink`): with a load or store under a branch the compiler cannot count
                // the accesses in flight and waits for ALL of them (s_waitcnt vmcnt(0)) before it parks a residual
                uint32_t rq, xq;
                s_in[lane][(u + 8) & (kRing - 1)] = (R)pf[j];
                const bool ahead = where(u + 16, &rq, &xq);
                pf[j] = (int32_t)src[ahead ? (size_t)rq * t.stride + xq : (size_t)0];
            }
            uint32_t r, ux;
            S st_value = 0;
            GlobalPtr<S> st_ptr = as_global((S*)a.sink + lane * 4 + 3);
            if (where(u, &r, &ux)) {
                const int32_t x = (int32_t)ux;
                const int32_t gwi = (int32_t)gw;
                if (x == 0) {
                    const uint32_t round = (uint32_t)u >> log2dp;
                    pb1 = ((round - wrap1) << log2dp) & (RO - 1);
                    pb2 = ((round - wrap2) << log2dp) & (RO - 1);
                    if (r == 0) {
                        w = n = nw = 0;
                    } else {
                        w = n = nw = prev[pb1];
                        te_w = 0;
                        te_n = s_true_err[0];
                        te_nw = te_n;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { se_n_w[i] = s_sub_err[i][0]; se_nw_ww[i] = se_n_w[i]; }
                        const int e1 = gw <= 1 ? 0 : 1;
                        te_ne = s_true_err[e1];
#pragma unroll
                        for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i][e1];
                    }
                }
                // LDS reads that do not depend on this step's arithmetic
                const int32_t res = s_in[lane][u & (kRing - 1)];
                const int32_t p_ne = prev[(pb1 + ux + 1) & (RO - 1)];
                const int32_t p_nn = prev2[(pb2 + ux) & (RO - 1)];
                const int32_t x2 = min(x + 2, gwi - 1);
                const int32_t l_te = s_true_err[x2];
                uint32_t l_se[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) l_se[i] = s_sub_err[i][x2];

                const int32_t ne = (r == 0 || x + 1 >= gwi) ? n : p_ne;
                const int32_t nn = r >= 2 ? p_nn : n;
                const int32_t n3 = n * 8, nw3 = nw * 8, ne3 = ne * 8, w3 = w * 8, nn3 = nn * 8;
                int32_t subpred[4];
                subpred[0] = w3 + ne3 - n3;
                subpred[1] = n3 - (((te_w + te_n + te_ne) * wp0) >> 5);
                subpred[2] = w3 - (((te_w + te_n + te_nw) * wp1) >> 5);
                subpred[3] = n3 - ((te_nw * wp2 + te_n * wp3 + te_ne * wp4 + (nn3 - n3) * wp5 + (nw3 - w3) * wp6) >> 5);
                uint32_t weight[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t err_sum = se_nw_ww[i] + se_n_w[i] + se_ne[i];
                    const uint32_t tt = (err_sum + 1u) >> 5;
                    const uint32_t shift = tt ? 31u - (uint32_t)__builtin_clz(tt) : 0u;
                    weight[i] = 4u + ((ww[i] * s_div[(err_sum >> shift) + 1]) >> shift);
                }
                uint32_t sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                const uint32_t log_weight = 31u - (uint32_t)__builtin_clz(sum_weights >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) weight[i] >>= log_weight;
                sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                int32_t acc = (int32_t)(sum_weights >> 1) - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += subpred[i] * (int32_t)weight[i];
                int32_t prediction = (int32_t)(((int64_t)acc * (int64_t)(int32_t)s_div[sum_weights]) >> 24);
                if (((te_n ^ te_w) | (te_n ^ te_nw)) <= 0) {
                    const int32_t mn = min(min(n3, w3), ne3), mx = max(max(n3, w3), ne3);
                    prediction = prediction < mn ? mn : (prediction > mx ? mx : prediction);
                }
                const int32_t pred = (prediction + 3) >> 3;
                const S diff = Wrap<S>::add(Wrap<S>::mul((S)res, (S)t.mul), (S)t.off);
                const S value = Wrap<S>::add(diff, (S)pred);
                st_value = value;
                st_ptr = as_global((S*)t.base + (size_t)r * t.stride + ux);
                const int32_t sample = (int32_t)value;
                s_out[lane][u & (RO - 1)] = (R)sample;

                const int32_t s8 = sample * 8;
                const int32_t true_err = prediction - s8;
                out_of_range |= (uint32_t)true_err + (1u << 19) >= (1u << 20) || (uint32_t)sample + (1u << 17) >= (1u << 18);
                uint32_t sub_err[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int32_t d = subpred[i] - s8;
                    sub_err[i] = ((uint32_t)(d < 0 ? -d : d) + 3u) >> 3;
                }
                s_true_err[x] = true_err;
#pragma unroll
                for (int i = 0; i < 4; ++i) s_sub_err[i][x] = sub_err[i];
                // SelfCorrectingPredictor::record + Properties::record; at a row's last column this state is dead
                // (the row start sets all of it), so the x + 1 < gw test of the reference is not needed
                const bool last2 = x + 2 >= gwi;
                te_w = true_err;
                te_nw = te_n;
                te_n = te_ne;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    se_nw_ww[i] = se_n_w[i];
                    se_n_w[i] = se_ne[i] + sub_err[i];
                }
                // row 0 reads the zeros its error rows still hold two columns ahead: the same as keeping its zeros
                te_ne = last2 ? te_n : l_te;
#pragma unroll
                for (int i = 0; i < 4; ++i) se_ne[i] = last2 ? se_n_w[i] : l_se[i];
                w = sample;
                nw = r == 0 ? sample : n;
                n = r == 0 ? sample : p_ne;
            }
            if constexpr (VEC) {
                sbuf.s[j & 3] = st_value;
                if ((j & 3) == 3) {
                    // columns x - 3 .. x of one row (or the sink: a lane is on the grid for all four steps of a group or none)
                    const GlobalPtr<S> g4 = st_ptr - 3;
                    *reinterpret_cast<GlobalPtr<V4>>(g4) = sbuf.v;
                }
            } else {
                *st_ptr = st_value;
            }
            lds_step_boundary();
        }
        if (__builtin_amdgcn_ballot_w64(out_of_range) != 0) {
            if (lane == 0) wave_flags[blockIdx.x] = 1;
            return;
        }
    }
}

// ---- the same pass for launches in which every wave has D = 4 (every subgrid at most 256 columns wide: all of a frame with
// 256 x 256 groups), round 6.  One wave per SIMD issues at most one instruction per ~5 cycles (tools/valu_cost_probe.hip) and a
// wave is one serial chain, so the pass lasts (instructions per step) x (steps of the longest wave): the step above is 185 VALU
// + 25 LDS instructions.  This form computes the same numbers with ~40 % fewer of them:
//  * no address arithmetic in the step.  A lane's rings are indexed by the STEP (s & 15), not by its stream position: the value
//    lane k needs from the row above (stream position u + 1 of lane k - 1, or of lane P - 1 one round earlier for lane 0) was
//    written at step s + 1 - D whatever k is, the one two rows up at s - 2 D; with the step loop unrolled by the ring length
//    every slot is an immediate offset on a per-lane base register;
//  * the self-correcting predictor's error rows (`true_err_row` / `subpred_err_row`, predictor.rs:176-190) are a hand-over from
//    row r - 1 to row r two columns behind it, like the samples: four-slot rings per lane (one ds_write_b128 + ds_write_b32 per
//    step, one ds_read_b128 + ds_read_b32) instead of five x-indexed rows; row 0 reads a ring of zeros;
//  * everything read from LDS that does not depend on the step's arithmetic is requested one step ahead (no wait at the top);
//  * the state update is unconditional (an off-grid lane computes garbage that the next row start overwrites; only the global
//    store and the range flag look at `on`): the register rotation is a renaming, not v_movs under an exec mask;
//  * arithmetic: `ww[i] * DIV_LOOKUP[j]` is a table of its own; shift = max(0, 26 - clz(err_sum + 1)) (one saturating
//    subtraction for the reference's `(err_sum + 1) >> 5` / floor(log2) pair, predictor.rs:372-380); samples are carried as
//    8 s + 2^23, which makes every sub-predictor a non-negative 24-bit number: |subpred - 8 s| is one v_sad_u32 and the
//    weighted sum four v_mad_u32_u24, the bias leaves again through - (sum_weights << 23); the clamp is one v_med3_i32.
// Range guard and redo pass as above.
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {   // low 24 bits of a, b
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int32_t med3_i32(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ int32_t mul_i24(int32_t a, int32_t b) {   // |a|, |b| < 2^23
    int32_t d;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "s"(b), "v"(a));
    return d;
}
__device__ __forceinline__ int32_t mad_i24(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}

template <typename S, bool VEC>
__global__ __launch_bounds__(64) void predict_lanes_wp4_kernel(PredArgs a, const PredWave* waves, const PredSrc* srcs,
                                                               uint32_t* wave_flags) {
    constexpr int RO = kRing, E = 4;
    constexpr int32_t B = 1 << 23;
    static_assert(RO == 16, "the step loop is unrolled by the ring length");
    using R = typename std::conditional<sizeof(S) == 2, int16_t, int32_t>::type;
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t s_wdiv[4][66];          // ww[i] * DIV_LOOKUP[j]
    __shared__ uint32_t s_div[65];
    __shared__ R s_out[65][RO + 1];             // finished samples of the lane's rows, slot = step & 15; [64] = zeros
    __shared__ R s_in[64][kRing + 1];           // residuals requested ahead, slot = step & 15
    __shared__ U4 s_se[65][E + 1];              // sub_err[4] of the lane's last four samples, slot = step & 3 (+1: bank spread); [64] = zeros
    __shared__ int32_t s_te[65][E + 1];         // true_err likewise
    const uint32_t lane = threadIdx.x;
    s_div[lane] = div_lookup_dev(lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) s_wdiv[i][lane] = (uint32_t)a.wp[7 + i] * div_lookup_dev(lane);
    if (lane == 0) {
        s_div[64] = div_lookup_dev(64);
#pragma unroll
        for (int i = 0; i < 4; ++i) s_wdiv[i][64] = (uint32_t)a.wp[7 + i] * div_lookup_dev(64);
    }
    for (uint32_t i = lane; i < 65 * (E + 1); i += 64) {
        (&s_se[0][0])[i] = U4{0, 0, 0, 0};
        (&s_te[0][0])[i] = 0;
    }
    if (lane < RO + 1) s_out[64][lane] = 0;
    __syncthreads();
    const uint32_t item = blockIdx.x;
    const PredWave wv = waves[item];
    switch (__builtin_amdgcn_readfirstlane((int)(wv.pad[1] & 3u))) {   // JXLGPU_PRED_PRIO (see predict_lanes_narrow_kernel); bit 8: all16, below
        case 3: __builtin_amdgcn_s_setprio(3); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        default: break;
    }
    const uint32_t log2p = wv.log2p, log2dp = wv.log2dp, P = 1u << log2p, DPm1 = (1u << log2dp) - 1u;
    const uint32_t slot = lane >> log2p, k = lane & (P - 1);
    const bool have_tile = slot < wv.count;
    const PredTile t = a.tiles[wv.first + (have_tile ? slot : 0)];
    const GlobalPtr<const S> src = as_global((const S*)srcs[wv.first + (have_tile ? slot : 0)].src);
    const int32_t gw = have_tile ? (int32_t)t.gw : 0;
    const uint32_t gh = have_tile ? t.gh : 0;
    if (lane == 0) wave_flags[item] = 0;
    const uint32_t lane0 = lane & ~(P - 1);
    const R* my_out = s_out[lane];
    R* my_out_w = s_out[lane];
    R* my_in = s_in[lane];
    const R* prev = s_out[lane0 + ((k + P - 1) & (P - 1))];
    const R* prev2 = s_out[lane0 + ((k + 2 * P - 2) & (P - 1))];
    const U4* prev_se = s_se[lane0 + ((k + P - 1) & (P - 1))];
    const int32_t* prev_te = s_te[lane0 + ((k + P - 1) & (P - 1))];
    (void)my_out;
    const U4* pse = s_se[64];        // the ring row r reads its NE errors from: the previous lane's, or the zeros (row 0)
    const int32_t* pte = s_te[64];
    const int32_t wp0 = a.wp[0], wp1 = a.wp[1], wp2 = a.wp[2], wp3 = a.wp[3], wp4 = a.wp[4], wp5 = a.wp[5], wp6 = a.wp[6];
    const int32_t offb = t.off;      // (the bias leaves the prediction as B / 8 = 2^20: see `value` below)

    int32_t w3b = B, n3b = B, nw3b = B;                      // 8 * sample + B of W, N, NW
    int32_t te_w = 0, te_nw = 0, te_n = 0, te_ne = 0;
    uint32_t se_nw_ww[4] = {0, 0, 0, 0}, se_n_w[4] = {0, 0, 0, 0}, se_ne[4] = {0, 0, 0, 0};
    bool r0 = true, r_ge2 = false, row_ok = false;
    GlobalPtr<S> row_base = as_global((S*)a.sink);
    // requested one step ahead
    int32_t nx_res = 0, nx_pne = 0, nx_pnn = 0, nx_te = 0;
    U4 nx_se = U4{0, 0, 0, 0};
    int32_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef uint32_t RawV2 __attribute__((ext_vector_type(2)));
    typedef uint32_t RawV4 __attribute__((ext_vector_type(4)));
    using V4 = typename std::conditional<sizeof(S) == 2, RawV2, RawV4>::type;
    union Pack4 { V4 v; S s[4]; };
    Pack4 pfv[2], packs[4];
    pfv[0].v = pfv[1].v = V4{};
    packs[0].v = packs[1].v = packs[2].v = packs[3].v = V4{};
    const bool all16 = __builtin_amdgcn_readfirstlane((int)(wv.pad[1] >> 8)) != 0;
    // (a subgrid whose width is not a multiple of 16 stores its last block up to 12 steps after its last sample)
    const int32_t steps = (int32_t)wv.steps + (VEC && !all16 ? 12 : 0);
    const int32_t u0 = -4 * (int32_t)k;
    auto where = [&](int32_t q, uint32_t* r, uint32_t* x) -> bool {
        *r = k + (((uint32_t)q >> log2dp) << log2p);
        *x = (uint32_t)q & DPm1;
        return q >= 0 && *r < gh && (int32_t)*x < gw;
    };
    for (int32_t s0 = -16; s0 < steps; s0 += 16) {
        bool out_of_range = false;
        uint32_t worst = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int32_t u = u0 + s0 + j;
            // ---- residual pipeline (global memory): park what arrived for the step 8 ahead, request the one 16 ahead; always issued
            if constexpr (VEC) {
                if ((j & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) my_in[(j + 8 + i) & 15] = (R)pfv[(j >> 2) & 1].s[i];
                    uint32_t rq, xq;
                    const bool ahead = where(u + 16, &rq, &xq);
                    pfv[(j >> 2) & 1].v = *reinterpret_cast<GlobalPtr<const V4>>(src + (ahead ? (size_t)rq * t.stride + xq : (size_t)0));
                }
            } else {
                uint32_t rq, xq;
                my_in[(j + 8) & 15] = (R)pf[j & 7];
                const bool ahead = where(u + 16, &rq, &xq);
                pf[j & 7] = (int32_t)src[ahead ? (size_t)rq * t.stride + xq : (size_t)0];
            }
            const int32_t x = (int32_t)((uint32_t)u & DPm1);
            // what this step was sent ahead
            const int32_t res = nx_res, p_ne = nx_pne, p_nn = nx_pnn;
            int32_t l_te = nx_te;
            U4 l_se = nx_se;
            // a lane's column is congruent to the step modulo 4 (D = 4 and DP is a multiple of 4): rows start at every fourth step only
            if ((j & 3) == 0 && x == 0) {
                // a row starts (also for positions before the lane's first row: row_ok stays false)
                const uint32_t r = k + (((uint32_t)u >> log2dp) << log2p);
                row_ok = u >= 0 && r < gh;
                r0 = r == 0;
                r_ge2 = r >= 2;
                row_base = row_ok ? as_global((S*)t.base + (size_t)r * t.stride) : as_global((S*)a.sink);
                // columns 0, 1 and 2 of the row above: written by the previous lane at steps s - 4, s - 3, s - 2; row 0 reads the
                // rings of zeros instead (no branch: every value below is a load into its register)
                const R* ps = r0 ? s_out[64] : prev;
                pse = r0 ? s_se[64] : prev_se;
                pte = r0 ? s_te[64] : prev_te;
                const int32_t c0 = ps[(j + 12) & 15];
                w3b = n3b = nw3b = c0 * 8 + B;
                te_w = 0;
                te_n = te_nw = pte[(j + 0) & 3];
                const U4 e0 = pse[(j + 0) & 3];
                const U4 e1v = pse[(j + 1) & 3];
                const int32_t t1 = pte[(j + 1) & 3];
                const bool one = gw <= 1;
                te_ne = one ? te_n : t1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    se_n_w[i] = se_nw_ww[i] = e0[i];
                    se_ne[i] = one ? e0[i] : e1v[i];
                }
                l_te = pte[(j + 2) & 3];
                l_se = pse[(j + 2) & 3];
            }
            // ---- requests for step s + 1 (NE sample: written at s - 2; NN: at s - 7; errors two columns ahead of it: at s - 1)
            nx_res = my_in[(j + 1) & 15];
            nx_pne = prev[(j + 14) & 15];
            nx_pnn = prev2[(j + 9) & 15];
            nx_te = pte[(j + 3) & 3];
            nx_se = pse[(j + 3) & 3];

            const bool on = row_ok && x < gw;
            const int32_t pne3b = p_ne * 8 + B;
            const int32_t ne3b = (r0 || x + 1 >= gw) ? n3b : pne3b;
            const int32_t nn3b = r_ge2 ? p_nn * 8 + B : n3b;
            uint32_t sp[4];   // sub-predictors + B
            sp[0] = (uint32_t)(w3b + ne3b - n3b);
            // (24-bit multiplies: |true_err| < 2^19, the sample differences < 2^21, the WpHeader factors < 32)
            const int32_t te_wn = te_w + te_n;
            sp[1] = (uint32_t)(n3b - (mul_i24(te_wn + te_ne, wp0) >> 5));
            sp[2] = (uint32_t)(w3b - (mul_i24(te_wn + te_nw, wp1) >> 5));
            sp[3] = (uint32_t)(n3b - (mad_i24(nw3b - w3b, wp6, mad_i24(nn3b - n3b, wp5, mad_i24(te_ne, wp4, mad_i24(te_n, wp3, mul_i24(te_nw, wp2))))) >> 5));
            uint32_t weight[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t err_sum = se_nw_ww[i] + se_n_w[i] + se_ne[i];
                const uint32_t shift = __builtin_elementwise_sub_sat(26u, (uint32_t)__builtin_clz(err_sum + 1u));
                weight[i] = 4u + (s_wdiv[i][(err_sum >> shift) + 1] >> shift);
            }
            uint32_t sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
            const uint32_t log_weight = 27u - (uint32_t)__builtin_clz(sum_weights);
#pragma unroll
            for (int i = 0; i < 4; ++i) weight[i] >>= log_weight;
            sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
            const uint32_t dv = s_div[sum_weights];
            uint32_t accb = (sum_weights >> 1) - (sum_weights << 23) - 1u;
#pragma unroll
            for (int i = 0; i < 4; ++i) accb = mad_u24(sp[i], weight[i], accb);
            const int32_t acc = (int32_t)accb;
            int32_t predb = (int32_t)(((int64_t)acc * (int64_t)(int32_t)dv) >> 24) + B;
            {
                const int32_t mn = min(min(n3b, w3b), ne3b), mx = max(max(n3b, w3b), ne3b);
                const int32_t clamped = med3_i32(predb, mn, mx);
                predb = ((te_n ^ te_w) | (te_n ^ te_nw)) <= 0 ? clamped : predb;
            }
            // (prediction + 3) >> 3 carries B / 8 = 2^20 with it: nothing, for 16-bit samples; taken off for 32-bit ones
            int32_t pred = (predb + 3) >> 3;
            if constexpr (sizeof(S) == 4) pred -= B / 8;
            const S diff = Wrap<S>::add(Wrap<S>::mul((S)res, (S)t.mul), (S)offb);
            const S value = Wrap<S>::add(diff, (S)pred);
            const int32_t sample = (int32_t)value;
            my_out_w[j & 15] = (R)sample;
            const int32_t s8b = sample * 8 + B;
            const int32_t true_err = predb - s8b;
            // range guard: the largest excursion of the group of four steps, looked at where the group ends (VEC: a lane is on the grid for
            // all four steps of a group or none); 16-bit samples cannot leave the sample range
            uint32_t exc = (uint32_t)true_err + (1u << 19);
            if constexpr (sizeof(S) == 4) exc = max(exc, (uint32_t)sample + (1u << 17) >= (1u << 18) ? (1u << 20) : 0u);
            if constexpr (VEC) {
                worst = (j & 3) == 0 ? exc : max(worst, exc);
                if ((j & 3) == 3) out_of_range |= on && worst >= (1u << 20);
            } else {
                out_of_range |= on && exc >= (1u << 20);
            }
            U4 sub_err;
#pragma unroll
            for (int i = 0; i < 4; ++i) sub_err[i] = sad_u32(sp[i], (uint32_t)s8b, 3u) >> 3;
            s_te[lane][j & 3] = true_err;
            s_se[lane][j & 3] = sub_err;
            // SelfCorrectingPredictor::record + Properties::record (unconditional: see above)
            const bool last2 = x + 2 >= gw;
            te_w = true_err;
            te_nw = te_n;
            te_n = te_ne;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                se_nw_ww[i] = se_n_w[i];
                se_n_w[i] = se_ne[i] + sub_err[i];
            }
            te_ne = last2 ? te_n : l_te;
#pragma unroll
            for (int i = 0; i < 4; ++i) se_ne[i] = last2 ? se_n_w[i] : l_se[i];
            w3b = s8b;
            nw3b = r0 ? s8b : n3b;
            n3b = r0 ? s8b : pne3b;

            if constexpr (VEC) {
                // Samples leave SIXTEEN at a time: a lane keeps the four groups of its current 16-column block (packs[g], g = the group
                // of the unrolled loop that wrote it) and stores them back to back when the block ends, i.e. at the end of the group in
                // which its column is 15 (mod 16) — a different one of the four for lanes k, k + 1, k + 2, k + 3.  With one 8-byte
                // store per lane per group, 64 lanes = 64 rows = 64 cache lines per instruction, each line took 16 visits 4 steps
                // apart and L2 wrote a partial sector out for nearly every one: WRITE_SIZE 890 MB for the 200 MB of an 8K frame
                // (profiles/r06_cfg3_pmc.txt).  Groups past the subgrid's last column, and lanes between rows, store to the sink.
                packs[(j >> 2) & 3].s[j & 3] = value;
                if ((j & 3) == 3) {
                    const int g = (j >> 2) & 3;   // (a constant of the unrolled loop)
                    const int32_t x0 = x - 15;
                    const bool block_ends = row_ok && (x & 12) == 12 && x0 < gw;
                    const GlobalPtr<S> sink = as_global((S*)a.sink + lane * 4);
                    if (all16) {   // (wave-uniform) every subgrid of the wave is a multiple of 16 columns wide: one test for the four groups
                        const GlobalPtr<S> b16 = block_ends ? row_base + x0 : sink;
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<GlobalPtr<V4>>(b16 + 4 * q) = packs[(g + 1 + q) & 3].v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const GlobalPtr<S> b4 = (block_ends && x0 + 4 * q < gw) ? row_base + (x0 + 4 * q) : sink;
                            *reinterpret_cast<GlobalPtr<V4>>(b4) = packs[(g + 1 + q) & 3].v;
                        }
                    }
                }
            } else {
                const GlobalPtr<S> g1 = on ? row_base + x : as_global((S*)a.sink + lane * 4 + 3);
                *g1 = value;
            }
            lds_step_boundary();
        }
        if (__builtin_amdgcn_ballot_w64(out_of_range) != 0) {
            if (lane == 0) wave_flags[item] = 1;
            return;
        }
    }
}

// ---------------------------------------------------------------- device: M3, delta-palette predictor pass
// The predictor pass of Palette::inverse_inner (palette.rs:112-142): ONE PredictorState over the
// whole channel, so the wavefront spans the image: workgroup b owns rows [256 b, 256 b + 256), lane r
// trails lane r-1 by three columns, and lane 0 of workgroup b trails the last lane of workgroup
// b-1 the same way through a progress counter in global memory (release/acquire at agent scope).
// All workgroups of the launch are resident together (a few dozen), lower bands are dispatched
// first, and the spin is bounded: if it ever expires the kernel raises `fail` and the host returns
// JXLGPU_ERR_DEVICE instead of hanging.  `rec` holds the recorded i32 `sample_value`s
// (palette.rs:130-139), which for i16 buffers may differ from the stored, truncated samples; the
// self-correcting predictor's two error rows are global arrays overwritten in place, as in the
// reference.
struct DeltaArgs {
    void* grid[3];
    int32_t* rec[3];
    uint32_t stride[3];
    const uint8_t* need_delta;
    uint32_t width, height, d_pred;
    int32_t wp[11];
    int32_t* true_err[3];     // width
    uint32_t* sub_err[3];     // 4 x width
    uint32_t* progress;       // [3][bands]: columns finished by the band's last row
    uint32_t bands;
    int* fail;
};

template <typename S>
__global__ __launch_bounds__(256) void palette_delta_kernel(DeltaArgs a) {
    const uint32_t band = blockIdx.x, ch = blockIdx.y;
    const uint32_t r = threadIdx.x;
    const uint32_t y = band * 256 + r;
    const uint32_t gw = a.width;
    const uint32_t rows = min(256u, a.height - band * 256);
    const bool active = r < rows;
    S* row = (S*)a.grid[ch] + (size_t)y * a.stride[ch];
    int32_t* rec = a.rec[ch] + (size_t)y * gw;
    const int32_t* prev = rec - gw;
    const int32_t* prev2 = prev - gw;
    const uint8_t* flags = a.need_delta + (size_t)y * gw;
    int32_t* s_true_err = a.true_err[ch];
    uint32_t* s_sub_err = a.sub_err[ch];  // [i * gw + x]
    uint32_t* my_progress = a.progress + ch * a.bands + band;
    const uint32_t* up_progress = a.progress + ch * a.bands + band - 1;
    const bool sc_on = a.d_pred == 6;

    int32_t w = 0, n = 0, nw = 0, ww1 = 0, ww2 = 0;
    int32_t te_w = 0, te_nw = 0, te_n = 0, te_ne = 0;
    uint32_t se_nw_ww[4] = {0, 0, 0, 0}, se_n_w[4] = {0, 0, 0, 0}, se_ne[4] = {0, 0, 0, 0};

    const uint32_t steps = gw + 3 * (rows - 1);
    bool gave_up = false;
    for (uint32_t s = 0; s < steps; ++s) {
        if (band > 0 && r == 0 && s < gw && !gave_up) {
            // row y-1 must have finished columns <= s + 2
            const uint32_t need = min(gw, s + 3);
            uint32_t spins = 0;
            while (__hip_atomic_load(up_progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {  // ~0.2 s: never in a healthy run; results are void, no hang
                    atomicExch(a.fail, 1);
                    gave_up = true;
                    break;
                }
            }
        }
        __syncthreads();
        const int32_t x = (int32_t)s - 3 * (int32_t)r;
        if (active && x >= 0 && x < (int32_t)gw) {
            const bool no_prev = y == 0;
            if (x == 0) {
                if (no_prev) {
                    w = n = nw = 0;
                } else {
                    w = n = nw = prev[0];
                    if (sc_on) {
                        te_w = 0;
                        te_n = s_true_err[0];
                        te_nw = te_n;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { se_n_w[i] = s_sub_err[i * gw]; se_nw_ww[i] = se_n_w[i]; }
                        if (gw <= 1) {
                            te_ne = te_n;
#pragma unroll
                            for (int i = 0; i < 4; ++i) se_ne[i] = se_n_w[i];
                        } else {
                            te_ne = s_true_err[1];
#pragma unroll
                            for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i * gw + 1];
                        }
                    }
                }
            }
            const int32_t ne = (no_prev || x + 1 >= (int32_t)gw) ? n : prev[x + 1];
            const int32_t nee = (no_prev || x + 2 >= (int32_t)gw) ? ne : prev[x + 2];
            const int32_t nn = y >= 2 ? prev2[x] : n;
            const int32_t ww = x >= 2 ? ww2 : w;

            int64_t sc_prediction = 0, subpred[4] = {0, 0, 0, 0};
            if (sc_on) {
                const int64_t tw = te_w, tnw = te_nw, tn = te_n, tne = te_ne;
                const int64_t n3 = (int64_t)n * 8, nw3 = (int64_t)nw * 8, ne3 = (int64_t)ne * 8, w3 = (int64_t)w * 8,
                              nn3 = (int64_t)nn * 8;
                subpred[0] = w3 + ne3 - n3;
                subpred[1] = n3 - (((tw + tn + tne) * (int64_t)a.wp[0]) >> 5);
                subpred[2] = w3 - (((tw + tn + tnw) * (int64_t)a.wp[1]) >> 5);
                subpred[3] = n3 - ((tnw * (int64_t)a.wp[2] + tn * (int64_t)a.wp[3] + tne * (int64_t)a.wp[4] +
                                    (nn3 - n3) * (int64_t)a.wp[5] + (nw3 - w3) * (int64_t)a.wp[6]) >> 5);
                uint32_t weight[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t err_sum = se_nw_ww[i] + se_n_w[i] + se_ne[i];
                    const uint64_t t = ((uint64_t)err_sum + 1) >> 5;
                    const uint32_t shift = t ? 63u - (uint32_t)__builtin_clzll(t) : 0u;
                    weight[i] = 4 + (((uint32_t)a.wp[7 + i] * div_lookup_dev((err_sum >> shift) + 1)) >> shift);
                }
                uint32_t sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                const uint32_t log_weight = 31u - (uint32_t)__builtin_clz(sum_weights >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) weight[i] >>= log_weight;
                sum_weights = weight[0] + weight[1] + weight[2] + weight[3];
                int64_t acc = ((int64_t)sum_weights >> 1) - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += subpred[i] * (int64_t)weight[i];
                int64_t prediction = (acc * (int64_t)div_lookup_dev(sum_weights)) >> 24;
                if (((tn ^ tw) | (tn ^ tnw)) <= 0) {
                    const int64_t mn = min(min(n3, w3), ne3), mx = max(max(n3, w3), ne3);
                    prediction = prediction < mn ? mn : (prediction > mx ? mx : prediction);
                }
                sc_prediction = prediction;
            }

            int32_t sample_value = (int32_t)row[x];
            if (flags[x]) {
                int32_t pred;
                const int64_t N = n, W = w, NW = nw;
                switch (a.d_pred) {
                    case 0: pred = 0; break;
                    case 1: pred = w; break;
                    case 2: pred = n; break;
                    case 3: pred = (int32_t)((W + N) / 2); break;
                    case 4: {
                        const int64_t dn = N > NW ? N - NW : NW - N, dw = W > NW ? W - NW : NW - W;
                        pred = dn < dw ? w : n;
                        break;
                    }
                    case 5: {
                        const int64_t g = N + W - NW, lo = W < N ? W : N, hi = W > N ? W : N;
                        pred = (int32_t)(g < lo ? lo : (g > hi ? hi : g));
                        break;
                    }
                    case 6: pred = (int32_t)((sc_prediction + 3) >> 3); break;
                    case 7: pred = ne; break;
                    case 8: pred = nw; break;
                    case 9: pred = ww; break;
                    case 10: pred = (int32_t)((W + NW) / 2); break;
                    case 11: pred = (int32_t)((N + NW) / 2); break;
                    case 12: pred = (int32_t)((N + (int64_t)ne) / 2); break;
                    default:
                        pred = (int32_t)((6 * N - 2 * (int64_t)nn + 7 * W + (int64_t)ww + (int64_t)nee + 3 * (int64_t)ne + 8) / 16);
                        break;
                }
                sample_value = (int32_t)((uint32_t)sample_value + (uint32_t)pred);  // palette.rs:133
                row[x] = (S)sample_value;                                            // S::from_i32
            }
            rec[x] = sample_value;

            if (sc_on) {
                const int64_t s8 = (int64_t)sample_value * 8;
                const int64_t true_err = sc_prediction - s8;
                uint32_t sub_err[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t d = subpred[i] - s8;
                    sub_err[i] = (uint32_t)(((uint64_t)(d < 0 ? -d : d) + 3) >> 3);
                }
                s_true_err[x] = (int32_t)true_err;
#pragma unroll
                for (int i = 0; i < 4; ++i) s_sub_err[i * gw + x] = sub_err[i];
                if (x + 1 < (int32_t)gw) {
                    te_w = (int32_t)true_err;
                    te_nw = te_n;
                    te_n = te_ne;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        se_nw_ww[i] = se_n_w[i];
                        se_n_w[i] = se_ne[i] + sub_err[i];
                    }
                    if (x + 2 >= (int32_t)gw) {
                        te_ne = te_n;
#pragma unroll
                        for (int i = 0; i < 4; ++i) se_ne[i] = se_n_w[i];
                    } else if (!no_prev) {
                        te_ne = s_true_err[x + 2];
#pragma unroll
                        for (int i = 0; i < 4; ++i) se_ne[i] = s_sub_err[i * gw + x + 2];
                    }
                }
            }
            if (x + 1 < (int32_t)gw) {
                ww2 = ww1;
                ww1 = sample_value;
                w = sample_value;
                if (no_prev) {
                    nw = sample_value;
                    n = sample_value;
                } else {
                    nw = n;
                    n = prev[x + 1];
                }
            }
            if (r == rows - 1) {
                // publish: everything this row wrote up to column x is visible before the counter moves
                __threadfence();
                __hip_atomic_store(my_progress, (uint32_t)x + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- device: int -> float
struct ToFloatArgs {
    const void* in[3];
    float* out[3];
    uint32_t in_stride[3], out_stride, width, height;
    uint32_t cw[3], ch[3];   // per-channel sizes (chroma-subsampled frames: smaller than width x height)
    uint32_t xyb, is_i16, bit_depth, float_sample, exp_bits;
    float m[3];
};

// convert_to_float_modular_xyb (jxl-render/src/image.rs:148-189) or BitDepth::parse_integer_sample
// (jxl-image/src/lib.rs:458-494); output planes in framebuffer order.
__global__ __launch_bounds__(256) void to_float_kernel(ToFloatArgs a) {
    uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.width) return;
    int32_t v[3];
    bool in_c[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        in_c[c] = x < a.cw[c] && y < a.ch[c];
        size_t i = (size_t)y * a.in_stride[c] + x;
        v[c] = !in_c[c] ? 0 : (a.is_i16 ? (int32_t)((const int16_t*)a.in[c])[i] : ((const int32_t*)a.in[c])[i]);
    }
    size_t o = (size_t)y * a.out_stride + x;
    if (a.xyb) {
        int32_t yv = v[0], xv = v[1], bv = v[2], bs;
        if (a.is_i16) {
            int32_t t = bv + yv;
            bs = t > 32767 ? 32767 : (t < -32768 ? -32768 : t);
        } else {
            int64_t t = (int64_t)bv + yv;
            bs = t > 2147483647ll ? 2147483647 : (t < -2147483648ll ? (int32_t)-2147483648ll : (int32_t)t);
        }
        a.out[0][o] = (float)xv * a.m[0];
        a.out[1][o] = (float)yv * a.m[1];
        a.out[2][o] = (float)bs * a.m[2];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float r;
            if (!a.float_sample) {
                int32_t div = (int32_t)((1u << a.bit_depth) - 1);
                r = (float)v[c] / (float)div;
            } else {
                uint32_t s = (uint32_t)v[c];
                uint32_t mantissa_bits = a.bit_depth - a.exp_bits - 1;
                uint32_t mantissa_mask = (1u << mantissa_bits) - 1;
                uint32_t exp_mask = ((1u << (a.bit_depth - 1)) - 1) ^ mantissa_mask;
                uint32_t is_signed = (s & (1u << (a.bit_depth - 1))) != 0;
                uint32_t mantissa = s & mantissa_mask;
                int32_t exp = (int32_t)((s & exp_mask) >> mantissa_bits) - ((1 << (a.exp_bits - 1)) - 1);
                if (mantissa_bits < 23) mantissa <<= (23 - mantissa_bits);
                else if (mantissa_bits > 23) mantissa >>= (mantissa_bits - 23);
                r = __uint_as_float((is_signed << 31) | ((uint32_t)(exp + 127) << 23) | mantissa);
            }
            if (in_c[c]) a.out[c][o] = r;
        }
    }
}

// ---------------------------------------------------------------- host: bookkeeping
struct Grid {
    int buf;                 // >= 0 channel buffer, < 0: ~meta index
    uint32_t x0, y0, w, h;
    int loc;                 // which working copy holds the data (channel buffers only)
    std::vector<int> members;  // palette: buffers of the merged member channels
    // ModularChannelInfo (jxl-modular/src/lib.rs:157-191): shifts accumulated by Squeeze (-1: unshiftable meta
    // channel) and the size of the untransformed channel, which fixes the number of groups
    int hshift = 0, vshift = 0;
    uint32_t orig_w = 0, orig_h = 0;
    int fwd_step = -1;       // residual rectangle: the (forward) Squeeze step that produced it — step 0's are the largest and the last to be consumed
    // inverse Squeeze: the chain (SqueezePlan::epoch, workgroup) of small levels the step that last wrote this rectangle may be queued in
    uint32_t chain_epoch = 0;
    int chain_slot = -1;
};

struct ModularState {
    JxlGpuModularDesc desc;  // scalar copy
    size_t esz = 4;
    std::vector<uint32_t> cw, ch;             // channel buffer sizes
    std::vector<void*> orig;                  // uploaded buffers (never modified)
    std::vector<void*> work[4];               // three working copies; [3] aliases `orig` (read-only location)
    std::vector<void*> meta;
    std::vector<void*> meta_work;             // writable copies (the predictor pass runs on the palette tables too)
    std::vector<uint32_t> mw, mh;
    std::vector<JxlGpuTransform> transforms;
    std::vector<std::vector<JxlGpuSqueezeStep>> explicit_steps;  // per transform (empty = default)
    std::vector<std::vector<JxlGpuSqueezeStep>> steps;
    std::vector<int> final_loc;               // after the last inverse run
    int* d_flag = nullptr;
    uint8_t* need_delta = nullptr;   // palette slow path: index < nb_deltas
    size_t need_delta_bytes = 0;
    uint32_t* delta_aux = nullptr;   // fail flag, band progress, error rows, recorded i32 samples
    size_t delta_aux_words = 0;
    void* chk = nullptr;                      // segment link values of the squeeze step in flight
    size_t chk_bytes = 0;
    int* d_redo = nullptr;                    // lines redone serially (diagnostics)
    PredTile* pred_tiles = nullptr;           // M4: every (group, channel) subgrid of the frame, longest first
    uint32_t n_pred_tiles = 0, pred_err_w = 1;   // ... of which the first n_pred_wide go through the workgroup-per-subgrid kernel
    uint32_t n_pred_wide = 0, n_pred_waves = 0, pred_lane_err_w = 256;
    PredWave* pred_waves = nullptr;            // lane-packed launch: one entry per wave
    PredSrc* pred_srcs = nullptr;              // per subgrid: its residuals in the read-only upload
    uint32_t* pred_flags = nullptr;            // per wave: left the 32-bit range (redone by the 64-bit kernel)
    void* pred_sink = nullptr;                 // 4 KB nobody reads: where off-grid lanes store
    uint32_t n_pred_vec_waves = 0;             // the first so many waves take four-sample accesses
    // "late" subgrids: residuals of the first JXLGPU_PRED_LATE_STEPS forward Squeeze steps (the top levels: half / three
    // quarters of the samples), consumed by the LAST inverse steps — their predictor waves run on a side stream beside the
    // deep Squeeze levels (serial chains beside HBM-bound launches) and the first step that reads them waits for ev_late.
    // Wave order: [early vec][early rest][late vec][late rest]
    uint32_t n_pred_early = 0, n_pred_late_vec = 0;
    hipEvent_t ev_late = nullptr;
    ~ModularState() { if (ev_late) (void)hipEventDestroy(ev_late); }
    bool pred_narrow = false;
    std::vector<JxlGpuMaLeaf> unit_leaves;   // copy of JxlGpuModularDesc::unit_leaves (empty: the frame's one leaf)
    std::vector<JxlGpuMaLeaf> axis_leaves;   // copy of JxlGpuModularDesc::axis_leaves (per-row / per-column leaves of the units marked BY_ROW / BY_COLUMN)
    JxlGpuMaLeaf* d_axis_leaves = nullptr;   // ... on the device
    bool pred_d4 = false;         // every wave of the lane-packed pass has D = 4 (no subgrid wider than 256 columns): predict_lanes_wp4_kernel
    bool pred_big_ring = false;   // a subgrid wider than 512 columns: rows trail by D = 16, the lane kernels with the 64-column sample ring
    float* fpix[3] = {};
};

// Squeeze::set_default_params, transform.rs:285-341
void default_squeeze(const std::vector<Grid>& l, int nb_meta, std::vector<JxlGpuSqueezeStep>* sp) {
    uint32_t first = (uint32_t)nb_meta;
    uint32_t w = l[first].w, h = l[first].h;
    if (l.size() - first >= 3 && l[first + 1].w == w && l[first + 1].h == h) {
        sp->push_back({1, 0, first + 1, 2});
        sp->push_back({0, 0, first + 1, 2});
    }
    uint32_t num_c = (uint32_t)l.size() - first;
    if (h >= w && h > 8) { sp->push_back({0, 1, first, num_c}); h = (h + 1) / 2; }
    while (w > 8 || h > 8) {
        if (w > 8) { sp->push_back({1, 1, first, num_c}); w = (w + 1) / 2; }
        if (h > 8) { sp->push_back({0, 1, first, num_c}); h = (h + 1) / 2; }
    }
}

template <typename T>
int malloc_dev(jxlgpu_ctx* ctx, jxlgpu_frame* f, T** out, size_t bytes) {
    void* p = nullptr;
    HIP_TRY(ctx, ctx_dev_malloc(ctx, &p, std::max<size_t>(bytes, 16)));
    f->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return JXLGPU_OK;
}

int fail(jxlgpu_ctx* ctx, int code, const char* msg) {
    ctx->last_error = msg;
    return code;
}

// One squeeze step over `count` (<= 3) independent channels.  Long chains are cut into segments
// (one launch for all channels + one check launch); steps too small for that are appended to
// `chain` (flushed by the caller as ONE launch for all the small levels) when they are tiny, or run
// through the whole-line kernels.
struct SqueezePlan {
    ChainArgs chain;
    bool chain_used = false;
    uint32_t epoch = 1;   // counts the flushes: a rectangle written in an earlier epoch is in memory for everything queued now
};

template <typename S>
void flush_chain(hipStream_t s, SqueezePlan& plan) {
    if (!plan.chain_used) return;
    squeeze_chain_kernel<S><<<3, 1024, 0, s>>>(plan.chain);
    memset(&plan.chain, 0, sizeof(plan.chain));
    plan.chain_used = false;
    ++plan.epoch;
}

template <typename S>
void launch_squeeze_step(hipStream_t s, const Tuning& tune, bool horizontal, const SqzArgs* a, const int* chain_slot,
                         int count, void* chk, size_t chk_bytes, int* redo_count, SqueezePlan& plan) {
    auto al = [](const void* p, uint32_t stride) { return ((uintptr_t)p % 16 == 0) && ((stride * sizeof(S)) % 16 == 0); };
    SegArgs3 g3;
    memset(&g3, 0, sizeof(g3));
    int nseg_ch = 0;
    uint32_t max_lines = 0, max_nseg = 0, max_grid_y = 0;
    bool all_vec = true, all_lanes = true;
    const size_t chk_each = chk_bytes / 3;
    for (int k = 0; k < count; ++k) {
        const SqzArgs& ak = a[k];
        const bool vec = al(ak.avg, ak.avg_stride) && al(ak.res, ak.res_stride) && al(ak.out, ak.out_stride);
        const uint32_t len = horizontal ? ak.width : ak.height, lines = horizontal ? ak.height : ak.width;
        const uint32_t pairs = len / 2;
        // segment length: about an eighth of the chain, a multiple of 16, within [32, JXLGPU_SQZ_SEG]
        uint32_t L = (pairs / 8 + 15) / 16 * 16;
        L = std::min<uint32_t>(std::max<uint32_t>(L, 32u), std::max<uint32_t>(32u, (uint32_t)tune.sqz_seg / 16u * 16u));
        uint32_t nseg = pairs / L;
        const uint32_t avg_len = (len + 1) / 2;
        // horizontal, aligned: lane = piece of 2N pairs (squeeze_h_lanes_kernel); a check segment is a wave's 64 pieces
        constexpr uint32_t SP = 2 * (16 / sizeof(S));
        const bool lanes = horizontal && vec && !tune.sqz_h_rows && pairs >= 64;
        uint32_t grid_y = ceil_div(lines, 64);
        if (lanes) {
            const uint32_t nls = ceil_div(pairs, SP);
            L = 64 * SP;
            nseg = nls > 32 ? ceil_div(nls, 64u) : 1u;
            uint32_t lpr = 64;
            if (nls <= 32) { lpr = 1; while (lpr < nls) lpr <<= 1; }
            grid_y = ceil_div(lines, 64 / lpr);
        }
        const bool segmented = (lanes || (nseg >= 2 && (!horizontal || (vec && avg_len >= 2u * (16 / sizeof(S)))))) && chk &&
                               (size_t)nseg * 2 * lines * sizeof(S) <= chk_each;
        if (segmented) {
            SegArgs& g = g3.g[nseg_ch];
            g.a = ak; g.seg_pairs = L; g.nseg = nseg; g.runin = tune.sqz_runin;
            g.chk = (char*)chk + (size_t)nseg_ch * chk_each;
            ++nseg_ch;
            max_lines = std::max(max_lines, lines);
            max_nseg = std::max(max_nseg, nseg);
            max_grid_y = std::max(max_grid_y, grid_y);
            all_vec &= vec;
            all_lanes &= lanes;
        } else if ((uint64_t)pairs * lines <= (1u << 16) && plan.chain.n[chain_slot[k]] < (uint32_t)kChainMaxSteps) {
            const int c = chain_slot[k];
            const uint32_t i = plan.chain.n[c]++;
            plan.chain.a[c][i] = ak;
            plan.chain.horizontal[c][i] = horizontal ? 1u : 0u;
            plan.chain_used = true;
        } else {
            flush_chain<S>(s, plan);
            if (horizontal) {
                if (vec) squeeze_h_kernel<S, true><<<ceil_div(ak.height, 64), 64, 0, s>>>(ak);
                else squeeze_h_kernel<S, false><<<ceil_div(ak.height, 64), 64, 0, s>>>(ak);
            } else {
                squeeze_v_kernel<S><<<ceil_div(ak.width, 64), 64, 0, s>>>(ak);
            }
        }
    }
    if (!nseg_ch) return;
    flush_chain<S>(s, plan);  // the segmented step reads what the small levels produced
    if (horizontal) {
        // (the channels of a step are all aligned or all not: they are rectangles of one geometry)
        if (all_lanes) squeeze_h_lanes_kernel<S><<<dim3(max_nseg, max_grid_y, nseg_ch), 64, 0, s>>>(g3);
        else squeeze_seg3_kernel<S, true><<<dim3(ceil_div(max_lines, 64), max_nseg, nseg_ch), 64, 0, s>>>(g3);
        if (max_nseg > 1) squeeze_check3_kernel<S, true><<<dim3(ceil_div(max_lines, 64), nseg_ch), 64, 0, s>>>(g3, redo_count);
    } else {
        constexpr uint32_t NC = 4 / sizeof(S);  // i16: two columns per lane; i32: the one-column kernel
        if (all_vec && NC > 1) {
            // lanes: width / NC vectors of columns + the width % NC leftover columns
            const uint32_t lanes = max_lines / NC + NC;
            squeeze_v_seg3_kernel<S, (NC > 1 ? NC : 2)><<<dim3(ceil_div(lanes, 64), max_nseg, nseg_ch), 64, 0, s>>>(g3);
        } else {
            squeeze_seg3_kernel<S, false><<<dim3(ceil_div(max_lines, 64), max_nseg, nseg_ch), 64, 0, s>>>(g3);
        }
        squeeze_check3_kernel<S, false><<<dim3(ceil_div(max_lines, 64), nseg_ch), 64, 0, s>>>(g3, redo_count);
    }
}

// Runs predictor application + all inverse transforms on the working copies.
int run_inverse(jxlgpu_ctx* ctx, jxlgpu_frame* f) {
    ModularState* m = static_cast<ModularState*>(f->modular);
    hipStream_t s = ctx->stream;
    const bool i16 = m->desc.sample_type == JXLGPU_SAMPLE_I16;
    const size_t esz = m->esz;
    const uint32_t nch = (uint32_t)m->orig.size();
    // The uploaded buffers are location 3, read-only: a squeeze step reads its rectangles where they
    // are and writes the merged one into a working copy, so nothing has to be copied up front.  Only
    // the in-place passes (predictor, RCT, palette) need a writable copy first.
    m->work[3] = m->orig;
    const bool predict = m->desc.residual_predictor <= 13 || !m->unit_leaves.empty();
    bool late_pending = false;   // predictor waves of the top-level residuals still running on the side stream (ModularState::ev_late)

    // forward bookkeeping (transform_channel_info): which rectangle is which transformed channel
    std::vector<Grid> l;
    int nb_meta = 0;
    for (uint32_t c = 0; c < nch; ++c) {
        Grid g{(int)c, 0, 0, m->cw[c], m->ch[c], predict ? 0 : 3, {}};
        g.orig_w = m->cw[c]; g.orig_h = m->ch[c];
        l.push_back(g);
    }
    int meta_next = 0;
    m->steps.assign(m->transforms.size(), {});
    for (size_t t = 0; t < m->transforms.size(); ++t) {
        const JxlGpuTransform& tr = m->transforms[t];
        if (tr.kind == JXLGPU_TR_RCT) {
            if (tr.begin_c + 3 > l.size()) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "RCT channel range");
            const Grid& a = l[tr.begin_c];
            for (int k = 1; k < 3; ++k)
                if (l[tr.begin_c + k].w != a.w || l[tr.begin_c + k].h != a.h) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "RCT size mismatch");
        } else if (tr.kind == JXLGPU_TR_PALETTE) {
            const uint32_t begin = tr.begin_c, end = tr.begin_c + tr.num_c;
            if (end > l.size() || tr.num_c == 0 || tr.num_c > 8 || meta_next >= (int)m->meta.size())
                return fail(ctx, JXLGPU_ERR_INVALID_ARG, "palette parameters");
            if (begin < (uint32_t)nb_meta) nb_meta = nb_meta + 2 - (int)tr.num_c; else nb_meta += 1;
            for (uint32_t i = begin + 1; i < end; ++i) {
                if (l[begin + 1].buf < 0 || l[begin + 1].w != l[begin].w || l[begin + 1].h != l[begin].h)
                    return fail(ctx, JXLGPU_ERR_INVALID_ARG, "palette member mismatch");
                l[begin].members.push_back(l[begin + 1].buf);
                l.erase(l.begin() + begin + 1);
            }
            Grid pal{~meta_next, 0, 0, tr.nb_colours, tr.num_c, 0, {}};
            pal.hshift = pal.vshift = -1;  // ModularChannelInfo::new_unshiftable, transform.rs:236
            pal.orig_w = tr.nb_colours; pal.orig_h = tr.num_c;
            l.insert(l.begin(), pal);
            ++meta_next;
        } else if (tr.kind == JXLGPU_TR_SQUEEZE) {
            std::vector<JxlGpuSqueezeStep>& sp = m->steps[t];
            if (!m->explicit_steps[t].empty()) sp = m->explicit_steps[t];
            else default_squeeze(l, nb_meta, &sp);
            for (const JxlGpuSqueezeStep& st : sp) {
                const uint32_t begin = st.begin_c, end = st.begin_c + st.num_c;
                if (end > l.size()) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "squeeze channel range");
                if (begin < (uint32_t)nb_meta) {
                    if (!st.in_place || end > (uint32_t)nb_meta) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "squeeze on meta channels");
                    nb_meta += (int)st.num_c;
                }
                std::vector<Grid> res;
                for (uint32_t i = begin; i < end; ++i) {
                    Grid& g = l[i];
                    if (g.w == 0 || g.h == 0 || g.buf < 0) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "cannot squeeze this channel");
                    Grid r = g;
                    r.fwd_step = (int)(&st - sp.data());
                    // a rectangle that is squeezed AGAIN (explicit steps over earlier residuals, a second Squeeze transform)
                    // is no longer "the residual of step k": its inverse reads the average half as soon as the later step is
                    // undone, long before step k is — its predictor waves must not run late (ev_late is only waited for by
                    // the step that reads a late rectangle as its residual input, or as its average: below)
                    g.fwd_step = -1;
                    if (g.hshift > 30 || g.vshift > 30) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "channel squeezed too much");
                    if (st.horizontal) {
                        uint32_t len = g.w; g.w = (len + 1) / 2; r.w = len / 2; r.x0 = g.x0 + g.w;
                        if (g.hshift >= 0) { ++g.hshift; ++r.hshift; }  // transform.rs:398-401
                    } else {
                        uint32_t len = g.h; g.h = (len + 1) / 2; r.h = len / 2; r.y0 = g.y0 + g.h;
                        if (g.vshift >= 0) { ++g.vshift; ++r.vshift; }
                    }
                    res.push_back(r);
                }
                l.insert(st.in_place ? l.begin() + end : l.end(), res.begin(), res.end());
            }
        } else {
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "unknown transform kind");
        }
    }

    auto ptr = [&](const Grid& g, int loc, uint32_t* stride) -> char* {
        if (g.buf >= 0) {
            *stride = m->cw[g.buf];
            return (char*)m->work[loc][g.buf] + ((size_t)g.y0 * *stride + g.x0) * esz;
        }
        *stride = m->mw[~g.buf];
        return (char*)(predict ? m->meta_work[~g.buf] : m->meta[~g.buf]);
    };

    // ---- M4: residuals -> samples, one (group, channel) subgrid at a time, BEFORE the inverse transforms.
    // The reference decodes every transformed channel on its own tile grid (prepare_groups,
    // jxl-modular/src/image.rs:209-340): the leading meta channels and the leading channels that fit one
    // group are whole-channel streams of GlobalModular (:224-228 skip_while); of the rest, channels with
    // hshift < 3 or vshift < 3 are cut into (group_dim >> hshift) x (group_dim >> vshift) pass-group tiles,
    // the others into LF-group tiles of (8 group_dim >> shift); the tile COUNT comes from the untransformed
    // size (:279-284, :300-305).  Every tile starts a fresh PredictorState (decode_single_node, :716-777).
    if (predict) {
        const uint32_t gd = m->desc.group_dim ? m->desc.group_dim : 256;
        // (decode_single_node's dispatch, image.rs:733-777, sends Gradient with offset 0 / multiplier 1 to
        //  decode_simple_grad: the same arithmetic as decode_one with Predictor::Gradient, one kernel here)
        bool global_phase = true;
        std::vector<PredTile> tiles;
        std::unordered_map<const void*, const void*> tile_src;   // PredTile.base -> the subgrid in the read-only upload
        std::unordered_map<const void*, int> tile_late;          // PredTile.base -> residual of one of the first `late_steps` Squeeze steps
        const int late_steps = ctx->tune.pred_late_steps;
        uint32_t max_w = 1;
        // per-unit leaves: units in the order of JxlGpuModularDesc::unit_leaves — channels in list order, ncols x nrows subgrids
        // each (the ones outside the channel included), one for a channel decoded whole
        size_t unit_base = 0;
        bool any_wp = false, all_wp = true;
        const JxlGpuMaLeaf one_leaf{m->desc.residual_predictor, m->desc.residual_multiplier, m->desc.residual_offset};
        for (size_t i = 0; i < l.size() && !m->pred_tiles; ++i) {
            const Grid& g = l[i];
            if (g.w == 0 || g.h == 0) continue;
            uint32_t tw, th, ncols, nrows;
            if (global_phase && ((int)i < nb_meta || (g.w <= gd && g.h <= gd))) {
                tw = g.w; th = g.h; ncols = nrows = 1;
            } else {
                global_phase = false;
                if (g.hshift < 0 || g.vshift < 0) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "unshiftable channel among the grouped channels");
                if (g.hshift < 3 || g.vshift < 3) {
                    tw = gd >> g.hshift; th = gd >> g.vshift;
                    ncols = ceil_div(g.orig_w, gd); nrows = ceil_div(g.orig_h, gd);
                } else {
                    tw = gd >> (g.hshift - 3); th = gd >> (g.vshift - 3);
                    ncols = ceil_div(g.orig_w, gd * 8); nrows = ceil_div(g.orig_h, gd * 8);
                }
                if (g.hshift > 31 || g.vshift > 31 || tw == 0 || th == 0)
                    return fail(ctx, JXLGPU_ERR_INVALID_ARG, "channel shift too large after transform");  // image.rs:265-273
            }
            // subgrids up to kPredLaneMaxW columns take the lane-packed kernel (any height); wider ones the
            // workgroup-per-subgrid kernel: a lane per row, at most 256 rows
            const bool lanes_ok = !ctx->tune.pred_wg && tw <= kPredLaneMaxW;
            if (!m->unit_leaves.empty() && unit_base + (size_t)ncols * nrows > m->unit_leaves.size())
                return fail(ctx, JXLGPU_ERR_INVALID_ARG, "num_unit_leaves is smaller than the number of decode units");
            const JxlGpuMaLeaf* leaves = m->unit_leaves.empty() ? nullptr : m->unit_leaves.data() + unit_base;
            unit_base += (size_t)ncols * nrows;
            bool chan_wp = !leaves && one_leaf.predictor == 6;
            for (size_t u = 0; leaves && u < (size_t)ncols * nrows; ++u) chan_wp |= leaves[u].predictor == 6 || leaves[u].predictor >= JXLGPU_LEAF_BY_ROW;
            if (!lanes_ok && (th > 256 || (chan_wp && tw > kPredMaxTileW)))
                return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "predictor tile wider than 1024 columns with more than 256 rows (or wider than 1024 with the self-correcting predictor)");
            uint32_t stride = 0;
            char* base = ptr(g, 0, &stride);
            // the same rectangle in the read-only upload (what the narrow kernel reads)
            const char* src_base = g.buf >= 0 ? (const char*)m->orig[g.buf] + ((size_t)g.y0 * stride + g.x0) * esz
                                              : (const char*)m->meta[~g.buf];
            // into_groups_with_fixed_count (jxl-grid/src/mutable_subgrid.rs:480-515): subgrids past the channel are empty
            for (uint32_t gy = 0; gy < nrows; ++gy)
                for (uint32_t gx = 0; gx < ncols; ++gx) {
                    const uint32_t x0 = std::min(gx * tw, g.w), y0 = std::min(gy * th, g.h);
                    const uint32_t gw = std::min(tw, g.w - x0), gh = std::min(th, g.h - y0);
                    if (gw == 0 || gh == 0) continue;
                    const JxlGpuMaLeaf& lf = leaves ? leaves[(size_t)gy * ncols + gx] : one_leaf;
                    uint32_t tile_sc = 0;
                    if (lf.predictor >= JXLGPU_LEAF_BY_ROW) {
                        // a tree that splits on y / x inside the unit: its per-row / per-column leaves
                        const size_t need = lf.predictor == JXLGPU_LEAF_BY_ROW ? gh : gw;
                        if (lf.multiplier < 0 || lf.offset != 0 || (size_t)lf.multiplier + need > m->axis_leaves.size())
                            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "a per-row / per-column unit leaf points outside axis_leaves");
                        for (size_t q = 0; q < need; ++q) tile_sc |= m->axis_leaves[(size_t)lf.multiplier + q].predictor == 6 ? 1u : 0u;
                        any_wp |= tile_sc != 0; all_wp = false;
                    } else {
                        any_wp |= lf.predictor == 6; all_wp &= lf.predictor == 6;
                    }
                    tiles.push_back(PredTile{base + ((size_t)y0 * stride + x0) * esz, stride, gw, gh, lanes_ok ? 1u : 0u,
                                             lf.predictor, lf.multiplier, lf.offset, tile_sc});
                    tile_src[tiles.back().base] = src_base + ((size_t)y0 * stride + x0) * esz;
                    tile_late[tiles.back().base] = (lanes_ok && g.fwd_step >= 0 && g.fwd_step < late_steps) ? 1 : 0;
                    if (!lanes_ok) max_w = std::max(max_w, gw);
                }
        }
        if (!m->pred_tiles && !m->unit_leaves.empty() && unit_base != m->unit_leaves.size())
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "num_unit_leaves is not the number of decode units");
        if (!m->pred_tiles) {
            auto pow2ceil = [](uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; };
            auto log2u = [](uint32_t v) { uint32_t l = 0; while ((1u << l) < v) ++l; return l; };
            // lanes per subgrid: P = pow2ceil(gw) / 4 (1..64); columns per round DP = max(4 P, pow2ceil(gw))
            auto lanes_of = [&](const PredTile& t) { return std::min(64u, std::max(1u, pow2ceil(t.gw) / 4)); };
            auto dp_of = [&](const PredTile& t) { return std::max(4 * lanes_of(t), pow2ceil(t.gw)); };
            auto steps_of = [&](const PredTile& t) { return t.packed ? t.gw + (dp_of(t) / lanes_of(t)) * (t.gh - 1) : t.gw + 3 * (t.gh - 1); };
            // four-sample accesses: both copies of the subgrid aligned to four samples, rows too, width a multiple of four
            auto vec_of = [&](const PredTile& t) {
                const uintptr_t al = 4 * esz;
                return t.predictor < JXLGPU_LEAF_BY_ROW && (uintptr_t)t.base % al == 0 && (uintptr_t)tile_src[t.base] % al == 0 && t.stride % 4 == 0 && t.gw % 4 == 0 &&
                       dp_of(t) >= 16;   // (a lane's 16-column blocks end inside a round: predict_lanes_wp4_kernel)
            };
            // wide subgrids first (own launch), then the lane-packed ones by (vec, P, DP), longest chains first inside a class
            std::stable_sort(tiles.begin(), tiles.end(), [&](const PredTile& x, const PredTile& y) {
                if (x.packed != y.packed) return x.packed < y.packed;
                if (x.packed) {
                    if (tile_late[x.base] != tile_late[y.base]) return tile_late[x.base] < tile_late[y.base];
                    if (vec_of(x) != vec_of(y)) return vec_of(x) > vec_of(y);
                    if (lanes_of(x) != lanes_of(y)) return lanes_of(x) > lanes_of(y);
                    if (dp_of(x) != dp_of(y)) return dp_of(x) > dp_of(y);
                    if (x.predictor != y.predictor) return x.predictor < y.predictor;   // a wave's subgrids share their predictor
                }
                return steps_of(x) > steps_of(y);
            });
            uint32_t n_wide = 0;
            while (n_wide < tiles.size() && !tiles[n_wide].packed) ++n_wide;
            std::vector<PredWave> waves;
            uint32_t lane_err_w = 256;
            bool all_d4 = true;
            for (uint32_t i = n_wide; i < tiles.size();) {
                const uint32_t P = lanes_of(tiles[i]), DP = dp_of(tiles[i]), T = 64 / P;
                PredWave w{};
                w.first = i; w.log2p = log2u(P); w.log2dp = log2u(DP); w.vec = vec_of(tiles[i]) ? 1u : 0u;
                w.pad[0] = (uint32_t)tile_late[tiles[i].base];
                const uint32_t wave_pred = tiles[i].predictor;
                while (i < tiles.size() && w.count < T && lanes_of(tiles[i]) == P && dp_of(tiles[i]) == DP && tiles[i].predictor == wave_pred &&
                       (vec_of(tiles[i]) ? 1u : 0u) == w.vec && (uint32_t)tile_late[tiles[i].base] == w.pad[0]) {
                    w.steps = std::max(w.steps, steps_of(tiles[i]));
                    ++w.count; ++i;
                }
                all_d4 &= DP == 4 * P;
                if (DP > 256) lane_err_w = std::max(lane_err_w, 512u);
                if (DP > 512) { lane_err_w = 1024; m->pred_big_ring = true; }
                waves.push_back(w);
            }
            // the four-sample waves first (a launch of their own), longest waves first inside each part
            std::stable_sort(waves.begin(), waves.end(), [](const PredWave& x, const PredWave& y) {
                if (x.pad[0] != y.pad[0]) return x.pad[0] < y.pad[0];
                if (x.vec != y.vec) return x.vec > y.vec;
                return x.steps > y.steps;
            });
            {   // issue priority by chain length (predict_lanes_narrow_kernel): the longest quarter 3, then 2, 1, 0
                uint32_t max_steps = 1;
                for (const PredWave& w : waves) max_steps = std::max(max_steps, w.steps);
                const bool on = ctx->tune.pred_prio;
                for (PredWave& w : waves)
                    w.pad[1] = !on ? 0u : (w.steps * 4 >= max_steps * 3 ? 3u : (w.steps * 2 >= max_steps ? 2u : (w.steps * 4 >= max_steps ? 1u : 0u)));
                // bit 8: every subgrid of the wave is a multiple of 16 columns wide (predict_lanes_wp4_kernel's 16-sample stores)
                for (PredWave& w : waves) {
                    bool all16 = true;
                    for (uint32_t q = 0; q < w.count; ++q) all16 &= tiles[w.first + q].gw % 16 == 0;
                    w.pad[1] |= all16 ? 0x100u : 0u;
                }
            }
            m->n_pred_vec_waves = m->n_pred_early = m->n_pred_late_vec = 0;
            for (const PredWave& w : waves) {
                if (!w.pad[0]) { ++m->n_pred_early; m->n_pred_vec_waves += w.vec; }
                else m->n_pred_late_vec += w.vec;
            }
            m->n_pred_tiles = (uint32_t)tiles.size();
            m->n_pred_wide = n_wide;
            m->n_pred_waves = (uint32_t)waves.size();
            if (getenv("JXLGPU_PRED_DUMP")) {   // diagnostics: the waves of the lane-packed pass in launch order
                for (size_t i = 0; i < waves.size(); ++i)
                    fprintf(stderr, "predwave %zu late=%u vec=%u P=%u DP=%u count=%u steps=%u gw=%u gh=%u\n", i, waves[i].pad[0], waves[i].vec,
                            1u << waves[i].log2p, 1u << waves[i].log2dp, waves[i].count, waves[i].steps, tiles[waves[i].first].gw, tiles[waves[i].first].gh);
            }
            m->pred_err_w = any_wp ? max_w : 1;
            m->pred_lane_err_w = any_wp ? lane_err_w : 64;
            // the 32-bit form of the self-correcting predictor: WpHeader fields in their coded ranges (5 / 4 bits)
            bool wp_coded = true;
            for (int k = 0; k < 11; ++k) wp_coded &= m->desc.wp_params[k] >= 0 && m->desc.wp_params[k] < (k < 7 ? 32 : 16);
            // (the 32-bit kernel is the self-correcting predictor only: every unit has to use it)
            m->pred_narrow = any_wp && all_wp && wp_coded && !ctx->tune.pred_wide && !waves.empty();
            m->pred_d4 = all_d4 && !ctx->tune.pred_step_v1;
            std::vector<PredSrc> srcs(tiles.size());
            for (size_t i = 0; i < tiles.size(); ++i) srcs[i].src = tile_src[tiles[i].base];
            if (int rc = malloc_dev(ctx, f, &m->pred_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(PredTile))) return rc;
            if (int rc = malloc_dev(ctx, f, &m->pred_waves, std::max<size_t>(waves.size(), 1) * sizeof(PredWave))) return rc;
            if (int rc = malloc_dev(ctx, f, &m->pred_srcs, std::max<size_t>(srcs.size(), 1) * sizeof(PredSrc))) return rc;
            if (int rc = malloc_dev(ctx, f, &m->pred_flags, std::max<size_t>(waves.size(), 1) * sizeof(uint32_t))) return rc;
            if (int rc = malloc_dev(ctx, f, &m->pred_sink, 4096)) return rc;   // 64 lanes x (4 + 12) samples of up to 4 bytes
            if (!m->axis_leaves.empty()) {
                if (int rc = malloc_dev(ctx, f, &m->d_axis_leaves, m->axis_leaves.size() * sizeof(JxlGpuMaLeaf))) return rc;
                HIP_TRY(ctx, hipMemcpy(m->d_axis_leaves, m->axis_leaves.data(), m->axis_leaves.size() * sizeof(JxlGpuMaLeaf), hipMemcpyHostToDevice));
            }
            // blocking copies from the host vectors: the lists are built once per frame (the geometry never changes)
            if (!tiles.empty()) {
                HIP_TRY(ctx, hipMemcpy(m->pred_tiles, tiles.data(), tiles.size() * sizeof(PredTile), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(m->pred_srcs, srcs.data(), srcs.size() * sizeof(PredSrc), hipMemcpyHostToDevice));
            }
            if (!waves.empty())
                HIP_TRY(ctx, hipMemcpy(m->pred_waves, waves.data(), waves.size() * sizeof(PredWave), hipMemcpyHostToDevice));
        }
        // The in-place kernels work on a copy of the residuals; the narrow kernel reads the upload itself and writes the
        // working copy, so when it serves every subgrid of the frame nothing is copied (200 MB for an 8K frame).
        if (!(m->pred_narrow && m->n_pred_wide == 0)) {
            for (uint32_t c = 0; c < nch; ++c)
                HIP_TRY(ctx, hipMemcpyAsync(m->work[0][c], m->orig[c], (size_t)m->cw[c] * m->ch[c] * esz, hipMemcpyDeviceToDevice, s));
            for (size_t c = 0; c < m->meta.size(); ++c)
                HIP_TRY(ctx, hipMemcpyAsync(m->meta_work[c], m->meta[c], (size_t)m->mw[c] * m->mh[c] * esz, hipMemcpyDeviceToDevice, s));
        }
        if (m->n_pred_tiles) {
            PredArgs pa;
            pa.tiles = m->pred_tiles; pa.sink = m->pred_sink; pa.axis = m->d_axis_leaves;
            for (int k = 0; k < 11; ++k) pa.wp[k] = m->desc.wp_params[k];
            // a frame with per-row / per-column leaves: the MAPPED instantiations (they serve plain units as well)
            const bool mapped = m->d_axis_leaves != nullptr;
            if (m->n_pred_wide) {
                pa.err_w = m->pred_err_w;
                const size_t lds = (size_t)5 * m->pred_err_w * 4;
                if (mapped) {
                    if (i16) predict_tiles_kernel<int16_t, true><<<m->n_pred_wide, 256, lds, s>>>(pa);
                    else predict_tiles_kernel<int32_t, true><<<m->n_pred_wide, 256, lds, s>>>(pa);
                } else if (i16) predict_tiles_kernel<int16_t><<<m->n_pred_wide, 256, lds, s>>>(pa);
                else predict_tiles_kernel<int32_t><<<m->n_pred_wide, 256, lds, s>>>(pa);
            }
            if (m->n_pred_waves) {
                pa.err_w = m->pred_lane_err_w;
                const size_t lds = (size_t)5 * m->pred_lane_err_w * 4;
                auto lanes_launch = [&](hipStream_t st, uint32_t first, uint32_t count, bool vec, const PredSrc* srcs, const uint32_t* flags) {
                    auto go = [&](auto kern) { kern<<<count, 64, lds, st>>>(pa, m->pred_waves + first, srcs, flags); };
                    const int sel = (i16 ? 4 : 0) | (vec ? 2 : 0) | (m->pred_big_ring ? 1 : 0);
                    if (mapped && !vec) {   // (the four-sample waves never hold a mapped unit: vec_of)
                        switch (sel) {
                            case 0: go(predict_lanes_kernel<int32_t, false, kRing, true>); break;
                            case 1: go(predict_lanes_kernel<int32_t, false, kBigRing, true>); break;
                            case 4: go(predict_lanes_kernel<int16_t, false, kRing, true>); break;
                            default: go(predict_lanes_kernel<int16_t, false, kBigRing, true>); break;
                        }
                        return;
                    }
                    switch (sel) {
                        case 0: go(predict_lanes_kernel<int32_t, false, kRing>); break;
                        case 1: go(predict_lanes_kernel<int32_t, false, kBigRing>); break;
                        case 2: go(predict_lanes_kernel<int32_t, true, kRing>); break;
                        case 3: go(predict_lanes_kernel<int32_t, true, kBigRing>); break;
                        case 4: go(predict_lanes_kernel<int16_t, false, kRing>); break;
                        case 5: go(predict_lanes_kernel<int16_t, false, kBigRing>); break;
                        case 6: go(predict_lanes_kernel<int16_t, true, kRing>); break;
                        default: go(predict_lanes_kernel<int16_t, true, kBigRing>); break;
                    }
                };
                if (m->pred_narrow) {
                    // 32-bit pass over everything, then the 64-bit kernel for the waves that left the 32-bit range (none, for
                    // images of up to 16 bits).  Wave ranges: [0, nv) early four-sample, [nv, ne) early rest, [ne, ne + lv) late
                    // four-sample, [ne + lv, n) late rest.
                    const uint32_t ne = m->n_pred_early, nv = m->n_pred_vec_waves, ns = ne - nv;
                    const uint32_t lv = m->n_pred_late_vec, ls = m->n_pred_waves - ne - lv;
                    // kernel of a launch by (sample type, four-sample accesses, sample ring): 16 ring columns unless a subgrid is wider than 512
                    auto narrow = [&](hipStream_t st, uint32_t first, uint32_t count, bool vec) {
                        if (!count) return;
                        auto go = [&](auto kern) { kern<<<count, 64, lds, st>>>(pa, m->pred_waves + first, m->pred_srcs, m->pred_flags + first); };
                        if (m->pred_d4) {   // static LDS only
                            auto go4 = [&](auto kern) { kern<<<count, 64, 0, st>>>(pa, m->pred_waves + first, m->pred_srcs, m->pred_flags + first); };
                            if (i16) { if (vec) go4(predict_lanes_wp4_kernel<int16_t, true>); else go4(predict_lanes_wp4_kernel<int16_t, false>); }
                            else { if (vec) go4(predict_lanes_wp4_kernel<int32_t, true>); else go4(predict_lanes_wp4_kernel<int32_t, false>); }
                            return;
                        }
                        const int sel = (i16 ? 4 : 0) | (vec ? 2 : 0) | (m->pred_big_ring ? 1 : 0);
                        switch (sel) {
                            case 0: go(predict_lanes_narrow_kernel<int32_t, false, kRing>); break;
                            case 1: go(predict_lanes_narrow_kernel<int32_t, false, kBigRing>); break;
                            case 2: go(predict_lanes_narrow_kernel<int32_t, true, kRing>); break;
                            case 3: go(predict_lanes_narrow_kernel<int32_t, true, kBigRing>); break;
                            case 4: go(predict_lanes_narrow_kernel<int16_t, false, kRing>); break;
                            case 5: go(predict_lanes_narrow_kernel<int16_t, false, kBigRing>); break;
                            case 6: go(predict_lanes_narrow_kernel<int16_t, true, kRing>); break;
                            default: go(predict_lanes_narrow_kernel<int16_t, true, kBigRing>); break;
                        }
                    };
                    auto redo = [&](hipStream_t st, uint32_t first, uint32_t count, bool vec) {
                        if (!count) return;
                        lanes_launch(st, first, count, vec, m->pred_srcs, m->pred_flags + first);
                    };
                    // the few waves of misaligned subgrids (deep Squeeze levels: long chains, three waves of an 8K frame)
                    // run beside the others on the side stream instead of behind them
                    const bool side = nv && ns;
                    const bool late = (lv || ls) && ctx->stream_tr2;
                    hipStream_t s2 = side ? ctx->stream2 : s;
                    if (side || late) HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
                    if (late) {
                        // the top levels' residuals: on their own stream, beside everything up to the Squeeze step that reads them
                        hipStream_t sl = ctx->stream_tr2;
                        if (!m->ev_late) HIP_TRY(ctx, hipEventCreateWithFlags(&m->ev_late, hipEventDisableTiming));
                        HIP_TRY(ctx, hipStreamWaitEvent(sl, ctx->ev_fork, 0));
                        narrow(sl, ne, lv, true);
                        narrow(sl, ne + lv, ls, false);
                        redo(sl, ne, lv, true);
                        redo(sl, ne + lv, ls, false);
                        HIP_TRY(ctx, hipEventRecord(m->ev_late, sl));
                        late_pending = true;
                    } else {
                        narrow(s, ne, lv, true);
                        narrow(s, ne + lv, ls, false);
                    }
                    if (side) HIP_TRY(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
                    narrow(s2, nv, ns, false);
                    narrow(s, 0, nv, true);
                    if (side) {
                        HIP_TRY(ctx, hipEventRecord(ctx->ev_join, s2));
                        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                    }
                    redo(s, 0, nv, true);
                    redo(s, nv, ns, false);
                    if (!late) {
                        redo(s, ne, lv, true);
                        redo(s, ne + lv, ls, false);
                    }
                    if (ctx->tune.debug_sync) {
                        std::vector<uint32_t> fl(m->n_pred_waves);
                        HIP_TRY(ctx, hipStreamSynchronize(s));
                        if (ctx->stream_tr2) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_tr2));
                        HIP_TRY(ctx, hipMemcpy(fl.data(), m->pred_flags, fl.size() * 4, hipMemcpyDeviceToHost));
                        size_t nfl = 0;
                        for (uint32_t v : fl) nfl += v != 0;
                        fprintf(stderr, "predictor waves redone in 64-bit arithmetic: %zu of %u\n", nfl, m->n_pred_waves);
                    }
                } else {
                    // any other predictor (or JXLGPU_PRED_WIDE): in place on the working copy, the four wave ranges one after the other
                    const uint32_t ne = m->n_pred_early, nv = m->n_pred_vec_waves, lv = m->n_pred_late_vec;
                    const uint32_t first[4] = {0, nv, ne, ne + lv}, count[4] = {nv, ne - nv, lv, m->n_pred_waves - ne - lv};
                    for (int r = 0; r < 4; ++r)
                        if (count[r]) lanes_launch(s, first[r], count[r], (r & 1) == 0, nullptr, nullptr);
                }
            }
        }
    }

    // in-place passes: move a rectangle that still lives in the read-only upload into working copy 0
    auto ensure_writable = [&](Grid& g) -> hipError_t {
        if (g.buf < 0 || g.loc != 3) return hipSuccess;
        uint32_t stride = 0;
        char* src = ptr(g, 3, &stride);
        char* dst = ptr(g, 0, &stride);
        g.loc = 0;
        return hipMemcpy2DAsync(dst, (size_t)stride * esz, src, (size_t)stride * esz, (size_t)g.w * esz, g.h, hipMemcpyDeviceToDevice, s);
    };

    // inverse, last transform first (transform.rs:75-86)
    for (int t = (int)m->transforms.size() - 1; t >= 0; --t) {
        const JxlGpuTransform& tr = m->transforms[t];
        if (late_pending && tr.kind != JXLGPU_TR_SQUEEZE) { HIP_TRY(ctx, hipStreamWaitEvent(s, m->ev_late, 0)); late_pending = false; }
        if (tr.kind == JXLGPU_TR_SQUEEZE) {
            const std::vector<JxlGpuSqueezeStep>& sp = m->steps[t];
            SqueezePlan plan;
            memset(&plan.chain, 0, sizeof(plan.chain));
            for (int i = (int)sp.size() - 1; i >= 0; --i) {
                const JxlGpuSqueezeStep& st = sp[i];
                const int begin = (int)st.begin_c, count = (int)st.num_c, end = begin + count;
                const int from = st.in_place ? end : (int)l.size() - count;
                std::vector<Grid> res(l.begin() + from, l.begin() + from + count);
                l.erase(l.begin() + from, l.begin() + from + count);
                if (late_pending) {
                    bool reads_late = false;
                    for (const Grid& r : res) reads_late |= r.fwd_step >= 0 && r.fwd_step < ctx->tune.pred_late_steps;
                    for (int k = 0; k < count; ++k)   // the average halves as well (never late after the forward bookkeeping above)
                        reads_late |= l[begin + k].fwd_step >= 0 && l[begin + k].fwd_step < ctx->tune.pred_late_steps;
                    if (reads_late) {
                        // (the chain of small levels queued so far is flushed first: it does not depend on the late residuals)
                        if (i16) flush_chain<int16_t>(s, plan); else flush_chain<int32_t>(s, plan);
                        HIP_TRY(ctx, hipStreamWaitEvent(s, m->ev_late, 0));
                        late_pending = false;
                    }
                }
                // the channels of a step are independent: up to three per launch
                for (int k0 = 0; k0 < count; k0 += 3) {
                    const int nk = std::min(3, count - k0);
                    SqzArgs args[3];
                    int slot[3];
                    // The three workgroups of the small-level launch run side by side: a step may only join workgroup c if
                    // whatever wrote its inputs is in memory (an earlier epoch) or queued in c itself.  With default parameters
                    // a channel keeps its index and residuals are never outputs, so this never fires; explicit steps that
                    // squeeze residual channels again, or move a channel to another index, read across workgroups (found by
                    // tests/tools/fuzz_parity.py in round 6: 73 x 50, `appended_then_squeezed`).
                    bool crosses = false;
                    for (int k = 0; k < nk; ++k) {
                        const int c = (begin + k0 + k) % 3;
                        const Grid& g = l[begin + k0 + k];
                        const Grid& r = res[k0 + k];
                        crosses |= g.chain_epoch == plan.epoch && g.chain_slot != c;
                        crosses |= r.chain_epoch == plan.epoch && r.chain_slot != c;
                    }
                    if (crosses) { if (i16) flush_chain<int16_t>(s, plan); else flush_chain<int32_t>(s, plan); }
                    for (int k = 0; k < nk; ++k) {
                        Grid& g = l[begin + k0 + k];
                        const Grid& r = res[k0 + k];
                        int out_loc = 0;
                        while (out_loc == g.loc || out_loc == r.loc) ++out_loc;
                        SqzArgs& a = args[k];
                        a.avg = ptr(g, g.loc, &a.avg_stride);
                        a.res = ptr(r, r.loc, &a.res_stride);
                        if (st.horizontal) g.w += r.w; else g.h += r.h;
                        a.out = ptr(g, out_loc, &a.out_stride);
                        a.width = g.w; a.height = g.h;
                        g.loc = out_loc;
                        slot[k] = (begin + k0 + k) % 3;  // chain of the small levels this channel's steps join
                    }
                    if (i16) launch_squeeze_step<int16_t>(s, ctx->tune, st.horizontal, args, slot, nk, m->chk, m->chk_bytes, m->d_redo, plan);
                    else launch_squeeze_step<int32_t>(s, ctx->tune, st.horizontal, args, slot, nk, m->chk, m->chk_bytes, m->d_redo, plan);
                    for (int k = 0; k < nk; ++k) {   // (conservative: also when the step was launched on its own)
                        l[begin + k0 + k].chain_epoch = plan.epoch;
                        l[begin + k0 + k].chain_slot = slot[k];
                    }
                    if (ctx->tune.debug_sync) {
                        if (i16) flush_chain<int16_t>(s, plan); else flush_chain<int32_t>(s, plan);
                        hipError_t e = hipStreamSynchronize(s);
                        for (int k = 0; k < nk; ++k)
                            fprintf(stderr, "squeeze %s ch%d %ux%u avg=%p(%u) res=%p(%u) out=%p(%u) -> %s\n", st.horizontal ? "H" : "V",
                                    begin + k0 + k, args[k].width, args[k].height, args[k].avg, args[k].avg_stride, args[k].res,
                                    args[k].res_stride, args[k].out, args[k].out_stride, hipGetErrorString(e));
                    }
                }
            }
            if (i16) flush_chain<int16_t>(s, plan); else flush_chain<int32_t>(s, plan);
            if (late_pending) { HIP_TRY(ctx, hipStreamWaitEvent(s, m->ev_late, 0)); late_pending = false; }
        } else if (tr.kind == JXLGPU_TR_RCT) {
            RctArgs a;
            for (int k = 0; k < 3; ++k) HIP_TRY(ctx, ensure_writable(l[tr.begin_c + k]));
            for (int k = 0; k < 3; ++k) a.p[k] = ptr(l[tr.begin_c + k], l[tr.begin_c + k].loc, &a.stride[k]);
            a.width = l[tr.begin_c].w; a.height = l[tr.begin_c].h; a.rct_type = tr.rct_type;
            dim3 grid(ceil_div(a.width, 256), a.height);
            if (i16) rct_kernel<int16_t><<<grid, 256, 0, s>>>(a);
            else rct_kernel<int32_t><<<grid, 256, 0, s>>>(a);
        } else {
            Grid pal = l.front();
            l.erase(l.begin());
            Grid& leader = l[tr.begin_c];
            HIP_TRY(ctx, ensure_writable(leader));
            PalArgs a;
            memset(&a, 0, sizeof(a));
            a.palette = ptr(pal, 0, &a.pal_stride);
            a.idx = ptr(leader, leader.loc, &a.idx_stride);
            a.dst[0] = a.idx; a.dst_stride[0] = a.idx_stride;
            for (size_t k = 0; k < leader.members.size(); ++k) {
                Grid mg{leader.members[k], leader.x0, leader.y0, leader.w, leader.h, 0, {}};
                a.dst[k + 1] = ptr(mg, 0, &a.dst_stride[k + 1]);
            }
            a.width = leader.w; a.height = leader.h; a.num_c = tr.num_c; a.nb_colours = tr.nb_colours;
            a.nb_deltas = (int32_t)tr.nb_deltas; a.bit_depth = m->desc.bit_depth;
            if (tr.num_c > 8) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "palette over more than 8 channels");
            const size_t npx = (size_t)a.width * a.height;
            if (m->need_delta_bytes < npx) {
                if (int rc = malloc_dev(ctx, f, &m->need_delta, npx)) return rc;
                m->need_delta_bytes = npx;
            }
            a.need_delta = m->need_delta;
            a.n_delta = m->d_flag;
            HIP_TRY(ctx, hipMemsetAsync(m->d_flag, 0, sizeof(int), s));
            dim3 grid(ceil_div(a.width, 256), a.height);
            if (i16) palette_kernel<int16_t><<<grid, 256, 0, s>>>(a);
            else palette_kernel<int32_t><<<grid, 256, 0, s>>>(a);
            int n_delta = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&n_delta, m->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            if (n_delta > 0) {
                // palette.rs:112-142: the predictor pass over every channel of the palette
                if (tr.d_pred > 13) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "palette d_pred is not a Predictor");
                const uint32_t bands = ceil_div(a.height, 256);
                const size_t aux_words = (size_t)3 * (npx + 5 * (size_t)a.width + bands) + 4;
                if (m->delta_aux_words < aux_words) {
                    if (int rc = malloc_dev(ctx, f, &m->delta_aux, aux_words * 4)) return rc;
                    m->delta_aux_words = aux_words;
                }
                for (uint32_t c0 = 0; c0 < tr.num_c; c0 += 3) {
                    const uint32_t nc = std::min<uint32_t>(3, tr.num_c - c0);
                    DeltaArgs da;
                    memset(&da, 0, sizeof(da));
                    uint32_t* aux = m->delta_aux;
                    // layout: [fail + pad (4)] [progress 3*bands] [err rows 3*5*width] [rec 3*npx]
                    da.fail = reinterpret_cast<int*>(aux);
                    da.progress = aux + 4;
                    uint32_t* err = aux + 4 + 3 * bands;
                    int32_t* rec = reinterpret_cast<int32_t*>(err + (size_t)15 * a.width);
                    HIP_TRY(ctx, hipMemsetAsync(aux, 0, (4 + (size_t)3 * bands + (size_t)15 * a.width) * 4, s));
                    for (uint32_t k = 0; k < nc; ++k) {
                        da.grid[k] = a.dst[c0 + k];
                        da.stride[k] = a.dst_stride[c0 + k];
                        da.rec[k] = rec + (size_t)k * npx;
                        da.true_err[k] = reinterpret_cast<int32_t*>(err + (size_t)k * 5 * a.width);
                        da.sub_err[k] = err + (size_t)k * 5 * a.width + a.width;
                    }
                    da.need_delta = m->need_delta;
                    da.width = a.width; da.height = a.height; da.d_pred = tr.d_pred; da.bands = bands;
                    for (int i = 0; i < 11; ++i) da.wp[i] = tr.wp_params[i];
                    if (i16) palette_delta_kernel<int16_t><<<dim3(bands, nc), 256, 0, s>>>(da);
                    else palette_delta_kernel<int32_t><<<dim3(bands, nc), 256, 0, s>>>(da);
                    int failed = 0;
                    HIP_TRY(ctx, hipMemcpyAsync(&failed, da.fail, sizeof(int), hipMemcpyDeviceToHost, s));
                    HIP_TRY(ctx, hipStreamSynchronize(s));
                    if (failed) return fail(ctx, JXLGPU_ERR_DEVICE, "delta-palette wavefront: a workgroup waited too long for the band above");
                }
            }
            std::vector<int> members = leader.members;
            leader.members.clear();
            const Grid lead_copy = leader;
            for (size_t k = 0; k < members.size(); ++k) {
                Grid mg{members[k], lead_copy.x0, lead_copy.y0, lead_copy.w, lead_copy.h, 0, {}};
                l.insert(l.begin() + tr.begin_c + 1 + k, mg);
            }
        }
    }
    if (ctx->tune.debug_sync) {
        int redo = 0;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(&redo, m->d_redo, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemset(m->d_redo, 0, sizeof(int));
        fprintf(stderr, "[jxlgpu] squeeze lines redone serially in this run: %d\n", redo);
    }
    if (late_pending) { HIP_TRY(ctx, hipStreamWaitEvent(s, m->ev_late, 0)); late_pending = false; }
    m->final_loc.assign(nch, 0);
    for (const Grid& g : l)
        if (g.buf >= 0 && g.x0 == 0 && g.y0 == 0 && g.w == m->cw[g.buf] && g.h == m->ch[g.buf]) m->final_loc[g.buf] = g.loc;
    HIP_TRY(ctx, hipGetLastError());
    return JXLGPU_OK;
}

}  // namespace

extern "C" {

int jxlgpu_modular_upload(jxlgpu_ctx* ctx, const JxlGpuModularDesc* d, jxlgpu_frame** out_frame) {
    if (!ctx || !d || !out_frame) return JXLGPU_ERR_INVALID_ARG;
    *out_frame = nullptr;
    if (d->abi != JXLGPU_ABI_VERSION) return fail(ctx, JXLGPU_ERR_ABI, "descriptor ABI version mismatch");
    if (d->num_channels == 0 || !d->channels) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "no channels");
    if (d->num_color_channels != 0 && d->num_color_channels != 1 && d->num_color_channels != 3)
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "num_color_channels is 1 (grayscale) or 3");
    if (d->num_color_channels == 1 && d->xyb_encoded) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "a grayscale frame cannot be XYB encoded");
    if ((d->num_color_channels ? d->num_color_channels : 3u) > d->num_channels)
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "fewer channels than colour channels");
    if (d->residual_predictor != 0xFFFFFFFFu && d->residual_predictor > 13)
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "residual_predictor is neither 0xFFFFFFFF nor a Predictor (0..13)");
    // every frame that asks for the colour pass (color.enabled), XYB or not: the same checks as a VarDCT upload
    if (const char* why = color_params_unsupported(d->color)) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, why);
    if ((d->residual_predictor <= 13 || d->num_unit_leaves) && d->group_dim > kPredLaneMaxW) return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "predictor tiles wider than 1024");
    if (d->num_unit_leaves && !d->unit_leaves) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null unit_leaves");
    if (d->num_axis_leaves && !d->axis_leaves) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null axis_leaves");
    for (uint32_t i = 0; i < d->num_axis_leaves; ++i)
        if (d->axis_leaves[i].predictor > 13) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "a per-row / per-column leaf's predictor is not a Predictor (0..13)");
    for (uint32_t i = 0; i < d->num_unit_leaves; ++i) {
        const JxlGpuMaLeaf& lf = d->unit_leaves[i];
        if (lf.predictor > JXLGPU_LEAF_BY_COLUMN) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "a unit leaf's predictor is neither a Predictor (0..13) nor JXLGPU_LEAF_BY_ROW / _BY_COLUMN");
        if (lf.predictor >= JXLGPU_LEAF_BY_ROW && (lf.multiplier < 0 || lf.offset != 0 || (uint32_t)lf.multiplier >= d->num_axis_leaves))
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "a per-row / per-column unit leaf: multiplier is the first index into axis_leaves, offset 0");
    }
    if (d->num_transforms && !d->transforms) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null transform list");
    if (d->num_meta_channels && !d->meta_channels) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "null meta channel list");
    for (uint32_t c = 0; c < d->num_meta_channels; ++c)
        if (!d->meta_channels[c].data || !d->meta_channels[c].width || !d->meta_channels[c].height)
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "empty meta channel");
    {
        uint32_t meta_used = 0;
        for (uint32_t t = 0; t < d->num_transforms; ++t) {
            const JxlGpuTransform& tr = d->transforms[t];
            if (tr.kind > JXLGPU_TR_SQUEEZE) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "unknown transform kind");
            if (tr.kind == JXLGPU_TR_PALETTE) {
                if (meta_used >= d->num_meta_channels) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "Palette transform without its meta channel");
                const JxlGpuModularChannel& mc = d->meta_channels[meta_used++];
                if (mc.width < tr.nb_colours || mc.height < tr.num_c)
                    return fail(ctx, JXLGPU_ERR_INVALID_ARG, "palette meta channel smaller than nb_colours x num_c");
            }
        }
    }
    {
        const uint32_t upf = d->upsampling.factor ? d->upsampling.factor : 1;
        if (upf != 1 && upf != 2 && upf != 4 && upf != 8) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "bad upsampling factor");
        const float* need = upf == 2 ? d->upsampling.up2_weight : upf == 4 ? d->upsampling.up4_weight
                          : upf == 8 ? d->upsampling.up8_weight : (const float*)d;
        if (!need) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "upsampling weights missing");
        if (d->filter.epf_iters > 3) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "epf_iters > 3");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    jxlgpu_frame* f = new (std::nothrow) jxlgpu_frame();
    ModularState* m = new (std::nothrow) ModularState();
    if (!f || !m) { delete f; delete m; return JXLGPU_ERR_OOM; }
    f->kind_of_frame = 1;
    f->modular = m;
    f->modular_free = [](void* p) { delete static_cast<ModularState*>(p); };
    struct Guard {
        jxlgpu_ctx* c; jxlgpu_frame* f; bool armed = true;
        ~Guard() { if (armed) jxlgpu_frame_free(c, f); }
    } guard{ctx, f};
    m->desc = *d;
    if (d->num_unit_leaves) m->unit_leaves.assign(d->unit_leaves, d->unit_leaves + d->num_unit_leaves);
    m->desc.unit_leaves = nullptr;   // (the caller's array may be gone after this call)
    if (d->num_axis_leaves) m->axis_leaves.assign(d->axis_leaves, d->axis_leaves + d->num_axis_leaves);
    m->desc.axis_leaves = nullptr;
    m->esz = d->sample_type == JXLGPU_SAMPLE_I16 ? 2 : 4;
    for (uint32_t c = 0; c < d->num_channels; ++c) {
        const JxlGpuModularChannel& ch = d->channels[c];
        if (!ch.data || ch.width == 0 || ch.height == 0) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "empty channel");
        size_t bytes = (size_t)ch.width * ch.height * m->esz;
        void* p = nullptr;
        int rc;
        if ((rc = malloc_dev(ctx, f, &p, bytes))) return rc;
        HIP_TRY(ctx, hipMemcpy(p, ch.data, bytes, hipMemcpyHostToDevice));
        m->orig.push_back(p);
        m->cw.push_back(ch.width);
        m->ch.push_back(ch.height);
        for (int k = 0; k < 3; ++k) {
            if ((rc = malloc_dev(ctx, f, &p, bytes))) return rc;
            m->work[k].push_back(p);
        }
    }
    for (uint32_t c = 0; c < d->num_meta_channels; ++c) {
        const JxlGpuModularChannel& ch = d->meta_channels[c];
        size_t bytes = (size_t)ch.width * ch.height * m->esz;
        void* p = nullptr;
        int rc;
        if ((rc = malloc_dev(ctx, f, &p, bytes))) return rc;
        HIP_TRY(ctx, hipMemcpy(p, ch.data, bytes, hipMemcpyHostToDevice));
        m->meta.push_back(p);
        m->mw.push_back(ch.width);
        m->mh.push_back(ch.height);
        if (d->residual_predictor <= 13 || d->num_unit_leaves) {  // the predictor pass rewrites the palette table: it needs a copy of its own
            if ((rc = malloc_dev(ctx, f, &p, bytes))) return rc;
            m->meta_work.push_back(p);
        }
    }
    for (uint32_t t = 0; t < d->num_transforms; ++t) {
        m->transforms.push_back(d->transforms[t]);
        const JxlGpuTransform& tr = d->transforms[t];
        // explicit squeeze parameters are copied now (the descriptor may be released after the call)
        if (tr.kind == JXLGPU_TR_SQUEEZE && tr.num_sq && tr.sq) m->explicit_steps.emplace_back(tr.sq, tr.sq + tr.num_sq);
        else m->explicit_steps.emplace_back();
        m->transforms.back().sq = nullptr;
    }
    int rc;
    if ((rc = malloc_dev(ctx, f, &m->d_flag, sizeof(int)))) return rc;
    if ((rc = malloc_dev(ctx, f, &m->d_redo, sizeof(int)))) return rc;
    HIP_TRY(ctx, hipMemset(m->d_redo, 0, sizeof(int)));
    {   // 2 link values per segment per line; segments are >= 16 pairs, so len/16 bounds nseg
        size_t worst = 0;
        for (size_t c = 0; c < m->cw.size(); ++c) {
            size_t w = m->cw[c], h = m->ch[c];
            worst = std::max(worst, std::max((w / 32 + 1) * 2 * h, (h / 32 + 1) * 2 * w) * m->esz);
        }
        worst = (worst + 15) / 16 * 16 * 3;  // three channels of a step share one launch
        m->chk_bytes = worst;
        if ((rc = malloc_dev(ctx, f, &m->chk, worst))) return rc;
    }

    // geometry of the colour image for the float tail
    // the colour channels of a chroma-subsampled frame (do_ycbcr + jpeg_upsampling) differ in size: the frame is the largest
    f->width = d->channels[0].width;
    f->height = d->channels[0].height;
    for (uint32_t c = 1; c < std::min<uint32_t>(d->num_color_channels == 1 ? 1u : 3u, d->num_channels); ++c) {
        f->width = std::max(f->width, d->channels[c].width);
        f->height = std::max(f->height, d->channels[c].height);
    }
    // several per-row kernels launch one grid row per image row (HIP: grid.y <= 65535)
    if ((uint64_t)f->height * (d->upsampling.factor ? d->upsampling.factor : 1) > 65535u)
        return fail(ctx, JXLGPU_ERR_UNSUPPORTED, "output taller than 65535 rows");
    f->w8 = ceil_div(f->width, 8); f->h8 = ceil_div(f->height, 8);
    f->wr = f->w8 * 8; f->hr = f->h8 * 8;
    f->desc.filter = d->filter;
    f->desc.upsampling = d->upsampling;
    f->desc.noise = d->noise;
    f->noise_group_dim = d->group_dim ? d->group_dim : 256;
    f->noise_corr_x = 0.0f;  // no VarDCT LfGlobal: base_correlations_xb = None -> (0, 1), noise.rs:35
    f->noise_corr_b = 1.0f;
    f->desc.color = d->color;
    if (!d->xyb_encoded) f->desc.color.enabled = 0;
    fill_color_args_public(f->desc.color, &f->color);
    {
        const size_t npix = (size_t)f->wr * f->hr;
        for (int c = 0; c < 3; ++c) {
            if ((rc = malloc_dev(ctx, f, &m->fpix[c], npix * 4))) return rc;
            if ((rc = malloc_dev(ctx, f, &f->buf_a[c], npix * 4))) return rc;
            if ((rc = malloc_dev(ctx, f, &f->buf_b[c], npix * 4))) return rc;
        }
        std::vector<float> sigma((size_t)f->w8 * f->h8, d->filter.epf_sigma_for_modular);
        if ((rc = malloc_dev(ctx, f, &f->sigma, sigma.size() * 4))) return rc;
        HIP_TRY(ctx, hipMemcpy(f->sigma, sigma.data(), sigma.size() * 4, hipMemcpyHostToDevice));
        const uint32_t upf = d->upsampling.factor ? d->upsampling.factor : 1;
        if (upf > 1) {
            for (int c = 0; c < 3; ++c)
                if ((rc = malloc_dev(ctx, f, &f->up[c], (size_t)f->width * upf * f->height * upf * 4))) return rc;
            if ((rc = upload_post_params(ctx, f, d->upsampling))) return rc;
        }
    }
    guard.armed = false;
    *out_frame = f;
    return JXLGPU_OK;
}

int jxlgpu_modular_inverse(jxlgpu_ctx* ctx, jxlgpu_frame* f, void* const* planes) {
    if (!ctx || !f || f->kind_of_frame != 1) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ModularState* m = static_cast<ModularState*>(f->modular);
    ctx->prof_begin(PROF_MODULAR);
    int rc = run_inverse(ctx, f);
    ctx->prof_end(PROF_MODULAR);
    if (rc) return rc;
    if (planes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t c = 0; c < m->orig.size(); ++c)
            if (planes[c])
                HIP_TRY(ctx, hipMemcpy(planes[c], m->work[m->final_loc[c]][c], (size_t)m->cw[c] * m->ch[c] * m->esz,
                                       hipMemcpyDeviceToHost));
    }
    return JXLGPU_OK;
}

static int modular_render_impl(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuRegion* region_in, const JxlGpuOut* out);

int jxlgpu_modular_render(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuOut* out) {
    return modular_render_impl(ctx, f, stages, nullptr, out);
}

int jxlgpu_modular_render_region(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuRegion* region,
                                 const JxlGpuOut* out) {
    if (!region) return JXLGPU_ERR_INVALID_ARG;
    return modular_render_impl(ctx, f, stages, region, out);
}

static int modular_render_impl(jxlgpu_ctx* ctx, jxlgpu_frame* f, uint32_t stages, const JxlGpuRegion* region_in, const JxlGpuOut* out) {
    if (!ctx || !f || f->kind_of_frame != 1) return JXLGPU_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ModularState* m = static_cast<ModularState*>(f->modular);
    // grayscale (render.rs:74-134): the one colour channel feeds all three filter inputs; plane 0 is the result
    const bool gray = m->desc.num_color_channels == 1;
    // chroma-subsampled colour channels (frame_header.do_ycbcr + jpeg_upsampling on a Modular frame): int -> float per
    // channel at its own size, then ImageWithRegion::upsample_jpeg (image.rs:448-485) as for the VarDCT JPEG path
    int hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
    bool subsampled = false;
    for (int c = 0; c < 3 && !gray; ++c) {
        if (m->cw[c] == f->width && m->ch[c] == f->height) continue;
        hs[c] = m->cw[c] != f->width; vs[c] = m->ch[c] != f->height;
        if ((hs[c] && m->cw[c] != (f->width + 1) / 2) || (vs[c] && m->ch[c] != (f->height + 1) / 2))
            return fail(ctx, JXLGPU_ERR_INVALID_ARG, "colour channel size is neither the frame's nor half of it");
        subsampled = true;
    }
    if (subsampled && (m->desc.xyb_encoded || !m->desc.color.ycbcr))
        return fail(ctx, JXLGPU_ERR_INVALID_ARG, "subsampled colour channels need a YCbCr frame (color.ycbcr, not XYB)");
    if (gray) stages &= ~(uint32_t)JXLGPU_STAGE_NOISE;  // render.rs:208-221: "Cannot render noise on grayscale buffer; skipping"
    if (!(stages & JXLGPU_STAGE_MODULAR_TO_FLOAT)) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "render needs JXLGPU_STAGE_MODULAR_TO_FLOAT");
    ctx->prof_begin(PROF_MODULAR);
    int rc = run_inverse(ctx, f);
    ctx->prof_end(PROF_MODULAR);
    if (rc) return rc;
    ToFloatArgs a;
    for (int c = 0; c < 3; ++c) {
        const int src = gray ? 0 : c;
        a.in[c] = m->work[m->final_loc[src]][src];
        a.in_stride[c] = m->cw[src];
        a.cw[c] = m->cw[src]; a.ch[c] = m->ch[src];
        a.out[c] = (hs[c] || vs[c]) ? f->buf_a[c] : m->fpix[c];   // a subsampled channel is upsampled into fpix below
        a.m[c] = m->desc.m_lf_unscaled[c];
    }
    a.out_stride = f->wr; a.width = f->width; a.height = f->height;
    a.xyb = m->desc.xyb_encoded; a.is_i16 = m->desc.sample_type == JXLGPU_SAMPLE_I16;
    a.bit_depth = m->desc.bit_depth; a.float_sample = m->desc.float_sample; a.exp_bits = m->desc.exp_bits;
    // JXLGPU_INT_POST=1 (measured option, off by default): XYB frames whose post stage is the packed streaming kernel (Gabor /
    // EPF iters 2: BASELINE config 3): that kernel and the border-ring kernel read the INTEGER planes and convert on the fly —
    // no float copy of the frame (12 B/px written + read).  Bit-identical; 4 % slower on config 3 (the copy is HBM-bound and
    // cheap, the conversion lands in a VALU-bound kernel)
    const int epf_iters_run = (stages & JXLGPU_STAGE_EPF) ? (int)f->desc.filter.epf_iters : 0;
    const bool int_post = a.xyb && !gray && !subsampled && !region_in && a.in_stride[0] == a.in_stride[1] && a.in_stride[0] == a.in_stride[2] &&
                          fused_int_input_supported(ctx, f, epf_iters_run, a.in, a.in_stride[0], m->esz);
    if (!int_post) to_float_kernel<<<dim3(ceil_div(f->width, 256), f->height), 256, 0, ctx->stream>>>(a);
    for (int c = 0; c < 3 && subsampled; ++c)
        if (hs[c] || vs[c])
            launch_upsample_jpeg_rows(ctx->stream, f->buf_a[c], f->wr, m->cw[c], m->ch[c], hs[c], vs[c], m->fpix[c], f->wr, f->width, f->height);
    float* cur[3] = {m->fpix[0], m->fpix[1], m->fpix[2]};
    uint32_t stride = f->wr, ow = f->width, oh = f->height;
    if (int_post) {
        for (int c = 0; c < 3; ++c) { cur[c] = reinterpret_cast<float*>(const_cast<void*>(a.in[c])); f->post_in_m[c] = a.m[c]; }
        stride = a.in_stride[0];
        f->post_in_int = a.is_i16 ? 1u : 2u;
    }
    ctx->prof_begin(PROF_POST);
    PixRect region{0, 0, 0, 0};
    bool cut = false;
    if (region_in) {
        uint32_t fw = 0, fh = 0;
        jxlgpu_frame_out_size(f, stages, &fw, &fh);
        if (!clip_region(region_in, fw, fh, &region)) return fail(ctx, JXLGPU_ERR_INVALID_ARG, "the region does not intersect the frame");
        const bool any_filter = ((stages & JXLGPU_STAGE_GABOR) && f->desc.filter.gab_enabled) || ((stages & JXLGPU_STAGE_EPF) && f->desc.filter.epf_iters);
        const bool noisy = (stages & JXLGPU_STAGE_NOISE) && f->desc.noise.enabled;  // seeded per absolute group: whole frame, then crop
        // the staged filters run on whole planes: crop afterwards; so does a chroma-subsampled frame (its planes are
        // upsampled whole above, as the VarDCT JPEG-transcode path does: whole frame, the region cropped from it)
        cut = !noisy && !subsampled && (!any_filter || fused_post_supported(ctx, f, true, 2));
    }
    rc = run_post_stages(ctx, f, stages, f->desc.filter, f->desc.upsampling.factor ? f->desc.upsampling.factor : 1,
                         cur, &stride, &ow, &oh, false, cut ? &region : nullptr);
    f->post_in_int = 0;
    ctx->prof_end(PROF_POST);
    if (rc) return rc;
    if (region_in) return finish_render_region(ctx, f, cur, stride, region, out);
    return finish_render(ctx, f, cur, stride, ow, oh, out);
}

}  // extern "C"
