// Varblocks with a side >= 128 (Dct128, Dct128x64, Dct64x128, Dct256, Dct256x128, Dct128x256).
//
// Legal but rare (none in typical d1 streams).  A 256x256 f32 block is 256 KiB — more than the
// 160 KiB of LDS on a gfx950 CU — so this path keeps the block in HBM/L2 and runs the same
// even/odd DCT recursion as dct_device.h level by level (de-interleave + neighbour add going
// down, register idct<8> at the leaves, sec-scaled butterflies going up), one lane per row, then
// one lane per column.  Same operation order as generic/dct.rs:272-291, so bit-identical.
#include "common.h"
#include "dct_device.h"

struct Strided {
    float* p;
    size_t s;
    __device__ __forceinline__ float& operator[](int i) const { return p[(size_t)i * s]; }
};

__device__ __forceinline__ const float* sec_table(int n, const SecLarge& sl) {
    switch (n) {
        case 16: return kSec16;
        case 32: return kSec32;
        case 64: return sl.s64;
        case 128: return sl.s128;
        default: return sl.s256;
    }
}

// In-place (result in `a`) inverse DCT of length n (power of two, >= 16) using `b` as the
// ping-pong buffer.
__device__ void idct_iterative(Strided a, Strided b, int n, const SecLarge& sl) {
    Strided src = a, dst = b;
    for (int len = n; len > 8; len >>= 1) {
        int h = len >> 1;
        for (int off = 0; off < n; off += len) {
            for (int i = 0; i < h; ++i) dst[off + i] = src[off + 2 * i];
            dst[off + h] = src[off + 1] * JXL_SQRT2F;
            for (int i = 1; i < h; ++i) dst[off + h + i] = src[off + 2 * i + 1] + src[off + 2 * i - 1];
        }
        Strided t = src; src = dst; dst = t;
    }
    for (int off = 0; off < n; off += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[off + i];
        idct<8>(v, sl);
#pragma unroll
        for (int i = 0; i < 8; ++i) src[off + i] = v[i];
    }
    for (int len = 16; len <= n; len <<= 1) {
        int h = len >> 1;
        const float* sec = sec_table(len, sl);
        for (int off = 0; off < n; off += len) {
            for (int i = 0; i < h; ++i) {
                float r = src[off + h + i] * sec[i];
                float e = src[off + i];
                dst[off + i] = e + r;
                dst[off + len - 1 - i] = e - r;
            }
        }
        Strided t = src; src = dst; dst = t;
    }
    // log2(n/8) down-swaps + log2(n/8) up-swaps: even, so `src` is `a` again
}

template <int N>
__device__ __forceinline__ void fdct_strided(Strided p, const SecLarge& sl) {
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[i];
    fdct<N>(v, sl);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = v[i];
}

__device__ __forceinline__ void fdct_dyn(Strided p, int n, const SecLarge& sl) {
    switch (n) {
        case 8: fdct_strided<8>(p, sl); break;
        case 16: fdct_strided<16>(p, sl); break;
        default: fdct_strided<32>(p, sl); break;
    }
}

__device__ __forceinline__ float dequant_big(int32_t qn, float quant_bias, float qbn, float m, float mul) {
    float q = (float)qn;
    if (fabsf(q) <= 1.0f) q *= quant_bias;
    else q -= qbn / q;
    q *= m;
    q *= mul;
    return q;
}

// One workgroup per (varblock, channel).
__global__ __launch_bounds__(256) void big_block_kernel(TransformArgs a, const uint4* list, float* tmp) {
    __shared__ float llf[32 * 33];
    const SecLarge sl{a.sec64, a.sec128, a.sec256};
    const int t = threadIdx.x;
    const int c = blockIdx.y;
    uint32_t e = list[blockIdx.x].x;
    uint32_t cx = e & 0xffffu, cy = e >> 16;
    size_t cell = (size_t)cy * a.w8 + cx;
    uint32_t type = a.kind[cell];
    int bw, bh;
    switch (type) {
        case JXLGPU_DCT128: bw = 16; bh = 16; break;
        case JXLGPU_DCT128X64: bw = 8; bh = 16; break;
        case JXLGPU_DCT64X128: bw = 16; bh = 8; break;
        case JXLGPU_DCT256: bw = 32; bh = 32; break;
        case JXLGPU_DCT256X128: bw = 16; bh = 32; break;
        default: bw = 32; bh = 16; break;  // JXLGPU_DCT128X256
    }
    const int W = bw * 8, H = bh * 8;
    const uint32_t px0 = cx * 8, py0 = cy * 8;
    float hm = (float)a.hf_mul[cell];
    float mul_c = 65536.0f / (a.global_scale * hm) * a.qm_scale[c];
    float mul_y = 65536.0f / (a.global_scale * hm) * a.qm_scale[1];
    const float* mat_c = a.dequant + a.deq_off[type * 3 + c];
    const float* mat_y = a.dequant + a.deq_off[type * 3 + 1];
    // both working copies are row-major scratch planes; the result moves to the tiled output last
    const size_t plane = (size_t)a.pstride * (a.h8 * 8);
    float* out = tmp + (size_t)(3 + c) * plane + (size_t)py0 * a.pstride + px0;
    float* scratch = tmp + (size_t)c * plane + (size_t)py0 * a.pstride + px0;

    // V4 + V5
    for (int i = t; i < W * H; i += 256) {
        int y = i / W, x = i % W;
        float v = dequant_big(a.coeff[coeff_tiled_index(px0 + x, py0 + y, c, a.w8)], a.quant_bias[c],
                              a.quant_bias_numerator, mat_c[y * W + x], mul_c);
        if (c != 1) {
            float yv = dequant_big(a.coeff[coeff_tiled_index(px0 + x, py0 + y, 1, a.w8)], a.quant_bias[1],
                                   a.quant_bias_numerator, mat_y[y * W + x], mul_y);
            uint32_t ti = ((py0 + y) >> 6) * a.w64 + ((px0 + x) >> 6);
            float k = c == 0 ? a.kx_map[ti] : a.kb_map[ti];
            v += k * yv;
        }
        out[(size_t)y * a.pstride + x] = v;
    }
    // V6: LLF (transform_common.rs:51-66), bw,bh >= 8: general dct_2d path (rows, then columns)
    for (int i = t; i < bw * bh; i += 256) {
        int y = i / bw, x = i % bw;
        llf[y * 33 + x] = a.lf[c][cell + (size_t)y * a.w8 + x];
    }
    __syncthreads();
    if (t < bh) fdct_dyn(Strided{llf + t * 33, 1}, bw, sl);
    __syncthreads();
    if (t < bw) fdct_dyn(Strided{llf + t, 33}, bh, sl);
    __syncthreads();
    {
        int sy = 5 - (31 - __builtin_clz(bh)), sx = 5 - (31 - __builtin_clz(bw));
        for (int i = t; i < bw * bh; i += 256) {
            int y = i / bw, x = i % bw;
            out[(size_t)y * a.pstride + x] = llf[y * 33 + x] / (kScaleF[y << sy] * kScaleF[x << sx]);
        }
    }
    __syncthreads();
    // V7: rows, then columns
    for (int r = t; r < H; r += 256)
        idct_iterative(Strided{out + (size_t)r * a.pstride, 1}, Strided{scratch + (size_t)r * a.pstride, 1}, W, sl);
    __syncthreads();
    for (int x = t; x < W; x += 256)
        idct_iterative(Strided{out + x, a.pstride}, Strided{scratch + x, a.pstride}, H, sl);
    __syncthreads();
    for (int i = t; i < W * H; i += 256) {
        const int y = i / W, x = i % W;
        a.pix[coeff_tiled_index(px0 + x, py0 + y, c, a.w8)] = out[(size_t)y * a.pstride + x];
    }
}

void launch_big_blocks(hipStream_t s, const TransformArgs& a, const uint4* list, uint32_t count) {
    big_block_kernel<<<dim3(count, 3), 256, 0, s>>>(a, list, a.big_tmp);
}
