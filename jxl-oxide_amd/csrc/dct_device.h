// 1-D DCT-II / DCT-III butterflies on register arrays, gfx950.
//
// Same recursion and operation order as the reference's generic scalar code
// (jxl-render/src/vardct/generic/dct.rs:144-293), expressed as compile-time recursion over
// register arrays so every butterfly is straight-line VALU code with immediate constants.  Built
// with -ffp-contract=off: the reference never fuses a*b+c, so neither may we — results are
// bit-identical to the CPU path, not merely close.
#pragma once

#include "common.h"

// sec_half(n)[k] = 1/(2 cos((2k+1)pi/2n)), jxl-render/src/vardct/dct_common.rs:10-50 (data).
__device__ constexpr float kSec8[4] = {0.5097955791041592f, 0.6013448869350453f,
                                       0.8999762231364156f, 2.5629154477415055f};
__device__ constexpr float kSec16[8] = {0.5024192861881557f, 0.5224986149396889f,
                                        0.5669440348163577f, 0.6468217833599901f,
                                        0.7881546234512502f, 1.060677685990347f,
                                        1.7224470982383342f, 5.101148618689155f};
__device__ constexpr float kSec32[16] = {
    0.5006029982351963f, 0.5054709598975436f, 0.5154473099226246f, 0.5310425910897841f,
    0.5531038960344445f, 0.5829349682061339f, 0.6225041230356648f, 0.6748083414550057f,
    0.7445362710022984f, 0.8393496454155268f, 0.9725682378619608f, 1.1694399334328847f,
    1.4841646163141662f, 2.057781009953411f,  3.407608418468719f,  10.190008123548033f};
// SCALE_F, dct_common.rs:80-113 (data)
__device__ constexpr float kScaleF[32] = {
    1.0000000000000000f, 0.9996047255830407f, 0.9984194528776054f, 0.9964458326264695f,
    0.9936866130906366f, 0.9901456355893141f, 0.9858278282666936f, 0.9807391980963174f,
    0.9748868211368796f, 0.9682788310563117f, 0.9609244059440204f, 0.9528337534340876f,
    0.9440180941651672f, 0.9344896436056892f, 0.9242615922757944f, 0.9133480844001980f,
    0.9017641950288744f, 0.8895259056651056f, 0.8766500784429904f, 0.8631544288990163f,
    0.8490574973847023f, 0.8343786191696513f, 0.8191378932865928f, 0.8033561501721485f,
    0.7870549181591013f, 0.7702563888779096f, 0.7529833816270532f, 0.7352593067735488f,
    0.7171081282466044f, 0.6985543251889097f, 0.6796228528314652f, 0.6603391026591464f};

// Tables for n = 64/128/256 come from the host (the reference computes them at run time with
// f32 cos(), dct_common.rs:56-66); addressed with compile-time indices off a uniform pointer, so
// they are scalar loads.
struct SecLarge {
    const float* s64;
    const float* s128;
    const float* s256;
};

template <int N>
__device__ __forceinline__ float sec_at(int k, const SecLarge& sl) {
    if constexpr (N == 8) return kSec8[k];
    else if constexpr (N == 16) return kSec16[k];
    else if constexpr (N == 32) return kSec32[k];
    else if constexpr (N == 64) return sl.s64[k];
    else if constexpr (N == 128) return sl.s128[k];
    else return sl.s256[k];
}

// ---------------------------------------------------------------- inverse (DCT-III, unscaled)
template <int N>
__device__ __forceinline__ void idct(float (&v)[N], const SecLarge& sl) {
    if constexpr (N == 1) {
        return;
    } else if constexpr (N == 2) {
        float t0 = v[0] + v[1];
        float t1 = v[0] - v[1];
        v[0] = t0;
        v[1] = t1;
    } else if constexpr (N == 4) {
        // dct4(), dct.rs:162-171
        const float sec0 = 0.5411961f, sec1 = 1.306563f;
        float tmp0 = v[1] * JXL_SQRT2F;
        float tmp1 = v[1] + v[3];
        float out0 = (tmp0 + tmp1) * sec0;
        float out1 = (tmp0 - tmp1) * sec1;
        float sum02 = v[0] + v[2];
        float sub02 = v[0] - v[2];
        v[0] = sum02 + out0;
        v[1] = sub02 + out1;
        v[2] = sub02 - out1;
        v[3] = sum02 - out0;
    } else {
        // dct.rs:229-244 (n == 8) and :272-291 (n >= 16) are the same even/odd split
        constexpr int H = N / 2;
        float in0[H], in1[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            in0[i] = v[2 * i];
            in1[i] = v[2 * i + 1];
        }
#pragma unroll
        for (int i = H - 1; i >= 1; --i) in1[i] += in1[i - 1];
        in1[0] *= JXL_SQRT2F;
        idct<H>(in0, sl);
        idct<H>(in1, sl);
#pragma unroll
        for (int i = 0; i < H; ++i) {
            float r = in1[i] * sec_at<N>(i, sl);
            v[i] = in0[i] + r;
            v[N - 1 - i] = in0[i] - r;
        }
    }
}

// ---------------------------------------------------------------- forward (DCT-II, halving)
template <int N>
__device__ __forceinline__ void fdct(float (&v)[N], const SecLarge& sl) {
    if constexpr (N == 1) {
        return;
    } else if constexpr (N == 2) {
        float t0 = v[0] + v[1];
        float t1 = v[0] - v[1];
        v[0] = t0 / 2.0f;
        v[1] = t1 / 2.0f;
    } else if constexpr (N == 4) {
        // dct4(), dct.rs:148-161
        const float sec0 = 0.5411961f, sec1 = 1.306563f;
        float sum03 = v[0] + v[3];
        float sum12 = v[1] + v[2];
        float tmp0 = (v[0] - v[3]) * sec0;
        float tmp1 = (v[1] - v[2]) * sec1;
        float out0 = (tmp0 + tmp1) / 4.0f;
        float out1 = (tmp0 - tmp1) / 4.0f;
        v[0] = (sum03 + sum12) / 4.0f;
        v[1] = out0 * JXL_SQRT2F + out1;
        v[2] = (sum03 - sum12) / 4.0f;
        v[3] = out1;
    } else if constexpr (N == 8) {
        // dct.rs:206-228
        float in0[4], in1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            in0[i] = (v[i] + v[7 - i]) / 2.0f;
            in1[i] = (v[i] - v[7 - i]) * kSec8[i] / 2.0f;
        }
        fdct<4>(in0, sl);
        fdct<4>(in1, sl);
        in1[0] *= JXL_SQRT2F;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[2 * i] = in0[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) v[2 * i + 1] = in1[i] + in1[i + 1];
        v[7] = in1[3];
    } else {
        // dct.rs:250-271
        constexpr int H = N / 2;
        float in0[H], in1[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            in0[i] = (v[i] + v[N - 1 - i]) / 2.0f;
            in1[i] = (v[i] - v[N - 1 - i]) / 2.0f;
        }
#pragma unroll
        for (int i = 0; i < H; ++i) in1[i] *= sec_at<N>(i, sl);
        fdct<H>(in0, sl);
        fdct<H>(in1, sl);
        in1[0] *= JXL_SQRT2F;
#pragma unroll
        for (int i = 0; i < H - 1; ++i) in1[i] += in1[i + 1];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            v[2 * i] = in0[i];
            v[2 * i + 1] = in1[i];
        }
    }
}

// 2-D forward DCT of a BW x BH block held as v[y][x], with every special case of
// dct_2d (dct.rs:5-141) in the reference's order.  Used for the LF -> LLF injection
// (transform_common.rs:51-66), BW,BH in {1,2,4,8,16,32}.
template <int BW, int BH>
__device__ __forceinline__ void fdct2d_small(float (&v)[BH][BW], const SecLarge& sl) {
    if constexpr (BW * BH <= 1) {
        return;
    } else if constexpr (BW == 2 && BH == 1) {
        float v0 = v[0][0], v1 = v[0][1];
        v[0][0] = (v0 + v1) * 0.5f;
        v[0][1] = (v0 - v1) * 0.5f;
    } else if constexpr (BW == 1 && BH == 2) {
        float v0 = v[0][0], v1 = v[1][0];
        v[0][0] = (v0 + v1) * 0.5f;
        v[1][0] = (v0 - v1) * 0.5f;
    } else if constexpr (BW == 2 && BH == 2) {
        float v00 = v[0][0], v01 = v[0][1], v10 = v[1][0], v11 = v[1][1];
        v[0][0] = (v00 + v01 + v10 + v11) * 0.5f * 0.5f;
        v[0][1] = (v00 - v01 + v10 - v11) * 0.5f * 0.5f;
        v[1][0] = (v00 + v01 - v10 - v11) * 0.5f * 0.5f;
        v[1][1] = (v00 - v01 - v10 + v11) * 0.5f * 0.5f;
    } else if constexpr (BH == 1) {
        fdct<BW>(v[0], sl);
    } else if constexpr (BW == 1) {
        float col[BH];
#pragma unroll
        for (int y = 0; y < BH; ++y) col[y] = v[y][0];
        fdct<BH>(col, sl);
#pragma unroll
        for (int y = 0; y < BH; ++y) v[y][0] = col[y];
    } else if constexpr (BH == 2) {
#pragma unroll
        for (int x = 0; x < BW; ++x) {
            float t0 = v[0][x], t1 = v[1][x];
            v[0][x] = (t0 + t1) * 0.5f;
            v[1][x] = (t0 - t1) * 0.5f;
        }
        fdct<BW>(v[0], sl);
        fdct<BW>(v[1], sl);
    } else if constexpr (BW == 2) {
        float c0[BH], c1[BH];
#pragma unroll
        for (int y = 0; y < BH; ++y) {
            float a = v[y][0], b = v[y][1];
            c0[y] = (a + b) * 0.5f;
            c1[y] = (a - b) * 0.5f;
        }
        fdct<BH>(c0, sl);
        fdct<BH>(c1, sl);
#pragma unroll
        for (int y = 0; y < BH; ++y) {
            v[y][0] = c0[y];
            v[y][1] = c1[y];
        }
    } else {
#pragma unroll
        for (int y = 0; y < BH; ++y) fdct<BW>(v[y], sl);
#pragma unroll
        for (int x = 0; x < BW; ++x) {
            float col[BH];
#pragma unroll
            for (int y = 0; y < BH; ++y) col[y] = v[y][x];
            fdct<BH>(col, sl);
#pragma unroll
            for (int y = 0; y < BH; ++y) v[y][x] = col[y];
        }
    }
}
