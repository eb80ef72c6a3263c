// Noise synthesis on the device (jxl-render/src/features/noise.rs).
//
// The reference fills one noise buffer per 256x256 group from an 8-lane xorshift128+ stream that
// runs serially through the group's three channels (NoiseGroup::new, noise.rs:199-231), convolves
// it with a 5x5 kernel whose rows are summed in ring-buffer order (convolve_fill, :240-318) and
// adds the modulated result to X, Y, B (render_noise, :12-90).  Here:
//   * noise_fill_kernel: the xorshift128+ state transition is linear over GF(2), so the state
//     after k steps is M^k * state.  M^(2^t), t = 0..17, are built once on the host (128x128 bit
//     matrices); every (group, channel, 16-row block, RNG lane) chain jumps straight to its first
//     batch and then steps normally — 6 144 independent chains per 256x256 group instead of 8.
//   * noise_apply_kernel: works on the assembled noise image with a mirrored 2-sample border,
//     which is what the reference's 9-neighbour padding amounts to (checked against the literal
//     adjacency logic by the oracle, tests/test_oracle_noise.py); the 25 taps are added in the
//     reference's order (ring-buffer rows, i.e. rotated by the group-local row mod 5).
// Integer work is exact; float operations follow the reference one by one (-ffp-contract=off).
#include <mutex>
#include <vector>

#include "common.h"
#include "pixel_device.h"

namespace {

constexpr int kJumpLevels = 18;   // jumps of up to 2^18 - 1 batches (a 1024x1024 group needs < 3 * 64 * 1024)
constexpr int kRowsPerChain = 16;

struct U128 {
    uint64_t lo, hi;  // lo = XorShift128Plus::s0[i], hi = ::s1[i]
};

// fill_batch's state update (noise.rs:437-447) for one lane
__host__ __device__ inline U128 xs_step(U128 s) {
    uint64_t s1 = s.lo;
    const uint64_t s0 = s.hi;
    U128 r;
    r.lo = s0;
    s1 ^= s1 << 23;
    r.hi = s1 ^ (s0 ^ (s1 >> 18) ^ (s0 >> 5));
    return r;
}

__host__ __device__ inline uint64_t split_mix_64(uint64_t z) {  // noise.rs:451-456
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// column-major bit matrix: col[j] = image of basis vector e_j
struct BitMat {
    U128 col[128];
};

U128 mat_vec(const BitMat& m, U128 v) {
    U128 r{0, 0};
    for (int j = 0; j < 128; ++j) {
        const uint64_t bit = j < 64 ? (v.lo >> j) & 1 : (v.hi >> (j - 64)) & 1;
        if (bit) { r.lo ^= m.col[j].lo; r.hi ^= m.col[j].hi; }
    }
    return r;
}

const std::vector<uint32_t>& jump_table_host() {
    static std::vector<uint32_t> table;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<BitMat> mats(kJumpLevels);
        for (int j = 0; j < 128; ++j) {
            U128 e{j < 64 ? 1ull << j : 0, j >= 64 ? 1ull << (j - 64) : 0};
            mats[0].col[j] = xs_step(e);
        }
        for (int t = 1; t < kJumpLevels; ++t)
            for (int j = 0; j < 128; ++j) mats[t].col[j] = mat_vec(mats[t - 1], mats[t - 1].col[j]);
        table.resize((size_t)kJumpLevels * 128 * 4);
        for (int t = 0; t < kJumpLevels; ++t)
            for (int j = 0; j < 128; ++j) {
                uint32_t* o = &table[((size_t)t * 128 + j) * 4];
                o[0] = (uint32_t)mats[t].col[j].lo; o[1] = (uint32_t)(mats[t].col[j].lo >> 32);
                o[2] = (uint32_t)mats[t].col[j].hi; o[3] = (uint32_t)(mats[t].col[j].hi >> 32);
            }
    });
    return table;
}

struct NoiseFillArgs {
    float* raw[3];          // W x H, tight
    uint32_t width, height; // frame size after upsampling (header.width/height, noise.rs:102-106)
    uint32_t group_dim, groups_per_row;
    uint64_t seed0;
    const uint4* jump;      // kJumpLevels x 128 columns
};

// grid (num_groups, 3 channels), block = (group_dim / kRowsPerChain) x 8 RNG lanes
__global__ __launch_bounds__(512) void noise_fill_kernel(NoiseFillArgs a) {
    const uint32_t g = blockIdx.x, c = blockIdx.y;
    const uint32_t lane = threadIdx.x & 7, blk = threadIdx.x >> 3;
    const uint32_t x0 = (g % a.groups_per_row) * a.group_dim, y0 = (g / a.groups_per_row) * a.group_dim;
    const uint32_t gw = min(a.group_dim, a.width - x0), gh = min(a.group_dim, a.height - y0);
    const uint32_t row0 = blk * kRowsPerChain;
    if (row0 >= gh) return;
    const uint32_t w16 = (gw + 15) / 16;

    // XorShift128Plus::new (noise.rs:408-425)
    const uint64_t seed1 = ((uint64_t)x0 << 32) + (uint64_t)y0;  // rng_seed1, noise.rs:174-177
    uint64_t s0 = split_mix_64(a.seed0 + 0x9E3779B97F4A7C15ull);
    uint64_t s1 = split_mix_64(seed1 + 0x9E3779B97F4A7C15ull);
    for (uint32_t i = 0; i < lane; ++i) { s0 = split_mix_64(s0); s1 = split_mix_64(s1); }

    // jump over the batches of the earlier channels and rows
    const uint32_t skip = c * w16 * gh + row0 * w16;
    uint32_t v[4] = {(uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)s1, (uint32_t)(s1 >> 32)};
    for (int t = 0; t < kJumpLevels; ++t) {
        if (!((skip >> t) & 1)) continue;
        const uint4* m = a.jump + (size_t)t * 128;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t word = v[w];
#pragma unroll 8
            for (int b = 0; b < 32; ++b) {
                const uint32_t mask = 0u - ((word >> b) & 1u);
                const uint4 col = m[w * 32 + b];
                r0 ^= col.x & mask; r1 ^= col.y & mask; r2 ^= col.z & mask; r3 ^= col.w & mask;
            }
        }
        v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3;
    }
    U128 s{(uint64_t)v[0] | ((uint64_t)v[1] << 32), (uint64_t)v[2] | ((uint64_t)v[3] << 32)};

    float* plane = a.raw[c];
    const uint32_t row_end = min(row0 + kRowsPerChain, gh);
    for (uint32_t r = row0; r < row_end; ++r) {
        float* dst = plane + (size_t)(y0 + r) * a.width + x0;
        for (uint32_t bx = 0; bx < w16; ++bx) {
            const uint64_t ret = s.lo + s.hi;  // fill_batch: ret = s1 + s0 of the old state
            s = xs_step(s);
            // get_u32_bits: little-endian halves; NoiseGroup::new: (x >> 9) | 0x3f800000
            const uint32_t x = bx * 16 + lane * 2;
            if (x < gw) dst[x] = __uint_as_float(((uint32_t)ret >> 9) | 0x3f800000u);
            if (x + 1 < gw) dst[x + 1] = __uint_as_float(((uint32_t)(ret >> 32) >> 9) | 0x3f800000u);
        }
    }
}

struct NoiseApplyArgs {
    const float* raw[3];
    float* ch[3];            // X, Y, B (or the three colour channels of a non-XYB Modular frame)
    uint32_t stride;
    uint32_t width, height, group_dim;
    float lut[9];
    float corr_x, corr_b;
};

constexpr int kTW = 64, kTH = 4;

// block 64 x 4 outputs; LDS tile (64+4) x (4+4) x 3 of the mirrored raw noise
__global__ __launch_bounds__(kTW * kTH) void noise_apply_kernel(NoiseApplyArgs a) {
    __shared__ float tile[3][kTH + 4][kTW + 4 + 1];
    const int bx0 = blockIdx.x * kTW, by0 = blockIdx.y * kTH;
    const int tid = threadIdx.y * kTW + threadIdx.x;
    const int W = (int)a.width, H = (int)a.height;
    for (int i = tid; i < 3 * (kTH + 4) * (kTW + 4); i += kTW * kTH) {
        const int c = i / ((kTH + 4) * (kTW + 4));
        const int rem = i - c * (kTH + 4) * (kTW + 4);
        const int ty = rem / (kTW + 4), tx = rem - ty * (kTW + 4);
        const int gx = mirror_idx(bx0 + tx - 2, W), gy = mirror_idx(by0 + ty - 2, H);
        tile[c][ty][tx] = a.raw[c][(size_t)gy * W + gx];
    }
    __syncthreads();
    const int x = bx0 + threadIdx.x, y = by0 + threadIdx.y;
    if (x >= W || y >= H) return;

    // convolve_fill (noise.rs:295-308): `rows` is a 5-row ring buffer; buffer row r holds the
    // group-local image row yy in [y-2, y+2] with (yy + 2) % 5 == r, and the taps are summed in
    // buffer order.
    const int yl = y % (int)a.group_dim;
    float conv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int off = (((r - yl) % 5) + 5) % 5;  // yy = y - 2 + off
            const float* row = &tile[c][threadIdx.y + off][threadIdx.x];
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) sum += row[dx] * 0.16f;
        }
        conv[c] = sum - tile[c][threadIdx.y + 2][threadIdx.x + 2] * 4.0f;
    }

    // render_noise (noise.rs:52-86)
    float* px = a.ch[0] + (size_t)y * a.stride + x;
    float* py = a.ch[1] + (size_t)y * a.stride + x;
    float* pb = a.ch[2] + (size_t)y * a.stride + x;
    const float grid_x = *px, grid_y = *py;
    const float in_x = grid_x + grid_y;
    const float in_y = grid_y - grid_x;
    const float in_scaled_x = fmaxf(0.0f, in_x * 3.0f);
    const float in_scaled_y = fmaxf(0.0f, in_y * 3.0f);
    const int in_x_int = in_scaled_x >= 8.0f ? 7 : (int)in_scaled_x;  // `as usize` then .min(7)
    const int in_y_int = in_scaled_y >= 8.0f ? 7 : (int)in_scaled_y;
    const float in_x_frac = in_scaled_x - (float)in_x_int;
    const float in_y_frac = in_scaled_y - (float)in_y_int;
    const float sx = (a.lut[in_x_int + 1] - a.lut[in_x_int]) * in_x_frac + a.lut[in_x_int];
    const float sy = (a.lut[in_y_int + 1] - a.lut[in_y_int]) * in_y_frac + a.lut[in_y_int];
    const float nx = 0.22f * sx * (0.0078125f * conv[0] + 0.9921875f * conv[2]);
    const float ny = 0.22f * sy * (0.0078125f * conv[1] + 0.9921875f * conv[2]);
    *px = grid_x + (a.corr_x * (nx + ny) + nx - ny);
    *py = grid_y + (nx + ny);
    *pb = *pb + a.corr_b * (nx + ny);
}

}  // namespace

size_t noise_jump_table_bytes() { return jump_table_host().size() * 4; }
const void* noise_jump_table_host() { return jump_table_host().data(); }

// noise.rs:326-333: the group row above a 1-row bottom group asks that group for its row 1 and the
// reference panics (shared_subgrid.rs:117-124); such frames stay on the CPU path.
bool noise_geometry_unsupported(uint32_t height, uint32_t group_dim) {
    return height > group_dim && height % group_dim == 1;
}

void launch_noise(hipStream_t s, const JxlGpuNoiseParams& np, const void* jump_dev, float* const raw[3],
                  float* const ch[3], uint32_t stride, uint32_t width, uint32_t height, uint32_t group_dim,
                  float corr_x, float corr_b) {
    NoiseFillArgs fa;
    for (int c = 0; c < 3; ++c) fa.raw[c] = raw[c];
    fa.width = width; fa.height = height; fa.group_dim = group_dim;
    fa.groups_per_row = (width + group_dim - 1) / group_dim;
    const uint32_t group_rows = (height + group_dim - 1) / group_dim;
    fa.seed0 = ((uint64_t)np.visible_frames << 32) + (uint64_t)np.invisible_frames;  // rng_seed0, noise.rs:168-170
    fa.jump = static_cast<const uint4*>(jump_dev);
    const uint32_t chains = (group_dim + kRowsPerChain - 1) / kRowsPerChain;
    hipLaunchKernelGGL(noise_fill_kernel, dim3(fa.groups_per_row * group_rows, 3), dim3(chains * 8), 0, s, fa);

    NoiseApplyArgs aa;
    for (int c = 0; c < 3; ++c) { aa.raw[c] = raw[c]; aa.ch[c] = ch[c]; }
    aa.stride = stride; aa.width = width; aa.height = height; aa.group_dim = group_dim;
    for (int i = 0; i < 8; ++i) aa.lut[i] = np.lut[i];
    aa.lut[8] = np.lut[7];
    aa.corr_x = corr_x; aa.corr_b = corr_b;
    hipLaunchKernelGGL(noise_apply_kernel, dim3((width + kTW - 1) / kTW, (height + kTH - 1) / kTH), dim3(kTW, kTH), 0,
                       s, aa);
}
