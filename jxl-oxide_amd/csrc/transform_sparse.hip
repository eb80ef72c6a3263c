// V4-V8 for JXLGPU_COEFF_GROUPED frames: the transform kernels of transform_kernels.hip fed by the
// decoder's per-varblock non-zero lists (jxl-vardct/src/hf_coeff.rs:188-254: `non_zeros`, then
// (dx, dy, coeff) triples) instead of dense coefficient cells.  A wave zeroes its LDS tile, scatters
// the lists of its channel into it (64 / NBI lanes per varblock) and reads its rows from there; the
// rest of the work item is the dense kernels' code (transform_items.inc), so the arithmetic — and
// the result, bit for bit — is the same.  Why: a d1 frame has ~5 % non-zero coefficients; the dense
// layout made every decoded frame pay a device-side layout pass (retile, or zero-fill + scatter of
// 99.5 MB at 4K) and made V4 read 12 B per pixel of mostly zeros.  Here the input is 4 B per
// non-zero coefficient and nothing is built in HBM first.
#include "common.h"
#include "transform_items.inc"

#include <mutex>

namespace {

__global__ __launch_bounds__(64) void transform_special_sparse_kernel(TransformArgs a, const uint4* __restrict__ entries,
                                                                      const uint32_t* __restrict__ nzc, uint32_t count) {
    __shared__ __attribute__((aligned(16))) int tile[kSpecialTileWords];
    special_body<true>(a, entries, nzc, count, nullptr, tile);
}

__global__ __launch_bounds__(64) void transform_special_sparse_batch_kernel(FrameBatch b) {
    __shared__ __attribute__((aligned(16))) int tile[kSpecialTileWords];
    JXL_SET_TR_PRIO();
    const FrameDevC fd = (FrameDevC)b.f[blockIdx.y];
    const uint32_t count = fd->special_count;
    if (blockIdx.x * kSpecialPerWave >= count) return;
    const TransformArgs a = load_transform_args(fd);
    special_body<true>(a, fd->entries + fd->special_first, fd->nzc + fd->special_first, count, nullptr, tile);
}

// Fallback: expand the lists into the dense cell-tiled coefficients (zeroed by the caller).  One wave
// per varblock, its three lists back to back.
template <bool ACC>
__global__ __launch_bounds__(256) void grouped_to_dense_kernel(const uint4* __restrict__ entries,
                                                               const uint32_t* __restrict__ nzc, uint32_t n_entries,
                                                               const uint32_t* __restrict__ nz, uint32_t w8,
                                                               int32_t* __restrict__ coeff) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (i >= n_entries) return;
    const uint4 e = entries[i];
    const uint32_t cyx = nzc[i];
    const uint32_t cnt[3] = {cyx & 0xffffu, cyx >> 16, e.y >> 16};  // decode order: Y, X, B
    const uint32_t chan[3] = {1, 0, 2};
    const uint32_t px0 = (e.x & 0xffffu) * 8, py0 = (e.x >> 16) * 8;
    // TransformType -> (bw, bh) in cells, jxl-vardct/src/dct_select.rs:52-76
    static constexpr uint8_t kCells[27][2] = {{1, 1}, {1, 1}, {1, 1}, {1, 1}, {2, 2}, {4, 4}, {1, 2}, {2, 1}, {1, 4},
                                              {4, 1}, {2, 4}, {4, 2}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {1, 1}, {1, 1},
                                              {8, 8}, {4, 8}, {8, 4}, {16, 16}, {8, 16}, {16, 8}, {32, 32}, {16, 32}, {32, 16}};
    const uint32_t t = min(e.y & 0xffffu, 26u);
    const uint32_t bw = kCells[t][0], bh = kCells[t][1];
    uint32_t off = e.w;
    for (int s = 0; s < 3; ++s) {
        for (uint32_t k = lane; k < cnt[s]; k += 64) {
            const uint32_t w = nz[off + k];
            const uint32_t dx = w & 255u, dy = (w >> 8) & 255u;
            if (dx < bw * 8 && dy < bh * 8) {  // positions outside the varblock are ignored, as in the list-fed kernels
                int32_t* q = &coeff[coeff_tiled_index(px0 + dx, py0 + dy, chan[s], w8)];
                // a later pass of a progressive frame adds to what the earlier ones left (a position appears at most
                // once per pass and channel: plain read-modify-write, launches are ordered by the stream)
                if (ACC) *q += (int32_t)w >> 16;
                else *q = (int32_t)w >> 16;
            }
        }
        off += cnt[s];
    }
}

template <int F>
void set_lds_attr_sparse() {
    constexpr size_t bytes = FamCfg<F>::WORDS_S * sizeof(float);
    if constexpr (bytes > 65536) {
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&transform_items_kernel<F, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&transform_items_batch_kernel<F, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        });
    }
}

}  // namespace

hipError_t launch_transform_items_sparse(hipStream_t s, int family, const TransformArgs& a, const uint4* entries,
                                         const uint32_t* nzc, const uint32_t class_first[CLS_COUNT],
                                         const uint32_t list_count[CLS_COUNT], uint32_t num_cus) {
    if (family < 0 || family > 3) return hipErrorInvalidValue;
    ClassTable ct;
    build_class_table(family, class_first, list_count, num_cus, 0, &ct);  // one workgroup per work item
    const uint32_t wgs = ct.wg_begin[ct.n_classes];
    if (!ct.n_classes || !wgs) return hipSuccess;
#define LAUNCH(F)                                                                                   \
    do {                                                                                            \
        constexpr size_t bytes = FamCfg<F>::WORDS_S * sizeof(float);                                \
        set_lds_attr_sparse<F>();                                                                   \
        transform_items_kernel<F, false, true><<<wgs, 192, bytes, s>>>(a, ct, entries, nzc);        \
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

void launch_transform_special_sparse(hipStream_t s, const TransformArgs& a, const uint4* entries, const uint32_t* nzc,
                                     uint32_t count) {
    if (count == 0) return;
    transform_special_sparse_kernel<<<ceil_div(count, kSpecialPerWave), 64, 0, s>>>(a, entries, nzc, count);
}

hipError_t launch_transform_batch_sparse(hipStream_t s, hipStream_t side, const FrameBatch& b, uint32_t n,
                                         const uint32_t max_wgs[4], uint32_t max_special, uint32_t mask) {
    if (!side) side = s;
#define LAUNCHB(F, ST)                                                                                              \
    if (max_wgs[F] && (mask >> F & 1u)) {                                                                          \
        set_lds_attr_sparse<F>();                                                                                  \
        transform_items_batch_kernel<F, true><<<dim3(max_wgs[F], n), 192, FamCfg<F>::WORDS_S * sizeof(float), ST>>>(b); \
    }
    LAUNCHB(3, side)
    LAUNCHB(2, side)
    if (max_special && (mask & 16u))
        transform_special_sparse_batch_kernel<<<dim3(ceil_div(max_special, kSpecialPerWave), n), 64, 0, side>>>(b);
    LAUNCHB(1, s)
    LAUNCHB(0, s)
#undef LAUNCHB
    return hipGetLastError();
}

void launch_grouped_to_dense(hipStream_t s, const uint4* entries, const uint32_t* nzc, uint32_t n_entries,
                             const uint32_t* nz, uint32_t w8, int32_t* coeff, bool accumulate) {
    if (!n_entries) return;
    if (accumulate) grouped_to_dense_kernel<true><<<ceil_div(n_entries, 4), 256, 0, s>>>(entries, nzc, n_entries, nz, w8, coeff);
    else grouped_to_dense_kernel<false><<<ceil_div(n_entries, 4), 256, 0, s>>>(entries, nzc, n_entries, nz, w8, coeff);
}
