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
#include "transform_items.inc"

#include <mutex>

namespace {

__global__ __launch_bounds__(64) void transform_special_kernel(TransformArgs a, const uint4* __restrict__ entries,
                                                               uint32_t count) {
    __shared__ float lut[kLutWords];
    special_body<false>(a, entries, nullptr, count, lut, nullptr);
}

__global__ __launch_bounds__(64) void transform_special_batch_kernel(FrameBatch b) {
    __shared__ float lut[kLutWords];
    JXL_SET_TR_PRIO();
    const FrameDevC fd = (FrameDevC)b.f[blockIdx.y];
    const uint32_t count = fd->special_count;
    if (blockIdx.x * kSpecialPerWave >= count) return;
    const TransformArgs a = load_transform_args(fd);
    special_body<false>(a, fd->entries + fd->special_first, nullptr, count, lut, nullptr);
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
        if (pipe) transform_items_kernel<F, true><<<wgs, 192, bytes, s>>>(a, ct, entries, nullptr); \
        else transform_items_kernel<F, false><<<wgs, 192, bytes, s>>>(a, ct, entries, nullptr);     \
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
                                  uint32_t max_special, uint32_t mask) {
    if (!side) side = s;
#define LAUNCHB(F, ST)                                                                                       \
    if (max_wgs[F] && (mask >> F & 1u)) {                                                                   \
        set_lds_attr<F>();                                                                                  \
        transform_items_batch_kernel<F><<<dim3(max_wgs[F], n), 192, FamCfg<F>::WORDS * sizeof(float), ST>>>(b); \
    }
    LAUNCHB(3, side)
    LAUNCHB(2, side)
    if (max_special && (mask & 16u))
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
