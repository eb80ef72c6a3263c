// Is the HIP virtual-memory API itself sound under churn (map / poison / use / unmap, same VAs coming back)?
// No libjxlgpu here: if THIS program sees corrupted buffers or dies, failures of the test suite under JXLGPU_GUARD
// after many frames are artefacts of the runtime's VMM path, not reads of uninitialised memory in our kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s (iter %d)\n", #x, hipGetErrorString(e_), it); return 1; } } while (0)
struct Rec { void* base; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; char* p; size_t user; };
__global__ void fill(unsigned* p, unsigned n, unsigned seed) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += seed + i; }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: unmap + release + free the VA; 1: unmap + release, keep the VA reserved; 2: leak everything
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; int it = -1;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t sizes[6] = {16, 4096 + 64, 300000, 1 << 20, 99532800 / 4, 52};
    long bad_total = 0;
    for (it = 0; it < 200; ++it) {
        std::vector<Rec> recs;
        for (size_t u : sizes) {
            Rec r; r.user = (u + 15) & ~(size_t)15; r.mapped = (r.user + gran - 1) / gran * gran; r.reserved = r.mapped + 2 * gran;
            CK(hipMemAddressReserve(&r.base, r.reserved, gran, nullptr, 0));
            CK(hipMemCreate(&r.h, r.mapped, &prop, 0));
            char* at = (char*)r.base + gran;
            CK(hipMemMap(at, r.mapped, 0, r.h, 0));
            hipMemAccessDesc ad; memset(&ad, 0, sizeof(ad)); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(at, r.mapped, &ad, 1));
            CK(hipMemset(at, 0xff, r.mapped)); CK(hipDeviceSynchronize());
            r.p = at + (r.mapped - r.user);
            recs.push_back(r);
        }
        for (Rec& r : recs) {
            CK(hipMemsetAsync(r.p, 0, r.user, s));
            const unsigned n = (unsigned)(r.user / 4);
            fill<<<(n + 255) / 256, 256, 0, s>>>((unsigned*)r.p, n, (unsigned)it);
        }
        for (Rec& r : recs) {
            std::vector<unsigned> h(r.user / 4);
            CK(hipMemcpyAsync(h.data(), r.p, r.user, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            long bad = 0;
            for (size_t i = 0; i < h.size(); ++i) bad += h[i] != (unsigned)it + (unsigned)i;
            if (bad) { printf("iter %d size %zu: %ld wrong words (first: %08x)\n", it, r.user, bad, h[0]); bad_total += bad; }
        }
        CK(hipStreamSynchronize(s));
        for (Rec& r : recs) {
            if (mode == 2) continue;
            CK(hipMemUnmap((char*)r.base + gran, r.mapped)); CK(hipMemRelease(r.h));
            if (mode == 0) CK(hipMemAddressFree(r.base, r.reserved));
        }
        if (it == 0 || it == 1 || it == 5) printf("iter %d done, wrong so far %ld\n", it, bad_total);
    }
    printf("vmm churn mode %d: 200 iterations, %ld wrong words\n", mode, bad_total);
    return bad_total != 0;
}
