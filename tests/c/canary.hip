// Pure-HIP canary: no libjxlgpu, no torch.  Exercises the runtime features libjxlgpu.so relies on — many small
// exact-size hipMallocs, non-blocking streams, blocking and async copies (1-D and 2-D, pageable and pinned),
// hipMemsetAsync, a kernel with > 64 KiB of dynamic LDS (hipFuncSetAttribute), events — with trivial,
// obviously in-bounds kernels.  Run as the first GPU process of a test session / smoke() to tell a faulting BOX
// from a faulting LIBRARY: if THIS program dies with "Memory access fault by GPU node", nothing of ours was loaded.
// Prints one line: "CANARY ok" (exit 0), "CANARY nodevice" (exit 3) or "CANARY fail: <why>" (exit 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void canary_kernel(unsigned* p, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 3u + i;
}

// 96 KiB of dynamic LDS: every thread writes its words, reads others' back
__global__ __launch_bounds__(256) void canary_lds_kernel(unsigned* out, unsigned words) {
    extern __shared__ unsigned lds[];
    for (unsigned i = threadIdx.x; i < words; i += 256) lds[i] = i ^ blockIdx.x;
    __syncthreads();
    unsigned acc = 0;
    for (unsigned i = threadIdx.x; i < words; i += 256) acc += lds[words - 1 - i] ^ blockIdx.x;
    atomicAdd(out + blockIdx.x, acc);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("CANARY fail: %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { printf("CANARY nodevice\n"); return 3; }
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    CK(hipSetDevice(0));
    hipStream_t s, s2;
    hipEvent_t ev;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int r = 0; r < rounds; ++r) {
        // the library's allocation pattern: many small exact-size buffers, then a few large ones
        const unsigned sizes[7] = {4, 16, 64, 1000, 4096, 1u << 16, 1u << 22};
        for (unsigned n : sizes) {
            unsigned* d = nullptr;
            CK(hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 4));
            unsigned* h = static_cast<unsigned*>(malloc((size_t)n * 4));
            for (unsigned i = 0; i < n; ++i) h[i] = 0x01010101u;
            if (r & 1) CK(hipMemsetAsync(d, 0x01, (size_t)n * 4, s));
            else CK(hipMemcpy(d, h, (size_t)n * 4, hipMemcpyHostToDevice));   // blocking, pageable source
            canary_kernel<<<(n + 255) / 256, 256, 0, s>>>(d, n);
            CK(hipGetLastError());
            memset(h, 0, (size_t)n * 4);
            CK(hipMemcpyAsync(h, d, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            for (unsigned i = 0; i < n; ++i)
                if (h[i] != 0x01010101u * 3u + i) { printf("CANARY fail: wrong value at %u of %u\n", i, n); return 1; }
            free(h);
            CK(hipFree(d));
        }
        // 2-D copies through a pinned staging buffer, cross-stream event ordering
        {
            const unsigned w = 264, hgt = 200, pitch = 272;
            unsigned *d = nullptr, *pin = nullptr;
            CK(hipMalloc(reinterpret_cast<void**>(&d), (size_t)pitch * hgt * 4));
            CK(hipHostMalloc(reinterpret_cast<void**>(&pin), (size_t)w * hgt * 4, hipHostMallocDefault));
            CK(hipMemsetAsync(d, 0x01, (size_t)pitch * hgt * 4, s));
            CK(hipEventRecord(ev, s));
            CK(hipStreamWaitEvent(s2, ev, 0));
            canary_kernel<<<(pitch * hgt + 255) / 256, 256, 0, s2>>>(d, pitch * hgt);
            CK(hipEventRecord(ev, s2));
            CK(hipStreamWaitEvent(s, ev, 0));
            CK(hipMemcpy2DAsync(pin, (size_t)w * 4, d, (size_t)pitch * 4, (size_t)w * 4, hgt, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            for (unsigned y = 0; y < hgt; ++y)
                for (unsigned x = 0; x < w; ++x)
                    if (pin[y * w + x] != 0x01010101u * 3u + y * pitch + x) { printf("CANARY fail: 2-D copy (%u,%u)\n", x, y); return 1; }
            CK(hipHostFree(pin));
            CK(hipFree(d));
        }
        // > 64 KiB of dynamic LDS
        {
            const unsigned words = 24576, blocks = 512;
            unsigned* d = nullptr;
            CK(hipMalloc(reinterpret_cast<void**>(&d), blocks * 4));
            CK(hipMemsetAsync(d, 0, blocks * 4, s));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&canary_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(words * 4)));
            canary_lds_kernel<<<blocks, 256, words * 4, s>>>(d, words);
            CK(hipGetLastError());
            static unsigned h[512];
            CK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            unsigned want = 0;
            for (unsigned i = 0; i < words; ++i) want += i;
            for (unsigned b = 0; b < blocks; ++b)
                if (h[b] != want) { printf("CANARY fail: LDS kernel block %u\n", b); return 1; }
            CK(hipFree(d));
        }
    }
    CK(hipEventDestroy(ev));
    CK(hipStreamDestroy(s));
    CK(hipStreamDestroy(s2));
    printf("CANARY ok\n");
    return 0;
}
