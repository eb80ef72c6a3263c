// Probe: semantics of the wave-wide DPP shifts on gfx950 (used by the streaming post kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    int lane = threadIdx.x;
    int shr = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xf, 0xf, false);  // wave_shr:1
    int shl = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xf, 0xf, false);  // wave_shl:1
    float f = (float)lane * 1.5f;
    float fs = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x138, 0xf, 0xf, false));
    out[lane] = shr;
    out[64 + lane] = shl;
    out[128 + lane] = (int)(fs * 2.0f);
}
int main() {
    int* d; int h[192];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1 ->"); for (int i : {0, 1, 2, 15, 16, 17, 31, 32, 33, 62, 63}) printf(" [%d]=%d", i, h[i]);
    printf("\nwave_shl:1 ->"); for (int i : {0, 1, 2, 15, 16, 17, 31, 32, 33, 62, 63}) printf(" [%d]=%d", i, h[64 + i]);
    printf("\nfloat shr  ->"); for (int i : {0, 1, 2, 16, 32, 63}) printf(" [%d]=%d", i, h[128 + i]);
    printf("\n");
    return 0;
}
