// Issue-rate probe for packed vs scalar f32 VALU work on gfx950 (groundwork for DESIGN.md §8 item 1).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/pk_probe.hip -o /tmp/pk_probe
//   /tmp/pk_probe
//
// For 1, 2, 3, 4 and 8 waves per SIMD it times a loop of (a) scalar v_mul_f32 + v_add_f32 and
// (b) v_pk_mul_f32 + v_pk_add_f32 with CHAINS independent dependency chains per lane, and prints
// wave-instructions per ns per SIMD.  What it answers: does a packed instruction issue at the
// scalar rate (one per 4 cycles per SIMD = 0.6 per ns at 2.4 GHz), and how many independent chains /
// resident waves does it take to reach that rate — the packed streaming kernel
// (csrc/stream_pk.inc) runs one or two waves per SIMD and came out slower than the scalar one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ __launch_bounds__(64) void scalar_kernel(float* out, const float* in, int iters) {
    float v[CHAINS], m = in[0], a = in[1];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = in[2 + c] + threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) v[c] = v[c] * m + a;   // -ffp-contract=off: mul, add
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(64) void packed_kernel(float* out, const f2* in, int iters) {
    f2 v[CHAINS], m = in[0], a = in[1];   // constants from memory: both halves are real data
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = in[2 + c] + (float)threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) v[c] = v[c] * m + a;   // v_pk_mul_f32, v_pk_add_f32
    f2 s = {0, 0};
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}

template <typename K, typename In>
double time_kernel(K kern, float* out, const In* in, int waves, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, out, in, 16);  // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, out, in, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int CHAINS>
void run(float* out, float* in, int simds) {
    const int iters = 1 << 16;
    for (int wps : {1, 2, 3, 4, 8}) {
        const int waves = simds * wps;
        const double ms_s = time_kernel(scalar_kernel<CHAINS>, out, (const float*)in, waves, iters);
        const double ms_p = time_kernel(packed_kernel<CHAINS>, out, (const f2*)in, waves, iters);
        const double instr = 2.0 * CHAINS * iters * wps;  // wave-instructions per SIMD
        printf("chains %d  waves/SIMD %d   scalar %.3f instr/ns/SIMD   packed %.3f instr/ns/SIMD (= %.3f f32 op pairs)\n",
               CHAINS, wps, instr / (ms_s * 1e6), instr / (ms_p * 1e6), instr / (ms_p * 1e6));
    }
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 3; }
    const int simds = p.multiProcessorCount * 4;
    printf("%s: %d CUs, %d SIMDs, %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, simds, p.clockRate / 1000.0);
    float *out, *in;
    (void)hipMalloc(&out, sizeof(float) * 64 * simds * 8);
    (void)hipMalloc(&in, sizeof(float) * 64);
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + 1e-6f * i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<1>(out, in, simds);
    run<4>(out, in, simds);
    run<12>(out, in, simds);
    return 0;
}
