// Latency of DEPENDENT instructions on gfx950 (round 6): what one step of a serial chain costs when nothing else
// can be issued between an instruction and its consumer — the self-correcting predictor's situation
// (csrc/modular.hip, predict_lanes_narrow_kernel: one chain per wave).
//
//   hipcc -O2 --offload-arch=gfx950 tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
//
// Every kernel is a loop of 32 instructions of one form in which each instruction reads the result of the one before.
// W waves per SIMD (W = 1, 2, 4) on every SIMD of the chip; printed: nanoseconds and cycles (at the nominal clock) per
// instruction of ONE wave's chain.  If W = 2 costs the same per chain as W = 1 the chain is latency-bound and a
// second wave is free; if it doubles, the form is issue-bound already with one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define R4(X) X X X X
#define R32(X) R4(R4(X)) R4(R4(X))

#define CHAIN_KERNEL(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(64) void NAME(uint32_t* out, const uint32_t* in, int iters) {     \
        __shared__ uint32_t lds[64 * 4];                                                           \
        for (int i = 0; i < 4; ++i) lds[threadIdx.x * 4 + i] = threadIdx.x * 16;                   \
        __syncthreads();                                                                           \
        uint32_t a = in[0] + threadIdx.x * 16, b = in[1], c = in[2];                               \
        uint32_t la = threadIdx.x * 16;                                                            \
        for (int i = 0; i < iters; ++i) {                                                          \
            asm volatile(R32(ASM) : "+v"(a), "+v"(la) : "v"(b), "v"(c) : "vcc", "memory");         \
        }                                                                                          \
        out[blockIdx.x * 64 + threadIdx.x] = a + la;                                               \
    }

CHAIN_KERNEL(k_add_u32, "v_add_u32 %0, %0, %2\n")
CHAIN_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %2, %3\n")
CHAIN_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %2\n")
CHAIN_KERNEL(k_mul_hi, "v_mul_hi_i32 %0, %0, %2\n")
CHAIN_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %2, %3\n")
CHAIN_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %2\n")
CHAIN_KERNEL(k_ffbh, "v_ffbh_u32 %0, %0\n")
CHAIN_KERNEL(k_lshr_v, "v_lshrrev_b32 %0, %2, %0\n")
CHAIN_KERNEL(k_min_i32, "v_min_i32 %0, %0, %2\n")
CHAIN_KERNEL(k_med3, "v_med3_i32 %0, %0, %2, %3\n")
CHAIN_KERNEL(k_cmp_cnd, "v_cmp_lt_i32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %3, vcc\n")
CHAIN_KERNEL(k_dpp_wshr, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
CHAIN_KERNEL(k_dpp_rshr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
// LDS: a pointer chase (every read's address is the previous read's data: lds[lane * 4] holds lane * 16)
CHAIN_KERNEL(k_ds_read, "ds_read_b32 %1, %1\ns_waitcnt lgkmcnt(0)\n")
// LDS write then read of the same word (the ring hand-over of the predictor kernels)
CHAIN_KERNEL(k_ds_wr_rd, "ds_write_b32 %1, %1\nds_read_b32 %1, %1\ns_waitcnt lgkmcnt(0)\n")
// four independent LDS reads behind one wait (what the top of a predictor step issues), then a dependent add
CHAIN_KERNEL(k_ds_read4, "ds_read_b32 %0, %1\nds_read_b32 %0, %1 offset:4\nds_read_b32 %0, %1 offset:8\nds_read_b32 %0, %1 offset:12\ns_waitcnt lgkmcnt(0)\nv_and_b32 %1, 0x3f0, %0\n")
// ds_bpermute: the cross-lane read that needs no LDS storage
CHAIN_KERNEL(k_bpermute, "ds_bpermute_b32 %1, %1, %1\ns_waitcnt lgkmcnt(0)\nv_and_b32 %1, 0xfc, %1\n")

struct Entry { const char* name; void (*fn)(uint32_t*, const uint32_t*, int); int per_rep; };

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4;
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, nominal %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, ghz * 1e3);
    uint32_t *out, *in;
    hipMalloc(&out, (size_t)simds * 8 * 64 * 4);
    hipMalloc(&in, 64);
    uint32_t h[16] = {0, 1, 3};
    hipMemcpy(in, h, 64, hipMemcpyHostToDevice);
    const Entry es[] = {
        {"v_add_u32", k_add_u32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_i32", k_mul_hi, 1},
        {"v_mad_u32_u24", k_mad_u24, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_lshrrev_b32 (vgpr)", k_lshr_v, 1},
        {"v_min_i32", k_min_i32, 1}, {"v_med3_i32", k_med3, 1}, {"v_cmp + v_cndmask (pair)", k_cmp_cnd, 1},
        {"v_mov_b32_dpp wave_shr:1", k_dpp_wshr, 1}, {"v_mov_b32_dpp row_shr:1", k_dpp_rshr, 1},
        {"ds_read_b32 (pointer chase)", k_ds_read, 1}, {"ds_write + ds_read same word", k_ds_wr_rd, 1},
        {"4 ds_read + wait + v_and", k_ds_read4, 1}, {"ds_bpermute + wait + v_and", k_bpermute, 1},
    };
    printf("%-34s %10s %10s %10s   (cycles per dependent instruction / group, one wave's chain)\n", "form", "1 w/SIMD", "2", "4");
    const int iters = 4000;
    for (const Entry& e : es) {
        printf("%-34s", e.name);
        for (int w : {1, 2, 4}) {
            // one workgroup of 64 per wave; simds * w workgroups: the dispatcher spreads them round-robin
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            e.fn<<<simds * w, 64>>>(out, in, 100);
            hipDeviceSynchronize();
            float best = 1e9f;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(a);
                e.fn<<<simds * w, 64>>>(out, in, iters);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double ns = best * 1e6 / ((double)iters * 32);
            printf(" %10.1f", ns * ghz);
            hipEventDestroy(a);
            hipEventDestroy(b);
        }
        printf("\n");
    }
    return 0;
}
