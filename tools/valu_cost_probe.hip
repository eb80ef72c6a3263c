// Issue cost of the VALU instruction forms the post / transform kernels are made of, on gfx950 (round 6).
//
//   hipcc -O2 --offload-arch=gfx950 tools/valu_cost_probe.hip -o /tmp/valu_cost_probe && /tmp/valu_cost_probe
//
// Every kernel is a loop of 64 instructions of ONE form on 8 independent accumulators (a dependent pair is 8
// instructions apart), W waves per SIMD on every SIMD of the chip.  Printed: SIMD cycles per wave-instruction RELATIVE to
// a pure v_fma_f32 run in the same column (= 4.00), for W = 1, 2, 4, 8, and the absolute v_fma_f32 rate at the end.
// What it found (profiles/r06_valu_cost_probe.txt): plain f32 / logic instructions on VGPR (or literal) operands —
// v_fma / v_add / v_mul / v_and — issue at ~2.4 cycles with >= 4 waves per SIMD (the guide's "2 cycles"); packed f32,
// every DPP form, an SGPR operand, v_cndmask, v_cmp, v_bfi, 3-operand integer forms and conversions cost ~4.1 cycles
// (1.75 x); v_rcp / v_sqrt ~8; ONE wave per SIMD issues at most one instruction per ~5 cycles whatever the form.  So
// a packed instruction buys nothing over two plain ones in issue cycles (it halves the instruction COUNT, which is what
// a kernel of two waves per SIMD needs), and the 4-cycle peak bench.py's roofline_valu uses is the packed / DPP cost,
// not a universal one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// scalar-f32 forms: accumulators a0..a7 (VGPR), operands b, c (VGPR), s (SGPR)
#define KERNEL_S(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(64) void NAME(float* out, const float* in, int iters) {            \
        float a0 = in[0] + threadIdx.x, a1 = in[1], a2 = in[2], a3 = in[3], a4 = in[4], a5 = in[5], \
              a6 = in[6], a7 = in[7];                                                               \
        float b = in[8], c = in[9];                                                                 \
        float s = in[10];                                                                           \
        s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));  \
        for (int i = 0; i < iters; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                         \
                asm volatile(ASM(a0) ASM(a1) ASM(a2) ASM(a3) ASM(a4) ASM(a5) ASM(a6) ASM(a7)        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c), "s"(s)                                               \
                             : "vcc", "s20", "s21", "s22", "s23", "v200", "v201");                                  \
            }                                                                                       \
        }                                                                                           \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                 \
    }
// inside ASM(x): %N of the accumulator is spelled by position: a0 = %0 ... a7 = %7, b = %8, c = %9, s = %10
#define A_(x) _IDX_##x
#define _IDX_a0 "%0"
#define _IDX_a1 "%1"
#define _IDX_a2 "%2"
#define _IDX_a3 "%3"
#define _IDX_a4 "%4"
#define _IDX_a5 "%5"
#define _IDX_a6 "%6"
#define _IDX_a7 "%7"

#define I_FMA(x) "v_fma_f32 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_ADD(x) "v_add_f32 " A_(x) ", " A_(x) ", %8\n"
#define I_MUL(x) "v_mul_f32 " A_(x) ", " A_(x) ", %8\n"
#define I_ADD_SGPR(x) "v_add_f32 " A_(x) ", %10, " A_(x) "\n"
#define I_ADD_WSHR(x) "v_add_f32_dpp " A_(x) ", " A_(x) ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADD_WSHL(x) "v_add_f32_dpp " A_(x) ", " A_(x) ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADD_RSHR(x) "v_add_f32_dpp " A_(x) ", " A_(x) ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADD_QP(x) "v_add_f32_dpp " A_(x) ", " A_(x) ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_MOV_WSHR(x) "v_mov_b32_dpp " A_(x) ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_RCP(x) "v_rcp_f32 " A_(x) ", " A_(x) "\n"
#define I_SQRT(x) "v_sqrt_f32 " A_(x) ", " A_(x) "\n"
#define I_AND(x) "v_and_b32 " A_(x) ", 0x7fffffff, " A_(x) "\n"
#define I_BFI(x) "v_bfi_b32 " A_(x) ", %8, " A_(x) ", %9\n"
#define I_MAX3(x) "v_max3_f32 " A_(x) ", |" A_(x) "|, |%8|, |%9|\n"
#define I_CNDMASK_VCC(x) "v_cndmask_b32 " A_(x) ", " A_(x) ", %8, vcc\n"
#define I_CNDMASK_SGPR(x) "v_cndmask_b32_e64 " A_(x) ", " A_(x) ", %8, s[20:21]\n"
#define I_CMP_VCC(x) "v_cmp_lt_f32 vcc, " A_(x) ", %8\n"
#define I_CMP_SGPR(x) "v_cmp_lt_f32_e64 s[22:23], " A_(x) ", %8\n"
#define I_CMP_CND(x) "v_cmp_lt_f32 vcc, " A_(x) ", %8\nv_cndmask_b32 " A_(x) ", " A_(x) ", %9, vcc\n"
#define I_READFIRST(x) "v_readfirstlane_b32 s20, " A_(x) "\n"
#define I_CVT(x) "v_cvt_f32_i32 " A_(x) ", " A_(x) "\n"
#define I_LSHL_ADD(x) "v_lshl_add_u32 " A_(x) ", " A_(x) ", 2, %8\n"
#define I_MAD_U32(x) "v_mad_u32_u24 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_MUL_LO(x) "v_mul_lo_u32 " A_(x) ", " A_(x) ", %8\n"
#define I_DIV_SCALE(x) "v_div_scale_f32 " A_(x) ", vcc, " A_(x) ", %8, " A_(x) "\n"
#define I_DIV_FIXUP(x) "v_div_fixup_f32 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_FMA_NOP(x) "v_fma_f32 " A_(x) ", " A_(x) ", %8, %9\ns_nop 1\n"
#define I_ADD_DPP_AFTER(x) "v_add_f32_dpp " A_(x) ", " A_(x) ", " A_(x) " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"

#define I_ADD_U32(x) "v_add_u32 " A_(x) ", " A_(x) ", %8\n"
#define I_SUB_U32(x) "v_sub_u32 " A_(x) ", " A_(x) ", %8\n"
#define I_ADD3_U32(x) "v_add3_u32 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_LSHL(x) "v_lshlrev_b32 " A_(x) ", 3, " A_(x) "\n"
#define I_LSHR(x) "v_lshrrev_b32 " A_(x) ", 3, " A_(x) "\n"
#define I_ASHR(x) "v_ashrrev_i32 " A_(x) ", 3, " A_(x) "\n"
#define I_LSHL_V(x) "v_lshlrev_b32 " A_(x) ", %8, " A_(x) "\n"
#define I_OR(x) "v_or_b32 " A_(x) ", " A_(x) ", %8\n"
#define I_XOR(x) "v_xor_b32 " A_(x) ", " A_(x) ", %8\n"
#define I_MIN_I32(x) "v_min_i32 " A_(x) ", " A_(x) ", %8\n"
#define I_MAX_U32(x) "v_max_u32 " A_(x) ", " A_(x) ", %8\n"
#define I_MED3_I32(x) "v_med3_i32 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_MUL_U24(x) "v_mul_u32_u24 " A_(x) ", " A_(x) ", %8\n"
#define I_MUL_I24(x) "v_mul_i32_i24 " A_(x) ", " A_(x) ", %8\n"
#define I_MAD_I24(x) "v_mad_i32_i24 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_MUL_HI(x) "v_mul_hi_u32 " A_(x) ", " A_(x) ", %8\n"
#define I_MOV(x) "v_mov_b32 " A_(x) ", %8\n"
#define I_FFBH(x) "v_ffbh_u32 " A_(x) ", " A_(x) "\n"
#define I_SUB_F32(x) "v_sub_f32 " A_(x) ", " A_(x) ", %8\n"
#define I_MAX_F32(x) "v_max_f32 " A_(x) ", " A_(x) ", %8\n"
#define I_MIN_F32(x) "v_min_f32 " A_(x) ", " A_(x) ", %8\n"
#define I_MAX_F32_ABS(x) "v_max_f32 " A_(x) ", |" A_(x) "|, |%8|\n"
#define I_FMAC(x) "v_fmac_f32 " A_(x) ", %8, %9\n"
#define I_FMA_LIT(x) "v_fma_f32 " A_(x) ", " A_(x) ", %8, 0.5\n"
#define I_MUL_LIT(x) "v_mul_f32 " A_(x) ", 0x3f9d70a4, " A_(x) "\n"
#define I_RSQ(x) "v_rsq_f32 " A_(x) ", " A_(x) "\n"
#define I_CVT_I32_F32(x) "v_cvt_i32_f32 " A_(x) ", " A_(x) "\n"
#define I_BFE(x) "v_bfe_u32 " A_(x) ", " A_(x) ", 4, 8\n"
#define I_AND_OR(x) "v_and_or_b32 " A_(x) ", " A_(x) ", %8, %9\n"
#define I_MAD_U64(x) "v_mad_u64_u32 v[200:201], s[22:23], " A_(x) ", %8, v[200:201]\n"
#define I_ADDC(x) "v_add_co_u32 " A_(x) ", vcc, " A_(x) ", %8\nv_addc_co_u32 " A_(x) ", vcc, " A_(x) ", %9, vcc\n"
KERNEL_S(k_fma, I_FMA)
KERNEL_S(k_add, I_ADD)
KERNEL_S(k_mul, I_MUL)
KERNEL_S(k_add_sgpr, I_ADD_SGPR)
KERNEL_S(k_add_wshr, I_ADD_WSHR)
KERNEL_S(k_add_wshl, I_ADD_WSHL)
KERNEL_S(k_add_rshr, I_ADD_RSHR)
KERNEL_S(k_add_qp, I_ADD_QP)
KERNEL_S(k_mov_wshr, I_MOV_WSHR)
KERNEL_S(k_rcp, I_RCP)
KERNEL_S(k_sqrt, I_SQRT)
KERNEL_S(k_and, I_AND)
KERNEL_S(k_bfi, I_BFI)
KERNEL_S(k_max3, I_MAX3)
KERNEL_S(k_cndmask_vcc, I_CNDMASK_VCC)
KERNEL_S(k_cndmask_sgpr, I_CNDMASK_SGPR)
KERNEL_S(k_cmp_vcc, I_CMP_VCC)
KERNEL_S(k_cmp_sgpr, I_CMP_SGPR)
KERNEL_S(k_cmp_cnd, I_CMP_CND)
KERNEL_S(k_readfirst, I_READFIRST)
KERNEL_S(k_cvt, I_CVT)
KERNEL_S(k_lshl_add, I_LSHL_ADD)
KERNEL_S(k_mad_u24, I_MAD_U32)
KERNEL_S(k_mul_lo, I_MUL_LO)
KERNEL_S(k_div_scale, I_DIV_SCALE)
KERNEL_S(k_div_fixup, I_DIV_FIXUP)
KERNEL_S(k_fma_nop1, I_FMA_NOP)
KERNEL_S(k_add_dpp_self, I_ADD_DPP_AFTER)
KERNEL_S(k_add_u32, I_ADD_U32)
KERNEL_S(k_sub_u32, I_SUB_U32)
KERNEL_S(k_add3_u32, I_ADD3_U32)
KERNEL_S(k_lshl, I_LSHL)
KERNEL_S(k_lshr, I_LSHR)
KERNEL_S(k_ashr, I_ASHR)
KERNEL_S(k_lshl_v, I_LSHL_V)
KERNEL_S(k_or, I_OR)
KERNEL_S(k_xor, I_XOR)
KERNEL_S(k_min_i32, I_MIN_I32)
KERNEL_S(k_max_u32, I_MAX_U32)
KERNEL_S(k_med3_i32, I_MED3_I32)
KERNEL_S(k_mul_u24, I_MUL_U24)
KERNEL_S(k_mul_i24, I_MUL_I24)
KERNEL_S(k_mad_i24, I_MAD_I24)
KERNEL_S(k_mul_hi, I_MUL_HI)
KERNEL_S(k_mov, I_MOV)
KERNEL_S(k_ffbh, I_FFBH)
KERNEL_S(k_sub_f32, I_SUB_F32)
KERNEL_S(k_max_f32, I_MAX_F32)
KERNEL_S(k_min_f32, I_MIN_F32)
KERNEL_S(k_max_f32_abs, I_MAX_F32_ABS)
KERNEL_S(k_fmac, I_FMAC)
KERNEL_S(k_fma_lit, I_FMA_LIT)
KERNEL_S(k_mul_lit, I_MUL_LIT)
KERNEL_S(k_rsq, I_RSQ)
KERNEL_S(k_cvt_i32_f32, I_CVT_I32_F32)
KERNEL_S(k_bfe, I_BFE)
KERNEL_S(k_and_or, I_AND_OR)
KERNEL_S(k_addc, I_ADDC)

// packed forms: accumulators are VGPR pairs, b / c VGPR pairs, s an SGPR pair
#define KERNEL_P(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(64) void NAME(float* out, const float* in, int iters) {            \
        f2 a0 = {in[0] + threadIdx.x, in[1]}, a1 = {in[1], in[2]}, a2 = {in[2], in[3]}, a3 = {in[3], in[4]}, \
           a4 = {in[4], in[5]}, a5 = {in[5], in[6]}, a6 = {in[6], in[7]}, a7 = {in[7], in[8]};      \
        f2 b = {in[8], in[9]}, c = {in[9], in[10]};                                                 \
        f2 s = {in[10], in[11]};                                                                    \
        s.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s.x))); \
        s.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s.y))); \
        for (int i = 0; i < iters; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                         \
                asm volatile(ASM(a0) ASM(a1) ASM(a2) ASM(a3) ASM(a4) ASM(a5) ASM(a6) ASM(a7)        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c), "s"(s));                                             \
            }                                                                                       \
        }                                                                                           \
        const f2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                         \
        out[blockIdx.x * 64 + threadIdx.x] = t.x + t.y;                                             \
    }
#define P_FMA(x) "v_pk_fma_f32 " A_(x) ", " A_(x) ", %8, %9\n"
#define P_ADD(x) "v_pk_add_f32 " A_(x) ", " A_(x) ", %8\n"
#define P_MUL(x) "v_pk_mul_f32 " A_(x) ", " A_(x) ", %8\n"
#define P_ADD_CLAMP(x) "v_pk_add_f32 " A_(x) ", " A_(x) ", %8 clamp\n"
#define P_MUL_SGPR(x) "v_pk_mul_f32 " A_(x) ", " A_(x) ", %10\n"
#define P_FMA_NEG(x) "v_pk_fma_f32 " A_(x) ", " A_(x) ", %8, %9 neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
#define P_MOV64(x) "v_mov_b64 " A_(x) ", %8\n"
KERNEL_P(k_pk_fma, P_FMA)
KERNEL_P(k_pk_add, P_ADD)
KERNEL_P(k_pk_mul, P_MUL)
KERNEL_P(k_pk_add_clamp, P_ADD_CLAMP)
KERNEL_P(k_pk_mul_sgpr, P_MUL_SGPR)
KERNEL_P(k_pk_fma_neg, P_FMA_NEG)
KERNEL_P(k_mov_b64, P_MOV64)

// LDS forms (issue cost of a ds_read beside nothing else; the address is the lane's own word)
__global__ __launch_bounds__(64) void k_ds_read_b32(float* out, const float* in, int iters) {
    __shared__ float lds[64 * 8];
    for (int i = 0; i < 8; ++i) lds[i * 64 + threadIdx.x] = in[i & 7];
    __syncthreads();
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds + threadIdx.x);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8 offset:256\nds_read_b32 %2, %8 offset:512\nds_read_b32 %3, %8 offset:768\n"
                         "ds_read_b32 %4, %8 offset:1024\nds_read_b32 %5, %8 offset:1280\nds_read_b32 %6, %8 offset:1536\nds_read_b32 %7, %8 offset:1792\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                         : "v"(addr));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(64) void k_ds_read_b64(float* out, const float* in, int iters) {
    __shared__ float lds[128 * 8];
    for (int i = 0; i < 16; ++i) lds[i * 64 + threadIdx.x] = in[i & 7];
    __syncthreads();
    f2 a0 = {0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds + 2 * threadIdx.x);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:512\nds_read_b64 %2, %8 offset:1024\nds_read_b64 %3, %8 offset:1536\n"
                         "ds_read_b64 %4, %8 offset:2048\nds_read_b64 %5, %8 offset:2560\nds_read_b64 %6, %8 offset:3072\nds_read_b64 %7, %8 offset:3584\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                         : "v"(addr));
        }
    }
    const f2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 64 + threadIdx.x] = t.x + t.y;
}

typedef void (*kern_t)(float*, const float*, int);

static double time_kernel(kern_t kern, float* out, const float* in, int waves, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, out, in, 8);  // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, out, in, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms;
}

struct Entry { const char* name; kern_t k; int instr_per_slot; };

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 3; }
    const int simds = p.multiProcessorCount * 4;
    printf("%s: %d CUs, %d SIMDs, nominal %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, simds, p.clockRate / 1000.0);
    float *out, *in;
    (void)hipMalloc(&out, sizeof(float) * 64 * simds * 8);
    (void)hipMalloc(&in, sizeof(float) * 64);
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + 1e-3f * i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const Entry tab[] = {
        {"v_fma_f32", k_fma, 1}, {"v_add_f32", k_add, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32 (sgpr src)", k_add_sgpr, 1},
        {"v_pk_fma_f32", k_pk_fma, 1}, {"v_pk_add_f32", k_pk_add, 1}, {"v_pk_mul_f32", k_pk_mul, 1},
        {"v_pk_add_f32 clamp", k_pk_add_clamp, 1}, {"v_pk_mul_f32 (sgpr pair)", k_pk_mul_sgpr, 1}, {"v_pk_fma_f32 neg", k_pk_fma_neg, 1},
        {"v_mov_b64", k_mov_b64, 1},
        {"v_add_f32_dpp wave_shr:1", k_add_wshr, 1}, {"v_add_f32_dpp wave_shl:1", k_add_wshl, 1}, {"v_add_f32_dpp row_shr:1", k_add_rshr, 1},
        {"v_add_f32_dpp quad_perm", k_add_qp, 1}, {"v_mov_b32_dpp wave_shr:1", k_mov_wshr, 1},
        {"v_add_f32_dpp wave_shr:1 (src0 = dst)", k_add_dpp_self, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_and_b32", k_and, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_max3_f32 |abs|", k_max3, 1},
        {"v_cndmask_b32 (vcc)", k_cndmask_vcc, 1}, {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask_sgpr, 1},
        {"v_cmp_lt_f32 -> vcc", k_cmp_vcc, 1}, {"v_cmp_lt_f32_e64 -> sgpr pair", k_cmp_sgpr, 1}, {"v_cmp + v_cndmask (per pair)", k_cmp_cnd, 2},
        {"v_readfirstlane_b32", k_readfirst, 1}, {"v_cvt_f32_i32", k_cvt, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
        {"v_mad_u32_u24", k_mad_u24, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_div_scale_f32", k_div_scale, 1}, {"v_div_fixup_f32", k_div_fixup, 1},
        {"v_sub_f32", k_sub_f32, 1}, {"v_max_f32", k_max_f32, 1}, {"v_min_f32", k_min_f32, 1}, {"v_max_f32 |a|,|b| (e64)", k_max_f32_abs, 1},
        {"v_fmac_f32", k_fmac, 1}, {"v_fma_f32 (inline const 0.5)", k_fma_lit, 1}, {"v_mul_f32 (32-bit literal)", k_mul_lit, 1},
        {"v_rsq_f32", k_rsq, 1}, {"v_cvt_i32_f32", k_cvt_i32_f32, 1}, {"v_mov_b32", k_mov, 1},
        {"v_add_u32", k_add_u32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_add_co_u32 + v_addc_co_u32 (per pair)", k_addc, 2},
        {"v_lshlrev_b32 (imm)", k_lshl, 1}, {"v_lshrrev_b32 (imm)", k_lshr, 1}, {"v_ashrrev_i32 (imm)", k_ashr, 1}, {"v_lshlrev_b32 (vgpr)", k_lshl_v, 1},
        {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_min_i32", k_min_i32, 1}, {"v_max_u32", k_max_u32, 1}, {"v_med3_i32", k_med3_i32, 1}, {"v_ffbh_u32", k_ffbh, 1},
        {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mul_i32_i24", k_mul_i24, 1}, {"v_mad_i32_i24", k_mad_i24, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
        {"v_fma_f32 + s_nop 1 (per pair)", k_fma_nop1, 1},
        {"ds_read_b32 (8 in flight)", k_ds_read_b32, 1}, {"ds_read_b64 (8 in flight)", k_ds_read_b64, 1},
    };
    const int iters = 1 << 13;
    printf("%-40s %8s %8s %8s %8s   (SIMD cycles per wave-instruction, relative to v_fma_f32 = 4.00 in the same column)\n", "form", "1 w/SIMD", "2", "4", "8");
    double base[4] = {0, 0, 0, 0};
    for (const Entry& e : tab) {
        printf("%-40s", e.name);
        int col = 0;
        for (int wps : {1, 2, 4, 8}) {
            double ms = 1e30;
            for (int r = 0; r < 3; ++r) { const double t = time_kernel(e.k, out, in, simds * wps, iters); if (t < ms) ms = t; }
            const double per = ms / ((double)iters * 64.0 * wps);   // ms per slot per wave
            if (e.k == (kern_t)k_fma) base[col] = per;
            printf(" %8.2f", 4.0 * per / base[col]);
            ++col;
        }
        printf("\n");
    }
    // absolute: v_fma_f32 wave-instructions per ns per SIMD
    for (int i = 0, wps = 1; i < 4; ++i, wps *= 2)
        printf("v_fma_f32 at %d w/SIMD: %.3f wave-instructions per ns per SIMD (= %.2f GHz / 4 cycles)\n", wps, 1e-6 / base[i], 4e-6 / base[i]);
    return 0;
}
