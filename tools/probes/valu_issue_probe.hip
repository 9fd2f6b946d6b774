// Issue cost of vector instructions on one SIMD of gfx950 with 1 or 2 resident waves: each wave runs REP x 32 independent
// copies of one instruction (8 destination registers round-robin, so no back-to-back dependency) and the probe reports
// shader cycles per instruction per SIMD = elapsed s_memtime / (instructions per wave x waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O2 -o valu_issue_probe valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 256
#define BODY32(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) \
                  I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

#define KERNEL(NAME, INSTR)                                                                         \
    __global__ void NAME(unsigned long long* out, float seed) {                                     \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, \
              a7 = seed + 7, b0 = seed * 0.5f, b1 = seed * 0.25f;                                   \
        float2 c0 = {seed, seed}, c1 = {seed, seed};                                                \
        __syncthreads();                                                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
        for (int i = 0; i < REP; ++i) { INSTR }                                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0.x + c1.y == 123.456f) out[1000000] = 0;     \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define OPS8(X) "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define ASM1(TXT) asm volatile(TXT : OPS8(0) : "v"(b0), "v"(b1));

// %0..%7 destinations, %8 %9 sources
#define I_ADD(n) "v_add_f32 %" #n ", %8, %9\n\t"
#define I_FMA(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n\t"
#define I_AND(n) "v_and_b32 %" #n ", 0xffff0000, %8\n\t"
#define I_SHL(n) "v_lshlrev_b32 %" #n ", 16, %8\n\t"
#define I_CVT(n) "v_cvt_pk_bf16_f32 %" #n ", %8, %9\n\t"
#define I_EXP(n) "v_exp_f32 %" #n ", %8\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %8\n\t"
#define I_MAX3(n) "v_max3_f32 %" #n ", %8, %9, %" #n "\n\t"
#define I_DPP(n) "v_max_f32_dpp %" #n ", %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPM(n) "v_add_f32_dpp %" #n ", %8, %9 row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n\t"
#define I_PERM(n) "v_perm_b32 %" #n ", %8, %9, %8\n\t"
#define I_CND(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n\t"
#define I_NOP(n) "s_nop 0\n\t"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %8, %9, s[10:11]\n\t"
#define I_OR(n) "v_or_b32 %" #n ", %8, %9\n\t"
#define I_CMP(n) "v_cmp_lt_i32 vcc, %8, %9\n\t"
#define I_CMPCND(n) "v_cmp_lt_i32 vcc, %8, %9\n\tv_cndmask_b32 %" #n ", %8, %9, vcc\n\t"

KERNEL(k_add, ASM1(BODY32(I_ADD)))
KERNEL(k_fma, ASM1(BODY32(I_FMA)))
KERNEL(k_and, ASM1(BODY32(I_AND)))
KERNEL(k_shl, ASM1(BODY32(I_SHL)))
KERNEL(k_cvt, ASM1(BODY32(I_CVT)))
KERNEL(k_exp, ASM1(BODY32(I_EXP)))
KERNEL(k_rcp, ASM1(BODY32(I_RCP)))
KERNEL(k_max3, ASM1(BODY32(I_MAX3)))
KERNEL(k_dpp, ASM1(BODY32(I_DPP)))
KERNEL(k_dppm, ASM1(BODY32(I_DPPM)))
KERNEL(k_mov, ASM1(BODY32(I_MOV)))
KERNEL(k_perm, ASM1(BODY32(I_PERM)))
KERNEL(k_cnd, ASM1(BODY32(I_CND)))
KERNEL(k_nop, ASM1(BODY32(I_NOP)))
KERNEL(k_cnd64, asm volatile(BODY32(I_CND64) : OPS8(0) : "v"(b0), "v"(b1) : "s10", "s11");)
KERNEL(k_or, ASM1(BODY32(I_OR)))
KERNEL(k_cmp, asm volatile(BODY32(I_CMP) : OPS8(0) : "v"(b0), "v"(b1) : "vcc");)
KERNEL(k_cmpcnd, asm volatile(BODY32(I_CMPCND) : OPS8(0) : "v"(b0), "v"(b1) : "vcc");)

// packed f32: 64-bit operands
#define PK8 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
#define KERNEL_PK(NAME, INSTR)                                                                      \
    __global__ void NAME(unsigned long long* out, float seed) {                                     \
        typedef float f2 __attribute__((ext_vector_type(2)));                                       \
        f2 p0 = {seed, 1}, p1 = {seed, 2}, p2 = {seed, 3}, p3 = {seed, 4}, p4 = {seed, 5}, p5 = {seed, 6}, p6 = {seed, 7}, \
           p7 = {seed, 8}, q0 = {seed, seed}, q1 = {seed * 0.5f, seed};                             \
        __syncthreads();                                                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
        for (int i = 0; i < REP; ++i) { asm volatile(INSTR : PK8 : "v"(q0), "v"(q1)); }             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
        if (p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y == 123.456f) out[1000000] = 0;   \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %8, %9\n\t"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n "\n\t"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %8, %9\n\t"
KERNEL_PK(k_pkadd, BODY32(I_PKADD))
KERNEL_PK(k_pkfma, BODY32(I_PKFMA))
KERNEL_PK(k_pkmul, BODY32(I_PKMUL))

// MFMA 16x16x32 bf16: 4 independent accumulators; and MFMA interleaved with VALU (1 mfma + K adds)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NADD>
__global__ void k_mfma(unsigned long long* out, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a0 = seed, a1 = seed, a2 = seed, a3 = seed, b0 = seed * 0.5f, b1 = 3.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"
                         "v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\t"
                         "v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n\t"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            if constexpr (NADD > 0) {
#pragma unroll
                for (int k = 0; k < NADD; ++k)
                    asm volatile("v_add_f32 %0, %4, %5\n\tv_add_f32 %1, %4, %5\n\tv_add_f32 %2, %4, %5\n\tv_add_f32 %3, %4, %5\n\t"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + a2 + a3 == 123.456f) out[1000000] = 0;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma16(unsigned long long* out, float seed) {
    bf16x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_16x16x16_bf16 %0, %4, %5, %0\n\t"
                         "v_mfma_f32_16x16x16_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_16x16x16_bf16 %2, %4, %5, %2\n\t"
                         "v_mfma_f32_16x16x16_bf16 %3, %4, %5, %3\n\t"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) out[1000000] = 0;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
// one accumulator, dependent chain
__global__ void k_mfma_dep(unsigned long long* out, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    f32x4 c0 = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                         : "+v"(c0) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (c0[0] == 123.456f) out[1000000] = 0;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int instr_per_wave, unsigned long long* d_out) {
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: block = 256 * wps threads, one block per CU
        const int block = 256 * wps, grid = 256;
        if (block > 1024) continue;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d_out, 1.5f);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d_out, 1.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * block / 64);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        printf("%-10s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD slot\n", name, wps,
               med / instr_per_wave, med / instr_per_wave / wps);
    }
}

int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, 8 * 1100000);
    const int n = REP * 32;
    run("v_add", k_add, n, d_out); run("v_fma", k_fma, n, d_out); run("v_and", k_and, n, d_out);
    run("v_lshl", k_shl, n, d_out); run("cvt_pk", k_cvt, n, d_out); run("v_exp", k_exp, n, d_out);
    run("v_rcp", k_rcp, n, d_out); run("v_max3", k_max3, n, d_out); run("max_dpp", k_dpp, n, d_out);
    run("add_dppm", k_dppm, n, d_out); run("v_mov", k_mov, n, d_out); run("v_perm", k_perm, n, d_out);
    run("v_cndmask", k_cnd, n, d_out); run("s_nop0", k_nop, n, d_out); run("cndmask_s", k_cnd64, n, d_out); run("v_or", k_or, n, d_out); run("v_cmp", k_cmp, n, d_out); run("cmp+cnd", k_cmpcnd, n, d_out);
    run("pk_add", k_pkadd, n, d_out); run("pk_fma", k_pkfma, n, d_out); run("pk_mul", k_pkmul, n, d_out);
    run("mfma", k_mfma<0>, REP * 32, d_out);
    run("mfma16x16x16", k_mfma16, REP * 32, d_out);
    run("mfma_dep", k_mfma_dep, REP * 32, d_out);
    run("mfma+1add", k_mfma<1>, REP * 32, d_out);      // per MFMA; each MFMA comes with NADD v_add
    run("mfma+2add", k_mfma<2>, REP * 32, d_out);
    run("mfma+4add", k_mfma<4>, REP * 32, d_out);
    return 0;
}
