// Sustained v_mfma_f32_32x32x16_bf16 rate on gfx950 under different per-CU shapes (device probe, not product code):
//   waves per SIMD 1 / 2, with / without LDS fragment reads (ds_read_b128 per MFMA as in conv_halo_kernel: 18 per 48),
// to price the conv kernels against what the matrix pipe really sustains (power / clocks), not the datasheet peak.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip ; run: ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS_READS, int TERM_MAJOR = 0>   // LDS_READS 0: operands stay in registers; 1: 18 ds_read_b128 per 48 MFMAs
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, int active_waves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 120 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f / (1 + (i & 255));
    __syncthreads();
    if (wave >= active_waves) return;
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a[4][3], b[2][3];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 3; ++q) for (int e = 0; e < 8; ++e) a[i][q][e] = (__bf16)(0.001f * (lane + i + q + e));
    for (int j = 0; j < 2; ++j) for (int q = 0; q < 3; ++q) for (int e = 0; e < 8; ++e) b[j][q][e] = (__bf16)(0.002f * (lane + j + q + e));
    const unsigned base = (lane & 31) * 208 + (lane >> 5) * 16 + wave * 4096;
    for (int it = 0; it < iters; ++it) {
        if (LDS_READS) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(smem + base + i * 7072 + q * 64 + (it & 1) * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(smem + 70720 + base + j * 6656 + q * 64 + (it & 1) * 32);
        }
        if (TERM_MAJOR) {       // consecutive MFMAs write different accumulators
            constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]], acc[i * 2 + j], 0, 0, 0);
        } else
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16& c = acc[i * 2 + j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int L, int TM = 0> static void run(const char* name, int active, int iters, float* out) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<L, TM>), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    hipLaunchKernelGGL((probe<L, TM>), dim3(grid), dim3(512), 124 * 1024, 0, out, 64, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<L, TM>), dim3(grid), dim3(512), 124 * 1024, 0, out, iters, active);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * active * iters * 48.0 * 32768.0;
    printf("%-44s %8.3f ms  %8.1f TFLOP/s bf16  (%.3f of 2500)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
}

int main() {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 512);
    run<0>("4 waves/CU (1/SIMD), registers only", 4, 4000, out);
    run<0>("8 waves/CU (2/SIMD), registers only", 8, 2000, out);
    run<1>("4 waves/CU (1/SIMD), 18 ds_read_b128 / 48 MFMA", 4, 4000, out);
    run<1>("8 waves/CU (2/SIMD), 18 ds_read_b128 / 48 MFMA", 8, 2000, out);
    run<0, 1>("4 waves/CU, registers only, term-major order", 4, 4000, out);
    run<1, 1>("4 waves/CU, 18 ds_read_b128 / 48 MFMA, term-major", 4, 4000, out);
    return 0;
}
