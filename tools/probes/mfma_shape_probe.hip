// Does the MFMA shape change the sustained FLOP/s of a power-limited loop?  Same operand bytes from LDS, same MAC count:
//   shape 0: v_mfma_f32_32x32x16_bf16, wave tile 32 x 64 (2 accumulators), 6 B fragments per 12 MFMAs
//   shape 1: v_mfma_f32_16x16x32_bf16, wave tile 32 x 64 (2 x 4 accumulators), 12 B fragments per 48 MFMAs (same bytes)
// One wave per SIMD, 256 workgroups of 256 threads, operands random (or zero with argv[1] = 0).
// hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip && ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(const unsigned short* __restrict__ src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 24 * 1024 / 2; i += 256) reinterpret_cast<unsigned short*>(smem)[i] = src[i];
    bf16x8 a[4][3];      // "stationary" operand: 4 k-steps x 3 planes
    for (int k = 0; k < 4; ++k)
        for (int u = 0; u < 3; ++u)
            a[k][u] = *reinterpret_cast<const bf16x8*>(src + 12288 + ((k * 3 + u) * 64 + lane) * 8);
    __syncthreads();
    const unsigned char* bs = smem + lane * 16;
    if (SHAPE == 0) {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bf16x8 b[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < 3; ++u) b[j][u] = *reinterpret_cast<const bf16x8*>(bs + ((k * 2 + j) * 3 + u) * 1024);
#define T(u_, v_) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][v_], a[k][u_], acc[j], 0, 0, 0);
                T(1, 1) T(2, 0) T(0, 2) T(1, 0) T(0, 1) T(0, 0)
#undef T
            }
        }
        float s = 0;
        for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        // 32 rows = 2 row blocks of 16; 64 columns = 4 column blocks of 16; k-step 32: per 64 k of the other shape, 2 steps
        f32x4 acc[2][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bf16x8 b[4][3];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < 3; ++u) b[j][u] = *reinterpret_cast<const bf16x8*>(bs + ((k * 4 + j) * 3 + u) * 1024);
#define T(u_, v_) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][v_], a[2 * k + i][u_], acc[i][j], 0, 0, 0);
                T(1, 1) T(2, 0) T(0, 2) T(1, 0) T(0, 1) T(0, 0)
#undef T
            }
        }
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + tid] = s;
    }
}

int main(int argc, char** argv) {
    const bool rnd = argc < 2 || atoi(argv[1]) != 0;
    std::vector<unsigned short> h(12288 + 12 * 64 * 8);
    srand(1);
    for (auto& v : h) v = rnd ? (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;
    unsigned short* d; float* o;
    (void)hipMalloc(&d, h.size() * 2); (void)hipMalloc(&o, 256 * 256 * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;      // x 48 (or 96) MFMAs
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            for (int w = 0; w < 3; ++w) {
                if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 24 * 1024, 0, d, o, iters);
                else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 24 * 1024, 0, d, o, iters);
            }
            (void)hipEventRecord(e0, 0);
            for (int w = 0; w < 10; ++w) {
                if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 24 * 1024, 0, d, o, iters);
                else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 24 * 1024, 0, d, o, iters);
            }
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = 10.0 * 256 * 4 * iters * 48.0 * 2 * 32 * 32 * 16;
            printf("%s operands, %s: %.3f ms per launch, %.1f TFLOP/s bf16 (dense peak 2500)\n", rnd ? "random" : "zero",
                   shape == 0 ? "32x32x16" : "16x16x32", ms / 10, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
