// Where does global_load_lds_dwordx4 with an instruction offset land in LDS?  (M0 base + offset + lane*16, or M0 base + lane*16)
// hipcc --offload-arch=gfx950 -O2 -o lds_dma_probe lds_dma_probe.hip && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s = (float*)smem;
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = -1.f;
    __syncthreads();
    const float* src = g + threadIdx.x * 4;
    unsigned keep, base = 4096;      // LDS byte address 4096
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(src), "s"(base) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = s[i];
}
int main() {
    float *g, *o;
    hipMalloc(&g, 65536); hipMalloc(&o, 16384);
    std::vector<float> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = (float)i;
    hipMemcpy(g, h.data(), 65536, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, g, o);
    std::vector<float> r(4096);
    hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
    int first = -1, cnt = 0;
    for (int i = 0; i < 4096; ++i) if (r[i] >= 0) { if (first < 0) first = i; ++cnt; }
    printf("first written float index %d (byte %d), count %d, value there %.0f (global float index)\n", first, first * 4, cnt, first >= 0 ? r[first] : -1.f);
    return 0;
}
