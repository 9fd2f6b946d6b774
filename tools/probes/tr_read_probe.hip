// Probe of gfx950's ds_read_b64_tr_b16 lane mapping (cdna_hip_programming.md T10): per 16-lane group a 4-row x 16-column
// block of 16-bit elements; lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i, row e in
// element e.  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe ; prints mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* src, short* dst) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 128];
    for (int i = threadIdx.x; i < 16 * 128; i += 64) tile[i] = src[i];
    __syncthreads();
    const int l = threadIdx.x, q = (l & 15) >> 2, p = l & 3, g = l >> 4;
    const short* a = tile + (8 * (g >> 1) + q) * 128 + 16 * (g & 1) + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int e = 0; e < 4; ++e) dst[l * 4 + e] = v[e];
}
int main() {
    std::vector<short> h(16 * 128), o(256);
    for (int i = 0; i < 16 * 128; ++i) h[i] = (short)i;
    short *d, *r;
    hipMalloc(&d, h.size() * 2); hipMalloc(&r, 512);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r);
    hipMemcpy(o.data(), r, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int g = l >> 4, want = (8 * (g >> 1) + e) * 128 + 16 * (g & 1) + (l & 15);
            if (o[l * 4 + e] != (short)want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", l, e, o[l * 4 + e], want); ++bad; }
        }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
