// Small workgroup tiles of the exact-fp32 engine (128x64, 64x64): problems that would not fill 1.5 workgroups per CU
// with 128x128 tiles (batch-1 rollout, small-batch training).  Separate translation unit for build parallelism.
#include "pa2d_gemm_kc_kernel.h"

// K-step 32 whenever the operand layout allows it (plain GEMMs always; conv when Cin % 32 == 0), else 16
#define KC_GO(BM_, BN_, WM_, WN_)                                                                            \
    {                                                                                                      \
        if (!im2col) hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, false, 32>), grid, dim3(256), 0, st, p); \
        else if (t.bk == 32) hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, true, 32>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, true, 16>), grid, dim3(256), 0, st, p);  \
    }

int launch_kc_f32_small(const KCParams& p, bool im2col, const KCTile& t, hipStream_t st) {
    const int tiles_n = ceil_div(p.N, t.bn);
    const dim3 grid(ceil_div(ceil_div(p.M, t.bm), 8) * 8 * tiles_n);
    if (t.bm == 128) KC_GO(128, 64, 4, 1)
    else KC_GO(64, 64, 2, 2)
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}
