// fp32 MFMA GEMM engine for the dense part of the Transolver block (gfx950 / CDNA4).
//
// Two kernels cover every dense contraction of the hot path (SURVEY.md §2b rows K1/K5 and their
// backward):
//
//   gemm_kc : C[M,N] = epi(A[M,K] . B[N,K]^T)            both operands K-contiguous in memory.
//             A may be an implicit im2col view of an NHWC image (3x3, pad 1): the conv of
//             Physics_Attention.py:94,96 and its data-gradient run as one implicit GEMM without
//             ever materialising the patches.  Used for linear fwd / bwd-data, conv fwd / bwd-data.
//   gemm_mc : S[i,j]  = sum_m A[m,i] . B[m,j]             contraction over the ROW index of both
//             operands (weight gradients: dW = dY^T X), split over workgroups along m; partial
//             slabs are summed by reduce_slabs (deterministic, no float atomics).  B may be the
//             im2col view (conv weight gradient).
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak), a
// 128x128x16 (or 128x64 / 64x64) workgroup tile, 4 waves = one per SIMD, two workgroups per CU so
// one workgroup's global->LDS staging hides behind the other's MFMAs, register-staged double
// buffering with one barrier per K-step.
//
// Fragment maps (cdna_hip_programming.md §3): A operand lane l holds A[i=l&31][k=l>>5], B operand
// holds B[k=l>>5][j=l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// In gemm_kc each lane fetches 4 consecutive k with one ds_read_b128 and feeds MFMA step t with
// element t of both fragments, i.e. MFMA-k {0,1} <-> real k {8kk+t, 8kk+4+t}: any permutation of k
// is fine as long as A and B use the same one.
#include "pa2d_gemm_kc_kernel.h"

// K-step 32 whenever the operand layout allows it (plain GEMMs always; conv when Cin % 32 == 0), else 16
#define KC_GO(BM_, BN_, WM_, WN_)                                                                            \
    {                                                                                                      \
        if (!im2col) hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, false, 32>), grid, dim3(256), 0, st, p); \
        else if (t.bk == 32) hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, true, 32>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, true, 16>), grid, dim3(256), 0, st, p);  \
    }

int launch_kc_f32_small(const KCParams& p, bool im2col, const KCTile& t, hipStream_t st);   // pa2d_gemm_kc_small.hip

int launch_kc_f32(const KCParams& p, bool im2col, const KCTile& t, hipStream_t st) {
    if (t.bm != 128 || t.bn != 128) return launch_kc_f32_small(p, im2col, t, st);
    const int tiles_n = ceil_div(p.N, t.bn);
    const dim3 grid(ceil_div(ceil_div(p.M, t.bm), 8) * 8 * tiles_n);
    KC_GO(128, 128, 2, 2)
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}
