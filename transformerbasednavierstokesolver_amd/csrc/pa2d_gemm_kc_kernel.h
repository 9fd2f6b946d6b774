// gemm_kc_kernel: the exact-fp32 MFMA GEMM / implicit-GEMM conv kernel template (see pa2d_gemm_kc.hip).
#pragma once
#include "pa2d_gemm_common.h"

template <int BM, int BN, int WAVES_M, int WAVES_N, bool IM2COL, int BK = 16>
__global__ __launch_bounds__(256, 2) void gemm_kc_kernel(const KCParams p) {
    constexpr int PITCH = BK + 4;              // 20 or 36 floats: an odd number of 16-byte slots
    constexpr int QPR = BK / 4;                // float4 per tile row
    constexpr int RPP = 256 / QPR;             // tile rows loaded per pass of the 256 threads
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * PITCH];
    float* const As = smem;
    float* const Bs = smem + 2 * BM * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // XCD-aware map (speed only): blocks b, b+8, ... share an XCD under round-robin dispatch.  Each
    // XCD walks a CONTIGUOUS range of row tiles, all column tiles of a row tile back to back, so
    // co-resident blocks share the A panel (and, for the conv, the halo rows) in that XCD's L2.
    const int tmx = (tiles_m + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_m = xcd * tmx + slot / tiles_n;
    const int tile_n = slot % tiles_n;
    if (tile_m >= tiles_m || slot / tiles_n >= tmx) return;

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = tid / QPR, lq = tid % QPR;

    const __amdgpu_buffer_rsrc_t ra_rsrc = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb_rsrc = make_rsrc(p.B, p.b_bytes);
    unsigned a_off[A_IT], b_off[B_IT];
    int a_y[A_IT], a_x[A_IT];
#pragma unroll
    for (int s = 0; s < A_IT; ++s) {
        const int gm = tile_m * BM + lr + RPP * s;
        a_off[s] = gm < p.M ? (unsigned)gm * (unsigned)p.lda * 4u + lq * 16u : OOB_OFF;
        if (IM2COL) {
            const int n = gm % (p.H * p.W);
            a_y[s] = n / p.W;
            a_x[s] = n - a_y[s] * p.W;
        } else {
            a_y[s] = a_x[s] = 0;
        }
    }
#pragma unroll
    for (int s = 0; s < B_IT; ++s) {
        const int gn = tile_n * BN + lr + RPP * s;
        b_off[s] = gn < p.N ? (unsigned)gn * (unsigned)p.ldb * 4u + lq * 16u : OOB_OFF;
    }

    float4 ra[A_IT], rb[B_IT];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    // K order of the implicit-GEMM conv: k = (ci_chunk*9 + tap)*16 + c, i.e. the 9 taps of one
    // 16-channel chunk are consecutive K-steps: they re-touch the same 64-B segments of the same
    // image rows (shifted by one pixel), which are then L1/L2 hits instead of fresh misses.
#define KC_LOAD(kc_)                                                                                   \
    {                                                                                                  \
        const int k0_ = (kc_) * BK;                                                                    \
        const bool kin_ = k0_ + lq * 4 < p.K;                                                          \
        if (IM2COL) {                                                                                  \
            const int cic_ = (kc_) / 9, tap_ = (kc_) - cic_ * 9;                                       \
            const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                             \
            const int sh_ = ((dy_ * p.W + dx_) * (int)p.lda + cic_ * BK) * 4;                          \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                         \
                const bool ok_ = (unsigned)(a_y[s] + dy_) < (unsigned)p.H &&                           \
                                 (unsigned)(a_x[s] + dx_) < (unsigned)p.W && a_off[s] != OOB_OFF;      \
                ra[s] = buf_load4(ra_rsrc, ok_ ? a_off[s] + (unsigned)sh_ : OOB_OFF);                  \
            }                                                                                          \
        } else {                                                                                       \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s)                                           \
                ra[s] = buf_load4(ra_rsrc, (kin_ && a_off[s] != OOB_OFF) ? a_off[s] + k0_ * 4u : OOB_OFF); \
        }                                                                                              \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                               \
            rb[s] = buf_load4(rb_rsrc, (kin_ && b_off[s] != OOB_OFF) ? b_off[s] + k0_ * 4u : OOB_OFF); \
    }
#define KC_STORE(buf_)                                                                                 \
    {                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < A_IT; ++s)                                               \
            *reinterpret_cast<float4*>(As + (buf_) * BM * PITCH + (lr + RPP * s) * PITCH + lq * 4) = ra[s]; \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                               \
            *reinterpret_cast<float4*>(Bs + (buf_) * BN * PITCH + (lr + RPP * s) * PITCH + lq * 4) = rb[s]; \
    }

    KC_LOAD(0)
    KC_STORE(0)
    __syncthreads();
    const int frag_off = (lane & 31) * PITCH + (lane >> 5) * 4;
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) KC_LOAD(kc + 1)
        const float* a_s = As + buf * BM * PITCH + wm * WM * PITCH + frag_off;
        const float* b_s = Bs + buf * BN * PITCH + wn * WN * PITCH + frag_off;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_s + i * 32 * PITCH + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b_s + j * 32 * PITCH + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kc + 1 < nk) KC_STORE(buf ^ 1)
        __syncthreads();
    }
#undef KC_LOAD
#undef KC_STORE

    // epilogue: lanes 0-31 of register r write 32 consecutive floats of one row (128 B)
    kc_epilogue<TM, TN, IM2COL>(p, acc, tile_m * BM + wm * WM + 4 * (lane >> 5), tile_n * BN + wn * WN + (lane & 31));
}

