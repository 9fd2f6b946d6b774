// bf16 engines of the dense path: fp32-accurate 6-term split and bf16 compute (gfx950 / CDNA4).
// gemm_kc_split: the same contraction as gemm_kc on the bf16 matrix cores at fp32 accuracy.
// gfx950 has no xf32/TF32 path and its f32 MFMA runs at 1/16 of the bf16 MFMA rate, so each fp32
// operand is split exactly into three bf16 terms  x = hi + mid + lo (+ <=2^-25 |x|)  while it is
// staged into LDS, and every product is evaluated as the six terms of order <= 2
//     hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid
// with v_mfma_f32_32x32x16_bf16 accumulating in fp32 (dropped terms are <= 2^-24 relative, the same
// order as the fp32 rounding of one product).  6 bf16 MFMAs of K=16 (192 cycles) replace 8 fp32
// MFMAs of K=2 (512 cycles) per 32x32x16 block.  Numerics are checked by the same parity tests and
// tolerances as the fp32 engine.
// LDS row = [hi(64 B) | mid(64 B) | lo(64 B) | pad 16 B] for 32 k: pitch 208 B = 13 x 16 B, an odd
// number of 16-byte slots, so the ds_read_b128 fragment reads (lane l: row l&31, k-half l>>5) of every
// 16-lane group hit 16 different slots.  One LDS stage (53 KB/workgroup, 3 workgroups per CU): the
// next K-step's global loads are issued before the MFMAs of the current one and converted after them.
#include "pa2d_gemm_common.h"

// Wave-specialised workgroup of 8 waves (one producer + one consumer wave per SIMD, so the VALU
// conversion work and the matrix pipe run concurrently):
//   waves 4-7 (producers): buffer-load the fp32 A/B tiles of K-step k+1, split them, write LDS stage
//                          (k+1)&1, then put the loads of K-step k+2 in flight;
//   waves 0-3 (consumers): 48 MFMAs per K-step on stage k&1 (2x2 tiles of 32x32, 2 k-halves, 6 terms).
// One __syncthreads per K-step hands stage (k+1)&1 over and frees stage k&1.
// NT = 3: the 6-term fp32-accuracy split above.  NT = 1: plain bf16 compute (operands rounded to bf16,
// ONE MFMA term, fp32 accumulate) — the autocast-style numerics of BASELINE configs[2]/[4].
// APRE: the A operand arrives pre-split as well ([pixel][32-channel chunk][plane][32] bf16, made once per tensor by
// split_planes_kernel), so BOTH tiles are staged with 16-byte copies and the producers do no conversion work:
// each activation element is converted once instead of once per (tap, column tile) = 36 times.
// TO: storage type of C / res / aux (float, or bf16 for the bf16-storage entry points).  APRE without IM2COL: A is a
// row-major bf16 matrix [M][K] (K % 32 == 0), i.e. already the 1-plane image of every 32-wide K-step.
template <int BM, int BN, bool IM2COL, int NT, bool APRE = false, typename TO = float>
__global__ __launch_bounds__(512, (BM + BN > 256) ? 1 : 2) void gemm_kc_split_kernel(const KCParams p) {
    constexpr int BK = 32, PITCHB = NT * 64 + 16;        // bytes per LDS row (NT planes x 64 B + 16): 208 / 80
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;        // float4 per producer thread per K-step
    constexpr int STAGE = (BM + BN) * PITCHB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tmx = (tiles_m + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_m = xcd * tmx + slot / tiles_n;
    const int tile_n = slot % tiles_n;
    if (tile_m >= tiles_m || slot / tiles_n >= tmx) return;
    const int nk = (p.K + BK - 1) / BK;
    const bool producer = wave >= 4;                     // wave-uniform

    if (producer) {
        const int ptid = tid - 256;
        const int lr = ptid >> 3, lq = ptid & 7;         // 8 lanes cover one 128-byte row segment
        const __amdgpu_buffer_rsrc_t ra_rsrc = make_rsrc(p.A, p.a_bytes);
        const __amdgpu_buffer_rsrc_t rb_rsrc = make_rsrc(p.B, p.b_bytes);
        // The conv weights arrive PRE-SPLIT from the pack kernel as the exact LDS row image
        // ([n][K-step][plane][32] bf16): the B tile is then copied with 16-byte pieces, no VALU.
        constexpr bool PRESPLIT = IM2COL;
        constexpr int PIECES = NT * 4;                            // 16-byte pieces per row and K-step
        constexpr int BP_IT = (BN * PIECES) / 256;                // pieces per producer thread
        constexpr int AP_IT = (BM * PIECES) / 256;                // A pieces per producer thread (APRE)
        unsigned a_off[A_IT], b_off[B_IT], bp_off[BP_IT], bp_lds[BP_IT];
        int a_y[A_IT], a_x[A_IT];
        unsigned ap_off[AP_IT], ap_lds[AP_IT];
        int ap_y[AP_IT], ap_x[AP_IT];
        const int nch_in = p.Cin / 32;                            // 32-channel chunks per pixel (APRE)
#pragma unroll
        for (int s = 0; s < AP_IT; ++s) {
            const int q = ptid + 256 * s, row = q / PIECES, piece = q - row * PIECES;
            const int gm = tile_m * BM + row;
            ap_off[s] = (APRE && gm < p.M) ? (IM2COL ? ((unsigned)gm * (unsigned)(nch_in * PIECES) + piece) * 16u
                                                     : (unsigned)gm * (unsigned)p.lda * 2u + piece * 16u) : OOB_OFF;
            ap_lds[s] = row * PITCHB + piece * 16;
            const int n = (APRE && IM2COL) ? gm % (p.H * p.W) : 0;
            ap_y[s] = n / p.W;
            ap_x[s] = n - ap_y[s] * p.W;
        }
#pragma unroll
        for (int s = 0; s < A_IT; ++s) {
            const int gm = tile_m * BM + lr + 32 * s;
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
            const int gn = tile_n * BN + lr + 32 * s;
            b_off[s] = gn < p.N ? (unsigned)gn * (unsigned)p.ldb * 4u + lq * 16u : OOB_OFF;
        }
#pragma unroll
        for (int s = 0; s < BP_IT; ++s) {
            const int q = ptid + 256 * s, row = q / PIECES, piece = q - row * PIECES;
            const int gn = tile_n * BN + row;
            bp_off[s] = gn < p.N ? ((unsigned)gn * (unsigned)nk * PIECES + piece) * 16u : OOB_OFF;
            bp_lds[s] = row * PITCHB + piece * 16;
        }
        // two register sets: K-step c lives in set c&1 and is loaded two barriers before it is stored
        float4 ra0[A_IT], ra1[A_IT], rb0[B_IT], rb1[B_IT];
        u32x4 rp0[BP_IT], rp1[BP_IT], rq0[AP_IT], rq1[AP_IT];
#define KS_LOAD(kc_, RA, RB, RP, RQ)                                                                   \
    {                                                                                                  \
        const int k0_ = (kc_) * BK;                                                                    \
        const bool kin_ = k0_ + lq * 4 < p.K;                                                          \
        if (APRE && !IM2COL) {                                                                         \
            _Pragma("unroll") for (int s = 0; s < AP_IT; ++s)                                          \
                RQ[s] = __builtin_amdgcn_raw_buffer_load_b128(                                         \
                    ra_rsrc, ap_off[s] != OOB_OFF ? ap_off[s] + (unsigned)(kc_) * (PIECES * 16u) : OOB_OFF, 0, 0); \
        } else if (APRE) {                                                                             \
            const int cic_ = (kc_) / 9, tap_ = (kc_) - cic_ * 9;                                       \
            const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                             \
            const int sh_ = ((dy_ * p.W + dx_) * nch_in + cic_) * (PIECES * 16);                       \
            _Pragma("unroll") for (int s = 0; s < AP_IT; ++s) {                                        \
                const bool ok_ = (unsigned)(ap_y[s] + dy_) < (unsigned)p.H &&                          \
                                 (unsigned)(ap_x[s] + dx_) < (unsigned)p.W && ap_off[s] != OOB_OFF;    \
                RQ[s] = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, ok_ ? ap_off[s] + (unsigned)sh_ : OOB_OFF, 0, 0); \
            }                                                                                          \
        } else if (IM2COL) {                                                                                  \
            const int cic_ = (kc_) / 9, tap_ = (kc_) - cic_ * 9;                                       \
            const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                             \
            const int sh_ = ((dy_ * p.W + dx_) * (int)p.lda + cic_ * BK) * 4;                          \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                         \
                const bool ok_ = (unsigned)(a_y[s] + dy_) < (unsigned)p.H &&                           \
                                 (unsigned)(a_x[s] + dx_) < (unsigned)p.W && a_off[s] != OOB_OFF;      \
                RA[s] = buf_load4(ra_rsrc, ok_ ? a_off[s] + (unsigned)sh_ : OOB_OFF);                  \
            }                                                                                          \
        } else {                                                                                       \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s)                                           \
                RA[s] = buf_load4(ra_rsrc, (kin_ && a_off[s] != OOB_OFF) ? a_off[s] + k0_ * 4u : OOB_OFF); \
        }                                                                                              \
        if (PRESPLIT) {                                                                                \
            _Pragma("unroll") for (int s = 0; s < BP_IT; ++s)                                          \
                RP[s] = __builtin_amdgcn_raw_buffer_load_b128(                                         \
                    rb_rsrc, bp_off[s] != OOB_OFF ? bp_off[s] + (unsigned)(kc_) * (PIECES * 16u) : OOB_OFF, 0, 0); \
        } else {                                                                                       \
            _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                           \
                RB[s] = buf_load4(rb_rsrc, (kin_ && b_off[s] != OOB_OFF) ? b_off[s] + k0_ * 4u : OOB_OFF); \
        }                                                                                              \
    }
#define KS_STORE(buf_, RA, RB, RP, RQ)                                                                 \
    {                                                                                                  \
        unsigned char* const As_ = smem + (buf_) * STAGE;                                              \
        unsigned char* const Bs_ = As_ + BM * PITCHB;                                                  \
        if (APRE) {                                                                                    \
            _Pragma("unroll") for (int s = 0; s < AP_IT; ++s)                                          \
                *reinterpret_cast<u32x4*>(As_ + ap_lds[s]) = RQ[s];                                    \
        } else {                                                                                       \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                         \
                bf16x4 h_, m_, l_;                                                                     \
                split3(RA[s], h_, m_, l_);                                                             \
                unsigned char* d_ = As_ + (lr + 32 * s) * PITCHB + lq * 8;                             \
                *reinterpret_cast<bf16x4*>(d_) = h_;                                                   \
                if (NT == 3) {                                                                         \
                    *reinterpret_cast<bf16x4*>(d_ + 64) = m_;                                          \
                    *reinterpret_cast<bf16x4*>(d_ + 128) = l_;                                         \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
        if (PRESPLIT) {                                                                                \
            _Pragma("unroll") for (int s = 0; s < BP_IT; ++s)                                          \
                *reinterpret_cast<u32x4*>(Bs_ + bp_lds[s]) = RP[s];                                    \
        } else {                                                                                       \
            _Pragma("unroll") for (int s = 0; s < B_IT; ++s) {                                         \
                bf16x4 h_, m_, l_;                                                                     \
                split3(RB[s], h_, m_, l_);                                                             \
                unsigned char* d_ = Bs_ + (lr + 32 * s) * PITCHB + lq * 8;                             \
                *reinterpret_cast<bf16x4*>(d_) = h_;                                                   \
                if (NT == 3) {                                                                         \
                    *reinterpret_cast<bf16x4*>(d_ + 64) = m_;                                          \
                    *reinterpret_cast<bf16x4*>(d_ + 128) = l_;                                         \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }
        KS_LOAD(0, ra0, rb0, rp0, rq0)
        if (nk > 1) KS_LOAD(1, ra1, rb1, rp1, rq1)
        KS_STORE(0, ra0, rb0, rp0, rq0)
        if (nk > 2) KS_LOAD(2, ra0, rb0, rp0, rq0)
        __syncthreads();
        for (int kc = 0; kc < nk; kc += 2) {
            if (kc + 1 < nk) {                               // K-step kc+1 lives in set 1
                KS_STORE((kc + 1) & 1, ra1, rb1, rp1, rq1)
                if (kc + 3 < nk) KS_LOAD(kc + 3, ra1, rb1, rp1, rq1)
            }
            __syncthreads();
            if (kc + 1 < nk) {
                if (kc + 2 < nk) {                           // K-step kc+2 lives in set 0
                    KS_STORE((kc + 2) & 1, ra0, rb0, rp0, rq0)
                    if (kc + 4 < nk) KS_LOAD(kc + 4, ra0, rb0, rp0, rq0)
                }
                __syncthreads();
            }
        }
#undef KS_LOAD
#undef KS_STORE
        return;
    }

    // ---------------- consumers
    // Ablation of the 6-term conv launch (2.1 ms): MFMAs skipped 1.16 ms, global loads skipped 1.48,
    // convert+LDS store skipped 1.56, both skipped 1.24 (= consumers alone; MFMA floor 0.84), loads alone
    // 0.73 (40 KB per K-step = 32 B/clk/CU, the CU's load-path limit), convert+store alone 0.78.  With one
    // workgroup per CU the staging phases do not hide behind the MFMAs; static MFMA-wave priority
    // (s_setprio) made no difference.  Next step: a tiling that converts each A tile once, not 4 times.
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frag_off = (lane & 31) * PITCHB + (lane >> 5) * 16;
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const unsigned char* a_s = smem + (kc & 1) * STAGE + wm * WM * PITCHB + frag_off;
        const unsigned char* b_s = smem + (kc & 1) * STAGE + BM * PITCHB + wn * WN * PITCHB + frag_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM][NT], bf[TN][NT];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    af[i][q] = *reinterpret_cast<const bf16x8*>(a_s + i * 32 * PITCHB + q * 64 + ks * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    bf[j][q] = *reinterpret_cast<const bf16x8*>(b_s + j * 32 * PITCHB + q * 64 + ks * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (NT == 3) {   // smallest terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    kc_epilogue<TM, TN, IM2COL, TO>(p, acc, tile_m * BM + wm * WM + 4 * (lane >> 5), tile_n * BN + wn * WN + (lane & 31));
}

// Variants in use: conv (im2col) GEMMs always take pre-split activation planes (NT = 3: fp32-accurate split,
// NT = 1: bf16 compute); plain GEMMs run only in bf16-compute mode and convert while staging.
// Small plain GEMMs of the split engine (rollout / training at batch 1-4: a few thousand rows): 64 x 64 tiles so that
// M = 4096, N = 256 still makes 256 workgroups, both operands split while staged.  The exact-fp32 kernel these launches
// used to take spends 1024 matrix-pipe cycles per 32-deep K-step and wave (16 x v_mfma_f32_32x32x2_f32), this one 384.
bool kc_split_small_applies(const KCParams& p, bool im2col) {
    if (!pa2d_env().lin_small_split || im2col || p.io_bf16 || p.engine != 1) return false;
    if ((p.K % 32) != 0 || p.K < 64 || p.N < 64 || p.M < 256) return false;
    return (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128) < 384;      // where kc_tile leaves the 128 x 128 tiles
}
int launch_kc_split_small(const KCParams& p, hipStream_t st) {
    const int smem = 2 * (64 + 64) * 208;
    {   // every launch: the attribute is per device, and the call is cheap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<64, 64, false, 3, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid(ceil_div(ceil_div(p.M, 64), 8) * 8 * ceil_div(p.N, 64));
    hipLaunchKernelGGL((gemm_kc_split_kernel<64, 64, false, 3, false>), grid, dim3(512), smem, st, p);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int launch_kc_split(KCParams& p, bool im2col, hipStream_t st) {
    const int tiles_m = ceil_div(p.M, 128), tiles_n = ceil_div(p.N, 128);
    const bool bf = p.engine == 2;
    const int planes = bf ? 1 : 3;
    if (im2col) {
        if (!p.apre) return PA2D_ERR_ARG;
        const unsigned long long pb = (unsigned long long)p.M * (p.Cin / 32) * planes * 64;
        if (pb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.a_bytes = (unsigned)pb;
        p.b_bytes = (unsigned)((size_t)p.N * (p.K / 32) * planes * 64);
    } else if (!bf) {
        return PA2D_ERR_ARG;
    }
    if (p.io_bf16) {          // bf16-storage entry points: A is the bf16 tensor itself, outputs are bf16
        if (!bf || (!im2col && (p.K % 32) != 0)) return PA2D_ERR_UNSUPPORTED;
        if (!im2col) {
            const unsigned long long ab = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2ull;
            if (ab >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
            p.a_bytes = (unsigned)ab;
        }
        if (im2col && conv_halo_applies(p)) return launch_conv_halo(p, st);
        const dim3 g16(ceil_div(tiles_m, 8) * 8 * tiles_n);
        if (im2col) hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 1, true, bf16_t>), g16, dim3(512), 2 * 256 * 80, st, p);
        else hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, false, 1, true, bf16_t>), g16, dim3(512), 2 * 256 * 80, st, p);
        PA2D_CHECK_LAUNCH();
        return PA2D_OK;
    }
    if (im2col && conv_halo_applies(p)) return launch_conv_halo(p, st);
    const dim3 grid(ceil_div(tiles_m, 8) * 8 * tiles_n);
    const int smem = 2 * (128 + 128) * (bf ? 80 : 208);
    {   // every launch: the attribute is per device, and the call is cheap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<128, 128, true, 3, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 128) * 208);
        if (e != hipSuccess) return (int)e;
    }
    if (pa2d_env().split_big && im2col && !bf && (p.M % 256) == 0 && (long long)(p.M / 256) * tiles_n >= 512) {
        // 256x128 workgroup tile (128x64 per consumer wave): 25 % less L2->LDS staging and LDS reads per MFMA
        const int smem_big = 2 * (256 + 128) * 208;
        {
            hipError_t eb = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<256, 128, true, 3, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, smem_big);
            if (eb != hipSuccess) return (int)eb;
        }
        const dim3 gbig(ceil_div(p.M / 256, 8) * 8 * tiles_n);
        hipLaunchKernelGGL((gemm_kc_split_kernel<256, 128, true, 3, true>), gbig, dim3(512), smem_big, st, p);
    } else if (im2col && !bf && (long long)tiles_m * tiles_n < 256) {
        // small launches (rollout at batch 1: 32 x 4 tiles of 128 x 128 leave half of the 256 CUs idle): 64-row tiles
        const int smem_s = 2 * (64 + 128) * 208;
        {
            hipError_t es = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<64, 128, true, 3, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, smem_s);
            if (es != hipSuccess) return (int)es;
        }
        const dim3 gs(ceil_div(ceil_div(p.M, 64), 8) * 8 * tiles_n);
        hipLaunchKernelGGL((gemm_kc_split_kernel<64, 128, true, 3, true>), gs, dim3(512), smem_s, st, p);
    } else
    if (!im2col) hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, false, 1, false>), grid, dim3(512), smem, st, p);
    else if (bf) hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 1, true>), grid, dim3(512), smem, st, p);
    else hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 3, true>), grid, dim3(512), smem, st, p);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// Activation pre-split for the bf16 engines: src [rows][ld >= C] fp32 -> dst [rows][C/32][NT][32] bf16 (NT = 3:
// hi | mid | lo with x = hi + mid + lo up to 2^-25 |x|; NT = 1: x rounded to bf16).  One thread per 4 channels.
__global__ void split_planes_kernel(const float* __restrict__ src, long long ld, __bf16* __restrict__ dst,
                                    long long rows, int C, int NT) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q4 = C / 4;
    if (idx >= rows * q4) return;
    const long long row = idx / q4;
    const int c4 = (int)(idx - row * q4), chunk = c4 >> 3, q = c4 & 7;
    const float4 v = *reinterpret_cast<const float4*>(src + row * ld + c4 * 4);
    bf16x4 h, m, l;
    split3(v, h, m, l);
    __bf16* d = dst + ((row * (C / 32) + chunk) * NT) * 32 + q * 4;
    *reinterpret_cast<bf16x4*>(d) = h;
    if (NT == 3) {
        *reinterpret_cast<bf16x4*>(d + 32) = m;
        *reinterpret_cast<bf16x4*>(d + 64) = l;
    }
}
size_t planes_bytes(long long rows, int C, int NT) { return (size_t)rows * C * NT * 2; }
int launch_split_planes(const float* src, long long ld, void* dst, long long rows, int C, int NT, hipStream_t st) {
    if (C & 31) return PA2D_ERR_UNSUPPORTED;
    const long long n = rows * (C / 4);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, st, src, ld, (__bf16*)dst,
                       rows, C, NT);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// conv weight pack for the bf16 engines: dst row n = [K-step kc = cic*9+tap][plane][32 channels] bf16,
// i.e. the LDS row image of gemm_kc_split_kernel (NT planes: hi | mid | lo).  bwd != 0: data-gradient
// layout (rows = input channels, contraction over the 2C output channels, taps mirrored).
__global__ void repack_split_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                    __bf16* __restrict__ dst, int bwd, int NT, int C, int Cin) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)2 * C * 9 * Cin) return;
    float v;
    int n, kc, c32;
    if (!bwd) {          // rows n = output channel co' in [0,2C), K = 9*Cin
        c32 = (int)(idx % 32);
        const int tap = (int)((idx / 32) % 9);
        const int cic = (int)((idx / 288) % (Cin / 32));
        n = (int)(idx / ((long long)Cin * 9));
        kc = cic * 9 + tap;
        const float* src = n < C ? w0 : w1;
        v = src[((size_t)(n % C) * Cin + cic * 32 + c32) * 9 + tap];
    } else {             // rows n = input channel ci in [0,Cin), K = 9*2C
        c32 = (int)(idx % 32);
        const int tap = (int)((idx / 32) % 9);
        const int cic = (int)((idx / 288) % (2 * C / 32));
        n = (int)(idx / ((long long)2 * C * 9));
        kc = cic * 9 + tap;
        const int co = cic * 32 + c32;
        const float* src = co < C ? w0 : w1;
        v = src[((size_t)(co % C) * Cin + n) * 9 + (8 - tap)];
    }
    const int nk = bwd ? (2 * C / 32) * 9 : (Cin / 32) * 9;
    __bf16* d = dst + ((size_t)n * nk + kc) * NT * 32 + c32;
    const __bf16 h = (__bf16)v;
    d[0] = h;
    if (NT == 3) {
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        d[32] = m;
        d[64] = (__bf16)(r1 - (float)m);
    }
}

int launch_repack_split(const float* w0, const float* w1, void* dst, int bwd, int NT, int C, int Cin, hipStream_t st) {
    hipLaunchKernelGGL(repack_split_kernel, dim3((unsigned)ceil_div_ll((long long)2 * C * 9 * Cin, 256)), dim3(256), 0, st, w0,
                       w1, (__bf16*)dst, bwd, NT, C, Cin);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}
