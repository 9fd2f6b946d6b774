// Row-panel GEMM of the bf16 engines for the K = C linears (to_out, MLP fc1 / fc2 and their data gradients):
//     C[M,N] = epilogue(A[M,K] . B[N,K]^T)        M = B*H*W rows (10^5 .. 10^6), N, K a few hundred.
// With K = 256 a 128x128 output tile lives for only 8 K-steps between a cold prologue and a 64 KB epilogue, so the
// per-tile kernels (gemm_kc, gemm_kc_split) run these layers at the latency of that start/stop sequence, not at the
// matrix pipe's rate (measured: 0.155-0.18 ms on either engine, 0.08 ms even with ONE bf16 term).  Here the workgroups
// are PERSISTENT (one per CU, XCD-contiguous tile ranges) and the K-steps of all their tiles form one stream:
//   waves 4-7 (producers): buffer-load the fp32 A / B tiles two to three K-steps ahead — across tile boundaries, so the
//                          next tile's first stages are in LDS before the consumers finish the current epilogue —
//                          split them into the NT bf16 planes (x = hi + mid + lo, exact to 2^-25 |x|) and write the
//                          LDS stage;
//   waves 0-3 (consumers): 256 x 128 tile, 128 x 64 per wave (96 MFMAs per K-step at NT = 3: 6 terms of order <= 2,
//                          fp32 accumulate), then the epilogue straight from the accumulators.
// Epilogues address elements as per-lane offset + SGPR row offset (no per-element address arithmetic); residual and
// pre-activation operands are fetched in 32-load batches.  NT = 3: PA2D_ENGINE_SPLIT (fp32 accuracy), NT = 1: bf16
// compute.  AF32 = false: A is a bf16 matrix (bf16-storage entry points), copied with 16-byte pieces.
// Phase ablation at B*N = 131072, N = K = 256, NT = 3 (ms): full 0.116; MFMAs alone 0.079 (0.056 + 0.023 epilogue =
// the HBM write of C); loads + convert alone 0.062; loads alone 0.046; nothing but barriers and the epilogue 0.023.
// The producers' VALU conversion and the consumers' MFMAs add up instead of overlapping (same finding as on the conv),
// which is what the plane-image operands remove on the conv path.  Starting every second pair of workgroups 3-12 us late
// (so that their epilogue writes fall into the other half's K loops) measured neutral.  Pre-split weights (a plane image made
// once per iteration, copied by the producers instead of converted) measured 0.4 % SLOWER on the whole step in a same-box
// A/B: the weight conversion is not what the layer waits for, and the image is 1.5x the bytes.
#include "pa2d_gemm_common.h"

namespace {

// buffer access with a wave-uniform (SGPR) offset on top of the per-lane one
template <typename TO> __device__ __forceinline__ float ld_so(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float ld_so<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <> __device__ __forceinline__ float ld_so<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
}
template <typename TO> __device__ __forceinline__ void st_so(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v);
template <> __device__ __forceinline__ void st_so<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
template <> __device__ __forceinline__ void st_so<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16_bits(v), r, voff, soff, 0);
}

// MODE 0: out = acc + bias (+ res; the bias is preloaded into the accumulators, the residual is added in batches);
// MODE 1: aux = acc + bias (dropped when aux == NULL), out = gelu(acc + bias);  MODE 2: out = acc * gelu'(aux).
// Requires M % 256 == 0 (the host sends a ragged tail of < 256 rows to the per-tile kernels); ragged N is masked.
template <int NT, bool AF32, typename TO, int MODE>
__global__ __launch_bounds__(512, 1) void gemm_panel_kernel(const KCParams p, const int tmx, const int S) {
    constexpr int BM = 256, BN = 128, BK = 32, PITCHB = NT * 64 + 16;
    constexpr int TM = 4, TN = 2;
    constexpr int STAGE = (BM + BN) * PITCHB;
    constexpr unsigned ES = Act<TO>::ES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // SGPR: everything derived from it is wave-uniform
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, s0 = blockIdx.x >> 3;
    int rows_here = tiles_m - xcd * tmx;
    rows_here = rows_here > tmx ? tmx : (rows_here < 0 ? 0 : rows_here);
    const int qmax = rows_here * tiles_n;            // tiles of this XCD, column tile fastest
    const int n_my = s0 < qmax ? (qmax - s0 + S - 1) / S : 0;
    if (n_my == 0) return;
    const int nk = p.K / BK;
    const int T = n_my * nk;                         // K-steps of this workgroup's stream

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int ptid = tid - 256, lr = ptid >> 3, lq = ptid & 7;
        const __amdgpu_buffer_rsrc_t ra_rsrc = make_rsrc(p.A, p.a_bytes);
        const __amdgpu_buffer_rsrc_t rb_rsrc = make_rsrc(p.B, p.b_bytes);
        constexpr int A_IT = BM / 32, B_IT = BN / 32, AQ_IT = AF32 ? 1 : (BM * 4) / 256;
        unsigned a_off[A_IT], b_off[B_IT], aq_off[AQ_IT];
        int q = s0, kc_ld = 0;
#define PN_TILE()                                                                                          \
    {                                                                                                      \
        const int row0_ = (xcd * tmx + q / tiles_n) * BM, col0_ = (q % tiles_n) * BN;                      \
        if (AF32) {                                                                                        \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                             \
                const int gm = row0_ + lr + 32 * s;                                                        \
                a_off[s] = gm < p.M ? (unsigned)gm * (unsigned)p.lda * 4u + lq * 16u : OOB_OFF;            \
            }                                                                                              \
        } else {                                                                                           \
            _Pragma("unroll") for (int s = 0; s < AQ_IT; ++s) {                                            \
                const int qq = ptid + 256 * s, gm = row0_ + (qq >> 2);                                     \
                aq_off[s] = gm < p.M ? (unsigned)gm * (unsigned)p.lda * 2u + (qq & 3) * 16u : OOB_OFF;     \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s) {                                                 \
            const int gn = col0_ + lr + 32 * s;                                                            \
            b_off[s] = gn < p.N ? (unsigned)gn * (unsigned)p.ldb * 4u + lq * 16u : OOB_OFF;                \
        }                                                                                                  \
    }
        PN_TILE()
        float4 ra0[AF32 ? A_IT : 1], ra1[AF32 ? A_IT : 1], rb0[B_IT], rb1[B_IT];
        u32x4 rq0[AQ_IT], rq1[AQ_IT];
#define PN_LOAD(RA, RB, RQ)                                                                                \
    {                                                                                                      \
        if (AF32) {                                                                                        \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s)                                               \
                RA[s] = buf_load4(ra_rsrc, a_off[s] != OOB_OFF ? a_off[s] + (unsigned)kc_ld * 128u : OOB_OFF); \
        } else {                                                                                           \
            _Pragma("unroll") for (int s = 0; s < AQ_IT; ++s)                                              \
                RQ[s] = __builtin_amdgcn_raw_buffer_load_b128(                                             \
                    ra_rsrc, aq_off[s] != OOB_OFF ? aq_off[s] + (unsigned)kc_ld * 64u : OOB_OFF, 0, 0);    \
        }                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                                   \
            RB[s] = buf_load4(rb_rsrc, b_off[s] != OOB_OFF ? b_off[s] + (unsigned)kc_ld * 128u : OOB_OFF); \
        if (++kc_ld == nk) { kc_ld = 0; q += S; PN_TILE() }                                                \
    }
#define PN_SPLIT_STORE(dst_, v_)                                                                           \
    {                                                                                                      \
        bf16x4 h_, m_, l_;                                                                                 \
        split3(v_, h_, m_, l_);                                                                            \
        *reinterpret_cast<bf16x4*>(dst_) = h_;                                                             \
        if (NT == 3) {                                                                                     \
            *reinterpret_cast<bf16x4*>((dst_) + 64) = m_;                                                  \
            *reinterpret_cast<bf16x4*>((dst_) + 128) = l_;                                                 \
        }                                                                                                  \
    }
#define PN_STORE(buf_, RA, RB, RQ)                                                                         \
    {                                                                                                      \
        unsigned char* const As_ = smem + (buf_) * STAGE;                                                  \
        unsigned char* const Bs_ = As_ + BM * PITCHB;                                                      \
        if (AF32) {                                                                                        \
            _Pragma("unroll") for (int s = 0; s < A_IT; ++s)                                               \
                PN_SPLIT_STORE(As_ + (lr + 32 * s) * PITCHB + lq * 8, RA[s])                               \
        } else {                                                                                           \
            _Pragma("unroll") for (int s = 0; s < AQ_IT; ++s) {                                            \
                const int qq = ptid + 256 * s;                                                             \
                *reinterpret_cast<u32x4*>(As_ + (qq >> 2) * PITCHB + (qq & 3) * 16) = RQ[s];               \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                                   \
            PN_SPLIT_STORE(Bs_ + (lr + 32 * s) * PITCHB + lq * 8, RB[s])                                   \
    }
        // K-step g of the stream lives in register set g & 1 and is loaded two barriers before it is stored
        PN_LOAD(ra0, rb0, rq0)
        if (T > 1) PN_LOAD(ra1, rb1, rq1)
        PN_STORE(0, ra0, rb0, rq0)
        if (T > 2) PN_LOAD(ra0, rb0, rq0)
        __syncthreads();
        for (int g = 0; g < T; g += 2) {
            if (g + 1 < T) {
                PN_STORE(1, ra1, rb1, rq1)
                if (g + 3 < T) PN_LOAD(ra1, rb1, rq1)
            }
            __syncthreads();
            if (g + 1 < T) {
                if (g + 2 < T) {
                    PN_STORE(0, ra0, rb0, rq0)
                    if (g + 4 < T) PN_LOAD(ra0, rb0, rq0)
                }
                __syncthreads();
            }
        }
#undef PN_TILE
#undef PN_LOAD
#undef PN_STORE
#undef PN_SPLIT_STORE
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.res ? p.res : p.C, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.C, p.aux ? p.aux_bytes : 0u);
    const bool has_res = p.res != nullptr;
    const int frag_off = (lane & 31) * PITCHB + (lane >> 5) * 16;
    // an element is addressed as (per-lane byte offset of the wave's first row) + (wave-uniform row offset in an SGPR):
    // no per-element address arithmetic; lanes of out-of-range COLUMNS carry OOB_OFF (dropped by the range check)
    const unsigned ldc_b = (unsigned)p.ldc * ES, ldres_b = (unsigned)p.ldres * ES, ldaux_b = (unsigned)p.ldaux * ES;
    int q = s0, g = 0;
    __syncthreads();
    for (int t = 0; t < n_my; ++t, q += S) {
        const int row0 = (xcd * tmx + q / tiles_n) * BM + wm * 128 + 4 * (lane >> 5);
        const int col0 = (q % tiles_n) * BN + wn * 64 + (lane & 31);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 32;
            const float bv = (MODE != 2 && p.bias && col < p.N) ? (col < p.bias_split ? p.bias[col] : p.bias2[col - p.bias_split]) : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
        }
        for (int kc = 0; kc < nk; ++kc, ++g) {
            const unsigned char* a_s = smem + (g & 1) * STAGE + wm * 128 * PITCHB + frag_off;
            const unsigned char* b_s = smem + (g & 1) * STAGE + BM * PITCHB + wn * 64 * PITCHB + frag_off;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[TM][NT], bf[TN][NT];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        af[i][u] = *reinterpret_cast<const bf16x8*>(a_s + i * 32 * PITCHB + u * 64 + ks * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        bf[j][u] = *reinterpret_cast<const bf16x8*>(b_s + j * 32 * PITCHB + u * 64 + ks * 32);
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
        // ---- epilogue (the producers are already staging the next tile).  Operand loads (residual / pre-activation) are
        // issued one 64-row-value column at a time and the second column's loads go out BEFORE the first column's stores,
        // so one load latency per tile is exposed and no load ever queues behind this tile's own stores.
        unsigned vc[TN], vx[TN], vr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 32;
            vc[j] = col < p.N ? (unsigned)row0 * ldc_b + (unsigned)col * ES : OOB_OFF;
            vx[j] = col < p.N ? (unsigned)row0 * ldaux_b + (unsigned)col * ES : OOB_OFF;
            vr[j] = col < p.N ? (unsigned)row0 * ldres_b + (unsigned)col * ES : OOB_OFF;
        }
#define PN_RD(i_, r_) ((unsigned)((i_) * 32 + ((r_) & 3) + 8 * ((r_) >> 2)))
#define PN_STORE_COL(j_)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r)          \
        st_so<TO>(rc, vc[j_], PN_RD(i, r) * ldc_b, acc[i][j_][r]);
        if ((MODE == 0 && has_res) || MODE == 2) {
            float ov[TM][16];
#define PN_LOAD_COL(j_)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r)          \
        ov[i][r] = MODE == 2 ? ld_so<TO>(raux, vx[j_], PN_RD(i, r) * ldaux_b)                              \
                             : ld_so<TO>(rres, vr[j_], PN_RD(i, r) * ldres_b);
#define PN_APPLY_COL(j_)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r)          \
        acc[i][j_][r] = MODE == 2 ? acc[i][j_][r] * (p.aux_deriv ? ov[i][r] : dgelu_f(ov[i][r])) : acc[i][j_][r] + ov[i][r];
            PN_LOAD_COL(0)
            PN_APPLY_COL(0)
            PN_LOAD_COL(1)
            PN_STORE_COL(0)
            PN_APPLY_COL(1)
            PN_STORE_COL(1)
#undef PN_LOAD_COL
#undef PN_APPLY_COL
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        st_so<TO>(raux, vx[j], PN_RD(i, r) * ldaux_b, p.aux_deriv ? dgelu_f(acc[i][j][r]) : acc[i][j][r]);      // zero-record descriptor when aux == NULL
                        st_so<TO>(rc, vc[j], PN_RD(i, r) * ldc_b, gelu_f(acc[i][j][r]));
                    }
        } else {
            PN_STORE_COL(0)
            PN_STORE_COL(1)
        }
#undef PN_STORE_COL
#undef PN_RD
    }
}

constexpr int PANEL_WGS_PER_XCD = 32;      // 256 CUs = 8 XCDs x 32: one persistent workgroup per CU

template <int NT, bool AF32, typename TO, int MODE>
int launch_panel_m(const KCParams& p, hipStream_t st) {
    constexpr int PITCHB = NT * 64 + 16;
    // >= 96 KB even when the stages are small, so that two workgroups never share a CU while another CU idles
    const int smem = 2 * 384 * PITCHB > 96 * 1024 ? 2 * 384 * PITCHB : 96 * 1024;
    {   // every launch: the attribute is per device, and the call is cheap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_kernel<NT, AF32, TO, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
    }
    const int tiles_m = p.M / 256, tiles_n = ceil_div(p.N, 128);
    const int tmx = ceil_div(tiles_m, 8);
    int S = tmx * tiles_n;
    if (S > PANEL_WGS_PER_XCD) S = PANEL_WGS_PER_XCD;
    hipLaunchKernelGGL((gemm_panel_kernel<NT, AF32, TO, MODE>), dim3(8 * S), dim3(512), smem, st, p, tmx, S);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int panel_mode(const KCParams& p) {
    const bool gelu = p.act == ACT_GELU, has_res = p.res != nullptr;
    if (p.epi == 0) return 0;
    if (gelu && !has_res && (p.epi == (EPI_ACT | EPI_STORE_PRE) || p.epi == EPI_ACT)) return 1;
    if (gelu && !has_res && p.epi == EPI_MUL_DACT && !p.bias) return 2;
    return -1;
}

template <int NT, bool AF32, typename TO>
int launch_panel_t(const KCParams& p, hipStream_t st) {
    switch (panel_mode(p)) {
        case 0: return launch_panel_m<NT, AF32, TO, 0>(p, st);
        case 1: return launch_panel_m<NT, AF32, TO, 1>(p, st);
        case 2: return launch_panel_m<NT, AF32, TO, 2>(p, st);
        default: return PA2D_ERR_ARG;
    }
}

}  // namespace

// Large-M plain GEMMs of the bf16 engines (fp32 storage: engines 1 / 2; bf16 storage: io_bf16).  PA2D_LIN_PANEL=off
// sends them back to the per-tile kernels (A/B measurements).
bool panel_applies(const KCParams& p, bool im2col) {
    if (!pa2d_env().lin_panel || im2col) return false;
    // (bf16 storage stays on the per-tile kernels: with one MFMA term and 2-byte operands the layer is a pure streaming
    //  problem and two small workgroups per CU keep more bytes in flight than one persistent one: measured 131 vs 138
    //  samples/s on the bf16-storage bench)
    if (p.io_bf16 || (p.engine != 1 && p.engine != 2)) return false;
    if ((p.K % 32) != 0 || p.N < 64) return false;
    if (p.io_bf16 && (p.lda % 8) != 0) return false;
    if (panel_mode(p) < 0) return false;
    if (p.epi == EPI_ACT && p.aux) return false;
    return (long long)(p.M / 256) * ceil_div(p.N, 128) >= 256;        // at least one tile per CU
}

// the first M - M % 256 rows (the caller runs the < 256-row tail on the per-tile kernels)

int launch_kc_panel(const KCParams& p_in, hipStream_t st) {
    KCParams p = p_in;
    if (p.io_bf16) {
        const unsigned long long ab = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2ull;      // A holds bf16
        if (ab >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.a_bytes = (unsigned)ab;
        return launch_panel_t<1, false, bf16_t>(p, st);
    }
    if (p.engine == 1) return launch_panel_t<3, true, float>(p, st);
    return launch_panel_t<1, true, float>(p, st);
}
