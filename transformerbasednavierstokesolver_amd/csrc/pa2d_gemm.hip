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
#include "pa2d_internal.h"
#include <stdlib.h>

#define KC_BK_SMALL_DEFAULT 1
#define EPI_ACT 1        // out = act(acc + bias)
#define EPI_STORE_PRE 2  // aux = acc + bias   (pre-activation, saved for backward)
#define EPI_MUL_DACT 4   // out = acc * act'(aux)

struct KCParams {
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    const float* bias;
    const float* bias2; int bias_split;   // columns >= bias_split take bias2[col - bias_split] (two stacked projections)
    const float* res; long long ldres;
    float* aux; long long ldaux;
    int M, N, K;
    int act, epi;
    int H, W, Cin;   // im2col view: A = image [B,H,W,Cin] with pixel pitch lda, K = 9*Cin
    unsigned a_bytes, b_bytes;   // extents of A and B for the buffer descriptors (filled by launch_kc)
    unsigned c_bytes, res_bytes, aux_bytes;
    int apre;   // split engine: A already holds the NT bf16 planes of every 32-channel chunk (split_planes_kernel)
};

__device__ __forceinline__ float gelu_f(float x) { return gelu_exact(x); }
__device__ __forceinline__ float dgelu_f(float x) { return dgelu_exact(x); }

// Branch-free epilogue of one 32x32 accumulator tile.  Ragged rows / columns are masked by the buffer
// range check (masked lanes get offset OOB_OFF: loads return 0, stores are dropped); all residual /
// pre-activation loads of the tile are issued before the first use.  ACT_ID < 0: runtime p.act.
template <bool HAS_RES, bool STORE_PRE, bool ACT, bool DACT, int ACT_ID>
__device__ __forceinline__ void kc_epilogue_tile(const KCParams& p, const f32x16& acc, int row_base, int col,
                                                 __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rres,
                                                 __amdgpu_buffer_rsrc_t raux) {
    const bool col_ok = col < p.N;
    const float bv = (p.bias && col_ok) ? (col < p.bias_split ? p.bias[col] : p.bias2[col - p.bias_split]) : 0.f;
    unsigned offc[16];
    float rv[16], av[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        const bool ok = col_ok && row < p.M;
        offc[r] = ok ? ((unsigned)row * (unsigned)p.ldc + (unsigned)col) * 4u : OOB_OFF;
        if (HAS_RES) rv[r] = buf_load1(rres, ok ? ((unsigned)row * (unsigned)p.ldres + (unsigned)col) * 4u : OOB_OFF);
        if (DACT) av[r] = buf_load1(raux, ok ? ((unsigned)row * (unsigned)p.ldaux + (unsigned)col) * 4u : OOB_OFF);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        float v = acc[r] + bv;
        if (STORE_PRE)
            buf_store1(raux, offc[r] != OOB_OFF ? ((unsigned)row * (unsigned)p.ldaux + (unsigned)col) * 4u : OOB_OFF, v);
        if (ACT) v = ACT_ID == ACT_GELU ? gelu_f(v) : act_fwd(p.act, v);
        if (DACT) v *= ACT_ID == ACT_GELU ? dgelu_f(av[r]) : act_bwd(p.act, av[r]);
        if (HAS_RES) v += rv[r];
        buf_store1(rc, offc[r], v);
    }
}

template <int TM, int TN>
__device__ __forceinline__ void kc_epilogue(const KCParams& p, f32x16 (&acc)[TM][TN], int row0, int col0) {
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.res ? p.res : p.C, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.C, p.aux ? p.aux_bytes : 0u);
    const bool has_res = p.res != nullptr;
    const bool gelu = p.act == ACT_GELU;
#define KC_EPI(HR, SP, AC, DA, ID)                                                                     \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i)      \
        kc_epilogue_tile<HR, SP, AC, DA, ID>(p, acc[i][j], row0 + i * 32, col0 + j * 32, rc, rres, raux);
    if (p.epi == 0) {
        if (has_res) { KC_EPI(true, false, false, false, 0) } else { KC_EPI(false, false, false, false, 0) }
    } else if (p.epi == (EPI_ACT | EPI_STORE_PRE) && gelu && !has_res) {
        KC_EPI(false, true, true, false, ACT_GELU)
    } else if (p.epi == EPI_ACT && gelu && !has_res) {          // inference: no pre-activation saved
        KC_EPI(false, false, true, false, ACT_GELU)
    } else if (p.epi == EPI_MUL_DACT && gelu && !has_res) {
        KC_EPI(false, false, false, true, ACT_GELU)
    } else {   // generic: any flag combination / activation (off the hot path)
        const bool sp = p.epi & EPI_STORE_PRE, ac = p.epi & EPI_ACT, da = p.epi & EPI_MUL_DACT;
        if (da) { if (has_res) { KC_EPI(true, false, false, true, -1) } else { KC_EPI(false, false, false, true, -1) } }
        else if (sp && ac) { if (has_res) { KC_EPI(true, true, true, false, -1) } else { KC_EPI(false, true, true, false, -1) } }
        else if (ac) { if (has_res) { KC_EPI(true, false, true, false, -1) } else { KC_EPI(false, false, true, false, -1) } }
        else { if (has_res) { KC_EPI(true, true, false, false, -1) } else { KC_EPI(false, true, false, false, -1) } }
    }
#undef KC_EPI
}

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
    kc_epilogue<TM, TN>(p, acc, tile_m * BM + wm * WM + 4 * (lane >> 5), tile_n * BN + wn * WN + (lane & 31));
}

// ---------------------------------------------------------------------------------------------
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)xs[i];
        const float r1 = xs[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
    }
}

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
template <int BM, int BN, bool IM2COL, int NT, bool APRE = false>
__global__ __launch_bounds__(512, 2) void gemm_kc_split_kernel(const KCParams p) {
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
            ap_off[s] = (APRE && gm < p.M) ? ((unsigned)gm * (unsigned)(nch_in * PIECES) + piece) * 16u : OOB_OFF;
            ap_lds[s] = row * PITCHB + piece * 16;
            const int n = APRE ? gm % (p.H * p.W) : 0;
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
        if (APRE) {                                                                                    \
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

    kc_epilogue<TM, TN>(p, acc, tile_m * BM + wm * WM + 4 * (lane >> 5), tile_n * BN + wn * WN + (lane & 31));
}

// engine selection: PA2D_GEMM=f32 (v_mfma_f32_32x32x2_f32, default) | split (6-term bf16 split, fp32
// accuracy, conv GEMMs only) | bf16 (1-term bf16 compute for every GEMM, fp32 accumulate/storage)
static int g_gemm_mode = -1;   // process-wide knob (pa2d_set_gemm_mode / env PA2D_GEMM)
static int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("PA2D_GEMM");
        g_gemm_mode = !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'b' ? 2 : 0));
    }
    return g_gemm_mode;
}
static int gemm_mode_split() { return gemm_mode() != 0; }
// chunk (channels per tap step) the conv weight packs must use for the selected engine
// the experimental split engine is only used for the conv implicit GEMMs (it is slower than the f32
// engine on the short-K linears)
// K-step of the big fp32 tiles: 32 by default (half the barriers, full 128-byte row segments, 72 KB of LDS
// -> 2 workgroups per CU; measured 3-4 % faster than 16 with 4 workgroups per CU: conv 2.61 -> 2.52 ms);
// PA2D_KC_BK=16 forces the 16-wide step.
static bool kc_bk32(bool im2col, int Cin) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PA2D_KC_BK"); v = (e && atoi(e) == 16) ? 0 : 1; }
    return v && (!im2col || (Cin % 32) == 0);
}
// K-step of the small tiles (128x64, 64x64: batch-1 rollout and small-batch training); PA2D_KC_BK_SMALL=16|32
static bool kc_bk32_small(bool im2col, int Cin) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PA2D_KC_BK_SMALL"); v = e ? (atoi(e) == 32) : KC_BK_SMALL_DEFAULT; }
    return v && (!im2col || (Cin % 32) == 0);
}
// tile choice of the fp32 engine: 128x128 when that already gives >= 1.5 workgroups per CU, otherwise smaller
// tiles so that small problems (rollout at batch 1: M = 4096) still fill the 256 CUs.  Shared by the launch and
// by the conv weight packs (the pack's channel chunk must equal the K-step).
struct KCTile { int bm, bn, bk; };
static KCTile kc_tile(int M, int N, bool im2col, int Cin) {
    const int tiles_m = ceil_div(M, 128);
    const long long t128 = (long long)tiles_m * ceil_div(N, 128);
    const long long t12864 = (long long)tiles_m * ceil_div(N, 64);
    if (N > 64 && t128 >= 384) return {128, 128, kc_bk32(im2col, Cin) ? 32 : 16};
    const int bk = kc_bk32_small(im2col, Cin) ? 32 : 16;
    if (t12864 >= 384 || M <= 64) return {128, 64, bk};
    return {64, 64, bk};
}
static bool use_split(int N, bool im2col, int Cin) {
    const int m = gemm_mode();
    if (m == 1) return im2col && N > 64 && (Cin % 32) == 0;
    if (m == 2) return N > 64 && (!im2col || (Cin % 32) == 0);
    return false;
}
static int conv_chunk(int Cin, int N) { return use_split(N, true, Cin) ? 32 : 16; }

static int launch_kc(const KCParams& p_in, bool im2col, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    KCParams p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return PA2D_OK;
    if (!p.bias2) p.bias_split = 0x7fffffff;
    {   // 32-bit byte offsets inside the buffer descriptors
        const unsigned long long ab = ((unsigned long long)(p.M - 1) * p.lda + (im2col ? p.Cin : p.K)) * 4ull;
        const unsigned long long bb = ((unsigned long long)(p.N - 1) * p.ldb + p.K) * 4ull;
        if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.a_bytes = (unsigned)ab;
        p.b_bytes = (unsigned)bb;
        const unsigned long long cb = ((unsigned long long)(p.M - 1) * p.ldc + p.N) * 4ull;
        const unsigned long long rb = p.res ? ((unsigned long long)(p.M - 1) * p.ldres + p.N) * 4ull : 0ull;
        const unsigned long long xb = p.aux ? ((unsigned long long)(p.M - 1) * p.ldaux + p.N) * 4ull : 0ull;
        if (cb >= 0xFFFFFFF0ull || rb >= 0xFFFFFFF0ull || xb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.c_bytes = (unsigned)cb; p.res_bytes = (unsigned)rb; p.aux_bytes = (unsigned)xb;
    }
    if ((p.K & 3) || (p.lda & 3) || (p.ldb & 3)) return PA2D_ERR_ARG;
    if (im2col && ((p.Cin & 15) || p.K != 9 * p.Cin)) return PA2D_ERR_UNSUPPORTED;
    const int tiles_m = ceil_div(p.M, 128);
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return PA2D_ERR_ARG;
    if (use_split(p.N, im2col, p.Cin)) {
        const int planes = gemm_mode() == 2 ? 1 : 3;
        if (im2col) p.b_bytes = (unsigned)((size_t)p.N * (p.K / 32) * planes * 64);
        if (p.apre) {
            if (!im2col) return PA2D_ERR_ARG;
            const unsigned long long pb = (unsigned long long)p.M * (p.Cin / 32) * planes * 64;
            if (pb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
            p.a_bytes = (unsigned)pb;
        }
        const int tiles_n = ceil_div(p.N, 128);
        const dim3 grid(ceil_div(tiles_m, 8) * 8 * tiles_n);
        const bool bf = gemm_mode() == 2;
        const int smem = 2 * (128 + 128) * (bf ? 80 : 208);
        static bool attr_done = false;
        if (!attr_done) {
            const int big = 2 * (128 + 128) * 208;
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<128, 128, true, 3>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, big);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<128, 128, false, 3>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, big);
            hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kc_split_kernel<128, 128, true, 3, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, big);
            if (e1 == hipSuccess) e1 = e3;
            if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
            attr_done = true;
        }
        if (p.apre) {
            if (bf) hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 1, true>), grid, dim3(512), smem, st, p);
            else hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 3, true>), grid, dim3(512), smem, st, p);
        } else if (bf) {
            if (im2col) hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 1>), grid, dim3(512), smem, st, p);
            else hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, false, 1>), grid, dim3(512), smem, st, p);
        } else {
            if (im2col) hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, true, 3>), grid, dim3(512), smem, st, p);
            else hipLaunchKernelGGL((gemm_kc_split_kernel<128, 128, false, 3>), grid, dim3(512), smem, st, p);
        }
    } else {
        const KCTile t = kc_tile(p.M, p.N, im2col, p.Cin);
        const int tiles_n = ceil_div(p.N, t.bn);
        const dim3 grid(ceil_div(ceil_div(p.M, t.bm), 8) * 8 * tiles_n);
#define KC_GO(BM_, BN_, WM_, WN_)                                                                            \
    {                                                                                                      \
        if (t.bk == 32) {                                                                                  \
            if (im2col) hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, true, 32>), grid, dim3(256), 0, st, p);  \
            else hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, false, 32>), grid, dim3(256), 0, st, p);        \
        } else {                                                                                           \
            if (im2col) hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, true, 16>), grid, dim3(256), 0, st, p);  \
            else hipLaunchKernelGGL((gemm_kc_kernel<BM_, BN_, WM_, WN_, false, 16>), grid, dim3(256), 0, st, p);        \
        }                                                                                                  \
    }
        if (t.bm == 128 && t.bn == 128) KC_GO(128, 128, 2, 2)
        else if (t.bm == 128) KC_GO(128, 64, 4, 1)
        else KC_GO(64, 64, 2, 2)
#undef KC_GO
    }
    PA2D_CHECK_LAUNCH();
    if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return PA2D_ERR_ARG;
    return PA2D_OK;
}

// ---------------------------------------------------------------------------------------------
struct MCParams {
    const float* A; long long lda; int Mi;
    const float* B; long long ldb; int Nj;
    float* slab;
    int Mk, chunks_per_split, splits;
    int H, W, Cin;   // im2col view of B: image [B,H,W,Cin] with pixel pitch ldb, j = tap*Cin + ci
    unsigned a_bytes, b_bytes;
};

// NT = 1: same staging (fp32 tiles [16 rows m][BM]), but each lane gathers its 8 consecutive m of one column
// with 8 ds_read_b32, rounds them to bf16 and issues ONE v_mfma_f32_32x32x16_bf16 per tile and 16-row chunk
// instead of 8 fp32 MFMAs (bf16-compute mode).  NT = 3: the gathered values are split exactly into hi+mid+lo
// bf16 terms in registers and the six products of order <= 2 are accumulated (fp32 accuracy, see
// gemm_kc_split_kernel): 6 bf16 MFMAs (192 cycles) instead of 8 fp32 MFMAs (512 cycles).  NT = 0: exact fp32.
template <int NT>
__device__ __forceinline__ void mc_split_elem(float v, bf16x8 (&pl)[NT], int e) {
    const __bf16 h = (__bf16)v;
    pl[0][e] = h;
    if constexpr (NT == 3) {
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        pl[1][e] = m;
        pl[2][e] = (__bf16)(r1 - (float)m);
    }
}
template <int BM, int BN, bool IM2COL, int NT, int BK = 16>
__global__ __launch_bounds__(256, 2) void gemm_mc_kernel(const MCParams p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_TPR = BM / 4, A_RPP = 256 / A_TPR, A_IT = BK / A_RPP;
    constexpr int B_TPR = BN / 4, B_RPP = 256 / B_TPR, B_IT = BK / B_RPP;
    static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    float* const As = smem;
    float* const Bs = smem + 2 * BK * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_i = (p.Mi + BM - 1) / BM, tiles_j = (p.Nj + BN - 1) / BN;
    // XCD-aware map (speed only): when the split count is a multiple of 8, blocks b, b+8, ... (one XCD
    // under round-robin dispatch) own a contiguous range of splits = a contiguous range of rows m, so
    // each XCD's L2 streams 1/8 of the operands instead of all of them.
    int tj, ti, split;
    {
        const int tiles = tiles_i * tiles_j;
        if ((p.splits & 7) == 0) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = p.splits >> 3;
            split = xcd * q + slot / tiles;
            const int t = slot % tiles;
            tj = t % tiles_j;
            ti = t / tiles_j;
        } else {
            tj = blockIdx.x % tiles_j;
            ti = (blockIdx.x / tiles_j) % tiles_i;
            split = blockIdx.x / tiles;
        }
    }
    const int wm = wave >> 1, wn = wave & 1;

    // the host plans in 16-row chunks; a K-step of BK rows covers BK/16 of them
    const int total_chunks = (p.Mk + BK - 1) / BK;
    const int cps = (p.chunks_per_split * 16 + BK - 1) / BK;
    const int c_begin = split * cps;
    const int c_end = min(total_chunks, c_begin + cps);

    const int a_r = tid / A_TPR, a_c = (tid % A_TPR) * 4;
    const int b_r = tid / B_TPR, b_c = (tid % B_TPR) * 4;
    const int gi = ti * BM + a_c;
    const int gj = tj * BN + b_c;
    const bool a_col_ok = gi < p.Mi, b_col_ok = gj < p.Nj;
    int tap_dy = 0, tap_dx = 0, ci = 0;
    if (IM2COL && b_col_ok) {
        const int tap = gj / p.Cin;
        ci = gj - tap * p.Cin;
        tap_dy = tap / 3 - 1;
        tap_dx = tap - (tap / 3) * 3 - 1;
    }
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t ra_rsrc = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb_rsrc = make_rsrc(p.B, p.b_bytes);
    const unsigned a_col = a_col_ok ? (unsigned)gi * 4u : OOB_OFF;
    const unsigned b_col = b_col_ok ? (unsigned)(IM2COL ? ci : gj) * 4u : OOB_OFF;
    const int tap_shift = tap_dy * p.W + tap_dx;
    float4 ra[A_IT], rb[B_IT];
#define MC_LOAD(c_)                                                                                       \
    {                                                                                                     \
        const int m0_ = (c_) * BK;                                                                        \
        _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                                \
            const int m_ = m0_ + a_r + s * A_RPP;                                                         \
            ra[s] = buf_load4(ra_rsrc, (a_col != OOB_OFF && m_ < p.Mk) ? (unsigned)m_ * (unsigned)p.lda * 4u + a_col : OOB_OFF); \
        }                                                                                                 \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s) {                                                \
            const int m_ = m0_ + b_r + s * B_RPP;                                                         \
            bool ok_ = b_col != OOB_OFF && m_ < p.Mk;                                                     \
            if (IM2COL) {                                                                                 \
                const int n_ = m_ % HW;                                                                   \
                const int y_ = n_ / p.W, x_ = n_ - y_ * p.W;                                              \
                ok_ = ok_ && (unsigned)(y_ + tap_dy) < (unsigned)p.H && (unsigned)(x_ + tap_dx) < (unsigned)p.W; \
                rb[s] = buf_load4(rb_rsrc, ok_ ? (unsigned)(m_ + tap_shift) * (unsigned)p.ldb * 4u + b_col : OOB_OFF); \
            } else {                                                                                      \
                rb[s] = buf_load4(rb_rsrc, ok_ ? (unsigned)m_ * (unsigned)p.ldb * 4u + b_col : OOB_OFF);  \
            }                                                                                             \
        }                                                                                                 \
    }
#define MC_STORE(buf_)                                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < A_IT; ++s)                                                  \
            *reinterpret_cast<float4*>(As + (buf_) * BK * BM + (a_r + s * A_RPP) * BM + a_c) = ra[s];     \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                                  \
            *reinterpret_cast<float4*>(Bs + (buf_) * BK * BN + (b_r + s * B_RPP) * BN + b_c) = rb[s];     \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        MC_LOAD(c_begin)
        MC_STORE(0)
    }
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        if (c + 1 < c_end) MC_LOAD(c + 1)
        const float* a_s = As + buf * BK * BM + kh * BM + wm * WM + li;
        const float* b_s = Bs + buf * BK * BN + kh * BN + wn * WN + li;
        if constexpr (NT > 0) {
            const float* a8 = As + buf * BK * BM + kh * 8 * BM + wm * WM + li;
            const float* b8 = Bs + buf * BK * BN + kh * 8 * BN + wn * WN + li;
            bf16x8 af[TM][NT], bf[TN][NT];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) mc_split_elem<NT>(a8[e * BM + i * 32], af[i], e);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) mc_split_elem<NT>(b8[e * BN + j * 32], bf[j], e);
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
        } else {
        // fragments of k-step kk+1 are fetched into the other register set before the MFMAs of
        // k-step kk issue, so the LDS latency hides behind 4 x 64 MFMA cycles
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = a_s[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = b_s[j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = a_s[(kk + 1) * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(kk + 1) & 1][j] = b_s[(kk + 1) * 2 * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
        }
        // pin the interleave: reads(0); { reads(kk+1); mfma(kk) } x 7; mfma(7)   (0x100 = DS read, 0x8 = MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2 - 1; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        if (c + 1 < c_end) MC_STORE(buf ^ 1)
        __syncthreads();
    }
#undef MC_LOAD
#undef MC_STORE

    float* out = p.slab + (size_t)split * p.Mi * p.Nj;
    const int col0 = tj * BN + wn * WN + (lane & 31);
    const int row0 = ti * BM + wm * WM + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.Nj) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.Mi) out[(size_t)row * p.Nj + col] = acc[i][j][r];
            }
    }
}

struct MCPlan { int big; int splits; int chunks_per_split; size_t slab_floats; };

static MCPlan plan_mc(int Mi, int Nj, int Mk) {
    MCPlan pl;
    if (Mk < 1) Mk = 1;       // empty contraction (batch 0): plan as one chunk, the entry points zero-fill instead
    pl.big = (Mi > 64 && Nj > 64) ? 1 : 0;
    const int bm = pl.big ? 128 : 64;
    const int tiles = ceil_div(Mi, bm) * ceil_div(Nj, bm);
    const int total_chunks = ceil_div(Mk, 16);
    // Split counts are multiples of 8 (one contiguous split range per XCD, see the kernel).  At most 4
    // workgroups per CU are resident, the rest run as slots free up, so the efficiency of a block
    // total is blocks / (256 * ceil(blocks/256)); take the smallest multiple of 8 that reaches
    // >= 0.97 with at least 2 workgroups per CU (fewer splits = less slab traffic; measured on the conv
    // weight gradient: 8/16/24/32 splits -> 3.63/3.02/2.84/2.75 ms).
    const int max_splits = total_chunks / 8 > 0 ? total_chunks / 8 : 1;   // >= 8 chunks per split
    int best = max_splits < 8 ? max_splits : 8;
    const char* env_s = getenv("PA2D_MC_SPLITS");                           // tuning knob
    if (env_s) {
        best = atoi(env_s);
    } else if (max_splits >= 8) {
        double best_eff = 0.0;
        for (int sp = 8; sp <= max_splits && sp <= 512; sp += 8) {
            const int blocks = tiles * sp;
            if (blocks < 512 && sp + 8 <= max_splits) continue;
            const double eff = (double)blocks / (256.0 * ceil_div(blocks, 256));
            if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
            if (eff >= 0.97) { best = sp; break; }
        }
    }
    if (best < 1) best = 1;
    if (best > max_splits) best = max_splits;
    pl.chunks_per_split = ceil_div(total_chunks, best);
    pl.splits = ceil_div(total_chunks, pl.chunks_per_split);
    pl.slab_floats = (size_t)pl.splits * Mi * Nj;
    return pl;
}

// out[idx] = sum_s slab[s][idx]; mode 1 additionally un-packs the conv weight gradient:
// slab row-major [2C][9][Cin] -> dWx / dWf in the reference's [C_out][C_in][3][3] layout.
// 64 consecutive idx x 4 slab lanes per workgroup: each lane sums slabs s = lane, lane+4, ... with
// 4 independent loads in flight; the 4 lane sums are added in fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int nslab, long long count,
                                                           float* __restrict__ out, float* __restrict__ out2,
                                                           int mode, int C, int Cin) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < count) {
        const float* p = slab + idx;
        int k = ty;
        for (; k + 12 < nslab; k += 16) {
            s0 += p[(size_t)k * count];
            s1 += p[(size_t)(k + 4) * count];
            s2 += p[(size_t)(k + 8) * count];
            s3 += p[(size_t)(k + 12) * count];
        }
        for (; k < nslab; k += 4) s0 += p[(size_t)k * count];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty != 0 || idx >= count) return;
    const float s = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
    if (mode == 0) {
        out[idx] = s;
    } else if (mode == 2) {      // one vector split over two outputs at element C
        if (idx < C) out[idx] = s; else out2[idx - C] = s;
    } else {
        const int ci = (int)(idx % Cin);
        const int tap = (int)((idx / Cin) % 9);
        const int co = (int)(idx / ((long long)Cin * 9));
        float* dst = co < C ? out : out2;
        dst[((size_t)(co % C) * Cin + ci) * 9 + tap] = s;
    }
}

static int launch_mc(const float* A, long long lda, int Mi, const float* B, long long ldb, int Nj, int Mk,
                     bool im2col, int H, int W, int Cin, float* slab, const MCPlan& pl, hipStream_t st) {
    if ((Mi & 3) || (Nj & 3) || (lda & 3) || (ldb & 3)) return PA2D_ERR_ARG;
    if (im2col && (Cin & 3)) return PA2D_ERR_UNSUPPORTED;
    MCParams p;
    p.A = A; p.lda = lda; p.Mi = Mi; p.B = B; p.ldb = ldb; p.Nj = Nj; p.slab = slab; p.Mk = Mk;
    p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits; p.H = H; p.W = W; p.Cin = Cin;
    {
        const unsigned long long ab = ((unsigned long long)(Mk - 1) * lda + Mi) * 4ull;
        const unsigned long long bb = ((unsigned long long)(Mk - 1) * ldb + (im2col ? Cin : Nj)) * 4ull;
        if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.a_bytes = (unsigned)ab;
        p.b_bytes = (unsigned)bb;
    }
    const int bm = pl.big ? 128 : 64;
    const dim3 grid(ceil_div(Mi, bm) * ceil_div(Nj, bm) * pl.splits);
    const bool bf = gemm_mode() == 2;
    static int mc_bk = -1;
    if (mc_bk < 0) { const char* e = getenv("PA2D_MC_BK"); mc_bk = (e && atoi(e) == 32) ? 32 : 16; }
    if (pl.big && !bf && mc_bk == 32 && (pl.chunks_per_split % 2) == 0) {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 0, 32>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<128, 128, false, 0, 32>), grid, dim3(256), 0, st, p);
    } else if (pl.big && im2col && gemm_mode() == 1 && (Cin % 32) == 0) {     // 6-term split, fp32 accuracy
        hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 3>), grid, dim3(256), 0, st, p);
    } else if (pl.big && bf) {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<128, 128, false, 1>), grid, dim3(256), 0, st, p);
    } else if (pl.big) {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 0>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<128, 128, false, 0>), grid, dim3(256), 0, st, p);
    } else {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<64, 64, true, 0>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<64, 64, false, 0>), grid, dim3(256), 0, st, p);
    }
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

static int launch_reduce(const float* slab, int nslab, long long count, float* out, float* out2, int mode,
                         int C, int Cin, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div_ll(count, 64));
    hipLaunchKernelGGL(reduce_slabs_kernel, grid, dim3(256), 0, st, slab, nslab, count, out, out2, mode, C, Cin);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st) {
    return launch_reduce(slab, nslab, count, out, nullptr, 0, 0, 0, st);
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): partial[blk][n] over row blocks, then reduce_slabs.
__global__ void colsum_partial_kernel(const float* __restrict__ X, long long ld, int M, int N, int rows_per_block,
                                      float* __restrict__ partial) {
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            s0 += X[(size_t)r * ld + c];
            s1 += X[(size_t)(r + 1) * ld + c];
            s2 += X[(size_t)(r + 2) * ld + c];
            s3 += X[(size_t)(r + 3) * ld + c];
        }
        for (; r < r1; ++r) s0 += X[(size_t)r * ld + c];
        partial[(size_t)blockIdx.x * N + c] = (s0 + s1) + (s2 + s3);
    }
}

static int colsum_blocks(int M) { int b = ceil_div(M, 128); return b > 1024 ? 1024 : (b < 1 ? 1 : b); }

// out2 != NULL: columns >= split go to out2[col - split]
static int launch_colsum(const float* X, long long ld, int M, int N, float* out, float* partial, hipStream_t st,
                         float* out2 = nullptr, int split = 0) {
    const int nb = colsum_blocks(M);
    const int rpb = ceil_div(M, nb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, X, ld, M, N, rpb, partial);
    PA2D_CHECK_LAUNCH();
    return launch_reduce(partial, nb, N, out, out2, out2 ? 2 : 0, split, 0, st);
}

// ---------------------------------------------------------------------------------------------
// weight re-layouts (tiny; weights change every optimizer step so they are redone per call)
// mode 0: plain transpose  dst[k][n] = src[n][k]                                  (linear bwd-data)
// mode 1: conv fwd pack    dst[co'][cic][tap][c16] = W_{co'<C ? x : f}[co][cic*16+c16][tap]       ([2C][9*Cin])
// mode 2: conv bwd pack    dst[ci][cic'][tap'][c16] = W_{..}[co = cic'*16+c16][ci][8 - tap']        ([Cin][9*2C])
//         (K order of gemm_kc's implicit GEMM: 16-channel chunk outer, tap inner)
__global__ void repack_kernel(const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ dst,
                              int mode, int N, int K, int C, int Cin) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) {
        if (idx >= (long long)N * K) return;
        const int n = (int)(idx % N), k = (int)(idx / N);
        dst[idx] = w0[(size_t)n * K + k];
    } else if (mode == 1) {
        if (idx >= (long long)2 * C * 9 * Cin) return;
        const int CH = K;      // channels per K-step (16 for the f32 engine, 32 for the split engine)
        const int c16 = (int)(idx % CH);
        const int tap = (int)((idx / CH) % 9);
        const int cic = (int)((idx / (9 * CH)) % (Cin / CH));
        const int co = (int)(idx / ((long long)Cin * 9));
        const int ci = cic * CH + c16;
        const float* src = co < C ? w0 : w1;
        dst[idx] = src[((size_t)(co % C) * Cin + ci) * 9 + tap];
    } else {
        if (idx >= (long long)2 * C * 9 * Cin) return;
        const int CH = K;
        const int c16 = (int)(idx % CH);
        const int tap = (int)((idx / CH) % 9);
        const int cic = (int)((idx / (9 * CH)) % (2 * C / CH));
        const int ci = (int)(idx / ((long long)2 * C * 9));
        const int co = cic * CH + c16;
        const float* src = co < C ? w0 : w1;
        dst[idx] = src[((size_t)(co % C) * Cin + ci) * 9 + (8 - tap)];
    }
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
static size_t planes_bytes(long long rows, int C, int NT) { return (size_t)rows * C * NT * 2; }
static int launch_split_planes(const float* src, long long ld, void* dst, long long rows, int C, int NT, hipStream_t st) {
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

static int launch_repack(const float* w0, const float* w1, float* dst, int mode, int N, int K, int C, int Cin,
                         hipStream_t st) {
    const long long count = mode == 0 ? (long long)N * K : (long long)2 * C * 9 * Cin;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)ceil_div_ll(count, 256)), dim3(256), 0, st, w0, w1, dst, mode,
                       N, K, C, Cin);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// =============================================================================================
// C ABI (declared in include/pa2d.h)
extern "C" {

// 0 = exact fp32 MFMA engine (default), 1 = experimental 6-term bf16-split engine for the conv GEMMs
void pa2d_set_gemm_mode(int mode) { g_gemm_mode = (mode == 1 || mode == 2) ? mode : 0; }
int pa2d_get_gemm_mode(void) { return gemm_mode(); }

int pa2d_gemm_bias_act_fwd(const float* x, long long ldx, const float* w, long long ldw, const float* bias,
                           const float* res, long long ldres, float* y, long long ldy, float* pre, long long ldpre,
                           int M, int N, int K, int act, hipStream_t st) {
    KCParams p = {};
    p.A = x; p.lda = ldx; p.B = w; p.ldb = ldw; p.C = y; p.ldc = ldy; p.bias = bias; p.res = res; p.ldres = ldres;
    p.aux = pre; p.ldaux = ldpre; p.M = M; p.N = N; p.K = K; p.act = act;
    p.epi = (act != ACT_NONE ? EPI_ACT : 0) | (pre ? EPI_STORE_PRE : 0);
    return launch_kc(p, false, st);
}

// dx[M,K] = (dy[M,N] . w[N,K]) * act'(pre[M,K])   (pre may be NULL -> plain product)
// wt_ws: K*N floats of scratch for the transposed weight.
int pa2d_gemm_bwd_data(const float* dy, long long lddy, const float* w, long long ldw, const float* pre,
                       long long ldpre, int act, float* dx, long long lddx, float* wt_ws, int M, int N, int K,
                       hipStream_t st) {
    if (ldw != K) return PA2D_ERR_ARG;
    if (N & 3) return PA2D_ERR_ARG;
    if (M <= 0) return PA2D_OK;
    int rc = launch_repack(w, nullptr, wt_ws, 0, N, K, 0, 0, st);
    if (rc) return rc;
    KCParams p = {};
    p.A = dy; p.lda = lddy; p.B = wt_ws; p.ldb = N; p.C = dx; p.ldc = lddx; p.M = M; p.N = K; p.K = N;
    p.aux = const_cast<float*>(pre); p.ldaux = ldpre; p.act = act;
    p.epi = (pre && act != ACT_NONE) ? EPI_MUL_DACT : 0;
    return launch_kc(p, false, st);
}

size_t pa2d_gemm_bwd_weight_workspace(int M, int N, int K) {
    const MCPlan pl = plan_mc(N, K, M);
    size_t a = pl.slab_floats, b = (size_t)colsum_blocks(M) * N;
    return (a > b ? a : b) * sizeof(float);
}

// dw[N,K] = dy[M,N]^T . x[M,K] ; db[N] = column sums of dy (db may be NULL)
int pa2d_gemm_bwd_weight(const float* dy, long long lddy, const float* x, long long ldx, float* dw, float* db,
                         void* ws, size_t ws_bytes, int M, int N, int K, hipStream_t st) {
    if (M <= 0) { const int rz = pa2d_zero(dw, sizeof(float) * N * K, st); return rz ? rz : pa2d_zero(db, sizeof(float) * N, st); }
    if (ws_bytes < pa2d_gemm_bwd_weight_workspace(M, N, K)) return PA2D_ERR_WORKSPACE;
    const MCPlan pl = plan_mc(N, K, M);
    int rc = launch_mc(dy, lddy, N, x, ldx, K, M, false, 0, 0, 0, (float*)ws, pl, st);
    if (rc) return rc;
    rc = launch_reduce((const float*)ws, pl.splits, (long long)N * K, dw, nullptr, 0, 0, 0, st);
    if (rc) return rc;
    if (db) rc = launch_colsum(dy, lddy, M, N, db, (float*)ws, st);
    return rc;
}

// bf16 engines: bytes of the pre-split activation planes of a [rows, Cin] operand (0 when the engine selected for
// this GEMM reads fp32 operands)
static size_t conv_planes_bytes(int M, int N, int Cin) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PA2D_SPLIT_APRE"); on = e ? atoi(e) : 1; }
    if (!on || !use_split(N, true, Cin)) return 0;
    return (planes_bytes(M, Cin, gemm_mode() == 2 ? 1 : 3) + 255) & ~(size_t)255;
}

// backward workspace: [weight pack | slabs or column-sum partials | activation planes (bf16 engines)]
size_t pa2d_conv3x3x2_workspace(int B, int H, int W, int C) {
    const size_t pack = (size_t)3 * C * 9 * C;      // fp32 pack (2C*9C floats) or 3 bf16 planes (1.5x)
    const MCPlan pl = plan_mc(2 * C, 9 * C, B * H * W);
    size_t sl = pl.slab_floats, cs = (size_t)colsum_blocks(B * H * W) * 2 * C;
    return (pack + (sl > cs ? sl : cs)) * sizeof(float) + conv_planes_bytes(B * H * W, C, 2 * C);
}

// forward workspace: [weight pack (unused if prepacked) | activation planes (bf16 engines)]
size_t pa2d_conv3x3x2_fwd_workspace(int B, int H, int W, int C) {
    return (size_t)3 * C * 9 * C * sizeof(float) + conv_planes_bytes(B * H * W, 2 * C, C);
}

// Packed conv weights in the layout the engine selected for these dims wants (channel chunk = K-step of the
// tile, fp32 or bf16 planes by GEMM mode).  direction 0: forward pack ([2C][9C]); 1: data-gradient pack
// ([C][9*2C], taps flipped).  pack: pa2d_conv3x3x2_pack_bytes(C) bytes.  A pack stays valid while the weights,
// the dims and the GEMM mode do not change.
static int conv_pack(const float* wx, const float* wf, float* pack, int M, int C, int direction, hipStream_t st) {
    const int N = direction ? C : 2 * C, Cin = direction ? 2 * C : C;
    if (use_split(N, true, Cin)) {
        hipLaunchKernelGGL(repack_split_kernel, dim3((unsigned)ceil_div_ll((long long)2 * C * 9 * C, 256)), dim3(256), 0, st,
                           wx, wf, (__bf16*)pack, direction, gemm_mode() == 2 ? 1 : 3, C, C);
        PA2D_CHECK_LAUNCH();
        return PA2D_OK;
    }
    return launch_repack(wx, wf, pack, direction ? 2 : 1, 0, kc_tile(M, N, true, Cin).bk, C, C, st);
}

size_t pa2d_conv3x3x2_pack_bytes(int C) { return (size_t)3 * C * 9 * C * sizeof(float); }

int pa2d_conv3x3x2_pack(const float* wx, const float* wf, void* pack, size_t pack_bytes, int B, int H, int W, int C,
                        int direction, hipStream_t st) {
    if (pack_bytes < pa2d_conv3x3x2_pack_bytes(C)) return PA2D_ERR_WORKSPACE;
    return conv_pack(wx, wf, (float*)pack, B * H * W, C, direction ? 1 : 0, st);
}

// out[B*H*W, 2C] = [conv3x3(xn, wx) + bx | conv3x3(xn, wf) + bf]   (zero padding 1, NHWC)
// Physics_Attention.py:94,96 — both projections read the same input, so they run as ONE implicit
// GEMM [B*N, 9C] x [9C, 2C].  prepacked: NULL (weights are packed into ws by this call) or a pack made by
// pa2d_conv3x3x2_pack(direction 0) for the same B, H, W, C.
int pa2d_conv3x3x2_fwd(const float* xn, const float* wx, const float* bx, const float* wf, const float* bf,
                       float* out, const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                       hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (B <= 0) return PA2D_OK;
    if (ws_bytes < pa2d_conv3x3x2_fwd_workspace(B, H, W, C)) return PA2D_ERR_WORKSPACE;
    const float* pack = (const float*)prepacked;
    if (!pack) {
        const int rc = conv_pack(wx, wf, (float*)ws, B * H * W, C, 0, st);
        if (rc) return rc;
        pack = (const float*)ws;
    }
    const size_t apl = conv_planes_bytes(B * H * W, 2 * C, C);
    void* const planes = (char*)ws + pa2d_conv3x3x2_pack_bytes(C);
    if (apl) {
        const int rc = launch_split_planes(xn, C, planes, (long long)B * H * W, C, gemm_mode() == 2 ? 1 : 3, st);
        if (rc) return rc;
    }
    KCParams p = {};
    p.A = apl ? (const float*)planes : xn; p.apre = apl ? 1 : 0;
    p.lda = C; p.B = pack; p.ldb = 9 * C; p.C = out; p.ldc = 2 * C;
    p.bias = bx; p.bias2 = bf; p.bias_split = C;
    p.M = B * H * W; p.N = 2 * C; p.K = 9 * C; p.H = H; p.W = W; p.Cin = C;
    return launch_kc(p, true, st, ev_start, ev_stop);
}

// dxn[B*N, C] (+= nothing; plain store), dwx/dwf [C,C,3,3], dbx/dbf [C]  from dout[B*N, 2C]
int pa2d_conv3x3x2_bwd(const float* dout, const float* xn, const float* wx, const float* wf, float* dxn, float* dwx,
                       float* dbx, float* dwf, float* dbf, const void* prepacked, void* ws, size_t ws_bytes, int B,
                       int H, int W, int C, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (B <= 0) {
        const size_t wb = sizeof(float) * (size_t)C * C * 9, bb = sizeof(float) * C;
        int rz = pa2d_zero(dwx, wb, st);
        if (!rz) rz = pa2d_zero(dwf, wb, st);
        if (!rz) rz = pa2d_zero(dbx, bb, st);
        return rz ? rz : pa2d_zero(dbf, bb, st);
    }
    if (ws_bytes < pa2d_conv3x3x2_workspace(B, H, W, C)) return PA2D_ERR_WORKSPACE;
    float* scratch = (float*)ws + (size_t)3 * C * 9 * C;
    const int M = B * H * W;
    int rc;
    if (dxn) {
        const float* pack = (const float*)prepacked;
        if (!pack) {
            rc = conv_pack(wx, wf, (float*)ws, M, C, 1, st);
            if (rc) return rc;
            pack = (const float*)ws;
        }
        const size_t apl = conv_planes_bytes(M, C, 2 * C);
        void* const planes = (char*)ws + pa2d_conv3x3x2_workspace(B, H, W, C) - apl;
        if (apl) {
            rc = launch_split_planes(dout, 2 * C, planes, M, 2 * C, gemm_mode() == 2 ? 1 : 3, st);
            if (rc) return rc;
        }
        KCParams p = {};
        p.A = apl ? (const float*)planes : dout; p.apre = apl ? 1 : 0;
        p.lda = 2 * C; p.B = pack; p.ldb = 9 * 2 * C; p.C = dxn; p.ldc = C;
        p.M = M; p.N = C; p.K = 9 * 2 * C; p.H = H; p.W = W; p.Cin = 2 * C;
        rc = launch_kc(p, true, st, ev_start, ev_stop);
        if (rc) return rc;
    }
    const MCPlan pl = plan_mc(2 * C, 9 * C, M);
    rc = launch_mc(dout, 2 * C, 2 * C, xn, C, 9 * C, M, true, H, W, C, scratch, pl, st);
    if (rc) return rc;
    rc = launch_reduce(scratch, pl.splits, (long long)2 * C * 9 * C, dwx, dwf, 1, C, C, st);
    if (rc) return rc;
    return launch_colsum(dout, 2 * C, M, 2 * C, dbx, scratch, st, dbf, C);
}

}  // extern "C"
