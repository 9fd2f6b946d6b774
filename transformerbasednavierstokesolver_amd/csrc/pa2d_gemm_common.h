// Internal declarations shared by the GEMM translation units of libpa2d (pa2d_gemm.hip = engine selection + C ABI,
// pa2d_gemm_kc.hip = exact fp32 engine, pa2d_gemm_split.hip = bf16 split / bf16 compute engines,
// pa2d_gemm_mc.hip = weight-gradient engine and the deterministic reductions).  Split only so that the
// translation units compile in parallel; nothing here is part of the C ABI.
#pragma once
#include "pa2d_internal.h"
#include <stdlib.h>

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
    int engine; // PA2D_ENGINE_* of this call (explicit per call: the library keeps no engine state)
    int aux_deriv; // PA2D_ACT_SAVE_DERIVATIVE: aux receives / holds act'(pre-activation) instead of the pre-activation
    void* wimg;  // scratch for the weight plane image of the row-stationary linear kernel (rowpanel_image_bytes), or NULL
    const float* wsrc; long long wsn, wsk;   // the image's source: B[n][k] = wsrc[n * wsn + k * wsk]
    int io_bf16; // bf16-storage entry points: A (row-major [M][K] or the NHWC image), C, res and aux hold bf16; lda / ldc /
                 // ldres / ldaux stay in ELEMENTS; a bf16 A is read as pre-made 1-plane "planes" (apre = 1)
};

__device__ __forceinline__ float gelu_f(float x) { return gelu_exact(x); }
__device__ __forceinline__ float dgelu_f(float x) { return dgelu_exact(x); }

// Branch-free epilogue of one 32x32 accumulator tile.  Ragged rows / columns are masked by the buffer
// range check (masked lanes get offset OOB_OFF: loads return 0, stores are dropped); all residual /
// pre-activation loads of the tile are issued before the first use.  ACT_ID < 0: runtime p.act.
template <bool HAS_RES, bool STORE_PRE, bool ACT, bool DACT, int ACT_ID, typename TO = float>
__device__ __forceinline__ void kc_epilogue_tile(const KCParams& p, const f32x16& acc, int row_base, int col,
                                                 __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rres,
                                                 __amdgpu_buffer_rsrc_t raux) {
    constexpr unsigned ES = Act<TO>::ES;
    const bool col_ok = col < p.N;
    const float bv = (p.bias && col_ok) ? (col < p.bias_split ? p.bias[col] : p.bias2[col - p.bias_split]) : 0.f;
    unsigned offc[16];
    float rv[16], av[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        const bool ok = col_ok && row < p.M;
        offc[r] = ok ? ((unsigned)row * (unsigned)p.ldc + (unsigned)col) * ES : OOB_OFF;
        if (HAS_RES) rv[r] = Act<TO>::bld1(rres, ok ? ((unsigned)row * (unsigned)p.ldres + (unsigned)col) * ES : OOB_OFF);
        if (DACT) av[r] = Act<TO>::bld1(raux, ok ? ((unsigned)row * (unsigned)p.ldaux + (unsigned)col) * ES : OOB_OFF);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        float v = acc[r] + bv;
        if (STORE_PRE)
            Act<TO>::bst1(raux, offc[r] != OOB_OFF ? ((unsigned)row * (unsigned)p.ldaux + (unsigned)col) * ES : OOB_OFF,
                          p.aux_deriv ? (ACT_ID == ACT_GELU ? dgelu_f(v) : act_bwd(p.act, v)) : v);
        if (ACT) v = ACT_ID == ACT_GELU ? gelu_f(v) : act_fwd(p.act, v);
        if (DACT) v *= p.aux_deriv ? av[r] : (ACT_ID == ACT_GELU ? dgelu_f(av[r]) : act_bwd(p.act, av[r]));
        if (HAS_RES) v += rv[r];
        Act<TO>::bst1(rc, offc[r], v);
    }
}

// BIAS_ONLY: the conv implicit GEMMs only ever store acc (+ bias): one variant instead of nine (compile time)
template <int TM, int TN, bool BIAS_ONLY = false, typename TO = float>
__device__ __forceinline__ void kc_epilogue(const KCParams& p, f32x16 (&acc)[TM][TN], int row0, int col0) {
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.res ? p.res : p.C, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.C, p.aux ? p.aux_bytes : 0u);
    const bool has_res = p.res != nullptr;
    const bool gelu = p.act == ACT_GELU;
#define KC_EPI(HR, SP, AC, DA, ID)                                                                     \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i)      \
        kc_epilogue_tile<HR, SP, AC, DA, ID, TO>(p, acc[i][j], row0 + i * 32, col0 + j * 32, rc, rres, raux);
    if constexpr (BIAS_ONLY) {
        KC_EPI(false, false, false, false, 0)
    } else if (p.epi == 0) {
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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Exact split x = hi + mid + lo (up to 2^-25 |x|) of four floats into three bf16 planes.  Written on the PACKED conversion
// result: the straightforward scalar form makes hipcc convert every element a second time on its own to build the residual
// (7.5 instructions per element); per pair this is 3 v_cvt_pk_bf16_f32 + 2 x (v_lshlrev, v_and, v_pk_add_f32) = 4.5.
__device__ __forceinline__ void split3(const float4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ a[2] = {{v.x, v.y}, {v.z, v.w}};
    u32x2_ ph, pm, pl;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(a[q], bf16x2_));
        const f32x2_ hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
        const f32x2_ r1 = a[q] - hf;
        const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_));
        const f32x2_ mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
        const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r1 - mf, bf16x2_));
        ph[q] = h; pm[q] = m; pl[q] = l;
    }
    hi = __builtin_bit_cast(bf16x4, ph);
    mid = __builtin_bit_cast(bf16x4, pm);
    lo = __builtin_bit_cast(bf16x4, pl);
}

struct MCParams {
    const float* A; long long lda; int Mi;
    const float* B; long long ldb; int Nj;
    float* slab;
    int Mk, chunks_per_split, splits;
    int H, W, Cin;   // im2col view of B: image [B,H,W,Cin] with pixel pitch ldb, j = tap*Cin + ci
    unsigned a_bytes, b_bytes;
    float* colsum;   // optional [splits][Mi]: per-split column sums of A (= the bias gradient of a linear layer), produced
                     // by the workgroups of column tile 0 from the A tiles they stage anyway; NULL = off
};
struct MCPlan { int big; int splits; int chunks_per_split; size_t slab_floats; };
struct KCTile { int bm, bn, bk; };

// engine selection / tile choice (pa2d_gemm.hip)
bool use_split(int engine, int N, bool im2col, int Cin);
KCTile kc_tile(int M, int N, bool im2col, int Cin);
// exact fp32 engine (pa2d_gemm_kc.hip): launches the tile variant `t` on a filled-in KCParams
int launch_kc_f32(const KCParams& p, bool im2col, const KCTile& t, hipStream_t st);
// bf16 engines (pa2d_gemm_split.hip)
int launch_kc_split(KCParams& p, bool im2col, hipStream_t st);
bool kc_split_small_applies(const KCParams& p, bool im2col);
int launch_kc_split_small(const KCParams& p, hipStream_t st);
size_t planes_bytes(long long rows, int C, int NT);
int launch_split_planes(const float* src, long long ld, void* dst, long long rows, int C, int NT, hipStream_t st);
int launch_repack_split(const float* w0, const float* w1, void* dst, int bwd, int NT, int C, int Cin, hipStream_t st);
// persistent row-panel kernel for large-M plain GEMMs of the bf16 engines (pa2d_gemm_panel.hip)
bool panel_applies(const KCParams& p, bool im2col);
int launch_kc_panel(const KCParams& p, hipStream_t st);
// row-stationary split-engine linears (pa2d_gemm_rowpanel.hip): need a weight plane image in caller scratch
size_t rowpanel_image_bytes(int N, int K, int engine);
bool rowpanel_applies(const KCParams& p);
int launch_pack_weight_image(const float* w, long long sn, long long sk, void* img, int N, int K, hipStream_t st);
int launch_kc_rowpanel(const KCParams& p, const void* img, hipStream_t st);
// conv with the halo tile resident in LDS (pa2d_conv_halo.hip): used by launch_kc_split for pre-split im2col operands
bool conv_halo_applies(const KCParams& p);
int launch_conv_halo(const KCParams& p, hipStream_t st);
// weight-gradient engine and reductions (pa2d_gemm_mc.hip)
MCPlan plan_mc(int Mi, int Nj, int Mk);
int launch_mc(const float* A, long long lda, int Mi, const float* B, long long ldb, int Nj, int Mk, bool im2col,
              int H, int W, int Cin, float* slab, const MCPlan& pl, int engine, hipStream_t st, float* colsum = nullptr);
// accumulate != 0: out += sum of slabs (gradient accumulation straight into the caller's buffer)
int launch_reduce(const float* slab, int nslab, long long count, float* out, float* out2, int mode, int C, int Cin,
                  hipStream_t st, int accumulate = 0);
int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st);
// weight gradient from pre-split planes (pa2d_gemm_mc_planes.hip)
bool mc_planes_supported(int C, int Cin);
int launch_mc_planes(const void* PA, const void* PB, int C, int Cin, int Mk, int H, int W, float* slab,
                     const MCPlan& pl, int NT, hipStream_t st);
// 256 x 256-tile variant (8 waves, one round of workgroups) and its own split plan
bool mc_planes_big_applies(int C, int Cin, int Mk);
MCPlan plan_mc_planes_big(int Mi, int Nj, int Mk);
int launch_mc_planes_big(const void* PA, const void* PB, int C, int Cin, int Mk, int H, int W, float* slab,
                         const MCPlan& pl, int NT, hipStream_t st);
int launch_mc_planes_big_raw(const void* PA, int Mi, const void* PB, int Cin, int taps, int Mk, int H, int W, float* slab,
                             const MCPlan& pl, int NT, hipStream_t st);
// plain dW of a linear layer straight from fp32 operands on the 128 x 128 planes kernel (split while staging)
bool mc_f32src_applies(int Mi, int Nj, int Mk, long long lda, long long ldb);
int launch_mc_f32src(const float* A, long long lda, int Mi, const float* B, long long ldb, int Nj, int Mk, float* slab,
                     const MCPlan& pl, int NT, float* colsum, hipStream_t st);
int colsum_blocks(int M);
int launch_colsum_bf16(const void* X, long long ld, int M, int N, float* out, float* partial, hipStream_t st,
                       float* out2 = nullptr, int split = 0, int accumulate = 0);
int launch_colsum(const float* X, long long ld, int M, int N, float* out, float* partial, hipStream_t st,
                  float* out2 = nullptr, int split = 0, int accumulate = 0);
