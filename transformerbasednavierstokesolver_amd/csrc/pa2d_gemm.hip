// Engine selection, weight re-layouts and the C ABI of the dense (GEMM / conv) stages.  Kernels live in
// pa2d_gemm_kc.hip (exact fp32), pa2d_gemm_split.hip (bf16 engines) and pa2d_gemm_mc.hip (weight gradients).
#include "pa2d_gemm_common.h"
#include <string.h>

// Engines (explicit `engine` argument of every dense entry point; the library keeps NO engine state, so two
// models in one process can use different engines): 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = 6-term bf16
// split at fp32 accuracy (conv GEMMs and large-M plain GEMMs; small GEMMs stay exact), 2 = bf16 compute for every GEMM (fp32 accumulate
// and storage).  pa2d_default_engine() only reads the environment (PA2D_GEMM=f32|split|bf16), default = split.
static bool engine_ok(int e) { return e >= 0 && e <= 2; }

// ---- the one place the library reads the environment (pa2d_internal.h: Pa2dEnv)
static Pa2dEnv read_env() {
    auto is = [](const char* n, const char* pre) { const char* e = getenv(n); return e && strncmp(e, pre, strlen(pre)) == 0; };
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return (e && e[0]) ? atoi(e) : dflt; };
    Pa2dEnv v;
    v.conv_halo = is("PA2D_CONV_HALO", "o") ? 0 : (is("PA2D_CONV_HALO", "f") ? 2 : 1);
    v.mc_big_off = is("PA2D_MC_BIG", "o") ? 1 : 0;
    v.kc_bk16 = num("PA2D_KC_BK", 0) == 16 ? 1 : 0;
    v.mc_bk = num("PA2D_MC_BK", 16) == 32 ? 32 : 16;
    v.mc_splits = num("PA2D_MC_SPLITS", 0);
    v.mcb_splits = num("PA2D_MCB_SPLITS", 0);
    v.lin_dw_split = is("PA2D_LIN_DW_SPLIT", "of") ? 0 : 1;
    v.lin_panel = is("PA2D_LIN_PANEL", "of") ? 0 : 1;
    v.lin_rowpanel = is("PA2D_LIN_ROWPANEL", "of") ? 0 : 1;
    v.lin_small_split = is("PA2D_LIN_SMALL_SPLIT", "of") ? 0 : 1;
    v.conv_mfma16 = num("PA2D_CONV_MFMA", 16) == 32 ? 0 : 1;
    v.split_big = num("PA2D_SPLIT_BIG", 1);
    v.slice_map = is("PA2D_SLICE_MAP", "l") ? 0 : 1;
    v.default_engine = is("PA2D_GEMM", "f") ? 0 : (is("PA2D_GEMM", "b") ? 2 : 1);
    return v;
}
static Pa2dEnv g_env = read_env();
const Pa2dEnv& pa2d_env() { return g_env; }
// K-step: 32 (half the barriers of 16, full 128-byte row segments; 3-4 % faster on the conv, ~10 % on the small tiles)
// whenever the layout allows it: plain GEMMs always, the conv when Cin % 32 == 0; otherwise 16.
static int kc_bk(bool im2col, int Cin) {
    // PA2D_KC_BK=16: 16-wide K-step for the conv (124 VGPRs, 40 KB LDS -> 4 workgroups per CU)
    if (im2col && pa2d_env().kc_bk16) return 16;
    return (!im2col || (Cin % 32) == 0) ? 32 : 16;
}
// tile choice of the fp32 engine: 128x128 when that already gives >= 1.5 workgroups per CU, otherwise smaller
// tiles so that small problems (rollout at batch 1: M = 4096) still fill the 256 CUs.  Shared by the launch and
// by the conv weight packs (the pack's channel chunk must equal the K-step).
KCTile kc_tile(int M, int N, bool im2col, int Cin) {
    const int tiles_m = ceil_div(M, 128);
    const long long t128 = (long long)tiles_m * ceil_div(N, 128);
    const long long t12864 = (long long)tiles_m * ceil_div(N, 64);
    const int bk = kc_bk(im2col, Cin);
    if (N > 64 && t128 >= 384) return {128, 128, bk};
    if (t12864 >= 384 || M <= 64) return {128, 64, bk};
    return {64, 64, bk};
}
// (per-TILE kernels: the split engine serves the conv implicit GEMMs only.  Measured round 2: the K = C linears as
// 6-term splits on 128x128 tiles run at the SAME 0.155 ms as on the exact engine — they are bound by their short K loop
// (8 K-steps between a cold prologue and an epilogue), not by the matrix pipe.  Large-M linears of the bf16 engines
// therefore take the persistent row-panel kernel instead, see panel_applies / pa2d_gemm_panel.hip.)
bool use_split(int engine, int N, bool im2col, int Cin) {
    const int m = engine;
    if (m == 1) return im2col && N > 64 && (Cin % 32) == 0;
    if (m == 2) return N > 64 && (!im2col || (Cin % 32) == 0);
    return false;
}

static int launch_kc(const KCParams& p_in, bool im2col, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
static int launch_kc(const KCParams& p_in, bool im2col, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    KCParams p = p_in;
    if (!engine_ok(p.engine)) return PA2D_ERR_ARG;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return PA2D_OK;
    if (!p.bias2) p.bias_split = 0x7fffffff;
    {   // 32-bit byte offsets inside the buffer descriptors
        const unsigned long long ab = ((unsigned long long)(p.M - 1) * p.lda + (im2col ? p.Cin : p.K)) * 4ull;
        const unsigned long long bb = ((unsigned long long)(p.N - 1) * p.ldb + p.K) * 4ull;
        if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.a_bytes = (unsigned)ab;
        p.b_bytes = (unsigned)bb;
        const unsigned long long es = p.io_bf16 ? 2ull : 4ull;      // storage size of C / res / aux
        const unsigned long long cb = ((unsigned long long)(p.M - 1) * p.ldc + p.N) * es;
        const unsigned long long rb = p.res ? ((unsigned long long)(p.M - 1) * p.ldres + p.N) * es : 0ull;
        const unsigned long long xb = p.aux ? ((unsigned long long)(p.M - 1) * p.ldaux + p.N) * es : 0ull;
        if (cb >= 0xFFFFFFF0ull || rb >= 0xFFFFFFF0ull || xb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.c_bytes = (unsigned)cb; p.res_bytes = (unsigned)rb; p.aux_bytes = (unsigned)xb;
    }
    if ((p.K & 3) || (p.lda & 3) || (p.ldb & 3)) return PA2D_ERR_ARG;
    if (im2col && ((p.Cin & 15) || p.K != 9 * p.Cin)) return PA2D_ERR_UNSUPPORTED;
    if (im2col && (p.epi != 0 || p.res)) return PA2D_ERR_ARG;      // conv kernels carry the bias-only epilogue
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return PA2D_ERR_ARG;
    int rc;
    if (!im2col && p.wimg && rowpanel_applies(p)) {
        // row-stationary kernel on the whole 128-row rounds, the per-tile kernels on the (< 128-row) tail
        if (p.wsrc) {      // NULL: the caller made the image (pa2d_gemm_weight_image) while the weights stand still
            rc = launch_pack_weight_image(p.wsrc, p.wsn, p.wsk, p.wimg, p.N, p.K, st);
            if (rc) return rc;
        }
        KCParams pp = p;
        pp.M = p.M - p.M % 128;
        rc = launch_kc_rowpanel(pp, p.wimg, st);
        if (!rc && pp.M < p.M) {
            KCParams pt = p_in;
            pt.wimg = nullptr;
            pt.M = p.M - pp.M;
            pt.A = p.A + (size_t)pp.M * p.lda;
            pt.C = p.C + (size_t)pp.M * p.ldc;
            if (p.res) pt.res = p.res + (size_t)pp.M * p.ldres;
            if (p.aux) pt.aux = p.aux + (size_t)pp.M * p.ldaux;
            rc = launch_kc(pt, false, st);
        }
    } else if (panel_applies(p, im2col)) {
        // persistent row-panel kernel on the whole 256-row blocks, the per-tile kernels on the (< 256-row) tail
        KCParams pp = p;
        pp.M = p.M - p.M % 256;
        rc = launch_kc_panel(pp, st);
        if (!rc && pp.M < p.M) {
            KCParams pt = p_in;
            const size_t ea = p.io_bf16 ? 2 : 4, es = p.io_bf16 ? 2 : 4;
            pt.M = p.M - pp.M;
            pt.A = (const float*)((const char*)p.A + (size_t)pp.M * p.lda * ea);
            pt.C = (float*)((char*)p.C + (size_t)pp.M * p.ldc * es);
            if (p.res) pt.res = (const float*)((const char*)p.res + (size_t)pp.M * p.ldres * es);
            if (p.aux) pt.aux = (float*)((char*)p.aux + (size_t)pp.M * p.ldaux * es);
            rc = launch_kc(pt, false, st);
        }
    } else if (p.io_bf16 || use_split(p.engine, p.N, im2col, p.Cin)) rc = launch_kc_split(p, im2col, st);
    else if (kc_split_small_applies(p, im2col)) rc = launch_kc_split_small(p, st);
    else rc = launch_kc_f32(p, im2col, kc_tile(p.M, p.N, im2col, p.Cin), st);
    if (rc) return rc;
    if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return PA2D_ERR_ARG;
    return PA2D_OK;
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

// The engine a caller should use when it has no preference: env PA2D_GEMM=f32|split|bf16, else the fp32-accurate
// split engine.  A pure function of the environment — nothing in the library reads it implicitly.
int pa2d_default_engine(void) { return pa2d_env().default_engine; }

void pa2d_reload_env(void) { g_env = read_env(); }

// bytes of scratch with which the forward / data-gradient linears of an [N, K] weight take their fastest kernel under
// `engine` (split engine, K in {128, 256}, N % 64 == 0: the row-stationary kernel's weight plane image); the calls
// work with less (data gradient: at least K*N floats for the transposed weight) on the other kernels
size_t pa2d_gemm_fwd_workspace(int N, int K, int engine) { return rowpanel_image_bytes(N, K, engine); }
size_t pa2d_gemm_bwd_data_workspace(int N, int K, int engine) {
    return (size_t)N * K * sizeof(float) + rowpanel_image_bytes(K, N, engine);
}

// The weight plane image on its own: callers that know the weights stand still over several calls (the model calls and
// the backward pass of one training iteration, a rollout) make it once and pass it as `wimg`.  transposed = 0: the image of
// w[N, K] for the forward of that layer; 1: the image of w^T for its data gradient (N = the layer's input width, K = its
// output width, w stored [K, N] with row pitch ldw).  img_bytes >= pa2d_gemm_fwd_workspace(N, K, engine) > 0.
int pa2d_gemm_weight_image(const float* w, long long ldw, int transposed, void* img, size_t img_bytes, int N, int K,
                           int engine, hipStream_t st) {
    const size_t need = rowpanel_image_bytes(N, K, engine);
    if (need == 0) return PA2D_ERR_UNSUPPORTED;
    if (!img || img_bytes < need) return PA2D_ERR_WORKSPACE;
    return transposed ? launch_pack_weight_image(w, 1, ldw, img, N, K, st) : launch_pack_weight_image(w, ldw, 1, img, N, K, st);
}

int pa2d_gemm_bias_act_fwd(const float* x, long long ldx, const float* w, long long ldw, const float* bias,
                           const float* res, long long ldres, float* y, long long ldy, float* pre, long long ldpre,
                           const void* wimg, void* ws, size_t ws_bytes, int M, int N, int K, int act, int engine,
                           hipStream_t st) {
    KCParams p = {};
    p.engine = engine;
    p.aux_deriv = (act & PA2D_ACT_SAVE_DERIVATIVE_BIT) ? 1 : 0;
    act &= ~PA2D_ACT_SAVE_DERIVATIVE_BIT;
    p.A = x; p.lda = ldx; p.B = w; p.ldb = ldw; p.C = y; p.ldc = ldy; p.bias = bias; p.res = res; p.ldres = ldres;
    p.aux = pre; p.ldaux = ldpre; p.M = M; p.N = N; p.K = K; p.act = act;
    p.epi = (act != ACT_NONE ? EPI_ACT : 0) | (pre ? EPI_STORE_PRE : 0);
    const size_t img = rowpanel_image_bytes(N, K, engine);
    if (img && wimg) p.wimg = const_cast<void*>(wimg);      // ready-made (p.wsrc stays NULL: nothing to pack)
    else if (ws && img && ws_bytes >= img) { p.wimg = ws; p.wsrc = w; p.wsn = ldw; p.wsk = 1; }
    return launch_kc(p, false, st);
}

// dx[M,K] = (dy[M,N] . w[N,K]) * act'(pre[M,K])   (pre may be NULL -> plain product)
// ws: at least K*N floats (the transposed weight of the per-tile kernels); pa2d_gemm_bwd_data_workspace(N, K, engine)
// bytes let the split engine's row-stationary kernel run: [transposed weight | weight plane image].
int pa2d_gemm_bwd_data(const float* dy, long long lddy, const float* w, long long ldw, const float* pre,
                       long long ldpre, int act, float* dx, long long lddx, const void* wimg, void* ws, size_t ws_bytes, int M,
                       int N, int K, int engine, hipStream_t st) {
    if (ldw != K) return PA2D_ERR_ARG;
    if (N & 3) return PA2D_ERR_ARG;
    if (M <= 0) return PA2D_OK;
    if (ws_bytes < (size_t)N * K * sizeof(float)) return PA2D_ERR_WORKSPACE;
    float* const wt_ws = (float*)ws;
    KCParams p = {};
    p.engine = engine;
    p.aux_deriv = (act & PA2D_ACT_SAVE_DERIVATIVE_BIT) ? 1 : 0;
    act &= ~PA2D_ACT_SAVE_DERIVATIVE_BIT;
    p.A = dy; p.lda = lddy; p.B = wt_ws; p.ldb = N; p.C = dx; p.ldc = lddx; p.M = M; p.N = K; p.K = N;
    p.aux = const_cast<float*>(pre); p.ldaux = ldpre; p.act = act;
    p.epi = (pre && act != ACT_NONE) ? EPI_MUL_DACT : 0;
    const size_t img = rowpanel_image_bytes(K, N, engine);
    if (img && wimg) p.wimg = const_cast<void*>(wimg);      // ready-made image of w^T (pa2d_gemm_weight_image, transposed = 1)
    else if (img && ws_bytes >= (size_t)N * K * sizeof(float) + img) {
        p.wimg = (char*)ws + (size_t)N * K * sizeof(float); p.wsrc = w; p.wsn = 1; p.wsk = ldw;
    }
    if (!(p.wimg && rowpanel_applies(p) && (M % 128) == 0)) {      // someone reads the fp32 transpose
        const int rc = launch_repack(w, nullptr, wt_ws, 0, N, K, 0, 0, st);
        if (rc) return rc;
    }
    return launch_kc(p, false, st);
}

size_t pa2d_gemm_bwd_weight_workspace(int M, int N, int K, int engine) {
    (void)engine;
    const MCPlan pl = plan_mc(N, K, M);
    return (pl.slab_floats + (size_t)pl.splits * N) * sizeof(float);      // [slabs | per-split column sums of dy]
}

// dw[N,K] (+)= dy[M,N]^T . x[M,K] ; db[N] (+)= column sums of dy (db may be NULL).  accumulate != 0: add to what dw / db
// hold (gradient accumulation straight into the caller's bucket, done inside the slab-reduce pass).
int pa2d_gemm_bwd_weight(const float* dy, long long lddy, const float* x, long long ldx, float* dw, float* db,
                         void* ws, size_t ws_bytes, int M, int N, int K, int accumulate, int engine, hipStream_t st) {
    if (!engine_ok(engine)) return PA2D_ERR_ARG;
    if (M <= 0) {
        if (accumulate) return PA2D_OK;
        const int rz = pa2d_zero(dw, sizeof(float) * N * K, st);
        return rz ? rz : pa2d_zero(db, sizeof(float) * N, st);
    }
    if (ws_bytes < pa2d_gemm_bwd_weight_workspace(M, N, K, engine)) return PA2D_ERR_WORKSPACE;
    const MCPlan pl = plan_mc(N, K, M);
    // the bias gradient (column sums of dy) rides in the GEMM: the workgroups of column tile 0 sum the dy tiles they stage
    float* cs = db ? (float*)ws + pl.slab_floats : nullptr;
    // bf16 engines: the same split plan on the transposed-read planes kernel, operands split while staging
    const bool bfk = (engine == 1 || engine == 2) && pl.big && mc_f32src_applies(N, K, M, lddy, ldx);
    int rc = bfk ? launch_mc_f32src(dy, lddy, N, x, ldx, K, M, (float*)ws, pl, engine == 2 ? 1 : 3, cs, st)
                 : launch_mc(dy, lddy, N, x, ldx, K, M, false, 0, 0, 0, (float*)ws, pl, engine, st, cs);
    if (rc) return rc;
    rc = launch_reduce((const float*)ws, pl.splits, (long long)N * K, dw, nullptr, 0, 0, 0, st, accumulate);
    if (rc) return rc;
    if (db) rc = launch_reduce(cs, pl.splits, N, db, nullptr, 0, 0, 0, st, accumulate);
    return rc;
}

// bf16 engines: bytes of the pre-split activation planes of a [rows, Cin] operand (0 when the engine selected for
// this GEMM reads fp32 operands)
static size_t conv_planes_bytes(int engine, int M, int N, int Cin) {
    if (!use_split(engine, N, true, Cin)) return 0;
    return (planes_bytes(M, Cin, engine == 2 ? 1 : 3) + 255) & ~(size_t)255;
}

// bf16 engines: the weight gradient also runs from pre-split planes (of dOut and of X) when the tile shapes allow
// 0 = fp32 operands (gather kernel), 1 = planes, 128 x 128 tiles, 2 = planes, 256 x 256 tiles (one round of workgroups)
static int conv_dw_kind(int engine, int M, int C) {
    if (conv_planes_bytes(engine, M, C, 2 * C) == 0) return 0;
    if (mc_planes_big_applies(C, C, M)) return 2;
    return (mc_planes_supported(C, C) && plan_mc(2 * C, 9 * C, M).big) ? 1 : 0;
}
static MCPlan conv_dw_plan(int engine, int M, int C) {
    return conv_dw_kind(engine, M, C) == 2 ? plan_mc_planes_big(2 * C, 9 * C, M) : plan_mc(2 * C, 9 * C, M);
}
static size_t conv_xplanes_bytes(int engine, int M, int C) {
    return conv_dw_kind(engine, M, C) ? ((planes_bytes(M, C, engine == 2 ? 1 : 3) + 255) & ~(size_t)255) : 0;
}

// backward workspace: [weight pack | slabs or column-sum partials | dOut planes | X planes]  (planes: bf16 engines)
size_t pa2d_conv3x3x2_workspace(int B, int H, int W, int C, int engine) {
    const size_t pack = (size_t)3 * C * 9 * C;      // fp32 pack (2C*9C floats) or 3 bf16 planes (1.5x)
    const MCPlan pl = conv_dw_plan(engine, B * H * W, C);
    size_t sl = pl.slab_floats, cs = (size_t)colsum_blocks(B * H * W) * 2 * C;
    return (pack + (sl > cs ? sl : cs)) * sizeof(float) + conv_planes_bytes(engine, B * H * W, C, 2 * C) +
           conv_xplanes_bytes(engine, B * H * W, C);
}

// forward workspace: [weight pack (unused if prepacked) | activation planes (bf16 engines)]
size_t pa2d_conv3x3x2_fwd_workspace(int B, int H, int W, int C, int engine) {
    return (size_t)3 * C * 9 * C * sizeof(float) + conv_planes_bytes(engine, B * H * W, 2 * C, C);
}

// Packed conv weights in the layout the engine selected for these dims wants (channel chunk = K-step of the
// tile, fp32 or bf16 planes by GEMM mode).  direction 0: forward pack ([2C][9C]); 1: data-gradient pack
// ([C][9*2C], taps flipped).  pack: pa2d_conv3x3x2_pack_bytes(C) bytes.  A pack stays valid while the weights,
// the dims and the GEMM mode do not change.
static int conv_pack(const float* wx, const float* wf, float* pack, int M, int C, int direction, int engine,
                     hipStream_t st) {
    const int N = direction ? C : 2 * C, Cin = direction ? 2 * C : C;
    if (use_split(engine, N, true, Cin)) {
        return launch_repack_split(wx, wf, pack, direction, engine == 2 ? 1 : 3, C, C, st);
    }
    return launch_repack(wx, wf, pack, direction ? 2 : 1, 0, kc_tile(M, N, true, Cin).bk, C, C, st);
}

size_t pa2d_conv3x3x2_pack_bytes(int C) { return (size_t)3 * C * 9 * C * sizeof(float); }

int pa2d_conv3x3x2_pack(const float* wx, const float* wf, void* pack, size_t pack_bytes, int B, int H, int W, int C,
                        int direction, int engine, hipStream_t st) {
    if (!engine_ok(engine)) return PA2D_ERR_ARG;
    if (pack_bytes < pa2d_conv3x3x2_pack_bytes(C)) return PA2D_ERR_WORKSPACE;
    return conv_pack(wx, wf, (float*)pack, B * H * W, C, direction ? 1 : 0, engine, st);
}

// out[B*H*W, 2C] = [conv3x3(xn, wx) + bx | conv3x3(xn, wf) + bf]   (zero padding 1, NHWC)
// Physics_Attention.py:94,96 — both projections read the same input, so they run as ONE implicit
// GEMM [B*N, 9C] x [9C, 2C].  prepacked: NULL (weights are packed into ws by this call) or a pack made by
// pa2d_conv3x3x2_pack(direction 0) for the same B, H, W, C.
int pa2d_conv3x3x2_fwd(const float* xn, const float* wx, const float* bx, const float* wf, const float* bf,
                       float* out, const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                       int engine, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (!engine_ok(engine)) return PA2D_ERR_ARG;
    if (B <= 0) return PA2D_OK;
    if (ws_bytes < pa2d_conv3x3x2_fwd_workspace(B, H, W, C, engine)) return PA2D_ERR_WORKSPACE;
    const float* pack = (const float*)prepacked;
    if (!pack) {
        const int rc = conv_pack(wx, wf, (float*)ws, B * H * W, C, 0, engine, st);
        if (rc) return rc;
        pack = (const float*)ws;
    }
    const size_t apl = conv_planes_bytes(engine, B * H * W, 2 * C, C);
    void* const planes = (char*)ws + pa2d_conv3x3x2_pack_bytes(C);
    if (apl) {
        const int rc = launch_split_planes(xn, C, planes, (long long)B * H * W, C, engine == 2 ? 1 : 3, st);
        if (rc) return rc;
    }
    KCParams p = {};
    p.engine = engine;
    p.A = apl ? (const float*)planes : xn; p.apre = apl ? 1 : 0;
    p.lda = C; p.B = pack; p.ldb = 9 * C; p.C = out; p.ldc = 2 * C;
    p.bias = bx; p.bias2 = bf; p.bias_split = C;
    p.M = B * H * W; p.N = 2 * C; p.K = 9 * C; p.H = H; p.W = W; p.Cin = C;
    return launch_kc(p, true, st, ev_start, ev_stop);
}

// dxn[B*N, C] (plain store), dwx/dwf [C,C,3,3], dbx/dbf [C] (accumulate != 0: added to) from dout[B*N, 2C]
int pa2d_conv3x3x2_bwd(const float* dout, const float* xn, const float* wx, const float* wf, float* dxn, float* dwx,
                       float* dbx, float* dwf, float* dbf, const void* prepacked, void* ws, size_t ws_bytes, int B,
                       int H, int W, int C, int accumulate, int engine, hipStream_t st, hipEvent_t ev_start,
                       hipEvent_t ev_stop) {
    if (!engine_ok(engine)) return PA2D_ERR_ARG;
    if (B <= 0) {
        if (accumulate) return PA2D_OK;
        const size_t wb = sizeof(float) * (size_t)C * C * 9, bb = sizeof(float) * C;
        int rz = pa2d_zero(dwx, wb, st);
        if (!rz) rz = pa2d_zero(dwf, wb, st);
        if (!rz) rz = pa2d_zero(dbx, bb, st);
        return rz ? rz : pa2d_zero(dbf, bb, st);
    }
    if (ws_bytes < pa2d_conv3x3x2_workspace(B, H, W, C, engine)) return PA2D_ERR_WORKSPACE;
    float* scratch = (float*)ws + (size_t)3 * C * 9 * C;
    const int M = B * H * W;
    int rc;
    const int NT = engine == 2 ? 1 : 3;
    const size_t apl = conv_planes_bytes(engine, M, C, 2 * C), xpl = conv_xplanes_bytes(engine, M, C);
    void* const planes = (char*)ws + pa2d_conv3x3x2_workspace(B, H, W, C, engine) - apl - xpl;     // dOut planes
    void* const xplanes = (char*)planes + apl;                                               // X planes
    if (apl && (dxn || xpl)) {
        rc = launch_split_planes(dout, 2 * C, planes, M, 2 * C, NT, st);
        if (rc) return rc;
    }
    if (dxn) {
        const float* pack = (const float*)prepacked;
        if (!pack) {
            rc = conv_pack(wx, wf, (float*)ws, M, C, 1, engine, st);
            if (rc) return rc;
            pack = (const float*)ws;
        }
        KCParams p = {};
        p.engine = engine;
        p.A = apl ? (const float*)planes : dout; p.apre = apl ? 1 : 0;
        p.lda = 2 * C; p.B = pack; p.ldb = 9 * 2 * C; p.C = dxn; p.ldc = C;
        p.M = M; p.N = C; p.K = 9 * 2 * C; p.H = H; p.W = W; p.Cin = 2 * C;
        rc = launch_kc(p, true, st, ev_start, ev_stop);
        if (rc) return rc;
    }
    const MCPlan pl = conv_dw_plan(engine, M, C);
    if (xpl) {      // bf16 engines: both operands as pre-split planes, transposed LDS reads, no conversion in the GEMM
        rc = launch_split_planes(xn, C, xplanes, M, C, NT, st);
        if (rc) return rc;
        rc = pl.big == 2 ? launch_mc_planes_big(planes, xplanes, C, C, M, H, W, scratch, pl, NT, st)
                         : launch_mc_planes(planes, xplanes, C, C, M, H, W, scratch, pl, NT, st);
    } else {
        rc = launch_mc(dout, 2 * C, 2 * C, xn, C, 9 * C, M, true, H, W, C, scratch, pl, engine, st);
    }
    if (rc) return rc;
    rc = launch_reduce(scratch, pl.splits, (long long)2 * C * 9 * C, dwx, dwf, 1, C, C, st, accumulate);
    if (rc) return rc;
    return launch_colsum(dout, 2 * C, M, 2 * C, dbx, scratch, st, dbf, C, accumulate);
}


// =============================================================================================
// Operand-planes interface of the bf16 engines (fp32 storage): the producers of the conv operands (LayerNorm forward,
// slice backward) can emit the bf16 plane image directly (pa2d_layernorm_fwd_planes, pa2d_slice_bwd_points_planes); these
// entry points consume it, so no fp32 copy of the operand and no split pre-pass exists.

// bytes of the plane image of a [rows, C] tensor under `engine` (0 for PA2D_ENGINE_F32)
size_t pa2d_planes_bytes(long long rows, int C, int engine) {
    if (engine != 1 && engine != 2) return 0;
    return planes_bytes(rows, C, engine == 2 ? 1 : 3);
}

// which conv operands `engine` consumes as planes at this shape: bit 0 = X in the forward GEMM, bit 1 = X in the weight
// gradient, bit 2 = dOut in the data AND weight gradient.  The *_planes entry points need all three (mask == 7).
int pa2d_conv3x3x2_planes_mask(int B, int H, int W, int C, int engine) {
    if (!engine_ok(engine) || B <= 0) return 0;
    const int M = B * H * W;
    int m = 0;
    if (conv_planes_bytes(engine, M, 2 * C, C)) m |= 1;
    if (conv_dw_kind(engine, M, C)) m |= 2;
    if (conv_planes_bytes(engine, M, C, 2 * C) && conv_dw_kind(engine, M, C)) m |= 4;
    return m;
}

int pa2d_conv3x3x2_fwd_planes(const void* xn_planes, const float* wx, const float* bx, const float* wf, const float* bf,
                              float* out, const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                              int engine, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (B <= 0) return PA2D_OK;
    if ((pa2d_conv3x3x2_planes_mask(B, H, W, C, engine) & 1) == 0) return PA2D_ERR_UNSUPPORTED;
    if (ws_bytes < pa2d_conv3x3x2_pack_bytes(C)) return PA2D_ERR_WORKSPACE;
    const float* pack = (const float*)prepacked;
    if (!pack) {
        const int rc = conv_pack(wx, wf, (float*)ws, B * H * W, C, 0, engine, st);
        if (rc) return rc;
        pack = (const float*)ws;
    }
    KCParams p = {};
    p.engine = engine;
    p.A = (const float*)xn_planes; p.apre = 1;
    p.lda = C; p.B = pack; p.ldb = 9 * C; p.C = out; p.ldc = 2 * C;
    p.bias = bx; p.bias2 = bf; p.bias_split = C;
    p.M = B * H * W; p.N = 2 * C; p.K = 9 * C; p.H = H; p.W = W; p.Cin = C;
    return launch_kc(p, true, st, ev_start, ev_stop);
}

size_t pa2d_conv3x3x2_workspace_planes(int B, int H, int W, int C, int engine) {
    const MCPlan pl = conv_dw_plan(engine, B * H * W, C);
    return pa2d_conv3x3x2_pack_bytes(C) + pl.slab_floats * sizeof(float);
}

// dxn (may be NULL), dwx / dwf ((+)= per accumulate) from the plane images of dOut [B*H*W, 2C] and X [B*H*W, C];
// the bias gradients come from pa2d_slice_bwd_points_planes
int pa2d_conv3x3x2_bwd_planes(const void* dout_planes, const void* xn_planes, const float* wx, const float* wf, float* dxn,
                              float* dwx, float* dwf, const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W,
                              int C, int accumulate, int engine, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (B <= 0) {
        if (accumulate) return PA2D_OK;
        const size_t wb = sizeof(float) * (size_t)C * C * 9;
        const int rz = pa2d_zero(dwx, wb, st);
        return rz ? rz : pa2d_zero(dwf, wb, st);
    }
    if (pa2d_conv3x3x2_planes_mask(B, H, W, C, engine) != 7) return PA2D_ERR_UNSUPPORTED;
    if (ws_bytes < pa2d_conv3x3x2_workspace_planes(B, H, W, C, engine)) return PA2D_ERR_WORKSPACE;
    float* scratch = (float*)((char*)ws + pa2d_conv3x3x2_pack_bytes(C));
    const int M = B * H * W, NT = engine == 2 ? 1 : 3;
    int rc;
    if (dxn) {
        const float* pack = (const float*)prepacked;
        if (!pack) {
            rc = conv_pack(wx, wf, (float*)ws, M, C, 1, engine, st);
            if (rc) return rc;
            pack = (const float*)ws;
        }
        KCParams p = {};
        p.engine = engine;
        p.A = (const float*)dout_planes; p.apre = 1;
        p.lda = 2 * C; p.B = pack; p.ldb = 9 * 2 * C; p.C = dxn; p.ldc = C;
        p.M = M; p.N = C; p.K = 9 * 2 * C; p.H = H; p.W = W; p.Cin = 2 * C;
        rc = launch_kc(p, true, st, ev_start, ev_stop);
        if (rc) return rc;
    }
    const MCPlan pl = conv_dw_plan(engine, M, C);
    rc = pl.big == 2 ? launch_mc_planes_big(dout_planes, xn_planes, C, C, M, H, W, scratch, pl, NT, st)
                     : launch_mc_planes(dout_planes, xn_planes, C, C, M, H, W, scratch, pl, NT, st);
    if (rc) return rc;
    return launch_reduce(scratch, pl.splits, (long long)2 * C * 9 * C, dwx, dwf, 1, C, C, st, accumulate);
}

// =============================================================================================
// bf16-STORAGE variants (BASELINE configs[2] / [4] as stated): activations, saved tensors and inter-kernel gradients are
// bf16 in HBM; weights, biases and every parameter gradient stay fp32; all products are ONE bf16 MFMA term with fp32
// accumulation (the arithmetic of PA2D_ENGINE_BF16, minus its fp32 round trips: a bf16 tensor IS the 1-plane operand
// image the bf16 kernels stage, so the activation pre-passes disappear).  ld* are in ELEMENTS.  Requires K % 32 == 0
// for the dense layers and C % 32 == 0 for the conv (PA2D_ERR_UNSUPPORTED otherwise, never a silent fallback).

int pa2d_gemm_bias_act_fwd_bf16(const void* x, long long ldx, const float* w, long long ldw, const float* bias,
                                const void* res, long long ldres, void* y, long long ldy, void* pre, long long ldpre,
                                int M, int N, int K, int act, hipStream_t st) {
    KCParams p = {};
    p.engine = 2; p.io_bf16 = 1; p.apre = 1;
    p.aux_deriv = (act & PA2D_ACT_SAVE_DERIVATIVE_BIT) ? 1 : 0;
    act &= ~PA2D_ACT_SAVE_DERIVATIVE_BIT;
    p.A = (const float*)x; p.lda = ldx; p.B = w; p.ldb = ldw; p.C = (float*)y; p.ldc = ldy; p.bias = bias;
    p.res = (const float*)res; p.ldres = ldres; p.aux = (float*)pre; p.ldaux = ldpre; p.M = M; p.N = N; p.K = K; p.act = act;
    p.epi = (act != ACT_NONE ? EPI_ACT : 0) | (pre ? EPI_STORE_PRE : 0);
    return launch_kc(p, false, st);
}

// dx[M,K] = (dy[M,N] . w[N,K]) * act'(pre[M,K]); dy, pre, dx bf16; w fp32; wt_ws: K*N floats (transposed weight)
int pa2d_gemm_bwd_data_bf16(const void* dy, long long lddy, const float* w, long long ldw, const void* pre,
                            long long ldpre, int act, void* dx, long long lddx, float* wt_ws, int M, int N, int K,
                            hipStream_t st) {
    if (ldw != K) return PA2D_ERR_ARG;
    if (N & 3) return PA2D_ERR_ARG;
    if (M <= 0) return PA2D_OK;
    int rc = launch_repack(w, nullptr, wt_ws, 0, N, K, 0, 0, st);
    if (rc) return rc;
    KCParams p = {};
    p.engine = 2; p.io_bf16 = 1; p.apre = 1;
    p.aux_deriv = (act & PA2D_ACT_SAVE_DERIVATIVE_BIT) ? 1 : 0;
    act &= ~PA2D_ACT_SAVE_DERIVATIVE_BIT;
    p.A = (const float*)dy; p.lda = lddy; p.B = wt_ws; p.ldb = N; p.C = (float*)dx; p.ldc = lddx; p.M = M; p.N = K; p.K = N;
    p.aux = (float*)const_cast<void*>(pre); p.ldaux = ldpre; p.act = act;
    p.epi = (pre && act != ACT_NONE) ? EPI_MUL_DACT : 0;
    return launch_kc(p, false, st);
}

size_t pa2d_gemm_bwd_weight_workspace_bf16(int M, int N, int K) {
    const MCPlan pl = plan_mc_planes_big(N, K, M);
    size_t a = pl.slab_floats, b = (size_t)colsum_blocks(M) * N;
    return (a > b ? a : b) * sizeof(float);
}

// dw[N,K] (+)= dy[M,N]^T . x[M,K] (fp32), db[N] (+)= column sums of dy; dy, x bf16 and CONTIGUOUS (lddy == N, ldx == K)
int pa2d_gemm_bwd_weight_bf16(const void* dy, long long lddy, const void* x, long long ldx, float* dw, float* db,
                              void* ws, size_t ws_bytes, int M, int N, int K, int accumulate, hipStream_t st) {
    if (lddy != N || ldx != K) return PA2D_ERR_ARG;
    if ((N & 31) || (K & 31)) return PA2D_ERR_UNSUPPORTED;
    if (M <= 0) {
        if (accumulate) return PA2D_OK;
        const int rz = pa2d_zero(dw, sizeof(float) * N * K, st);
        return rz ? rz : pa2d_zero(db, sizeof(float) * N, st);
    }
    if (ws_bytes < pa2d_gemm_bwd_weight_workspace_bf16(M, N, K)) return PA2D_ERR_WORKSPACE;
    const MCPlan pl = plan_mc_planes_big(N, K, M);
    int rc = launch_mc_planes_big_raw(dy, N, x, K, 1, M, 1, 1, (float*)ws, pl, 1, st);
    if (rc) return rc;
    rc = launch_reduce((const float*)ws, pl.splits, (long long)N * K, dw, nullptr, 0, 0, 0, st, accumulate);
    if (rc) return rc;
    if (db) rc = launch_colsum_bf16(dy, lddy, M, N, db, (float*)ws, st, nullptr, 0, accumulate);
    return rc;
}

// weight pack of the bf16-storage conv: ALWAYS the 1-plane bf16 K-step image (these entry points use the bf16 kernels
// for every shape, also the narrow ones the fp32-I/O engines hand to the exact kernel)
int pa2d_conv3x3x2_pack_bf16(const float* wx, const float* wf, void* pack, size_t pack_bytes, int C, int direction,
                             hipStream_t st) {
    if (C & 31) return PA2D_ERR_UNSUPPORTED;
    if (pack_bytes < pa2d_conv3x3x2_pack_bytes(C)) return PA2D_ERR_WORKSPACE;
    return launch_repack_split(wx, wf, pack, direction ? 1 : 0, 1, C, C, st);
}

size_t pa2d_conv3x3x2_fwd_workspace_bf16(int B, int H, int W, int C) {
    (void)B; (void)H; (void)W;
    return pa2d_conv3x3x2_pack_bytes(C);
}
size_t pa2d_conv3x3x2_workspace_bf16(int B, int H, int W, int C) {
    const MCPlan pl = plan_mc_planes_big(2 * C, 9 * C, B * H * W);
    size_t sl = pl.slab_floats, cs = (size_t)colsum_blocks(B * H * W) * 2 * C;
    return pa2d_conv3x3x2_pack_bytes(C) + (sl > cs ? sl : cs) * sizeof(float);
}

// xn [B*H*W, C] bf16 -> out [B*H*W, 2C] bf16; weights / biases fp32 (prepacked: pa2d_conv3x3x2_pack with PA2D_ENGINE_BF16)
int pa2d_conv3x3x2_fwd_bf16(const void* xn, const float* wx, const float* bx, const float* wf, const float* bf, void* out,
                            const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W, int C, hipStream_t st,
                            hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (C & 31) return PA2D_ERR_UNSUPPORTED;
    if (B <= 0) return PA2D_OK;
    if (ws_bytes < pa2d_conv3x3x2_fwd_workspace_bf16(B, H, W, C)) return PA2D_ERR_WORKSPACE;
    const float* pack = (const float*)prepacked;
    if (!pack) {
        const int rc = launch_repack_split(wx, wf, ws, 0, 1, C, C, st);
        if (rc) return rc;
        pack = (const float*)ws;
    }
    KCParams p = {};
    p.engine = 2; p.io_bf16 = 1; p.apre = 1;
    p.A = (const float*)xn; p.lda = C; p.B = pack; p.ldb = 9 * C; p.C = (float*)out; p.ldc = 2 * C;
    p.bias = bx; p.bias2 = bf; p.bias_split = C;
    p.M = B * H * W; p.N = 2 * C; p.K = 9 * C; p.H = H; p.W = W; p.Cin = C;
    return launch_kc(p, true, st, ev_start, ev_stop);
}

// dout [B*H*W, 2C] bf16, xn bf16 -> dxn bf16 (may be NULL); dwx/dwf/dbx/dbf fp32 ((+)= per accumulate)
int pa2d_conv3x3x2_bwd_bf16(const void* dout, const void* xn, const float* wx, const float* wf, void* dxn, float* dwx,
                            float* dbx, float* dwf, float* dbf, const void* prepacked, void* ws, size_t ws_bytes, int B,
                            int H, int W, int C, int accumulate, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (C & 31) return PA2D_ERR_UNSUPPORTED;
    if (B <= 0) {
        if (accumulate) return PA2D_OK;
        const size_t wb = sizeof(float) * (size_t)C * C * 9, bb = sizeof(float) * C;
        int rz = pa2d_zero(dwx, wb, st);
        if (!rz) rz = pa2d_zero(dwf, wb, st);
        if (!rz) rz = pa2d_zero(dbx, bb, st);
        return rz ? rz : pa2d_zero(dbf, bb, st);
    }
    if (ws_bytes < pa2d_conv3x3x2_workspace_bf16(B, H, W, C)) return PA2D_ERR_WORKSPACE;
    float* scratch = (float*)((char*)ws + pa2d_conv3x3x2_pack_bytes(C));
    const int M = B * H * W;
    int rc;
    if (dxn) {
        const float* pack = (const float*)prepacked;
        if (!pack) {
            rc = launch_repack_split(wx, wf, ws, 1, 1, C, C, st);
            if (rc) return rc;
            pack = (const float*)ws;
        }
        KCParams p = {};
        p.engine = 2; p.io_bf16 = 1; p.apre = 1;
        p.A = (const float*)dout; p.lda = 2 * C; p.B = pack; p.ldb = 9 * 2 * C; p.C = (float*)dxn; p.ldc = C;
        p.M = M; p.N = C; p.K = 9 * 2 * C; p.H = H; p.W = W; p.Cin = 2 * C;
        rc = launch_kc(p, true, st, ev_start, ev_stop);
        if (rc) return rc;
    }
    const MCPlan pl = plan_mc_planes_big(2 * C, 9 * C, M);
    rc = launch_mc_planes_big_raw(dout, 2 * C, xn, C, 9, M, H, W, scratch, pl, 1, st);
    if (rc) return rc;
    rc = launch_reduce(scratch, pl.splits, (long long)2 * C * 9 * C, dwx, dwf, 1, C, C, st, accumulate);
    if (rc) return rc;
    return launch_colsum_bf16(dout, 2 * C, M, 2 * C, dbx, scratch, st, dbf, C, accumulate);
}

}  // extern "C"
