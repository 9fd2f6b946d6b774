/* libpa2d — C ABI of the MI355X-native Transolver Physics-Attention hot path.
 *
 * Drop-in boundary.  The reference has NO native layer: every stage below is what
 * stock PyTorch dispatches from the Python lines cited per function (paths relative to the
 * reference repo OnurBasci/TransformerBasedNavierStokeSolver).  A maintainer binds these symbols
 * with ctypes from a replacement of model/Physics_Attention.py /
 * model/Transolver_Structured_Mesh_2D.py; the shipped binding is
 * transformerbasednavierstokesolver_amd/_lib.py.
 *
 * Conventions
 *   - all tensors fp32, row-major, device pointers; "ld*" = row pitch in floats; an operand may not extend
 *     past 4 GiB (32-bit buffer descriptors): larger problems return PA2D_ERR_UNSUPPORTED before any launch;
 *   - empty problems (batch 0) are valid: maps are no-ops, reductions are zero-filled;
 *   - the caller owns every buffer (outputs and workspaces); nothing here allocates, frees or
 *     synchronises, and the library keeps NO mutable state (the GEMM engine is an explicit argument of every
 *     dense entry point) -> re-entrant, safe under hipGraph capture and on any stream;
 *   - gradient outputs of parameters take an `accumulate` flag: 0 = overwrite, 1 = add to what the buffer holds
 *     (done inside the deterministic partial-sum reduce pass, so gradient accumulation over the T/step model
 *     calls of an iteration, exp_ns.py:198-211, costs no extra launch or copy);
 *   - hipStream_t is passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return value: 0 = ok, otherwise a hipError_t or PA2D_ERR_* (never a silent fallback);
 *   - activation ids follow the reference's ACTIVATION table (…_2D.py:9-10).
 */
#ifndef PA2D_H
#define PA2D_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pa2d_stream_t;

#define PA2D_OK 0
#define PA2D_ERR_ARG 1001          /* alignment / shape contract violated */
#define PA2D_ERR_UNSUPPORTED 1002  /* size outside the compiled kernel grid */
#define PA2D_ERR_WORKSPACE 1003    /* caller workspace too small */

/* OR-ed into the `act` argument of the dense-layer entry points: `pre` receives (forward) / holds (data gradient)
 * act'(pre-activation) instead of the pre-activation itself — what the backward of `act(x.w^T + b)` needs; for GELU
 * the derivative shares every term with the activation, so saving it costs nothing and the data-gradient epilogue
 * becomes one multiplication. */
#define PA2D_ACT_SAVE_DERIVATIVE 0x100
enum pa2d_act { PA2D_ACT_NONE = 0, PA2D_ACT_GELU = 1, PA2D_ACT_TANH = 2, PA2D_ACT_SIGMOID = 3,
                PA2D_ACT_RELU = 4, PA2D_ACT_SOFTPLUS = 5, PA2D_ACT_ELU = 6, PA2D_ACT_SILU = 7 };

const char* pa2d_version(void);

/* GEMM engines (the `engine` argument of the dense entry points):
 * PA2D_ENGINE_F32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 * PA2D_ENGINE_SPLIT fp32-accurate split: operands split exactly into 3 bf16 terms, 6 bf16 MFMA terms per product, fp32
 *                   accumulate — same parity tolerances as the exact engine.  Used by the conv GEMMs
 *                   (forward, data and weight gradients) and by the large-M plain GEMMs (K % 32 == 0, >= 256 tiles of
 *                   256 x 128 rows x columns: forward / data gradient; weight gradient from 2048 rows); small GEMMs
 *                   stay on the exact kernels;
 * PA2D_ENGINE_BF16  bf16 compute: every GEMM rounds its operands to bf16 and uses ONE bf16 MFMA term with fp32
 *                   accumulation; tensors stay fp32 in HBM (autocast-style numerics, tolerance rel-L2 <= 3e-2).
 * Workspace sizes and weight-pack layouts depend on the engine: query them, and make packs, with the engine used.
 * pa2d_default_engine(): what a caller without a preference should pass — env PA2D_GEMM=f32|split|bf16, else
 * PA2D_ENGINE_SPLIT.  It only reads the environment; no entry point consults it implicitly. */
enum pa2d_engine { PA2D_ENGINE_F32 = 0, PA2D_ENGINE_SPLIT = 1, PA2D_ENGINE_BF16 = 2 };
int pa2d_default_engine(void);
/* Kernel-selection overrides (PA2D_CONV_HALO, PA2D_MC_BIG, PA2D_LIN_PANEL, ...: A/B timing and the parity tests that
 * force one of two equivalent kernels) and PA2D_GEMM are read from the environment ONCE, when the library is loaded;
 * pa2d_reload_env() reads them again (tests).  They never change the numerics contract of an entry point. */
void pa2d_reload_env(void);

/* ---- LayerNorm: nn.LayerNorm(C) of Transolver_block, model/Transolver_Structured_Mesh_2D.py:58,62,65,70-73 */
int pa2d_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                       float* rstd, int rows, int C, float eps, pa2d_stream_t stream);
size_t pa2d_layernorm_bwd_workspace(int rows, int C);
/* dx = LN'(dy) (+ dres: gradient of the residual branch `+ fx`, …_2D.py:70-71); dgamma/dbeta reduced ((+)= per
 * `accumulate`) */
int pa2d_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                       const float* gamma, const float* dres, float* dx, float* dgamma, float* dbeta,
                       void* ws, size_t ws_bytes, int rows, int C, int accumulate, pa2d_stream_t stream);

/* ---- dense layers: nn.Linear of MLP (…_2D.py:26-38), to_out (Physics_Attention.py:81-84,119)
 * y[M,N] = act(x[M,K] . w[N,K]^T + bias) (+ res);  pre (optional) receives the pre-activation.
 * Requires K % 4 == 0 and ldx, ldw % 4 == 0.
 * ws / ws_bytes: optional scratch (NULL / 0 allowed).  With pa2d_gemm_fwd_workspace(N, K, engine) bytes the split
 * engine's large-M layers (K in {128, 256}, N % 64 == 0, M >= 32768) run on the row-stationary kernel, which reads
 * the weight as a bf16 plane image made in ws by this call; results do not depend on which kernel ran beyond fp32
 * summation order. */
size_t pa2d_gemm_fwd_workspace(int N, int K, int engine);
/* The image on its own, for callers that know the weights stand still over several calls (the model calls and the
 * backward pass of one training iteration; a rollout): made once, passed as `wimg` (then ws may be NULL).
 * transposed = 0: image of w[N, K] for the layer's forward; 1: image of w^T for its data gradient (N = the layer's input
 * width, K = its output width, w stored [K, N]).  PA2D_ERR_UNSUPPORTED where pa2d_gemm_fwd_workspace(N, K, engine) is 0. */
int pa2d_gemm_weight_image(const float* w, long long ldw, int transposed, void* img, size_t img_bytes, int N, int K,
                           int engine, pa2d_stream_t stream);
int pa2d_gemm_bias_act_fwd(const float* x, long long ldx, const float* w, long long ldw, const float* bias,
                           const float* res, long long ldres, float* y, long long ldy, float* pre,
                           long long ldpre, const void* wimg, void* ws, size_t ws_bytes, int M, int N, int K, int act,
                           int engine, pa2d_stream_t stream);
/* dx[M,K] = (dy[M,N] . w[N,K]) * act'(pre[M,K])  (pre NULL -> no activation factor);
 * ws: at least K*N floats (transposed weight; PA2D_ERR_WORKSPACE below that); pa2d_gemm_bwd_data_workspace(N, K,
 * engine) bytes also hold the weight plane image of the row-stationary kernel (as for the forward); wimg (may be NULL):
 * a ready-made image of w^T, pa2d_gemm_weight_image(w, K, 1, img, bytes, K, N, engine). */
size_t pa2d_gemm_bwd_data_workspace(int N, int K, int engine);
int pa2d_gemm_bwd_data(const float* dy, long long lddy, const float* w, long long ldw, const float* pre,
                       long long ldpre, int act, float* dx, long long lddx, const void* wimg, void* ws, size_t ws_bytes,
                       int M, int N, int K, int engine, pa2d_stream_t stream);
size_t pa2d_gemm_bwd_weight_workspace(int M, int N, int K, int engine);
/* dw[N,K] (+)= dy[M,N]^T . x[M,K];  db[N] (+)= column sums of dy (db may be NULL) */
int pa2d_gemm_bwd_weight(const float* dy, long long lddy, const float* x, long long ldx, float* dw, float* db,
                         void* ws, size_t ws_bytes, int M, int N, int K, int accumulate, int engine,
                         pa2d_stream_t stream);

/* ---- in_project_x / in_project_fx: two Conv2d(C, C, 3, 1, 1) on the same input,
 * Physics_Attention.py:74-75,91-97, as ONE implicit GEMM on the NHWC ([B,N,C]) tensor.
 * out[B*H*W, 2C] = [x_mid | fx_mid] (heads are channel groups h*D..h*D+D-1 of each half).
 * Weights in the checkpoint layout [C_out, C_in, 3, 3].  Requires C % 16 == 0.
 * ev_start / ev_stop: optional hipEvent_t (NULL = none) recorded on `stream` immediately around the
 * implicit-GEMM launch (forward: the [B*N,9C]x[9C,2C] product; backward: the data-gradient product),
 * so a benchmark can time that kernel alone without a profiler. */
size_t pa2d_conv3x3x2_workspace(int B, int H, int W, int C, int engine);       /* backward */
size_t pa2d_conv3x3x2_fwd_workspace(int B, int H, int W, int C, int engine);   /* forward  */
/* Packed weights.  The implicit GEMM reads the two [C,C,3,3] kernels from a K-major pack whose layout depends on
 * the tile / K-step chosen for (B,H,W,C) and on the engine.  By default fwd/bwd build it per call into `ws`
 * (weights change every optimizer step).  A caller that knows the weights are constant over several calls (the
 * T/step calls of one training iteration, a rollout) packs once with pa2d_conv3x3x2_pack and passes the result
 * as `prepacked`; direction 0 = forward pack, 1 = data-gradient pack (taps flipped, in/out swapped). */
size_t pa2d_conv3x3x2_pack_bytes(int C);
int pa2d_conv3x3x2_pack(const float* wx, const float* wf, void* pack, size_t pack_bytes, int B, int H, int W,
                        int C, int direction, int engine, pa2d_stream_t stream);
int pa2d_conv3x3x2_fwd(const float* xn, const float* wx, const float* bx, const float* wf, const float* bf,
                       float* out, const void* prepacked /* NULL = pack here */, void* ws, size_t ws_bytes,
                       int B, int H, int W, int C, int engine, pa2d_stream_t stream, void* ev_start, void* ev_stop);
/* dxn may be NULL (input needs no gradient); dwx/dbx/dwf/dbf (+)= per `accumulate` */
int pa2d_conv3x3x2_bwd(const float* dout, const float* xn, const float* wx, const float* wf, float* dxn,
                       float* dwx, float* dbx, float* dwf, float* dbf, const void* prepacked /* NULL = pack here */,
                       void* ws, size_t ws_bytes, int B, int H, int W, int C, int accumulate, int engine,
                       pa2d_stream_t stream, void* ev_start, void* ev_stop);

/* ---- slice: softmax((x_mid . Ws^T + bs) / clamp(temperature, .1, 5)) and the weighted scatter of
 * N points into M tokens, Physics_Attention.py:98-101.  Emits per-chunk partial sums
 * spart [B,heads,nchunk,M,D] and npart [B,heads,nchunk,M] (npart NULL = skip), nchunk =
 * pa2d_slice_nchunk(B,N,heads); `v` is fx_mid in the forward and dY in backward phase A.
 * D in {8,16,32,64}, M <= 128.  clamp_temperature: 1 = clamp(temperature, 0.1, 5) as the structured-mesh
 * attention does (Physics_Attention.py:98-99); 0 = raw temperature (irregular mesh, :40).
 * `engine` (here, on de-slice and on the slice backward): PA2D_ENGINE_F32 = every contraction on the exact-fp32 matrix
 * instruction (v_mfma_f32_16x16x4_f32); PA2D_ENGINE_SPLIT / _BF16 = bf16 MFMA on exact 3-plane operand splits with fp32
 * accumulation (24-bit significand; the same parity tolerances).  The bf16-storage variants below have no such argument.
 * ev_start / ev_stop (here and on de-slice / slice backward): optional hipEvent_t recorded on `stream` right around
 * the point kernel, as for the conv. */
int pa2d_slice_nchunk(int B, int N, int heads);
int pa2d_slice_scatter(const float* xm, long long ldx, const float* v, long long ldv, const float* ws,
                       const float* bs, const float* temperature, float* spart, float* npart, int B, int N,
                       int heads, int D, int M, int clamp_temperature, int engine, pa2d_stream_t stream, void* ev_start,
                       void* ev_stop);

/* ---- token attention among the M slice tokens of each (batch, head): normalisation by
 * (slice_norm + 1e-5), to_q/to_k/to_v, softmax(q k^T D^-0.5), attn.v — Physics_Attention.py:102-111.
 * Outputs s (raw sums), nrm, o (out_slice_token), all [B*heads, M, (D)]. */
size_t pa2d_token_attn_lds_bytes(int M, int D, int backward);
int pa2d_token_attn_fwd(const float* spart, const float* npart, const float* wq, const float* wk,
                        const float* wv, float* s, float* nrm, float* o, int BH, int nchunk, int M, int D,
                        pa2d_stream_t stream);
size_t pa2d_token_attn_bwd_workspace(int BH, int D);
int pa2d_token_attn_bwd(const float* s, const float* nrm, const float* wq, const float* wk, const float* wv,
                        const float* dopart, float* ds, float* dn, float* dwq, float* dwk, float* dwv, void* ws,
                        size_t ws_bytes, int BH, int nchunk, int M, int D, int accumulate, pa2d_stream_t stream);

/* ---- de-slice: out_x = slice_weights . out_slice_token, written directly as [B,N,(h d)]
 * (Physics_Attention.py:116-117); slice weights are recomputed from x_mid, never stored. */
int pa2d_deslice_fwd(const float* xm, long long ldx, const float* o, const float* ws, const float* bs,
                     const float* temperature, float* y, long long ldy, int B, int N, int heads, int D, int M,
                     int clamp_temperature, int engine, pa2d_stream_t stream, void* ev_start, void* ev_stop);

/* ---- backward of slice + de-slice w.r.t. the points (autograd of
 * Physics_Attention.py:98-101,116): given dY, O, dS, dn produces dx_mid, dfx_mid and the fully
 * reduced dWs [M,D], dbs [M], dtemperature [heads] (clamp mask applied; (+)= per `accumulate`). */
size_t pa2d_slice_bwd_workspace(int B, int N, int heads, int D, int M);
int pa2d_slice_bwd_points(const float* xm, long long ldx, const float* fm, long long ldf, const float* dy,
                          long long lddy, const float* ws, const float* bs, const float* temperature,
                          const float* o, const float* ds, const float* dn, float* dxm, long long lddx,
                          float* dfm, long long lddf, float* dws, float* dbs, float* dtemperature, void* ws_buf,
                          size_t ws_bytes, int B, int N, int heads, int D, int M, int clamp_temperature,
                          int accumulate, int engine, pa2d_stream_t stream, void* ev_start, void* ev_stop);

/* ---- output head mlp2 = nn.Linear(C, out_dim), out_dim <= 8 (…_2D.py:66,73) */
int pa2d_head_fwd(const float* xn, const float* w, const float* b, float* y, int rows, int C, int out_dim,
                  pa2d_stream_t stream);
size_t pa2d_head_bwd_workspace(int rows, int C, int out_dim);
int pa2d_head_bwd(const float* dy, const float* xn, const float* w, float* dxn, float* dw, float* db, void* ws,
                  size_t ws_bytes, int rows, int C, int out_dim, int accumulate, pa2d_stream_t stream);

/* ---- elementwise out = dy * act'(pre): backward of the activation of a generic Linear+act layer
 * (MLP hidden layers with n_layers > 0, …_2D.py:28,32-36; not used by the NS/Darcy configurations) */
int pa2d_act_bwd(const float* dy, const float* pre, float* out, long long n, int act, pa2d_stream_t stream);

/* ---- optimizer / loss side of the exp_ns iteration over flat fp32 buffers
 * (exp_ns.py:198-218: TestLoss rel-L2 sum, clip_grad_norm_, AdamW.step; lr/beta1 of the step come from
 * OneCycleLR on the host).  Buffers 16-byte aligned. */
size_t pa2d_sumsq_workspace(long long n);
int pa2d_sumsq(const float* g, long long n, float* out, void* ws, size_t ws_bytes, pa2d_stream_t stream);
/* one multi-tensor AdamW update of p (decoupled weight decay; bias corrections for step_index >= 1 are evaluated
 * in double on the host, like torch's Python scalars);
 * gnorm_sq (device scalar, may be NULL) + max_norm > 0 apply clip_grad_norm_'s coefficient on the fly */
int pa2d_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int step_index, const float* gnorm_sq,
                    float max_norm, pa2d_stream_t stream);
/* per-sample ||pred-y||, ||y|| and their ratio (utils/testloss.py:31-42), and the gradient w.r.t. pred:
 * dpred[b] = gout[b] * (pred_b - y_b) / (dnorm_b * ynorm_b); gout is the per-sample upstream gradient [B];
 * a sample with dnorm_b == 0 gets the zero sub-gradient (as torch.norm's backward), never inf/NaN */
int pa2d_rel_l2_fwd(const float* pred, const float* y, float* dnorm, float* ynorm, float* ratio, int B,
                    long long L, pa2d_stream_t stream);
int pa2d_rel_l2_bwd(const float* pred, const float* y, const float* dnorm, const float* ynorm,
                    const float* gout, float* dpred, int B, long long L, pa2d_stream_t stream);

/* ==== operand-planes interface of the bf16 engines (fp32 storage; PA2D_ENGINE_SPLIT: 3 planes, PA2D_ENGINE_BF16: 1).
 * The conv GEMMs of these engines stage their activation operand as a bf16 plane image [row][C/32][NT][32].  By default
 * pa2d_conv3x3x2_fwd/bwd make it from the fp32 tensor in a pre-pass; with the entry points below the PRODUCER of the
 * operand writes the image and nothing else: LayerNorm forward (…_2D.py:70 feeding Physics_Attention.py:94,96) and the
 * slice backward (autograd of Physics_Attention.py:98-101 feeding the autograd of :94,96), which also yields the conv
 * bias gradients (`nrm` [B*heads, M] = the forward's slice norms, pa2d_token_attn_fwd: the column sums of dF are
 * sum_m nrm[m] dS[m][:], those of dX sum_m dbs_partial[m] Ws[m][:] — token-level sums, no pass over the points).
 * Use when pa2d_conv3x3x2_planes_mask(...) == 7. */
size_t pa2d_planes_bytes(long long rows, int C, int engine);
int pa2d_conv3x3x2_planes_mask(int B, int H, int W, int C, int engine);
int pa2d_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, void* planes, float* mean,
                              float* rstd, int rows, int C, float eps, int engine, pa2d_stream_t stream);
int pa2d_conv3x3x2_fwd_planes(const void* xn_planes, const float* wx, const float* bx, const float* wf, const float* bf,
                              float* out, const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                              int engine, pa2d_stream_t stream, void* ev_start, void* ev_stop);
size_t pa2d_conv3x3x2_workspace_planes(int B, int H, int W, int C, int engine);
int pa2d_conv3x3x2_bwd_planes(const void* dout_planes, const void* xn_planes, const float* wx, const float* wf, float* dxn,
                              float* dwx, float* dwf, const void* prepacked, void* ws, size_t ws_bytes, int B, int H,
                              int W, int C, int accumulate, int engine, pa2d_stream_t stream, void* ev_start,
                              void* ev_stop);
int pa2d_slice_bwd_points_planes(const float* xm, long long ldx, const float* fm, long long ldf, const float* dy,
                                 long long lddy, const float* ws, const float* bs, const float* temperature,
                                 const float* o, const float* ds, const float* dn, const float* nrm, void* dxf_planes,
                                 float* dbx, float* dbf, float* dws, float* dbs, float* dtemperature, void* ws_buf,
                                 size_t ws_bytes, int B, int N, int heads, int D, int M, int clamp_temperature,
                                 int accumulate, int engine, pa2d_stream_t stream, void* ev_start, void* ev_stop);

/* ==== bf16-STORAGE variants (BASELINE configs[2] "NS 64x64 bf16 ... DDP" and configs[4] "Darcy ... bf16"; the reference
 * would reach these numerics with torch.autocast(bfloat16) around model/Transolver_Structured_Mesh_2D.py:202-220).
 * Same stages, same argument order as the fp32 entry points above, but every ACTIVATION pointer (inputs, outputs,
 * saved tensors, inter-kernel gradients: the `void*` arguments) holds bf16; parameters, biases, LayerNorm statistics,
 * slice partial sums / norms, token tensors and every parameter gradient stay fp32; all accumulation is fp32.  GEMMs
 * use ONE bf16 MFMA term (the arithmetic of PA2D_ENGINE_BF16; conv weight packs are made by pa2d_conv3x3x2_pack_bf16).
 * ld* are in elements.  Dense layers need K % 32 == 0 (and N, K % 32 == 0 for the weight
 * gradient, whose operands must be contiguous), the conv C % 32 == 0: otherwise PA2D_ERR_UNSUPPORTED. */
int pa2d_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                            int rows, int C, float eps, pa2d_stream_t stream);
int pa2d_layernorm_bwd_bf16(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                            const void* dres, void* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                            int rows, int C, int accumulate, pa2d_stream_t stream);
int pa2d_gemm_bias_act_fwd_bf16(const void* x, long long ldx, const float* w, long long ldw, const float* bias,
                                const void* res, long long ldres, void* y, long long ldy, void* pre, long long ldpre,
                                int M, int N, int K, int act, pa2d_stream_t stream);
int pa2d_gemm_bwd_data_bf16(const void* dy, long long lddy, const float* w, long long ldw, const void* pre,
                            long long ldpre, int act, void* dx, long long lddx, float* wt_ws, int M, int N, int K,
                            pa2d_stream_t stream);
size_t pa2d_gemm_bwd_weight_workspace_bf16(int M, int N, int K);
int pa2d_gemm_bwd_weight_bf16(const void* dy, long long lddy, const void* x, long long ldx, float* dw, float* db,
                              void* ws, size_t ws_bytes, int M, int N, int K, int accumulate, pa2d_stream_t stream);
/* weight pack for pa2d_conv3x3x2_{fwd,bwd}_bf16 (`prepacked`): pa2d_conv3x3x2_pack_bytes(C) bytes; direction as above */
int pa2d_conv3x3x2_pack_bf16(const float* wx, const float* wf, void* pack, size_t pack_bytes, int C, int direction,
                             pa2d_stream_t stream);
size_t pa2d_conv3x3x2_workspace_bf16(int B, int H, int W, int C);       /* backward */
size_t pa2d_conv3x3x2_fwd_workspace_bf16(int B, int H, int W, int C);   /* forward  */
int pa2d_conv3x3x2_fwd_bf16(const void* xn, const float* wx, const float* bx, const float* wf, const float* bf,
                            void* out, const void* prepacked, void* ws, size_t ws_bytes, int B, int H, int W, int C,
                            pa2d_stream_t stream, void* ev_start, void* ev_stop);
int pa2d_conv3x3x2_bwd_bf16(const void* dout, const void* xn, const float* wx, const float* wf, void* dxn, float* dwx,
                            float* dbx, float* dwf, float* dbf, const void* prepacked, void* ws, size_t ws_bytes,
                            int B, int H, int W, int C, int accumulate, pa2d_stream_t stream, void* ev_start,
                            void* ev_stop);
int pa2d_slice_scatter_bf16(const void* xm, long long ldx, const void* v, long long ldv, const float* ws,
                            const float* bs, const float* temperature, float* spart, float* npart, int B, int N,
                            int heads, int D, int M, int clamp_temperature, pa2d_stream_t stream, void* ev_start,
                            void* ev_stop);
int pa2d_deslice_fwd_bf16(const void* xm, long long ldx, const float* o, const float* ws, const float* bs,
                          const float* temperature, void* y, long long ldy, int B, int N, int heads, int D, int M,
                          int clamp_temperature, pa2d_stream_t stream, void* ev_start, void* ev_stop);
int pa2d_slice_bwd_points_bf16(const void* xm, long long ldx, const void* fm, long long ldf, const void* dy,
                               long long lddy, const float* ws, const float* bs, const float* temperature,
                               const float* o, const float* ds, const float* dn, void* dxm, long long lddx,
                               void* dfm, long long lddf, float* dws, float* dbs, float* dtemperature, void* ws_buf,
                               size_t ws_bytes, int B, int N, int heads, int D, int M, int clamp_temperature,
                               int accumulate, pa2d_stream_t stream, void* ev_start, void* ev_stop);
int pa2d_head_fwd_bf16(const void* xn, const float* w, const float* b, float* y, int rows, int C, int out_dim,
                       pa2d_stream_t stream);
int pa2d_head_bwd_bf16(const float* dy, const void* xn, const float* w, void* dxn, float* dw, float* db, void* ws,
                       size_t ws_bytes, int rows, int C, int out_dim, int accumulate, pa2d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PA2D_H */
