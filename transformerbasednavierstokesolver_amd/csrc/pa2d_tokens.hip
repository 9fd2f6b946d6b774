// Token attention among the M slice tokens of one (batch, head): Physics_Attention.py:102-111 and
// its backward.  M <= 128 tokens of D <= 64 channels: the whole problem
// (T, Q, K, V, the M x M attention matrix) lives in the LDS of one workgroup; 7 reference launches
// (3 linears, 2 matmuls, softmax, normalisation) collapse into one.  ~7 MFLOP per sample-layer, so the
// stage is latency-critical (it is 1/5 of a single-trajectory rollout step if done with scalar FMAs):
//  * every contraction runs on v_mfma_f32_16x16x4_f32 with both operands read from LDS; row pitches
//    D+4 / Mp+4 (4 x odd) make the k-contiguous fragment reads (16 rows x 4 k) hit 64 distinct banks;
//  * 8 waves per workgroup, one 16x16 output tile per wave and phase at the NS shape (M=64, D=32);
//  * the per-chunk partial sums of the slice scatter are fetched with all loads of an element in
//    flight (branch-free buffer loads) and added in a fixed order, so results stay deterministic.
#include "pa2d_internal.h"

#define SLICE_EPS 1e-5f
#define TOK_THREADS 512
#define TOK_WAVES 8

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st);

__device__ __forceinline__ f32x4 tok_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sum over the chunk partials of element `off` (bytes) of a [nchunk][stride] panel: 16 loads in flight
// per pass (chunks >= nchunk read 0 through the descriptor's range check), fixed summation tree.
__device__ __forceinline__ float sum_chunks(__amdgpu_buffer_rsrc_t r, unsigned off, int nchunk, unsigned stride) {
    float tot = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = buf_load1(r, c0 + j < nchunk ? off + (unsigned)(c0 + j) * stride : OOB_OFF);
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int j = 0; j < w; ++j) v[j] += v[j + w];
        tot += v[0];
    }
    return tot;
}

struct TokParams {
    const float* spart; const float* npart;   // [B*heads, nchunk, M, D], [B*heads, nchunk, M]
    const float* wq; const float* wk; const float* wv;   // [D, D] (out, in), shared by all heads
    float* s; float* nrm; float* o;           // [B*heads, M, D], [B*heads, M], [B*heads, M, D]
    int M, D, nchunk;
};

// LDS carve shared by forward and backward.  Mp / Dp = M / D rounded up to the 16-wide MFMA tile;
// padding rows of T and of the weights are zero, so padded outputs are exact zeros or masked.
struct TokLds {
    float *T, *Q, *K, *Vt, *A, *Wq, *Wk, *Wv, *nr;
    int P, PA, Mp, Dp;
};
static __host__ __device__ inline int up16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ TokLds carve(float* smem, int M, int D) {
    TokLds l;
    l.Mp = up16(M); l.Dp = up16(D);
    l.P = D + 4; l.PA = l.Mp + 4;
    l.T = smem; l.Q = l.T + l.Mp * l.P; l.K = l.Q + l.Mp * l.P;
    l.Vt = l.K + l.Mp * l.P;                  // V transposed: Vt[d][n]
    l.A = l.Vt + l.Dp * l.PA;
    l.Wq = l.A + l.Mp * l.PA; l.Wk = l.Wq + l.Dp * l.P; l.Wv = l.Wk + l.Dp * l.P;
    l.nr = l.Wv + l.Dp * l.P;
    return l;
}
static size_t tok_fwd_floats(int M, int D) {
    const size_t Mp = up16(M), Dp = up16(D), P = D + 4, PA = Mp + 4;
    return 3 * Mp * P + Dp * PA + Mp * PA + 3 * Dp * P + Mp;
}

// C[i][j] = sum_{k<K} A[i*sai + k*sak] * B[j*sbj + k*sbk] for the 16x16 tiles of an (I16 x J16) tile grid,
// tiles dealt round-robin to the waves; epi(i, j, value) receives each element once.  K % 4 == 0.
template <class Epi>
__device__ __forceinline__ void lds_gemm(const float* A, int sai, int sak, const float* B, int sbj, int sbk,
                                         int I16, int J16, int K, int wave, int lane, int first, Epi epi) {
    const int li = lane & 15, kq = lane >> 4;
    for (int t = (wave + TOK_WAVES - first % TOK_WAVES) % TOK_WAVES; t < I16 * J16; t += TOK_WAVES) {
        const int it = t / J16, jt = t - it * J16;
        const float* a = A + (16 * it + li) * sai + kq * sak;
        const float* b = B + (16 * jt + li) * sbj + kq * sbk;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; k += 4) acc = tok_mfma(a[k * sak], b[k * sbk], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(16 * it + 4 * kq + r, 16 * jt + li, acc[r]);
    }
}

__device__ __forceinline__ void load_weights(const TokLds& l, const float* wq, const float* wk, const float* wv,
                                             int D, int tid) {
    for (int i = tid; i < l.Dp * D; i += TOK_THREADS) {
        const int e = i / D, d = i - e * D;
        const bool ok = e < D;
        l.Wq[e * l.P + d] = ok ? wq[i] : 0.f;
        l.Wk[e * l.P + d] = ok ? wk[i] : 0.f;
        l.Wv[e * l.P + d] = ok ? wv[i] : 0.f;
    }
}

// shared recompute from T (and the weights) in LDS: Q, K, V^T projections and A = softmax(Q K^T / sqrt(D));
// rows and columns of A beyond M are exact zeros.
__device__ __forceinline__ void tokens_forward_core(const TokLds& l, int M, int D, int tid) {
    const int P = l.P, PA = l.PA, M16 = l.Mp / 16, D16 = l.Dp / 16;
    const int lane = tid & 63, wave = tid >> 6;
    float* const Q = l.Q; float* const Kk = l.K; float* const Vt = l.Vt; float* const A = l.A;
    lds_gemm(l.T, P, 1, l.Wq, P, 1, M16, D16, D, wave, lane, 0,
             [=](int m, int e, float v) { if (e < D) Q[m * P + e] = v; });
    lds_gemm(l.T, P, 1, l.Wk, P, 1, M16, D16, D, wave, lane, M16 * D16,
             [=](int m, int e, float v) { if (e < D) Kk[m * P + e] = v; });
    lds_gemm(l.Wv, P, 1, l.T, P, 1, D16, M16, D, wave, lane, 2 * M16 * D16,
             [=](int e, int m, float v) { Vt[e * PA + m] = v; });
    __syncthreads();
    const float scale = rsqrtf((float)D);
    lds_gemm(Q, P, 1, Kk, P, 1, M16, M16, D, wave, lane, 0,
             [=](int m, int n, float v) { A[m * PA + n] = v * scale; });
    __syncthreads();
    for (int m = wave; m < l.Mp; m += TOK_WAVES) {
        float z[2], mx = -1e30f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = lane + 64 * k;
            z[k] = (m < M && n < M) ? A[m * PA + n] : -1e30f;
            mx = fmaxf(mx, z[k]);
        }
        mx = wave_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = lane + 64 * k;
            z[k] = (m < M && n < M) ? expf(z[k] - mx) : 0.f;
            sm += z[k];
        }
        sm = wave_sum(sm);
        const float inv = m < M ? 1.0f / sm : 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = lane + 64 * k;
            if (n < l.Mp) A[m * PA + n] = z[k] * inv;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(TOK_THREADS) void token_attn_fwd_kernel(const TokParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = p.M, D = p.D, tid = threadIdx.x;
    const TokLds l = carve(smem, M, D);
    const size_t bh = blockIdx.x;
    const unsigned MD4 = (unsigned)(M * D) * 4u;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.spart + bh * p.nchunk * M * D, (unsigned)p.nchunk * MD4);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(p.npart + bh * p.nchunk * M, (unsigned)(p.nchunk * M) * 4u);
    load_weights(l, p.wq, p.wk, p.wv, D, tid);
    for (int m = tid; m < l.Mp; m += TOK_THREADS) {
        const float s = m < M ? sum_chunks(rn, m * 4u, p.nchunk, M * 4u) : 0.f;
        l.nr[m] = s;
        if (m < M) p.nrm[bh * M + m] = s;
    }
    __syncthreads();
    for (int i = tid; i < l.Mp * D; i += TOK_THREADS) {
        const int m = i / D, d = i - m * D;
        float t = 0.f;
        if (m < M) {
            const float s = sum_chunks(rs, i * 4u, p.nchunk, MD4);
            p.s[bh * M * D + i] = s;
            t = s / (l.nr[m] + SLICE_EPS);
        }
        l.T[m * l.P + d] = t;
    }
    __syncthreads();
    tokens_forward_core(l, M, D, tid);
    float* const o = p.o + bh * M * D;
    lds_gemm(l.A, l.PA, 1, l.Vt, l.PA, 1, l.Mp / 16, l.Dp / 16, l.Mp, tid >> 6, tid & 63, 0,
             [=](int m, int d, float v) { if (m < M && d < D) o[m * D + d] = v; });
}

struct TokBwdParams {
    const float* s; const float* nrm;         // saved by forward
    const float* wq; const float* wk; const float* wv;
    const float* dopart;                      // [B*heads, nchunk, M, D]  (phase-A partials of dO)
    float* ds; float* dn;                     // [B*heads, M, D], [B*heads, M]
    float* dwpart;                            // [B*heads, 3, D, D]
    int M, D, nchunk;
};
static size_t tok_bwd_floats(int M, int D) { return tok_fwd_floats(M, D) + (size_t)3 * up16(M) * (D + 4) + up16(M); }

__global__ __launch_bounds__(TOK_THREADS) void token_attn_bwd_kernel(const TokBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = p.M, D = p.D, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TokLds l = carve(smem, M, D);
    const int P = l.P, PA = l.PA, Mp = l.Mp, M16 = l.Mp / 16, D16 = l.Dp / 16;
    float* const G1 = l.nr + Mp;         // dO, later dQ
    float* const G2 = G1 + Mp * P;       // dV
    float* const G3 = G2 + Mp * P;       // O (recomputed), later dK
    float* const rsum = G3 + Mp * P;     // rowsum(dA * A) = rowsum(dO * O)
    float* const A = l.A; float* const Q = l.Q;
    const size_t bh = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.dopart + bh * p.nchunk * M * D, (unsigned)(p.nchunk * M * D) * 4u);
    load_weights(l, p.wq, p.wk, p.wv, D, tid);
    for (int m = tid; m < Mp; m += TOK_THREADS) l.nr[m] = m < M ? p.nrm[bh * M + m] : 0.f;
    __syncthreads();
    for (int i = tid; i < Mp * D; i += TOK_THREADS) {
        const int m = i / D, d = i - m * D;
        float t = 0.f, g = 0.f;
        if (m < M) {
            t = p.s[bh * M * D + i] / (l.nr[m] + SLICE_EPS);
            g = sum_chunks(rd, i * 4u, p.nchunk, (unsigned)(M * D) * 4u);
        }
        l.T[m * P + d] = t;
        G1[m * P + d] = g;
    }
    __syncthreads();
    tokens_forward_core(l, M, D, tid);
    // O = A V (for the softmax-backward row term) and dV[n][d] = sum_m A[m][n] dO[m][d]
    lds_gemm(A, PA, 1, l.Vt, PA, 1, M16, D16, Mp, wave, lane, 0,
             [=](int m, int d, float v) { if (d < D) G3[m * P + d] = v; });
    lds_gemm(A, 1, PA, G1, 1, P, M16, D16, Mp, wave, lane, M16 * D16,
             [=](int n, int d, float v) { if (d < D) G2[n * P + d] = v; });
    __syncthreads();
    for (int m = tid; m < Mp; m += TOK_THREADS) {
        float g = 0.f;
        for (int d = 0; d < D; ++d) g += G1[m * P + d] * G3[m * P + d];
        rsum[m] = g;
    }
    __syncthreads();
    // dA[m][n] = dO[m].V[n];  dP = A * (dA - rowsum), written over A (each element by the lane that read it)
    lds_gemm(G1, P, 1, l.Vt, 1, PA, M16, M16, D, wave, lane, 0,
             [=](int m, int n, float v) { A[m * PA + n] *= (v - rsum[m]); });
    __syncthreads();
    // dQ = scale * dP K (into G1: dO is dead), dK = scale * dP^T Q (into G3: O is dead)
    const float scale = rsqrtf((float)D);
    lds_gemm(A, PA, 1, l.K, 1, P, M16, D16, Mp, wave, lane, 0,
             [=](int m, int e, float v) { if (e < D) G1[m * P + e] = v * scale; });
    lds_gemm(A, 1, PA, Q, 1, P, M16, D16, Mp, wave, lane, M16 * D16,
             [=](int n, int e, float v) { if (e < D) G3[n * P + e] = v * scale; });
    __syncthreads();
    // weight-gradient partials of this (b,h): dWq[e][d] = sum_m dQ[m][e] T[m][d]  (same for K, V)
    float* const dwp = p.dwpart + bh * 3 * D * D;
    lds_gemm(G1, 1, P, l.T, 1, P, D16, D16, Mp, wave, lane, 0,
             [=](int e, int d, float v) { if (e < D && d < D) dwp[e * D + d] = v; });
    lds_gemm(G3, 1, P, l.T, 1, P, D16, D16, Mp, wave, lane, D16 * D16,
             [=](int e, int d, float v) { if (e < D && d < D) dwp[D * D + e * D + d] = v; });
    lds_gemm(G2, 1, P, l.T, 1, P, D16, D16, Mp, wave, lane, 2 * D16 * D16,
             [=](int e, int d, float v) { if (e < D && d < D) dwp[2 * D * D + e * D + d] = v; });
    // dT = dQ Wq + dK Wk + dV Wv  (one accumulator, three operand pairs) ; dS = dT/(n+eps)
    // Q is still read by nobody (dK is done): it receives dT for the dn pass.
    {
        const int li = lane & 15, kq = lane >> 4;
        float* const ds = p.ds + bh * M * D;
        for (int t = (wave + TOK_WAVES - (3 * D16 * D16) % TOK_WAVES) % TOK_WAVES; t < M16 * D16; t += TOK_WAVES) {
            const int it = t / D16, jt = t - it * D16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int ar = (16 * it + li) * P + kq, bc = kq * P + 16 * jt + li;
            for (int k = 0; k < D; k += 4) {
                acc = tok_mfma(G1[ar + k], l.Wq[bc + k * P], acc);
                acc = tok_mfma(G3[ar + k], l.Wk[bc + k * P], acc);
                acc = tok_mfma(G2[ar + k], l.Wv[bc + k * P], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * it + 4 * kq + r, d = 16 * jt + li;
                if (d < D) {
                    Q[m * P + d] = acc[r];
                    if (m < M) ds[m * D + d] = acc[r] / (l.nr[m] + SLICE_EPS);
                }
            }
        }
    }
    __syncthreads();
    for (int m = tid; m < M; m += TOK_THREADS) {
        const float den = l.nr[m] + SLICE_EPS;
        float g = 0.f;
        for (int d = 0; d < D; ++d) g += Q[m * P + d] * l.T[m * P + d];   // T*den = S
        p.dn[bh * M + m] = -g / den;   // -(dT . S)/den^2 with S = T*den
    }
}

extern "C" {

size_t pa2d_token_attn_lds_bytes(int M, int D, int backward) {
    return sizeof(float) * (backward ? tok_bwd_floats(M, D) : tok_fwd_floats(M, D));
}

// spart/npart: partial sums from pa2d_slice_scatter ([B*heads, nchunk, ...]).
// outputs: s (raw slice sums), nrm (slice norms), o (out_slice_token), all [B*heads, M, (D)].
int pa2d_token_attn_fwd(const float* spart, const float* npart, const float* wq, const float* wk, const float* wv,
                        float* s, float* nrm, float* o, int BH, int nchunk, int M, int D, hipStream_t st) {
    const size_t smem = pa2d_token_attn_lds_bytes(M, D, 0);
    if (smem > 160 * 1024 || M > 128 || (D & 3)) return PA2D_ERR_UNSUPPORTED;
    if (BH <= 0) return PA2D_OK;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_attn_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    TokParams p;
    p.spart = spart; p.npart = npart; p.wq = wq; p.wk = wk; p.wv = wv; p.s = s; p.nrm = nrm; p.o = o;
    p.M = M; p.D = D; p.nchunk = nchunk;
    hipLaunchKernelGGL(token_attn_fwd_kernel, dim3(BH), dim3(TOK_THREADS), smem, st, p);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

size_t pa2d_token_attn_bwd_workspace(int BH, int D) { return sizeof(float) * ((size_t)BH + 1) * 3 * D * D; }
// dwq/dwk/dwv (+)= (accumulate != 0: added to what they hold)

// dopart: phase-A partials of dO = W^T dY.  Outputs ds, dn per (b,h) and fully reduced dwq/dwk/dwv [D,D].
int pa2d_token_attn_bwd(const float* s, const float* nrm, const float* wq, const float* wk, const float* wv,
                        const float* dopart, float* ds, float* dn, float* dwq, float* dwk, float* dwv, void* ws,
                        size_t ws_bytes, int BH, int nchunk, int M, int D, int accumulate, hipStream_t st) {
    const size_t smem = pa2d_token_attn_lds_bytes(M, D, 1);
    if (smem > 160 * 1024 || M > 128 || (D & 3)) return PA2D_ERR_UNSUPPORTED;
    if (BH <= 0) {
        if (accumulate) return PA2D_OK;
        const size_t wb = sizeof(float) * (size_t)D * D;
        int rz = pa2d_zero(dwq, wb, st);
        if (!rz) rz = pa2d_zero(dwk, wb, st);
        return rz ? rz : pa2d_zero(dwv, wb, st);
    }
    if (ws_bytes < pa2d_token_attn_bwd_workspace(BH, D)) return PA2D_ERR_WORKSPACE;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_attn_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    TokBwdParams p;
    p.s = s; p.nrm = nrm; p.wq = wq; p.wk = wk; p.wv = wv; p.dopart = dopart; p.ds = ds; p.dn = dn;
    p.dwpart = (float*)ws; p.M = M; p.D = D; p.nchunk = nchunk;
    hipLaunchKernelGGL(token_attn_bwd_kernel, dim3(BH), dim3(TOK_THREADS), smem, st, p);
    PA2D_CHECK_LAUNCH();
    const long long dd = (long long)D * D;
    ReduceSegs segs;
    segs.nseg = 3;
    segs.begin[0] = 0; segs.begin[1] = dd; segs.begin[2] = 2 * dd; segs.begin[3] = 3 * dd; segs.begin[4] = 3 * dd;
    segs.dst[0] = dwq; segs.dst[1] = dwk; segs.dst[2] = dwv; segs.dst[3] = nullptr;
    return pa2d_launch_reduce_segs((const float*)ws, BH, 3 * dd, segs, accumulate, st);
}

}  // extern "C"
