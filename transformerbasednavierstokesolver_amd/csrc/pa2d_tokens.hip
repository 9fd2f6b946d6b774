// Token attention among the M slice tokens of one (batch, head): Physics_Attention.py:102-111 and
// its backward.  M <= 128 tokens of D <= 64 channels: the whole problem
// (T, Q, K, V, the M x M attention matrix) lives in the LDS of one workgroup; 7 reference launches
// (3 linears, 2 matmuls, softmax, normalisation) collapse into one.  ~7 MFLOP per sample-layer, so
// this stage is latency- not throughput-critical; plain fp32 FMAs.
#include "pa2d_internal.h"

#define SLICE_EPS 1e-5f

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st);

// sum of the per-chunk partials of one element, 4 loads in flight, fixed order (deterministic)
__device__ __forceinline__ float sum_chunks(const float* __restrict__ base, int nchunk, size_t stride) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < nchunk; c += 4) {
        s0 += base[(size_t)c * stride];
        s1 += base[(size_t)(c + 1) * stride];
        s2 += base[(size_t)(c + 2) * stride];
        s3 += base[(size_t)(c + 3) * stride];
    }
    for (; c < nchunk; ++c) s0 += base[(size_t)c * stride];
    return (s0 + s1) + (s2 + s3);
}

struct TokParams {
    const float* spart; const float* npart;   // [B*heads, nchunk, M, D], [B*heads, nchunk, M]
    const float* wq; const float* wk; const float* wv;   // [D, D] (out, in), shared by all heads
    float* s; float* nrm; float* o;           // [B*heads, M, D], [B*heads, M], [B*heads, M, D]
    int M, D, nchunk;
};

// LDS carve shared by forward and backward
struct TokLds {
    float *T, *Q, *K, *V, *A, *Wq, *Wk, *Wv, *nr;
    int P, PA;
};
__device__ __forceinline__ TokLds carve(float* smem, int M, int D) {
    TokLds l;
    l.P = D + 1; l.PA = M + 1;
    l.T = smem; l.Q = l.T + M * l.P; l.K = l.Q + M * l.P; l.V = l.K + M * l.P;
    l.A = l.V + M * l.P;
    l.Wq = l.A + M * l.PA; l.Wk = l.Wq + D * l.P; l.Wv = l.Wk + D * l.P;
    l.nr = l.Wv + D * l.P;
    return l;
}
static size_t tok_fwd_floats(int M, int D) { return (size_t)4 * M * (D + 1) + (size_t)M * (M + 1) + 3 * D * (D + 1) + M; }

// shared recompute: T = S/(n+eps), Q/K/V projections, A = softmax(Q K^T / sqrt(D))
__device__ __forceinline__ void tokens_forward_core(const TokLds& l, int M, int D, int tid) {
    const int P = l.P, PA = l.PA;
    for (int i = tid; i < M * D; i += 256) {
        const int m = i / D, e = i % D;
        float q = 0.f, k = 0.f, v = 0.f;
        for (int d = 0; d < D; ++d) {
            const float t = l.T[m * P + d];
            q += t * l.Wq[e * P + d];
            k += t * l.Wk[e * P + d];
            v += t * l.Wv[e * P + d];
        }
        l.Q[m * P + e] = q; l.K[m * P + e] = k; l.V[m * P + e] = v;
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    for (int i = tid; i < M * M; i += 256) {
        const int m = i / M, n = i % M;
        float a = 0.f;
        for (int e = 0; e < D; ++e) a += l.Q[m * P + e] * l.K[n * P + e];
        l.A[m * PA + n] = a * scale;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int m = wave; m < M; m += 4) {
        float mx = -1e30f;
        for (int n = lane; n < M; n += 64) mx = fmaxf(mx, l.A[m * PA + n]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int n = lane; n < M; n += 64) {
            const float e = expf(l.A[m * PA + n] - mx);
            l.A[m * PA + n] = e;
            sm += e;
        }
        sm = wave_sum(sm);
        const float inv = 1.0f / sm;
        for (int n = lane; n < M; n += 64) l.A[m * PA + n] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void token_attn_fwd_kernel(const TokParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = p.M, D = p.D, tid = threadIdx.x;
    const TokLds l = carve(smem, M, D);
    const size_t bh = blockIdx.x;
    for (int i = tid; i < D * D; i += 256) {
        const int e = i / D, d = i % D;
        l.Wq[e * l.P + d] = p.wq[i]; l.Wk[e * l.P + d] = p.wk[i]; l.Wv[e * l.P + d] = p.wv[i];
    }
    for (int m = tid; m < M; m += 256) {
        const float s = sum_chunks(p.npart + bh * p.nchunk * M + m, p.nchunk, M);
        l.nr[m] = s;
        p.nrm[bh * M + m] = s;
    }
    __syncthreads();
    for (int i = tid; i < M * D; i += 256) {
        const float s = sum_chunks(p.spart + bh * p.nchunk * M * D + i, p.nchunk, (size_t)M * D);
        p.s[bh * M * D + i] = s;
        l.T[(i / D) * l.P + (i % D)] = s / (l.nr[i / D] + SLICE_EPS);
    }
    __syncthreads();
    tokens_forward_core(l, M, D, tid);
    for (int i = tid; i < M * D; i += 256) {
        const int m = i / D, d = i % D;
        float o = 0.f;
        for (int n = 0; n < M; ++n) o += l.A[m * l.PA + n] * l.V[n * l.P + d];
        p.o[bh * M * D + i] = o;
    }
}

struct TokBwdParams {
    const float* s; const float* nrm;         // saved by forward
    const float* wq; const float* wk; const float* wv;
    const float* dopart;                      // [B*heads, nchunk, M, D]  (phase-A partials of dO)
    float* ds; float* dn;                     // [B*heads, M, D], [B*heads, M]
    float* dwpart;                            // [B*heads, 3, D, D]
    int M, D, nchunk;
};
static size_t tok_bwd_floats(int M, int D) { return tok_fwd_floats(M, D) + (size_t)3 * M * (D + 1); }

__global__ __launch_bounds__(256) void token_attn_bwd_kernel(const TokBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = p.M, D = p.D, tid = threadIdx.x;
    const TokLds l = carve(smem, M, D);
    const int P = l.P, PA = l.PA;
    float* const G1 = l.nr + M;          // dO, later dQ
    float* const G2 = G1 + M * P;        // dV
    float* const G3 = G2 + M * P;        // dK
    const size_t bh = blockIdx.x;
    for (int i = tid; i < D * D; i += 256) {
        const int e = i / D, d = i % D;
        l.Wq[e * P + d] = p.wq[i]; l.Wk[e * P + d] = p.wk[i]; l.Wv[e * P + d] = p.wv[i];
    }
    for (int m = tid; m < M; m += 256) l.nr[m] = p.nrm[bh * M + m];
    __syncthreads();
    for (int i = tid; i < M * D; i += 256) {
        const int m = i / D, d = i % D;
        l.T[m * P + d] = p.s[bh * M * D + i] / (l.nr[m] + SLICE_EPS);
        G1[m * P + d] = sum_chunks(p.dopart + bh * p.nchunk * M * D + i, p.nchunk, (size_t)M * D);
    }
    __syncthreads();
    tokens_forward_core(l, M, D, tid);
    // dV[n][d] = sum_m A[m][n] dO[m][d]
    for (int i = tid; i < M * D; i += 256) {
        const int n = i / D, d = i % D;
        float g = 0.f;
        for (int m = 0; m < M; ++m) g += l.A[m * PA + n] * G1[m * P + d];
        G2[n * P + d] = g;
    }
    __syncthreads();
    // per row m: dA[m][n] = dO[m].V[n]; dP = A * (dA - sum_n dA*A), written over A
    const int lane = tid & 63, wave = tid >> 6;
    for (int m = wave; m < M; m += 4) {
        float da[2], rs = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = lane + 64 * k;
            da[k] = 0.f;
            if (n < M) {
                for (int d = 0; d < D; ++d) da[k] += G1[m * P + d] * l.V[n * P + d];
                rs += da[k] * l.A[m * PA + n];
            }
        }
        rs = wave_sum(rs);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = lane + 64 * k;
            if (n < M) l.A[m * PA + n] *= (da[k] - rs);
        }
    }
    __syncthreads();
    // dQ = scale * dP K (into G1, dO is dead), dK = scale * dP^T Q (G3)
    const float scale = rsqrtf((float)D);
    for (int i = tid; i < M * D; i += 256) {
        const int m = i / D, e = i % D;
        float gq = 0.f, gk = 0.f;
        for (int n = 0; n < M; ++n) {
            gq += l.A[m * PA + n] * l.K[n * P + e];
            gk += l.A[n * PA + m] * l.Q[n * P + e];
        }
        G3[m * P + e] = gk * scale;
        // dQ must not overwrite G1 before every thread finished reading dO: dO is no longer read here
        G1[m * P + e] = gq * scale;
    }
    __syncthreads();
    // weight-gradient partials of this (b,h): dWq[e][d] = sum_m dQ[m][e] T[m][d]
    float* dwp = p.dwpart + bh * 3 * D * D;
    for (int i = tid; i < D * D; i += 256) {
        const int e = i / D, d = i % D;
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int m = 0; m < M; ++m) {
            const float t = l.T[m * P + d];
            gq += G1[m * P + e] * t;
            gk += G3[m * P + e] * t;
            gv += G2[m * P + e] * t;
        }
        dwp[i] = gq; dwp[D * D + i] = gk; dwp[2 * D * D + i] = gv;
    }
    // dT = dQ Wq + dK Wk + dV Wv ; dS = dT/(n+eps) ; dn = -sum_d dT*S/(n+eps)^2  (row per thread group)
    for (int i = tid; i < M * D; i += 256) {
        const int m = i / D, d = i % D;
        float g = 0.f;
        for (int e = 0; e < D; ++e)
            g += G1[m * P + e] * l.Wq[e * P + d] + G3[m * P + e] * l.Wk[e * P + d] + G2[m * P + e] * l.Wv[e * P + d];
        l.Q[m * P + d] = g;    // Q is dead: reuse for dT
        p.ds[bh * M * D + i] = g / (l.nr[m] + SLICE_EPS);
    }
    __syncthreads();
    for (int m = tid; m < M; m += 256) {
        const float den = l.nr[m] + SLICE_EPS;
        float g = 0.f;
        for (int d = 0; d < D; ++d) g += l.Q[m * P + d] * l.T[m * P + d];   // T*den = S
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
    if (smem > 160 * 1024 || M > 128) return PA2D_ERR_UNSUPPORTED;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_attn_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    TokParams p;
    p.spart = spart; p.npart = npart; p.wq = wq; p.wk = wk; p.wv = wv; p.s = s; p.nrm = nrm; p.o = o;
    p.M = M; p.D = D; p.nchunk = nchunk;
    hipLaunchKernelGGL(token_attn_fwd_kernel, dim3(BH), dim3(256), smem, st, p);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

size_t pa2d_token_attn_bwd_workspace(int BH, int D) { return sizeof(float) * ((size_t)BH + 1) * 3 * D * D; }

// dopart: phase-A partials of dO = W^T dY.  Outputs ds, dn per (b,h) and fully reduced dwq/dwk/dwv [D,D].
int pa2d_token_attn_bwd(const float* s, const float* nrm, const float* wq, const float* wk, const float* wv,
                        const float* dopart, float* ds, float* dn, float* dwq, float* dwk, float* dwv, void* ws,
                        size_t ws_bytes, int BH, int nchunk, int M, int D, hipStream_t st) {
    const size_t smem = pa2d_token_attn_lds_bytes(M, D, 1);
    if (smem > 160 * 1024 || M > 128) return PA2D_ERR_UNSUPPORTED;
    if (ws_bytes < pa2d_token_attn_bwd_workspace(BH, D)) return PA2D_ERR_WORKSPACE;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_attn_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    TokBwdParams p;
    p.s = s; p.nrm = nrm; p.wq = wq; p.wk = wk; p.wv = wv; p.dopart = dopart; p.ds = ds; p.dn = dn;
    p.dwpart = (float*)ws; p.M = M; p.D = D; p.nchunk = nchunk;
    hipLaunchKernelGGL(token_attn_bwd_kernel, dim3(BH), dim3(256), smem, st, p);
    PA2D_CHECK_LAUNCH();
    float* tail = (float*)ws + (size_t)BH * 3 * D * D;
    int rc = pa2d_launch_reduce((const float*)ws, BH, (long long)3 * D * D, tail, st);
    if (rc) return rc;
    const size_t n = sizeof(float) * D * D;
    hipError_t e = hipMemcpyAsync(dwq, tail, n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(dwk, tail + D * D, n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(dwv, tail + 2 * D * D, n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    return PA2D_OK;
}

}  // extern "C"
