// the optimizer / loss side of one exp_ns iteration (exp_ns.py:198-218,
// utils/testloss.py:31-42) over the SAME flat fp32 buffers the DDP gradient bucket uses:
//   * pa2d_sumsq            : sum of squares of the flat gradient (for clip_grad_norm_, exp_ns.py:215-216)
//   * pa2d_adamw_step       : one multi-tensor AdamW update (decoupled weight decay, bias correction) of
//                             the whole flat parameter buffer in ONE launch; lr and beta1 are the values
//                             OneCycleLR computed for this step; the clip coefficient
//                             min(1, max_norm / (||g|| + 1e-6)) is applied on the fly
//   * pa2d_rel_l2_fwd / bwd : sum_b ||pred_b - y_b||_2 / ||y_b||_2 and its gradient
// All HBM-bound streaming kernels (16-byte accesses, wave-shuffle + LDS block reductions, second
// deterministic pass instead of float atomics).
#include "pa2d_internal.h"

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st);

__device__ __forceinline__ float block_sum_256(float v, float* red /* [4] */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n,
                                                            float* __restrict__ partial) {
    __shared__ float red[4];
    const long long n4 = n >> 2;
    float s = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    const float t = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

struct AdamParams {
    float* p; const float* g; float* m; float* v;
    long long n;
    float lr, beta1, beta2, eps, step, decay, bias2_sqrt, max_norm, omb1, omb2;   // omb = 1 - beta, rounded from double
    const float* gnorm_sq;   // device scalar (sum of squares of g) or null
};

// torch.optim.AdamW (foreach path) arithmetic: p *= 1 - lr*wd ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2) g^2 ;
// p -= (lr/bias1) * m / (sqrt(v)/sqrt(bias2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(const AdamParams a) {
    float coef = 1.f;
    if (a.gnorm_sq && a.max_norm > 0.f) {
        const float c = a.max_norm / (sqrtf(*a.gnorm_sq) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const float step = a.step, decay = a.decay, omb1 = a.omb1, omb2 = a.omb2;
    const long long n4 = a.n >> 2;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
#define ADAM1(P, G, M, V)                                        \
    {                                                            \
        const float g_ = (G) * coef;                             \
        (P) *= decay;                                            \
        (M) += omb1 * (g_ - (M));                                \
        (V) = a.beta2 * (V) + omb2 * g_ * g_;                    \
        (P) -= step * (M) / (sqrtf(V) / a.bias2_sqrt + a.eps);   \
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        ADAM1(p.x, g.x, m.x, v.x) ADAM1(p.y, g.y, m.y, v.y) ADAM1(p.z, g.z, m.z, v.z) ADAM1(p.w, g.w, m.w, v.w)
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) ADAM1(a.p[i], a.g[i], a.m[i], a.v[i])
#undef ADAM1
}

// one workgroup per sample: d = ||pred-y||, yn = ||y||, ratio = d / yn
__global__ __launch_bounds__(256) void rel_l2_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                         long long L, float* __restrict__ dnorm,
                                                         float* __restrict__ ynorm, float* __restrict__ ratio) {
    __shared__ float red[4];
    const float* pb = pred + (size_t)blockIdx.x * L;
    const float* yb = y + (size_t)blockIdx.x * L;
    float sd = 0.f, sy = 0.f;
    for (long long i = threadIdx.x; i < L; i += 256) {
        const float yy = yb[i], d = pb[i] - yy;
        sd += d * d;
        sy += yy * yy;
    }
    const float td = block_sum_256(sd, red);
    const float ty = block_sum_256(sy, red);
    if (threadIdx.x == 0) {
        const float d = sqrtf(td), n = sqrtf(ty);
        dnorm[blockIdx.x] = d;
        ynorm[blockIdx.x] = n;
        ratio[blockIdx.x] = d / n;
    }
}

// dpred[b][i] = gout[b] * (pred - y) / (dnorm_b * ynorm_b)   (gout: per-sample upstream gradient, e.g. 1 for sum,
// 1/B for mean, arbitrary weights with reduction=False).  dnorm_b == 0 (pred == y): the sub-gradient 0, as the
// backward of torch.norm returns at 0 — never inf/NaN into the flat gradient bucket.
__global__ __launch_bounds__(256) void rel_l2_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                         const float* __restrict__ dnorm,
                                                         const float* __restrict__ ynorm,
                                                         const float* __restrict__ gout, long long L,
                                                         float* __restrict__ dpred) {
    const float dn = dnorm[blockIdx.y];
    const float s = dn > 0.f ? gout[blockIdx.y] / (dn * ynorm[blockIdx.y]) : 0.f;
    const size_t base = (size_t)blockIdx.y * L;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long long)gridDim.x * 256)
        dpred[base + i] = s * (pred[base + i] - y[base + i]);
}

static int stream_blocks(long long n) {
    long long b = ceil_div_ll(n, 1024);
    if (b > 2048) b = 2048;
    return b < 1 ? 1 : (int)b;
}

extern "C" {

size_t pa2d_sumsq_workspace(long long n) { return sizeof(float) * (size_t)stream_blocks(n); }

// out[0] = sum_i g[i]^2
int pa2d_sumsq(const float* g, long long n, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0) return pa2d_zero(out, sizeof(float), st);
    if (ws_bytes < pa2d_sumsq_workspace(n) || (((uintptr_t)g) & 15)) return PA2D_ERR_ARG;
    const int nb = stream_blocks(n);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, (float*)ws);
    PA2D_CHECK_LAUNCH();
    return pa2d_launch_reduce((const float*)ws, nb, 1, out, st);
}

// One AdamW step over flat buffers p/g/m/v of n floats (16-byte aligned).  step_index >= 1.
// Hyper-parameters travel as double and the bias corrections are evaluated in double on the host, like
// torch.optim.AdamW's Python scalars (a float powf is off by ~1e-5 relative at step 1).
int pa2d_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                    double eps, double weight_decay, int step_index, const float* gnorm_sq, float max_norm,
                    hipStream_t st) {
    if (step_index < 1 || ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15)) return PA2D_ERR_ARG;
    if (n <= 0) return PA2D_OK;
    AdamParams a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.eps = (float)eps; a.max_norm = max_norm; a.gnorm_sq = gnorm_sq;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);      // 1 - float(beta2) would be off by 1e-5 relative
    const double bias1 = 1.0 - pow(beta1, (double)step_index);
    a.step = (float)(lr / bias1);
    a.decay = (float)(1.0 - lr * weight_decay);
    a.bias2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step_index));
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, a);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// dnorm, ynorm, ratio: [B]; the caller sums `ratio` (B values) for the batch-summed loss
int pa2d_rel_l2_fwd(const float* pred, const float* y, float* dnorm, float* ynorm, float* ratio, int B, long long L,
                    hipStream_t st) {
    if (B <= 0) return PA2D_OK;
    hipLaunchKernelGGL(rel_l2_fwd_kernel, dim3(B), dim3(256), 0, st, pred, y, L, dnorm, ynorm, ratio);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int pa2d_rel_l2_bwd(const float* pred, const float* y, const float* dnorm, const float* ynorm, const float* gout,
                    float* dpred, int B, long long L, hipStream_t st) {
    if (B <= 0 || L <= 0) return PA2D_OK;
    int bx = (int)ceil_div_ll(L, 1024);
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(rel_l2_bwd_kernel, dim3(bx, B), dim3(256), 0, st, pred, y, dnorm, ynorm, gout, L, dpred);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

}  // extern "C"
