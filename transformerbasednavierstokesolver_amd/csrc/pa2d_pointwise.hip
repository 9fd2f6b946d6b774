// Row-wise (HBM-bound) kernels of the Transolver block for gfx950: LayerNorm forward/backward
// (model/Transolver_Structured_Mesh_2D.py:58-65,70-74) and the narrow output head
// mlp2: Linear(C, out_dim<=8) (…_2D.py:66,73).  One wave per row, 16-byte loads, statistics in
// registers, wave-shuffle reductions; parameter gradients are reduced per workgroup in LDS and
// across workgroups by a second deterministic pass (no float atomics).
#include "pa2d_internal.h"

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st);

#define LN_MAXV 4   // float4 per lane -> C <= 1024

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const T* xrow = x + (size_t)row * C;
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int i = lane + 64 * k;
            v[k] = i < nvec ? Act<T>::ld4(xrow + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += v[k].x + v[k].y + v[k].z + v[k].w;
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            if (lane + 64 * k < nvec) {
                const float a = v[k].x - mu, bb = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
                q += a * a + bb * bb + c * c + d * d;
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / C + eps);
        T* yrow = y + (size_t)row * C;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int i = lane + 64 * k;
            if (i < nvec) {
                const float4 gg = g4[i], bb = b4[i];
                Act<T>::st4(yrow + 4 * i, make_float4((v[k].x - mu) * rs * gg.x + bb.x, (v[k].y - mu) * rs * gg.y + bb.y,
                                                      (v[k].z - mu) * rs * gg.z + bb.z, (v[k].w - mu) * rs * gg.w + bb.w));
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// LayerNorm whose output leaves ONLY as the bf16 plane image [row][C/32][NT][32] that the conv GEMMs of the bf16 engines
// stage (NT = 3: exact hi+mid+lo split, NT = 1: rounded): no fp32 y, no split pre-pass in front of the conv
template <int NT>
__global__ __launch_bounds__(256) void layernorm_fwd_planes_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                   const float* __restrict__ b,
                                                                   unsigned short* __restrict__ planes,
                                                                   float* __restrict__ mean, float* __restrict__ rstd,
                                                                   int rows, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xrow = x + (size_t)row * C;
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int i = lane + 64 * k;
            v[k] = i < nvec ? *reinterpret_cast<const float4*>(xrow + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += v[k].x + v[k].y + v[k].z + v[k].w;
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            if (lane + 64 * k < nvec) {
                const float a = v[k].x - mu, bb = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
                q += a * a + bb * bb + c * c + d * d;
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / C + eps);
        unsigned short* prow = planes + (size_t)row * C * NT;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int i = lane + 64 * k;
            if (i < nvec) {
                const float4 gg = g4[i], bb = b4[i];
                float r[4] = {(v[k].x - mu) * rs * gg.x + bb.x, (v[k].y - mu) * rs * gg.y + bb.y,
                              (v[k].z - mu) * rs * gg.z + bb.z, (v[k].w - mu) * rs * gg.w + bb.w};
                const int c0 = 4 * i;                               // channels c0 .. c0+3: chunk c0/32, offset c0%32
                unsigned short* dst = prow + (size_t)(c0 >> 5) * NT * 32 + (c0 & 31);
#pragma unroll
                for (int pl = 0; pl < NT; ++pl) {
                    unsigned short h[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] = f32_to_bf16_bits(r[e]);
                        r[e] -= bf16_bits_to_f32(h[e]);
                    }
                    *reinterpret_cast<uint2*>(dst + pl * 32) =
                        make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                }
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)) (+ dres);  partial[blk] = [dg | db]
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ g,
                                                            const T* __restrict__ dres, T* __restrict__ dx,
                                                            float* __restrict__ partial, int rows, int C,
                                                            int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4 gv[NV], dga[NV], dba[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        gv[k] = i < nvec ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        dga[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        dba[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    // one wave per row, software-pipelined by one row: the three operand loads of row + 4 are in flight while the wave
    // reduces and stores row (the kernel is latency-bound otherwise: two dependent load rounds per row)
    float4 xn_[NV], dn_[NV], rn_[NV];
    float mun = 0.f, rsn = 0.f;
#define LNB_LOAD(row_)                                                                            \
    {                                                                                             \
        const T* xrow_ = x + (size_t)(row_) * C;                                                  \
        const T* drow_ = dy + (size_t)(row_) * C;                                                 \
        _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                     \
            const int i = lane + 64 * k;                                                          \
            if (i < nvec) {                                                                       \
                xn_[k] = Act<T>::ld4(xrow_ + 4 * i);                                              \
                dn_[k] = Act<T>::ld4(drow_ + 4 * i);                                              \
                rn_[k] = dres ? Act<T>::ld4(dres + (size_t)(row_) * C + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f); \
            }                                                                                     \
        }                                                                                         \
        mun = mean[row_]; rsn = rstd[row_];                                                       \
    }
    if (r0 + wave < r1) LNB_LOAD(r0 + wave)
    for (int row = r0 + wave; row < r1; row += 4) {
        float4 xc[NV], dc[NV], rc_[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) { xc[k] = xn_[k]; dc[k] = dn_[k]; rc_[k] = rn_[k]; }
        const float mu = mun, rs = rsn;
        if (row + 4 < r1) LNB_LOAD(row + 4)
        float4 xh[NV], gd[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nvec) {
                const float4 xv = xc[k], dv = dc[k];
                xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gd[k] = make_float4(gv[k].x * dv.x, gv[k].y * dv.y, gv[k].z * dv.z, gv[k].w * dv.w);
                s1 += gd[k].x + gd[k].y + gd[k].z + gd[k].w;
                s2 += gd[k].x * xh[k].x + gd[k].y * xh[k].y + gd[k].z * xh[k].z + gd[k].w * xh[k].w;
                dga[k].x += dv.x * xh[k].x; dga[k].y += dv.y * xh[k].y; dga[k].z += dv.z * xh[k].z; dga[k].w += dv.w * xh[k].w;
                dba[k].x += dv.x; dba[k].y += dv.y; dba[k].z += dv.z; dba[k].w += dv.w;
            } else {
                xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                gd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float c1 = wave_sum(s1) / C, c2 = wave_sum(s2) / C;
        T* orow = dx + (size_t)row * C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nvec) {
                float4 o = make_float4(rs * (gd[k].x - c1 - xh[k].x * c2), rs * (gd[k].y - c1 - xh[k].y * c2),
                                       rs * (gd[k].z - c1 - xh[k].z * c2), rs * (gd[k].w - c1 - xh[k].w * c2));
                o.x += rc_[k].x; o.y += rc_[k].y; o.z += rc_[k].z; o.w += rc_[k].w;
                Act<T>::st4(orow + 4 * i, o);
            }
        }
    }
#undef LNB_LOAD
    float4* s4 = reinterpret_cast<float4*>(smem) + (size_t)wave * 2 * nvec;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nvec) {
            s4[i] = dga[k];
            s4[nvec + i] = dba[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        partial[(size_t)blockIdx.x * 2 * C + i] = smem[i] + smem[2 * C + i] + smem[4 * C + i] + smem[6 * C + i];
}

// ---------------------------------------------------------------------------------------------
// narrow head: y[n][o] = xn[n] . W[o] + b[o],  O <= 8
template <int O, typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ xn, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, int rows,
                                                       int C, int out_dim) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const T* xrow = xn + (size_t)row * C;
        float acc[O];
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            const float4 xv = Act<T>::ld4(xrow + 4 * i);
#pragma unroll
            for (int o = 0; o < O; ++o)
                if (o < out_dim) {
                    const float4 wv = reinterpret_cast<const float4*>(w + (size_t)o * C)[i];
                    acc[o] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                }
        }
#pragma unroll
        for (int o = 0; o < O; ++o) {
            const float s = wave_sum(acc[o]);
            if (lane == 0 && o < out_dim) y[(size_t)row * out_dim + o] = s + b[o];
        }
    }
}

// dxn[n][c] = sum_o dy[n][o] W[o][c];  partial[blk] = [dW (out_dim*C) | db (out_dim)]
template <int O, typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dy, const T* __restrict__ xn,
                                                       const float* __restrict__ w, T* __restrict__ dxn,
                                                       float* __restrict__ partial, int rows, int C, int out_dim,
                                                       int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][out_dim*C + out_dim]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    const int rec = out_dim * C + out_dim;
    float* mine = smem + (size_t)wave * rec;
    for (int i = lane; i < rec; i += 64) mine[i] = 0.f;
    float dbacc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) dbacc[o] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += 4) {
        float g[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            g[o] = o < out_dim ? dy[(size_t)row * out_dim + o] : 0.f;
            dbacc[o] += g[o];
        }
        const T* xrow = xn + (size_t)row * C;
        T* orow = dxn + (size_t)row * C;
        for (int i = lane; i < nvec; i += 64) {
            const float4 xv = Act<T>::ld4(xrow + 4 * i);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < O; ++o)
                if (o < out_dim) {
                    const float4 wv = reinterpret_cast<const float4*>(w + (size_t)o * C)[i];
                    acc.x += g[o] * wv.x; acc.y += g[o] * wv.y; acc.z += g[o] * wv.z; acc.w += g[o] * wv.w;
                    float4* m4 = reinterpret_cast<float4*>(mine + (size_t)o * C) + i;   // lane-private slot
                    float4 mv = *m4;
                    mv.x += g[o] * xv.x; mv.y += g[o] * xv.y; mv.z += g[o] * xv.z; mv.w += g[o] * xv.w;
                    *m4 = mv;
                }
            Act<T>::st4(orow + 4 * i, acc);
        }
    }
    if (lane == 0)
        for (int o = 0; o < out_dim; ++o) mine[out_dim * C + o] = dbacc[o];
    __syncthreads();
    for (int i = threadIdx.x; i < rec; i += 256)
        partial[(size_t)blockIdx.x * rec + i] = smem[i] + smem[rec + i] + smem[2 * rec + i] + smem[3 * rec + i];
}

// out = dy * act'(pre)   (elementwise; used only by the generic Linear+act layers off the hot path)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ out,
                               long long n, int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = dy[i] * act_bwd(act, pre[i]);
}

static int row_blocks(int rows) {
    int b = ceil_div(rows, 64);
    if (b > 1024) b = 1024;
    return b < 1 ? 1 : b;
}

template <typename T>
static int ln_fwd_t(const T* x, const float* g, const float* b, T* y, float* mean, float* rstd, int rows, int C, float eps,
                    hipStream_t st) {
    if ((C & 3) || C > 256 * LN_MAXV) return PA2D_ERR_UNSUPPORTED;
    if (rows <= 0) return PA2D_OK;
    int grid = ceil_div(rows, 4);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

template <typename T>
static int ln_bwd_t(const T* dy, const T* x, const float* mean, const float* rstd, const float* g, const T* dres, T* dx,
                    float* dg, float* db, void* ws, size_t ws_bytes, int rows, int C, int accumulate, hipStream_t st) {
    if ((C & 3) || C > 256 * LN_MAXV) return PA2D_ERR_UNSUPPORTED;
    if (rows <= 0) {
        if (accumulate) return PA2D_OK;
        const int rz = pa2d_zero(dg, sizeof(float) * C, st);
        return rz ? rz : pa2d_zero(db, sizeof(float) * C, st);
    }
    if (ws_bytes < sizeof(float) * ((size_t)row_blocks(rows) + 1) * 2 * C) return PA2D_ERR_WORKSPACE;
    const int nb = row_blocks(rows);
    const int rpb = ceil_div(rows, nb);
    float* part = (float*)ws;
    // NV = float4 per lane (C <= 256 NV): the common C <= 256 gets the small-register variant
    auto kern = C <= 256 ? &layernorm_bwd_kernel<T, 1> : (C <= 512 ? &layernorm_bwd_kernel<T, 2> : &layernorm_bwd_kernel<T, LN_MAXV>);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), sizeof(float) * 8 * C, st, dy, x, mean, rstd, g,
                       dres, dx, part, rows, C, rpb);
    PA2D_CHECK_LAUNCH();
    ReduceSegs segs;
    segs.nseg = 2;
    segs.begin[0] = 0; segs.begin[1] = C; segs.begin[2] = 2 * C; segs.begin[3] = segs.begin[4] = 2 * C;
    segs.dst[0] = dg; segs.dst[1] = db; segs.dst[2] = segs.dst[3] = nullptr;
    return pa2d_launch_reduce_segs(part, nb, 2 * C, segs, accumulate, st);
}

template <typename T>
static int head_fwd_t(const T* xn, const float* w, const float* b, float* y, int rows, int C, int out_dim, hipStream_t st) {
    if ((C & 3) || out_dim < 1 || out_dim > 8) return PA2D_ERR_UNSUPPORTED;
    if (rows <= 0) return PA2D_OK;
    int grid = ceil_div(rows, 4);
    if (grid > 8192) grid = 8192;
    if (out_dim == 1) hipLaunchKernelGGL((head_fwd_kernel<1, T>), dim3(grid), dim3(256), 0, st, xn, w, b, y, rows, C, out_dim);
    else if (out_dim == 2) hipLaunchKernelGGL((head_fwd_kernel<2, T>), dim3(grid), dim3(256), 0, st, xn, w, b, y, rows, C, out_dim);
    else if (out_dim <= 4) hipLaunchKernelGGL((head_fwd_kernel<4, T>), dim3(grid), dim3(256), 0, st, xn, w, b, y, rows, C, out_dim);
    else hipLaunchKernelGGL((head_fwd_kernel<8, T>), dim3(grid), dim3(256), 0, st, xn, w, b, y, rows, C, out_dim);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

template <typename T>
static int head_bwd_t(const float* dy, const T* xn, const float* w, T* dxn, float* dw, float* db, void* ws, size_t ws_bytes,
                      int rows, int C, int out_dim, int accumulate, hipStream_t st) {
    if ((C & 3) || out_dim < 1 || out_dim > 8) return PA2D_ERR_UNSUPPORTED;
    if (rows <= 0) {
        if (accumulate) return PA2D_OK;
        const int rz = pa2d_zero(dw, sizeof(float) * out_dim * C, st);
        return rz ? rz : pa2d_zero(db, sizeof(float) * out_dim, st);
    }
    const int rec = out_dim * C + out_dim;
    if (ws_bytes < sizeof(float) * ((size_t)row_blocks(rows) + 1) * rec) return PA2D_ERR_WORKSPACE;
    const size_t smem = sizeof(float) * 4 * rec;
    if (smem > 64 * 1024) return PA2D_ERR_UNSUPPORTED;
    const int nb = row_blocks(rows);
    const int rpb = ceil_div(rows, nb);
    float* part = (float*)ws;
    if (out_dim == 1) hipLaunchKernelGGL((head_bwd_kernel<1, T>), dim3(nb), dim3(256), smem, st, dy, xn, w, dxn, part, rows, C, out_dim, rpb);
    else if (out_dim == 2) hipLaunchKernelGGL((head_bwd_kernel<2, T>), dim3(nb), dim3(256), smem, st, dy, xn, w, dxn, part, rows, C, out_dim, rpb);
    else if (out_dim <= 4) hipLaunchKernelGGL((head_bwd_kernel<4, T>), dim3(nb), dim3(256), smem, st, dy, xn, w, dxn, part, rows, C, out_dim, rpb);
    else hipLaunchKernelGGL((head_bwd_kernel<8, T>), dim3(nb), dim3(256), smem, st, dy, xn, w, dxn, part, rows, C, out_dim, rpb);
    PA2D_CHECK_LAUNCH();
    ReduceSegs segs;
    segs.nseg = 2;
    segs.begin[0] = 0; segs.begin[1] = (long long)out_dim * C; segs.begin[2] = rec; segs.begin[3] = segs.begin[4] = rec;
    segs.dst[0] = dw; segs.dst[1] = db; segs.dst[2] = segs.dst[3] = nullptr;
    return pa2d_launch_reduce_segs(part, nb, rec, segs, accumulate, st);
}

extern "C" {

int pa2d_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int rows,
                       int C, float eps, hipStream_t st) {
    return ln_fwd_t<float>(x, g, b, y, mean, rstd, rows, C, eps, st);
}
int pa2d_layernorm_fwd_bf16(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int rows,
                            int C, float eps, hipStream_t st) {
    return ln_fwd_t<bf16_t>((const bf16_t*)x, g, b, (bf16_t*)y, mean, rstd, rows, C, eps, st);
}

// planes: pa2d_planes_bytes(rows, C, engine) bytes; engine PA2D_ENGINE_SPLIT (3 planes) or PA2D_ENGINE_BF16 (1 plane)
int pa2d_layernorm_fwd_planes(const float* x, const float* g, const float* b, void* planes, float* mean, float* rstd,
                              int rows, int C, float eps, int engine, hipStream_t st) {
    if ((C & 31) || C > 256 * LN_MAXV) return PA2D_ERR_UNSUPPORTED;
    if (engine != 1 && engine != 2) return PA2D_ERR_ARG;
    if (rows <= 0) return PA2D_OK;
    int grid = ceil_div(rows, 4);
    if (grid > 8192) grid = 8192;
    if (engine == 1)
        hipLaunchKernelGGL((layernorm_fwd_planes_kernel<3>), dim3(grid), dim3(256), 0, st, x, g, b, (unsigned short*)planes,
                           mean, rstd, rows, C, eps);
    else
        hipLaunchKernelGGL((layernorm_fwd_planes_kernel<1>), dim3(grid), dim3(256), 0, st, x, g, b, (unsigned short*)planes,
                           mean, rstd, rows, C, eps);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

size_t pa2d_layernorm_bwd_workspace(int rows, int C) { return sizeof(float) * ((size_t)row_blocks(rows) + 1) * 2 * C; }

// dres (optional): gradient flowing through the residual branch, added to dx (dx may alias dres)
int pa2d_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* g,
                       const float* dres, float* dx, float* dg, float* db, void* ws, size_t ws_bytes, int rows, int C,
                       int accumulate, hipStream_t st) {
    return ln_bwd_t<float>(dy, x, mean, rstd, g, dres, dx, dg, db, ws, ws_bytes, rows, C, accumulate, st);
}
int pa2d_layernorm_bwd_bf16(const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                            const void* dres, void* dx, float* dg, float* db, void* ws, size_t ws_bytes, int rows, int C,
                            int accumulate, hipStream_t st) {
    return ln_bwd_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)x, mean, rstd, g, (const bf16_t*)dres, (bf16_t*)dx, dg, db, ws,
                            ws_bytes, rows, C, accumulate, st);
}

int pa2d_head_fwd(const float* xn, const float* w, const float* b, float* y, int rows, int C, int out_dim,
                  hipStream_t st) {
    return head_fwd_t<float>(xn, w, b, y, rows, C, out_dim, st);
}
int pa2d_head_fwd_bf16(const void* xn, const float* w, const float* b, float* y, int rows, int C, int out_dim,
                       hipStream_t st) {
    return head_fwd_t<bf16_t>((const bf16_t*)xn, w, b, y, rows, C, out_dim, st);
}

size_t pa2d_head_bwd_workspace(int rows, int C, int out_dim) {
    return sizeof(float) * ((size_t)row_blocks(rows) + 1) * ((size_t)out_dim * C + out_dim);
}

int pa2d_head_bwd(const float* dy, const float* xn, const float* w, float* dxn, float* dw, float* db, void* ws,
                  size_t ws_bytes, int rows, int C, int out_dim, int accumulate, hipStream_t st) {
    return head_bwd_t<float>(dy, xn, w, dxn, dw, db, ws, ws_bytes, rows, C, out_dim, accumulate, st);
}
int pa2d_head_bwd_bf16(const float* dy, const void* xn, const float* w, void* dxn, float* dw, float* db, void* ws,
                       size_t ws_bytes, int rows, int C, int out_dim, int accumulate, hipStream_t st) {
    return head_bwd_t<bf16_t>(dy, (const bf16_t*)xn, w, (bf16_t*)dxn, dw, db, ws, ws_bytes, rows, C, out_dim, accumulate, st);
}

int pa2d_act_bwd(const float* dy, const float* pre, float* out, long long n, int act, hipStream_t st) {
    if (n <= 0) return PA2D_OK;
    long long blocks = ceil_div_ll(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, pre, out, n, act);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

const char* pa2d_version(void) { return "pa2d 0.3 gfx950"; }

}  // extern "C"
