// Slice / de-slice kernels of Physics-Attention (Physics_Attention.py:98-102,116) and their backward
//  for gfx950.  The N x M slice-weight matrix W is never written to HBM: it
// is recomputed from x_mid in every kernel that needs it and lives only in MFMA accumulators.
//
// All contractions run on v_mfma_f32_16x16x4_f32 (exact fp32).  Fragment maps: A operand lane l
// holds A[i=l&15][k=l>>4], B operand B[k=l>>4][j=l&15], C/D element r is (row 4*(l>>4)+r, col l&15).
// Two register layouts of a 16-point tile of W are used:
//   N-layout (rows = points, cols = slices)   L  = X . Ws^T     -> softmax across 16 lanes (row of a
//       16-lane group), accumulator register r of lane (m, kq) is W[point 4kq+r][m], i.e. exactly
//       the A operand (i=m, k=point) of the scatter  S += W^T . F  (MFMA step r).
//   T-layout (rows = slices, cols = points)   L^T = Ws . X^T    -> softmax in-lane (+2 shuffles),
//       register (mt,r) of lane (pt, kq) is W[pt][16mt+4kq+r], i.e. the B operand (k=slice, j=pt)
//       of  Y^T = O^T . W^T  (de-slice) and of the backward products dF^T, dX^T.
// The k-order inside a contraction over d is permuted (lane (.,kq) loads VEC consecutive d and
// feeds MFMA step t with element t): legal because A and B use the same permutation.
// X / F / dY fragments are loaded straight from global memory (each element exactly once, 64-B
// contiguous per point per instruction); no LDS in the forward kernels.
#include "pa2d_internal.h"
#include <stdlib.h>

#define NEG_BIG (-1e30f)
#define PA2D_ENGINE_BF16_ID 2

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st);

template <int D>
struct SCfg {
    static constexpr int VEC = D >= 16 ? 4 : 2;
    static constexpr int NV = D / (4 * VEC);
    static constexpr int KS = D / 4;          // MFMA steps of a contraction over d
    static constexpr int DT = (D + 15) / 16;  // 16-wide tiles over d
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Reductions across the 16 lanes of a DPP row (lanes 16g..16g+15) with VALU-DPP operands instead
// of ds_bpermute: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.  After the
// four steps every lane of the row holds the row result.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// exp(x) for x <= 0 (softmax after max subtraction) on the hardware exponential: 2^(x*log2 e).
// Relative error ~1e-7 near 0 where the weights matter, growing only for terms that are ~0 anyway.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float kq_max(float v) {   // across the 4 lane groups l, l^16, l^32, l^48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float kq_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float clamp_tau(float t) { return fminf(fmaxf(t, 0.1f), 5.0f); }

// fragment of a [16 rows][D] panel for a contraction over d: lane (row = l&15, kq = l>>4) gets
// elements k(v,t) = 4*VEC*v + VEC*kq + t of its row.
template <int D>
__device__ __forceinline__ void load_kfrag(const float* row, bool valid, int kq, float (&f)[SCfg<D>::KS]) {
    constexpr int VEC = SCfg<D>::VEC, NV = SCfg<D>::NV;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if constexpr (VEC == 4) {
            const float4 q = valid ? *reinterpret_cast<const float4*>(row + 16 * v + 4 * kq) : make_float4(0, 0, 0, 0);
            f[4 * v + 0] = q.x; f[4 * v + 1] = q.y; f[4 * v + 2] = q.z; f[4 * v + 3] = q.w;
        } else {
            const float2 q = valid ? *reinterpret_cast<const float2*>(row + 8 * v + 2 * kq) : make_float2(0, 0);
            f[2 * v + 0] = q.x; f[2 * v + 1] = q.y;
        }
    }
}

// same fragment through a buffer descriptor: masked lanes (off == OOB_OFF) read 0, no branch
// (T = storage type of the activations: float or bf16; offsets are BYTE offsets, ES = sizeof(T))
template <int D, typename T>
__device__ __forceinline__ void buf_kfrag(__amdgpu_buffer_rsrc_t r, unsigned off, int kq, float (&f)[SCfg<D>::KS]) {
    constexpr int VEC = SCfg<D>::VEC, NV = SCfg<D>::NV;
    constexpr unsigned ES = Act<T>::ES;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if constexpr (VEC == 4) {
            const float4 q = Act<T>::bld4(r, off == OOB_OFF ? OOB_OFF : off + (16 * v + 4 * kq) * ES);
            f[4 * v + 0] = q.x; f[4 * v + 1] = q.y; f[4 * v + 2] = q.z; f[4 * v + 3] = q.w;
        } else {
            const float2 q = Act<T>::bld2(r, off == OOB_OFF ? OOB_OFF : off + (8 * v + 2 * kq) * ES);
            f[2 * v + 0] = q.x; f[2 * v + 1] = q.y;
        }
    }
}

struct SliceParams {
    const void* xm; long long ldx;      // x_mid rows: xm[(b*N+n)*ldx + h*D + d]   (float or bf16 storage)
    const void* v; long long ldv;       // values scattered (fx_mid forward, dY in backward phase A)
    const float* ws; const float* bs; const float* temperature;   // [M,D], [M], [heads]
    float* spart; float* npart;         // [B,heads,nchunk,M,D], [B,heads,nchunk,M] (npart may be null)
    int B, N, heads, M, nchunk, ppc;    // ppc = points per chunk (multiple of 16)
    unsigned x_bytes, v_bytes;          // extents for the buffer descriptors
    int clamp;                          // 1: clamp(temperature, .1, 5) (structured mesh); 0: raw (irregular mesh)
    int xcd_map;                        // workgroup numbering, see slice_decode
};

// S_partial[m][d] = sum_{n in chunk} W[n][m] * V[n][d];  n_partial[m] = sum_n W[n][m]
template <int D, int MT, typename T>
__global__ __launch_bounds__(256) void slice_scatter_kernel(const SliceParams p) {
    constexpr int KS = SCfg<D>::KS, DT = SCfg<D>::DT;
    constexpr unsigned ES = Act<T>::ES;
    constexpr int MP = 16 * MT, DP = 16 * DT;
    __shared__ float sbuf[MP * DP];
    __shared__ float nbuf[MP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    int b, hh, chunk, bid;
    if (!slice_decode(p.xcd_map, p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float inv_tau = 1.0f / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);

    float wsf[MT][KS], bsv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + li;
        load_kfrag<D>(p.ws + (size_t)(m < p.M ? m : 0) * D, m < p.M, kq, wsf[mt]);
        bsv[mt] = m < p.M ? p.bs[m] : 0.f;
    }
    f32x4 sacc[MT][DT];
    float nacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        nacc[mt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) sacc[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int p_begin = chunk * p.ppc;
    const int p_end = min(p.N, p_begin + p.ppc);
    const unsigned row0 = (unsigned)b * (unsigned)p.N;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_v(p.xm, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc_v(p.v, p.v_bytes);
    const unsigned ldx4 = (unsigned)p.ldx * ES, ldv4 = (unsigned)p.ldv * ES, hcol = (unsigned)(hh * D) * ES;
    // group loads: X as k-fragments (lane = point li), V as rows (lane = channel li) — branch-free
    // buffer loads, issued one group ahead so that they fly under the 64 MFMAs of the current group
#define SC_LOAD(g_, XF, FV)                                                                            \
    {                                                                                                  \
        const int pt_ = (g_) + li;                                                                     \
        buf_kfrag<D, T>(rx, pt_ < p_end ? (row0 + pt_) * ldx4 + hcol : OOB_OFF, kq, XF);                \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
            const int pr_ = (g_) + 4 * kq + r;                                                         \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                        \
                const int d_ = 16 * dt + li;                                                           \
                FV[r][dt] = Act<T>::bld1(rv, (pr_ < p_end && d_ < D) ? (row0 + pr_) * ldv4 + hcol + d_ * ES : OOB_OFF); \
            }                                                                                          \
        }                                                                                              \
    }
    float xf[KS], fv[4][DT], xf_n[KS], fv_n[4][DT];
    int g = p_begin + wave * 16;
    if (g < p_end) SC_LOAD(g, xf, fv)
    for (; g < p_end; g += 64) {
        if (g + 64 < p_end) SC_LOAD(g + 64, xf_n, fv_n)
        f32x4 w[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            w[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[mt] = mfma16(xf[ks], wsf[mt][ks], w[mt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = NEG_BIG;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float z = (w[mt][r] + bsv[mt]) * inv_tau;
                if (16 * mt + li >= p.M) z = NEG_BIG;
                w[mt][r] = z;
                mx = fmaxf(mx, z);
            }
            mx = row16_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float e = fast_exp(w[mt][r] - mx);
                w[mt][r] = e;
                sm += e;
            }
            sm = row16_sum(sm);
            const float inv = (g + 4 * kq + r) < p_end ? 1.0f / sm : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                w[mt][r] *= inv;
                nacc[mt] += w[mt][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sacc[mt][dt] = mfma16(w[mt][r], fv[r][dt], sacc[mt][dt]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = xf_n[ks];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) fv[r][dt] = fv_n[r][dt];
    }
#undef SC_LOAD
    // deterministic cross-wave reduction: waves add in order 0,1,2,3
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) nacc[mt] = kq_sum(nacc[mt]);
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (16 * mt + 4 * kq + r) * DP + 16 * dt + li;
                        sbuf[idx] = (wv == 0 ? 0.f : sbuf[idx]) + sacc[mt][dt][r];
                    }
                if (kq == 0) nbuf[16 * mt + li] = (wv == 0 ? 0.f : nbuf[16 * mt + li]) + nacc[mt];
            }
        }
        __syncthreads();
    }
    float* so = p.spart + (size_t)bid * p.M * D;
    for (int i = tid; i < p.M * D; i += 256) so[i] = sbuf[(i / D) * DP + (i % D)];
    if (p.npart)
        for (int i = tid; i < p.M; i += 256) p.npart[(size_t)bid * p.M + i] = nbuf[i];
}

struct DesliceParams {
    const void* xm; long long ldx;
    const float* o;                     // [B,heads,M,D]
    const float* ws; const float* bs; const float* temperature;
    void* y; long long ldy;             // y[(b*N+n)*ldy + h*D + d]
    int B, N, heads, M, nchunk, ppc;
    unsigned x_bytes, y_bytes;
    int clamp, xcd_map;
};

// Y[n][h*D+d] = sum_m W[n][m] * O[m][d]
template <int D, int MT, typename T>
__global__ __launch_bounds__(256) void deslice_kernel(const DesliceParams p) {
    constexpr int KS = SCfg<D>::KS, DT = SCfg<D>::DT;
    constexpr unsigned ES = Act<T>::ES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    int b, hh, chunk, bid;
    if (!slice_decode(p.xcd_map, p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float inv_tau = 1.0f / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);

    float wsf[MT][KS], bst[MT][4], of[MT][4][DT];
    const float* ob = p.o + (size_t)(b * p.heads + hh) * p.M * D;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + li;
        load_kfrag<D>(p.ws + (size_t)(m < p.M ? m : 0) * D, m < p.M, kq, wsf[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mr = 16 * mt + 4 * kq + r;
            bst[mt][r] = mr < p.M ? p.bs[mr] : 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = 16 * dt + li;
                of[mt][r][dt] = (mr < p.M && d < D) ? ob[(size_t)mr * D + d] : 0.f;
            }
        }
    }
    const int p_begin = chunk * p.ppc;
    const int p_end = min(p.N, p_begin + p.ppc);
    const unsigned row0 = (unsigned)b * (unsigned)p.N;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_v(p.xm, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc_v(p.y, p.y_bytes);
    const unsigned ldx4 = (unsigned)p.ldx * ES, ldy4 = (unsigned)p.ldy * ES, hcol = (unsigned)(hh * D) * ES;
    float xf[KS], xf_n[KS];
    int g = p_begin + wave * 16;
    if (g < p_end) buf_kfrag<D, T>(rx, g + li < p_end ? (row0 + g + li) * ldx4 + hcol : OOB_OFF, kq, xf);
    for (; g < p_end; g += 64) {
        const int pt = g + li;
        const bool pv = pt < p_end;
        if (g + 64 < p_end) buf_kfrag<D, T>(rx, pt + 64 < p_end ? (row0 + pt + 64) * ldx4 + hcol : OOB_OFF, kq, xf_n);
        f32x4 w[MT];
        float mx = NEG_BIG;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            w[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[mt] = mfma16(wsf[mt][ks], xf[ks], w[mt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float z = (w[mt][r] + bst[mt][r]) * inv_tau;
                if (16 * mt + 4 * kq + r >= p.M) z = NEG_BIG;
                w[mt][r] = z;
                mx = fmaxf(mx, z);
            }
        }
        mx = kq_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fast_exp(w[mt][r] - mx);
                w[mt][r] = e;
                sm += e;
            }
        sm = kq_sum(sm);
        const float inv = 1.0f / sm;
        f32x4 yacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) yacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wv = w[mt][r] * inv;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) yacc[dt] = mfma16(of[mt][r][dt], wv, yacc[dt]);
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * kq;
            Act<T>::bst4(ry, (pv && d < D) ? (row0 + pt) * ldy4 + hcol + d * ES : OOB_OFF,
                         make_float4(yacc[dt][0], yacc[dt][1], yacc[dt][2], yacc[dt][3]));
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = xf_n[ks];
    }
}

struct SliceBwdParams {
    const void* xm; long long ldx;      // x_mid
    const void* fm; long long ldf;      // fx_mid
    const void* dy; long long lddy;     // gradient w.r.t. de-sliced y
    const float* ws; const float* bs; const float* temperature;
    const float* o; const float* ds; const float* dn;   // [B,heads,M,D] x2, [B,heads,M]
    void* dxm; long long lddx;          // outputs
    void* dfm; long long lddf;
    void* planes; unsigned planes_bytes; int planes_nt;   // planes_nt > 0: [dX | dF] is written ONLY as the bf16 plane image
                                        // [row][2C/32][nt][32] the conv GEMMs stage (no fp32 dxm / dfm), and the
                                        // column sums of dX / dF (= the conv bias gradients) go to the block record
    int stride;                         // floats per block record: M*D (dWs) + M (dbs) + 1 (dtau) + 2*D (dbx | dbf)
    float* part;                        // per block: [M*D (dWs) | M (dbs) | 1 (dtau)]
    int B, N, heads, M, nchunk, ppc;
    unsigned x_bytes, f_bytes, dy_bytes, dx_bytes, df_bytes;
    int clamp, xcd_map;
};

// Backward phase C (per point): recompute W, then
//   dW = dY.O^T + F.dS^T + dn ; dL = W*(dW - rowsum(dW*W)) ; dF = W.dS ; dX = dL.Ws/tau
//   dWs += (dL/tau)^T.X ; dbs += sum dL/tau ; dtau -= sum(dL*L)/tau
template <int D, int MT, typename T, int PL = 0>
__global__ __launch_bounds__(256, (MT <= 4 && D <= 32 && sizeof(T) == 4) ? 2 : 1) void slice_bwd_kernel(const SliceBwdParams p) {
    constexpr int KS = SCfg<D>::KS, DT = SCfg<D>::DT, VEC = SCfg<D>::VEC, NV = SCfg<D>::NV;
    constexpr unsigned ES = Act<T>::ES;
    constexpr int MP = 16 * MT, DP = 16 * DT, P = DP + 4;   // LDS row pitch (floats), 16-B aligned
    constexpr int TP = MP + 4;                              // pitch of the per-wave dL transpose tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const WsL = smem;                 // [MP][P]
    float* const OL = WsL + MP * P;          // [MP][P]
    float* const dSL = OL + MP * P;          // [MP][P]
    float* const bsL = dSL + MP * P;         // [MP]
    float* const dnL = bsL + MP;             // [MP]
    float* const TL = dnL + MP;              // [4 waves][16][TP]
    float* const red = smem;                 // [MP*DP + MP + 4] cross-wave reduction, aliases the
                                             // fragment matrices (dead after the point loop)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    int b, hh, chunk, bid;
    if (!slice_decode(p.xcd_map, p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float inv_tau = 1.0f / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);
    const size_t bh = (size_t)(b * p.heads + hh);

    for (int i = tid; i < MP * P; i += 256) {
        const int m = i / P, d = i % P;
        const bool ok = m < p.M && d < D;
        WsL[i] = ok ? p.ws[(size_t)m * D + d] : 0.f;
        OL[i] = ok ? p.o[(bh * p.M + m) * D + d] : 0.f;
        dSL[i] = ok ? p.ds[(bh * p.M + m) * D + d] : 0.f;
    }
    for (int i = tid; i < MP; i += 256) {
        bsL[i] = i < p.M ? p.bs[i] : 0.f;
        dnL[i] = i < p.M ? p.dn[bh * p.M + i] : 0.f;
    }
    __syncthreads();

    f32x4 wsacc[MT][DT];
    float dbacc[MT][4];
    float dtacc = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dbacc[mt][r] = 0.f;
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) wsacc[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float bxs[DT][4], bfs[DT][4];            // PL: column sums of dX / dF over this lane's points
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bxs[dt][r] = bfs[dt][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rpl = make_rsrc_v(PL ? p.planes : nullptr, PL ? p.planes_bytes : 0u);
    const int Ctot = p.heads * D;
    float* const myT = TL + wave * 16 * TP;
    const int p_begin = chunk * p.ppc;
    const int p_end = min(p.N, p_begin + p.ppc);
    const unsigned row0 = (unsigned)b * (unsigned)p.N;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_v(p.xm, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rf = make_rsrc_v(p.fm, p.f_bytes);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc_v(p.dy, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t rdx = make_rsrc_v(p.dxm, p.dx_bytes);
    const __amdgpu_buffer_rsrc_t rdf = make_rsrc_v(p.dfm, p.df_bytes);
    const unsigned ldx4 = (unsigned)p.ldx * ES, ldf4 = (unsigned)p.ldf * ES, ldg4 = (unsigned)p.lddy * ES;
    const unsigned lddx4 = (unsigned)p.lddx * ES, lddf4 = (unsigned)p.lddf * ES, hcol = (unsigned)(hh * D) * ES;
    // one-group-ahead prefetch of the three k-fragments and of the X rows used by dWs
#define BW_LOAD(g_, XF, FF, GF, XV)                                                                    \
    {                                                                                                  \
        const int pt_ = (g_) + li;                                                                     \
        const bool ok_ = pt_ < p_end;                                                                  \
        buf_kfrag<D, T>(rx, ok_ ? (row0 + pt_) * ldx4 + hcol : OOB_OFF, kq, XF);                        \
        buf_kfrag<D, T>(rf, ok_ ? (row0 + pt_) * ldf4 + hcol : OOB_OFF, kq, FF);                        \
        buf_kfrag<D, T>(rg, ok_ ? (row0 + pt_) * ldg4 + hcol : OOB_OFF, kq, GF);                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
            const int pr_ = (g_) + 4 * kq + r;                                                         \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                        \
                const int d_ = 16 * dt + li;                                                           \
                XV[r][dt] = Act<T>::bld1(rx, (pr_ < p_end && d_ < D) ? (row0 + pr_) * ldx4 + hcol + d_ * ES : OOB_OFF); \
            }                                                                                          \
        }                                                                                              \
    }
    float xf[KS], ff[KS], gf[KS], xv[4][DT], xf_n[KS], ff_n[KS], gf_n[KS], xv_n[4][DT];
    int g = p_begin + wave * 16;
    if (g < p_end) BW_LOAD(g, xf, ff, gf, xv)
    for (; g < p_end; g += 64) {
        const int pt = g + li;
        const bool pv = pt < p_end;
        if (g + 64 < p_end) BW_LOAD(g + 64, xf_n, ff_n, gf_n, xv_n)

        // logits^T and dW^T (rows m = 16mt+4kq+r, cols pt = li)
        f32x4 w[MT], dw[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            w[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dw[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* wrow = WsL + (16 * mt + li) * P;
            const float* orow = OL + (16 * mt + li) * P;
            const float* srow = dSL + (16 * mt + li) * P;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float wa[VEC], oa[VEC], sa[VEC];
                if constexpr (VEC == 4) {
                    const float4 q0 = *reinterpret_cast<const float4*>(wrow + 16 * v + 4 * kq);
                    const float4 q1 = *reinterpret_cast<const float4*>(orow + 16 * v + 4 * kq);
                    const float4 q2 = *reinterpret_cast<const float4*>(srow + 16 * v + 4 * kq);
                    wa[0] = q0.x; wa[1] = q0.y; wa[2] = q0.z; wa[3] = q0.w;
                    oa[0] = q1.x; oa[1] = q1.y; oa[2] = q1.z; oa[3] = q1.w;
                    sa[0] = q2.x; sa[1] = q2.y; sa[2] = q2.z; sa[3] = q2.w;
                } else {
                    const float2 q0 = *reinterpret_cast<const float2*>(wrow + 8 * v + 2 * kq);
                    const float2 q1 = *reinterpret_cast<const float2*>(orow + 8 * v + 2 * kq);
                    const float2 q2 = *reinterpret_cast<const float2*>(srow + 8 * v + 2 * kq);
                    wa[0] = q0.x; wa[1] = q0.y; oa[0] = q1.x; oa[1] = q1.y; sa[0] = q2.x; sa[1] = q2.y;
                }
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    w[mt] = mfma16(wa[t], xf[VEC * v + t], w[mt]);
                    dw[mt] = mfma16(oa[t], gf[VEC * v + t], dw[mt]);
                    dw[mt] = mfma16(sa[t], ff[VEC * v + t], dw[mt]);
                }
            }
        }
        // softmax over m (in-lane + across kq groups); keep scaled logits z for dtau
        f32x4 z[MT];
        float mx = NEG_BIG;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mt + 4 * kq + r;
                float zz = (w[mt][r] + bsL[m]) * inv_tau;
                if (m >= p.M) zz = NEG_BIG;
                z[mt][r] = zz;
                mx = fmaxf(mx, zz);
            }
        mx = kq_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fast_exp(z[mt][r] - mx);
                w[mt][r] = e;
                sm += e;
            }
        sm = kq_sum(sm);
        const float inv = pv ? 1.0f / sm : 0.f;
        float rd = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mt + 4 * kq + r;
                w[mt][r] *= inv;
                dw[mt][r] += dnL[m];
                rd += dw[mt][r] * w[mt][r];
            }
        rd = kq_sum(rd);
        // dL (stored in dw), dbs / dtau accumulation
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dl = w[mt][r] * (dw[mt][r] - rd);
                dw[mt][r] = dl;
                dbacc[mt][r] += dl;
                if (16 * mt + 4 * kq + r < p.M) dtacc += dl * z[mt][r];
            }
        // dF^T = dS^T . W^T ; dX^T = Ws^T . dL^T / tau      (rows d = 16dt+4kq+r', cols pt)
        f32x4 facc[DT], xacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            facc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            xacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mrow = (16 * mt + 4 * kq + r) * P;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    facc[dt] = mfma16(dSL[mrow + 16 * dt + li], w[mt][r], facc[dt]);
                    xacc[dt] = mfma16(WsL[mrow + 16 * dt + li], dw[mt][r], xacc[dt]);
                }
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * kq;
            const bool ok = pv && d < D;
            const float4 fo = make_float4(facc[dt][0], facc[dt][1], facc[dt][2], facc[dt][3]);
            const float4 xo = make_float4(xacc[dt][0] * inv_tau, xacc[dt][1] * inv_tau, xacc[dt][2] * inv_tau,
                                          xacc[dt][3] * inv_tau);
            if constexpr (PL == 0) {
                Act<T>::bst4(rdf, ok ? (row0 + pt) * lddf4 + hcol + d * ES : OOB_OFF, fo);
                Act<T>::bst4(rdx, ok ? (row0 + pt) * lddx4 + hcol + d * ES : OOB_OFF, xo);
            } else {
                // plane image of the [rows, 2C] tensor [dX | dF]: 4 consecutive channels = 8 bytes in each plane
                const int cx = hh * D + d, cf = Ctot + cx;
                const unsigned rowb = (row0 + pt) * (unsigned)((2 * Ctot / 32) * PL * 64);
                const unsigned ox = ok ? rowb + (unsigned)(((cx >> 5) * PL) * 64 + (cx & 31) * 2) : OOB_OFF;
                const unsigned of_ = ok ? rowb + (unsigned)(((cf >> 5) * PL) * 64 + (cf & 31) * 2) : OOB_OFF;
                const float xv[4] = {xo.x, xo.y, xo.z, xo.w}, fv[4] = {fo.x, fo.y, fo.z, fo.w};
                typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
                typedef float f32x4_ __attribute__((ext_vector_type(4)));
                // exact split x = hi + mid + lo with packed conversions (v_cvt_pk_bf16_f32: two values per instruction)
                f32x4_ xr = {xo.x, xo.y, xo.z, xo.w}, fr = {fo.x, fo.y, fo.z, fo.w};
#pragma unroll
                for (int q = 0; q < PL; ++q) {
                    const bf16x4_ hx = __builtin_convertvector(xr, bf16x4_), hf = __builtin_convertvector(fr, bf16x4_);
                    if (q + 1 < PL) {
                        xr -= __builtin_convertvector(hx, f32x4_);
                        fr -= __builtin_convertvector(hf, f32x4_);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, hx), rpl, ox == OOB_OFF ? OOB_OFF : ox + q * 64u, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, hf), rpl, of_ == OOB_OFF ? OOB_OFF : of_ + q * 64u, 0, 0);
                }
                if (ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bxs[dt][e] += xv[e]; bfs[dt][e] += fv[e]; }
                }
            }
        }
        // dWs += dL^T . X : transpose the 16 x M tile of dL through wave-private LDS so that the
        // slice index lands on the lane (A operand i = m, k = point)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<float4*>(myT + li * TP + 16 * mt + 4 * kq) = make_float4(dw[mt][0], dw[mt][1], dw[mt][2], dw[mt][3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float a = myT[(4 * kq + r) * TP + 16 * mt + li];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) wsacc[mt][dt] = mfma16(a, xv[r][dt], wsacc[mt][dt]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { xf[ks] = xf_n[ks]; ff[ks] = ff_n[ks]; gf[ks] = gf_n[ks]; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) xv[r][dt] = xv_n[r][dt];
    }
#undef BW_LOAD

    // block partials: dWs [M][D], dbs [M], dtau — waves add in fixed order
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbacc[mt][r] = row16_sum(dbacc[mt][r]);
    dtacc = wave_sum(dtacc);
    float* const rW = red;
    float* const rB = red + MP * DP;
    float* const rT = rB + MP;
    float* const rX = rT + 4;                // [2][DP] column sums of dX | dF (PL only)
    if constexpr (PL != 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { bxs[dt][r] = row16_sum(bxs[dt][r]); bfs[dt][r] = row16_sum(bfs[dt][r]); }
    }
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mt + 4 * kq + r;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int idx = m * DP + 16 * dt + li;
                        rW[idx] = (wv == 0 ? 0.f : rW[idx]) + wsacc[mt][dt][r];
                    }
                    if (li == 0) rB[m] = (wv == 0 ? 0.f : rB[m]) + dbacc[mt][r];
                }
            if (lane == 0) rT[0] = (wv == 0 ? 0.f : rT[0]) + dtacc;
            if constexpr (PL != 0) {
                if (li == 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int d = 16 * dt + 4 * kq + r;
                            rX[d] = (wv == 0 ? 0.f : rX[d]) + bxs[dt][r];
                            rX[DP + d] = (wv == 0 ? 0.f : rX[DP + d]) + bfs[dt][r];
                        }
                }
            }
        }
        __syncthreads();
    }
    float* po = p.part + (size_t)bid * p.stride;
    for (int i = tid; i < p.M * D; i += 256) po[i] = rW[(i / D) * DP + (i % D)] * inv_tau;
    for (int i = tid; i < p.M; i += 256) po[p.M * D + i] = rB[i] * inv_tau;
    if (tid == 0) po[p.M * D + p.M] = -rT[0] * inv_tau;
    if constexpr (PL != 0)
        for (int i = tid; i < 2 * D; i += 256) po[p.M * D + p.M + 1 + i] = rX[(i / D) * DP + (i % D)];
}

// dtemperature[h] = mask(0.1 <= t <= 5) * sum over (b, chunk) of the per-block dtau partials
// both finalize passes sum B*nchunk per-workgroup records per output: latency-bound, so the records are spread over the
// lanes (fixed lane -> record assignment and a fixed reduction tree: deterministic)
__global__ __launch_bounds__(64) void dtau_finalize_kernel(const float* __restrict__ part, const float* __restrict__ temperature,
                                     float* __restrict__ dtemp, int B, int heads, int nchunk, int stride, int off,
                                     int clamp, int accumulate) {
    const int hh = blockIdx.x, lane = threadIdx.x;      // one wave per head
    float s = 0.f;
    for (int r = lane; r < B * nchunk; r += 64) {
        const int b = r / nchunk, c = r - b * nchunk;
        s += part[(size_t)((b * heads + hh) * nchunk + c) * stride + off];
    }
    s = wave_sum(s);
    if (lane != 0) return;
    const float t = temperature[hh];
    const float v = (!clamp || (t >= 0.1f && t <= 5.0f)) ? s : 0.f;
    dtemp[hh] = accumulate ? dtemp[hh] + v : v;
}
__global__ __launch_bounds__(256) void conv_bias_finalize_kernel(const float* __restrict__ part, float* __restrict__ dbx,
                                          float* __restrict__ dbf, int B, int heads, int nchunk, int stride, int off,
                                          int D, int accumulate) {
    __shared__ float red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 outputs x 16 record lanes
    const int idx = blockIdx.x * 16 + tx;
    const bool ok = idx < 2 * heads * D;
    const int which = ok ? idx / (heads * D) : 0, c = idx - which * heads * D, hh = ok ? c / D : 0, d = c - hh * D;
    float s = 0.f;
    if (ok)
        for (int r = ty; r < B * nchunk; r += 16) {
            const int b = r / nchunk, ch = r - b * nchunk;
            s += part[(size_t)((b * heads + hh) * nchunk + ch) * stride + off + which * D + d];
        }
    red[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || !ok) return;
    float t = red[0][tx];
#pragma unroll
    for (int l = 1; l < 16; ++l) t += red[l][tx];
    float* dst = which ? dbf : dbx;
    dst[c] = accumulate ? dst[c] + t : t;
}

// ---------------------------------------------------------------------------------------------
// dispatch over the compile-time (D, MT) grid
template <int D, int MT>
static void launch_scatter_t(const SliceParams& p, int grid, hipStream_t st, bool bf) {
    if (bf) hipLaunchKernelGGL((slice_scatter_kernel<D, MT, bf16_t>), dim3(slice_grid(grid)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((slice_scatter_kernel<D, MT, float>), dim3(slice_grid(grid)), dim3(256), 0, st, p);
}
template <int D, int MT>
static void launch_deslice_t(const DesliceParams& p, int grid, hipStream_t st, bool bf) {
    if (bf) hipLaunchKernelGGL((deslice_kernel<D, MT, bf16_t>), dim3(slice_grid(grid)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((deslice_kernel<D, MT, float>), dim3(slice_grid(grid)), dim3(256), 0, st, p);
}
template <int D, int MT>
static size_t bwd_smem_bytes() {
    constexpr int DT = SCfg<D>::DT, MP = 16 * MT, DP = 16 * DT, P = DP + 4, TP = MP + 4;
    static_assert(MP * DP + MP + 4 + 2 * DP <= 3 * MP * P, "reduction scratch must fit in the aliased region");
    return sizeof(float) * (size_t)(3 * MP * P + 2 * MP + 4 * 16 * TP);
}
template <int D, int MT, typename T, int PL>
static int launch_bwd_one(const SliceBwdParams& p, int grid, hipStream_t st, size_t smem) {
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&slice_bwd_kernel<D, MT, T, PL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((slice_bwd_kernel<D, MT, T, PL>), dim3(slice_grid(grid)), dim3(256), smem, st, p);
    return PA2D_OK;
}
template <int D, int MT>
static int launch_bwd_t(const SliceBwdParams& p, int grid, hipStream_t st, bool bf) {
    const size_t smem = bwd_smem_bytes<D, MT>();
    if (p.planes_nt == 3) return bf ? PA2D_ERR_ARG : launch_bwd_one<D, MT, float, 3>(p, grid, st, smem);
    if (p.planes_nt == 1) return bf ? PA2D_ERR_ARG : launch_bwd_one<D, MT, float, 1>(p, grid, st, smem);
    return bf ? launch_bwd_one<D, MT, bf16_t, 0>(p, grid, st, smem) : launch_bwd_one<D, MT, float, 0>(p, grid, st, smem);
}

#define DISPATCH_MT(D_, CALL)                                    \
    switch (mt) {                                                \
        case 1: CALL(D_, 1); break;                              \
        case 2: CALL(D_, 2); break;                              \
        case 4: CALL(D_, 4); break;                              \
        case 8: CALL(D_, 8); break;                              \
        default: return PA2D_ERR_UNSUPPORTED;                    \
    }
#define DISPATCH_D(CALL)                                         \
    switch (D) {                                                 \
        case 8: DISPATCH_MT(8, CALL) break;                      \
        case 16: DISPATCH_MT(16, CALL) break;                    \
        case 32: DISPATCH_MT(32, CALL) break;                    \
        case 64: DISPATCH_MT(64, CALL) break;                    \
        default: return PA2D_ERR_UNSUPPORTED;                    \
    }

static int mt_for(int M) {
    if (M <= 16) return 1;
    if (M <= 32) return 2;
    if (M <= 64) return 4;
    if (M <= 128) return 8;
    return 0;
}

extern "C" {

// number of point chunks per (batch, head) and points per chunk used by every slice-stage kernel
int pa2d_slice_nchunk(int B, int N, int heads) {
    // one WAVE of the v3 kernels owns a (batch, head, chunk) unit: enough units for two waves on each of the 1024 SIMDs,
    // at least 128 points (4 groups of 32) per unit; the token kernels sum any number of chunk records
    const int bh = B * heads > 0 ? B * heads : 1;
    if (N < 1) return 1;
    int nchunk = ceil_div(2048, bh);
    const int maxc = ceil_div(N, 128);
    if (nchunk > maxc) nchunk = maxc;
    if (nchunk < 1) nchunk = 1;
    const int ppc = ceil_div(ceil_div(N, nchunk), 32) * 32;
    return ceil_div(N, ppc);
}
static int ppc_for(int N, int nchunk) { return ceil_div(ceil_div(N, nchunk), 32) * 32; }
// The slice stages exist twice.  PA2D_ENGINE_F32: the kernels of this file, every contraction on the exact-fp32 matrix
// instruction (v_mfma_f32_16x16x4_f32).  PA2D_ENGINE_SPLIT / _BF16 and the bf16-storage entry points: the v3 kernels
// (pa2d_slice3.hip, pa2d_slice3_bwd.hip), bf16 MFMA on exact 3-plane operand splits with fp32 accumulation.  The choice
// is the caller's `engine` argument — nothing here reads the environment for it.
__attribute__((visibility("hidden"))) int pa2d_launch_scatter3(const void* xm, long long ldx, const void* v, long long ldv, const float* ws, const float* bs,
                           const float* temperature, float* spart, float* npart, int B, int N, int heads, int D, int M,
                           int mt, int nchunk, int ppc, unsigned x_bytes, unsigned v_bytes, int clamp, int xcd_map, bool bf,
                           hipStream_t st);
__attribute__((visibility("hidden"))) int pa2d_launch_deslice3(const void* xm, long long ldx, const float* o, const float* ws, const float* bs,
                           const float* temperature, void* y, long long ldy, int B, int N, int heads, int D, int M, int mt,
                           int nchunk, int ppc, unsigned x_bytes, unsigned y_bytes, int clamp, bool bf, hipStream_t st);
__attribute__((visibility("hidden"))) int pa2d_launch_slice_bwd3(
    const void* xm, long long ldx, const void* fm, long long ldf, const void* dy, long long lddy, const float* ws,
    const float* bs, const float* temperature, const float* o, const float* ds, const float* dn, const float* nrm, void* dxm,
    long long lddx, void* dfm, long long lddf, void* planes, unsigned planes_bytes, int planes_nt, int stride, float* part,
    int B, int N, int heads, int D, int M, int mt, int nchunk, int ppc, unsigned x_bytes, unsigned f_bytes,
    unsigned dy_bytes, unsigned dx_bytes, unsigned df_bytes, int clamp, int xcd_map, bool bf, hipStream_t st);
// chunking of the slice BACKWARD kernel (its partial-sum records are its own): one 4-wave workgroup per (batch, head,
// chunk), about two workgroups per CU in one round, at least 128 points per workgroup
static int bwd_nchunk(int B, int N, int heads) {
    const int bh = B * heads > 0 ? B * heads : 1;
    if (N < 1) return 1;
    int nchunk = ceil_div(512, bh);
    const int maxc = ceil_div(N, 128);
    if (nchunk > maxc) nchunk = maxc;
    if (nchunk < 1) nchunk = 1;
    const int ppc = ceil_div(ceil_div(N, nchunk), 32) * 32;
    return ceil_div(N, ppc);
}
static int slice_xcd_map() { return pa2d_env().slice_map; }      // PA2D_SLICE_MAP=legacy: chunk-fastest numbering (A/B timing)
static bool slice_v3(int engine, bool bf) { return bf || engine != 0; }
size_t pa2d_slice_bwd_workspace(int B, int N, int heads, int D, int M);

// spart [B,heads,nchunk,M,D], npart [B,heads,nchunk,M] (NULL to skip the norm); bf: activations stored as bf16
static int slice_scatter_impl(const void* xm, long long ldx, const void* v, long long ldv, const float* ws,
                              const float* bs, const float* temperature, float* spart, float* npart, int B, int N,
                              int heads, int D, int M, int clamp_temperature, int engine, hipStream_t st, hipEvent_t ev_start,
                              hipEvent_t ev_stop, bool bf) {
    const int mt = mt_for(M);
    if ((ldx & 3) || (D & 7) || engine < 0 || engine > 2) return PA2D_ERR_ARG;
    if (B <= 0 || N <= 0) return PA2D_OK;
    const unsigned long long es = bf ? 2ull : 4ull;
    SliceParams p;
    p.xm = xm; p.ldx = ldx; p.v = v; p.ldv = ldv; p.ws = ws; p.bs = bs; p.temperature = temperature;
    p.spart = spart; p.npart = npart; p.B = B; p.N = N; p.heads = heads; p.M = M; p.clamp = clamp_temperature; p.xcd_map = slice_xcd_map();
    p.nchunk = pa2d_slice_nchunk(B, N, heads);
    p.ppc = ppc_for(N, p.nchunk);
    {
        const unsigned long long rows = (unsigned long long)B * N;
        const unsigned long long xb = ((rows - 1) * ldx + (unsigned long long)heads * D) * es;
        const unsigned long long vb = ((rows - 1) * ldv + (unsigned long long)heads * D) * es;
        if (xb >= 0xFFFFFFF0ull || vb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.x_bytes = (unsigned)xb; p.v_bytes = (unsigned)vb;
    }
    const int grid = B * heads * p.nchunk;
    if (ev_start && hipEventRecord(ev_start, st) != hipSuccess) return PA2D_ERR_ARG;
    if (slice_v3(engine, bf)) {
        const int rc = pa2d_launch_scatter3(xm, ldx, v, ldv, ws, bs, temperature, spart, npart, B, N, heads, D, M, mt,
                                            p.nchunk, p.ppc, p.x_bytes, p.v_bytes, clamp_temperature, p.xcd_map, bf, st);
        if (rc) return rc;
    } else {
#define CALL_SC(D_, MT_) launch_scatter_t<D_, MT_>(p, grid, st, bf)
        DISPATCH_D(CALL_SC)
    }
    PA2D_CHECK_LAUNCH();
    if (ev_stop && hipEventRecord(ev_stop, st) != hipSuccess) return PA2D_ERR_ARG;
    return PA2D_OK;
}

static int deslice_impl(const void* xm, long long ldx, const float* o, const float* ws, const float* bs,
                        const float* temperature, void* y, long long ldy, int B, int N, int heads, int D, int M,
                        int clamp_temperature, int engine, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, bool bf) {
    const int mt = mt_for(M);
    if ((ldx & 3) || (ldy & 3) || (D & 7) || engine < 0 || engine > 2) return PA2D_ERR_ARG;
    if (B <= 0 || N <= 0) return PA2D_OK;
    const unsigned long long es = bf ? 2ull : 4ull;
    DesliceParams p;
    p.xm = xm; p.ldx = ldx; p.o = o; p.ws = ws; p.bs = bs; p.temperature = temperature; p.y = y; p.ldy = ldy;
    p.B = B; p.N = N; p.heads = heads; p.M = M; p.clamp = clamp_temperature; p.xcd_map = slice_xcd_map();
    p.nchunk = pa2d_slice_nchunk(B, N, heads);
    p.ppc = ppc_for(N, p.nchunk);
    {
        const unsigned long long rows = (unsigned long long)B * N;
        const unsigned long long xb = ((rows - 1) * ldx + (unsigned long long)heads * D) * es;
        const unsigned long long yb = ((rows - 1) * ldy + (unsigned long long)heads * D) * es;
        if (xb >= 0xFFFFFFF0ull || yb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
    }
    const int grid = B * heads * p.nchunk;
    if (ev_start && hipEventRecord(ev_start, st) != hipSuccess) return PA2D_ERR_ARG;
    if (slice_v3(engine, bf)) {
        const int rc = pa2d_launch_deslice3(xm, ldx, o, ws, bs, temperature, y, ldy, B, N, heads, D, M, mt, p.nchunk, p.ppc,
                                            p.x_bytes, p.y_bytes, clamp_temperature, bf, st);
        if (rc) return rc;
    } else {
#define CALL_DS(D_, MT_) launch_deslice_t<D_, MT_>(p, grid, st, bf)
        DISPATCH_D(CALL_DS)
    }
    PA2D_CHECK_LAUNCH();
    if (ev_stop && hipEventRecord(ev_stop, st) != hipSuccess) return PA2D_ERR_ARG;
    return PA2D_OK;
}

// Backward phase C.  dws [M,D], dbs [M], dtemperature [heads] are fully reduced on return.
static int slice_bwd_impl(const void* xm, long long ldx, const void* fm, long long ldf, const void* dy, long long lddy,
                          const float* ws, const float* bs, const float* temperature, const float* o, const float* ds,
                          const float* dn, void* dxm, long long lddx, void* dfm, long long lddf, float* dws, float* dbs,
                          float* dtemperature, void* ws_buf, size_t ws_bytes, int B, int N, int heads, int D, int M,
                          int clamp_temperature, int accumulate, int engine, hipStream_t st, hipEvent_t ev_start,
                          hipEvent_t ev_stop, bool bf, void* planes = nullptr, int planes_nt = 0, float* dbx = nullptr,
                          float* dbf = nullptr, const float* nrm = nullptr) {
    const int mt = mt_for(M);
    if (engine < 0 || engine > 2) return PA2D_ERR_ARG;
    if ((ldx & 3) || (ldf & 3) || (lddy & 3) || (lddx & 3) || (lddf & 3) || (D & 7)) return PA2D_ERR_ARG;
    if (B <= 0 || N <= 0) {
        if (accumulate) return PA2D_OK;
        int rz = pa2d_zero(dws, sizeof(float) * M * D, st);
        if (!rz) rz = pa2d_zero(dbs, sizeof(float) * M, st);
        return rz ? rz : pa2d_zero(dtemperature, sizeof(float) * heads, st);
    }
    if (ws_bytes < pa2d_slice_bwd_workspace(B, N, heads, D, M)) return PA2D_ERR_WORKSPACE;
    const unsigned long long es = bf ? 2ull : 4ull;
    SliceBwdParams p;
    p.xm = xm; p.ldx = ldx; p.fm = fm; p.ldf = ldf; p.dy = dy; p.lddy = lddy; p.ws = ws; p.bs = bs;
    p.temperature = temperature; p.o = o; p.ds = ds; p.dn = dn; p.dxm = dxm; p.lddx = lddx; p.dfm = dfm;
    p.lddf = lddf; p.part = (float*)ws_buf; p.B = B; p.N = N; p.heads = heads; p.M = M; p.clamp = clamp_temperature;
    p.xcd_map = slice_xcd_map();
    p.nchunk = bwd_nchunk(B, N, heads);
    p.ppc = ppc_for(N, p.nchunk);
    p.stride = M * D + M + 1 + 2 * D;
    p.planes = planes; p.planes_nt = planes ? planes_nt : 0; p.planes_bytes = 0;
    if (planes) {
        if ((planes_nt != 1 && planes_nt != 3) || ((heads * D) & 31) || (D & 3)) return PA2D_ERR_UNSUPPORTED;
        const unsigned long long pb = (unsigned long long)B * N * (2ull * heads * D) * planes_nt * 2ull;
        if (pb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.planes_bytes = (unsigned)pb;
        if (!dxm) { p.dxm = const_cast<void*>(xm); lddx = ldx; p.lddx = ldx; }     // descriptors need a base; nothing is stored there
        if (!dfm) { p.dfm = const_cast<void*>(fm); lddf = ldf; p.lddf = ldf; }
    }
    {
        const unsigned long long rows = (unsigned long long)B * N, w = (unsigned long long)heads * D;
        const unsigned long long e[5] = {((rows - 1) * ldx + w) * es, ((rows - 1) * ldf + w) * es,
                                         ((rows - 1) * lddy + w) * es, ((rows - 1) * lddx + w) * es,
                                         ((rows - 1) * lddf + w) * es};
        for (int i = 0; i < 5; ++i)
            if (e[i] >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.x_bytes = (unsigned)e[0]; p.f_bytes = (unsigned)e[1]; p.dy_bytes = (unsigned)e[2];
        p.dx_bytes = (unsigned)e[3]; p.df_bytes = (unsigned)e[4];
    }
    const int grid = B * heads * p.nchunk;
    int rc = PA2D_OK;
    if (ev_start && hipEventRecord(ev_start, st) != hipSuccess) return PA2D_ERR_ARG;
    if (slice_v3(engine, bf)) {
        if (planes && !nrm) return PA2D_ERR_ARG;
        rc = pa2d_launch_slice_bwd3(p.xm, p.ldx, p.fm, p.ldf, p.dy, p.lddy, ws, bs, temperature, o, ds, dn, nrm, p.dxm, p.lddx, p.dfm,
                                    p.lddf, p.planes, p.planes_bytes, p.planes_nt, p.stride, p.part, B, N, heads, D, M, mt,
                                    p.nchunk, p.ppc, p.x_bytes, p.f_bytes, p.dy_bytes, p.dx_bytes, p.df_bytes,
                                    clamp_temperature, p.xcd_map, bf, st);
    } else {
        rc = PA2D_ERR_UNSUPPORTED;
    }
    if (rc == PA2D_ERR_UNSUPPORTED) {
#define CALL_BW(D_, MT_) rc = launch_bwd_t<D_, MT_>(p, grid, st, bf)
        DISPATCH_D(CALL_BW)
    }
    if (rc) return rc;
    PA2D_CHECK_LAUNCH();
    if (ev_stop && hipEventRecord(ev_stop, st) != hipSuccess) return PA2D_ERR_ARG;
    // sum the per-block records [dWs | dbs | dtau | ..] straight into dws / dbs (the dtau column is finalised below)
    const int stride = p.stride;
    ReduceSegs segs;
    segs.nseg = 2;
    segs.begin[0] = 0; segs.begin[1] = (long long)M * D; segs.begin[2] = (long long)M * D + M;
    segs.begin[3] = segs.begin[4] = segs.begin[2];
    segs.dst[0] = dws; segs.dst[1] = dbs; segs.dst[2] = segs.dst[3] = nullptr;
    rc = pa2d_launch_reduce_segs(p.part, grid, stride, segs, accumulate, st);
    if (rc) return rc;
    hipLaunchKernelGGL(dtau_finalize_kernel, dim3(heads), dim3(64), 0, st, p.part, temperature,
                       dtemperature, B, heads, p.nchunk, stride, M * D + M, clamp_temperature, accumulate);
    PA2D_CHECK_LAUNCH();
    if (planes && dbx && dbf) {
        hipLaunchKernelGGL(conv_bias_finalize_kernel, dim3(ceil_div(2 * heads * D, 16)), dim3(256), 0, st, p.part, dbx, dbf, B,
                           heads, p.nchunk, stride, M * D + M + 1, D, accumulate);
        PA2D_CHECK_LAUNCH();
    }
    return PA2D_OK;
}

int pa2d_slice_scatter(const float* xm, long long ldx, const float* v, long long ldv, const float* ws,
                       const float* bs, const float* temperature, float* spart, float* npart, int B, int N,
                       int heads, int D, int M, int clamp_temperature, int engine, hipStream_t st, hipEvent_t ev_start,
                       hipEvent_t ev_stop) {
    return slice_scatter_impl(xm, ldx, v, ldv, ws, bs, temperature, spart, npart, B, N, heads, D, M, clamp_temperature,
                              engine, st, ev_start, ev_stop, false);
}
int pa2d_slice_scatter_bf16(const void* xm, long long ldx, const void* v, long long ldv, const float* ws,
                            const float* bs, const float* temperature, float* spart, float* npart, int B, int N,
                            int heads, int D, int M, int clamp_temperature, hipStream_t st, hipEvent_t ev_start,
                            hipEvent_t ev_stop) {
    return slice_scatter_impl(xm, ldx, v, ldv, ws, bs, temperature, spart, npart, B, N, heads, D, M, clamp_temperature,
                              PA2D_ENGINE_BF16_ID, st, ev_start, ev_stop, true);
}

int pa2d_deslice_fwd(const float* xm, long long ldx, const float* o, const float* ws, const float* bs,
                     const float* temperature, float* y, long long ldy, int B, int N, int heads, int D, int M,
                     int clamp_temperature, int engine, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    return deslice_impl(xm, ldx, o, ws, bs, temperature, y, ldy, B, N, heads, D, M, clamp_temperature, engine, st, ev_start,
                        ev_stop, false);
}
int pa2d_deslice_fwd_bf16(const void* xm, long long ldx, const float* o, const float* ws, const float* bs,
                          const float* temperature, void* y, long long ldy, int B, int N, int heads, int D, int M,
                          int clamp_temperature, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    return deslice_impl(xm, ldx, o, ws, bs, temperature, y, ldy, B, N, heads, D, M, clamp_temperature, PA2D_ENGINE_BF16_ID, st,
                        ev_start, ev_stop, true);
}

size_t pa2d_slice_bwd_workspace(int B, int N, int heads, int D, int M) {
    const int nchunk = bwd_nchunk(B, N, heads);
    return sizeof(float) * ((size_t)B * heads * nchunk + 1) * ((size_t)M * D + M + 1 + 2 * D);
}

int pa2d_slice_bwd_points(const float* xm, long long ldx, const float* fm, long long ldf, const float* dy,
                          long long lddy, const float* ws, const float* bs, const float* temperature,
                          const float* o, const float* ds, const float* dn, float* dxm, long long lddx, float* dfm,
                          long long lddf, float* dws, float* dbs, float* dtemperature, void* ws_buf, size_t ws_bytes,
                          int B, int N, int heads, int D, int M, int clamp_temperature, int accumulate, int engine,
                          hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    return slice_bwd_impl(xm, ldx, fm, ldf, dy, lddy, ws, bs, temperature, o, ds, dn, dxm, lddx, dfm, lddf, dws, dbs,
                          dtemperature, ws_buf, ws_bytes, B, N, heads, D, M, clamp_temperature, accumulate, engine, st,
                          ev_start, ev_stop, false);
}
int pa2d_slice_bwd_points_bf16(const void* xm, long long ldx, const void* fm, long long ldf, const void* dy,
                               long long lddy, const float* ws, const float* bs, const float* temperature,
                               const float* o, const float* ds, const float* dn, void* dxm, long long lddx, void* dfm,
                               long long lddf, float* dws, float* dbs, float* dtemperature, void* ws_buf,
                               size_t ws_bytes, int B, int N, int heads, int D, int M, int clamp_temperature,
                               int accumulate, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    return slice_bwd_impl(xm, ldx, fm, ldf, dy, lddy, ws, bs, temperature, o, ds, dn, dxm, lddx, dfm, lddf, dws, dbs,
                          dtemperature, ws_buf, ws_bytes, B, N, heads, D, M, clamp_temperature, accumulate,
                          PA2D_ENGINE_BF16_ID, st, ev_start, ev_stop, true);
}

// Same, but [dX | dF] leaves ONLY as the bf16 plane image the conv GEMMs of `engine` stage (NT = 3 planes for
// PA2D_ENGINE_SPLIT, 1 for PA2D_ENGINE_BF16; pa2d_planes_bytes(B*N, 2*heads*D, engine) bytes) — no fp32 tensor, no split
// pre-pass in the conv backward — and the conv bias gradients dbx / dbf [heads*D] (column sums of dX / dF) come out of the
// same partial-sum records.
int pa2d_slice_bwd_points_planes(const float* xm, long long ldx, const float* fm, long long ldf, const float* dy,
                                 long long lddy, const float* ws, const float* bs, const float* temperature,
                                 const float* o, const float* ds, const float* dn, const float* nrm, void* dxf_planes,
                                 float* dbx, float* dbf, float* dws, float* dbs, float* dtemperature, void* ws_buf,
                                 size_t ws_bytes, int B, int N, int heads, int D, int M, int clamp_temperature,
                                 int accumulate, int engine, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (engine != 1 && engine != 2) return PA2D_ERR_ARG;
    if (!dxf_planes || !dbx || !dbf || !nrm) return PA2D_ERR_ARG;
    return slice_bwd_impl(xm, ldx, fm, ldf, dy, lddy, ws, bs, temperature, o, ds, dn, nullptr, 0, nullptr, 0, dws, dbs,
                          dtemperature, ws_buf, ws_bytes, B, N, heads, D, M, clamp_temperature, accumulate, engine, st,
                          ev_start, ev_stop, false, dxf_planes, engine == 2 ? 1 : 3, dbx, dbf, nrm);
}

}  // extern "C"
