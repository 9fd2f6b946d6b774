// Slice scatter / de-slice on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) — gfx950 / CDNA4.
//
// The kernels in pa2d_slice.hip run every contraction on v_mfma_f32_16x16x4_f32.  They read each operand exactly once,
// yet sit at 0.30 of the HBM roofline because that instruction retires 64 FLOP/clk/SIMD: 112 us per scatter launch at
// the bench shape against a 34 us HBM floor (fp32 MFMA floor 62 us).  v_mfma_f32_16x16x32_bf16 retires 1024 FLOP/clk/SIMD,
// so the same contractions are evaluated here as a few bf16 terms of EXACT operand splits with fp32 accumulation:
//   fp32 storage: x = hi + mid + lo (3 bf16 planes, exact to 2^-25 |x|), six product terms of order <= 2 — the same
//                 fp32-accurate scheme as the conv GEMMs of the split engine, same parity tolerances;
//   bf16 storage: the activations ARE one plane; parameters (Ws, O) keep 3 planes, the slice weights 2.
// What remains is the fp32 softmax (VALU, v_exp_f32) and the HBM stream.
//
// Register layouts (lane l: i = l & 15, kq = l >> 4):
//   A operand of 16x16x32: row i, k = 8*kq .. 8*kq+7;  B operand: column i, k = 8*kq .. 8*kq+7;
//   C/D: column i, rows 4*kq + r (r = 0..3).
// N-layout (rows = points, cols = slices) for the scatter: the accumulator of the logits tile of 16 points holds
// W[point 4kq+r][slice i]; two such tiles (32 points) give lane (slice i, kq) the 8 values W[points 4kq..4kq+3 and
// 16+4kq..16+4kq+3][i] = the A operand (rows = slices, k = points) of S += W^T F, with the SAME point permutation
// applied to the rows of F loaded for the B operand.
// T-layout (rows = slices, cols = points) for the de-slice: two logits tiles of 16 slices give lane (point i, kq) the 8
// values W[i][slices 4kq.. of tile 2u and of tile 2u+1] = the B operand (k = slices, cols = points) of Y^T = O^T W^T.
#include "pa2d_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NEG_BIG (-1e30f)

namespace {

__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
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
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float kq_max(float v) {
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

// exact split of 8 floats into NP bf16 planes (x = p0 + p1 + p2 up to 2^-25 |x|)
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&pl)[NP]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        pl[0][e] = h;
        if constexpr (NP > 1) {
            const float r1 = x[e] - (float)h;
            const __bf16 m = (__bf16)r1;
            pl[1][e] = m;
            if constexpr (NP > 2) pl[2][e] = (__bf16)(r1 - (float)m);
        }
    }
}

// planes per operand and the product terms kept (smallest first): fp32 storage = six terms of the 3 x 3 split;
// bf16 storage = activations exact in one plane
template <typename T> struct Planes;
template <> struct Planes<float> { static constexpr int ACT = 3, PAR = 3, WGT = 3; };
template <> struct Planes<bf16_t> { static constexpr int ACT = 1, PAR = 3, WGT = 2; };

// acc += sum over the kept terms of a[i] * b[j]  (terms with i + j <= 2, smallest first)
template <int NA, int NB>
__device__ __forceinline__ f32x4 mfma_terms(const bf16x8 (&a)[NA], const bf16x8 (&b)[NB], f32x4 acc) {
#pragma unroll
    for (int s = 2; s >= 0; --s)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int j = s - i;
            if (j >= 0 && j < NB) acc = mfma_bf(a[i], b[j], acc);
        }
    return acc;
}

// 8 consecutive activation elements of one row: loaded RAW (prefetch registers stay small: 4 VGPRs for bf16, 8 for
// fp32) and turned into planes only when consumed; off = byte offset of the first element (OOB_OFF -> zeros)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u32x4 q; };
template <> struct Raw8<float> { float4 a, b; };
template <typename T>
__device__ __forceinline__ void load_raw8(__amdgpu_buffer_rsrc_t r, unsigned off, Raw8<T>& x) {
    if constexpr (sizeof(T) == 2) {
        x.q = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    } else {
        x.a = buf_load4(r, off);
        x.b = buf_load4(r, off == OOB_OFF ? OOB_OFF : off + 16u);
    }
}
template <typename T>
__device__ __forceinline__ void raw_planes(const Raw8<T>& x, bf16x8 (&pl)[Planes<T>::ACT]) {
    if constexpr (sizeof(T) == 2) {
        pl[0] = __builtin_bit_cast(bf16x8, x.q);
    } else {
        const float v[8] = {x.a.x, x.a.y, x.a.z, x.a.w, x.b.x, x.b.y, x.b.z, x.b.w};
        split8<3>(v, pl);
    }
}

template <int D> struct BCfg {
    static constexpr int KST = (D + 31) / 32;      // 32-wide k-steps of a contraction over d
    static constexpr int DT = (D + 15) / 16;       // 16-wide tiles over d
};

struct BfSliceParams {
    const void* xm; long long ldx;
    const void* v; long long ldv;
    const float* ws; const float* bs; const float* temperature;
    float* spart; float* npart;
    int B, N, heads, M, nchunk, ppc;
    unsigned x_bytes, v_bytes;
    int clamp, xcd_map;
};

// parameter fragment: row `m` of a [M][D] fp32 matrix, columns 32*s + 8*kq .. +7, as 3 planes (zeros outside)
template <int D>
__device__ __forceinline__ void load_par8(const float* mat, int m, int M, int s, int kq, bf16x8 (&pl)[3]) {
    float x[8];
    const int d0 = 32 * s + 8 * kq;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (m < M && d0 + e < D) ? mat[(size_t)m * D + d0 + e] : 0.f;
    split8<3>(x, pl);
}

}  // namespace

// S_partial[m][d] = sum_{n in chunk} W[n][m] * V[n][d];  n_partial[m] = sum_n W[n][m]
template <int D, int MT, typename T>
__global__ __launch_bounds__(256) void slice_scatter_bf_kernel(const BfSliceParams p) {
    constexpr int KST = BCfg<D>::KST, DT = BCfg<D>::DT;
    constexpr int NA = Planes<T>::ACT, NW = Planes<T>::WGT;
    constexpr int MP = 16 * MT, DP = 16 * DT;
    constexpr unsigned ES = Act<T>::ES;
    __shared__ float sbuf[MP * DP];
    __shared__ float nbuf[MP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    int b, hh, chunk, bid;
    if (!slice_decode(p.xcd_map, p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float inv_tau = 1.0f / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);

    bf16x8 wsp[MT][KST][3];      // B operand of the logits: column = slice 16mt+li, k = d
    float bsv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + li;
#pragma unroll
        for (int s = 0; s < KST; ++s) load_par8<D>(p.ws, m, p.M, s, kq, wsp[mt][s]);
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
    const unsigned ldxb = (unsigned)p.ldx * ES, ldvb = (unsigned)p.ldv * ES, hcol = (unsigned)(hh * D) * ES;

    // one group = 32 points = two logits tiles.  X: lane (point li of tile t, kq) loads 8 consecutive d per k-step;
    // V: lane (channel li, kq) loads the 8 rows 4kq+e (e < 4) and 16+4kq+e-4 of the group
#define SB_LOAD(g_, XP, FV)                                                                               \
    {                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                   \
            const int pt_ = (g_) + 16 * t + li;                                                           \
            _Pragma("unroll") for (int s = 0; s < KST; ++s) {                                             \
                const int d0_ = 32 * s + 8 * kq;                                                          \
                load_raw8<T>(rx, (pt_ < p_end && d0_ < D) ? (row0 + pt_) * ldxb + hcol + d0_ * ES : OOB_OFF, XP[t][s]); \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                   \
            const int pr_ = (g_) + (e < 4 ? 4 * kq + e : 12 + 4 * kq + e);                                \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                           \
                const int d_ = 16 * dt + li;                                                              \
                FV[dt][e] = Act<T>::bld1(rv, (pr_ < p_end && d_ < D) ? (row0 + pr_) * ldvb + hcol + d_ * ES : OOB_OFF); \
            }                                                                                             \
        }                                                                                                 \
    }
    Raw8<T> xp[2][KST], xp_n[2][KST];
    float fv[DT][8], fv_n[DT][8];
    int g = p_begin + wave * 32;
    if (g < p_end) SB_LOAD(g, xp, fv)
    for (; g < p_end; g += 128) {
        if (g + 128 < p_end) SB_LOAD(g + 128, xp_n, fv_n)
        float w[2][MT][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                bf16x8 xpl[NA];
                raw_planes<T>(xp[t][s], xpl);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_terms<NA, 3>(xpl, wsp[mt][s], acc[mt]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mx = NEG_BIG;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float z = (acc[mt][r] + bsv[mt]) * inv_tau;
                    if (16 * mt + li >= p.M) z = NEG_BIG;
                    w[t][mt][r] = z;
                    mx = fmaxf(mx, z);
                }
                mx = row16_max(mx);
                float sm = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float e = fast_exp(w[t][mt][r] - mx);
                    w[t][mt][r] = e;
                    sm += e;
                }
                sm = row16_sum(sm);
                const float inv = (g + 16 * t + 4 * kq + r) < p_end ? 1.0f / sm : 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    w[t][mt][r] *= inv;
                    nacc[mt] += w[t][mt][r];
                }
            }
        }
        // S += W^T F over the 32 points of the group
        bf16x8 fp[DT][NA];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if constexpr (NA == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) fp[dt][0][e] = (__bf16)fv[dt][e];      // exact: the values are bf16
            } else {
                split8<NA>(fv[dt], fp[dt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float wv[8] = {w[0][mt][0], w[0][mt][1], w[0][mt][2], w[0][mt][3],
                                 w[1][mt][0], w[1][mt][1], w[1][mt][2], w[1][mt][3]};
            bf16x8 wp[NW];
            split8<NW>(wv, wp);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) sacc[mt][dt] = mfma_terms<NW, NA>(wp, fp[dt], sacc[mt][dt]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < KST; ++s) xp[t][s] = xp_n[t][s];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[dt][e] = fv_n[dt][e];
    }
#undef SB_LOAD
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

struct BfDesliceParams {
    const void* xm; long long ldx;
    const float* o;
    const float* ws; const float* bs; const float* temperature;
    void* y; long long ldy;
    int B, N, heads, M, nchunk, ppc;
    unsigned x_bytes, y_bytes;
    int clamp, xcd_map;
};

// Y[n][h*D+d] = sum_m W[n][m] * O[m][d]
template <int D, int MT, typename T>
__global__ __launch_bounds__(256) void deslice_bf_kernel(const BfDesliceParams p) {
    constexpr int KST = BCfg<D>::KST, DT = BCfg<D>::DT;
    constexpr int NA = Planes<T>::ACT, NW = Planes<T>::WGT;
    constexpr int MU = (MT + 1) / 2;                 // 32-slice k-steps of the contraction over m
    constexpr unsigned ES = Act<T>::ES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    int b, hh, chunk, bid;
    if (!slice_decode(p.xcd_map, p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float inv_tau = 1.0f / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);

    bf16x8 wsp[MT][KST][3];      // A operand of the transposed logits: row = slice 16mt+li, k = d
    float bst[MT][4];
    bf16x8 op[MU][DT][3];        // A operand of Y^T = O^T W^T: row = channel 16dt+li, k = slices of k-step u
    const float* ob = p.o + (size_t)(b * p.heads + hh) * p.M * D;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int s = 0; s < KST; ++s) load_par8<D>(p.ws, 16 * mt + li, p.M, s, kq, wsp[mt][s]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mr = 16 * mt + 4 * kq + r;
            bst[mt][r] = mr < p.M ? p.bs[mr] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float x[8];
            const int d = 16 * dt + li;
#pragma unroll
            for (int e = 0; e < 8; ++e) {       // slot e of k-block kq: slice 16(2u) + 4kq + e, or 16(2u+1) + 4kq + e - 4
                const int m = e < 4 ? 32 * u + 4 * kq + e : 32 * u + 12 + 4 * kq + e;
                x[e] = (m < p.M && d < D) ? ob[(size_t)m * D + d] : 0.f;
            }
            split8<3>(x, op[u][dt]);
        }
    const int p_begin = chunk * p.ppc;
    const int p_end = min(p.N, p_begin + p.ppc);
    const unsigned row0 = (unsigned)b * (unsigned)p.N;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_v(p.xm, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc_v(p.y, p.y_bytes);
    const unsigned ldxb = (unsigned)p.ldx * ES, ldyb = (unsigned)p.ldy * ES, hcol = (unsigned)(hh * D) * ES;
#define DB_LOAD(g_, XP)                                                                                   \
    {                                                                                                     \
        const int pt_ = (g_) + li;                                                                        \
        _Pragma("unroll") for (int s = 0; s < KST; ++s) {                                                 \
            const int d0_ = 32 * s + 8 * kq;                                                              \
            load_raw8<T>(rx, (pt_ < p_end && d0_ < D) ? (row0 + pt_) * ldxb + hcol + d0_ * ES : OOB_OFF, XP[s]); \
        }                                                                                                 \
    }
    Raw8<T> xp[KST], xp_n[KST];
    int g = p_begin + wave * 16;
    if (g < p_end) DB_LOAD(g, xp)
    for (; g < p_end; g += 64) {
        const int pt = g + li;
        const bool pv = pt < p_end;
        if (g + 64 < p_end) DB_LOAD(g + 64, xp_n)
        f32x4 w[MT];
        float mx = NEG_BIG;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) w[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            bf16x8 xpl[NA];
            raw_planes<T>(xp[s], xpl);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) w[mt] = mfma_terms<3, NA>(wsp[mt][s], xpl, w[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
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
        for (int u = 0; u < MU; ++u) {
            float wv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wv[e] = w[2 * u][e] * inv;
                wv[4 + e] = (2 * u + 1 < MT) ? w[(2 * u + 1 < MT) ? 2 * u + 1 : 0][e] * inv : 0.f;
            }
            bf16x8 wp[NW];
            split8<NW>(wv, wp);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) yacc[dt] = mfma_terms<3, NW>(op[u][dt], wp, yacc[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * kq;
            Act<T>::bst4(ry, (pv && d < D) ? (row0 + pt) * ldyb + hcol + d * ES : OOB_OFF,
                         make_float4(yacc[dt][0], yacc[dt][1], yacc[dt][2], yacc[dt][3]));
        }
#pragma unroll
        for (int s = 0; s < KST; ++s) xp[s] = xp_n[s];
    }
#undef DB_LOAD
}

// ---------------------------------------------------------------------------------------------- host side
#define BF_DISPATCH_MT(D_, CALL)                                 \
    switch (mt) {                                                \
        case 1: CALL(D_, 1); break;                              \
        case 2: CALL(D_, 2); break;                              \
        case 4: CALL(D_, 4); break;                              \
        case 8: CALL(D_, 8); break;                              \
        default: return PA2D_ERR_UNSUPPORTED;                    \
    }
#define BF_DISPATCH_D(CALL)                                      \
    switch (D) {                                                 \
        case 8: BF_DISPATCH_MT(8, CALL) break;                   \
        case 16: BF_DISPATCH_MT(16, CALL) break;                 \
        case 32: BF_DISPATCH_MT(32, CALL) break;                 \
        case 64: BF_DISPATCH_MT(64, CALL) break;                 \
        default: return PA2D_ERR_UNSUPPORTED;                    \
    }

// called by pa2d_slice.hip (from inside its extern "C" block) with the fields of its own parameter structs
extern "C" __attribute__((visibility("hidden"))) int pa2d_launch_scatter_bf(const void* xm, long long ldx, const void* v, long long ldv, const float* ws, const float* bs,
                           const float* temperature, float* spart, float* npart, int B, int N, int heads, int D, int M,
                           int mt, int nchunk, int ppc, unsigned x_bytes, unsigned v_bytes, int clamp, int xcd_map, bool bf,
                           hipStream_t st) {
    BfSliceParams p;
    p.xm = xm; p.ldx = ldx; p.v = v; p.ldv = ldv; p.ws = ws; p.bs = bs; p.temperature = temperature;
    p.spart = spart; p.npart = npart; p.B = B; p.N = N; p.heads = heads; p.M = M; p.nchunk = nchunk; p.ppc = ppc;
    p.x_bytes = x_bytes; p.v_bytes = v_bytes; p.clamp = clamp; p.xcd_map = xcd_map;
    const int grid = B * heads * nchunk;
#define CALL_SB(D_, MT_)                                                                                          \
    if (bf) hipLaunchKernelGGL((slice_scatter_bf_kernel<D_, MT_, bf16_t>), dim3(slice_grid(grid)), dim3(256), 0, st, p);      \
    else hipLaunchKernelGGL((slice_scatter_bf_kernel<D_, MT_, float>), dim3(slice_grid(grid)), dim3(256), 0, st, p)
    BF_DISPATCH_D(CALL_SB)
    return PA2D_OK;
}

extern "C" __attribute__((visibility("hidden"))) int pa2d_launch_deslice_bf(const void* xm, long long ldx, const float* o, const float* ws, const float* bs,
                           const float* temperature, void* y, long long ldy, int B, int N, int heads, int D, int M, int mt,
                           int nchunk, int ppc, unsigned x_bytes, unsigned y_bytes, int clamp, int xcd_map, bool bf, hipStream_t st) {
    BfDesliceParams p;
    p.xm = xm; p.ldx = ldx; p.o = o; p.ws = ws; p.bs = bs; p.temperature = temperature; p.y = y; p.ldy = ldy;
    p.B = B; p.N = N; p.heads = heads; p.M = M; p.nchunk = nchunk; p.ppc = ppc; p.x_bytes = x_bytes; p.y_bytes = y_bytes;
    p.clamp = clamp; p.xcd_map = xcd_map;
    const int grid = B * heads * nchunk;
#define CALL_DB(D_, MT_)                                                                                          \
    if (bf) hipLaunchKernelGGL((deslice_bf_kernel<D_, MT_, bf16_t>), dim3(slice_grid(grid)), dim3(256), 0, st, p);            \
    else hipLaunchKernelGGL((deslice_bf_kernel<D_, MT_, float>), dim3(slice_grid(grid)), dim3(256), 0, st, p)
    BF_DISPATCH_D(CALL_DB)
    return PA2D_OK;
}
