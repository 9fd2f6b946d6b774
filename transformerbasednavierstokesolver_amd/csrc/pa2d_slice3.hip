// Slice scatter / de-slice / slice backward on the bf16 matrix cores with a lean vector stream — gfx950 / CDNA4.
//
// Same math and the same exact operand splits as pa2d_slice_bf.hip (x = hi + mid + lo in bf16 planes, six product terms,
// fp32 accumulation: a 24-bit significand), but the kernels there are bound by the VECTOR unit, not by HBM or the matrix
// pipe (PMC, profiles/r02_e_pmc_slice_kernels.json: 1,230 VALU instructions per 32 points per wave).  What is removed:
//   * the logits come out of the MFMA chain already scaled and biased: Ws and bs are multiplied by log2(e) / tau once per
//     workgroup, the bias (or -1e30 for padding slices) is the C operand of the first MFMA, so softmax is exp2(acc - max);
//   * the softmax normalisation is applied to the D values of a point (scatter: F' = F / Z) or to the D outputs of a point
//     (de-slice: Y = (W~ O) / Z) instead of to its M weights;
//   * the 16-lane row reductions are DPP-fused v_max / v_add (4 instructions per row, hipcc emits mov_dpp + 2 x canonicalise
//     + op), the cross-row ones v_permlane16/32_swap instead of ds_bpermute;
//   * 1 / Z is v_rcp_f32 (hipcc expands 1.0f / x to a 10-instruction IEEE division);
//   * the prefetch registers are swapped by unrolling the group loop twice, not copied.
//
// Register layouts (lane l: li = l & 15, kq = l >> 4) of v_mfma_f32_16x16x32_bf16:
//   A operand: row li, k = 8 kq .. 8 kq + 7;  B operand: column li, same k;  C/D: column li, rows 4 kq + r.
// The k order inside a contraction over d is permuted (element j of lane group kq <-> d = 32 s + 4 kq + j for j < 4,
// 32 s + 16 + 4 kq + (j - 4) otherwise): each of the two 16-byte loads of a point's k-fragment then covers 64 contiguous
// bytes of the row; A and B use the same permutation.
#include "pa2d_internal.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define NEG_BIG (-1e30f)
#define LOG2E 1.44269504088896340736f

namespace {

__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float clamp_tau(float t) { return fminf(fmaxf(t, 0.1f), 5.0f); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// Reductions over the 16 lanes of a DPP row, four independent values at a time (the interleave covers the two wait
// states a DPP read needs after a VALU write of the same register; the leading s_nop covers the producers of the inputs,
// which the compiler's hazard recogniser does not see through the asm).  Every lane of the row ends with the row result.
#ifndef S3_PRE
#define S3_PRE "s_nop 1\n\t"
#endif
#ifndef S3_BC
#define S3_BC ""
#endif
#define ROW16_OP4(OP)                                                                                        \
    asm volatile(S3_PRE                                                                               \
                 OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                         \
                 OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                             \
                 OP " %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                             \
                 OP " %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                             \
                 OP " %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                             \
                 OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                                  \
                 OP " %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                                  \
                 OP " %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                                  \
                 OP " %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf" S3_BC "\n\t"                                  \
                 "s_nop 1"                                                                                   \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max_b(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float row16_sum_b(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ void row16_max4(float& a, float& b, float& c, float& d) { ROW16_OP4("v_max_f32_dpp"); }
__device__ __forceinline__ void row16_sum4(float& a, float& b, float& c, float& d) { ROW16_OP4("v_add_f32_dpp"); }

// Reductions over the four lane groups l, l ^ 16, l ^ 32, l ^ 48 (the kq index) with the gfx950 row swaps:
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second, v_permlane32_swap the
// upper half of the first with the lower half of the second.
#define KQ_OP(OP)                                                                                            \
    float t;                                                                                                 \
    asm volatile("v_mov_b32 %1, %0\n\t"                                                                      \
                 "s_nop 1\n\t"                                                                               \
                 "v_permlane16_swap_b32 %0, %1\n\t"                                                          \
                 OP " %0, %0, %1\n\t"                                                                        \
                 "v_mov_b32 %1, %0\n\t"                                                                      \
                 "s_nop 1\n\t"                                                                               \
                 "v_permlane32_swap_b32 %0, %1\n\t"                                                          \
                 OP " %0, %0, %1\n\t"                                                                        \
                 "s_nop 0"                                                                                   \
                 : "+v"(v), "=&v"(t));                                                                       \
    return v
__device__ __forceinline__ float kq_max(float v) { KQ_OP("v_max_f32"); }
__device__ __forceinline__ float kq_sum(float v) { KQ_OP("v_add_f32"); }

// exact split of 8 floats into NP bf16 planes (x = p0 + p1 + p2 up to 2^-25 |x|).  Written on the PACKED conversion result
// (hipcc otherwise converts every element a second time on its own to form the residual: 7.5 instead of 4.5 instructions
// per element): per pair 3 v_cvt_pk_bf16_f32 + 2 x (v_lshlrev, v_and, v_pk_add_f32).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split8(const f32x8 x, bf16x8 (&pl)[NP]) {
    u32x4 p0, p1, p2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 a = {x[2 * q], x[2 * q + 1]};
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
        p0[q] = h;
        if constexpr (NP > 1) {
            const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
            const f32x2 r = a - hf;                                  // explicit vector op -> v_pk_add_f32
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
            p1[q] = m;
            if constexpr (NP > 2) {
                const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
                p2[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r - mf, bf16x2));
            }
        }
    }
    pl[0] = __builtin_bit_cast(bf16x8, p0);
    if constexpr (NP > 1) pl[1] = __builtin_bit_cast(bf16x8, p1);
    if constexpr (NP > 2) pl[2] = __builtin_bit_cast(bf16x8, p2);
}

template <typename T> struct Planes;
template <> struct Planes<float> { static constexpr int ACT = 3, PAR = 3, WGT = 3; };
template <> struct Planes<bf16_t> { static constexpr int ACT = 1, PAR = 3, WGT = 2; };

// acc = c + sum over the kept terms a[i] * b[j] (i + j <= 2, smallest first)
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

// d index of element j (0..7) of lane group kq in k-step s
__device__ __forceinline__ constexpr int kd(int s, int kq, int j) { return 32 * s + (j < 4 ? 4 * kq + j : 16 + 4 * kq + (j - 4)); }

// k-fragment of one activation row, RAW (planes are made when consumed): the two 16-byte (fp32) / 8-byte (bf16) pieces at
// d = 32 s + 4 kq and 32 s + 16 + 4 kq;  off = byte offset of element (row, head column 0), OOB_OFF -> zeros
template <typename T> struct RawK;
template <> struct RawK<float> { float4 a, b; };
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <> struct RawK<bf16_t> { u32x2 a, b; };       // stays packed: the values ARE the one plane
// `off` = byte offset of the lane's first piece (row, head column + 4 kq) of k-step 0, OOB_OFF -> zeros
template <int D, typename T>
__device__ __forceinline__ void load_rawk(__amdgpu_buffer_rsrc_t r, unsigned off, int s, RawK<T>& x) {
    constexpr unsigned ES = Act<T>::ES;
    // pieces beyond D (D = 8 / 16: the second piece, and lane groups kq >= D / 4 of the first) must read zeros
    const int kq = (threadIdx.x >> 4) & 3;
    const bool ok0 = D == 16 || 32 * s + 4 * kq < D, ok1 = D == 16 || 32 * s + 16 + 4 * kq < D;
    const unsigned o0 = (off != OOB_OFF && ok0) ? off + 32 * s * ES : OOB_OFF;
    const unsigned o1 = (off != OOB_OFF && ok1) ? off + (D == 16 ? 4 : 32 * s + 16) * ES : OOB_OFF;   // D = 16: 8 adjacent d
    if constexpr (sizeof(T) == 2) {
        x.a = __builtin_amdgcn_raw_buffer_load_b64(r, o0, 0, 0);
        x.b = __builtin_amdgcn_raw_buffer_load_b64(r, o1, 0, 0);
    } else {
        x.a = buf_load4(r, o0);
        x.b = buf_load4(r, o1);
    }
}
template <typename T>
__device__ __forceinline__ void rawk_planes(const RawK<T>& x, bf16x8 (&pl)[Planes<T>::ACT]) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 q = {x.a.x, x.a.y, x.b.x, x.b.y};
        pl[0] = __builtin_bit_cast(bf16x8, q);
    } else {
        const f32x8 v = {x.a.x, x.a.y, x.a.z, x.a.w, x.b.x, x.b.y, x.b.z, x.b.w};
        split8<3>(v, pl);
    }
}

// parameter fragment: row m of a [M][D] fp32 matrix scaled by `scale`, k-step s, as 3 planes (zeros outside)
template <int D>
__device__ __forceinline__ void load_par8(const float* mat, int m, int M, int s, int kq, float scale, bf16x8 (&pl)[3]) {
    f32x8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = kd(s, kq, j);
        x[j] = (m < M && d < D) ? mat[(size_t)m * D + d] * scale : 0.f;
    }
    split8<3>(x, pl);
}


// ---- D = 16 (Darcy: 8 heads of 16 channels): a contraction over d fills only half of a 16x16x32 MFMA, so the other half of
// the k index carries a second plane pair — the six terms of the 3 x 3 split are THREE MFMAs (fp32 storage; two for bf16
// storage) instead of six half-empty ones, and every lane converts useful data.  k-slot (kq, j): plane pair member
// "lo" at d = 8 kq + j for kq < 2, member "hi" at d = 8 (kq - 2) + j for kq >= 2:
//   activations [x0|x1] [x0|x2] [x1|x0]  x  parameters [w0|w0] [w1|w0] [w1|w2]  =  w0x0 + w0x1, w1x0 + w0x2, w1x1 + w2x0.
// The parameter planes live UNSCALED in LDS ([plane][slice][16 d], shared by the heads of the workgroup; log2(e)/tau is
// applied inside exp2's fma), which also frees the 96 registers the M = 128 instantiations kept them in.
template <int MT, typename T> struct W16 {
    static constexpr int NA = Planes<T>::ACT, NC = NA == 3 ? 3 : 2, MP = 16 * MT, PIMG = MP * 32;
    static constexpr int BIAS = 3 * PIMG, BYTES = BIAS + MP * 16;      // + bias as the MFMA C operand (fp32)
    unsigned off[NC];
    const unsigned char* img;
    __device__ __forceinline__ void init(const unsigned char* base, int li, int kq) {
        img = base;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int pa = NA == 3 ? (c == 0 ? 0 : 1) : (c == 0 ? 0 : 2);
            const int pb = NA == 3 ? (c == 2 ? 2 : 0) : (c == 0 ? 1 : 2);
            off[c] = (unsigned)((kq < 2 ? pa : pb) * PIMG + li * 32 + (kq & 1) * 16);
        }
    }
    __device__ __forceinline__ f32x4 bias_n(int mt, int li) const {      // N-layout: lane's slice 16 mt + li, all 4 rows
        return *reinterpret_cast<const f32x4*>(img + BIAS + (16 * mt + li) * 16);
    }
    __device__ __forceinline__ f32x4 bias_t(int mt, int kq) const {      // T-layout: slices 16 mt + 4 kq .. + 3
        return *reinterpret_cast<const f32x4*>(img + BIAS + (16 * mt + 4 * kq) * 4);
    }
    __device__ __forceinline__ bf16x8 frag(int c, int mt) const {
        return *reinterpret_cast<const bf16x8*>(img + off[c] + mt * 512);
    }
    // every thread of the workgroup (before any wave leaves): split Ws [M][16] into the three plane images
    // splat: bias[m] four times (N-layout logits: column = slice) or once (T-layout: rows = slices 4 kq + r)
    static __device__ __forceinline__ void fill(unsigned char* base, const float* ws, const float* bs, int M, bool splat) {
        for (int i = threadIdx.x; i < MP; i += blockDim.x) {
            const float bv = i < M ? bs[i] : NEG_BIG;
            if (splat) *reinterpret_cast<f32x4*>(base + BIAS + i * 16) = (f32x4){bv, bv, bv, bv};
            else *reinterpret_cast<float*>(base + BIAS + i * 4) = bv;
        }
        for (int i = threadIdx.x; i < MP * 2; i += blockDim.x) {
            const int m = i >> 1, d0 = 8 * (i & 1);
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = m < M ? ws[(size_t)m * 16 + d0 + e] : 0.f;
            bf16x8 pl[3];
            split8<3>(v, pl);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(base + q * PIMG + m * 32 + d0 * 2) = pl[q];
        }
        __syncthreads();
    }
    // activation side of the K-packed terms
    static __device__ __forceinline__ void combos(const bf16x8 (&xpl)[NA], int kq, bf16x8 (&xc)[NC]) {
        const bool lo = kq < 2;
        if constexpr (NA == 3) {
            const u32x4 x0 = __builtin_bit_cast(u32x4, xpl[0]), x1 = __builtin_bit_cast(u32x4, xpl[1]),
                        x2 = __builtin_bit_cast(u32x4, xpl[2]);
            u32x4 c0, c1, c2;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0[i] = lo ? x0[i] : x1[i]; c1[i] = lo ? x0[i] : x2[i]; c2[i] = lo ? x1[i] : x0[i]; }
            xc[0] = __builtin_bit_cast(bf16x8, c0); xc[1] = __builtin_bit_cast(bf16x8, c1); xc[2] = __builtin_bit_cast(bf16x8, c2);
        } else {
            const u32x4 x0 = __builtin_bit_cast(u32x4, xpl[0]);
            u32x4 c1;
#pragma unroll
            for (int i = 0; i < 4; ++i) c1[i] = lo ? x0[i] : 0u;
            xc[0] = xpl[0]; xc[1] = __builtin_bit_cast(bf16x8, c1);      // [x0|x0] . [w0|w1], [x0|0] . [w2|-]
        }
    }
};

template <int D> struct BCfg {
    static constexpr int KST = (D + 31) / 32;      // 32-wide k-steps of a contraction over d
    static constexpr int DT = (D + 15) / 16;       // 16-wide tiles over d
};

// Work decomposition of the v3 kernels: ONE WAVE owns a (batch, head, point chunk) unit and a workgroup is the `hpw` =
// min(heads, 8) heads of the same (batch, chunk): its waves read the neighbouring 128-byte head segments of the SAME
// activation rows at the same time, so HBM sees whole rows (loads-only timing at the bench shape: 69 us with one
// workgroup per (batch, head, chunk), rows 2 KB apart — 4 TB/s).  No LDS, no barrier, no cross-wave reduction: the wave
// writes its own partial-sum record `bid`.  grid = B * nchunk * ceil(heads / hpw), block = 64 * hpw.
__device__ __forceinline__ bool slice3_decode(int B, int heads, int nchunk, int& b, int& hh, int& chunk, int& bid) {
    const int hpw = (int)(blockDim.x >> 6), ngroups = (heads + hpw - 1) / hpw;
    const int L = (int)blockIdx.x, hg = L % ngroups;
    chunk = (L / ngroups) % nchunk;
    b = L / (ngroups * nchunk);
    hh = hg * hpw + (int)(threadIdx.x >> 6);
    bid = (b * heads + hh) * nchunk + chunk;
    return hh < heads && b < B;
}

struct Scatter3Params {
    const void* xm; long long ldx;
    const void* v; long long ldv;
    const float* ws; const float* bs; const float* temperature;
    float* spart; float* npart;
    int B, N, heads, M, nchunk, ppc;
    unsigned x_bytes, v_bytes;
    int clamp, xcd_map;
};

}  // namespace

// S_partial[m][d] = sum_{n in chunk} W[n][m] * V[n][d];  n_partial[m] = sum_n W[n][m]
// N-layout: the logits tile of 16 points has rows = points, columns = slices; two tiles (32 points) give lane (slice li,
// kq) the 8 weights of points 4kq..4kq+3 and 16+4kq..16+4kq+3 = the A operand (rows = slices, k = points) of S += W^T V,
// and the SAME lane holds the 8 normalisers 1/Z of exactly these points, which scale its 8 rows of V (B operand).
template <int D, int MT, typename T>
__global__ __launch_bounds__((MT <= 4 || D == 16) ? 512 : 256, (MT <= 4 || D == 16) ? 2 : 1) void scatter3_kernel(const Scatter3Params p) {
    constexpr int KST = BCfg<D>::KST, DT = BCfg<D>::DT;
    constexpr int NA = Planes<T>::ACT, NW = Planes<T>::WGT;
    constexpr unsigned ES = Act<T>::ES;
    constexpr bool K16 = D == 16;
    using W16T = W16<MT, T>;
    __shared__ __attribute__((aligned(16))) unsigned char wimg[K16 ? W16T::BYTES : 16];
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    if constexpr (K16) W16T::fill(wimg, p.ws, p.bs, p.M, true);
    int b, hh, chunk, bid;
    if (!slice3_decode(p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float scale = LOG2E / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);
    const float esc = K16 ? scale : 1.0f;      // generic: Ws and bs are pre-scaled;  D = 16: applied inside exp2's fma

    bf16x8 wsp[K16 ? 1 : MT][KST][3];      // B operand of the logits: column = slice 16mt+li, k = d
    W16T w16;
    if constexpr (K16) w16.init(wimg, li, kq);
    f32x4 bias[K16 ? 1 : MT];    // C operand of the first MFMA of a logits chain (D = 16: read from LDS)
    if constexpr (!K16) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * mt + li;
#pragma unroll
            for (int s = 0; s < KST; ++s) load_par8<D>(p.ws, m, p.M, s, kq, scale, wsp[mt][s]);
            const float bv = m < p.M ? p.bs[m] * scale : NEG_BIG;
            bias[mt] = (f32x4){bv, bv, bv, bv};
        }
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
    const unsigned row0 = (unsigned)b * (unsigned)p.N + (unsigned)p_begin;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_v(p.xm, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc_v(p.v, p.v_bytes);
    const unsigned ldxb = (unsigned)p.ldx * ES, ldvb = (unsigned)p.ldv * ES, hcol = (unsigned)(hh * D) * ES;
    const bool want_n = p.npart != nullptr;

    // Per-lane byte offsets of group 0, advanced by 32 rows per group.  X: lane (point li of tile t, kq) loads the two
    // 16-byte pieces of its k-fragment; V: lane (channel li, kq) loads the 8 rows 4kq+e (e < 4) and 16+4kq+(e-4) of the
    // group, one dword each: the row offsets e * ldv are wave-uniform (scalar offsets of the loads).
    unsigned xo[2], vo = (row0 + 4 * kq) * ldvb + hcol + li * ES;
#pragma unroll
    for (int t = 0; t < 2; ++t) xo[t] = (row0 + 16 * t + li) * ldxb + hcol + (K16 ? 8 * (kq & 1) : 4 * kq) * ES;
    const unsigned xstep = 32u * ldxb, vstep = 32u * ldvb;
    unsigned vrow[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) vrow[e] = (unsigned)(e < 4 ? e : 12 + e) * ldvb;

    RawK<T> xr[2][KST];
    float fv[DT][8];
    // TAIL = false: all 32 points of the group exist (no masks anywhere); TAIL = true: the ragged last group of a chunk
    auto load_x = [&](auto tail, int n_left) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned off_ = (!decltype(tail)::value || 16 * t + li < n_left) ? xo[t] : OOB_OFF;
#pragma unroll
            for (int s = 0; s < KST; ++s) load_rawk<D, T>(rx, off_, s, xr[t][s]);
        }
    };
    auto load_v = [&](auto tail, int n_left) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok_ = !decltype(tail)::value || 4 * kq + (e < 4 ? e : 12 + e) < n_left;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bool okd_ = 16 * dt + li < D;            // folds to true unless D < 16
                fv[dt][e] = Act<T>::bld1s(rv, (ok_ && okd_) ? vo + 16 * dt * ES : OOB_OFF, vrow[e]);
            }
        }
    };
    auto group = [&](auto tail, int n_left, int n_next) {        // n_left: points left in the chunk from this group on
        constexpr bool TAIL = decltype(tail)::value;
        f32x4 w[2][MT];          // logits, then unnormalised weights exp2(z - max)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                bf16x8 xpl[NA];
                rawk_planes<T>(xr[t][s], xpl);
                if constexpr (K16) {
                    bf16x8 xc[W16T::NC];
                    W16T::combos(xpl, kq, xc);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x4 a = w16.bias_n(mt, li);
#pragma unroll
                        for (int c = W16T::NC - 1; c >= 0; --c) a = mfma_bf(xc[c], w16.frag(c, mt), a);
                        w[t][mt] = a;
                    }
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) w[t][mt] = mfma_terms<NA, 3>(xpl, wsp[mt][s], s == 0 ? bias[mt] : w[t][mt]);
                }
            }
        xo[0] += xstep; xo[1] += xstep;
        if (n_next >= 32) load_x(std::false_type{}, n_next);
        else if (n_next > 0) load_x(std::true_type{}, n_next);
        f32x4 inv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mx[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mx[r] = w[t][0][r];
#pragma unroll
                for (int mt = 1; mt < MT; ++mt) mx[r] = fmaxf(mx[r], w[t][mt][r]);
            }
            row16_max4(mx[0], mx[1], mx[2], mx[3]);
            float sm[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sm[r] = 0.f;
                const float nm = -mx[r] * esc;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float e = K16 ? ex2(fmaf(w[t][mt][r], esc, nm)) : ex2(w[t][mt][r] - mx[r]);
                    w[t][mt][r] = e;
                    sm[r] += e;
                }
            }
            row16_sum4(sm[0], sm[1], sm[2], sm[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float iv = __builtin_amdgcn_rcpf(sm[r]);
                inv[t][r] = (!TAIL || 16 * t + 4 * kq + r < n_left) ? iv : 0.f;
            }
        }
        if (want_n) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nacc[mt] = fmaf(w[t][mt][r], inv[t][r], nacc[mt]);
        }
        // S += W^T V over the 32 points of the group
        bf16x8 fp[DT][NA];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = NA == 1 ? fv[dt][e] : fv[dt][e] * inv[e >> 2][e & 3];
            split8<NA>(f, fp[dt]);          // bf16 storage: the values ARE one plane (exact)
        }
        vo += vstep;
        if (n_next >= 32) load_v(std::false_type{}, n_next);
        else if (n_next > 0) load_v(std::true_type{}, n_next);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x8 wv;
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[e] = NA == 1 ? w[e >> 2][mt][e & 3] * inv[e >> 2][e & 3] : w[e >> 2][mt][e & 3];
            bf16x8 wp[NW];
            split8<NW>(wv, wp);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) sacc[mt][dt] = mfma_terms<NW, NA>(wp, fp[dt], sacc[mt][dt]);
        }
    };

    int n_left = p_end - p_begin;
    if (n_left >= 32) { load_x(std::false_type{}, n_left); load_v(std::false_type{}, n_left); }
    else if (n_left > 0) { load_x(std::true_type{}, n_left); load_v(std::true_type{}, n_left); }
    for (; n_left >= 32; n_left -= 32) group(std::false_type{}, n_left, n_left - 32);
    if (n_left > 0) group(std::true_type{}, n_left, 0);

    // the wave's own record: S [M][D] and n [M]
    float* so = p.spart + (size_t)bid * p.M * D;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mt + 4 * kq + r, d = 16 * dt + li;
                if (m < p.M && d < D) so[m * D + d] = sacc[mt][dt][r];
            }
    if (want_n) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float v = kq_sum(nacc[mt]);
            if (kq == 0 && 16 * mt + li < p.M) p.npart[(size_t)bid * p.M + 16 * mt + li] = v;
        }
    }
}

namespace {
struct Deslice3Params {
    const void* xm; long long ldx;
    const float* o;
    const float* ws; const float* bs; const float* temperature;
    void* y; long long ldy;
    int B, N, heads, M, nchunk, ppc;
    unsigned x_bytes, y_bytes;
    int clamp;
};
}  // namespace

// Y[n][h*D+d] = sum_m W[n][m] * O[m][d]
// T-layout: the logits tile of 16 points has rows = slices, columns = points: lane (point li, kq) holds the weights of its
// point for the slices 16mt + 4kq + r, i.e. the B operand (k = slices, columns = points) of Y^T = O^T W^T; softmax is
// in-lane plus the two row swaps, and the normalisation 1/Z multiplies the D outputs of the point.
template <int D, int MT, typename T>
__global__ __launch_bounds__((MT <= 4 || D == 16) ? 512 : 256, (MT <= 4 || D == 16) ? 2 : 1) void deslice3_kernel(const Deslice3Params p) {
    constexpr int KST = BCfg<D>::KST, DT = BCfg<D>::DT;
    constexpr int NA = Planes<T>::ACT, NW = Planes<T>::WGT;
    constexpr int MU = (MT + 1) / 2;                 // 32-slice k-steps of the contraction over m
    constexpr unsigned ES = Act<T>::ES;
    constexpr bool K16 = D == 16;
    using W16T = W16<MT, T>;
    __shared__ __attribute__((aligned(16))) unsigned char wimg[K16 ? W16T::BYTES : 16];
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    if constexpr (K16) W16T::fill(wimg, p.ws, p.bs, p.M, false);
    int b, hh, chunk, bid;
    if (!slice3_decode(p.B, p.heads, p.nchunk, b, hh, chunk, bid)) return;
    const float scale = LOG2E / (p.clamp ? clamp_tau(p.temperature[hh]) : p.temperature[hh]);
    const float esc = K16 ? scale : 1.0f;

    bf16x8 wsp[K16 ? 1 : MT][KST][3];      // A operand of the transposed logits: row = slice 16mt+li, k = d
    W16T w16;
    if constexpr (K16) w16.init(wimg, li, kq);
    f32x4 bias[K16 ? 1 : MT];    // C operand: rows = slices 16mt + 4kq + r (D = 16: read from LDS)
    bf16x8 op[MU][DT][3];        // A operand of Y^T = O^T W^T: row = channel 16dt+li, k = slices of k-step u
    const float* ob = p.o + (size_t)(b * p.heads + hh) * p.M * D;
    if constexpr (!K16) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int s = 0; s < KST; ++s) load_par8<D>(p.ws, 16 * mt + li, p.M, s, kq, scale, wsp[mt][s]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mr = 16 * mt + 4 * kq + r;
                bias[mt][r] = mr < p.M ? p.bs[mr] * scale : NEG_BIG;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x8 x;
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
    const unsigned row0 = (unsigned)b * (unsigned)p.N + (unsigned)p_begin;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_v(p.xm, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc_v(p.y, p.y_bytes);
    const unsigned ldxb = (unsigned)p.ldx * ES, ldyb = (unsigned)p.ldy * ES, hcol = (unsigned)(hh * D) * ES;
    unsigned xo[2], yo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        xo[t] = (row0 + 16 * t + li) * ldxb + hcol + (K16 ? 8 * (kq & 1) : 4 * kq) * ES;
        yo[t] = (row0 + 16 * t + li) * ldyb + hcol + 4 * kq * ES;
    }
    const unsigned xstep = 32u * ldxb, ystep = 32u * ldyb;

    RawK<T> xr[2][KST];
    auto load_x = [&](auto tail, int n_left) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned off_ = (!decltype(tail)::value || 16 * t + li < n_left) ? xo[t] : OOB_OFF;
#pragma unroll
            for (int s = 0; s < KST; ++s) load_rawk<D, T>(rx, off_, s, xr[t][s]);
        }
    };
    auto group = [&](auto tail, int n_left, int n_next) {
        constexpr bool TAIL = decltype(tail)::value;
        f32x4 w[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                bf16x8 xpl[NA];
                rawk_planes<T>(xr[t][s], xpl);
                if constexpr (K16) {
                    bf16x8 xc[W16T::NC];
                    W16T::combos(xpl, kq, xc);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x4 a = w16.bias_t(mt, kq);
#pragma unroll
                        for (int c = W16T::NC - 1; c >= 0; --c) a = mfma_bf(w16.frag(c, mt), xc[c], a);
                        w[t][mt] = a;
                    }
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) w[t][mt] = mfma_terms<3, NA>(wsp[mt][s], xpl, s == 0 ? bias[mt] : w[t][mt]);
                }
            }
        xo[0] += xstep; xo[1] += xstep;
        if (n_next >= 32) load_x(std::false_type{}, n_next);
        else if (n_next > 0) load_x(std::true_type{}, n_next);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mx = w[t][0][0];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, w[t][mt][r]);
            mx = kq_max(mx);
            float sm = 0.f;
            const float nm = -mx * esc;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = K16 ? ex2(fmaf(w[t][mt][r], esc, nm)) : ex2(w[t][mt][r] - mx);
                    w[t][mt][r] = e;
                    sm += e;
                }
            sm = kq_sum(sm);
            const float inv = __builtin_amdgcn_rcpf(sm);
            f32x4 yacc[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) yacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                f32x8 wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wv[e] = w[t][2 * u][e];
                    wv[4 + e] = (2 * u + 1 < MT) ? w[t][(2 * u + 1 < MT) ? 2 * u + 1 : 0][e] : 0.f;
                }
                bf16x8 wp[NW];
                split8<NW>(wv, wp);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) yacc[dt] = mfma_terms<3, NW>(op[u][dt], wp, yacc[dt]);
            }
            const bool pv = !TAIL || 16 * t + li < n_left;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bool okd = 16 * dt + 4 * kq < D;
                Act<T>::bst4(ry, (pv && okd) ? yo[t] + 16 * dt * ES : OOB_OFF,
                             make_float4(yacc[dt][0] * inv, yacc[dt][1] * inv, yacc[dt][2] * inv, yacc[dt][3] * inv));
            }
        }
        yo[0] += ystep; yo[1] += ystep;
#ifndef S3_SGB
#define S3_SGB 4      // VALU instructions the scheduler places behind each MFMA of the group (measured 0.082 -> 0.075 ms)
#endif
#if S3_SGB > 0
#pragma unroll
        for (int i = 0; i < 96; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, S3_SGB, 0); }
#endif
    };
    int n_left = p_end - p_begin;
    if (n_left >= 32) load_x(std::false_type{}, n_left);
    else if (n_left > 0) load_x(std::true_type{}, n_left);
    for (; n_left >= 32; n_left -= 32) group(std::false_type{}, n_left, n_left - 32);
    if (n_left > 0) group(std::true_type{}, n_left, 0);
}

// ---------------------------------------------------------------------------------------------- host side
// heads per workgroup: up to 8 waves (4 for the 256-register M = 128 instantiations) that read neighbouring head segments
// of the same rows; small problems (rollout at batch 1) keep fewer heads per workgroup so that every CU gets a wave
static int slice3_hpw(int B, int heads, int nchunk, int mt, int D) {
    int hpw = (mt <= 4 || D == 16) ? 8 : 4;
    if (hpw > heads) hpw = heads;
    while (hpw > 1 && (long long)B * nchunk * ((heads + hpw - 1) / hpw) < 256) hpw >>= 1;
    return hpw;
}
#define S3_DISPATCH_MT(D_, CALL)                                 \
    switch (mt) {                                                \
        case 1: CALL(D_, 1); break;                              \
        case 2: CALL(D_, 2); break;                              \
        case 4: CALL(D_, 4); break;                              \
        case 8: CALL(D_, 8); break;                              \
        default: return PA2D_ERR_UNSUPPORTED;                    \
    }
#define S3_DISPATCH_D(CALL)                                      \
    switch (D) {                                                 \
        case 8: S3_DISPATCH_MT(8, CALL) break;                   \
        case 16: S3_DISPATCH_MT(16, CALL) break;                 \
        case 32: S3_DISPATCH_MT(32, CALL) break;                 \
        case 64: S3_DISPATCH_MT(64, CALL) break;                 \
        default: return PA2D_ERR_UNSUPPORTED;                    \
    }

extern "C" __attribute__((visibility("hidden"))) int pa2d_launch_scatter3(const void* xm, long long ldx, const void* v, long long ldv, const float* ws, const float* bs,
                           const float* temperature, float* spart, float* npart, int B, int N, int heads, int D, int M,
                           int mt, int nchunk, int ppc, unsigned x_bytes, unsigned v_bytes, int clamp, int xcd_map, bool bf,
                           hipStream_t st) {
    Scatter3Params p;
    p.xm = xm; p.ldx = ldx; p.v = v; p.ldv = ldv; p.ws = ws; p.bs = bs; p.temperature = temperature;
    p.spart = spart; p.npart = npart; p.B = B; p.N = N; p.heads = heads; p.M = M; p.nchunk = nchunk; p.ppc = ppc;
    p.x_bytes = x_bytes; p.v_bytes = v_bytes; p.clamp = clamp; p.xcd_map = xcd_map;
    const int hpw = slice3_hpw(B, heads, nchunk, mt, D);
    const dim3 grid(B * nchunk * ((heads + hpw - 1) / hpw)), block(64 * hpw);
#define CALL_S3(D_, MT_)                                                                                          \
    if (bf) hipLaunchKernelGGL((scatter3_kernel<D_, MT_, bf16_t>), grid, block, 0, st, p);                        \
    else hipLaunchKernelGGL((scatter3_kernel<D_, MT_, float>), grid, block, 0, st, p)
    S3_DISPATCH_D(CALL_S3)
    return PA2D_OK;
}

extern "C" __attribute__((visibility("hidden"))) int pa2d_launch_deslice3(const void* xm, long long ldx, const float* o, const float* ws, const float* bs,
                           const float* temperature, void* y, long long ldy, int B, int N, int heads, int D, int M, int mt,
                           int nchunk, int ppc, unsigned x_bytes, unsigned y_bytes, int clamp, bool bf, hipStream_t st) {
    Deslice3Params p;
    p.xm = xm; p.ldx = ldx; p.o = o; p.ws = ws; p.bs = bs; p.temperature = temperature; p.y = y; p.ldy = ldy;
    p.B = B; p.N = N; p.heads = heads; p.M = M; p.nchunk = nchunk; p.ppc = ppc; p.x_bytes = x_bytes; p.y_bytes = y_bytes;
    p.clamp = clamp;
    const int hpw = slice3_hpw(B, heads, nchunk, mt, D);
    const dim3 grid(B * nchunk * ((heads + hpw - 1) / hpw)), block(64 * hpw);
#define CALL_D3(D_, MT_)                                                                                          \
    if (bf) hipLaunchKernelGGL((deslice3_kernel<D_, MT_, bf16_t>), grid, block, 0, st, p);                        \
    else hipLaunchKernelGGL((deslice3_kernel<D_, MT_, float>), grid, block, 0, st, p)
    S3_DISPATCH_D(CALL_D3)
    return PA2D_OK;
}
