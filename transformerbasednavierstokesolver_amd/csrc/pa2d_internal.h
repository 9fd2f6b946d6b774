// Internal helpers shared by the HIP translation units of libpa2d (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PA2D_OK 0
#define PA2D_ERR_ARG 1001
#define PA2D_ERR_UNSUPPORTED 1002
#define PA2D_ERR_WORKSPACE 1003

#define PA2D_CHECK_LAUNCH()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// activation ids (ACTIVATION table of the reference, model/Transolver_Structured_Mesh_2D.py:9-10)
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_RELU = 4, ACT_SOFTPLUS = 5,
       ACT_ELU = 6, ACT_SILU = 7 };

// Exact-erf GELU (nn.GELU() default) without ocml erff: Phi(x) = 0.5*(1+erf(x/sqrt2)) from the
// Abramowitz-Stegun 7.1.26 rational form q(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1/(1+p z),
// erfc(z) = q(z) + eps, |eps| <= 1.5e-7 (fp32 level), evaluated branch-free on z = |x|/sqrt2.
// e_out returns exp(-x^2/2) so the derivative can reuse it for the density term.
// (contraction off: hipcc's default -ffp-contract=fast fuses `1 - hq` / `x * cdf` differently depending on what else the
//  inlining epilogue computes from the same values, and the training forward — which also saves gelu' — must give the
//  same bits as the inference forward)
__device__ __forceinline__ float normal_cdf(float x, float& e_out) {
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    e_out = e;
    const float hq = 0.5f * poly * t * e;          // 0.5 * erfc(z)
    return x < 0.f ? hq : 1.0f - hq;
}
__device__ __forceinline__ float gelu_exact(float x) {
#pragma clang fp contract(off)
    float e;
    return x * normal_cdf(x, e);
}
__device__ __forceinline__ float dgelu_exact(float x) {
#pragma clang fp contract(off)
    float e;
    const float cdf = normal_cdf(x, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

__device__ __forceinline__ float act_fwd(int id, float x) {
    switch (id) {
        case ACT_GELU: return gelu_exact(x);
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        case ACT_SILU: return x / (1.0f + expf(-x));
        default: return x;
    }
}

// d act(x) / dx evaluated at the pre-activation x
__device__ __forceinline__ float act_bwd(int id, float x) {
    switch (id) {
        case ACT_GELU: return dgelu_exact(x);
        case ACT_TANH: { const float t = tanhf(x); return 1.0f - t * t; }
        case ACT_SIGMOID: { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f - s); }
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_SOFTPLUS: return x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
        case ACT_ELU: return x > 0.f ? 1.f : expf(x);
        case ACT_SILU: { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f + x * (1.0f - s)); }
        default: return 1.f;
    }
}

// 64-lane butterfly reductions (wave = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Raw buffer loads: lanes whose byte offset is >= num_records return 0 from the hardware range
// check, so masked (padding / out-of-tile) elements need neither branches nor selects and the loads
// stay asynchronous until the compiler's vmcnt wait in front of the LDS store.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define OOB_OFF 0xFFFFFFFFu

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}


__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, 0, 0);
}

__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
    u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, 0);
}

// Kernel-selection overrides (A/B timing, and the parity tests that force one of two equivalent kernels): read from the
// environment ONCE per process (first use) into this table; pa2d_reload_env() re-reads it (tests only).  They choose
// between kernels that implement the same stage to the same tolerance — no numerical mode depends on them.
#define PA2D_ACT_SAVE_DERIVATIVE_BIT 0x100      // include/pa2d.h: PA2D_ACT_SAVE_DERIVATIVE

struct Pa2dEnv {
    int conv_halo;        // PA2D_CONV_HALO=off|auto|force : 0 / 1 / 2
    int mc_big_off;       // PA2D_MC_BIG=off
    int kc_bk16;          // PA2D_KC_BK=16
    int mc_bk;            // PA2D_MC_BK=32 (else 16)
    int mc_splits;        // PA2D_MC_SPLITS=n (0 = automatic)
    int mcb_splits;       // PA2D_MCB_SPLITS=n (0 = automatic)
    int lin_dw_split;     // PA2D_LIN_DW_SPLIT=off -> 0
    int lin_panel;        // PA2D_LIN_PANEL=off -> 0
    int conv_mfma16;      // PA2D_CONV_MFMA=32 -> 0 (halo conv consumers back on v_mfma_f32_32x32x16_bf16)
    int lin_small_split;  // PA2D_LIN_SMALL_SPLIT=off -> 0 (small split-engine linears back on the exact-fp32 kernels)
    int lin_rowpanel;     // PA2D_LIN_ROWPANEL=off -> 0 (row-stationary linears back on the panel / per-tile kernels)
    int split_big;        // PA2D_SPLIT_BIG=0 -> 0
    int slice_map;        // PA2D_SLICE_MAP=legacy -> 0
    int default_engine;   // PA2D_GEMM=f32|split|bf16 : what pa2d_default_engine() returns
};
const Pa2dEnv& pa2d_env();

// empty problems (batch 0): maps are no-ops, reductions produce exact zeros
static inline int pa2d_zero(void* p, size_t bytes, hipStream_t st) {
    if (!p || !bytes) return PA2D_OK;
    return (int)hipMemsetAsync(p, 0, bytes, st);
}
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// Deterministic second pass of every partial-sum reduction: out = sum over the nslab records of `count` floats each,
// scattered over up to 4 destination segments (record index range [begin[i], begin[i+1]) -> dst[i]); accumulate != 0
// adds to the destination instead of overwriting it (gradient accumulation into the caller's bucket).
struct ReduceSegs { int nseg; long long begin[5]; float* dst[4]; };
int pa2d_launch_reduce_segs(const float* slab, int nslab, long long count, const ReduceSegs& segs, int accumulate,
                            hipStream_t st);

// ---- activation element types: fp32, or bf16 STORAGE (the *_bf16 entry points: activations, saved tensors and
// inter-kernel gradients live in HBM as bf16; parameters, statistics, partial sums and accumulators stay fp32).
// Everything is computed in fp32 registers; these helpers are the only place where the storage type shows.
typedef __bf16 bf16_t;
template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr unsigned ES = 4;
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    // buffer-descriptor forms (byte offset; OOB_OFF lanes read 0 / drop the store)
    static __device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) { return buf_load4(r, off); }
    static __device__ __forceinline__ float2 bld2(__amdgpu_buffer_rsrc_t r, unsigned off) { return buf_load2(r, off); }
    static __device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t r, unsigned off) { return buf_load1(r, off); }
    // with a wave-uniform scalar offset added to the address (not to the range check: mask through `off`)
    static __device__ __forceinline__ float bld1s(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
    }
    static __device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) { buf_store4(r, off, v); }
    static __device__ __forceinline__ void bst1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) { buf_store1(r, off, v); }
};
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float x) {        // round to nearest even (hardware cvt)
    const bf16_t h = (bf16_t)x;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
}
template <> struct Act<bf16_t> {
    static constexpr unsigned ES = 2;
    static __device__ __forceinline__ float4 unpack4(uint2 q) {
        return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16),
                           __uint_as_float(q.y & 0xFFFF0000u));
    }
    static __device__ __forceinline__ float4 ld4(const bf16_t* p) { return unpack4(*reinterpret_cast<const uint2*>(p)); }
    static __device__ __forceinline__ void st4(bf16_t* p, float4 v) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
    static __device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
    static __device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        return unpack4(make_uint2(q.x, q.y));
    }
    static __device__ __forceinline__ float2 bld2(__amdgpu_buffer_rsrc_t r, unsigned off) {
        const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
        return make_float2(__uint_as_float(q << 16), __uint_as_float(q & 0xFFFF0000u));
    }
    static __device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
    }
    static __device__ __forceinline__ float bld1s(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(r, off, soff, 0));
    }
    static __device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ q = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        __builtin_amdgcn_raw_buffer_store_b64(q, r, off, 0, 0);
    }
    static __device__ __forceinline__ void bst1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
        __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16_bits(v), r, off, 0, 0);
    }
};
// buffer descriptor over any element type
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_v(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// Workgroup -> (batch, head, point chunk) of the slice-stage kernels.  xcd_map = 1: workgroups are numbered so that each
// XCD (blockIdx & 7 under round-robin dispatch) owns a CONTIGUOUS range with the head index fastest: the `heads`
// workgroups that read the 128-byte head segments of the SAME activation rows run next to each other behind one L2, so
// HBM sees whole rows instead of eight interleaved strided streams.  xcd_map = 0: legacy order (chunk fastest).
// bid = index of the workgroup's partial-sum record, independent of the map.  Launch with slice_grid(total) blocks.
__device__ __forceinline__ bool slice_decode(int xcd_map, int B, int heads, int nchunk, int& b, int& hh, int& chunk, int& bid) {
    const int total = B * heads * nchunk;
    int L;
    if (xcd_map) {
        const int per = (total + 7) >> 3;
        if ((int)(blockIdx.x >> 3) >= per) return false;
        L = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if (L >= total) return false;
        hh = L % heads;
        chunk = (L / heads) % nchunk;
        b = L / (heads * nchunk);
    } else {
        L = (int)blockIdx.x;
        if (L >= total) return false;
        chunk = L % nchunk;
        hh = (L / nchunk) % heads;
        b = L / (nchunk * heads);
    }
    bid = (b * heads + hh) * nchunk + chunk;
    return true;
}
static inline int slice_grid(int total) { return ((total + 7) / 8) * 8; }
