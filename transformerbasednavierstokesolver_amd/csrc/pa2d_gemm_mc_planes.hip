// Conv weight gradient on the bf16 matrix cores from PRE-SPLIT operand planes (bf16 engines, gfx950 / CDNA4):
//     dW[i][j = tap*Cin + ci] = sum_m dOut[m][i] * X[m shifted by tap][ci]
// Both operands are contracted over their ROW index m, so the MFMA operands (8 consecutive k = m per lane) are
// k-strided in memory.  The planes made by split_planes_kernel ([row][32-channel chunk][plane][32] bf16) are staged
// ROW-MAJOR into LDS with plain 16-byte copies (no conversion, half the bytes of an fp32 tile per plane) and the
// operands are fetched with gfx950's transposed LDS read ds_read_b64_tr_b16 (per 16-lane group a 4-row x 16-column
// block, delivered column-major: lane i gets column i, row e in element e — mapping checked on the device by
// tools/probes/tr_read_probe.hip).  LDS image per plane: [16 rows][128 columns] bf16 = 256-byte rows with the 16-byte
// chunk index XOR-swizzled by ((row&3)<<2 | (row>>2)&3), which keeps the transposed reads conflict-free.
// NT = 3: six MFMA terms of the exact hi+mid+lo split (fp32 accuracy); NT = 1: bf16 compute.
#include "pa2d_gemm_common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct MCPlanesParams {
    const void* PA; int chA;      // dOut planes, chA = 2C/32 chunks per row
    const void* PB; int chB;      // X planes, chB = Cin/32
    float* slab;
    int Mi, Nj, Mk, chunks_per_split, splits, H, W, Cin;
    unsigned a_bytes, b_bytes;
    int taps;                     // 9: columns are (tap, input channel) of a 3x3 conv; 1: plain dW = A^T . B (big kernel only)
    long long lda, ldb;           // F32SRC (128 x 128 kernel): PA / PB are fp32 row-major matrices [Mk][lda], [Mk][ldb]
    float* colsum;                // F32SRC, optional: [splits][Mi] per-split column sums of A (= a linear layer's bias gradient)
};

__device__ __forceinline__ unsigned img_off(int row, int ch) {       // byte offset inside one 4 KB plane image
    return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// KS = 16-row K-steps per LDS stage (one barrier per stage): 1 for NT = 3 (24 MFMAs per wave and barrier, 48 KB of LDS),
// 4 for NT = 1 (16 MFMAs per barrier instead of 4, 64 KB).
// F32SRC: plain dW = A^T . B of a linear layer straight from the fp32 operands (no tap shift): each thread loads the 8
// floats of its (row, 8-column piece) of both operands, splits them into the NT bf16 planes in registers and writes the
// same LDS plane images; the column sums of A (the bias gradient) are accumulated from the registers of column tile 0.
template <int NT, int KS, bool F32SRC = false>
__global__ __launch_bounds__(256, 2) void gemm_mc_planes_kernel(const MCPlanesParams p) {
    constexpr int PIMG = 4096;                       // one plane image of one K-step: 16 rows x 256 B
    constexpr int STAGE = 2 * NT * KS * PIMG;        // A planes then B planes, KS sub-images each
    constexpr int NP = 2 * NT * KS;                  // 16-byte pieces per thread and stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_i = (p.Mi + 127) / 128, tiles_j = (p.Nj + 127) / 128;
    int tj, ti, split;
    {
        const int tiles = tiles_i * tiles_j;
        if ((p.splits & 7) == 0) {                   // XCD-contiguous split ranges, as in gemm_mc_kernel
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = p.splits >> 3;
            split = xcd * q + slot / tiles;
            const int t = slot % tiles;
            tj = t % tiles_j;
            ti = t / tiles_j;
        } else {
            tj = blockIdx.x % tiles_j;
            ti = (blockIdx.x / tiles_j) % tiles_i;
            split = blockIdx.x / tiles;
        }
    }
    const int wm = wave >> 1, wn = wave & 1;
    const int total_chunks = (p.Mk + 15) / 16;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(total_chunks, c_begin + p.chunks_per_split);

    // the 128 columns of this tile are one tap and 4 consecutive 32-channel chunks of X
    const int j0 = tj * 128, tap = j0 / p.Cin, cb0 = (j0 - tap * p.Cin) >> 5, ca0 = (ti * 128) >> 5;
    const int tap_dy = tap / 3 - 1, tap_dx = tap - (tap / 3) * 3 - 1, tap_shift = tap_dy * p.W + tap_dx;
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.PA), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.PB), 0, p.b_bytes, 0x00020000);

    // staging assignment: piece id q = tid + 256*s in [0, 512*NT): operand, row, chunk, plane, 16-byte piece
    int s_row[NP];
    unsigned s_goff[NP], s_lds[NP];
    bool s_isb[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const int q = tid + 256 * s;
        const int op = q / (256 * NT * KS), r = q - op * (256 * NT * KS);
        const int row = r / (16 * NT), rr = r - row * (16 * NT);          // row in [0, 16*KS)
        const int c = rr / (4 * NT), pp = rr - c * (4 * NT);
        const int plane = pp >> 2, piece = pp & 3;
        s_row[s] = row;
        s_isb[s] = op != 0;
        s_goff[s] = (unsigned)((((op ? cb0 : ca0) + c) * NT + plane) * 64 + piece * 16);
        s_lds[s] = (unsigned)(((op * NT + plane) * KS + (row >> 4)) * PIMG) + img_off(row & 15, c * 4 + piece);
    }
    u32x4 rg[F32SRC ? 4 * KS : NP];
    // F32SRC staging: thread = (row, 8-column piece) of both operands
    const int frow = tid >> 4, fc8 = tid & 15;
    const unsigned flds = img_off(frow, fc8);
    const bool fok_a = ti * 128 + fc8 * 8 < p.Mi, fok_b = tj * 128 + fc8 * 8 < p.Nj;
    const unsigned fcol_a = (unsigned)(ti * 128 + fc8 * 8) * 4u, fcol_b = (unsigned)(tj * 128 + fc8 * 8) * 4u;
    const unsigned fpitch_a = (unsigned)p.lda * 4u, fpitch_b = (unsigned)p.ldb * 4u;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#define MP_LOAD(c_)                                                                                       \
    if constexpr (F32SRC) {                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                               \
            const int m_ = ((c_) + ks) * 16 + frow;                                                       \
            const bool okr_ = m_ < p.Mk && (c_) + ks < c_end;                                             \
            const unsigned oa_ = (okr_ && fok_a) ? (unsigned)m_ * fpitch_a + fcol_a : OOB_OFF;            \
            const unsigned ob_ = (okr_ && fok_b) ? (unsigned)m_ * fpitch_b + fcol_b : OOB_OFF;            \
            rg[ks * 4 + 0] = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, oa_, 0, 0);                   \
            rg[ks * 4 + 1] = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, oa_ == OOB_OFF ? OOB_OFF : oa_ + 16u, 0, 0); \
            rg[ks * 4 + 2] = __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, ob_, 0, 0);                   \
            rg[ks * 4 + 3] = __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, ob_ == OOB_OFF ? OOB_OFF : ob_ + 16u, 0, 0); \
        }                                                                                                 \
    } else {                                                                                              \
        const int m0_ = (c_) * 16;                                                                        \
        _Pragma("unroll") for (int s = 0; s < NP; ++s) {                                                  \
            const int m_ = m0_ + s_row[s];                                                                \
            bool ok_ = m_ < p.Mk && (KS == 1 || (c_) + (s_row[s] >> 4) < c_end);   /* next split's rows stay out */ \
            int mm_ = m_;                                                                                 \
            if (s_isb[s]) {                                                                               \
                const int n_ = m_ % HW;                                                                   \
                const int y_ = n_ / p.W, x_ = n_ - y_ * p.W;                                              \
                ok_ = ok_ && (unsigned)(y_ + tap_dy) < (unsigned)p.H && (unsigned)(x_ + tap_dx) < (unsigned)p.W; \
                mm_ = m_ + tap_shift;                                                                     \
            }                                                                                             \
            const unsigned rowb_ = (unsigned)mm_ * (unsigned)((s_isb[s] ? p.chB : p.chA) * NT * 64);      \
            rg[s] = s_isb[s] ? __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, ok_ ? rowb_ + s_goff[s] : OOB_OFF, 0, 0) \
                             : __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, ok_ ? rowb_ + s_goff[s] : OOB_OFF, 0, 0); \
        }                                                                                                 \
    }
#define MP_STORE(buf_)                                                                                    \
    if constexpr (F32SRC) {                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int op = 0; op < 2; ++op) { \
            const u32x4 u0_ = rg[ks * 4 + op * 2], u1_ = rg[ks * 4 + op * 2 + 1];                         \
            const float4 v0_ = make_float4(__uint_as_float(u0_.x), __uint_as_float(u0_.y), __uint_as_float(u0_.z), __uint_as_float(u0_.w)); \
            const float4 v1_ = make_float4(__uint_as_float(u1_.x), __uint_as_float(u1_.y), __uint_as_float(u1_.z), __uint_as_float(u1_.w)); \
            if (op == 0 && do_cs) {                                                                       \
                cs[0] += v0_.x; cs[1] += v0_.y; cs[2] += v0_.z; cs[3] += v0_.w;                           \
                cs[4] += v1_.x; cs[5] += v1_.y; cs[6] += v1_.z; cs[7] += v1_.w;                           \
            }                                                                                             \
            bf16x4 h0_, m0_, l0_, h1_, m1_, l1_;                                                          \
            split3(v0_, h0_, m0_, l0_);                                                                   \
            split3(v1_, h1_, m1_, l1_);                                                                   \
            unsigned char* d_ = smem + (buf_) * STAGE + flds;                                             \
            *reinterpret_cast<bf16x8*>(d_ + ((op * NT + 0) * KS + ks) * PIMG) = __builtin_shufflevector(h0_, h1_, 0, 1, 2, 3, 4, 5, 6, 7); \
            if (NT == 3) {                                                                                \
                *reinterpret_cast<bf16x8*>(d_ + ((op * NT + 1) * KS + ks) * PIMG) = __builtin_shufflevector(m0_, m1_, 0, 1, 2, 3, 4, 5, 6, 7); \
                *reinterpret_cast<bf16x8*>(d_ + ((op * NT + 2) * KS + ks) * PIMG) = __builtin_shufflevector(l0_, l1_, 0, 1, 2, 3, 4, 5, 6, 7); \
            }                                                                                             \
        }                                                                                                 \
    } else {                                                                                              \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                    \
            *reinterpret_cast<u32x4*>(smem + (buf_) * STAGE + s_lds[s]) = rg[s];                          \
    }
    const bool do_cs = F32SRC && p.colsum != nullptr && tj == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read addresses of this lane: group g = lane>>4 covers columns 16*(g&1).. of a 32-wide tile and
    // rows 8*(g>>1)..; lane 4q+pq of the group supplies row q, columns 4pq..4pq+3 (8 bytes)
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pq = lane & 3;
    unsigned fa[2][2], fb[2][2];                      // [tile][read r]: byte offset inside a plane image
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 8 * (g >> 1) + 4 * r + q4;
            const int cha = ((wm * 64 + t * 32) >> 3) + 2 * (g & 1) + (pq >> 1);
            const int chb = ((wn * 64 + t * 32) >> 3) + 2 * (g & 1) + (pq >> 1);
            fa[t][r] = img_off(row, cha) + 8u * (pq & 1);
            fb[t][r] = img_off(row, chb) + 8u * (pq & 1);
        }

    if (c_begin < c_end) {
        MP_LOAD(c_begin)
        MP_STORE(0)
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; c += KS) {
        const int buf = ((c - c_begin) / KS) & 1;
        if (c + KS < c_end) { MP_LOAD(c + KS) }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
        const unsigned char* st = smem + buf * STAGE + ks * PIMG;
        bf16x8 af[2][NT], bf[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < NT; ++pl) {
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + pl * KS * PIMG + fa[t][0]));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + pl * KS * PIMG + fa[t][1]));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + (NT + pl) * KS * PIMG + fb[t][0]));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + (NT + pl) * KS * PIMG + fb[t][1]));
                const s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                const s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                af[t][pl] = __builtin_bit_cast(bf16x8, av);
                bf[t][pl] = __builtin_bit_cast(bf16x8, bv);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (NT == 3) {   // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
        }
        if (c + KS < c_end) { MP_STORE(buf ^ 1) }
        __syncthreads();
    }
#undef MP_LOAD
#undef MP_STORE

    if constexpr (F32SRC) {
        if (do_cs) {       // the 16 row-threads of a column piece add up through LDS (free after the last barrier)
            float* sm = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[frow * 128 + fc8 * 8 + e] = cs[e];
            __syncthreads();
            if (tid < 128) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += sm[r * 128 + tid];
                if (ti * 128 + tid < p.Mi) p.colsum[(size_t)split * p.Mi + ti * 128 + tid] = t;
            }
        }
    }
    float* out = p.slab + (size_t)split * p.Mi * p.Nj;
    const int col0 = tj * 128 + wn * 64 + (lane & 31);
    const int row0 = ti * 128 + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (!F32SRC || (row < p.Mi && col0 + j * 32 < p.Nj)) out[(size_t)row * p.Nj + col0 + j * 32] = acc[i][j][r];
            }
}

bool mc_planes_supported(int C, int Cin) { return (C % 64) == 0 && (Cin % 128) == 0; }

// PA: planes of dOut [Mk][2C], PB: planes of X [Mk][Cin] (both NT planes per 32-channel chunk); slab as launch_mc
int launch_mc_planes(const void* PA, const void* PB, int C, int Cin, int Mk, int H, int W, float* slab,
                     const MCPlan& pl, int NT, hipStream_t st) {
    if (!mc_planes_supported(C, Cin) || !pl.big) return PA2D_ERR_UNSUPPORTED;
    MCPlanesParams p;
    p.PA = PA; p.chA = 2 * C / 32; p.PB = PB; p.chB = Cin / 32; p.slab = slab;
    p.Mi = 2 * C; p.Nj = 9 * Cin; p.Mk = Mk; p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits;
    p.H = H; p.W = W; p.Cin = Cin; p.taps = 9;
    const unsigned long long ab = (unsigned long long)Mk * p.chA * NT * 64, bb = (unsigned long long)Mk * p.chB * NT * 64;
    if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    const dim3 grid((p.Mi / 128) * (p.Nj / 128) * pl.splits);
    if (NT == 3) hipLaunchKernelGGL((gemm_mc_planes_kernel<3, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_mc_planes_kernel<1, 4>), grid, dim3(256), 0, st, p);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// =============================================================================================================
// 256 x 256 tile, 8 waves (2 per SIMD), every wave stages AND multiplies: the 128 x 128 kernel above moves 24 KB per 24
// MFMAs per wave and sits at the ~70 GB/s per CU at which L2 feeds a CU (1.75 ms per launch at the bench shape, MFMA
// floor 0.74).  Here a workgroup owns 256 dOut channels x 256 columns (= 8 groups of 32 input channels, each group
// with its own tap, so any Cin % 32 == 0 works) and each wave a 128 x 64 quarter-strip (4 x 2 MFMA tiles): 48 KB per
// K-step of 16 pixel rows for 8 x 48 MFMAs, i.e. HALF the staging bytes and 3/4 of the LDS fragment reads per MFMA,
// and 48 MFMAs between barriers instead of 24.  LDS: 2 stages x NT x (8 KB dOut image + 8 KB X image) = 96 KB, one
// workgroup per CU.  The launch is ONE round of <= 256 workgroups (tiles x splits); workgroups are numbered so that each
// XCD (blockIdx & 7) owns a contiguous range of (split, i-tile, j-tile): the 9+ workgroups that re-read the same dOut
// rows and the same X rows (at different tap shifts) sit behind one L2.
template <int NT, int KS>
__global__ __launch_bounds__(512, 1) void gemm_mc_planes_big_kernel(const MCPlanesParams p, const int per_xcd) {
    constexpr int PIMG = 8192;                       // one plane image of one K-step: 16 rows x 512 B (256 bf16 columns)
    constexpr int STAGE = 2 * NT * KS * PIMG;        // dOut planes then X planes
    constexpr int NP = (2 * NT * KS * 512) / 512;    // 16-byte pieces per thread and stage (16 rows x 32 pieces x NT x 2 / 512)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_i = (p.Mi + 255) / 256, tiles_j = (p.Nj + 255) / 256;
    const int tiles = tiles_i * tiles_j;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // XCD-contiguous numbering
    if ((int)(blockIdx.x >> 3) >= per_xcd || lin >= tiles * p.splits) return;
    const int split = lin / tiles, t_ = lin - split * tiles;
    const int ti = t_ / tiles_j, tj = t_ - ti * tiles_j;
    const int wm = wave >> 2, wn = wave & 3;         // 2 x 4 waves: 128 rows x 64 columns each
    const int total_chunks = (p.Mk + 15) / 16;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(total_chunks, c_begin + p.chunks_per_split);
    const int HW = p.taps == 1 ? 1 : p.H * p.W;

    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.PA), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.PB), 0, p.b_bytes, 0x00020000);

    // staging assignment: the 512 threads cover ONE plane image of ONE 16-row K-step (16 rows x 32 pieces of 16 bytes);
    // piece s of a thread = (operand, plane, k-step) of the same (row, column group c, piece) -> the metadata is per
    // thread, not per piece (a second register set for a 2-stage-deep prefetch was tried: 256 VGPRs + 160 spilled)
    const int row16 = tid >> 5, cgrp = (tid & 31) >> 2, piece = tid & 3;
    const bool colok_a = ti * 256 + cgrp * 32 < p.Mi;
    const unsigned goff_a = (unsigned)((((ti * 256) >> 5) + cgrp) * NT * 64 + piece * 16);
    const int jc = tj * 256 + cgrp * 32;             // X: column group = one tap, one 32-channel chunk
    const int tap = p.taps == 1 ? 4 : jc / p.Cin;    // plain GEMM: the centre tap (no shift)
    const bool colok_b = jc < p.Nj;
    const unsigned goff_b = (unsigned)((p.taps == 1 ? (jc >> 5) : ((jc - tap * p.Cin) >> 5)) * NT * 64 + piece * 16);
    const int tap_dy = tap / 3 - 1, tap_dx = tap - (tap / 3) * 3 - 1, tap_shift = tap_dy * p.W + tap_dx;
    const unsigned pitch_a = (unsigned)(p.chA * NT * 64), pitch_b = (unsigned)(p.chB * NT * 64);
    const unsigned lds_off = 512u * row16 + 16u * ((unsigned)(cgrp * 4 + piece) ^ (unsigned)(((row16 & 3) << 2) | ((row16 >> 2) & 3)));
#define MB_LOAD(c_, RG)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                               \
            const int m_ = ((c_) + ks) * 16 + row16;                                                      \
            const bool okr_ = m_ < p.Mk && (c_) + ks < c_end;      /* next split's rows stay out */       \
            bool okb_ = okr_ && colok_b;                                                                  \
            int mm_ = m_;                                                                                 \
            if (p.taps != 1) {      /* (ly, lx) = pixel of row m_ in its image, advanced 16 rows per K-step */ \
                okb_ = okb_ && (unsigned)(ly + tap_dy) < (unsigned)p.H && (unsigned)(lx + tap_dx) < (unsigned)p.W; \
                mm_ = m_ + tap_shift;                                                                     \
                lx += 16;                                                                                 \
                while (lx >= p.W) { lx -= p.W; ++ly; }                                                    \
                while (ly >= p.H) ly -= p.H;                                                              \
            }                                                                                             \
            const unsigned oa_ = (okr_ && colok_a) ? (unsigned)m_ * pitch_a + goff_a : OOB_OFF;           \
            const unsigned ob_ = okb_ ? (unsigned)mm_ * pitch_b + goff_b : OOB_OFF;                       \
            _Pragma("unroll") for (int pl = 0; pl < NT; ++pl) {                                           \
                RG[pl * KS + ks] = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, oa_ == OOB_OFF ? OOB_OFF : oa_ + pl * 64u, 0, 0); \
                RG[(NT + pl) * KS + ks] = __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, ob_ == OOB_OFF ? OOB_OFF : ob_ + pl * 64u, 0, 0); \
            }                                                                                             \
        }                                                                                                 \
    }
#define MB_STORE(buf_, RG)                                                                                \
    {                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                    \
            *reinterpret_cast<u32x4*>(smem + (buf_) * STAGE + s * PIMG + lds_off) = RG[s];                \
    }
    u32x4 rg0[NP];
    // pixel coordinates of this thread's row of the NEXT K-step to be loaded: two integer divisions once, not per K-step
    // (the loads walk the rows in order, 16 per K-step; PMC before: 2.4 vector instructions per MFMA, most of them these)
    int ly = 0, lx = 0;
    if (p.taps != 1) {
        const int n0 = (c_begin * 16 + row16) % HW;
        ly = n0 / p.W;
        lx = n0 - ly * p.W;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read addresses (see the 128 x 128 kernel): group g = lane>>4 covers columns 16*(g&1).. of a 32-wide tile
    // and rows 8*(g>>1)..; lane 4q+pq of the group supplies row q, columns 4pq..4pq+3 (8 bytes)
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pq = lane & 3;
    unsigned fa[4][2], fb[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * (g >> 1) + 4 * r + q4;
        const unsigned swz = (unsigned)(((row & 3) << 2) | ((row >> 2) & 3));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cha = ((wm * 128 + t * 32) >> 3) + 2 * (g & 1) + (pq >> 1);
            fa[t][r] = 512u * row + 16u * ((unsigned)cha ^ swz) + 8u * (pq & 1);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int chb = ((wn * 64 + t * 32) >> 3) + 2 * (g & 1) + (pq >> 1);
            fb[t][r] = 512u * row + 16u * ((unsigned)chb ^ swz) + 8u * (pq & 1);
        }
    }

#define MB_COMPUTE(buf_)                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                   \
        const unsigned char* st = smem + (buf_) * STAGE + ks * PIMG;                                      \
        bf16x8 af[4][NT], bf[2][NT];                                                                      \
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;                                        \
        _Pragma("unroll") for (int pl = 0; pl < NT; ++pl) {                                               \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                               \
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + pl * KS * PIMG + fa[t][0])); \
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + pl * KS * PIMG + fa[t][1])); \
                af[t][pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7)); \
            }                                                                                             \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                               \
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + (NT + pl) * KS * PIMG + fb[t][0])); \
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + (NT + pl) * KS * PIMG + fb[t][1])); \
                bf[t][pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7)); \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
                if constexpr (NT == 3) {   /* smallest terms first */                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0); \
                }                                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0); \
            }                                                                                             \
    }
    // write-after-barrier order: at the top of stage c the registers (loaded during stage c-KS) are stored into the
    // buffer that stage c-KS just released, the loads of stage c+2KS are issued at once, then the MFMAs of stage c run
    // -> LDS always holds the current AND the next stage, the registers the one after, with ONE register set
    if (c_begin < c_end) {
        MB_LOAD(c_begin, rg0)
        MB_STORE(0, rg0)
        if (c_begin + KS < c_end) MB_LOAD(c_begin + KS, rg0)
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; c += KS) {
        const int buf = ((c - c_begin) / KS) & 1;
        if (c + KS < c_end) {
            MB_STORE(buf ^ 1, rg0)
            if (c + 2 * KS < c_end) MB_LOAD(c + 2 * KS, rg0)
        }
        MB_COMPUTE(buf)
        __syncthreads();
    }
#undef MB_COMPUTE
#undef MB_LOAD
#undef MB_STORE

    float* out = p.slab + (size_t)split * p.Mi * p.Nj;
    const int col0 = tj * 256 + wn * 64 + (lane & 31);
    const int row0 = ti * 256 + wm * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.Nj) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.Mi) out[(size_t)row * p.Nj + col] = acc[i][j][r];
            }
    }
}

// one round of <= 256 workgroups: splits = 256 / tiles (>= 8 K-steps of 16 rows per split)
MCPlan plan_mc_planes_big(int Mi, int Nj, int Mk) {
    MCPlan pl;
    if (Mk < 1) Mk = 1;
    pl.big = 2;
    const int tiles = ceil_div(Mi, 256) * ceil_div(Nj, 256);
    const int total_chunks = ceil_div(Mk, 16);
    int splits = 256 / tiles;
    if (pa2d_env().mcb_splits > 0) splits = pa2d_env().mcb_splits;         // tuning knob PA2D_MCB_SPLITS
    const int max_splits = total_chunks / 8 > 0 ? total_chunks / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    pl.chunks_per_split = ceil_div(total_chunks, splits);
    pl.splits = ceil_div(total_chunks, pl.chunks_per_split);
    pl.slab_floats = (size_t)pl.splits * Mi * Nj;
    return pl;
}

// 0 = never, 1 = when the shape suits it (default), env PA2D_MC_BIG=off|auto
bool mc_planes_big_applies(int C, int Cin, int Mk) {
    if (pa2d_env().mc_big_off) return false;
    return (C % 16) == 0 && (Cin % 32) == 0 && 2 * C >= 256 && 9 * Cin >= 256 && Mk >= 16 * 8 * 4;
}

int launch_mc_planes_big(const void* PA, const void* PB, int C, int Cin, int Mk, int H, int W, float* slab,
                         const MCPlan& pl, int NT, hipStream_t st) {
    return launch_mc_planes_big_raw(PA, 2 * C, PB, Cin, 9, Mk, H, W, slab, pl, NT, st);
}

// General form: slab[split][Mi][Nj] += A[m][Mi]^T . B[m (shifted by tap)][Cin]; taps = 9 -> Nj = 9*Cin (3x3 conv weight
// gradient), taps = 1 -> Nj = Cin (plain dW = dY^T . X).  A, B: NT-plane images with Mi / Cin multiples of 32.
int launch_mc_planes_big_raw(const void* PA, int Mi, const void* PB, int Cin, int taps, int Mk, int H, int W, float* slab,
                             const MCPlan& pl, int NT, hipStream_t st) {
    if ((Mi & 31) || (Cin & 31) || (taps != 1 && taps != 9)) return PA2D_ERR_UNSUPPORTED;
    MCPlanesParams p;
    p.PA = PA; p.chA = Mi / 32; p.PB = PB; p.chB = Cin / 32; p.slab = slab;
    p.Mi = Mi; p.Nj = taps * Cin; p.Mk = Mk; p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits;
    p.H = H; p.W = W; p.Cin = Cin; p.taps = taps;
    const unsigned long long ab = (unsigned long long)Mk * p.chA * NT * 64, bb = (unsigned long long)Mk * p.chB * NT * 64;
    if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    const int wgs = ceil_div(p.Mi, 256) * ceil_div(p.Nj, 256) * pl.splits;
    const int per_xcd = ceil_div(wgs, 8);
    const dim3 grid(per_xcd * 8);
    // the LDS attribute is set on every launch: it is per device and the call is cheap
    if (NT == 3) {
        const int smem = 2 * 2 * 3 * 1 * 8192;
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mc_planes_big_kernel<3, 1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((gemm_mc_planes_big_kernel<3, 1>), grid, dim3(512), smem, st, p, per_xcd);
    } else {
        const int smem = 2 * 2 * 1 * 4 * 8192;      // NT = 1: four 16-row K-steps per stage (64 MFMAs per wave and barrier)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mc_planes_big_kernel<1, 4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((gemm_mc_planes_big_kernel<1, 4>), grid, dim3(512), smem, st, p, per_xcd);
    }
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// Plain dW = A^T . B of the bf16 engines straight from fp32 operands (A [Mk][lda] with Mi columns, B [Mk][ldb] with Nj
// columns; Mi, Nj multiples of 8), 128 x 128 tiles on plan_mc's split plan; colsum (optional): [splits][Mi].
bool mc_f32src_applies(int Mi, int Nj, int Mk, long long lda, long long ldb) {
    return pa2d_env().lin_dw_split && (Mi % 8) == 0 && (Nj % 8) == 0 && Mi > 64 && Nj > 64 && Mk >= 16 * 8 * 16 && (lda % 4) == 0 && (ldb % 4) == 0;
}

int launch_mc_f32src(const float* A, long long lda, int Mi, const float* B, long long ldb, int Nj, int Mk, float* slab,
                     const MCPlan& pl, int NT, float* colsum, hipStream_t st) {
    if (!pl.big) return PA2D_ERR_UNSUPPORTED;
    MCPlanesParams p = {};
    p.PA = A; p.PB = B; p.slab = slab; p.lda = lda; p.ldb = ldb; p.colsum = colsum;
    p.Mi = Mi; p.Nj = Nj; p.Mk = Mk; p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits;
    p.H = 1; p.W = 1; p.Cin = Nj; p.taps = 1; p.chA = 0; p.chB = 0;
    const unsigned long long ab = ((unsigned long long)(Mk - 1) * lda + Mi) * 4ull, bb = ((unsigned long long)(Mk - 1) * ldb + Nj) * 4ull;
    if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    const dim3 grid(ceil_div(Mi, 128) * ceil_div(Nj, 128) * pl.splits);
    if (NT == 3) hipLaunchKernelGGL((gemm_mc_planes_kernel<3, 1, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_mc_planes_kernel<1, 4, true>), grid, dim3(256), 0, st, p);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}
