// Weight-gradient GEMM engine (contraction over the row index of both operands), slab / column-sum reductions.
#include "pa2d_gemm_common.h"

// ---------------------------------------------------------------------------------------------

// NT = 1: same staging (fp32 tiles [16 rows m][BM]), but each lane gathers its 8 consecutive m of one column
// with 8 ds_read_b32, rounds them to bf16 and issues ONE v_mfma_f32_32x32x16_bf16 per tile and 16-row chunk
// instead of 8 fp32 MFMAs (bf16-compute mode).  NT = 3: the gathered values are split exactly into hi+mid+lo
// bf16 terms in registers and the six products of order <= 2 are accumulated (fp32 accuracy, see
// gemm_kc_split_kernel): 6 bf16 MFMAs (192 cycles) instead of 8 fp32 MFMAs (512 cycles).  NT = 0: exact fp32.
template <int NT>
__device__ __forceinline__ void mc_split_elem(float v, bf16x8 (&pl)[NT], int e) {
    const __bf16 h = (__bf16)v;
    pl[0][e] = h;
    if constexpr (NT == 3) {
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        pl[1][e] = m;
        pl[2][e] = (__bf16)(r1 - (float)m);
    }
}
template <int BM, int BN, bool IM2COL, int NT, int BK = 16>
__global__ __launch_bounds__(256, 2) void gemm_mc_kernel(const MCParams p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_TPR = BM / 4, A_RPP = 256 / A_TPR, A_IT = BK / A_RPP;
    constexpr int B_TPR = BN / 4, B_RPP = 256 / B_TPR, B_IT = BK / B_RPP;
    static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    float* const As = smem;
    float* const Bs = smem + 2 * BK * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_i = (p.Mi + BM - 1) / BM, tiles_j = (p.Nj + BN - 1) / BN;
    // XCD-aware map (speed only): when the split count is a multiple of 8, blocks b, b+8, ... (one XCD
    // under round-robin dispatch) own a contiguous range of splits = a contiguous range of rows m, so
    // each XCD's L2 streams 1/8 of the operands instead of all of them.
    int tj, ti, split;
    {
        const int tiles = tiles_i * tiles_j;
        if ((p.splits & 7) == 0) {
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

    // the host plans in 16-row chunks; a K-step of BK rows covers BK/16 of them
    const int total_chunks = (p.Mk + BK - 1) / BK;
    const int cps = (p.chunks_per_split * 16 + BK - 1) / BK;
    const int c_begin = split * cps;
    const int c_end = min(total_chunks, c_begin + cps);

    const int a_r = tid / A_TPR, a_c = (tid % A_TPR) * 4;
    const int b_r = tid / B_TPR, b_c = (tid % B_TPR) * 4;
    const int gi = ti * BM + a_c;
    const int gj = tj * BN + b_c;
    const bool a_col_ok = gi < p.Mi, b_col_ok = gj < p.Nj;
    int tap_dy = 0, tap_dx = 0, ci = 0;
    if (IM2COL && b_col_ok) {
        const int tap = gj / p.Cin;
        ci = gj - tap * p.Cin;
        tap_dy = tap / 3 - 1;
        tap_dx = tap - (tap / 3) * 3 - 1;
    }
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t ra_rsrc = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb_rsrc = make_rsrc(p.B, p.b_bytes);
    const unsigned a_col = a_col_ok ? (unsigned)gi * 4u : OOB_OFF;
    const unsigned b_col = b_col_ok ? (unsigned)(IM2COL ? ci : gj) * 4u : OOB_OFF;
    const int tap_shift = tap_dy * p.W + tap_dx;
    float4 ra[A_IT], rb[B_IT];
#define MC_LOAD(c_)                                                                                       \
    {                                                                                                     \
        const int m0_ = (c_) * BK;                                                                        \
        _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                                \
            const int m_ = m0_ + a_r + s * A_RPP;                                                         \
            ra[s] = buf_load4(ra_rsrc, (a_col != OOB_OFF && m_ < p.Mk) ? (unsigned)m_ * (unsigned)p.lda * 4u + a_col : OOB_OFF); \
        }                                                                                                 \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s) {                                                \
            const int m_ = m0_ + b_r + s * B_RPP;                                                         \
            bool ok_ = b_col != OOB_OFF && m_ < p.Mk;                                                     \
            if (IM2COL) {                                                                                 \
                const int n_ = m_ % HW;                                                                   \
                const int y_ = n_ / p.W, x_ = n_ - y_ * p.W;                                              \
                ok_ = ok_ && (unsigned)(y_ + tap_dy) < (unsigned)p.H && (unsigned)(x_ + tap_dx) < (unsigned)p.W; \
                rb[s] = buf_load4(rb_rsrc, ok_ ? (unsigned)(m_ + tap_shift) * (unsigned)p.ldb * 4u + b_col : OOB_OFF); \
            } else {                                                                                      \
                rb[s] = buf_load4(rb_rsrc, ok_ ? (unsigned)m_ * (unsigned)p.ldb * 4u + b_col : OOB_OFF);  \
            }                                                                                             \
        }                                                                                                 \
    }
    const bool do_cs = p.colsum != nullptr && tj == 0;      // column sums of A: once per (split, row tile)
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#define MC_STORE(buf_)                                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < A_IT; ++s) {                                                \
            *reinterpret_cast<float4*>(As + (buf_) * BK * BM + (a_r + s * A_RPP) * BM + a_c) = ra[s];     \
            if (do_cs) { cs.x += ra[s].x; cs.y += ra[s].y; cs.z += ra[s].z; cs.w += ra[s].w; }            \
        }                                                                                                 \
        _Pragma("unroll") for (int s = 0; s < B_IT; ++s)                                                  \
            *reinterpret_cast<float4*>(Bs + (buf_) * BK * BN + (b_r + s * B_RPP) * BN + b_c) = rb[s];     \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        MC_LOAD(c_begin)
        MC_STORE(0)
    }
    __syncthreads();
    const int li = lane & 31, kh = lane >> 5;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        if (c + 1 < c_end) MC_LOAD(c + 1)
        const float* a_s = As + buf * BK * BM + kh * BM + wm * WM + li;
        const float* b_s = Bs + buf * BK * BN + kh * BN + wn * WN + li;
        if constexpr (NT > 0) {
            const float* a8 = As + buf * BK * BM + kh * 8 * BM + wm * WM + li;
            const float* b8 = Bs + buf * BK * BN + kh * 8 * BN + wn * WN + li;
            bf16x8 af[TM][NT], bf[TN][NT];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) mc_split_elem<NT>(a8[e * BM + i * 32], af[i], e);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) mc_split_elem<NT>(b8[e * BN + j * 32], bf[j], e);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (NT == 3) {   // smallest terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        } else {
        // fragments of k-step kk+1 are fetched into the other register set before the MFMAs of
        // k-step kk issue, so the LDS latency hides behind 4 x 64 MFMA cycles
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = a_s[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = b_s[j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = a_s[(kk + 1) * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(kk + 1) & 1][j] = b_s[(kk + 1) * 2 * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
        }
        // pin the interleave: reads(0); { reads(kk+1); mfma(kk) } x 7; mfma(7)   (0x100 = DS read, 0x8 = MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2 - 1; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        if (c + 1 < c_end) MC_STORE(buf ^ 1)
        __syncthreads();
    }
#undef MC_LOAD
#undef MC_STORE

    if (do_cs) {       // the A_RPP threads that share a column group add up through LDS (free after the last barrier)
        *reinterpret_cast<float4*>(smem + a_r * BM + a_c) = cs;
        __syncthreads();
        for (int c = tid; c < BM; c += 256) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < A_RPP; ++r) t += smem[r * BM + c];
            if (ti * BM + c < p.Mi) p.colsum[(size_t)split * p.Mi + ti * BM + c] = t;
        }
    }
    float* out = p.slab + (size_t)split * p.Mi * p.Nj;
    const int col0 = tj * BN + wn * WN + (lane & 31);
    const int row0 = ti * BM + wm * WM + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.Nj) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.Mi) out[(size_t)row * p.Nj + col] = acc[i][j][r];
            }
    }
}


MCPlan plan_mc(int Mi, int Nj, int Mk) {
    MCPlan pl;
    if (Mk < 1) Mk = 1;       // empty contraction (batch 0): plan as one chunk, the entry points zero-fill instead
    pl.big = (Mi > 64 && Nj > 64) ? 1 : 0;
    const int bm = pl.big ? 128 : 64;
    const int tiles = ceil_div(Mi, bm) * ceil_div(Nj, bm);
    const int total_chunks = ceil_div(Mk, 16);
    // Split counts are multiples of 8 (one contiguous split range per XCD, see the kernel).  At most 4
    // workgroups per CU are resident, the rest run as slots free up, so the efficiency of a block
    // total is blocks / (256 * ceil(blocks/256)); take the smallest multiple of 8 that reaches
    // >= 0.97 with at least 2 workgroups per CU (fewer splits = less slab traffic; measured on the conv
    // weight gradient: 8/16/24/32 splits -> 3.63/3.02/2.84/2.75 ms).
    const int max_splits = total_chunks / 8 > 0 ? total_chunks / 8 : 1;   // >= 8 chunks per split
    int best = max_splits < 8 ? max_splits : 8;
    if (pa2d_env().mc_splits > 0) {                                        // tuning knob PA2D_MC_SPLITS
        best = pa2d_env().mc_splits;
    } else if (max_splits >= 8) {
        double best_eff = 0.0;
        for (int sp = 8; sp <= max_splits && sp <= 512; sp += 8) {
            const int blocks = tiles * sp;
            if (blocks < 512 && sp + 8 <= max_splits) continue;
            const double eff = (double)blocks / (256.0 * ceil_div(blocks, 256));
            if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
            if (eff >= 0.97) { best = sp; break; }
        }
    }
    if (best < 1) best = 1;
    if (best > max_splits) best = max_splits;
    pl.chunks_per_split = ceil_div(total_chunks, best);
    pl.splits = ceil_div(total_chunks, pl.chunks_per_split);
    pl.slab_floats = (size_t)pl.splits * Mi * Nj;
    return pl;
}

// out[idx] = sum_s slab[s][idx]; mode 1 additionally un-packs the conv weight gradient:
// slab row-major [2C][9][Cin] -> dWx / dWf in the reference's [C_out][C_in][3][3] layout.
// 64 consecutive idx x 4 slab lanes per workgroup: each lane sums slabs s = lane, lane+4, ... with
// 4 independent loads in flight; the 4 lane sums are added in fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int nslab, long long count,
                                                           float* __restrict__ out, float* __restrict__ out2,
                                                           int mode, int C, int Cin, int accumulate) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < count) {
        const float* p = slab + idx;
        int k = ty;
        for (; k + 12 < nslab; k += 16) {
            s0 += p[(size_t)k * count];
            s1 += p[(size_t)(k + 4) * count];
            s2 += p[(size_t)(k + 8) * count];
            s3 += p[(size_t)(k + 12) * count];
        }
        for (; k < nslab; k += 4) s0 += p[(size_t)k * count];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty != 0 || idx >= count) return;
    const float s = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
    float* dst;
    if (mode == 0) {
        dst = out + idx;
    } else if (mode == 2) {      // one vector split over two outputs at element C
        dst = idx < C ? out + idx : out2 + (idx - C);
    } else {
        const int ci = (int)(idx % Cin);
        const int tap = (int)((idx / Cin) % 9);
        const int co = (int)(idx / ((long long)Cin * 9));
        dst = (co < C ? out : out2) + ((size_t)(co % C) * Cin + ci) * 9 + tap;
    }
    *dst = accumulate ? *dst + s : s;
}

int launch_mc(const float* A, long long lda, int Mi, const float* B, long long ldb, int Nj, int Mk,
                     bool im2col, int H, int W, int Cin, float* slab, const MCPlan& pl, int engine, hipStream_t st,
                     float* colsum) {
    if ((Mi & 3) || (Nj & 3) || (lda & 3) || (ldb & 3)) return PA2D_ERR_ARG;
    if (im2col && (Cin & 3)) return PA2D_ERR_UNSUPPORTED;
    MCParams p;
    p.A = A; p.lda = lda; p.Mi = Mi; p.B = B; p.ldb = ldb; p.Nj = Nj; p.slab = slab; p.Mk = Mk;
    p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits; p.H = H; p.W = W; p.Cin = Cin;
    p.colsum = colsum;
    {
        const unsigned long long ab = ((unsigned long long)(Mk - 1) * lda + Mi) * 4ull;
        const unsigned long long bb = ((unsigned long long)(Mk - 1) * ldb + (im2col ? Cin : Nj)) * 4ull;
        if (ab >= 0xFFFFFFF0ull || bb >= 0xFFFFFFF0ull) return PA2D_ERR_UNSUPPORTED;
        p.a_bytes = (unsigned)ab;
        p.b_bytes = (unsigned)bb;
    }
    const int bm = pl.big ? 128 : 64;
    const dim3 grid(ceil_div(Mi, bm) * ceil_div(Nj, bm) * pl.splits);
    const bool bf = engine == 2;
    const int mc_bk = pa2d_env().mc_bk;
    if (pl.big && !bf && mc_bk == 32 && (pl.chunks_per_split % 2) == 0) {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 0, 32>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<128, 128, false, 0, 32>), grid, dim3(256), 0, st, p);
    } else if (pl.big && im2col && engine == 1 && (Cin % 32) == 0) {     // 6-term split, fp32 accuracy
        hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 3>), grid, dim3(256), 0, st, p);

    } else if (pl.big && bf) {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<128, 128, false, 1>), grid, dim3(256), 0, st, p);
    } else if (pl.big) {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<128, 128, true, 0>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<128, 128, false, 0>), grid, dim3(256), 0, st, p);
    } else {
        if (im2col) hipLaunchKernelGGL((gemm_mc_kernel<64, 64, true, 0>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_mc_kernel<64, 64, false, 0>), grid, dim3(256), 0, st, p);
    }
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int launch_reduce(const float* slab, int nslab, long long count, float* out, float* out2, int mode,
                         int C, int Cin, hipStream_t st, int accumulate) {
    const dim3 grid((unsigned)ceil_div_ll(count, 64));
    hipLaunchKernelGGL(reduce_slabs_kernel, grid, dim3(256), 0, st, slab, nslab, count, out, out2, mode, C, Cin,
                       accumulate);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int pa2d_launch_reduce(const float* slab, int nslab, long long count, float* out, hipStream_t st) {
    return launch_reduce(slab, nslab, count, out, nullptr, 0, 0, 0, st);
}

// small records (LayerNorm / head / slice / token parameter gradients): all slabs of a record element summed in a fixed
// order, result routed to its destination segment.  These reductions are latency-bound (a few MB spread over up to 1024
// slabs), so the 256 threads of a workgroup are IDX record elements x SL slab lanes with SL chosen by the slab count:
// every lane walks nslab / SL slabs four loads at a time (1024 slabs: 4 rounds instead of 64), then the SL lane sums
// are added in lane order.  Deterministic for a given nslab.
template <int SL>
__global__ __launch_bounds__(256) void reduce_segs_kernel(const float* __restrict__ slab, int nslab, long long count,
                                                          const ReduceSegs segs, int accumulate) {
    constexpr int IDX = 256 / SL;
    __shared__ float red[SL][IDX];
    const int tx = threadIdx.x % IDX, ty = threadIdx.x / IDX;
    const long long idx = (long long)blockIdx.x * IDX + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < count) {
        const float* p = slab + idx;
        int k = ty;
        for (; k + 3 * SL < nslab; k += 4 * SL) {
            s0 += p[(size_t)k * count];
            s1 += p[(size_t)(k + SL) * count];
            s2 += p[(size_t)(k + 2 * SL) * count];
            s3 += p[(size_t)(k + 3 * SL) * count];
        }
        for (; k < nslab; k += SL) s0 += p[(size_t)k * count];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty != 0 || idx >= count) return;
    float s = red[0][tx];
#pragma unroll
    for (int l = 1; l < SL; ++l) s += red[l][tx];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < segs.nseg && idx >= segs.begin[i] && idx < segs.begin[i + 1]) {
            float* d = segs.dst[i] + (idx - segs.begin[i]);
            *d = accumulate ? *d + s : s;
        }
}

int pa2d_launch_reduce_segs(const float* slab, int nslab, long long count, const ReduceSegs& segs, int accumulate,
                            hipStream_t st) {
    if (count <= 0) return PA2D_OK;
    if (nslab >= 256)
        hipLaunchKernelGGL(reduce_segs_kernel<64>, dim3((unsigned)ceil_div_ll(count, 4)), dim3(256), 0, st, slab, nslab, count,
                           segs, accumulate);
    else if (nslab >= 32)
        hipLaunchKernelGGL(reduce_segs_kernel<16>, dim3((unsigned)ceil_div_ll(count, 16)), dim3(256), 0, st, slab, nslab, count,
                           segs, accumulate);
    else
        hipLaunchKernelGGL(reduce_segs_kernel<4>, dim3((unsigned)ceil_div_ll(count, 64)), dim3(256), 0, st, slab, nslab, count,
                           segs, accumulate);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): partial[blk][n] over row blocks, then reduce_slabs.
template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ X, long long ld, int M, int N, int rows_per_block,
                                      float* __restrict__ partial) {
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            s0 += Act<T>::ld1(&X[(size_t)r * ld + c]);
            s1 += Act<T>::ld1(&X[(size_t)(r + 1) * ld + c]);
            s2 += Act<T>::ld1(&X[(size_t)(r + 2) * ld + c]);
            s3 += Act<T>::ld1(&X[(size_t)(r + 3) * ld + c]);
        }
        for (; r < r1; ++r) s0 += Act<T>::ld1(&X[(size_t)r * ld + c]);
        partial[(size_t)blockIdx.x * N + c] = (s0 + s1) + (s2 + s3);
    }
}

int colsum_blocks(int M) { int b = ceil_div(M, 128); return b > 1024 ? 1024 : (b < 1 ? 1 : b); }

// out2 != NULL: columns >= split go to out2[col - split]
int launch_colsum(const float* X, long long ld, int M, int N, float* out, float* partial, hipStream_t st,
                         float* out2, int split, int accumulate) {
    const int nb = colsum_blocks(M);
    const int rpb = ceil_div(M, nb);
    hipLaunchKernelGGL((colsum_partial_kernel<float>), dim3(nb), dim3(256), 0, st, X, ld, M, N, rpb, partial);
    PA2D_CHECK_LAUNCH();
    return launch_reduce(partial, nb, N, out, out2, out2 ? 2 : 0, split, 0, st, accumulate);
}
int launch_colsum_bf16(const void* X, long long ld, int M, int N, float* out, float* partial, hipStream_t st,
                       float* out2, int split, int accumulate) {
    const int nb = colsum_blocks(M);
    const int rpb = ceil_div(M, nb);
    hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)X, ld, M, N, rpb, partial);
    PA2D_CHECK_LAUNCH();
    return launch_reduce(partial, nb, N, out, out2, out2 ? 2 : 0, split, 0, st, accumulate);
}
