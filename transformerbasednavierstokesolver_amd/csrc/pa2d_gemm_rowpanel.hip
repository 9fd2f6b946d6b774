// Row-stationary GEMM of the split engine for the K = C linears (to_out, MLP fc1 / fc2 and their data gradients):
//     C[M,N] = epilogue(A[M,K] . B[N,K]^T)        M = B*H*W rows (10^5 .. 10^6), K in {128, 256}, N % 64 == 0.
// gemm_panel_kernel (pa2d_gemm_panel.hip) splits every A element once per 128-column tile in dedicated producer waves
// whose VALU work, and whose consumers' store-only epilogue, add to the MFMA time instead of hiding behind it
// (measured there: 0.056 ms of MFMAs inside a 0.116 ms launch).  Here the ACTIVATION rows are the stationary operand:
//   * a wave owns 32 rows (two 16-row blocks) for a whole round: it loads them once (fragment-shaped: lane (row l&15,
//     k-group l>>4) reads the 8 consecutive k of its MFMA fragment), splits them once into the three bf16 planes
//     (x = hi + mid + lo) and keeps all K/32 x 2 fragments x 3 planes in registers (192 VGPRs at K = 256; one wave per SIMD);
//   * the WEIGHTS come as a fragment-ordered plane image made once per launch by pack_weight_image_kernel (N*K*6 bytes,
//     L2-resident, read by every workgroup): 24 KB stages (64 columns x 64 k x 3 planes) go global -> registers -> LDS
//     one stage ahead and are read back conflict-free with ds_read_b128 (the image IS the lane order);
//   * the four waves of a workgroup (128 rows) share each stage; per stage a wave issues 96 MFMAs 16x16x32 (6 terms of
//     order <= 2 per fragment pair, smallest first, fp32 accumulate).  16x16x32, not 32x32x16: the same fragments, bytes
//     and matrix-pipe cycles, but these kernels are power-limited on real data and the narrower shape holds ~1.10x the
//     FLOP/s at the clock the chip sustains (tools/probes/mfma_shape_probe.hip);
//   * a 64-column pair of accumulators is finished every K/64 stages and stored straight away, so the C writes, the
//     residual / pre-activation reads (fetched one pair ahead) and the next round's A rows (fetched during the first
//     stages of this round) are spread evenly over the launch instead of arriving in bursts.
// Workgroups are persistent (one per CU); the stage stream continues across rounds.
#include "pa2d_gemm_common.h"
#include <type_traits>
#ifndef RP_SGB
#define RP_SGB 1
#endif

namespace {

constexpr int RP_STAGE = 24 * 1024;      // 4 k-steps x 2 column tiles x 3 planes x 1 KB fragments
constexpr int RP_SLOTS = 3;              // LDS ring of weight stages
constexpr int RP_BIAS = 4096;            // bias image in front of the ring (small ds_read offsets)

typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st4s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float (&v)[4]) {
    const u32x4_ u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// image[stage = np * KC + kc][s = 0..3][j = 0..1][plane][lane] (16 B): sub-step s = (32-k step s >> 1, column half s & 1),
// column tile nt = 2 (s & 1) + j of 16 columns; lane (li = l & 15, kq = l >> 4) holds the plane of
// B[64 np + 16 nt + li][64 kc + 32 (s >> 1) + 8 kq + 0..7]
// B[n][k] = src[n * sn + k * sk] (sk = 1: the weight as stored, forward; sn = 1: its transpose, data gradient)
__global__ void pack_weight_image_kernel(const float* __restrict__ src, long long sn, long long sk, unsigned char* __restrict__ img,
                                         int N, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one fragment lane: (stage, s, j, lane)
    const int KC = K / 64;
    if (idx >= (N / 64) * KC * 8 * 64) return;
    const int lane = idx & 63, j = (idx >> 6) & 1, sub = (idx >> 7) & 3, stage = idx >> 9;
    const int np = stage / KC, kc = stage % KC;
    const int n = 64 * np + 16 * (2 * (sub & 1) + j) + (lane & 15), k0 = 64 * kc + 32 * (sub >> 1) + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[(long long)n * sn + (long long)(k0 + e) * sk];
    bf16x4 h0, m0, l0, h1, m1, l1;
    split3(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
    split3(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
    unsigned char* dst = img + (size_t)stage * RP_STAGE + (size_t)((sub * 2 + j) * 3) * 1024 + lane * 16;
    *reinterpret_cast<bf16x8*>(dst) = cat8(h0, h1);
    *reinterpret_cast<bf16x8*>(dst + 1024) = cat8(m0, m1);
    *reinterpret_cast<bf16x8*>(dst + 2048) = cat8(l0, l1);
}

// MODE 0: out = acc + bias (+ res);  MODE 1: aux = acc + bias (dropped when aux == NULL), out = gelu(acc + bias);
// MODE 2: out = acc * gelu'(aux).  M % 128 == 0, N % 64 == 0, K = 16 KS.
template <int KS, int MODE, bool HAS_RES>
__global__ __launch_bounds__(256, 1) void gemm_rowpanel_kernel(const KCParams p, const unsigned char* __restrict__ img,
                                                               const unsigned img_bytes) {
    constexpr int KC = KS / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int NP = p.N / 64, SPR = NP * KC;                  // stages per round
    float* const bias_s = reinterpret_cast<float*>(smem);      // [N <= 1024] (zeros without a bias); the stage ring follows
    unsigned char* const ring = smem + RP_BIAS;
    for (int c = tid; c < p.N; c += 256)
        bias_s[c] = (MODE != 2 && p.bias) ? (c < p.bias_split ? p.bias[c] : p.bias2[c - p.bias_split]) : 0.f;
    const int rounds = p.M / 128;
    const int n_my = ((int)blockIdx.x < rounds) ? (rounds - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (n_my == 0) return;
    const int T = n_my * SPR;                                // stages of this workgroup's stream

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t ri = make_rsrc_v(img, img_bytes);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.res ? p.res : p.C, p.res ? p.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.C, p.aux ? p.aux_bytes : 0u);
    const unsigned lda_b = (unsigned)p.lda * 4u, ldc_b = (unsigned)p.ldc * 4u, ldres_b = (unsigned)p.ldres * 4u,
                   ldaux_b = (unsigned)p.ldaux * 4u;

    // ---- weight stages: 24 wave-pieces of 1 KB, six per wave
    const unsigned piece = (unsigned)(wave * 6) * 1024u + (unsigned)lane * 16u;
    u32x4_ breg[6];
    int ld_stage = 0;                                        // stage (mod SPR) of the next stage load
    auto stage_load = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            breg[i] = __builtin_amdgcn_raw_buffer_load_b128(ri, piece + i * 1024u, (unsigned)ld_stage * RP_STAGE, 0);
        if (++ld_stage == SPR) ld_stage = 0;
    };
    auto stage_write = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4_*>(ring + slot * RP_STAGE + piece + i * 1024) = breg[i];
    };

    // ---- activation rows: raw fragment-shaped loads (two float4 per unit) and their split.  Unit u = (32-k step u >> 1,
    // 16-row block u & 1): K/16 units per round, four per 64-k chunk
    const unsigned a_voff = (unsigned)li * lda_b + (unsigned)kq * 32u;
    float4 raw[KS][2];
    bf16x8 ap[KS][3];
    auto a_load = [&](int u, unsigned voff, unsigned row0) {       // voff == OOB_OFF: reads nothing (zeros)
        const unsigned so = (row0 + 16u * (u & 1)) * lda_b;
        raw[u][0] = ld4s(ra, voff == OOB_OFF ? OOB_OFF : voff + (u >> 1) * 128u, so);
        raw[u][1] = ld4s(ra, voff == OOB_OFF ? OOB_OFF : voff + (u >> 1) * 128u + 16u, so);
    };
    auto a_split = [&](int ks) {
        bf16x4 h0, m0, l0, h1, m1, l1;
        split3(raw[ks][0], h0, m0, l0);
        split3(raw[ks][1], h1, m1, l1);
        ap[ks][0] = cat8(h0, h1); ap[ks][1] = cat8(m0, m1); ap[ks][2] = cat8(l0, l1);
    };

    int round = blockIdx.x;
    unsigned row0 = (unsigned)(round * 128 + wave * 32);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_load(ks, a_voff, row0);
    stage_load();
    stage_write(0);
    stage_load();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_split(ks);
    __syncthreads();

    // The weights are the MFMA's A operand and the activation rows its B operand, so a lane's accumulator quad holds ONE
    // output row (row0 + 16 rb + li) and four CONSECUTIVE columns 4 kq + 0..3 of a 16-column tile: the epilogue moves 16
    // bytes per lane and instruction.  Per-lane byte offsets of (row li, column 4 kq); row blocks and tiles add SGPR offsets.
    const unsigned vc = li * ldc_b + kq * 16u, vr = li * ldres_b + kq * 16u, vx = li * ldaux_b + kq * 16u;

    // B fragments of one sub-step (2 column tiles x 3 planes), double-buffered: the reads of sub-step s + 1 are issued
    // before the MFMAs of sub-step s
    bf16x8 bfb[2][2][3];
    int cur = 0;                                             // LDS slot of the current stage (ring of 3)
    auto frag_read = [&](int buf, int slot, int ks4) {
        const unsigned char* bs = ring + slot * RP_STAGE + lane * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int u = 0; u < 3; ++u) bfb[buf][j][u] = *reinterpret_cast<const bf16x8*>(bs + ((ks4 * 2 + j) * 3 + u) * 1024);
    };
    frag_read(0, 0, 0);

    f32x4 acc[2][4];                                         // [row block][16-column tile of the pair]
    float4 ov[2][4];                                         // residual / pre-activation operands, fetched at the pair's start
    auto ep_group = [&](const int q8, const int np) {        // 16 rows x 16 columns: one register quad per lane
        const int rb = q8 >> 2, nt = q8 & 3;
        const unsigned cb = (unsigned)(np * 64 + nt * 16) * 4u;
        const unsigned rw = row0 + 16u * rb;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[rb][nt][e];
        if (MODE == 1) {
            if (p.aux_deriv) {               // the backward wants gelu'(pre), not pre: it shares every term with gelu(pre)
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = dgelu_f(v[e]);
                st4s(raux, vx, rw * ldaux_b + cb, d);
            } else {
                st4s(raux, vx, rw * ldaux_b + cb, v);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
        } else if (MODE == 2) {
            if (p.aux_deriv) {
                v[0] *= ov[rb][nt].x; v[1] *= ov[rb][nt].y; v[2] *= ov[rb][nt].z; v[3] *= ov[rb][nt].w;
            } else {
                v[0] *= dgelu_f(ov[rb][nt].x); v[1] *= dgelu_f(ov[rb][nt].y); v[2] *= dgelu_f(ov[rb][nt].z); v[3] *= dgelu_f(ov[rb][nt].w);
            }
        } else if (HAS_RES) {
            v[0] += ov[rb][nt].x; v[1] += ov[rb][nt].y; v[2] += ov[rb][nt].z; v[3] += ov[rb][nt].w;
        }
#ifdef RP_NO_STORE
        if (p.M == 12345)
#endif
        st4s(rc, vc, rw * ldc_b + cb, v);
    };
    auto ov_load = [&](const int q8, const int np) {         // operands of group q8 of the CURRENT pair
        const int rb = q8 >> 2, nt = q8 & 3;
        const unsigned cb = (unsigned)(np * 64 + nt * 16) * 4u;
        if (MODE == 2) ov[rb][nt] = ld4s(raux, vx, (row0 + 16u * rb) * ldaux_b + cb);
        else if (HAS_RES) ov[rb][nt] = ld4s(rres, vr, (row0 + 16u * rb) * ldres_b + cb);
    };

    unsigned row0n = 0, a_voff_n = OOB_OFF;
    // one 64-column pair of one round: KC stages.  FIRST: the pair that also fetches the next round's rows; LAST: the pair
    // behind whose MFMAs those rows are split into the plane registers (chunk c of the K range is free after stage c).
    auto pair = [&](auto first_tag, auto last_tag, const int np) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 bv = *reinterpret_cast<const float4*>(bias_s + np * 64 + nt * 16 + 4 * kq);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                acc[rb][nt] = f32x4{bv.x, bv.y, bv.z, bv.w};
                ov_load(rb * 4 + nt, np);
            }
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            // Stage g sits in slot cur.  Stage g + 1 is in registers and goes to slot nxt now: its last readers (stage g - 2)
            // were done before anyone passed the mid-stage barrier of stage g - 1.  Stage g + 2 is requested.  (Past the end
            // of the stream the extra write / load are harmless: nobody reads them.)
            const int nxt = cur == 2 ? 0 : cur + 1;
#ifndef RP_NO_BSTAGE
            stage_write(nxt);
            stage_load();
#endif
#ifndef RP_NO_ANEXT
            if (FIRST) {                                     // next round's rows, four k-steps per stage
#pragma unroll
                for (int q = 0; q < 4; ++q) a_load(kc * 4 + q, a_voff_n, row0n);
            }
#endif
#pragma unroll
            for (int ks4 = 0; ks4 < 4; ++ks4) {
                if (ks4 < 3) frag_read((ks4 + 1) & 1, cur, ks4 + 1);
                else frag_read(0, nxt, 0);                   // written before the barrier below, by every wave
                const int ks = kc * 4 + ks4;
#ifndef RP_NO_ANEXT
                // the rows fetched for the next round: k-step ks - 4 is free once stage kc - 1 of the last pair is through
                // (the last chunk is split behind the first stage of the next round)
                if (LAST && kc >= 1) a_split(ks - 4);
                if (FIRST && kc == 0) a_split(KS - 4 + ks4);
#endif
                // sub-step ks4: 32-k step ks4 >> 1 of the chunk, column tiles 2 (ks4 & 1) + j; units (k-step, row block rb)
#define RP_MFMA(u_, v_)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)            \
        acc[rb][2 * (ks4 & 1) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                 \
            bfb[ks4 & 1][j][v_], ap[kc * 4 + (ks4 >> 1) * 2 + rb][u_], acc[rb][2 * (ks4 & 1) + j], 0, 0, 0);
#ifndef RP_NO_MFMA
                RP_MFMA(1, 1) RP_MFMA(2, 0) RP_MFMA(0, 2) RP_MFMA(1, 0) RP_MFMA(0, 1) RP_MFMA(0, 0)
#endif
#undef RP_MFMA
#if RP_SGB == 1
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#endif
#ifndef RP_NO_BARRIER
                if (ks4 == 1) __syncthreads();
#endif
            }
            cur = nxt;
        }
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) ep_group(q8, np);
    };
    using T_ = std::true_type; using F_ = std::false_type;

    for (int t = 0; t < n_my; ++t) {
        const bool more = t + 1 < n_my;
        row0n = more ? (unsigned)((round + (int)gridDim.x) * 128 + wave * 32) : 0u;
        a_voff_n = more ? a_voff : OOB_OFF;                  // past the last round the prefetch reads nothing
        pair(T_{}, F_{}, 0);
        for (int np = 1; np < NP - 1; ++np) pair(F_{}, F_{}, np);
        pair(F_{}, T_{}, NP - 1);
        round += gridDim.x;
        row0 = row0n;
    }
}

template <int KS, int MODE, bool HAS_RES>
int launch_rowpanel_t(const KCParams& p, const unsigned char* img, hipStream_t st) {
    const int smem = RP_BIAS + RP_SLOTS * RP_STAGE;
    {   // every launch: the attribute is per device, and the call is cheap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rowpanel_kernel<KS, MODE, HAS_RES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
    }
    const int rounds = p.M / 128;
    const unsigned img_bytes = (unsigned)((size_t)p.N * p.K * 6);
    hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, MODE, HAS_RES>), dim3(rounds < 256 ? rounds : 256), dim3(256), smem, st, p,
                       img, img_bytes);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

int rowpanel_mode(const KCParams& p) {
    const bool gelu = p.act == ACT_GELU, has_res = p.res != nullptr;
    if (p.epi == 0) return 0;
    if (gelu && !has_res && (p.epi == (EPI_ACT | EPI_STORE_PRE) || p.epi == EPI_ACT)) return 1;
    if (gelu && !has_res && p.epi == EPI_MUL_DACT && !p.bias) return 2;
    return -1;
}

template <int KS>
int launch_rowpanel_k(const KCParams& p, const unsigned char* img, hipStream_t st) {
    switch (rowpanel_mode(p)) {
        case 0: return p.res ? launch_rowpanel_t<KS, 0, true>(p, img, st) : launch_rowpanel_t<KS, 0, false>(p, img, st);
        case 1: return launch_rowpanel_t<KS, 1, false>(p, img, st);
        case 2: return launch_rowpanel_t<KS, 2, false>(p, img, st);
        default: return PA2D_ERR_ARG;
    }
}

}  // namespace

// bytes of the weight plane image of an [N, K] layer (0: the row-stationary kernel does not take this shape / engine)
size_t rowpanel_image_bytes(int N, int K, int engine) {
    if (engine != 1 || (K != 128 && K != 256) || N < 128 || N > 1024 || (N % 128) != 0) return 0;      // pairs of 64 columns, two in flight
    return (size_t)N * K * 6;
}

bool rowpanel_applies(const KCParams& p) {
    if (!pa2d_env().lin_rowpanel || p.io_bf16) return false;
    if (rowpanel_image_bytes(p.N, p.K, p.engine) == 0) return false;
    if (rowpanel_mode(p) < 0 || (p.epi == EPI_ACT && p.aux)) return false;
    // 16-byte loads / stores of accumulator quads: every leading dimension a multiple of 4 elements
    if ((p.lda & 3) || (p.ldc & 3) || (p.res && (p.ldres & 3)) || (p.aux && (p.ldaux & 3))) return false;
    return p.M / 128 >= 256;                                     // at least one round per CU
}

// sn / sk: element strides of B[n][k] in `w` (forward: ldw, 1; data gradient: 1, ldw)
int launch_pack_weight_image(const float* w, long long sn, long long sk, void* img, int N, int K, hipStream_t st) {
    const int count = (N / 64) * (K / 64) * 8 * 64;
    hipLaunchKernelGGL(pack_weight_image_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, st, w, sn, sk, (unsigned char*)img, N, K);
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}

// the first M - M % 128 rows of p (the caller runs the tail on the per-tile kernels)
int launch_kc_rowpanel(const KCParams& p, const void* img, hipStream_t st) {
    if (p.K == 256) return launch_rowpanel_k<16>(p, (const unsigned char*)img, st);
    return launch_rowpanel_k<8>(p, (const unsigned char*)img, st);
}
