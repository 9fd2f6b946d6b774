// 3x3 implicit-GEMM conv on the bf16 matrix cores with the im2col halo tile RESIDENT in LDS (gfx950 / CDNA4).
//
// gemm_kc_split_kernel treats the conv as a plain GEMM over K = (32-channel chunk, tap): every one of the nine taps
// re-stages the SAME activation pixels (shifted by one) from L2 into LDS, so a 256x128 tile moves 80 KB per K-step
// and the kernel runs at the ~32 B/clk/CU at which a CU can stage operands (1.31 ms per launch at the bench shape,
// MFMA floor 0.74 ms).  Here a workgroup owns a SPATIAL tile of 8 x 32 pixels (BM = 256 GEMM rows) x 128 output
// channels and keeps, per 32-channel chunk, the (8+2) x (32+2) halo tile of the pre-split activation planes in LDS
// ONCE; the nine taps are nine K-steps that read the MFMA A fragments from that tile at a shifted row offset and
// only the weight tile (26 KB) is staged per K-step: 34.5 KB instead of 80 KB per K-step (2.3x less L2->LDS
// traffic), which puts the kernel back under the MFMA roof.
//
// Layout (NT = 3 planes: the exact hi+mid+lo split, six MFMA terms, fp32 accuracy; NT = 1: bf16 compute):
//   A (activations): planes [pixel][Cin/32 chunks][NT][32] bf16 made by split_planes_kernel (p.apre);
//   B (weights):     pack [n][K-step = chunk*9 + tap][NT][32] bf16 made by repack_split_kernel;
//   LDS: [halo tile 340 rows | weight stage 0 | weight stage 1], row pitch NT*64 + 16 bytes (an odd number of
//        16-byte slots: conflict-free ds_read_b128 fragment reads) = 70.7 + 2 x 26.6 KB = 124 KB, one workgroup/CU.
// Workgroup = 8 waves: waves 4-7 produce (buffer loads -> registers -> LDS, no conversion work), waves 0-3 consume
// (each a 128-pixel x 64-channel quarter: 4 x 2 tiles of v_mfma_f32_32x32x16_bf16).  The halo tile is single-
// buffered: the producers fetch chunk c+1 into registers while the consumers run the nine taps of chunk c and
// store it between two barriers at the chunk boundary (0.2 us per 11.6 us of MFMA work).
// Zero padding, ragged tiles (any H, W) and the K tail cost nothing: out-of-image pixels are buffer loads with an
// out-of-range offset (hardware returns 0), out-of-image outputs are stores with an out-of-range offset (dropped).
#include "pa2d_gemm_common.h"

namespace {
constexpr int TH = 8, TW = 32, BM = TH * TW, BN = 128;
constexpr int HH = TH + 2, HWD = TW + 2, HROWS = HH * HWD;          // 10 x 34 = 340 halo pixels
}

// TPB = taps per weight stage (= per barrier): 1 for NT = 3 (96 MFMAs per consumer wave and barrier; LDS is full), 3 for
// NT = 1 (one tap would be 16 MFMAs per barrier: the kernel would spend its time in s_barrier)
// MF16: the consumers issue v_mfma_f32_16x16x32_bf16 (8 x 4 tiles of 16 pixels x 16 channels per wave) instead of
// v_mfma_f32_32x32x16_bf16 (4 x 2 tiles of 32 x 32): the same fragments, bytes, MACs and matrix-pipe cycles, but the kernel
// is POWER-limited on real data (1.30 ms on random operands, 0.99 ms on zeros) and the 16x16x32 shape sustains ~1.10x the
// FLOP/s at the clock the chip holds under that load (tools/probes/mfma_shape_probe.hip: 1905 vs 1730 TFLOP/s).  The weights
// are the A operand there, so a lane's accumulator quad is four consecutive channels of one pixel: 16-byte stores.
// Row pitch: an even number (14) of 16-byte slots keeps the 16-row fragment reads conflict-free (13 is the 32-row optimum).
template <int NT, typename TO = float, int TPB = (NT == 1 ? 3 : 1), bool MF16 = false>
__global__ __launch_bounds__(512, 1) void conv_halo_kernel(const KCParams p, const int tiles_x, const int tiles_y,
                                                           const int nimg) {
    constexpr int PITCHB = NT * 64 + (MF16 ? 32 : 16);
    constexpr int PIECES = NT * 4;                                   // 16-byte pieces per row and chunk
    constexpr int A_BYTES = HROWS * PITCHB;
    constexpr int B_TAP = BN * PITCHB;                               // weight tile of one tap
    constexpr int B_STAGE = TPB * B_TAP;
    constexpr int SPC = 9 / TPB;                                     // weight stages per 32-channel chunk
    constexpr int AP_IT = (HROWS * PIECES + 255) / 256;              // halo pieces per producer thread (16 / 6)
    constexpr int BP_IT = (TPB * BN * PIECES) / 256;                 // weight pieces per producer thread and stage (6)
    constexpr int TM = 4, TN = 2;
    static_assert(9 % TPB == 0, "taps per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const a_s = smem;
    unsigned char* const b_s = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = nimg * tiles_y * tiles_x;
    const int tmx = (tiles_m + 7) / 8;                               // XCD-contiguous ranges of spatial tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_m = xcd * tmx + slot / tiles_n;
    const int tile_n = slot % tiles_n;
    if (tile_m >= tiles_m || slot / tiles_n >= tmx) return;
    const int img = tile_m / (tiles_y * tiles_x);
    const int trem = tile_m - img * (tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int nch = p.Cin / 32;                                      // 32-channel chunks
    const int nk = nch * 9;                                          // K-steps (chunk, tap) of the weight pack
    const int nst = nch * SPC;                                       // weight stages = barriers

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int ptid = tid - 256;
        const __amdgpu_buffer_rsrc_t ra_rsrc = make_rsrc(p.A, p.a_bytes);
        const __amdgpu_buffer_rsrc_t rb_rsrc = make_rsrc(p.B, p.b_bytes);
        unsigned ap_off[AP_IT], ap_lds[AP_IT];
#pragma unroll
        for (int s = 0; s < AP_IT; ++s) {
            const int q = ptid + 256 * s, row = q / PIECES, piece = q - row * PIECES;
            const int hy = row / HWD, hx = row - hy * HWD;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool ok = row < HROWS && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            ap_off[s] = ok ? ((unsigned)((img * p.H + yy) * p.W + xx) * (unsigned)(nch * PIECES) + piece) * 16u : OOB_OFF;
            ap_lds[s] = row < HROWS ? (unsigned)(row * PITCHB + piece * 16) : 0xFFFFFFFFu;
        }
        unsigned bp_off[BP_IT], bp_lds[BP_IT];
#pragma unroll
        for (int s = 0; s < BP_IT; ++s) {
            const int q = ptid + 256 * s, tapi = q / (BN * PIECES), r = q - tapi * (BN * PIECES);
            const int row = r / PIECES, piece = r - row * PIECES;
            const int gn = tile_n * BN + row;
            bp_off[s] = gn < p.N ? ((unsigned)gn * (unsigned)nk * PIECES + (unsigned)(tapi * PIECES) + piece) * 16u : OOB_OFF;
            bp_lds[s] = tapi * B_TAP + row * PITCHB + piece * 16;
        }
        u32x4 rq[AP_IT], rp0[BP_IT], rp1[BP_IT];
#define HA_LOAD(chunk_)                                                                                    \
    {                                                                                                      \
        const unsigned sh_ = (unsigned)(chunk_) * (PIECES * 16u);                                          \
        _Pragma("unroll") for (int s = 0; s < AP_IT; ++s)                                                  \
            rq[s] = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, ap_off[s] != OOB_OFF ? ap_off[s] + sh_ : OOB_OFF, 0, 0); \
    }
#define HA_STORE()                                                                                         \
    {                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < AP_IT; ++s)                                                  \
            if (ap_lds[s] != 0xFFFFFFFFu) *reinterpret_cast<u32x4*>(a_s + ap_lds[s]) = rq[s];              \
    }
#define HB_LOAD(sg_, RP)      /* stage sg_ = K-steps sg_*TPB .. sg_*TPB + TPB - 1 of the pack */                 \
    {                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < BP_IT; ++s)                                                  \
            RP[s] = __builtin_amdgcn_raw_buffer_load_b128(                                                 \
                rb_rsrc, bp_off[s] != OOB_OFF ? bp_off[s] + (unsigned)((sg_) * TPB) * (PIECES * 16u) : OOB_OFF, 0, 0); \
    }
#define HB_STORE(stage_, RP)                                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < BP_IT; ++s)                                                  \
            *reinterpret_cast<u32x4*>(b_s + (stage_) * B_STAGE + bp_lds[s]) = RP[s];                       \
    }
        // Weight stages are fetched FOUR stages ahead (register sets rp0..rp3: set s % 4 holds stage s from its load, issued
        // while the consumers run stage s-4, to its store during stage s-1): the 7 MB weight pack does not stay in a 4 MB L2,
        // so a tile often comes from the Infinity Cache, and with two stages of lead the producers reached the barrier late
        // (ablation: 15 % of the kernel's time was staging the consumers waited for).
        // Tried and measured neutral on a same-box A/B (1.2176 vs 1.2168 ms): a ring of THREE weight slots (stage s + 2
        // stored during stage s) with the consumers fetching the first fragments of stage s + 1 before the barrier that
        // ends stage s — the LDS latency behind the barrier is not what limits this kernel.
#define HB_STEP(sg_, RP_NEXT)   /* the consumers run stage sg_; RP_NEXT = set (sg_ + 1) % 4 */                   \
    {                                                                                                      \
        if ((sg_) + 1 < nst) {                                                                             \
            HB_STORE(((sg_) + 1) & 1, RP_NEXT)                                                             \
            if ((sg_) + 5 < nst) HB_LOAD((sg_) + 5, RP_NEXT)                                               \
        }                                                                                                  \
        __syncthreads();                                  /* end of stage sg_ */                            \
        if ((sg_) % SPC == SPC - 1 && (sg_) + 1 < nst) {  /* chunk boundary: swap the halo tile between two barriers */ \
            HA_STORE()                                                                                     \
            if ((sg_) / SPC + 2 < nch) HA_LOAD((sg_) / SPC + 2)                                            \
            __syncthreads();                                                                               \
        }                                                                                                  \
    }
        u32x4 rp2[BP_IT], rp3[BP_IT];
        HA_LOAD(0)
        HB_LOAD(0, rp0)
        if (nst > 1) HB_LOAD(1, rp1)
        if (nst > 2) HB_LOAD(2, rp2)
        if (nst > 3) HB_LOAD(3, rp3)
        HA_STORE()
        HB_STORE(0, rp0)
        if (nch > 1) HA_LOAD(1)
        if (nst > 4) HB_LOAD(4, rp0)
        __syncthreads();                                  // #0: halo tile of chunk 0 + weight stage 0 are ready
        for (int sg = 0; sg < nst; sg += 4) {
            HB_STEP(sg, rp1)
            if (sg + 1 < nst) HB_STEP(sg + 1, rp2)
            if (sg + 2 < nst) HB_STEP(sg + 2, rp3)
            if (sg + 3 < nst) HB_STEP(sg + 3, rp0)
        }
#undef HB_STEP
#undef HA_LOAD
#undef HA_STORE
#undef HB_LOAD
#undef HB_STORE
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    if constexpr (MF16) {
        static_assert(NT == 3 && TPB == 1, "the 16x16x32 consumers are written for the split engine");
        const int li = lane & 15, kq = lane >> 4;
        f32x4 acc[8][4];
#pragma unroll
        for (int ib = 0; ib < 8; ++ib)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ib][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // lane l holds row (l & 15), k-group (l >> 4) of a 16-row x 32-k fragment.  Row block ib = (image row i = ib >> 1,
        // pixel half h = ib & 1) of the wave's 4 x 32 pixels; column block j = 16 output channels.
        const unsigned a_frag = (unsigned)((((wm * 4 + 1) * HWD) + 1 + li) * PITCHB + kq * 16);
        const unsigned b_frag = (unsigned)((wn * 64 + li) * PITCHB + kq * 16);
        __syncthreads();                                      // #0
        int tap = 0;
        for (int sg = 0; sg < nst; ++sg) {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const unsigned char* af_base = a_s + a_frag + (dy * HWD + dx) * PITCHB;
            const unsigned char* bf_base = b_s + (sg & 1) * B_STAGE + b_frag;
            bf16x8 bf[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < NT; ++q) bf[j][q] = *reinterpret_cast<const bf16x8*>(bf_base + j * 16 * PITCHB + q * 64);
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                bf16x8 af[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    af[q] = *reinterpret_cast<const bf16x8*>(af_base + (ib >> 1) * (HWD * PITCHB) + (ib & 1) * 16 * PITCHB + q * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) {                 // smallest terms first; weights = A operand (rows = channels)
                    acc[ib][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][1], af[1], acc[ib][j], 0, 0, 0);
                    acc[ib][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[2], acc[ib][j], 0, 0, 0);
                    acc[ib][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][2], af[0], acc[ib][j], 0, 0, 0);
                    acc[ib][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[1], acc[ib][j], 0, 0, 0);
                    acc[ib][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][1], af[0], acc[ib][j], 0, 0, 0);
                    acc[ib][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[0], acc[ib][j], 0, 0, 0);
                }
            }
            ++tap;
            __syncthreads();                                  // end of stage sg
            if (tap == 9) {
                tap = 0;
                if (sg + 1 < nst) __syncthreads();            // the producers swapped the halo tile in between
            }
        }
        // epilogue: accumulator quad r = 0..3 of lane l = channels col0 + j*16 + 4*(l>>4) + r of pixel x0 + (ib&1)*16 + (l&15)
        // of image row y0 + wm*4 + (ib>>1): one 16-byte (fp32) / 8-byte (bf16) store per quad
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
        constexpr unsigned ES = Act<TO>::ES;
        const unsigned ldc_b = (unsigned)p.ldc * ES, roww_b = (unsigned)p.W * ldc_b;
        const bool inside = y0 + TH <= p.H && x0 + TW <= p.W;      // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = tile_n * BN + wn * 64 + j * 16 + 4 * kq;
            const bool col_ok = col < p.N;                    // N % 4 == 0 (checked by the launcher)
            float bv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                bv[e] = (p.bias && col_ok) ? (col + e < p.bias_split ? p.bias[col + e] : p.bias2[col + e - p.bias_split]) : 0.f;
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                const int y = y0 + wm * 4 + (ib >> 1), x = x0 + (ib & 1) * 16 + li;
                const bool ok = col_ok && (inside || (y < p.H && x < p.W));
                const unsigned off = ok ? (unsigned)((img * p.H + y) * p.W + x) * ldc_b + (unsigned)col * ES : OOB_OFF;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[ib][j][e] + bv[e];
                if constexpr (ES == 4) {
                    const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(u, rc, off, 0, 0);
                } else {
                    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                    const u32x2_ u = {(unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16),
                                      (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16)};
                    __builtin_amdgcn_raw_buffer_store_b64(u, rc, off, 0, 0);
                }
            }
        }
        (void)roww_b;
        return;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment addresses: lane l holds row (l & 31), k-half (l >> 5) of a 32-row tile.  A rows are the 32 pixels of
    // image row (wm*4 + i) of the tile, found in the halo tile at [(row + 1 + dy)][1 + dx + pixel].
    const unsigned a_frag = (unsigned)((((wm * 4 + 1) * HWD) + 1 + (lane & 31)) * PITCHB + (lane >> 5) * 16);
    const unsigned b_frag = (unsigned)((wn * 64 + (lane & 31)) * PITCHB + (lane >> 5) * 16);
    __syncthreads();                                      // #0
    int tap = 0;
    for (int sg = 0; sg < nst; ++sg) {
#pragma unroll
        for (int tt = 0; tt < TPB; ++tt, ++tap) {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const unsigned char* af_base = a_s + a_frag + (dy * HWD + dx) * PITCHB;
            const unsigned char* bf_base = b_s + (sg & 1) * B_STAGE + tt * B_TAP + b_frag;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[TM][NT], bf[TN][NT];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        af[i][q] = *reinterpret_cast<const bf16x8*>(af_base + i * (HWD * PITCHB) + q * 64 + ks * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        bf[j][q] = *reinterpret_cast<const bf16x8*>(bf_base + j * 32 * PITCHB + q * 64 + ks * 32);
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
            }
        }
        __syncthreads();                                  // end of stage sg
        if (tap == 9) {
            tap = 0;
            if (sg + 1 < nst) __syncthreads();            // the producers swapped the halo tile in between
        }
    }

    // epilogue: out[pixel][col] = acc + bias.  Accumulator register r of lane l is pixel x0 + 4*(l>>5) + (r&3) + 8*(r>>2)
    // of image row y0 + wm*4 + i, column tile_n*128 + wn*64 + j*32 + (l & 31): 32 lanes write 128 contiguous bytes.
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, p.c_bytes);
    constexpr unsigned ES = Act<TO>::ES;
    if (y0 + TH <= p.H && x0 + TW <= p.W) {
        // tile inside the image (wave-uniform): element = per-lane offset of the wave's first pixel + an SGPR offset per
        // (image row i, register r); no per-element address arithmetic or bounds tests
        const unsigned ldc_b = (unsigned)p.ldc * ES, roww_b = (unsigned)p.W * ldc_b;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tile_n * BN + wn * 64 + j * 32 + (lane & 31);
            const bool col_ok = col < p.N;
            const float bv = (p.bias && col_ok) ? (col < p.bias_split ? p.bias[col] : p.bias2[col - p.bias_split]) : 0.f;
            const unsigned vo = col_ok ? (unsigned)((img * p.H + y0 + wm * 4) * p.W + x0 + 4 * (lane >> 5)) * ldc_b + (unsigned)col * ES
                                       : OOB_OFF;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = (unsigned)i * roww_b + (unsigned)((r & 3) + 8 * (r >> 2)) * ldc_b;
                    if constexpr (ES == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r] + bv), rc, vo, so, 0);
                    else __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16_bits(acc[i][j][r] + bv), rc, vo, so, 0);
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = tile_n * BN + wn * 64 + j * 32 + (lane & 31);
        const bool col_ok = col < p.N;
        const float bv = (p.bias && col_ok) ? (col < p.bias_split ? p.bias[col] : p.bias2[col - p.bias_split]) : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int y = y0 + wm * 4 + i;
            const unsigned rowbase = (unsigned)((img * p.H + y) * p.W);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && y < p.H && x < p.W;
                Act<TO>::bst1(rc, ok ? ((rowbase + (unsigned)x) * (unsigned)p.ldc + (unsigned)col) * ES : OOB_OFF,
                              acc[i][j][r] + bv);
            }
        }
    }
}

// 0 = never, 1 = when the launch fills the chip (default), 2 = always (tests): env PA2D_CONV_HALO=off|auto|force
static int halo_policy() { return pa2d_env().conv_halo; }

bool conv_halo_applies(const KCParams& p) {
    const int pol = halo_policy();
    if (pol == 0 || !p.apre || (p.Cin % 32) != 0 || p.H <= 0 || p.W <= 0 || (p.M % (p.H * p.W)) != 0) return false;
    const long long tiles = (long long)(p.M / (p.H * p.W)) * ceil_div(p.H, TH) * ceil_div(p.W, TW) * ceil_div(p.N, BN);
    return pol == 2 || tiles >= 512;
}

// p.a_bytes / p.b_bytes must already describe the plane image and the weight pack (launch_kc_split fills them)
int launch_conv_halo(const KCParams& p, hipStream_t st) {
    const int NT = p.engine == 2 ? 1 : 3;
    const int tiles_x = ceil_div(p.W, TW), tiles_y = ceil_div(p.H, TH), nimg = p.M / (p.H * p.W);
    const int tiles_m = nimg * tiles_y * tiles_x, tiles_n = ceil_div(p.N, BN);
    const dim3 grid(ceil_div(tiles_m, 8) * 8 * tiles_n);
    const int pitch = NT * 64 + 16;
    const int smem = HROWS * pitch + 2 * (NT == 1 ? 3 : 1) * BN * pitch;
    // the LDS attribute is set on every launch: it is per device and the call is cheap
    if (NT == 3 && pa2d_env().conv_mfma16 && (p.N % 4) == 0 && (p.ldc % 4) == 0) {      // 16-byte stores of accumulator quads
        const int smem16 = (HROWS + 2 * BN) * (3 * 64 + 32);
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<3, float, 1, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem16);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv_halo_kernel<3, float, 1, true>), grid, dim3(512), smem16, st, p, tiles_x, tiles_y, nimg);
    } else if (NT == 3) {
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<3>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv_halo_kernel<3>), grid, dim3(512), smem, st, p, tiles_x, tiles_y, nimg);
    } else if (p.io_bf16) {
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<1, bf16_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv_halo_kernel<1, bf16_t>), grid, dim3(512), smem, st, p, tiles_x, tiles_y, nimg);
    } else {
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv_halo_kernel<1>), grid, dim3(512), smem, st, p, tiles_x, tiles_y, nimg);
    }
    PA2D_CHECK_LAUNCH();
    return PA2D_OK;
}
