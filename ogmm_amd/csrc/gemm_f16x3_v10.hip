// fp16x3 split GEMM, LDS-DMA structure, FOUR waves (one per SIMD) with 64 rows x 256 columns each: 16 accumulators = 256 accumulation registers.
//
// Same operand path and arithmetic as gemm_f16x3_v8.hip (both operands by global_load_lds_dwordx4, activations as wave-private stages of raw
// fp32 rows with the XOR chunk swizzle, weights as stages of the fragment-major image, split of A in registers after the fragment read, per
// accumulator the products lo*hi, hi*lo, hi*hi per k16 block in ascending k: bit-identical output).  What changes is the blocking:
//   * a weight fragment read from LDS feeds TWO row blocks (12 MFMAs per 4 ds_read_b128 instead of 6): 160 KiB of LDS reads per CU and K step
//     instead of 288 -- the fragment reads were the largest stall of the v8 loop and, on this power-limited part, a large share of its energy;
//   * one wave per SIMD: nothing covers a stall, so every non-MFMA instruction has a fixed slot BETWEEN two matrix instructions of its group
//     (sched_barrier after every slot), at most ~one MFMA duration of issue per slot, and the one barrier of a K step sits in the middle of
//     the step's last group, followed by the first weight fragments of the next step (six MFMAs cover their latency).
#include <cstdlib>
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BK8 = 32;
constexpr int RB = 2, NT = 8, WM = 4;
constexpr int BM = RB * 32 * WM, BN = NT * 32, T = WM * 64;                 // 256, 256, 256
constexpr int A_STAGE = BM * BK8 * 4;                                        // 32768 B
constexpr int B_STAGE = BN * BK8 * 2 * 2;                                    // 32768 B
constexpr int A_STAGES = 2, B_STAGES = 2;
constexpr int AFF_OFF = A_STAGES * A_STAGE + B_STAGES * B_STAGE, AFF_MAX_K = 4096;
constexpr int B_OFF = A_STAGES * A_STAGE;
constexpr int LDS_BYTES = A_STAGES * A_STAGE + B_STAGES * B_STAGE;          // 131072 B (+ 32768 B with AFF)

__device__ unsigned long long g_v10_probe[4];

// OVL: the GEMM is the batched similarity S = fn_src fn_tgt^T of the overlap block (models/gmmreg.py:75-80) and S is never stored: the epilogue
// forms e = exp(S - 1) (|S| <= 1 for normalised rows, so no running maximum is needed: softmax(S) = e / sum e) and leaves, per tile, the partial
// softmax-dots of its 256 rows against o_tgt and of its 256 columns against o_src as (1, sum e, sum e o) triples; ogmm_overlap_finalize merges them.
// TERMS: matrix instructions per product block (struct ogmm_gemm.terms; the per-layer term budget of HISTORY.md section 4).
//   3  lo*hi + hi*lo + hi*hi: fp32-class, the default
//   2  lo*hi + hi*hi = (a_hi + a_lo) w_hi: the WEIGHT is rounded to binary16, the activation keeps both terms.  The lo plane of the weight image
//      is not even fetched: 4 instead of 8 weight DMA instructions and 16 instead of 32 fragment reads per K step and wave.
// Both run the same instruction schedule: a group is 4 TERMS matrix instructions, and everything else sits in the gaps m = 0..7 of a group.
//   1  hi*hi: BOTH operands rounded to binary16 (11 significand bits); no lo term is formed at all.  Four matrix instructions per group: the raw
//      activation fragments are read in gap 3 of a half step's first group, converted (v_cvt_pk_f16_f32 only) in gap 3 of its second and third.
//      With a third of the matrix work the loop is bound by its operand DMA (48 KiB per step and CU).
// TA: the A operand is read TRANSPOSED (struct ogmm_gemm.a_trans): A[m][k] = A_mem[k * lda + m], the weight gradient's dY^T without the transposed copy.
//   A wave's stage is then [32 k][64 m] fp32 = the same 8 KiB from the same 8 DMA instructions, each fetching 4 k-rows x 256 B; lane l of instruction i
//   fetches row 4 i + (l >> 4), 16-byte chunk (l & 15) ^ (8 * ((i >> 1) & 1)), so that element (k, m) lies at byte k * 256 + ((m / 4) ^ (8 * ((k >> 3) & 1))) * 16
//   + (m % 4) * 4.  Lane (lr, lh) of the MFMA's A operand holds k = 16 s + 8 lh + e, e = 0..7, of row m = 32 rb + lr: eight dwords 256 B apart = four
//   ds_read2st64_b32 (offsets 16 s + e, + 1 in units of 256 B) from lane base lh * 2048 + (((rb ^ lh) * 8 + lr / 4) * 16) + (lr % 4) * 4 -- the chunk swizzle
//   puts the two lane halves on disjoint banks.  Same k order, same split, same products: bit-identical to the plain form on a transposed copy.
//   TA = 2: the same, and the column sums of A_mem (sum over k of A[m][k] = the bias gradient dy.sum(0)) ride along: every lane adds the eight raw values of
//   each fragment it reads (fp32 tree of 8, then fp64) in gaps the split leaves free; the n-tile-0 workgroups add them to a_colsum[m] (fp64 atomics).
template <int ABL, bool AFF, bool OVL, int TERMS = 3, bool NBS = false, int TA = 0>
__global__ __launch_bounds__(T) void gemm_f16x3_v10_kernel(const ogmm_gemm g, const int m_tiles_signed, const int n_tiles) {
    static_assert(TERMS == 3 || ((TERMS == 2 || TERMS == 1) && !AFF), "TERMS < 3 has no InstanceNorm-on-A form (its transform pieces need the gaps of 12 MFMAs)");
    static_assert(!TA || (!AFF && !OVL && !NBS && TERMS >= 2), "the transposed-A form exists for the plain product with three or two terms");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem10[];

    const int bid = blockIdx.x;
    long long probe_c0 = 0, probe_w0 = 0;
    if (ABL & 2048) { probe_c0 = clock64(); probe_w0 = wall_clock64(); }
    int tile_m, tile_n;
    if (m_tiles_signed < 0) {
        tile_m = bid / n_tiles;
        tile_n = bid % n_tiles;
    } else {            // XCD-aware map (block b runs on XCD b % 8): all N tiles of an M panel on one XCD
        const int xcd = bid & 7, local = bid >> 3;
        tile_m = (local / n_tiles) * 8 + xcd;
        tile_n = local % n_tiles;
        if (tile_m >= m_tiles_signed) return;
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int zb = blockIdx.z;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int m_end = min(g.M, m0 + BM);
    const int nk1 = g.K1 / BK8, nk2 = g.K2 / BK8, nk = nk1 + nk2;

    // ---- DMA sources.  A: wave w stages rows [64 w, 64 w + 64), instruction i rows 8 i .. 8 i + 7, lane l -> row (l >> 3), LDS chunk (l & 7)
    // <- global chunk (l & 7) ^ ((row >> 1) & 7).  Rows beyond M are clamped (their results are never stored).
    // A rows gathered on the fly (ogmm_gemm.a_gather_ids): output row m = c S + s reads source row map(c) N + ids[map(c)][s]; the DMA's per-lane row
    // offsets then come from the index list, relative to A itself instead of the tile's first row
    const bool gathered = g.a_gather_ids != nullptr;
    const float* __restrict__ A1p = g.A + zb * g.sA_o + (TA ? (int64_t)m0 : (gathered ? 0 : (int64_t)m0 * g.lda));
    const float* __restrict__ A2p = g.A2 ? g.A2 + zb * g.sA2_o + (int64_t)m0 * g.lda2 : nullptr;
    const unsigned lds0 = (unsigned)(size_t)smem10;
    unsigned aoff[8];
    auto set_aoff = [&](int ld) {
        if (TA) {          // k-row 4 i + (lane >> 4) of the stage, columns m0 + 64 wave + 4 chunk .. + 3 (clamped to the last whole quad of M: never stored)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = min(wave * 64 + (((lane & 15) ^ (((i >> 1) & 1) * 8)) << 2), g.M - 4 - m0);
                aoff[i] = (unsigned)((i * 4 + (lane >> 4)) * ld + col) * 4u;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 64 + i * 8 + (lane >> 3);
            int src_row = min(r, g.M - 1 - m0);
            if (gathered) {
                const int mrow = m0 + src_row, c = mrow / g.a_gather_S, sidx = mrow - c * g.a_gather_S;
                const int sc = g.a_gather_map ? g.a_gather_map[c] : c;
                src_row = sc * g.a_gather_N + g.a_gather_ids[(int64_t)sc * g.a_gather_S + sidx];
            }
            aoff[i] = (unsigned)(src_row * ld + ((lane & 7) ^ ((r >> 1) & 7)) * 4) * 4u;
        }
    };
    set_aoff((int)g.lda);
    // B: wave w stages column blocks 2 w, 2 w + 1: instruction i = (column block i >> 2, k16 = (i >> 1) & 1, plane = i & 1)
    const int KB = (int)(g.ldb_h / 16);
    const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_hi) + zb * g.sB_o) + ((int64_t)(n0 / 32 + 2 * wave) * KB) * 64;
    const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_lo) + zb * g.sB_o) + ((int64_t)(n0 / 32 + 2 * wave) * KB) * 64;
    const unsigned boff = lane * 16;

    auto issue_a_piece = [&](int t, int i) {
        const bool second = t >= nk1;
        if (i == 0 && t == nk1 && nk2 > 0) set_aoff((int)g.lda2);          // stages are issued in order and piece 0 first: switch to the second A piece once
        const float* Ap = TA ? A1p + (int64_t)t * BK8 * g.lda : (second ? A2p + (t - nk1) * BK8 : A1p + t * BK8);
        lds_dma16(aoff[i], Ap, lds0 + (t % A_STAGES) * A_STAGE + wave * 8192 + i * 1024);
    };
    auto issue_b_piece = [&](int t, int i) {
        const int kb = (t < nk1 ? t * 2 : (g.K1 / 16) + (t - nk1) * 2) * 64;
        lds_dma16(boff, ((i & 1) ? BL : BH) + (int64_t)(i >> 2) * KB * 64 + kb + ((i >> 1) & 1) * 64,
                  lds0 + B_OFF + (t % B_STAGES) * B_STAGE + (2 * wave + (i >> 2)) * 4096 + (i & 3) * 1024);
    };

    // row block 0 / 1 of this wave against the eight column blocks.  With three or more K steps the accumulators are not cleared: the first product
    // on each of them takes C = 0 (256 v_accvgpr_write per lane and tile less: ~1000 cycles); shorter loops clear them.
    f32x16 acc0[NT], acc1[NT];
    if (nk < 3) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.0f; acc1[j][r] = 0.0f; }
    }

    // fragment read offsets: A rows (wave*64 + rb*32 + lr), chunk (s*4 + lh*2 + q) ^ ((lr >> 1) & 7); B: all column blocks
    const int a_rd = (wave * 64 + lr) * 128;
    const int a_sw = (lr >> 1) & 7;
    const int a_c0 = ((lh * 2) ^ a_sw) << 4, a_c1 = ((lh * 2 + 1) ^ a_sw) << 4;        // k16 block 0; block 1 = chunk ^ 4 = byte offset ^ 64
    const int b_rd = lane * 16;

    f32x4 ra[RB][2];
    f16x2 h01[4], h23[4], l01[4], l23[4];          // split in progress: [which = row block * 2 + half of the fragment's 8 k]
    f16x4 ahh[RB][2][2], alh[RB][2][2];            // [row block][k16 block][half of the fragment's 8 k]
    f16x8 bh[2][2], bl[2][2];                      // [group parity][column block of the pair]
    f32x4 rsc[2], rsh[2];                          // AFF: the 8 scales / shifts of the fragment's k positions
    const float aff_lo = (AFF && g.a_relu) ? 0.0f : -__builtin_inff();
    const int ta_rd = wave * 8192 + lh * 2048 + ((lr >> 2) << 4) + ((lr & 3) << 2);
    auto read_a = [&](int tau, int s, int rb) {          // raw fp32 fragment of row block rb, k16 block s of stage tau
        if (TA) {
            const unsigned char* As = smem10 + (tau % A_STAGES) * A_STAGE + ta_rd + ((rb ^ lh) << 7) + s * 4096;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ra[rb][0][e] = *reinterpret_cast<const float*>(As + e * 256);
                ra[rb][1][e] = *reinterpret_cast<const float*>(As + (4 + e) * 256);
            }
            return;
        }
        const unsigned char* As = smem10 + (tau % A_STAGES) * A_STAGE + a_rd + rb * 4096;
        ra[rb][0] = *reinterpret_cast<const f32x4*>(As + (a_c0 ^ (s * 64)));
        ra[rb][1] = *reinterpret_cast<const f32x4*>(As + (a_c1 ^ (s * 64)));
    };
    auto read_aff = [&](int tau, int s) {
        if (AFF) {
            const int Kt = g.K1 + g.K2;
            const int k0 = (tau < nk1 ? tau * BK8 : g.K1 + (tau - nk1) * BK8) + s * 16 + lh * 8;
            const float* tab = reinterpret_cast<const float*>(smem10 + AFF_OFF) + k0;
            rsc[0] = *reinterpret_cast<const f32x4*>(tab);
            rsc[1] = *reinterpret_cast<const f32x4*>(tab + 4);
            rsh[0] = *reinterpret_cast<const f32x4*>(tab + Kt);
            rsh[1] = *reinterpret_cast<const f32x4*>(tab + Kt + 4);
        }
    };
    // The split of one raw fragment half (4 values; `which` = row block * 2 + half) in pieces of TWO vector instructions, so that each piece fits
    // beside one matrix instruction (one wave per SIMD hides about five single-issue instructions per MFMA; a whole split4 in one gap costs its
    // full issue time: measured 41 cycles per split4, 10 % of the loop).  hi = rn16(x) (v_cvt_pk_f16_f32), lo = rn16(x - hi) (v_fma_mix*_f16 reads
    // the binary16 source in place; x - hi is exact in fp32).  The two instructions of a piece are independent and consecutive pieces of one
    // split4 are at least one MFMA apart, so no wait states are needed inside the asm.
    auto piece_aff = [&](int which, int sub) {          // sub 0 / 1: v = v * scale + shift on values 0,1 / 2,3; sub 2 / 3: the lower clamp
        f32x4& v = ra[which >> 1][which & 1];
        const int h = which & 1, e = (sub & 1) * 2;
        if (sub < 2) { v[e] = fmaf(v[e], rsc[h][e], rsh[h][e]); v[e + 1] = fmaf(v[e + 1], rsc[h][e + 1], rsh[h][e + 1]); }
        else { v[e] = fmaxf(v[e], aff_lo); v[e + 1] = fmaxf(v[e + 1], aff_lo); }
    };
    auto piece_split = [&](int s, int which, int stage) {
        const f32x4 v = ra[which >> 1][which & 1];
        if (stage == 0) {
            asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5" : "=&v"(h01[which]), "=&v"(h23[which]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
            if (TERMS == 1) ahh[which >> 1][s][which & 1] = f16x4{h01[which][0], h01[which][1], h23[which][0], h23[which][1]};          // (no lo term: done)
        } else if (stage == 1)
            asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %1, %3, -1.0, %5 op_sel_hi:[1,0,0]"
                : "=&v"(l01[which]), "=&v"(l23[which]) : "v"(h01[which]), "v"(h23[which]), "v"(v[0]), "v"(v[2]));
        else {
            asm("v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "+v"(l01[which]), "+v"(l23[which]) : "v"(h01[which]), "v"(h23[which]), "v"(v[1]), "v"(v[3]));
            ahh[which >> 1][s][which & 1] = f16x4{h01[which][0], h01[which][1], h23[which][0], h23[which][1]};
            alh[which >> 1][s][which & 1] = f16x4{l01[which][0], l01[which][1], l23[which][0], l23[which][1]};
        }
    };
    // piece number pi of a half step (k16 block s): AFF: 16 transform pieces, then 4 x cvt, 4 x lo, 4 x hi; else the 12 split pieces
    auto piece = [&](int s, int pi) {
        if (AFF) {
            if (pi < 16) piece_aff(pi >> 2, pi & 3);
            else piece_split(s, (pi - 16) & 3, (pi - 16) >> 2);
        } else piece_split(s, pi & 3, pi >> 2);
    };
    // TA = 2: column sums of the raw fragments, in pieces of <= 3 vector instructions (gaps m = 8..11 of the groups whose gaps m = 4..7 hold the split)
    constexpr bool CS = TA == 2;
    float cs_a[4];
    double cs_acc[2] = {0.0, 0.0};
    auto cs_piece = [&](int gl, int j) {
        if (gl == 1) { const f32x4 v = ra[j >> 1][j & 1]; cs_a[j] = (v[0] + v[1]) + (v[2] + v[3]); }
        else if (gl == 2) { if (j < 2) cs_a[2 * j] += cs_a[2 * j + 1]; }
        else if (j < 2) cs_acc[j] += (double)cs_a[2 * j];
    };
    auto read_b = [&](int tau, int grp, int c) {          // MFMA group grp = k16 block grp >> 2, column blocks 2q, 2q+1 with q = grp & 3
        const unsigned char* Bs = smem10 + B_OFF + (tau % B_STAGES) * B_STAGE + b_rd;
        const int s = grp >> 2, q = grp & 3;
        bh[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 0) * 1024);
        if (TERMS == 3) bl[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 1) * 1024);
    };

    // ---- prologue.  DMA order B(0), A(0), A(1): the counted waits below rely on it.  (TERMS < 3: only the hi plane = the even pieces)
#pragma unroll
    for (int i = 0; i < 8; i += (TERMS == 3 ? 1 : 2)) issue_b_piece(0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a_piece(0, i);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_a_piece(1, i);
    }
    if (AFF) {          // the tile's rows belong to one group (group_rows is a multiple of the tile): its K scales, then its K shifts
        const int Kt = g.K1 + g.K2;
        const float* __restrict__ sc = g.a_scale + (int64_t)(m0 / g.group_rows) * Kt;
        const float* __restrict__ sh = g.a_shift + (int64_t)(m0 / g.group_rows) * Kt;
        float* tab = reinterpret_cast<float*>(smem10 + AFF_OFF);
        for (int i = tid; i < Kt; i += T) { tab[i] = sc[i]; tab[Kt + i] = sh[i]; }
    }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // B(0), A(0) of this wave landed (A(1) may be in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                       // everybody's weight pieces (and the AFF table) are visible
    read_a(0, 0, 0);
    read_a(0, 0, 1);
    read_aff(0, 0);
    read_b(0, 0, 0);
    read_b(0, 0, 1);
#pragma unroll
    for (int pi = 0; pi < (AFF ? 28 : (TERMS == 1 ? 4 : 12)); ++pi) piece(0, pi);
    if (CS) {
#pragma unroll
        for (int gl = 1; gl < 4; ++gl)
#pragma unroll
            for (int j = 0; j < 4; ++j) cs_piece(gl, j);
    }

    // One K step = 8 MFMA groups of 12: k16 block s = grp >> 2 against the column-block pair q = grp & 3, for both row blocks; MFMA m of a group is
    // product m >> 2 (lo*hi, hi*lo, hi*hi), row block (m >> 1) & 1, column block m & 1 -- per accumulator the same order as v8.  What the wave issues
    // in the gap after MFMA m of group grp (while that MFMA executes):
    //   m = 0, 3          one DMA instruction each: weights of stage t+1 in groups 0-3, activations of stage t+2 in groups 4-7
    //   m = 1, 2          the next group's weight fragments (2 ds_read_b128 each); in group 7, m = 2: the step's barrier -- this wave's weight
    //                     pieces of stage t+1 landed, all its reads of stage t are done -- followed by the first fragments of step t+1
    //   groups 0 / 4      m = 4, 5 (6): raw activation fragments of k16 block 1 of this stage / block 0 of the next stage (after the vmcnt wait for them)
    //   groups 1-3 / 5-7  m = 4..7 (AFF: groups 0-3 / 4-7, m = 8..11 and 4..11): their split, two vector instructions per gap
    auto step = [&](int t, auto has_b_c, auto has_a_c, auto first_c) {
        constexpr bool HAS_B = decltype(has_b_c)::value, HAS_A = decltype(has_a_c)::value;          // stage t+1 / t+2 exist
        constexpr bool FIRST = decltype(first_c)::value;                                             // the tile's first step: C = 0 for the first products
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int s = grp >> 2, q = grp & 3, p = grp & 1, gl = grp & 3;
            const bool second = grp >= 4;                 // second half of the step: prepares k16 block 0 of stage t+1
            const bool prep = second ? HAS_B : true;      // (first half: block 1 of stage t)
            const f16x8 ah0 = __builtin_shufflevector(ahh[0][s][0], ahh[0][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 al0 = __builtin_shufflevector(alh[0][s][0], alh[0][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 ah1 = __builtin_shufflevector(ahh[1][s][0], ahh[1][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 al1 = __builtin_shufflevector(alh[1][s][0], alh[1][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int m = 0; m < 4 * TERMS; ++m) {
                const int prod = TERMS == 3 ? m >> 2 : (TERMS == 2 ? (m >> 2) * 2 : 2), rb = (m >> 1) & 1, c = m & 1;          // TERMS = 2: products 0 (lo*hi) and 2 (hi*hi); 1: hi*hi
                const f16x8 av = prod == 0 ? (rb ? al1 : al0) : (rb ? ah1 : ah0);
                const f16x8 bv = prod == 1 ? bl[p][c] : bh[p][c];
                const bool fresh = FIRST && s == 0 && (m >> 2) == 0;          // the accumulator's first product of the tile
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (rb == 0) acc0[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, fresh ? zero : acc0[2 * q + c], 0, 0, 0);
                else acc1[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, fresh ? zero : acc1[2 * q + c], 0, 0, 0);
                // ---- the gap after MFMA m
                if (!(ABL & 1) && (m == 0 || m == 3)) {
                    const int pc = 2 * gl + (m == 3);          // (weights: even pieces = hi plane, odd = lo plane)
                    if (!second) { if (HAS_B && (TERMS == 3 || m == 0)) issue_b_piece(t + 1, pc); }
                    else { if (HAS_A) issue_a_piece(t + 2, pc); }
                }
                if (!(ABL & 4) && grp < 7 && (m == 1 || m == 2)) read_b(t, grp + 1, m - 1);
                if (HAS_B && grp == 7 && m == 2) {
                    // younger than this wave's weight pieces of stage t+1: the 7 activation pieces of stage t+2 issued so far
                    if (!(ABL & 1)) { if (HAS_A) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (!(ABL & 4)) { read_b(t + 1, 0, 0); read_b(t + 1, 0, 1); }
                }
                if (TERMS == 1) {
                    // four gaps per group: raw fragments of both row blocks in gap 3 of the half step's first group, their conversion in gap 3 of the next two
                    if (prep && m == 3) {
                        const int tau = second ? t + 1 : t, sn = second ? 0 : 1;
                        if (gl == 0) {
                            if (second && !(ABL & 1)) { if (HAS_A) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                            if (!(ABL & 16)) { read_a(tau, sn, 0); read_a(tau, sn, 1); }
                        } else if (gl <= 2 && !(ABL & 2)) {
                            piece_split(sn, 2 * (gl - 1), 0);
                            piece_split(sn, 2 * (gl - 1) + 1, 0);
                        }
                    }
                } else
                if (prep && gl == 0) {
                    const int tau = second ? t + 1 : t, sn = second ? 0 : 1;
                    if (m == 4) {
                        // own activation pieces of stage t+1 landed: younger are the 8 weight pieces of this step and the 2 activation pieces of this group
                        if (second && !(ABL & 1)) {
                            if (TERMS == 3) { if (HAS_A) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                            else { if (HAS_A) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }          // 4 weight pieces per step
                        }
                        if (!(ABL & 16)) read_a(tau, sn, 0);
                    }
                    if (m == 5 && !(ABL & 16)) read_a(tau, sn, 1);
                    if (m == 6) read_aff(tau, sn);
                }
                if (TERMS != 1 && prep && !(ABL & 2)) {
                    const int sn = second ? 0 : 1;
                    if (AFF) {
                        if (gl == 0 && m >= 8) piece(sn, m - 8);
                        if (gl > 0 && m >= 4) piece(sn, 4 + (gl - 1) * 8 + (m - 4));
                    } else if (gl > 0 && m >= 4 && m < 8) piece(sn, (gl - 1) * 4 + (m - 4));
                    // (three terms: the gaps m = 8..11 are free; two terms: a group has eight gaps, the sums share m = 4..7 with the split: <= 5 instructions per gap)
                    if (CS && gl > 0 && m >= (TERMS == 3 ? 8 : 4) && m < (TERMS == 3 ? 12 : 8)) cs_piece(gl, m - (TERMS == 3 ? 8 : 4));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    {
        int t = 0;
        if (nk >= 3) { step(0, std::true_type{}, std::true_type{}, std::true_type{}); t = 1; }
        for (; t + 2 < nk; ++t) step(t, std::true_type{}, std::true_type{}, std::false_type{});
        if (t + 1 < nk) { step(t, std::true_type{}, std::false_type{}, std::false_type{}); ++t; }
        step(t, std::false_type{}, std::false_type{}, std::false_type{});
    }
    if ((ABL & 2048) && threadIdx.x == 0) {
        atomicAdd(&g_v10_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_v10_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
        atomicAdd(&g_v10_probe[2], 1ull);
    }
    if (g.overflow) {
        // binary16 overflow flag: some |a| > 65504 made its hi part infinite, and then EVERY output of that row is inf or nan (inf * 0 = nan): one
        // column block per row block tells (32 instructions per tile instead of two per split4 in the loop)
        float chk = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { chk = fmaf(acc0[0][r], 0.0f, chk); chk = fmaf(acc1[0][r], 0.0f, chk); }
        if (chk != chk) atomicOr(g.overflow, 1);
    }
    if constexpr (CS) {
        if (tile_n == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                double v = cs_acc[rb];
                v += __shfl_xor(v, 32, 64);
                const int mcol = m0 + wave * 64 + rb * 32 + lr;
                if (lh == 0 && mcol < g.M) atomicAdd(g.a_colsum + mcol, v);
            }
        }
    }
    __builtin_amdgcn_s_barrier();          // every wave is done with the last stage: LDS becomes the epilogue's scratch
    if (ABL & 8) {          // ablation: no output stores
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc0[j][r] + acc1[j][r];
        if (sum == 1.2345f) g.C[0] = sum;
        return;
    }
    if constexpr (OVL) {
        // LDS (floats): o of the tile's rows | o of its columns | row scale | row results [256][2] | column partials [4 waves][256][2]
        float* s_orow = reinterpret_cast<float*>(smem10);
        float* s_ocol = s_orow + 256;
        float* s_rinv = s_orow + 512;
        float* s_rowres = s_orow + 768;
        float* s_col = s_orow + 1280;
        s_orow[tid] = g.ovl_orow[((int64_t)zb * g.M + m0 + tid) * g.ovl_ld];
        s_ocol[tid] = g.ovl_ocol[((int64_t)zb * g.N + n0 + tid) * g.ovl_ld];
        s_rinv[tid] = g.row_rscale ? g.row_rscale[(int64_t)zb * g.M + m0 + tid] : 1.0f;
        __syncthreads();
        constexpr float L2E = 1.4426950408889634f;
        float csum[NT], cdot[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { csum[j] = 0.0f; cdot[j] = 0.0f; }
        auto row_block = [&](f32x16 (&acc)[NT], int rb) {
            float rs[16], rd[16], orow[16], fac[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wave * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                orow[r] = s_orow[rl];
                fac[r] = g.alpha * s_rinv[rl] * L2E;
                rs[r] = 0.0f; rd[r] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float oc = s_ocol[j * 32 + lr];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(acc[j][r], fac[r], -L2E));          // exp(s - 1), s = acc * alpha * row scale
                    rs[r] += e;
                    rd[r] = fmaf(e, oc, rd[r]);
                    csum[j] += e;
                    cdot[j] = fmaf(e, orow[r], cdot[j]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { rs[r] = half_wave_sum(rs[r]); rd[r] = half_wave_sum(rd[r]); }
            if (lr == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wave * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s_rowres[rl * 2] = rs[r];
                    s_rowres[rl * 2 + 1] = rd[r];
                }
            }
        };
        row_block(acc0, 0);
        row_block(acc1, 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {          // the two half waves hold the same columns
            csum[j] += __shfl_xor(csum[j], 32, 64);
            cdot[j] += __shfl_xor(cdot[j], 32, 64);
            if (lh == 0) {
                s_col[(wave * 256 + j * 32 + lr) * 2] = csum[j];
                s_col[(wave * 256 + j * 32 + lr) * 2 + 1] = cdot[j];
            }
        }
        __syncthreads();
        const int m_tiles_z = g.M / BM;
        float* rp = g.ovl_rowpart + ((((int64_t)zb * n_tiles + tile_n) * g.M) + m0 + tid) * 3;
        rp[0] = 1.0f; rp[1] = s_rowres[tid * 2]; rp[2] = s_rowres[tid * 2 + 1];
        float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { c1 += s_col[(w * 256 + tid) * 2]; c2 += s_col[(w * 256 + tid) * 2 + 1]; }
        float* cp = g.ovl_colpart + ((((int64_t)zb * m_tiles_z + tile_m) * g.N) + n0 + tid) * 3;
        cp[0] = 1.0f; cp[1] = c1; cp[2] = c2;
        return;
    }
    ogmm_gemm gz = g;
    if (gz.C) gz.C += zb * g.sC_o;
    if (gz.Res) gz.Res += zb * g.sR_o;
    const bool inside = m0 + BM <= m_end && n0 + BN <= g.N && !g.row_affine;
    if (inside) {
        float* stat_lds = reinterpret_cast<float*>(smem10);          // [8 row blocks][256 columns][2]: the rings are dead (barrier above)
        gemm_epilogue_rowblock<NT, true, NBS>(gz, acc0, m0 + wave * 64, n0, g.alpha, stat_lds, wave * 2);
        gemm_epilogue_rowblock<NT, true, NBS>(gz, acc1, m0 + wave * 64 + 32, n0, g.alpha, stat_lds, wave * 2 + 1);
        if (g.col_stats) {
            __syncthreads();
            // thread = column: add the eight row blocks' partial sums (fp64), one atomic per column and statistic per tile
            const int c = tid;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                double tot = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) tot += NBS ? (double)stat_lds[(w * 256 + c) * 2 + which] : reinterpret_cast<const double*>(stat_lds)[(w * 256 + c) * 2 + which];
                atomicAdd(g.col_stats + (int64_t)((m0 >> 8) & g.col_stats_slot_mask) * g.col_stats_slot_stride + ((int64_t)(m0 / g.group_rows) * g.N + n0 + c) * 2 + which, tot);
            }
        }
    } else {
        auto fallback = [&](auto qc) {          // (a run-time q would index the accumulators dynamically: they would live in scratch)
            constexpr int q = decltype(qc)::value;
            f32x16 quad[2][2] = {{acc0[2 * q], acc0[2 * q + 1]}, {acc1[2 * q], acc1[2 * q + 1]}};
            gemm_epilogue<2, 2, 4, 1, false>(gz, quad, reinterpret_cast<float*>(smem10), m0, n0 + q * 64, m_end, 0, 0, g.alpha);
        };
        fallback(std::integral_constant<int, 0>{});
        fallback(std::integral_constant<int, 1>{});
        fallback(std::integral_constant<int, 2>{});
        fallback(std::integral_constant<int, 3>{});
    }
}

}  // namespace

namespace ogmm {

bool gemm_f16x3_v10_applicable(const ogmm_gemm& g) {
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch_outer;
    static const int enabled = [] { const char* e = getenv("OGMM_V10"); return e ? atoi(e) : 1; }();
    static const long long min_tiles = [] { const char* e = getenv("OGMM_V10_MIN_TILES"); return e ? atoll(e) : 256LL; }();
    const bool whole_tiles = g.M % BM == 0 && g.N % BN == 0 && !g.row_affine;
    const bool gather_ok = !g.a_gather_ids || (g.K2 == 0 && !g.a_scale && g.batch_outer * g.batch_inner == 1 && g.a_gather_S > 0 && g.a_gather_N > 0 &&
                                               (int64_t)g.a_gather_rows * g.lda * 4 < (1ll << 32));
    const bool ovl_ok = !g.ovl_rowpart || (whole_tiles && g.ovl_colpart && g.ovl_orow && g.ovl_ocol && g.ovl_ld >= 1 && !g.a_scale && !g.col_stats && !g.Res && g.batch_inner == 1);
    const bool nb_ok = !g.nb_mean || (whole_tiles && g.nb_rstd && g.nb_scale && g.nb_shift && g.col_stats && g.Res && g.C && !g.a_scale && !g.ovl_rowpart && !g.a_gather_ids &&
                                      g.group_rows > 0 && g.group_rows % BM == 0 && g.batch_outer * g.batch_inner == 1 && g.N >= 512 &&
                                      (g.nb_act == OGMM_ACT_RELU || g.nb_act == OGMM_ACT_LEAKY02));
    const bool ta_ok = !g.a_trans || (g.K2 == 0 && !g.a_scale && !g.a_gather_ids && !g.ovl_rowpart && !g.nb_mean && g.M % 4 == 0 && g.M >= 4 && g.terms != 1 &&
                                      (int64_t)BK8 * g.lda * 4 + 1024 < (1ll << 31));
    return enabled && g.pool_k == 0 && (!g.col_stats || whole_tiles) && ovl_ok && gather_ok && nb_ok && ta_ok &&
           (!g.a_scale || (g.a_shift && g.group_rows > 0 && g.group_rows % BM == 0 && g.K1 + g.K2 <= AFF_MAX_K && (g.K1 + g.K2) % 4 == 0)) && g.N >= 256 && tiles >= min_tiles && g.K1 % BK8 == 0 && g.K2 % BK8 == 0 && g.ldb_h % 64 == 0 &&
           (g.K2 == 0 || g.K1 % 64 == 0) && (g.K1 + 63) / 64 * 64 + (g.K2 + 63) / 64 * 64 <= g.ldb_h && (g.lda % 4) == 0 && (g.K2 == 0 || (g.lda2 % 4) == 0);
}

template <int ABL, bool AFF = false, bool OVL = false, int TERMS = 3, bool NBS = false, int TA = 0>
static int launch_v10(const ogmm_gemm& g, hipStream_t s) {
    const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    static ogmm::PerDeviceOnce attr_once;          // per template instance and device
    if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v10_kernel<ABL, AFF, OVL, TERMS, NBS, TA>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + (AFF ? 32768 : 0));
    if (m_tiles % 8 != 0 && m_tiles < 32)
        hipLaunchKernelGGL((gemm_f16x3_v10_kernel<ABL, AFF, OVL, TERMS, NBS, TA>), dim3((unsigned)(m_tiles * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES + (AFF ? 32768 : 0), s, g, -m_tiles, n_tiles);
    else
        hipLaunchKernelGGL((gemm_f16x3_v10_kernel<ABL, AFF, OVL, TERMS, NBS, TA>), dim3((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES + (AFF ? 32768 : 0), s, g, m_tiles, n_tiles);
    return check_launch("ogmm_gemm_nt(f16x3 v10)");
}

}  // namespace ogmm

#ifdef OGMM_ABLATIONS          // tools-only build (libogmm_probe.so): the product library carries neither the ablation instantiations nor this symbol
// diagnostic (tools/gemm_v6_check.py): read and clear the clock probe {shader cycles, 100 MHz wall ticks, workgroups}
extern "C" int ogmm_debug_v10_probe(unsigned long long* host3) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_v10_probe), 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v10_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif

namespace ogmm {

int gemm_nt_f16x3_v10(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {
#ifdef OGMM_ABLATIONS
        case 111: return launch_v10<8>(g, s);                    // no output stores
        case 112: return launch_v10<2048>(g, s);                 // clock probe
        case 113: return launch_v10<2048 + 8>(g, s);             // clock probe, no stores
        case 114: return launch_v10<2048 + 8 + 1>(g, s);         //   no DMA after the prologue
        case 115: return launch_v10<2048 + 8 + 2>(g, s);         //   no split arithmetic in the loop (stale fragments)
        case 116: return launch_v10<2048 + 8 + 4>(g, s);         //   no weight-fragment reads in the loop
        case 117: return launch_v10<2048 + 8 + 2 + 4 + 16>(g, s);    //   DMA + MFMA + barrier only
        case 118: return launch_v10<2048 + 8 + 1 + 2 + 4 + 16>(g, s);    //   MFMA + barrier only
        case 119: return launch_v10<2048 + 8 + 1 + 2>(g, s);     //   fragment reads + MFMA (no DMA, no split)
        case 120: return launch_v10<2048, false, false, 2>(g, s);          // clock probe, two-term form
        case 121: return launch_v10<2048 + 8, false, false, 2>(g, s);      //   no stores
#endif
        default:
            if (g.ovl_rowpart) return g.terms == 1 ? launch_v10<0, false, true, 1>(g, s) : g.terms == 2 ? launch_v10<0, false, true, 2>(g, s) : launch_v10<0, false, true>(g, s);
            if (g.nb_mean) return launch_v10<0, false, false, 3, true>(g, s);          // normalisation-backward fusion (training): its own instantiation
            if (g.a_trans) {
                if (g.terms == 2) return g.a_colsum ? launch_v10<0, false, false, 2, false, 2>(g, s) : launch_v10<0, false, false, 2, false, 1>(g, s);
                return g.a_colsum ? launch_v10<0, false, false, 3, false, 2>(g, s) : launch_v10<0, false, false, 3, false, 1>(g, s);
            }   // transposed A (training: the weight gradient's dY^T read as dY lies)
            if (g.a_scale) return launch_v10<0, true>(g, s);          // (the InstanceNorm-on-A form has no reduced variant: terms is a permission, not an order)
            return g.terms == 1 ? launch_v10<0, false, false, 1>(g, s) : g.terms == 2 ? launch_v10<0, false, false, 2>(g, s) : launch_v10<0>(g, s);
    }
}

}  // namespace ogmm
