// fp16x3 split GEMM, LDS-DMA structure, 8 (M) x 1 (N) waves: a wave owns 32 rows x all 256 columns of the 256 x 256 tile.
//
// Same operand path as gemm_f16x3_v6.hip (both operands by global_load_lds_dwordx4, activations as stages of [256][32] fp32 with the XOR chunk
// swizzle, weights 2 stages of the fragment-major image, split of A in registers after the fragment read) -- but v6's 4 x 2 wave layout splits
// every activation TWICE (the two waves that share a row block) and its loop turned out to be bound by instruction issue between the matrix
// instructions, not by the operand traffic (PMC: SQ_ACTIVE_INST_VALU x3 against the MFMA-only loop, matrix pipe 65 % busy; clock probes:
// MFMA + barrier 101 k cycles per tile, + DMA 112 k, + fragment reads and split 129 k, all 147 k).  With one row block per wave
//   * every activation is split once: 40 VALU per wave and K step instead of 96;
//   * a wave reads only the activation rows it staged itself (A needs no cross-wave ordering at all), and all eight weight column blocks:
//     36 ds_read_b128 per step instead of 24 (288 KiB per CU and step, 37 % of the LDS read rate);
//   * the MFMA order alternates two accumulators (column blocks 2q, 2q+1) exactly as v4 / v6 alternate two row blocks: same products,
//     same order per accumulator -> bit-identical output.
#include <cstdlib>
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK8 = 32;
constexpr int MT = 1, NT = 8, WM = 8, WN = 1;
constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;      // 256, 256, 512
constexpr int A_STAGE = BM * BK8 * 4;                                        // 32768 B
constexpr int B_STAGE = BN * BK8 * 2 * 2;                                    // 32768 B
// Two stages per operand (128 KiB): the activations are wave-private here, so the slot of stage t is free as soon as the wave has read its second
// k16 block (group 1 of step t) and takes stage t+2 in groups 4-7 of the same step -- v6 needs a third stage because other waves may still be reading.
// (Measured: 2 stages 121 k cycles per tile, 3 stages 123 k; a THIRD WEIGHT stage with the weights requested two steps ahead: 131 k -- slower.)
// The InstanceNorm-fusing form (AFF) keeps the tile's [scale | shift] table of the A transform in the 32 KiB behind the rings.
constexpr int A_STAGES = 2, B_STAGES = 2;
constexpr int AFF_OFF = A_STAGES * A_STAGE + B_STAGES * B_STAGE, AFF_MAX_K = 4096;
constexpr int B_OFF = A_STAGES * A_STAGE;
constexpr int LDS_BYTES = A_STAGES * A_STAGE + B_STAGES * B_STAGE;          // 131072 B (+ 32768 B with AFF)

// clock probe (ablation 2048): every workgroup adds its duration in shader cycles (s_memtime) and in 100 MHz wall ticks: the ratio is the
// shader clock the kernel actually ran at (the chip's power management picks it per workload; rocprofv3 pins it, so counters cannot tell)
__device__ unsigned long long g_v8_probe[4];

// AFF: A is read as relu?(a * a_scale[group][k] + a_shift[group][k]) (InstanceNorm of the producing layer, models/attn.py:24-25), applied to the raw
// fragment right after the ds_read, with the constants of the tile's row group staged once in LDS (a half wave reads the same 8 k: broadcast reads).
// TERMS (struct ogmm_gemm.terms): 3 = lo*hi + hi*lo + hi*hi; 2 = lo*hi + hi*hi = (a_hi + a_lo) w_hi, the weight rounded to binary16 -- its lo plane is
// neither fetched (2 instead of 4 weight DMA instructions per step and wave) nor read from LDS.  Same schedule, groups of 2 TERMS matrix instructions.
template <int ABL, bool AFF, bool HEAD, int TERMS = 3>
__global__ __launch_bounds__(T) void gemm_f16x3_v8_kernel(const ogmm_gemm g, const int m_tiles_signed, const int n_tiles, const int direct_stores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

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

    // ---- DMA sources.  A: wave w stages rows [32 w, 32 w + 32), instruction i rows 8 i .. 8 i + 7, lane l -> row (l >> 3), LDS chunk (l & 7)
    // <- global chunk (l & 7) ^ ((row >> 1) & 7).  Rows beyond M are clamped (their results are never stored).
    const float* __restrict__ A1p = g.A + zb * g.sA_o + (int64_t)m0 * g.lda;
    const float* __restrict__ A2p = g.A2 ? g.A2 + zb * g.sA2_o + (int64_t)m0 * g.lda2 : nullptr;
    // byte offset of this lane's 16 bytes in each of its four pieces, relative to the stage's first element (row panel start + k0); one set per A piece
    const unsigned lds0 = (unsigned)(size_t)smem8;
    unsigned aoff[4];
    auto set_aoff = [&](int ld) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 32 + i * 8 + (lane >> 3);
            aoff[i] = (unsigned)(min(r, g.M - 1 - m0) * ld + ((lane & 7) ^ ((r >> 1) & 7)) * 4) * 4u;
        }
    };
    set_aoff((int)g.lda);
    // B: wave w stages column block w: instruction i = (k16 = i >> 1, plane = i & 1); 1 KiB fragments of the fragment-major image
    const int KB = (int)(g.ldb_h / 16);
    const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_hi) + zb * g.sB_o) + ((int64_t)(n0 / 32 + wave) * KB) * 64;
    const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_lo) + zb * g.sB_o) + ((int64_t)(n0 / 32 + wave) * KB) * 64;
    const unsigned boff = lane * 16;

    // one DMA instruction (1 KiB) of stage t: piece i of this wave's four activation row groups / four weight fragments
    auto issue_a_piece = [&](int t, int i) {
        const bool second = t >= nk1;
        if (i == 0 && t == nk1 && nk2 > 0) set_aoff((int)g.lda2);          // stages are issued in order and piece 0 first: switch to the second A piece once
        const float* Ap = second ? A2p + (t - nk1) * BK8 : A1p + t * BK8;
        lds_dma16(aoff[i], Ap, lds0 + (t % A_STAGES) * A_STAGE + wave * 4096 + i * 1024);
    };
    auto issue_b_piece = [&](int t, int i) {
        const int kb = (t < nk1 ? t * 2 : (g.K1 / 16) + (t - nk1) * 2) * 64;
        lds_dma16(boff, ((i & 1) ? BL : BH) + kb + (i >> 1) * 64, lds0 + B_OFF + (t % B_STAGES) * B_STAGE + wave * 4096 + i * 1024);
    };
    auto issue_a = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_a_piece(t, i);
    };
    auto issue_b = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; i += (TERMS == 3 ? 1 : 2)) issue_b_piece(t, i);          // (even pieces: the hi plane)
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // fragment read offsets: A row (wave*32 + lr) -- the rows this wave staged itself --, chunk (s*4 + lh*2 + q) ^ ((lr >> 1) & 7); B: all column blocks
    const int a_rd = (wave * 32 + lr) * 128;
    const int a_sw = (lr >> 1) & 7;
    const int a_c0 = ((lh * 2) ^ a_sw) << 4, a_c1 = ((lh * 2 + 1) ^ a_sw) << 4;        // k16 step 0; step 1 = chunk ^ 4 = byte offset ^ 64
    const int b_rd = lane * 16;
    float ovf = 0.0f;          // += hi . hi per pair of split values: becomes inf / nan iff some |a| > 65504 (binary16 overflow flag)

    f32x4 ra[2];
    f16x8 ah[2], al[2];                    // [k16 block]
    f16x8 bh[2][2], bl[2][2];              // [group parity][column block of the pair]
    f32x4 rsc[2], rsh[2];          // AFF: the 8 scales / shifts of the fragment's k positions
    const float aff_lo = (AFF && g.a_relu) ? 0.0f : -__builtin_inff();
    auto read_a = [&](int tau, int s) {          // raw fp32 fragment of this wave's rows, k16 block s of stage tau
        const unsigned char* As = smem8 + (tau % A_STAGES) * A_STAGE + a_rd;
        ra[0] = *reinterpret_cast<const f32x4*>(As + (a_c0 ^ (s * 64)));
        ra[1] = *reinterpret_cast<const f32x4*>(As + (a_c1 ^ (s * 64)));
        if (AFF) {
            const int Kt = g.K1 + g.K2;
            const int k0 = (tau < nk1 ? tau * BK8 : g.K1 + (tau - nk1) * BK8) + s * 16 + lh * 8;
            const float* tab = reinterpret_cast<const float*>(smem8 + AFF_OFF) + k0;
            rsc[0] = *reinterpret_cast<const f32x4*>(tab);
            rsc[1] = *reinterpret_cast<const f32x4*>(tab + 4);
            rsh[0] = *reinterpret_cast<const f32x4*>(tab + Kt);
            rsh[1] = *reinterpret_cast<const f32x4*>(tab + Kt + 4);
        }
    };
    auto split_a = [&](int s) {
        if (AFF) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[h][e] = fmaxf(fmaf(ra[h][e], rsc[h][e], rsh[h][e]), aff_lo);
        }
        f16x4 h0, l0, h1, l1;
        split4_f16_pure(ra[0], h0, l0, ovf);
        split4_f16_pure(ra[1], h1, l1, ovf);
        ah[s] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        al[s] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    };
    auto read_b = [&](int tau, int grp) {          // MFMA group grp = k16 block grp >> 2, column blocks 2q, 2q+1 with q = grp & 3
        const unsigned char* Bs = smem8 + B_OFF + (tau % B_STAGES) * B_STAGE + b_rd;
        const int s = grp >> 2, q = grp & 3;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bh[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 0) * 1024);
            if (TERMS == 3) bl[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 1) * 1024);
        }
    };

    // ---- prologue.  DMA order B(0), A(0), A(1): the counted waits below rely on it.
    issue_b(0);
    issue_a(0);
    if (nk > 1) issue_a(1);
    if (AFF) {          // the tile's rows belong to one group (group_rows is a multiple of the tile): its K scales, then its K shifts
        const int Kt = g.K1 + g.K2;
        const float* __restrict__ sc = g.a_scale + (int64_t)(m0 / g.group_rows) * Kt;
        const float* __restrict__ sh = g.a_shift + (int64_t)(m0 / g.group_rows) * Kt;
        float* tab = reinterpret_cast<float*>(smem8 + AFF_OFF);
        for (int i = tid; i < Kt; i += T) { tab[i] = sc[i]; tab[Kt + i] = sh[i]; }
        __syncthreads();          // (the compiler's wait for these loads also drains the DMA issued above: it is needed right below anyway)
    }
    // The activation rows a wave reads are the rows it staged itself: its own vmcnt orders them, no barrier.  A(0) landed (A(1) may be in flight):
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_a(0, 0);
    split_a(0);

    // One K step = 8 MFMA groups of 6 (k16 block s = grp >> 2 against the column-block pair q = grp & 3).  A wave's eight DMA instructions of the
    // step (weights of stage t+1 in groups 0-3, activations of stage t+2 in groups 4-7) are issued one per group: eight back to back fill the
    // CU's vector-memory queue and hold every wave in the issue of its own DMA instructions.  The fragments of group g+1 are read before group
    // g's MFMAs; the first activation fragment of the NEXT step is read and split in groups 5-6 (wave-private data: see above), so that a step
    // starts with the weight fragments of group 0 as its only exposed LDS latency.
    auto step = [&](int t, auto has_b_c, auto has_a_c) {
        constexpr bool HAS_B = decltype(has_b_c)::value, HAS_A = decltype(has_a_c)::value;          // stage t+1 / t+2 exist
        // this wave's weight pieces of stage t landed (the 4 youngest DMA instructions, activations of stage t+1, may stay in flight); the barrier
        // makes all eight waves' pieces visible and tells that everybody is done reading stage t-1
        if (HAS_B) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_b(t, 0);
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int s = grp >> 2, q = grp & 3, p = grp & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) {
                if (grp < 4) { if (HAS_B && (TERMS == 3 || !(grp & 1))) issue_b_piece(t + 1, grp); }
                else { if (HAS_A) issue_a_piece(t + 2, grp - 4); }
            }
            if (grp == 1) read_a(t, 1);                        // raw fragment of k16 block 1 (ra is free: block 0 was split in the previous step)
            if (grp + 1 < 8) read_b(t, grp + 1);
            if (grp == 2) split_a(1);                          // VALU in the shadow of this group's MFMAs
            if (HAS_B && grp == 5) {
                // own activation pieces of stage t+1 landed: younger are the 4 weight pieces of this step and the activation pieces of groups 4, 5
                if (!(ABL & 1)) {
                    if (TERMS == 3) { if (HAS_A) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                    else { if (HAS_A) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }          // 2 weight pieces per step
                }
                read_a(t + 1, 0);
            }
            if (HAS_B && grp == 6) split_a(0);                 // ah[0] / al[0] were last used by group 3
            __builtin_amdgcn_sched_barrier(0);
            // the two accumulators of the pair alternate (as v4 / v6 alternate two row blocks); per accumulator: lo*hi, hi*lo, hi*hi
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[p][c], acc[2 * q + c], 0, 0, 0);
            if (TERMS == 3) {
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[p][c], acc[2 * q + c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[p][c], acc[2 * q + c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        int t = 0;
        for (; t + 2 < nk; ++t) step(t, std::true_type{}, std::true_type{});
        if (t + 1 < nk) { step(t, std::true_type{}, std::false_type{}); ++t; }
        step(t, std::false_type{}, std::false_type{});
    }
    if ((ABL & 2048) && threadIdx.x == 0) {
        atomicAdd(&g_v8_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_v8_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
        atomicAdd(&g_v8_probe[2], 1ull);
    }
    if (g.overflow && !(fabsf(ovf) <= 3.0e38f)) atomicOr(g.overflow, 1);
    __builtin_amdgcn_s_barrier();          // every wave is done with the last stage: LDS becomes the epilogue's transposition patch
    if (ABL & 8) {          // ablation: no output stores
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[j][r];
        if (sum == 1.2345f) g.C[0] = sum;
        return;
    }
    ogmm_gemm gz = g;
    if (gz.C) gz.C += zb * g.sC_o;
    if (gz.Res) gz.Res += zb * g.sR_o;
    // a wave's 32 x 256 slab: straight from the accumulators when it lies inside the matrix (the common case), else the general per-element form
    const bool inside = m0 + BM <= m_end && n0 + BN <= g.N && !g.row_affine;
    if constexpr (HEAD) {          // a Cout = 1 head behind this layer (N == 256, whole tiles: gemm_f16x3_v8_applicable); its own instantiation --
        gemm_epilogue_rowblock_rowdot<NT>(gz, acc, m0 + wave * 32, g.alpha);          // as a run-time branch it cost the default kernel 87 spilled registers
        return;
    }
    if (inside) {
        float* stat_lds = reinterpret_cast<float*>(smem8);          // [8 waves][256 columns][2]: the rings are dead (barrier above)
        gemm_epilogue_rowblock<NT>(gz, acc, m0 + wave * 32, n0, g.alpha, stat_lds);
        if (g.col_stats) {
            __syncthreads();
            // thread = (statistic, column): add the eight waves' partial sums (fp64), one atomic per column and statistic per tile
            const int c = tid & 255, which = tid >> 8;
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += reinterpret_cast<const double*>(stat_lds)[(w * 256 + c) * 2 + which];
            atomicAdd(g.col_stats + (int64_t)((m0 >> 8) & g.col_stats_slot_mask) * g.col_stats_slot_stride + ((int64_t)(m0 / g.group_rows) * g.N + n0 + c) * 2 + which, tot);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x16 pair[1][2] = {{acc[2 * q], acc[2 * q + 1]}};
            gemm_epilogue<1, 2, 8, 1, false>(gz, pair, reinterpret_cast<float*>(smem8), m0, n0 + q * 64, m_end, 0, 0, g.alpha);
        }
    }
}

}  // namespace

namespace ogmm {

bool gemm_f16x3_v8_applicable(const ogmm_gemm& g) {
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch_outer;
    static const int enabled = [] { const char* e = getenv("OGMM_V8"); return e ? atoi(e) : 1; }();
    static const long long min_tiles = [] { const char* e = getenv("OGMM_V8_MIN_TILES"); return e ? atoll(e) : 256LL; }();
    const bool whole_tiles = g.M % BM == 0 && g.N % BN == 0 && !g.row_affine;          // the statistics come out of the row-block epilogue only
    const bool rd_ok = !g.rd_out || (!g.a_scale && whole_tiles && g.N == BN && g.rd_w && g.rd_ld >= 1 && !g.col_stats && !g.ovl_rowpart && g.batch_outer * g.batch_inner == 1);
    return enabled && g.pool_k == 0 && (!g.col_stats || whole_tiles) && rd_ok && !g.ovl_rowpart && (!g.a_scale || (g.a_shift && g.group_rows > 0 && g.group_rows % BM == 0 && g.K1 + g.K2 <= AFF_MAX_K && (g.K1 + g.K2) % 4 == 0)) && g.N >= 256 && tiles >= min_tiles && g.K1 % BK8 == 0 && g.K2 % BK8 == 0 && g.ldb_h % 64 == 0 &&
           (g.K2 == 0 || g.K1 % 64 == 0) && (g.K1 + 63) / 64 * 64 + (g.K2 + 63) / 64 * 64 <= g.ldb_h && (g.lda % 4) == 0 && (g.K2 == 0 || (g.lda2 % 4) == 0);
}

template <int ABL, bool AFF = false, bool HEAD = false, int TERMS = 3>
static int launch_v8(const ogmm_gemm& g, hipStream_t s) {
    const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    static const int direct = [] { const char* e = getenv("OGMM_V8_DIRECT"); return e ? atoi(e) : 1; }();
    static ogmm::PerDeviceOnce attr_once;          // per template instance and device
    if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v8_kernel<ABL, AFF, HEAD, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + (AFF ? 32768 : 0));
    if (m_tiles % 8 != 0 && m_tiles < 32)
        hipLaunchKernelGGL((gemm_f16x3_v8_kernel<ABL, AFF, HEAD, TERMS>), dim3((unsigned)(m_tiles * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES + (AFF ? 32768 : 0), s, g, -m_tiles, n_tiles, direct);
    else
        hipLaunchKernelGGL((gemm_f16x3_v8_kernel<ABL, AFF, HEAD, TERMS>), dim3((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES + (AFF ? 32768 : 0), s, g, m_tiles, n_tiles, direct);
    return check_launch("ogmm_gemm_nt(f16x3 v8)");
}

}  // namespace ogmm

#ifdef OGMM_ABLATIONS          // tools-only build (libogmm_probe.so)
// diagnostic (tools/gemm_v6_check.py): read and clear the clock probe {shader cycles, 100 MHz wall ticks, workgroups}
extern "C" int ogmm_debug_v8_probe(unsigned long long* host3) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_v8_probe), 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v8_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif

namespace ogmm {

int gemm_nt_f16x3_v8(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {
#ifdef OGMM_ABLATIONS
        case 101: return launch_v8<8>(g, s);                    // no output stores
        case 102: return launch_v8<2048>(g, s);                 // clock probe
        case 103: return launch_v8<2048 + 8>(g, s);             // clock probe, no stores
        case 104: return launch_v8<2048 + 8 + 1>(g, s);         //   no DMA after the prologue
#endif
        default:
            if (g.rd_out) return g.terms == 2 ? launch_v8<0, false, true, 2>(g, s) : launch_v8<0, false, true>(g, s);
            if (g.a_scale) return launch_v8<0, true>(g, s);          // (terms is a permission: the InstanceNorm-on-A form runs all three)
            return g.terms == 2 ? launch_v8<0, false, false, 2>(g, s) : launch_v8<0>(g, s);
    }
}

}  // namespace ogmm
