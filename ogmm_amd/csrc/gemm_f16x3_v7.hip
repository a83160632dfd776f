// fp16x3 split GEMM, LDS-DMA structure with fragment reads carried ACROSS the barriers (third form of the large-shape engine).
//
// gemm_f16x3_v6.hip showed (clock probes inside the kernel, tools/gemm_v6_check.py --clock; 131072 x 1024 x 1024, cycles per
// 256 x 256 tile of which the matrix pipe needs 98 304):
//     MFMA + barrier                    101 008     97 % busy
//     + LDS-DMA of both operands        112 181     88 %
//     + fragment reads and split        129 310     76 %   (no DMA)
//     everything but the stores         146 880     67 %
// i.e. the cost is not the traffic but two exposed latencies per K step: every wave starts a step with ds_read -> wait -> 40 VALU
// (split) -> first MFMA, both waves of a SIMD at the same time, and stage t+1's weights are requested only one step before the barrier
// that needs them.  This form removes both:
//   * two barriers per K step (before MFMA group 0 and before group 4), and everything a wave reads in the first groups after a barrier
//     is guaranteed by the PREVIOUS barrier, so those reads (and the split of the next step's first fragments) are issued in the shadow of
//     the preceding groups' MFMAs;
//   * every DMA piece is in flight for at least one full K step: activations in 3 stages of [256][32] fp32 (as v6), weights in FOUR
//     half stages of one k16 block each ([8 column blocks][hi, lo][64 lanes][16 B] = 16 KiB), 160 KiB in all.
// Schedule of a wave's eight DMA instructions in step t (one per MFMA group g; stage tau of A has pieces p0..p3 = 8-row groups of the
// wave's 32 rows, half (tau, s) of B has the pieces hi, lo of the wave's column block):
//     g0, g1: B(t+1, 1) hi, lo      g2, g3: A(t+2) p2, p3      -- after barrier a_t: frees B(t-1, 1) and A(t-1)
//     g4, g5: B(t+2, 0) hi, lo      g6, g7: A(t+3) p0, p1      -- after barrier b_t: frees B(t, 0) and A(t)
// and what the barriers guarantee (every wave waits for its own pieces with a counted vmcnt first):
//     a_t: everything issued up to g1 of step t-1  -> B(t, 1), read from g3 on                      (vmcnt(6): g2..g7 of t-1 stay in flight)
//     b_t: everything issued up to g5 of step t-1  -> A(t+1), B(t+1, 0), read from g5 / g7 on      (vmcnt(6): g6, g7 of t-1, g0..g3 of t)
// Arithmetic and result are those of v4 / v6 (same products in the same order: bit-identical output).
#include <cstdlib>
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK7 = 32;
constexpr int MT = 2, NT = 4, WM = 4, WN = 2;
constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;      // 256, 256, 512
constexpr int A_STAGE = BM * BK7 * 4;                                        // 32768 B
constexpr int B_HALF = BN * 16 * 2 * 2;                                      // 16384 B: one k16 block, hi + lo
constexpr int A_STAGES = 3, B_HALVES = 4;
constexpr int B_OFF = A_STAGES * A_STAGE;
constexpr int LDS_BYTES = A_STAGES * A_STAGE + B_HALVES * B_HALF;           // 163840 B

__device__ unsigned long long g_v7_probe[4];          // clock probe, see gemm_f16x3_v6.hip

template <int ABL>
__global__ __launch_bounds__(T) void gemm_f16x3_v7_kernel(const ogmm_gemm g, const int m_tiles_signed, const int n_tiles, const int direct_stores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];

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
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int zb = blockIdx.z;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int m_end = min(g.M, m0 + BM);
    const int nk1 = g.K1 / BK7, nk2 = g.K2 / BK7, nk = nk1 + nk2;

    // ---- DMA sources.  A: wave w stages rows [32 w, 32 w + 32), piece i rows 8 i .. 8 i + 7, lane l -> row (l >> 3), LDS chunk (l & 7)
    // <- global chunk (l & 7) ^ ((row >> 1) & 7) (the read applies the same XOR).  Rows beyond M are clamped (their results are never stored).
    const float* __restrict__ A1p = g.A + zb * g.sA_o + (int64_t)m0 * g.lda;
    const float* __restrict__ A2p = g.A2 ? g.A2 + zb * g.sA2_o + (int64_t)m0 * g.lda2 : nullptr;
    const unsigned lds0 = (unsigned)(size_t)smem7;
    unsigned aoff[4];          // byte offset of this lane's 16 bytes in each of its four pieces, relative to the stage's first element
    auto set_aoff = [&](int ld) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 32 + i * 8 + (lane >> 3);
            aoff[i] = (unsigned)(min(r, g.M - 1 - m0) * ld + ((lane & 7) ^ ((r >> 1) & 7)) * 4) * 4u;
        }
    };
    set_aoff((int)g.lda);
    // B: wave w stages column block w of every half stage: pieces hi, lo (1 KiB fragments of the fragment-major image)
    const int KB = (int)(g.ldb_h / 16);
    const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_hi) + zb * g.sB_o) + ((int64_t)(n0 / 32 + wave) * KB) * 64;
    const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_lo) + zb * g.sB_o) + ((int64_t)(n0 / 32 + wave) * KB) * 64;
    const unsigned boff = lane * 16;

    auto issue_a_piece = [&](int tau, int i) {          // (inline-assembly DMA: gemm_common.h)
        const bool second = tau >= nk1;
        if (i == 0 && tau == nk1 && nk2 > 0) set_aoff((int)g.lda2);          // stages are issued in order and piece 0 first: switch to the second A piece once
        const float* Ap = second ? A2p + (tau - nk1) * BK7 : A1p + tau * BK7;
        lds_dma16(aoff[i], Ap, lds0 + (tau % A_STAGES) * A_STAGE + wave * 4096 + i * 1024);
    };
    auto issue_b_piece = [&](int tau, int s, int plane) {
        const int kb = ((tau < nk1 ? tau * 2 : (g.K1 / 16) + (tau - nk1) * 2) + s) * 64;
        lds_dma16(boff, (plane ? BL : BH) + kb, lds0 + B_OFF + ((2 * tau + s) & 3) * B_HALF + (wave * 2 + plane) * 1024);
    };
    // the DMA piece that the schedule places in step st at group grp (st may be negative: the prologue walks steps -3 .. -1)
    auto dma_dyn = [&](int st, int grp) {
        if (grp < 2) { if (st + 1 >= 0 && st + 1 < nk) issue_b_piece(st + 1, 1, grp); }
        else if (grp < 4) { if (st + 2 >= 0 && st + 2 < nk) issue_a_piece(st + 2, grp); }
        else if (grp < 6) { if (st + 2 >= 0 && st + 2 < nk) issue_b_piece(st + 2, 0, grp - 4); }
        else { if (st + 3 >= 0 && st + 3 < nk) issue_a_piece(st + 3, grp - 6); }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets: A row (wm*64 + u*32 + lr), chunk (s*4 + lh*2 + q) ^ ((lr >> 1) & 7); B column block (wn*4 + j)
    const int a_rd = (wm * 64 + lr) * 128;
    const int a_sw = (lr >> 1) & 7;
    const int a_c0 = ((lh * 2) ^ a_sw) << 4, a_c1 = ((lh * 2 + 1) ^ a_sw) << 4;        // k16 block 0; block 1 = byte offset ^ 64
    const int b_rd = wn * 4 * 2048 + lane * 16;
    float ovf = 0.0f;          // += hi . hi per pair of split values: inf / nan iff some |a| > 65504 (binary16 overflow flag)

    f32x4 ra[MT][2];
    f16x8 ah[2][MT], al[2][MT];            // [k16 block][row block]
    f16x8 bh[2], bl[2];                    // [group parity]
    auto read_a = [&](int tau, int s) {
        const unsigned char* As = smem7 + (tau % A_STAGES) * A_STAGE + a_rd;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            ra[u][0] = *reinterpret_cast<const f32x4*>(As + u * 4096 + (a_c0 ^ (s * 64)));
            ra[u][1] = *reinterpret_cast<const f32x4*>(As + u * 4096 + (a_c1 ^ (s * 64)));
        }
    };
    auto split_a = [&](int s, int u) {
        f16x4 h0, l0, h1, l1;
        split4_f16_pure(ra[u][0], h0, l0, ovf);
        split4_f16_pure(ra[u][1], h1, l1, ovf);
        ah[s][u] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        al[s][u] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    };
    auto read_b = [&](int tau, int grp) {          // fragments of MFMA group grp (k16 block grp >> 2, column block grp & 3) into the registers of its parity
        const int s = grp >> 2, j = grp & 3;
        const unsigned char* Bs = smem7 + B_OFF + ((2 * tau + s) & 3) * B_HALF + b_rd + j * 2048;
        bh[grp & 1] = *reinterpret_cast<const f16x8*>(Bs);
        bl[grp & 1] = *reinterpret_cast<const f16x8*>(Bs + 1024);
    };

    // ---- prologue: the pieces that the schedule issues in steps -3, -2, -1 (in that order: the counted waits below rely on it)
#pragma unroll
    for (int st = -3; st < 0; ++st)
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) { dma_dyn(st, grp); __builtin_amdgcn_sched_barrier(0); }
    // A(0) and B(0, 0) landed: younger pieces are A(1) p0 p1, B(0,1), A(1) p2 p3, B(1,0), A(2) p0 p1
    if (nk >= 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (nk == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(0, 0);
    read_b(0, 0);
    split_a(0, 0);
    split_a(0, 1);

    // One K step; REM = min(3, steps remaining including this one) resolves which DMA pieces still exist and the counted waits
    auto step = [&](int t, auto rem_c) {
        constexpr int REM = decltype(rem_c)::value;
        constexpr bool N1 = REM >= 2, N2 = REM >= 3;          // stage t+1 / t+2 exist (t+3 is tested at run time: one scalar branch per step)
        // barrier a_t: B(t, 1) landed; in flight stay g2..g7 of step t-1 = A(t+1) p2 p3, B(t+1, 0), A(t+2) p0 p1
        if (N2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (N1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool n3 = t + 3 < nk;
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int s = grp >> 2, j = grp & 3, p = grp & 1;
            if (grp == 4) {
                // barrier b_t: A(t+1) and B(t+1, 0) landed; in flight stay g6, g7 of step t-1 and g0..g3 of this step
                __builtin_amdgcn_sched_barrier(0);
                if (N2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (N1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) {
                if (grp < 2) { if (N1) issue_b_piece(t + 1, 1, grp); }
                else if (grp < 4) { if (N2) issue_a_piece(t + 2, grp); }
                else if (grp < 6) { if (N2) issue_b_piece(t + 2, 0, grp - 4); }
                else { if (N2 && n3) issue_a_piece(t + 3, grp - 6); }
            }
            if (grp == 1) read_a(t, 1);                        // raw fragments of k16 block 1 (ra is free: block 0 was split in the previous step)
            if (grp == 2) split_a(1, 0);                       // VALU in the shadow of this group's MFMAs
            if (grp == 3) split_a(1, 1);
            if (grp < 7) read_b(t, grp + 1);
            if (N1) {                                          // the next step's first fragments: guaranteed by barrier b_t
                if (grp == 5) read_a(t + 1, 0);
                if (grp == 6) split_a(0, 0);
                if (grp == 7) { split_a(0, 1); read_b(t + 1, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < MT; ++u) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][u], bh[p], acc[u][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < MT; ++u) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][u], bl[p], acc[u][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < MT; ++u) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][u], bh[p], acc[u][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        int t = 0;
        for (; t + 2 < nk; ++t) step(t, std::integral_constant<int, 3>{});
        if (t + 1 < nk) { step(t, std::integral_constant<int, 2>{}); ++t; }
        step(t, std::integral_constant<int, 1>{});
    }
    if ((ABL & 2048) && threadIdx.x == 0) {
        atomicAdd(&g_v7_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_v7_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
        atomicAdd(&g_v7_probe[2], 1ull);
    }
    if (g.overflow && !(fabsf(ovf) <= 3.0e38f)) atomicOr(g.overflow, 1);
    __builtin_amdgcn_s_barrier();          // every wave is done with the last stage (all DMA landed: vmcnt(0) above): LDS becomes the epilogue's patch
    if (ABL & 8) {          // ablation: no output stores
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 1.2345f) g.C[0] = sum;
        return;
    }
    ogmm_gemm gz = g;
    if (gz.C) gz.C += zb * g.sC_o;
    if (gz.Res) gz.Res += zb * g.sR_o;
    if (wide_epilogue_ok(g)) gemm_epilogue_wide<MT, NT, WM, WN>(gz, acc, reinterpret_cast<float*>(smem7), m0, n0, m_end, g.alpha, direct_stores != 0);
    else gemm_epilogue<MT, NT, WM, WN, false>(gz, acc, reinterpret_cast<float*>(smem7), m0, n0, m_end, 0, 0, g.alpha);
}

}  // namespace

// diagnostic (tools/gemm_v6_check.py): read and clear the clock probe {shader cycles, 100 MHz wall ticks, workgroups}
extern "C" int ogmm_debug_v7_probe(unsigned long long* host3) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_v7_probe), 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v7_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}

namespace ogmm {

bool gemm_f16x3_v7_applicable(const ogmm_gemm& g) {
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch_outer;
    static const int enabled = [] { const char* e = getenv("OGMM_V7"); return e ? atoi(e) : 1; }();
    static const long long min_tiles = [] { const char* e = getenv("OGMM_V7_MIN_TILES"); return e ? atoll(e) : 256LL; }();
    return enabled && g.pool_k == 0 && !g.a_scale && g.N >= 256 && tiles >= min_tiles && g.K1 % BK7 == 0 && g.K2 % BK7 == 0 && g.ldb_h % 64 == 0 &&
           (g.K2 == 0 || g.K1 % 64 == 0) && (g.K1 + 63) / 64 * 64 + (g.K2 + 63) / 64 * 64 <= g.ldb_h && (g.lda % 4) == 0 && (g.K2 == 0 || (g.lda2 % 4) == 0);
}

template <int ABL>
static int launch_v7(const ogmm_gemm& g, hipStream_t s) {
    const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    static const int direct = [] { const char* e = getenv("OGMM_V7_DIRECT"); return e ? atoi(e) : 1; }();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v7_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (m_tiles % 8 != 0 && m_tiles < 32)
        hipLaunchKernelGGL(gemm_f16x3_v7_kernel<ABL>, dim3((unsigned)(m_tiles * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES, s, g, -m_tiles, n_tiles, direct);
    else
        hipLaunchKernelGGL(gemm_f16x3_v7_kernel<ABL>, dim3((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES, s, g, m_tiles, n_tiles, direct);
    return check_launch("ogmm_gemm_nt(f16x3 v7)");
}

int gemm_nt_f16x3_v7(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {
        case 91: return launch_v7<8>(g, s);                 // no output stores
        case 92: return launch_v7<2048>(g, s);              // clock probe
        case 93: return launch_v7<2048 + 8>(g, s);          // clock probe, no stores
        case 94: return launch_v7<8 + 1>(g, s);             // no stores, no DMA after the prologue
        default: return launch_v7<0>(g, s);
    }
}

}  // namespace ogmm
