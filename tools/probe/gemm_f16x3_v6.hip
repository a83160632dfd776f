// fp16x3 split GEMM, LDS-DMA structure: BOTH operands reach LDS by `global_load_lds_dwordx4` (no VGPR staging, no ds_write,
// no split arithmetic before the barrier), through rings that keep the activations two K steps and the weights one K step
// ahead of the matrix pipe; the fp32 -> (hi, lo) binary16 split of A happens in registers AFTER the fragment read.
//
// Why (HISTORY.md section 4): in the register-staged engine (gemm_f16x3_v4.hip) the activation loads ride the waves' in-order
// vector-memory queue next to the weight-fragment loads, their split + ds_write pass sits in front of the tile's only
// barrier, and the matrix pipe is ~50 % busy.  Here a wave's instruction stream between two barriers is 8 DMA issues,
// 24 ds_read_b128, ~80 VALU and 48 MFMAs, and nothing in it waits for HBM except the counted `s_waitcnt vmcnt(4)` in front
// of the barrier, which leaves the next activation stage in flight.
//
// Geometry: 256 x 256 tile, K step 32, 8 waves as 4 (M) x 2 (N): a wave owns 64 rows x 128 columns (2 x 4 accumulators).
//   A stage  = [256 rows][32 k] fp32, 128-byte rows, 16-byte chunks XOR-swizzled with (row >> 1) & 7 so that the 16 lanes of a
//              ds_read_b128 group (16 different rows, same k) hit 16 different 16-byte slots of the 256-byte bank row.  The DMA
//              writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE chunk (a permutation inside one
//              128-byte line: coalescing is unchanged) and again on the read.                    3 stages x 32 KiB
//   B stage  = the fragment-major weight image as it is in memory: [8 column blocks][2 k16][hi, lo][64 lanes][16 B]; every
//              DMA instruction copies one 1 KiB fragment, reads are lane-linear (conflict-free).  2 stages x 32 KiB
//   160 KiB of LDS in all: one workgroup per CU.
// Each A element is split by the two waves that share its row block (16 split4 per wave and K step, VALU in the MFMA shadow).
#include <cstdlib>
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK6 = 32;
constexpr int MT = 2, NT = 4, WM = 4, WN = 2;
constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;      // 256, 256, 512
constexpr int A_STAGE = BM * BK6 * 4;                                        // 32768 B
constexpr int B_STAGE = BN * BK6 * 2 * 2;                                    // 32768 B
constexpr int A_STAGES = 3, B_STAGES = 2;
constexpr int B_OFF = A_STAGES * A_STAGE;
constexpr int LDS_BYTES = A_STAGES * A_STAGE + B_STAGES * B_STAGE;          // 163840 B

// clock probe (ablation 2048): every workgroup adds its duration in shader cycles (s_memtime) and in 100 MHz wall ticks: the ratio is the
// shader clock the kernel actually ran at (the chip's power management picks it per workload; rocprofv3 pins it, so counters cannot tell)
__device__ unsigned long long g_v6_probe[4];

template <int ABL>
__global__ __launch_bounds__(T) void gemm_f16x3_v6_kernel(const ogmm_gemm g, const int m_tiles_signed, const int n_tiles, const int direct_stores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];

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
    const int nk1 = g.K1 / BK6, nk2 = g.K2 / BK6, nk = nk1 + nk2;

    // ---- DMA sources.  A: wave w stages rows [32 w, 32 w + 32), instruction i rows 8 i .. 8 i + 7, lane l -> row (l >> 3), LDS chunk (l & 7)
    // <- global chunk (l & 7) ^ ((row >> 1) & 7).  Rows beyond M are clamped (their results are never stored).
    const float* __restrict__ A1p = g.A + zb * g.sA_o + (int64_t)((ABL & 128) ? 0 : m0) * g.lda;          // ablation 128: every tile reads row panel 0 (L2-resident A)
    const float* __restrict__ A2p = g.A2 ? g.A2 + zb * g.sA2_o + (int64_t)m0 * g.lda2 : nullptr;
    // byte offset of this lane's 16 bytes in each of its four pieces, relative to the stage's first element (row panel start + k0); one set per A piece
    const unsigned lds0 = (unsigned)(size_t)smem6;
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

    // one DMA instruction (1 KiB) of stage t: piece i of this wave's four activation row groups / four weight fragments (inline assembly: gemm_common.h)
    auto issue_a_piece = [&](int t, int i) {
        const bool second = t >= nk1;
        if (i == 0 && t == nk1 && nk2 > 0) set_aoff((int)g.lda2);          // stages are issued in order and piece 0 first: switch to the second A piece once
        const float* Ap = second ? A2p + (t - nk1) * BK6 : A1p + t * BK6;
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
        for (int i = 0; i < 4; ++i) issue_b_piece(t, i);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets: A row (wm*64 + u*32 + lr), chunk (s*4 + lh*2 + q) ^ ((lr >> 1) & 7); B block (wn*4 + j)
    const int a_rd = (wm * 64 + lr) * 128;
    const int a_sw = (lr >> 1) & 7;
    const int a_c0 = ((lh * 2) ^ a_sw) << 4, a_c1 = ((lh * 2 + 1) ^ a_sw) << 4;        // k16 step 0; step 1 = chunk ^ 4 = byte offset ^ 64
    const int b_rd = wn * 4 * 4096 + lane * 16;
    const int wpos = (0x30524130u >> (4 * wave)) & 7;          // waves 0..7 -> 0 3 1 4 2 5 0 3 (see ablation 1024)
    float ovf = 0.0f;          // += hi . hi per pair of split values: becomes inf / nan iff some |a| > 65504 (binary16 overflow flag)

    issue_b(0);
    issue_a(0);
    if (nk > 1) issue_a(1);

    f32x4 ra[MT][2];
    f16x8 ah[2][MT], al[2][MT];            // [k16 step][row block]
    f16x8 bh[2], bl[2];                    // [group parity]
    // One K step.  The eight DMA instructions of a wave (weights of stage t+1, activations of stage t+2) are issued ONE per MFMA group:
    // issued back to back after the barrier -- 64 KiB per CU at once -- they fill the CU's vector-memory queue and every wave sits in
    // the issue of its own DMA instructions until the address unit has taken them; measured, the operand traffic was then purely
    // additive to the matrix time (MFMA + barrier 0.53 ms, DMA + barrier 0.28 ms, both 0.76 ms at 131072 x 1024 x 1024).
    auto step = [&](int t, auto has_b_c, auto has_a_c) {
        constexpr bool HAS_B = decltype(has_b_c)::value, HAS_A = decltype(has_a_c)::value;          // stage t+1 / t+2 exist
        // stage t landed (this wave's share); stage t+1's activations (the 4 youngest DMA instructions) may stay in flight
        if (!(ABL & 4) || t == 0) {
            if (HAS_B && !(ABL & (32 | 64))) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if ((ABL & 512) && !(ABL & 1)) {          // ablation 512: the previous order (all eight DMA instructions right after the barrier)
            if (HAS_B && !(ABL & 32)) issue_b(t + 1);
            if (HAS_A && !(ABL & 64)) issue_a(t + 2);
        }
        const unsigned char* As = smem6 + (t % A_STAGES) * A_STAGE + a_rd;
        const unsigned char* Bs = smem6 + B_OFF + (t % B_STAGES) * B_STAGE + b_rd;

        auto read_a = [&](int s) {
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
        auto read_b = [&](int grp) {
            const int s = grp >> 2, j = grp & 3;
            bh[grp & 1] = *reinterpret_cast<const f16x8*>(Bs + ((j * 2 + s) * 2 + 0) * 1024);
            bl[grp & 1] = *reinterpret_cast<const f16x8*>(Bs + ((j * 2 + s) * 2 + 1) * 1024);
        };
        const bool frozen = (ABL & 2) && t > 0;          // ablation 2: the fragments of step 0 are reused by every later step (no LDS reads, no split)
        if (!frozen) {
            read_a(0);
            read_b(0);
            split_a(0, 0);
            split_a(0, 1);
            if (ABL & 2) { read_a(1); read_b(1); split_a(1, 0); split_a(1, 1); }
            if (ABL & 16) {          // ablation 16: all-zero operands (what a register-only MFMA loop measures when its fragments were never loaded)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bh[q] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; bl[q] = bh[q];
#pragma unroll
                    for (int u = 0; u < MT; ++u) { ah[q][u] = bh[q]; al[q][u] = bh[q]; }
                }
            }
        }
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int s = grp >> 2, j = grp & 3, p = grp & 1;
            __builtin_amdgcn_sched_barrier(0);
            auto dma = [&]() {
                if (grp < 4) { if (HAS_B && !(ABL & 32)) issue_b_piece(t + 1, grp); }
                else { if (HAS_A && !(ABL & 64)) issue_a_piece(t + 2, grp - 4); }
            };
            if (!(ABL & (1 | 512 | 1024))) dma();
            if (!(ABL & 2)) {
                if (grp == 1) read_a(1);                       // raw fragments of k16 step 1 (ra is free: step 0 is split)
                if (grp + 1 < 8) read_b(grp + 1);
                if (grp == 2) split_a(1, 0);                   // VALU in the shadow of this group's MFMAs
                if (grp == 3) split_a(1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ABL & 256) { if (ABL & 1024) dma(); continue; }            // ablation 256: no matrix instructions (what the operand path delivers on its own)
            // ablation 1024: the wave's DMA instruction of this group goes in front of MFMA number `wpos` (a per-wave constant), so that the eight
            // waves of the workgroup -- which leave the barrier together -- do not present their DMA instructions to the CU's address unit at once
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                if ((ABL & 1024) && !(ABL & 1) && wpos == m) { __builtin_amdgcn_sched_barrier(0); dma(); __builtin_amdgcn_sched_barrier(0); }
                const int u = m & 1;
                if (m < 2) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][u], bh[p], acc[u][j], 0, 0, 0);
                else if (m < 4) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][u], bl[p], acc[u][j], 0, 0, 0);
                else acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][u], bh[p], acc[u][j], 0, 0, 0);
            }
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
        atomicAdd(&g_v6_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_v6_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
        atomicAdd(&g_v6_probe[2], 1ull);
    }
    if (g.overflow && !(fabsf(ovf) <= 3.0e38f)) atomicOr(g.overflow, 1);
    __builtin_amdgcn_s_barrier();          // every wave is done with the last stage: LDS becomes the epilogue's transposition patch
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
    if (wide_epilogue_ok(g)) gemm_epilogue_wide<MT, NT, WM, WN>(gz, acc, reinterpret_cast<float*>(smem6), m0, n0, m_end, g.alpha, direct_stores != 0);
    else gemm_epilogue<MT, NT, WM, WN, false>(gz, acc, reinterpret_cast<float*>(smem6), m0, n0, m_end, 0, 0, g.alpha);
}

}  // namespace

namespace ogmm {

bool gemm_f16x3_v6_applicable(const ogmm_gemm& g) {
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch_outer;
    static const int enabled = [] { const char* e = getenv("OGMM_V6"); return e ? atoi(e) : 1; }();
    static const long long min_tiles = [] { const char* e = getenv("OGMM_V6_MIN_TILES"); return e ? atoll(e) : 256LL; }();
    return enabled && g.pool_k == 0 && !g.a_scale && g.N >= 256 && tiles >= min_tiles && g.K1 % BK6 == 0 && g.K2 % BK6 == 0 && g.ldb_h % 64 == 0 &&
           (g.K2 == 0 || g.K1 % 64 == 0) && (g.K1 + 63) / 64 * 64 + (g.K2 + 63) / 64 * 64 <= g.ldb_h && (g.lda % 4) == 0 && (g.K2 == 0 || (g.lda2 % 4) == 0);
}

template <int ABL>
static int launch_v6(const ogmm_gemm& g, hipStream_t s) {
    const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    static const int direct = [] { const char* e = getenv("OGMM_V6_DIRECT"); return e ? atoi(e) : 1; }();
    static ogmm::PerDeviceOnce attr_once;          // per template instance and device
    if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v6_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (m_tiles % 8 != 0 && m_tiles < 32)
        hipLaunchKernelGGL(gemm_f16x3_v6_kernel<ABL>, dim3((unsigned)(m_tiles * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES, s, g, -m_tiles, n_tiles, direct);
    else
        hipLaunchKernelGGL(gemm_f16x3_v6_kernel<ABL>, dim3((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS_BYTES, s, g, m_tiles, n_tiles, direct);
    return check_launch("ogmm_gemm_nt(f16x3 v6)");
}

}  // namespace ogmm

// diagnostic (tools/gemm_v6_check.py): read and clear the clock probe {shader cycles, 100 MHz wall ticks, workgroups}
extern "C" int ogmm_debug_v6_probe(unsigned long long* host3) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_v6_probe), 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v6_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}

namespace ogmm {

int gemm_nt_f16x3_v6(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {
        case 61: return launch_v6<8>(g, s);         // no output stores
        case 62: return launch_v6<8 + 1>(g, s);     // no stores, no DMA after the prologue
        case 63: return launch_v6<8 + 1 + 2>(g, s); // no stores, no DMA, fragments of step 0 reused (MFMA + barrier)
        case 64: return launch_v6<8 + 1 + 2 + 4>(g, s);          // MFMA only, operands = real data
        case 65: return launch_v6<8 + 1 + 2 + 4 + 16>(g, s);     // MFMA only, operands = zeros
        case 66: return launch_v6<8 + 2>(g, s);     // no stores, DMA running, no LDS reads / split
        case 67: return launch_v6<8 + 2 + 32>(g, s);         // same, activations only
        case 68: return launch_v6<8 + 2 + 64>(g, s);         // same, weights only
        case 69: return launch_v6<8 + 2 + 128>(g, s);        // same, activations from row panel 0 (L2-resident)
        case 70: return launch_v6<8 + 2 + 256>(g, s);        // DMA + barrier only (no MFMA, no LDS reads)
        case 71: return launch_v6<8 + 2 + 256 + 32>(g, s);   // DMA of the activations only
        case 72: return launch_v6<8 + 2 + 256 + 64>(g, s);   // DMA of the weights only
        case 73: return launch_v6<8 + 2 + 256 + 128>(g, s);  // DMA, activations from row panel 0
        case 74: return launch_v6<8 + 256>(g, s);            // DMA + LDS reads + split, no MFMA
        case 77: return launch_v6<1024>(g, s);               // DMA instructions staggered across the waves
        case 78: return launch_v6<1024 + 8>(g, s);           // same, no stores
        case 80: return launch_v6<2048>(g, s);               // clock probes: default
        case 81: return launch_v6<2048 + 8>(g, s);           //   no stores
        case 83: return launch_v6<2048 + 8 + 1 + 2>(g, s);   //   MFMA + barrier
        case 86: return launch_v6<2048 + 8 + 2>(g, s);       //   DMA + MFMA
        case 84: return launch_v6<2048 + 8 + 1 + 2 + 4>(g, s);          // MFMA only
        case 85: return launch_v6<2048 + 8 + 1 + 2 + 4 + 16>(g, s);     // MFMA only, zeros
        case 82: return launch_v6<2048 + 8 + 1>(g, s);       //   LDS reads + split + MFMA, no DMA
        case 75: return launch_v6<512>(g, s);                // all eight DMA instructions of a step right after the barrier (first form)
        case 76: return launch_v6<512 + 8>(g, s);            // same, no stores
        default: return launch_v6<0>(g, s);
    }
}

}  // namespace ogmm
