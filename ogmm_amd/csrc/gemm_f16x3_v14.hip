// fp16x3 split GEMM, PERSISTENT form of gemm_f16x3_v10.hip with the two row blocks of a wave HALF A K LOOP APART, so that the output of one can be
// stored while the other one keeps the matrix pipe busy (the store tail of DESIGN.md: 10 % of a K = 1024 launch, 25 % of a K = 512 one).
//
// A workgroup (4 waves, one per SIMD, 512 registers) owns one column tile (256 columns) and a list of row tiles; time runs in K steps of 32,
// tau = 0, 1, 2, ...; the weight stage in LDS at step tau is k block tau mod P (P = K / 32) -- the weights of the column tile cycle through LDS
// once per P steps.  Each wave runs two STREAMS (its row blocks 0 and 1, 32 rows x 256 columns, 128 accumulation registers each):
//   stream 0 starts a new row tile at tau = i P and sums k = 0 .. P-1 in order (bit-identical to the other engines);
//   stream 1 starts at tau = i P + P/2 and sums k = P/2 .. P-1, 0 .. P/2-1 (the same products, a rotated order of summation: fp32-rounding apart).
// Both use the weight stage of the moment, each stages its own activation rows (wave-private, as v10).  Every P/2 steps one stream finishes:
// its 128 values per lane move to 128 otherwise idle registers (scale / shift / activation applied) and go out as a few stores per MFMA
// group over the next steps, while the accumulators start the next row tile at once.
//
// Restrictions (else gemm_f16x3_v10.hip): plain epilogue (per-column scale / shift, none / ReLU / LeakyReLU), one A piece, K % 64 == 0, K >= 512,
// M, N multiples of 256, no batching, the row tiles divide evenly over the chip's 256 workgroup slots.
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
constexpr int B_OFF = 2 * A_STAGE;
constexpr int LDS_BYTES = 2 * A_STAGE + 2 * B_STAGE;                         // 131072 B
constexpr int WG_SLOTS = 256;                                                // one workgroup per CU

__device__ unsigned long long g_v14_probe[4];

template <int ABL>
__global__ __launch_bounds__(T) void gemm_f16x3_v14_kernel(const ogmm_gemm g, const int n_tiles, const int cnt, const int mgroups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem14[];

    const int bid = blockIdx.x;
    long long probe_c0 = 0, probe_w0 = 0;
    if (ABL & 2048) { probe_c0 = clock64(); probe_w0 = wall_clock64(); }
    // workgroup b runs on XCD b % 8; the n_tiles workgroups that share a row-tile list sit on one XCD (the activations are read from HBM once)
    const int xcd = bid & 7, local = bid >> 3;
    const int tile_n = local % n_tiles, mgroup = local / n_tiles;
    auto tile_m_of = [&](int i) { return (i * mgroups + mgroup) * 8 + xcd; };          // i-th row tile of this workgroup

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int n0 = tile_n * BN;
    const int P = g.K1 / BK8, half = P >> 1;

    // ---- DMA sources (as v10): A pieces 0-3 = rows of stream 0, 4-7 = rows of stream 1, each from its own row tile
    const unsigned lds0 = (unsigned)(size_t)smem14;
    unsigned aoff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = wave * 64 + i * 8 + (lane >> 3);
        aoff[i] = (unsigned)(r * (int)g.lda + ((lane & 7) ^ ((r >> 1) & 7)) * 4) * 4u;
    }
    const int KB = (int)(g.ldb_h / 16);
    const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(g.B_hi) + ((int64_t)(n0 / 32 + 2 * wave) * KB) * 64;
    const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(g.B_lo) + ((int64_t)(n0 / 32 + 2 * wave) * KB) * 64;
    const unsigned boff = lane * 16;

    // stage bookkeeping for the DMA two steps ahead (activations) / one step ahead (weights): k block and row tile of each stream at that step
    int ka = 0, ia0 = 0, ia1 = 0, seen_half = 0;          // state of "stage tau_a" while tau_a advances by one per call of advance_a()
    auto advance_a = [&]() {
        ++ka;
        if (ka == P) { ka = 0; ia0 = min(ia0 + 1, cnt - 1); }
        if (ka == half) { if (seen_half) ia1 = min(ia1 + 1, cnt - 1); seen_half = 1; }
    };
    auto issue_a_piece = [&](int slot, int i) {          // piece i of the stage described by (ka, ia0, ia1) into ring slot `slot`
        const int tm = tile_m_of(i < 4 ? ia0 : ia1);
        const float* Ap = g.A + (int64_t)tm * BM * g.lda + ka * BK8;
        lds_dma16(aoff[i], Ap, lds0 + slot * A_STAGE + wave * 8192 + i * 1024);
    };
    int kbn = 0;                                          // k block of the weight stage to be requested next
    auto issue_b_piece = [&](int slot, int i) {
        lds_dma16(boff, ((i & 1) ? BL : BH) + (int64_t)(i >> 2) * KB * 64 + kbn * 128 + ((i >> 1) & 1) * 64,
                  lds0 + B_OFF + slot * B_STAGE + (2 * wave + (i >> 2)) * 4096 + (i & 3) * 1024);
    };

    f32x16 acc0[NT], acc1[NT];          // stream 0 / 1 against the eight column blocks
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.0f; acc1[j][r] = 0.0f; }

    const int a_rd = (wave * 64 + lr) * 128;
    const int a_sw = (lr >> 1) & 7;
    const int a_c0 = ((lh * 2) ^ a_sw) << 4, a_c1 = ((lh * 2 + 1) ^ a_sw) << 4;
    const int b_rd = lane * 16;

    f32x4 ra[RB][2];
    f16x2 h01[4], h23[4], l01[4], l23[4];
    f16x4 ahh[RB][2][2], alh[RB][2][2];
    f16x8 bh[2][2], bl[2][2];
    auto read_a = [&](int slot, int s, int rb) {
        const unsigned char* As = smem14 + slot * A_STAGE + a_rd + rb * 4096;
        ra[rb][0] = *reinterpret_cast<const f32x4*>(As + (a_c0 ^ (s * 64)));
        ra[rb][1] = *reinterpret_cast<const f32x4*>(As + (a_c1 ^ (s * 64)));
    };
    auto piece_split = [&](int s, int which, int stage) {
        const f32x4 v = ra[which >> 1][which & 1];
        if (stage == 0)
            asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5" : "=&v"(h01[which]), "=&v"(h23[which]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        else if (stage == 1)
            asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %1, %3, -1.0, %5 op_sel_hi:[1,0,0]"
                : "=&v"(l01[which]), "=&v"(l23[which]) : "v"(h01[which]), "v"(h23[which]), "v"(v[0]), "v"(v[2]));
        else {
            asm("v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "+v"(l01[which]), "+v"(l23[which]) : "v"(h01[which]), "v"(h23[which]), "v"(v[1]), "v"(v[3]));
            ahh[which >> 1][s][which & 1] = f16x4{h01[which][0], h01[which][1], h23[which][0], h23[which][1]};
            alh[which >> 1][s][which & 1] = f16x4{l01[which][0], l01[which][1], l23[which][0], l23[which][1]};
        }
    };
    auto read_b = [&](int slot, int grp, int c) {
        const unsigned char* Bs = smem14 + B_OFF + slot * B_STAGE + b_rd;
        const int s = grp >> 2, q = grp & 3;
        bh[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 0) * 1024);
        bl[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 1) * 1024);
    };

    // ---- prologue.  DMA order B(0), A(0), A(1) (the counted waits rely on it); stream 1 has no row tile yet: it chews on row tile 0 until P/2
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_b_piece(0, i);
    kbn = 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a_piece(0, i);
    advance_a();
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a_piece(1, i);
    advance_a();          // (ka, ia0, ia1) now describe stage 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    read_a(0, 0, 0);
    read_a(0, 0, 1);
    read_b(0, 0, 0);
    read_b(0, 0, 1);
#pragma unroll
    for (int pi = 0; pi < 12; ++pi) piece_split(0, pi & 3, pi >> 2);

    // One K step (v10's steady-state step: 8 MFMA groups of 12, every other instruction in a fixed gap); sl = tau & 1 is the ring slot of the step
    auto step = [&](int sl) __attribute__((always_inline)) {
        const int ns = sl ^ 1;
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            const int s = grp >> 2, q = grp & 3, p = grp & 1, gl = grp & 3;
            const bool second = grp >= 4;
            const f16x8 ah0 = __builtin_shufflevector(ahh[0][s][0], ahh[0][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 al0 = __builtin_shufflevector(alh[0][s][0], alh[0][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 ah1 = __builtin_shufflevector(ahh[1][s][0], ahh[1][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 al1 = __builtin_shufflevector(alh[1][s][0], alh[1][s][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int prod = m >> 2, rb = (m >> 1) & 1, c = m & 1;
                const f16x8 av = prod == 0 ? (rb ? al1 : al0) : (rb ? ah1 : ah0);
                const f16x8 bv = prod == 1 ? bl[p][c] : bh[p][c];
                if (rb == 0) acc0[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc0[2 * q + c], 0, 0, 0);
                else acc1[2 * q + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc1[2 * q + c], 0, 0, 0);
                // ---- the gap after MFMA m
                if (m == 0 || m == 3) {
                    const int pc = 2 * gl + (m == 3);
                    if (!second) issue_b_piece(ns, pc);          // weights of the next step
                    else issue_a_piece(sl, pc);                  // activations two steps ahead: into this step's slot (its raw data is dead)
                }
                if (grp < 7 && (m == 1 || m == 2)) read_b(sl, grp + 1, m - 1);
                if (grp == 7 && m == 2) {
                    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    read_b(ns, 0, 0);
                    read_b(ns, 0, 1);
                }
                if (gl == 0) {
                    if (m == 4) {
                        if (second) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                        read_a(second ? ns : sl, second ? 0 : 1, 0);
                    }
                    if (m == 5) read_a(second ? ns : sl, second ? 0 : 1, 1);
                }
                if (gl > 0 && m >= 4 && m < 8) piece_split(second ? 0 : 1, (gl - 1) * 4 + (m - 4) & 3, ((gl - 1) * 4 + (m - 4)) >> 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // the finished stream's slab: scale / shift / activation and stores (stage A of this engine: directly, as v10's epilogue)
    ogmm_gemm gz = g;
    auto drain = [&](f32x16 (&acc)[NT], int rb, int tm) __attribute__((always_inline)) {          // tm < 0: nothing to store (stream 1 before its first row tile)
        if (!(ABL & 8) && tm >= 0) gemm_epilogue_rowblock<NT, false>(gz, acc, tm * BM + wave * 64 + rb * 32, n0, g.alpha, nullptr, 0);
        if (ABL & 8) {          // ablation without stores: keep the accumulators alive
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[j][r];
            if (sum == 1.2345f) g.C[0] = sum;
        }
        if (g.overflow && tm >= 0) {
            float chk = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) chk = fmaf(acc[0][r], 0.0f, chk);
            if (chk != chk) atomicOr(g.overflow, 1);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    };

    const int t_end = cnt * P + half;
    int kmod = 0, i0 = 0, i1 = -1;          // k block of the current step, row-tile index of stream 0 / 1 (-1: none yet)
    for (int tau = 0; tau <= t_end; ++tau) {          // (the last round only drains)
        if (kmod == 0 && tau > 0 && i0 < cnt) { drain(acc0, 0, tile_m_of(i0)); ++i0; }
        if (kmod == half) { drain(acc1, 1, i1 >= 0 ? tile_m_of(i1) : -1); ++i1; }
        if (tau < t_end) {
            step(tau & 1);
            kbn = kbn + 1 == P ? 0 : kbn + 1;
            advance_a();
            kmod = kmod + 1 == P ? 0 : kmod + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the DMA requested beyond the last step
    if ((ABL & 2048) && threadIdx.x == 0) {
        atomicAdd(&g_v14_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_v14_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
        atomicAdd(&g_v14_probe[2], (unsigned long long)cnt);
    }
}

}  // namespace

namespace ogmm {

bool gemm_f16x3_v14_applicable(const ogmm_gemm& g) {
    static const int enabled = [] { const char* e = getenv("OGMM_V14"); return e ? atoi(e) : 1; }();
    static const int cus = [] { int d = 0; hipDeviceProp_t p; if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&p, d) != hipSuccess) return 0; return p.multiProcessorCount; }();
    if (!enabled || cus != WG_SLOTS) return false;
    if (g.M % BM || g.N % BN || g.K2 != 0 || g.K1 % 64 || g.K1 < 512 || g.batch_outer * g.batch_inner != 1) return false;
    const int n_tiles = g.N / BN, m_tiles = g.M / BM;
    if (n_tiles > 32 || 32 % n_tiles) return false;
    const int mgroups = 32 / n_tiles;                       // row-tile lists per XCD
    if (m_tiles % (8 * mgroups)) return false;
    const int cnt = m_tiles / (8 * mgroups);
    return cnt >= 4 && g.pool_k == 0 && !g.col_stats && !g.a_scale && !g.Res && !g.row_affine && !g.ovl_rowpart && !g.rd_out && g.C && g.act != OGMM_ACT_SIGMOID &&
           g.ldb_h % 64 == 0 && (g.K1 + 63) / 64 * 64 <= g.ldb_h && (g.lda % 4) == 0 && (int64_t)BM * g.lda * 4 < (1ll << 31);
}

template <int ABL>
static int launch_v14(const ogmm_gemm& g, hipStream_t s) {
    const int n_tiles = g.N / BN, m_tiles = g.M / BM, mgroups = 32 / n_tiles, cnt = m_tiles / (8 * mgroups);
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v14_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((gemm_f16x3_v14_kernel<ABL>), dim3(WG_SLOTS), dim3(T), LDS_BYTES, s, g, n_tiles, cnt, mgroups);
    return check_launch("ogmm_gemm_nt(f16x3 v14)");
}

}  // namespace ogmm

extern "C" int ogmm_debug_v14_probe(unsigned long long* host3) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_v14_probe), 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v14_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}

namespace ogmm {

int gemm_nt_f16x3_v14(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {
        case 131: return launch_v14<8>(g, s);                    // no output stores
        case 132: return launch_v14<2048>(g, s);                 // clock probe (per row tile)
        case 133: return launch_v14<2048 + 8>(g, s);             // clock probe, no stores
        default: return launch_v14<0>(g, s);
    }
}

}  // namespace ogmm
