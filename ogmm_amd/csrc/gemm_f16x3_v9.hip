// fp16x3 split GEMM, LDS-DMA structure, PERSISTENT form of gemm_f16x3_v8.hip: one workgroup per CU walks a list of 256 x 256 tiles and the
// operand stream never drains between them.
//
// What the one-tile-per-workgroup form (v8) leaves on the table (131072 x 1024 x 1024: 0.63 ms without the output stores, 0.70 ms with them):
// a tile ends with 128 dword stores per wave, the wave cannot retire before they are acknowledged, the CU then waits for the next workgroup
// and for its first operand stages -- with one workgroup per CU (160 KiB of LDS) nothing overlaps any of that.  Here the K steps of a
// workgroup's tiles form ONE stream of stages n = 0, 1, 2, ...:
//   * stage n's weights go to B slot n % 2, its activations to A slot n % 3; in step n a wave issues the weight pieces of stage n+1 (MFMA groups
//     0-3) and the activation pieces of stage n+3 (groups 4-7) -- whatever tile they belong to.  Activations are wave-private (a wave reads the 32
//     rows it staged itself), so A slot n % 3 is free as soon as the wave has read k16 block 1 of stage n (group 1 of step n).
//   * one counted wait per step, `s_waitcnt vmcnt(4)` before the barrier: everything but the 4 youngest DMA instructions (activations of stage
//     n+2) has landed, i.e. the weights of stage n and, long before, the activations of stage n+1.
//   * tile boundary: after the last step the wave waits vmcnt(4) once more -- the next tile's first weights have landed, its first TWO activation
//     stages landed earlier -- and only then issues the epilogue's stores.  The first step of the next tile therefore needs no vmcnt wait at all,
//     its MFMAs run while the stores drain, and the next wait (top of the second step) asks for "at most 4 outstanding", which holds once the
//     stores are acknowledged -- a full K step later.  No assumption about the completion order of loads relative to stores is made.
//   * the stream's end is handled by clamping: when the look-ahead runs past the last stage it re-loads the last one into a ring slot nobody
//     reads any more, so the step body has no conditionals and the wait counts never change.
// Tiles are dealt per XCD (block b runs on XCD b % 8): the workgroups of XCD x walk the tiles of the row panels m = x (mod 8) in (m, n) order,
// so that the N tiles of a panel run at the same time on one XCD and the panel is fetched into one L2 (as v4 / v8).
// Arithmetic and result are those of v4 / v8 (same products in the same order per accumulator: bit-identical output).
#include <cstdlib>
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BK9 = 32;
constexpr int NT = 8;                                                        // a wave: one 32-row block x 8 column blocks
constexpr int BM = 256, BN = 256, T = 512;
constexpr int A_STAGE = BM * BK9 * 4;                                        // 32768 B
constexpr int B_STAGE = BN * BK9 * 2 * 2;                                    // 32768 B
constexpr int A_STAGES = 3, B_STAGES = 2;
constexpr int B_OFF = A_STAGES * A_STAGE;
constexpr int LDS_BYTES = A_STAGES * A_STAGE + B_STAGES * B_STAGE;          // 163840 B

__device__ unsigned long long g_v9_probe[4];          // clock probe, see gemm_f16x3_v6.hip

// SWAP: the weight fragment is the MFMA's first operand, i.e. the accumulators hold the TRANSPOSED 32 x 32 blocks (lane = row, register = column):
// the output then leaves as dwordx4 stores (gemm_epilogue_rowblock_t).  Needs scale == NULL (folded into the weights by the caller), no column
// statistics, a power-of-two alpha; the column shift enters through the accumulators' start value.  Same products, same order per accumulator.
template <int ABL, bool SWAP>
__global__ __launch_bounds__(T) void gemm_f16x3_v9_kernel(const ogmm_gemm g, const int m_tiles, const int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem9[];

    long long probe_c0 = 0, probe_w0 = 0;
    if (ABL & 2048) { probe_c0 = clock64(); probe_w0 = wall_clock64(); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int nk1 = g.K1 / BK9, nk2 = g.K2 / BK9, nk = nk1 + nk2;

    // ---- this workgroup's tile list: entries kk = j0, j0 + stride, ... of its XCD's list [batch z][row panel m = xcd (mod 8)][n]
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, stride = gridDim.x >> 3;
    const int list_len = ((m_tiles - xcd + 7) >> 3) * n_tiles;          // tiles of this XCD per batch entry
    const int total = list_len * g.batch_outer;
    if (j0 >= total) return;
    const int my_tiles = (total - j0 + stride - 1) / stride;
    auto tile_of = [&](int kk, int& z, int& m0, int& n0) {
        z = kk / list_len;
        const int k = kk - z * list_len;
        m0 = ((k / n_tiles) * 8 + xcd) * BM;
        n0 = (k % n_tiles) * BN;
    };

    const unsigned lds0 = (unsigned)(size_t)smem9;
    // ---- activation look-ahead cursor: stage (a_kk, a_t) is the next one to request; base pointer of the stage and this lane's byte offsets
    // (piece i = rows 8 i .. 8 i + 7 of the wave's 32 rows, lane l -> row (l >> 3), LDS chunk (l & 7) <- global chunk (l & 7) ^ ((row >> 1) & 7))
    int a_kk = j0, a_t = 0;
    const float* a_base;
    unsigned aoff[4];
    auto a_setup = [&]() {          // pointers / offsets of stage (a_kk, a_t): called when the cursor enters a tile or its second A piece
        int z, m0, n0;
        tile_of(a_kk, z, m0, n0);
        const bool second = a_t >= nk1;
        const int ld = second ? (int)g.lda2 : (int)g.lda;
        a_base = second ? g.A2 + z * g.sA2_o + (int64_t)m0 * g.lda2 + (a_t - nk1) * BK9 : g.A + z * g.sA_o + (int64_t)m0 * g.lda + a_t * BK9;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 32 + i * 8 + (lane >> 3);
            aoff[i] = (unsigned)(min(r, g.M - 1 - m0) * ld + ((lane & 7) ^ ((r >> 1) & 7)) * 4) * 4u;          // rows beyond M are clamped (never stored)
        }
    };
    auto a_advance = [&]() {        // to the next stage of the stream; past the end the cursor stays on the last stage (harmless re-load)
        if (a_t + 1 < nk) {
            ++a_t;
            if (a_t == nk1) a_setup(); else a_base += BK9;
        } else if (a_kk + stride < total) {
            a_kk += stride; a_t = 0;
            a_setup();
        }
    };
    // ---- weight look-ahead cursor (column block `wave` of the tile's 8; pieces: k16 block i >> 1, plane i & 1)
    int b_kk = j0, b_t = 0;
    const f16x8* b_hi; const f16x8* b_lo;          // fragment (column block, k-block 0) of the cursor's tile
    const int KB = (int)(g.ldb_h / 16);
    auto b_setup = [&]() {
        int z, m0, n0;
        tile_of(b_kk, z, m0, n0);
        const int64_t off = z * g.sB_o / 8 + ((int64_t)(n0 / 32 + wave) * KB) * 64;
        b_hi = reinterpret_cast<const f16x8*>(g.B_hi) + off;
        b_lo = reinterpret_cast<const f16x8*>(g.B_lo) + off;
    };
    auto b_advance = [&]() {
        if (b_t + 1 < nk) ++b_t;
        else if (b_kk + stride < total) { b_kk += stride; b_t = 0; b_setup(); }
    };
    const unsigned boff = lane * 16;
    int a_n = 0, b_n = 0;          // stream index of the cursors' stages (ring slots)
    auto issue_a_piece = [&](int i) { lds_dma16(aoff[i], a_base, lds0 + (a_n % A_STAGES) * A_STAGE + wave * 4096 + i * 1024); };
    auto issue_b_piece = [&](int i) {
        const int kb = ((b_t < nk1 ? b_t * 2 : (g.K1 / 16) + (b_t - nk1) * 2) + (i >> 1)) * 64;
        lds_dma16(boff, ((i & 1) ? b_lo : b_hi) + kb, lds0 + B_OFF + (b_n % B_STAGES) * B_STAGE + wave * 4096 + i * 1024);
    };

    // fragment read offsets: A row (wave*32 + lr), chunk (s*4 + lh*2 + q) ^ ((lr >> 1) & 7); B: all column blocks
    const int a_rd = (wave * 32 + lr) * 128;
    const int a_sw = (lr >> 1) & 7;
    const int a_c0 = ((lh * 2) ^ a_sw) << 4, a_c1 = ((lh * 2 + 1) ^ a_sw) << 4;        // k16 block 0; block 1 = byte offset ^ 64
    const int b_rd = lane * 16;
    float ovf = 0.0f;          // += hi . hi per pair of split values: inf / nan iff some |a| > 65504 (binary16 overflow flag)

    // SWAP: the column shift varies along the REGISTER index of the transposed accumulators, so it is added by one extra MFMA per accumulator and
    // tile: weight-side fragment {hi, lo, 0, ...} of b' = shift * 2^-15 / alpha in the k = 0, 1 positions (lanes of the lower half wave; the upper
    // half holds k = 8..15: zeros), activation-side fragment 2^15 in k = 0, 1 -- exact products, b' carried to 22 bits like every other operand.
    // The packed (hi, lo) pairs of all (at most 4) column tiles are loaded once, before any DMA is in flight: a vector load inside the stream would
    // be waited for by the compiler with the in-order vmcnt counter, i.e. together with everything this kernel keeps in flight on purpose.
    unsigned bias_v[4][NT];
    if (SWAP) {
        const float bscale = 1.0f / (g.alpha * 32768.0f);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = nt * BN + j * 32 + lr;
                const float b = (g.shift && nt < n_tiles && col < g.N) ? g.shift[col] * bscale : 0.0f;
                const _Float16 h = (_Float16)b, l = (_Float16)(b - (float)h);
                const f16x2 hl = {h, l};
                bias_v[nt][j] = lh ? 0u : __builtin_bit_cast(unsigned, hl);
            }
    }
    f32x16 acc[NT];
    f32x4 ra[2];
    f16x8 ah[2], al[2];                    // [k16 block]
    f16x8 bh[2][2], bl[2][2];              // [group parity][column block of the pair]
    auto read_a = [&](int n, int s) {
        const unsigned char* As = smem9 + (n % A_STAGES) * A_STAGE + a_rd;
        ra[0] = *reinterpret_cast<const f32x4*>(As + (a_c0 ^ (s * 64)));
        ra[1] = *reinterpret_cast<const f32x4*>(As + (a_c1 ^ (s * 64)));
    };
    auto split_a = [&](int s) {
        f16x4 h0, l0, h1, l1;
        split4_f16_pure(ra[0], h0, l0, ovf);
        split4_f16_pure(ra[1], h1, l1, ovf);
        ah[s] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        al[s] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    };
    auto read_b = [&](int n, int grp) {          // MFMA group grp = k16 block grp >> 2, column blocks 2q, 2q+1 with q = grp & 3
        const unsigned char* Bs = smem9 + B_OFF + (n % B_STAGES) * B_STAGE + b_rd;
        const int s = grp >> 2, q = grp & 3;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bh[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 0) * 1024);
            bl[grp & 1][c] = *reinterpret_cast<const f16x8*>(Bs + (((2 * q + c) * 2 + s) * 2 + 1) * 1024);
        }
    };

    // ---- prologue.  DMA order A(0), A(1), B(0), A(2): afterwards "activations of stage n+1 are older than the weights of stage n" holds for all n
    a_setup();
    b_setup();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a_piece(i);
    a_advance(); ++a_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a_piece(i);
    a_advance(); ++a_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_b_piece(i);
    b_advance(); ++b_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a_piece(i);
    a_advance(); ++a_n;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // all but A(2)
    read_a(0, 0);
    split_a(0);

    int n = 0;          // stream index of the current stage
    int c_kk = j0;      // current tile
    for (int tile_i = 0; tile_i < my_tiles; ++tile_i, c_kk += stride) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        for (int t = 0; t < nk; ++t, ++n) {
            // weights of stage n landed (this wave's pieces; only the activations of stage n+2 may be in flight).  Not in the first step of a later
            // tile: the same wait was taken before the previous tile's stores went out, and repeating it would wait for the stores.
            if (t > 0 || tile_i == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // all eight waves' weight pieces visible; everybody is done reading stage n-1
            read_b(n, 0);
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                const int s = grp >> 2, q = grp & 3, p = grp & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 1)) {
                    if (grp < 4) issue_b_piece(grp);          // weights of stage n+1
                    else issue_a_piece(grp - 4);              // activations of stage n+3
                }
                if (grp == 1) read_a(n, 1);                    // raw fragment of k16 block 1 (ra is free: block 0 was split in the previous step)
                if (grp + 1 < 8) read_b(n, grp + 1);
                if (grp == 2) split_a(1);                      // VALU in the shadow of this group's MFMAs
                if (grp == 5) read_a(n + 1, 0);                // the next stage's first activation fragment: own rows, landed (older than stage n's weights)
                if (grp == 6) split_a(0);                      // ah[0] / al[0] were last used by group 3
                __builtin_amdgcn_sched_barrier(0);
                // the two accumulators of the pair alternate; per accumulator: lo*hi, hi*lo, hi*hi
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[2 * q + c] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[p][c], al[s], acc[2 * q + c], 0, 0, 0)
                                                                  : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[p][c], acc[2 * q + c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[2 * q + c] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[p][c], ah[s], acc[2 * q + c], 0, 0, 0)
                                                                  : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[p][c], acc[2 * q + c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[2 * q + c] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[p][c], ah[s], acc[2 * q + c], 0, 0, 0)
                                                                  : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[p][c], acc[2 * q + c], 0, 0, 0);
                if (grp == 3) { __builtin_amdgcn_sched_barrier(0); b_advance(); ++b_n; }
            }
            __builtin_amdgcn_sched_barrier(0);
            a_advance(); ++a_n;
        }
        // ---- tile boundary: the next tile's first weights (and everything older) landed BEFORE this tile's stores enter the queue
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        int z, m0, n0;
        tile_of(c_kk, z, m0, n0);
        const int m_end = min(g.M, m0 + BM);
        if (SWAP && g.shift) {
            using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
            const u32x4 ones_u = {lh ? 0u : 0x78007800u, 0u, 0u, 0u};          // binary16 2^15 in k = 0, 1
            const f16x8 ones = __builtin_bit_cast(f16x8, ones_u);
            auto add_bias = [&](int nt) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    unsigned b0 = bias_v[nt][j];
                    asm volatile("" : "+v"(b0));          // opaque per tile: otherwise the 32 four-register fragments are built once, before the loop, and spilled
                    const u32x4 bu = {b0, 0u, 0u, 0u};
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bu), ones, acc[j], 0, 0, 0);
                }
            };
            const int nt = n0 / BN;
            if (nt == 0) add_bias(0); else if (nt == 1) add_bias(1); else if (nt == 2) add_bias(2); else add_bias(3);
        }
        if (ABL & 8) {          // ablation: no output stores
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[j][r];
            if (sum == 1.2345f) g.C[0] = sum;
            continue;
        }
        ogmm_gemm gz = g;
        if (gz.C) gz.C += z * g.sC_o;
        if (gz.Res) gz.Res += z * g.sR_o;
        // a wave's 32 x 256 slab: straight from the accumulators (no LDS: the ring is live) when it lies inside the matrix, else per element
        const bool inside = m0 + BM <= m_end && n0 + BN <= g.N && !g.row_affine;
        if (SWAP) {
            if (inside) {
                gemm_epilogue_rowblock_t<NT>(gz, acc, m0 + wave * 32, n0, g.alpha);
            } else {          // edge tile: per element with bounds (lane = row, register = column)
                const int row = m0 + wave * 32 + lr;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = n0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (row < m_end && col < g.N) {
                            float y = apply_act(acc[j][r] * g.alpha, g.act);
                            if (gz.Res) y += gz.Res[(int64_t)row * g.ldr + col];
                            gz.C[(int64_t)row * g.ldc + col] = y;
                        }
                    }
            }
        } else if (inside) {
            gemm_epilogue_rowblock<NT>(gz, acc, m0 + wave * 32, n0, g.alpha);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x16 pair[1][2] = {{acc[2 * q], acc[2 * q + 1]}};
                gemm_epilogue<1, 2, 8, 1, false>(gz, pair, nullptr, m0, n0 + q * 64, m_end, 0, 0, g.alpha);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the clamped look-ahead's last loads
    if ((ABL & 2048) && threadIdx.x == 0) {
        atomicAdd(&g_v9_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_v9_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
        atomicAdd(&g_v9_probe[2], 1ull);
    }
    if (g.overflow && !(fabsf(ovf) <= 3.0e38f)) atomicOr(g.overflow, 1);
}

}  // namespace

// diagnostic (tools/gemm_v6_check.py): read and clear the clock probe {shader cycles, 100 MHz wall ticks, workgroups}
extern "C" int ogmm_debug_v9_probe(unsigned long long* host3) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_v9_probe), 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v9_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}

namespace ogmm {

bool gemm_f16x3_v9_applicable(const ogmm_gemm& g) {
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch_outer;
    static const int enabled = [] { const char* e = getenv("OGMM_V9"); return e ? atoi(e) : 1; }();
    static const long long min_tiles = [] { const char* e = getenv("OGMM_V9_MIN_TILES"); return e ? atoll(e) : 256LL; }();
    return enabled && g.pool_k == 0 && !g.a_scale && g.N >= 256 && tiles >= min_tiles && g.K1 % BK9 == 0 && g.K2 % BK9 == 0 && g.ldb_h % 64 == 0 &&
           (g.K2 == 0 || g.K1 % 64 == 0) && (g.K1 + 63) / 64 * 64 + (g.K2 + 63) / 64 * 64 <= g.ldb_h && (g.lda % 4) == 0 && (g.K2 == 0 || (g.lda2 % 4) == 0) &&
           g.sB_o % 8 == 0 && (g.K1 + g.K2) / BK9 >= 1;
}

template <int ABL, bool SWAP>
static int launch_v9(const ogmm_gemm& g, hipStream_t s) {
    const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
    // workgroups: 8 XCDs x min(CUs per XCD, longest per-XCD tile list)
    const long long longest = (long long)((m_tiles + 7) / 8) * n_tiles * g.batch_outer;
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    const int per_xcd = (int)(longest < cus / 8 ? longest : cus / 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v9_kernel<ABL, SWAP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((gemm_f16x3_v9_kernel<ABL, SWAP>), dim3((unsigned)(per_xcd * 8)), dim3(T), LDS_BYTES, s, g, m_tiles, n_tiles);
    return check_launch("ogmm_gemm_nt(f16x3 v9)");
}

// the transposed-accumulator form: see the kernel's SWAP comment
static bool swap_ok(const ogmm_gemm& g) {
    static const int enabled = [] { const char* e = getenv("OGMM_V9_SWAP"); return e ? atoi(e) : 1; }();
    unsigned bits; float a = g.alpha; memcpy(&bits, &a, 4);
    return enabled && g.N <= 4 * BN && !g.scale && !g.col_stats && !g.row_affine && g.C && g.alpha > 0.0f && (bits & 0x7FFFFFu) == 0u && (g.N % 4) == 0 && (g.ldc % 4) == 0 && aligned16(g.C) &&
           (!g.Res || ((g.ldr % 4) == 0 && aligned16(g.Res)));
}

int gemm_nt_f16x3_v9(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {
        case 111: return launch_v9<8, false>(g, s);                    // no output stores
        case 112: return launch_v9<2048, false>(g, s);                 // clock probe
        case 113: return launch_v9<2048 + 8, false>(g, s);             // clock probe, no stores
        case 114: return launch_v9<0, false>(g, s);                    // dword-store epilogue (lane = column)
        case 115: OGMM_REQUIRE(swap_ok(g), "transposed form not applicable"); return launch_v9<2048, true>(g, s);      // clock probe, transposed accumulators
        default: return swap_ok(g) ? launch_v9<0, true>(g, s) : launch_v9<0, false>(g, s);
    }
}

}  // namespace ogmm
