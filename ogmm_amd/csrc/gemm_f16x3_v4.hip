// fp16x3 split GEMM for large shapes: 256 x 256 tile, 8 waves of 256 x 32 (all rows, one 32-column block each), K tile of 64,
// one barrier per 96 MFMAs per wave, everything else issued in the shadow of the matrix pipe.
//
// With one column block per wave every B fragment (1 KiB straight from the L2-resident weight image) is fetched by exactly
// ONE wave of the workgroup (a 2 x 4 wave arrangement fetches each twice: measured -3 %); the price is 16 A-fragment LDS
// reads per k-step, which LDS has room for.  Other structures measured on the same shapes and dropped: 2 x 4 waves of
// 128 x 64 (v3), the same pipeline on 128 x 256 tiles with two independent 4-wave workgroups per CU so that one's store
// burst overlaps the other's MFMA loop (v5), K tiles of 32 with two barriers (v2's 256 x 256 instance): all land within
// +-4 % of each other (~300-330 TF algorithmic at 131072 x 1024 x 1024), i.e. the plateau is set by the matrix pipe's
// effective clock (MFMA-only loop: 543), L2->CU operand traffic and the un-overlapped output burst, not by the tiling.
//
// Per K tile (4 k-steps of 16) a wave runs 4 x 4 groups of 6 MFMAs (row block i against both column blocks: lo*hi, hi*lo,
// hi*hi).  Between groups it issues, in program order pinned with sched_barrier:
//   * the two ds_read_b128 of the NEXT group's A fragments (hi, lo)      -> LDS latency hidden behind 6 MFMAs (192 cycles)
//   * once per k-step the four 1 KiB loads of the next k-step's B fragments from the fragment-major weight image
//   * during the last two k-steps one eighth of the NEXT tile's A staging (fp32 -> hi/lo split -> ds_write into the
//     other LDS buffer), whose global loads were issued at the start of the tile
// so the only full stop is the single barrier at the end of the tile.  A and B operand formats are those of
// gemm_f16x3_v2.hip (which remains the engine for small N, small M and the EdgeConv pooling epilogue).
#include <cstdlib>
#include "gemm_common.h"
#include <stdlib.h>

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BK3 = 64;
constexpr int LD3 = BK3 + 8;          // 144-byte rows: conflict-free ds_read_b128 / ds_write_b64
constexpr int MT = 8, NT = 1, WM = 1, WN = 8;
constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;      // 256, 256, 512
constexpr int A_P = BM * (BK3 / 4) / T;                                      // 8 float4 per thread per tile
constexpr int PLANE = BM * LD3;

__device__ __forceinline__ void split4w(const f32x4 v, f16x4& hi, f16x4& lo, bool& ovf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        ovf |= fabsf(x) > 65504.0f;
        x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
        const _Float16 h = (_Float16)x;
        hi[e] = h;
        lo[e] = (_Float16)(x - (float)h);
    }
}

template <int ABL>
__global__ __launch_bounds__(T) void gemm_f16x3_v4_kernel(const ogmm_gemm g, const int m_tiles_signed, const int n_tiles, const int direct_stores) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];      // [2 buffers][hi, lo][BM][LD3]

    // XCD-aware map (block b runs on XCD b % 8): all N tiles of an M panel on one XCD, so the panel is fetched into one L2.
    // With fewer than 8 M panels (weight-gradient GEMMs: M = output channels, the long axis is the split-K batch) that map
    // would leave XCDs idle -- m_tiles < 0 selects the plain row-major tile order instead.
    const int bid = blockIdx.x;
    int tile_m, tile_n;
    if (m_tiles_signed < 0) {
        tile_m = bid / n_tiles;
        tile_n = bid % n_tiles;
    } else {
        const int xcd = bid & 7, local = bid >> 3;
        tile_m = (local / n_tiles) * 8 + xcd;
        tile_n = local % n_tiles;
        if (tile_m >= m_tiles_signed) return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int zb = blockIdx.z;                                   // batch index (outer)
    const float* __restrict__ Abase = g.A + zb * g.sA_o;
    const float* __restrict__ A2base = g.A2 ? g.A2 + zb * g.sA2_o : nullptr;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int m_end = min(g.M, m0 + BM);
    const int nk1 = (g.K1 + BK3 - 1) / BK3, nk2 = (g.K2 + BK3 - 1) / BK3, nk = nk1 + nk2;

    const int KB = (int)(g.ldb_h / 16);
    const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_hi) + zb * g.sB_o);
    const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(g.B_lo) + zb * g.sB_o);
    int64_t bbase[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bbase[j] = ((int64_t)(n0 / 32 + wn * NT + j) * KB) * 64 + lane;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[A_P];
    unsigned ra_ok = 0;
    float amax = 0.0f;          // running max |a| of everything this thread staged (fp16 overflow flag)
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4v = {0.f, 0.f, 0.f, 0.f};
    f32x4 asc = one4, ash = zero4v;      // fused InstanceNorm: A is read as relu(a * asc + ash); one k-quad per thread and tile
    const int64_t agroup = g.a_scale ? (int64_t)(m0 / g.group_rows) * (g.K1 + g.K2) : 0;
    auto load_a = [&](int t) {
        const bool second = t >= nk1;
        const float* Ap = second ? A2base : Abase;
        const int64_t ld = second ? g.lda2 : g.lda;
        const int kbase = second ? (t - nk1) * BK3 : t * BK3;
        const int Kp = second ? g.K2 : g.K1;
        ra_ok = 0;
        if (g.a_scale) {
            const int kq0 = (tid & 15) * 4;
            const int kk = (kbase + kq0 < Kp) ? (second ? g.K1 : 0) + kbase + kq0 : 0;
            asc = *reinterpret_cast<const f32x4*>(g.a_scale + agroup + kk);
            ash = *reinterpret_cast<const f32x4*>(g.a_shift + agroup + kk);
        }
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int f = tid + i * T, row = f >> 4, kq = (f & 15) * 4;
            const int gm = m0 + row;
            const bool ok = gm < m_end && kbase + kq < Kp;
            ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)min(gm, g.M - 1) * ld + (ok ? kbase + kq : 0));
            ra_ok |= (ok ? 1u : 0u) << i;
        }
    };
    // one eighth of load_a (deep mode spreads the tile's A loads over the first two k-steps: eight back-to-back 1 KiB loads per
    // wave, from all eight waves at once, fill the CU's vector-memory queue and stall MFMA issue behind them)
    auto load_a_piece = [&](int t, int i) {
        const bool second = t >= nk1;
        const float* Ap = second ? A2base : Abase;
        const int64_t ld = second ? g.lda2 : g.lda;
        const int kbase = second ? (t - nk1) * BK3 : t * BK3;
        const int Kp = second ? g.K2 : g.K1;
        if (i == 0) {
            ra_ok = 0;
            if (g.a_scale) {
                const int kq0 = (tid & 15) * 4;
                const int kk = (kbase + kq0 < Kp) ? (second ? g.K1 : 0) + kbase + kq0 : 0;
                asc = *reinterpret_cast<const f32x4*>(g.a_scale + agroup + kk);
                ash = *reinterpret_cast<const f32x4*>(g.a_shift + agroup + kk);
            }
        }
        const int f = tid + i * T, row = f >> 4, kq = (f & 15) * 4;
        const int gm = m0 + row;
        const bool ok = gm < m_end && kbase + kq < Kp;
        ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)min(gm, g.M - 1) * ld + (ok ? kbase + kq : 0));
        ra_ok |= (ok ? 1u : 0u) << i;
    };
    auto store_a_piece = [&](int buf, int i) {
        _Float16* Ah = smem_h + buf * 2 * PLANE;
        _Float16* Al = Ah + PLANE;
        const int f = tid + i * T;
        f16x4 hi, lo;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 val = ra[i];
        if (g.a_scale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                val[e] = fmaf(val[e], asc[e], ash[e]);
                if (g.a_relu) val[e] = fmaxf(val[e], 0.0f);
            }
        }
        split4_f16(((ra_ok >> i) & 1u) ? val : zero, hi, lo, amax);
        const int off = (f >> 4) * LD3 + (f & 15) * 4;
        if (ABL & 256) {            // ablation 256: no LDS writes (loads + split kept alive)
            asm volatile("" :: "v"(hi), "v"(lo));
            return;
        }
        *reinterpret_cast<f16x4*>(&Ah[off]) = hi;
        *reinterpret_cast<f16x4*>(&Al[off]) = lo;
    };
    auto kblk = [&](int t, int s) { return (t < nk1 ? t * 4 : (g.K1 / 16) + (t - nk1) * 4) + s; };
    auto load_b = [&](f16x8 (&bh)[NT], f16x8 (&bl)[NT], int t, int s) {
        const int64_t kb = (int64_t)kblk(t, s) * 64;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = BH[bbase[j] + kb];
            bl[j] = (ABL & (64 | 1024)) ? bh[j] : BL[bbase[j] + kb];  // 64: ablation (half the weight traffic); 1024: single-term mode, lo unused
        }
    };
    auto read_a = [&](f16x8 (&ah)[2], f16x8 (&al)[2], int buf, int s, int gI) {
        const _Float16* Ah = smem_h + buf * 2 * PLANE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int off = ((2 * gI + u) * 32 + lr) * LD3 + s * 16 + lh * 8;
            ah[u] = *reinterpret_cast<const f16x8*>(&Ah[off]);
            al[u] = *reinterpret_cast<const f16x8*>(&Ah[PLANE + off]);
        }
    };
    // group gI = row blocks 2*gI and 2*gI+1: six MFMAs alternating between their two accumulators
    auto mma6 = [&](int gI, const f16x8 (&ah)[2], const f16x8 (&al)[2], const f16x8 (&bh)[NT], const f16x8 (&bl)[NT]) {
        if (!(ABL & 1024)) {        // 1024 = OGMM_PREC_F16_FRAG: plain binary16 product hi*hi only (reduced precision, 1/3 of the MFMAs)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[2 * gI + u][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u], bh[0], acc[2 * gI + u][0], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[2 * gI + u][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bl[0], acc[2 * gI + u][0], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[2 * gI + u][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bh[0], acc[2 * gI + u][0], 0, 0, 0);
    };

    // one k-step: 4 groups of two row blocks; A fragments of the next group (or of step s+1's first group) are read one group ahead
    f16x8 bhA[NT], blA[NT], bhB[NT], blB[NT];      // B fragments: even / odd k-steps
    f16x8 bhC[NT], blC[NT], bhD[NT], blD[NT];      // (deep mode) k-steps 2, 3 of a tile; A, B then hold k-steps 0, 1
    f16x8 ah0[2], al0[2], ah1[2], al1[2];          // A fragments: even / odd groups (two row blocks each)
    constexpr bool DEEP = (ABL & 2048) != 0;       // B fragments fetched 3 k-steps ahead (one register set per k-step of a tile)

    load_a(0);
    load_b(bhA, blA, 0, 0);
    if (DEEP) { load_b(bhB, blB, 0, 1); load_b(bhC, blC, 0, 2); }
    if (ABL & 7) { load_b(bhB, blB, 0, 1); }
#pragma unroll
    for (int i = 0; i < A_P; ++i) store_a_piece(0, i);
    __syncthreads();
    read_a(ah0, al0, 0, 0, 0);
    if (ABL & 7) { read_a(ah1, al1, 0, 0, 1); }

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nk;
        if ((ABL & 512) && more && !(ABL & 4) && !(ABL & 128)) load_a(t + 1);      // A/B switch: A loads at the tile start
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // B fragments of the next k-step (next tile's first step after the last one)
            if (DEEP) {
                // k-step s consumes set s; the set freed by k-step s-1 is refilled with the fragments needed 3 k-steps from now.
                // Every fragment waited for during this tile was requested BEFORE this tile's A loads (in-order vmcnt): the matrix
                // pipe never waits for the HBM-latency A loads, which get three k-steps before their split.
                if (s == 0) load_b(bhD, blD, t, 3);
                else if (more) { if (s == 1) load_b(bhA, blA, t + 1, 0); else if (s == 2) load_b(bhB, blB, t + 1, 1); else load_b(bhC, blC, t + 1, 2); }
            } else if (!(ABL & 2)) {
            if (s < 3) { if (s & 1) load_b(bhA, blA, t, s + 1); else load_b(bhB, blB, t, s + 1); }
            else if (more) load_b(bhA, blA, t + 1, 0);
            }
            // The next tile's A loads go out AFTER step 0's B-fragment loads: vmcnt retires in order, so every fragment
            // consumed after this point waits for these (HBM-latency) loads too; issued here the first such consumer is
            // step 2's, two k-steps (~3 us) away, instead of step 1's (measured: the A loads cost 20 % of the loop).
            if (s == 0 && !(ABL & 512) && !(ABL & 8192)) {
                __builtin_amdgcn_sched_barrier(0);
                if (more && !(ABL & 4) && !(ABL & 128)) load_a(t + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MT / 2; ++i) {
                if (!(ABL & 32)) __builtin_amdgcn_sched_barrier(0);
                // prefetch the next group's A fragments (same tile only: the next tile's first group is read after the barrier)
                if (ABL & 4096) {           // single A-fragment buffer: this group's fragments are read right before its MFMAs (the other
                    read_a(ah0, al0, buf, s, i);   // wave of the SIMD covers the LDS latency); frees 16 registers for the deep B prefetch
                } else
                if (!(ABL & 1)) {
                if (i < MT / 2 - 1) { if (i & 1) read_a(ah0, al0, buf, s, i + 1); else read_a(ah1, al1, buf, s, i + 1); }
                else if (s < 3) read_a(ah0, al0, buf, s + 1, 0);
                }
                if ((ABL & 8192) && more && s < 2) load_a_piece(t + 1, s * 4 + i);          // spread A loads: one piece per MFMA group
                if (DEEP) { if (more && s == 3 && !(ABL & 4)) { store_a_piece(buf ^ 1, 2 * i); store_a_piece(buf ^ 1, 2 * i + 1); } }
                else if (more && s >= 2 && !(ABL & 4)) store_a_piece(buf ^ 1, (s - 2) * 4 + i);
                if (!(ABL & 32)) __builtin_amdgcn_sched_barrier(0);
                if (DEEP && (ABL & 4096)) {
                    if (s == 0) mma6(i, ah0, al0, bhA, blA); else if (s == 1) mma6(i, ah0, al0, bhB, blB);
                    else if (s == 2) mma6(i, ah0, al0, bhC, blC); else mma6(i, ah0, al0, bhD, blD);
                } else if (DEEP) {
                    if (s == 0)      { if (i & 1) mma6(i, ah1, al1, bhA, blA); else mma6(i, ah0, al0, bhA, blA); }
                    else if (s == 1) { if (i & 1) mma6(i, ah1, al1, bhB, blB); else mma6(i, ah0, al0, bhB, blB); }
                    else if (s == 2) { if (i & 1) mma6(i, ah1, al1, bhC, blC); else mma6(i, ah0, al0, bhC, blC); }
                    else             { if (i & 1) mma6(i, ah1, al1, bhD, blD); else mma6(i, ah0, al0, bhD, blD); }
                } else
                if (s & 1) { if (i & 1) mma6(i, ah1, al1, bhB, blB); else mma6(i, ah0, al0, bhB, blB); }
                else       { if (i & 1) mma6(i, ah1, al1, bhA, blA); else mma6(i, ah0, al0, bhA, blA); }
            }
        }
        __syncthreads();
        if (more && !(ABL & 1)) read_a(ah0, al0, buf ^ 1, 0, 0);
    }
    if (g.overflow && amax > 65504.0f) atomicOr(g.overflow, 1);
    if (ABL & 8) {          // ablation: no output stores (one dummy store keeps the accumulators live)
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
    ogmm_gemm gz = g;                  // per-batch views for the epilogue
    if (gz.C) gz.C += zb * g.sC_o;
    if (gz.Res) gz.Res += zb * g.sR_o;
    if (ABL & 16384) {      // ablation: every tile writes C[0:256, 0:256] (all instructions of the epilogue, no HBM write traffic)
        gemm_epilogue_wide<MT, NT, WM, WN>(gz, acc, reinterpret_cast<float*>(smem_h), 0, 0, BM, g.alpha);
        return;
    }
    if (wide_epilogue_ok(g)) gemm_epilogue_wide<MT, NT, WM, WN>(gz, acc, reinterpret_cast<float*>(smem_h), m0, n0, m_end, g.alpha, direct_stores != 0);
    else gemm_epilogue<MT, NT, WM, WN, false>(gz, acc, reinterpret_cast<float*>(smem_h), m0, n0, m_end, 0, 0, g.alpha);
}

}  // namespace

namespace ogmm {

bool gemm_f16x3_large_applicable(const ogmm_gemm& g) {
    const long long tiles = (long long)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.batch_outer;
    static const long long min_tiles = [] { const char* e = getenv("OGMM_V4_MIN_TILES"); return e ? atoll(e) : 256LL; }();      // one full round of 256 CUs; below that the 128x128 engine fills the chip better
    return g.pool_k == 0 && g.N >= 256 && tiles >= min_tiles && g.ldb_h % 64 == 0 && (g.K2 == 0 || g.K1 % 64 == 0) &&
           (g.K1 + 63) / 64 * 64 + (g.K2 + 63) / 64 * 64 <= g.ldb_h;
}

template <int ABL>
static int launch_v4(const ogmm_gemm& g, hipStream_t s) {
    constexpr size_t LDS = (size_t)2 * 2 * PLANE * sizeof(_Float16);      // 147456 B
    const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    static const int direct = [] { const char* e = getenv("OGMM_V4_DIRECT"); return e ? atoi(e) : 1; }();      // 0: transposed dwordx4 stores also without residual / statistics
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v4_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    if (m_tiles % 8 != 0 && m_tiles < 32)
        hipLaunchKernelGGL(gemm_f16x3_v4_kernel<ABL>, dim3((unsigned)(m_tiles * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS, s, g, -m_tiles, n_tiles, direct);
    else
        hipLaunchKernelGGL(gemm_f16x3_v4_kernel<ABL>, dim3((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)g.batch_outer), dim3(T), LDS, s, g, m_tiles, n_tiles, direct);
    return check_launch("ogmm_gemm_nt(f16x3 v4)");
}

int gemm_nt_f16x3_v4(const ogmm_gemm& g, hipStream_t s) {
    switch (g.precision) {          // 26..29: ablations for tools/gemm_bench.py (wrong results by construction)
#ifdef OGMM_ABLATIONS          // tools-only build (libogmm_probe.so)
        case 26: return launch_v4<7>(g, s);     // MFMA + barrier only
        case 29: return launch_v4<3>(g, s);     // + A global loads / split / LDS writes only
        case 19: return launch_v4<15>(g, s);    // MFMA only, no epilogue stores
        case 27: return launch_v4<32 + 8 + 128>(g, s);   // no stores, no A global loads
        case 28: return launch_v4<32 + 512>(g, s);   // A loads at the tile start (older order)
        case 18: return launch_v4<8>(g, s);     // full loop, no epilogue stores
        // single-term (1/3 of the MFMAs) ablations: what bounds the loop once the matrix pipe is light
        case 30: return launch_v4<32 + 1024 + 8>(g, s);              // no stores
        case 31: return launch_v4<32 + 1024 + 8 + 128>(g, s);        // no stores, no A global loads
        case 32: return launch_v4<32 + 1024 + 8 + 256>(g, s);        // no stores, no A LDS writes
        case 33: return launch_v4<32 + 1024 + 8 + 2>(g, s);          // no stores, no B fragment loads
        case 34: return launch_v4<32 + 1024 + 8 + 1>(g, s);          // no stores, no A fragment LDS reads
        case 35: return launch_v4<32 + 1024 + 8 + 4>(g, s);          // no stores, no A staging at all (loads, split, LDS writes)
        case 36: return launch_v4<32 + 1024 + 8 + 4 + 1 + 2>(g, s);  // no stores, MFMA + barrier only
        case 37: return launch_v4<32 + 2048>(g, s);                  // B fragments 3 k-steps ahead, A split in the last k-step
        case 38: return launch_v4<32 + 2048 + 8>(g, s);              // same, no stores
        case 39: return launch_v4<32 + 2048 + 4096>(g, s);           // deep B prefetch + single A-fragment buffer
        case 24: return launch_v4<32 + 2048 + 4096 + 8192>(g, s);    // + A loads spread over the first two k-steps
        case 25: return launch_v4<32 + 2048 + 4096 + 8192 + 8>(g, s);   // same, no stores
        case 40: return launch_v4<32 + 2048 + 4096 + 8192 + 16384>(g, s);   // same, all tiles store to one 256 x 256 patch of C
#endif
        case OGMM_PREC_F16_FRAG: return launch_v4<32 + 1024>(g, s);     // single binary16 term
        default: {
#ifdef OGMM_ABLATIONS
            static const char* env = getenv("OGMM_V4_OLD");
            if (env && env[0] == '1') return launch_v4<32>(g, s);     // previous default (1-step B prefetch, double-buffered A fragments)
#endif
            // B fragments 3 k-steps ahead (no matrix-pipe wait ever falls behind the A loads in the in-order vmcnt queue), one
            // A-fragment register set (the SIMD's other wave covers the LDS latency), A loads spread over two k-steps: +3 %
            return launch_v4<32 + 2048 + 4096 + 8192>(g, s);
        }
    }
}

}  // namespace ogmm
