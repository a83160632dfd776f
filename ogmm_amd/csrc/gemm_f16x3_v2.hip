// fp16x3 split GEMM, second structure: B fragments straight from a fragment-major pre-packed weight image,
// A through a double-buffered LDS image, ONE barrier per K tile.
//
// Why (measured on MI355X with tools/gemm_bench.py): in the first structure (gemm_f16x3.hip: A and B both staged
// through LDS, two barriers per tile) removing every global load still left the matrix pipe ~45 % busy -- the loop is
// bound by its LDS write phase (64 KB per K tile at the ~80 B/clk/CU ds_write rate, with all waves parked between two
// barriers) -- and the on-the-fly split arithmetic costs nothing measurable.  Weights are static, so they are packed once
// in the exact per-lane order of the v_mfma_f32_32x32x16_f16 B operand:
//     image[plane][n/32][k/16][lane 0..63][8 halfs],  lane = (k % 16 / 8) * 32 + n % 32
// and every wave fetches its B fragments as fully coalesced 1 KiB loads (L2-resident: <= 4 MB per layer), one k-step ahead,
// with no LDS traffic and no barrier dependence.  Only A (fp32 activations, shared by the WN waves of a row block) goes
// through LDS: split on the fly into hi/lo binary16 planes, two buffers, so tile t+1 is written while tile t is read
// and a single barrier per tile orders both.
#include "gemm_common.h"

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BKH = 32;
constexpr int LDH = BKH + 8;

__device__ __forceinline__ void split4v(const f32x4 v, f16x4& hi, f16x4& lo, bool& ovf) {
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

template <int MT, int NT, int WM, int WN, bool POOL>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && MT * NT == 8) ? 2 : 1) void gemm_f16x3_v2_kernel(const ogmm_gemm g, const int rows_per_tile, const int m_tiles,
                                                                     const int n_tiles) {
    constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;
    constexpr int A_PIECES = BM * 8, A_P = (A_PIECES + T - 1) / T;
    constexpr int PLANE = BM * LDH;                      // halfs per plane per buffer
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];      // [2 buffers][hi, lo][BM][LDH]

    const int bid = blockIdx.x;
    int tile_m, tile_n;
    if (m_tiles < 0) {                  // few M panels (see gemm_f16x3_v4.hip): plain tile order keeps every XCD busy
        tile_m = bid / n_tiles;
        tile_n = bid % n_tiles;
    } else {
        const int xcd = bid & 7, local = bid >> 3;
        tile_m = (local / n_tiles) * 8 + xcd;
        tile_n = local % n_tiles;
        if (tile_m >= m_tiles) return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    const int zb = blockIdx.z;                                   // batch index (outer): whole-problem strides
    const float* __restrict__ A = g.A + zb * g.sA_o;
    const float* __restrict__ A2 = g.A2 ? g.A2 + zb * g.sA2_o : nullptr;
    const int m0 = tile_m * rows_per_tile, n0 = tile_n * BN;
    const int m_end = min(g.M, m0 + rows_per_tile);
    const int nk1 = (g.K1 + BKH - 1) / BKH, nk2 = (g.K2 + BKH - 1) / BKH, nk = nk1 + nk2;

    // B image: k-blocks of 16; piece 2 starts at k = K1 (a multiple of 32)
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
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4v = {0.f, 0.f, 0.f, 0.f};
    f32x4 asc = one4, ash = zero4v;      // fused InstanceNorm: A is read as relu(a * asc + ash); one k-quad per thread and tile
    const int64_t agroup = g.a_scale ? (int64_t)(m0 / g.group_rows) * (g.K1 + g.K2) : 0;
    unsigned ra_ok = 0;          // validity bits of ra[]: the zero-select is applied when the data is CONSUMED (store_a), so the
    float amax = 0.0f;          // running max |a| of everything this thread staged (fp16 overflow flag)            // loads stay in flight across the MFMAs (a select right after the load forces vmcnt(0) there)
    auto load_a = [&](int t) {
        const bool second = t >= nk1;
        const float* Ap = second ? A2 : A;
        const int64_t ld = second ? g.lda2 : g.lda;
        const int kbase = second ? (t - nk1) * BKH : t * BKH;
        const int Kp = second ? g.K2 : g.K1;
        ra_ok = 0;
        if (g.a_scale) {
            const int kq0 = (tid & 7) * 4;
            const int kk = (kbase + kq0 < Kp) ? (second ? g.K1 : 0) + kbase + kq0 : 0;
            asc = *reinterpret_cast<const f32x4*>(g.a_scale + agroup + kk);
            ash = *reinterpret_cast<const f32x4*>(g.a_shift + agroup + kk);
        }
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int f = tid + i * T, row = f >> 3, kq = (f & 7) * 4;
            const int gm = m0 + row;
            const bool ok = (A_PIECES % T == 0 || f < A_PIECES) && gm < m_end && kbase + kq < Kp;
            ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)min(gm, g.M - 1) * ld + (ok ? kbase + kq : 0));
            ra_ok |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_a = [&](int buf) {
        _Float16* Ah = smem_h + buf * 2 * PLANE;
        _Float16* Al = Ah + PLANE;
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int f = tid + i * T;
            if (A_PIECES % T == 0 || f < A_PIECES) {
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
                const int off = (f >> 3) * LDH + (f & 7) * 4;
                *reinterpret_cast<f16x4*>(&Ah[off]) = hi;
                *reinterpret_cast<f16x4*>(&Al[off]) = lo;
            }
        }
    };
    // k-block index (in the B image) of k-step s of tile t
    auto kblk = [&](int t, int s) { return (t < nk1 ? t * 2 : (g.K1 / 16) + (t - nk1) * 2) + s; };
    auto load_b = [&](f16x8 (&bh)[NT], f16x8 (&bl)[NT], int t, int s) {
        const int kb = kblk(t, s);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = BH[bbase[j] + (int64_t)kb * 64];
            bl[j] = BL[bbase[j] + (int64_t)kb * 64];
        }
    };
    auto mma_step = [&](int buf, int s, const f16x8 (&bh)[NT], const f16x8 (&bl)[NT]) {
        const _Float16* Ah = smem_h + buf * 2 * PLANE;
        const _Float16* Al = Ah + PLANE;
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int off = ((wm * MT + i) * 32 + lr) * LDH + s * 16 + lh * 8;
            ah[i] = *reinterpret_cast<const f16x8*>(&Ah[off]);
            al[i] = *reinterpret_cast<const f16x8*>(&Al[off]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };

    f16x8 bh0[NT], bl0[NT], bh1[NT], bl1[NT];      // fragments of k-step 0 / k-step 1 (static names: no runtime indexing)
    load_a(0);
    load_b(bh0, bl0, 0, 0);
    store_a(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nk;
        // issue order is pinned (sched_barrier): hipcc otherwise sinks the B loads next to their use and then waits
        // vmcnt(0) for everything in flight.  In-order vmcnt: step 1 needs B(t,1) with only A(t+1) newer, the next
        // step 0 needs B(t+1,0) with B(t+1,1) and A(t+2) newer, store_a needs A(t+1) with B(t+1,0) newer.
        load_b(bh1, bl1, t, 1);
        if (more) load_a(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(buf, 0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_b(bh0, bl0, t + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(buf, 1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_a(buf ^ 1);
        __syncthreads();
    }
    if (g.overflow && amax > 65504.0f) atomicOr(g.overflow, 1);
    ogmm_gemm gz = g;                  // per-batch views for the epilogue
    if (gz.C) gz.C += zb * g.sC_o;
    if (gz.Res) gz.Res += zb * g.sR_o;
    if (!POOL && wide_epilogue_ok(g)) gemm_epilogue_wide<MT, NT, WM, WN>(gz, acc, reinterpret_cast<float*>(smem_h), m0, n0, m_end, g.alpha);
    else gemm_epilogue<MT, NT, WM, WN, POOL>(gz, acc, reinterpret_cast<float*>(smem_h), m0, n0, m_end, 0, 0, g.alpha);
}

template <int MT, int NT, int WM, int WN, bool POOL>
int launch_v2(const ogmm_gemm& g, hipStream_t stream) {
    constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;
    constexpr size_t LDS = (size_t)2 * 2 * BM * LDH * sizeof(_Float16);
    static_assert(LDS >= (size_t)(BM / 4) * BN * sizeof(int), "pool scratch must fit");
    static_assert(LDS >= (size_t)WM * WN * 32 * (NT * 32 + 4) * sizeof(float), "wide-epilogue patches must fit");
    const int rows_per_tile = POOL ? (BM / g.pool_k) * g.pool_k : BM;
    const int m_tiles = (g.M + rows_per_tile - 1) / rows_per_tile;
    const int n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_v2_kernel<MT, NT, WM, WN, POOL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    const bool plain = m_tiles % 8 != 0 && m_tiles < 32;
    dim3 grid((unsigned)((plain ? m_tiles : m_tiles8) * n_tiles), 1, (unsigned)g.batch_outer);
    hipLaunchKernelGGL((gemm_f16x3_v2_kernel<MT, NT, WM, WN, POOL>), grid, dim3(T), LDS, stream, g, rows_per_tile, plain ? -m_tiles : m_tiles, n_tiles);
    return ogmm::check_launch("ogmm_gemm_nt(f16x3 frag)");
}

}  // namespace

namespace ogmm {

bool gemm_f16x3_large_applicable(const ogmm_gemm& g);
int gemm_nt_f16x3_v4(const ogmm_gemm& g, hipStream_t s);
bool gemm_f16x3_v8_applicable(const ogmm_gemm& g);
int gemm_nt_f16x3_v8(const ogmm_gemm& g, hipStream_t s);
bool gemm_f16x3_v10_applicable(const ogmm_gemm& g);
int gemm_nt_f16x3_v10(const ogmm_gemm& g, hipStream_t s);

int gemm_nt_f16x3_frag(const ogmm_gemm& g, hipStream_t s) {
    OGMM_REQUIRE(g.B_hi && g.B_lo && aligned16(g.B_hi) && aligned16(g.B_lo), "ogmm_gemm_nt(f16x3 frag): needs the fragment-major B image");
    OGMM_REQUIRE(g.ldb_h > 0 && g.ldb_h % 32 == 0 && (g.K1 + 31) / 32 * 32 + (g.K2 + 31) / 32 * 32 <= g.ldb_h,
                 "ogmm_gemm_nt(f16x3 frag): ldb_h (padded K) must be a multiple of 32 covering the padded K pieces");
    OGMM_REQUIRE(g.batch_inner == 1 && (g.batch_outer == 1 || (!g.col_stats && !g.a_scale && g.pool_k == 0)),
                 "ogmm_gemm_nt(f16x3 frag): only outer batching (whole-problem strides sA_o, sB_o in binary16 elements, sC_o, sR_o), no fusion flags");
    OGMM_REQUIRE(g.sB_o % 8 == 0, "ogmm_gemm_nt(f16x3 frag): sB_o must be a multiple of 8 halfs");
    OGMM_REQUIRE(g.K2 == 0 || g.K1 % 32 == 0, "ogmm_gemm_nt(f16x3 frag): two A pieces need K1 %% 32 == 0");
    if (g.col_stats || g.a_scale)
        OGMM_REQUIRE(g.group_rows > 0 && g.group_rows % 256 == 0 && g.pool_k == 0 && (!g.a_scale || (g.a_shift && aligned16(g.a_scale) && aligned16(g.a_shift))),
                     "ogmm_gemm_nt(f16x3 frag): InstanceNorm fusion needs group_rows %% 256 == 0, no pooling, aligned a_scale/a_shift");
    if (g.col_stats) {
        ogmm_gemm probe = g;
        OGMM_REQUIRE(g.C && (g.N & 3) == 0 && (g.ldc & 3) == 0 && aligned16(g.C) && (!g.Res || g.nb_mean), "ogmm_gemm_nt(f16x3 frag): col_stats needs the wide epilogue (N, ldc %% 4 == 0, no residual)");
        (void)probe;
    }
    if (g.pool_k > 0) return g.N <= 64 ? launch_v2<5, 1, 1, 2, true>(g, s) : launch_v2<5, 1, 1, 4, true>(g, s);
    switch (g.precision) {
        case 21: return launch_v2<2, 2, 2, 2, false>(g, s);    // 128 x 128, 4 waves
#ifdef OGMM_ABLATIONS          // tools-only build (libogmm_probe.so): tile-shape experiment; the ablation codes below reach engines that only carry them in that build
        case 22: return launch_v2<4, 2, 1, 4, false>(g, s);    // 128 x 256, 4 waves of 128 x 64: two independent workgroups per CU
#endif
        case 18: case 19: case 23: case 24: case 25: case 26: case 27: case 28: case 29:            // large-shape engine and its ablations (tools/gemm_bench.py)
        case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39: case 40:
            OGMM_REQUIRE(gemm_f16x3_large_applicable(g), "large-shape engine not applicable"); return gemm_nt_f16x3_v4(g, s);
        case 100: case 101: case 102: case 103: case 104: case 105: case 106: case 107: case 108: case 109:            // LDS-DMA engine, 8 x 1 waves (v8)
            OGMM_REQUIRE(gemm_f16x3_v8_applicable(g), "LDS-DMA engine (v8) not applicable"); return gemm_nt_f16x3_v8(g, s);
        case 110: case 111: case 112: case 113: case 114: case 115: case 116: case 117: case 118: case 119: case 120: case 121:            // LDS-DMA engine, 4 waves x (64 x 256) (v10)
            OGMM_REQUIRE(gemm_f16x3_v10_applicable(g), "LDS-DMA engine (v10) not applicable"); return gemm_nt_f16x3_v10(g, s);
        default: break;
    }
    // the LDS-DMA engines (v10: 4 waves of 64 x 256; v8: 8 waves of 32 x 256) wherever they apply (their first form, v6, lives in the tools-only libogmm_probe.so)
    if (g.a_gather_ids) {
        OGMM_REQUIRE(g.precision == OGMM_PREC_F16X3_FRAG && g.N >= 512 && gemm_f16x3_v10_applicable(g), "ogmm_gemm_nt: gathered A rows need the fragment-major fp16x3 engine, N >= 512, one A piece (ogmm_gemm_gather_fusable)");
        return gemm_nt_f16x3_v10(g, s);
    }
    if (g.a_trans) {
        OGMM_REQUIRE(g.precision == OGMM_PREC_F16X3_FRAG && gemm_f16x3_v10_applicable(g), "ogmm_gemm_nt: a transposed A operand (a_trans) needs the fragment-major fp16x3 engine: one A piece, K %% 32 == 0, M %% 4 == 0, N >= 256, >= 256 tiles (ogmm_gemm_atrans_supported)");
        return gemm_nt_f16x3_v10(g, s);
    }
    if (g.nb_mean) {
        OGMM_REQUIRE(g.precision == OGMM_PREC_F16X3_FRAG && g.N >= 512 && gemm_f16x3_v10_applicable(g), "ogmm_gemm_nt: the normalisation-backward fusion needs the fragment-major fp16x3 engine on whole 256 x 256 tiles, N >= 512, col_stats, Res = x, group_rows %% 256 == 0 (ogmm_gemm_normbwd_fusable)");
        return gemm_nt_f16x3_v10(g, s);
    }
    if (g.rd_out) {
        OGMM_REQUIRE(g.precision == OGMM_PREC_F16X3_FRAG && gemm_f16x3_v8_applicable(g), "ogmm_gemm_nt: a fused Cout = 1 head needs the fragment-major fp16x3 engine, N == 256 and whole row tiles (ogmm_gemm_rowdot_fusable)");
        return gemm_nt_f16x3_v8(g, s);
    }
    if (g.ovl_rowpart) {
        OGMM_REQUIRE(g.precision == OGMM_PREC_F16X3_FRAG && gemm_f16x3_v10_applicable(g), "ogmm_gemm_nt: the overlap-block fusion needs the fragment-major fp16x3 engine on whole 256 x 256 tiles (ogmm_gemm_overlap_fusable)");
        return gemm_nt_f16x3_v10(g, s);
    }
    if (g.precision == OGMM_PREC_F16X3_FRAG) {
        // four waves of 64 x 256 (v10) from 512 output columns on; at N = 256 its longer prologue / epilogue per tile costs more than its loop gains
        // (131072 x 256 x 512: 0.109 against 0.105 ms; x 1024 x 1024: 0.645 against 0.673)
        if (g.N >= 512 && gemm_f16x3_v10_applicable(g)) return gemm_nt_f16x3_v10(g, s);
        if (gemm_f16x3_v8_applicable(g)) return gemm_nt_f16x3_v8(g, s);
    }
    if (gemm_f16x3_large_applicable(g)) return gemm_nt_f16x3_v4(g, s);
    if (g.N <= 64) return launch_v2<2, 1, 2, 2, false>(g, s);
    // 256 x 256 tiles (8 waves) once they still give >= 2 workgroups per CU, else 128 x 128 (4 waves)
    const long long big_tiles = (long long)((g.M + 255) / 256) * ((g.N + 255) / 256) * g.batch_outer;
    if (g.N >= 256 && big_tiles >= 512) return launch_v2<4, 2, 2, 4, false>(g, s);
    return launch_v2<2, 2, 2, 2, false>(g, s);
}

}  // namespace ogmm
