// The 1x1-convolution engine: exact-fp32 NT GEMM on v_mfma_f32_32x32x2_f32 with fused epilogues.
//
//   C[z][m][n] = act(alpha * sum_k [A|A2][z][m][k] * B[z][n][k] * scale + shift) + Res
//
// Replaces every nn.Conv1d/Conv2d(kernel_size=1) (+ eval BatchNorm + activation) of
// models/dgcnn.py:19-35,121-152 and models/attn.py:17-27,34-57,91-99, the attention products of
// models/attn.py:79,82 and the N x N similarity of models/gmmreg.py:75.
//
// Tiling (gfx950, wave64): a workgroup of WM x WN waves owns a (MT*32*WM) x (NT*32*WN) tile; each wave
// keeps MT x NT 32x32 accumulators (16 VGPRs each).  K is walked in tiles of 32 floats staged through
// LDS as row-major [rows][36] (4 floats of padding make the ds_read_b128 fragment loads conflict-free);
// the next tile's global loads are issued before the MFMAs of the current one.  The k order inside a
// tile is permuted (lane half h takes k = 8g+4h+s at MFMA step s) so that one ds_read_b128 feeds four
// MFMAs; A and B use the same permutation, so the sum is over the same set of products.
//
// Workgroup -> tile map is XCD-aware: all N-tiles of one M-tile run on the same XCD (block b is
// dispatched to XCD b % 8), so the A panel is fetched into one L2 only.
#include "gemm_common.h"

namespace {

using namespace ogmm_gemm_detail;

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int MT, int NT, int WM, int WN, bool POOL>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const ogmm_gemm g, const int rows_per_tile,
                                                               const int m_tiles, const int n_tiles) {
    constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;
    constexpr int A_F4 = BM * 8 / T, B_F4 = BN * 8 / T;
    static_assert((BM * 8) % T == 0 && (BN * 8) % T == 0, "tile / thread mismatch");
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    // XCD-aware tile assignment
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int tile_m = (local / n_tiles) * 8 + xcd;
    const int tile_n = local % n_tiles;
    if (tile_m >= m_tiles) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    const int z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
    const float* __restrict__ A = g.A + zo * g.sA_o + zi * g.sA_i;
    const float* __restrict__ A2 = g.A2 ? g.A2 + zo * g.sA2_o + zi * g.sA2_i : nullptr;
    const float* __restrict__ Bm = g.B + zo * g.sB_o + zi * g.sB_i;

    const int m0 = tile_m * rows_per_tile, n0 = tile_n * BN;
    const int m_end = min(g.M, m0 + rows_per_tile);
    const int nk1 = (g.K1 + BK - 1) / BK, nk2 = (g.K2 + BK - 1) / BK, nk = nk1 + nk2;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[A_F4], rb[B_F4];
    unsigned ra_ok = 0, rb_ok = 0;     // zero-selects are applied when the registers are consumed, not right after the load
    auto load_tile = [&](int t) {
        const bool second = t >= nk1;
        const float* Ap = second ? A2 : A;
        const int64_t ld = second ? g.lda2 : g.lda;
        const int kbase = second ? (t - nk1) * BK : t * BK;
        const int Kp = second ? g.K2 : g.K1;
        const int kB = second ? g.K1 + kbase : kbase;
        // unconditional loads from clamped (always valid) addresses + select (no branches around loads)
        ra_ok = 0;
        rb_ok = 0;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * T, row = f >> 3, kq = (f & 7) * 4;
            const int gm = m0 + row;
            const bool ok = gm < m_end && kbase + kq < Kp;
            ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)min(gm, g.M - 1) * ld + (ok ? kbase + kq : 0));
            ra_ok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * T, row = f >> 3, kq = (f & 7) * 4;
            const int gn = n0 + row;
            const bool ok = gn < g.N && kbase + kq < Kp;
            rb[i] = *reinterpret_cast<const f32x4*>(Bm + (int64_t)min(gn, g.N - 1) * g.ldb + (ok ? kB + kq : 0));
            rb_ok |= (ok ? 1u : 0u) << i;
        }
    };

    load_tile(0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * T;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&As[(f >> 3) * LDS_LD + (f & 7) * 4]) = ((ra_ok >> i) & 1u) ? ra[i] : zero;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * T;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&Bs[(f >> 3) * LDS_LD + (f & 7) * 4]) = ((rb_ok >> i) & 1u) ? rb[i] : zero;
        }
        __syncthreads();
        if (t + 1 < nk) load_tile(t + 1);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MT + i) * 32 + lr) * LDS_LD + kg * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NT + j) * 32 + lr) * LDS_LD + kg * 8 + lh * 4]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }

    gemm_epilogue<MT, NT, WM, WN, POOL>(g, acc, smem, m0, n0, m_end, zo, zi, g.alpha);
}

template <int MT, int NT, int WM, int WN, bool POOL>
int launch(const ogmm_gemm& g, hipStream_t stream) {
    constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;
    const int rows_per_tile = POOL ? (BM / g.pool_k) * g.pool_k : BM;
    const int m_tiles = (g.M + rows_per_tile - 1) / rows_per_tile;
    const int n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    dim3 grid((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)(g.batch_outer * g.batch_inner));
    hipLaunchKernelGGL((gemm_nt_kernel<MT, NT, WM, WN, POOL>), grid, dim3(T), 0, stream, g, rows_per_tile, m_tiles, n_tiles);
    return ogmm::check_launch("ogmm_gemm_nt");
}

}  // namespace

namespace ogmm {
int gemm_nt_f16x3(const ogmm_gemm& g, hipStream_t s);
int gemm_nt_f16x3_frag(const ogmm_gemm& g, hipStream_t s);
bool gemm_f16x3_v10_applicable(const ogmm_gemm& g);
bool gemm_f16x3_v8_applicable(const ogmm_gemm& g);
}

extern "C" int ogmm_gemm_nt(const ogmm_gemm* d, void* stream) {
    OGMM_REQUIRE(d != nullptr, "ogmm_gemm_nt: null descriptor");
    const ogmm_gemm& g = *d;
    OGMM_REQUIRE(g.A && (g.B || g.precision != OGMM_PREC_F32) && g.M > 0 && g.N > 0 && g.K1 > 0, "ogmm_gemm_nt: A, B, M, N, K1 required");
    OGMM_REQUIRE(g.precision == OGMM_PREC_F32 || g.precision == OGMM_PREC_F16X3 || g.precision == OGMM_PREC_F16X3_FRAG || g.precision == OGMM_PREC_F16_FRAG || (g.precision > 10 && g.precision <= 129), "ogmm_gemm_nt: bad precision %d", g.precision);
    OGMM_REQUIRE(g.K2 >= 0 && (g.K2 == 0 || g.A2), "ogmm_gemm_nt: K2 > 0 needs A2");
    OGMM_REQUIRE(g.K1 % 4 == 0 && g.K2 % 4 == 0 && g.lda % 4 == 0 && (g.ldb % 4 == 0 || g.precision != OGMM_PREC_F32) && (g.K2 == 0 || g.lda2 % 4 == 0),
                 "ogmm_gemm_nt: K1, K2, lda, lda2, ldb must be multiples of 4 (got %d %d %lld %lld %lld)", g.K1, g.K2,
                 (long long)g.lda, (long long)g.lda2, (long long)g.ldb);
    OGMM_REQUIRE(g.K2 == 0 || g.K1 % BK == 0, "ogmm_gemm_nt: with two A pieces K1 must be a multiple of %d", BK);
    OGMM_REQUIRE(ogmm::aligned16(g.A) && (g.precision != OGMM_PREC_F32 || ogmm::aligned16(g.B)) && (!g.A2 || ogmm::aligned16(g.A2)),
                 "ogmm_gemm_nt: operand pointers must be 16-byte aligned");
    OGMM_REQUIRE(g.sA_o % 4 == 0 && g.sA_i % 4 == 0 && g.sB_o % 4 == 0 && g.sB_i % 4 == 0 && g.sA2_o % 4 == 0 && g.sA2_i % 4 == 0,
                 "ogmm_gemm_nt: batch strides of A/B must be multiples of 4");
    OGMM_REQUIRE(g.batch_outer >= 1 && g.batch_inner >= 1, "ogmm_gemm_nt: batch counts must be >= 1");
    OGMM_REQUIRE(g.C || g.pool_k > 0 || g.ovl_rowpart || g.rd_out, "ogmm_gemm_nt: no output");
    OGMM_REQUIRE(g.act >= OGMM_ACT_NONE && g.act <= OGMM_ACT_SIGMOID, "ogmm_gemm_nt: bad act %d", g.act);
    hipStream_t s = ogmm::as_stream(stream);
    const bool frag = g.precision == OGMM_PREC_F16X3_FRAG || g.precision == OGMM_PREC_F16_FRAG || g.precision >= 18;
    OGMM_REQUIRE(g.col_stats_slot_mask >= 0 && ((g.col_stats_slot_mask + 1) & g.col_stats_slot_mask) == 0 &&
                 (g.col_stats_slot_mask == 0 || (g.col_stats && g.group_rows > 0 && g.col_stats_slot_stride >= (int64_t)((g.M + g.group_rows - 1) / g.group_rows) * g.N * 2)),
                 "ogmm_gemm_nt: col_stats_slot_mask must be 2^n - 1 and col_stats_slot_stride must hold one [groups][N][2] copy");
    OGMM_REQUIRE(frag || (!g.col_stats && !g.a_scale && !g.ovl_rowpart && !g.rd_out && !g.a_gather_ids && !g.nb_mean && !g.a_trans), "ogmm_gemm_nt: InstanceNorm / overlap-block / Cout = 1 head / row gather / normalisation-backward fusion is only available with OGMM_PREC_F16X3_FRAG");
    if (g.pool_k > 0)
        OGMM_REQUIRE(g.pool_out && g.act == OGMM_ACT_RELU && g.pool_k >= 4 && g.pool_k <= 160 && g.M % g.pool_k == 0 &&
                         g.batch_outer * g.batch_inner == 1,
                     "ogmm_gemm_nt: pooling needs pool_out, ReLU, 4 <= pool_k <= 160, M %% pool_k == 0, no batching");
    if (frag) return ogmm::gemm_nt_f16x3_frag(g, s);
#ifdef OGMM_ABLATIONS
    if (g.precision != OGMM_PREC_F32) return ogmm::gemm_nt_f16x3(g, s);
#else
    // the first fp16x3 engine (row-major split planes, gemm_f16x3.hip) is not reachable from any model path: it lives in the tools-only libogmm_probe.so
    OGMM_REQUIRE(g.precision == OGMM_PREC_F32, "ogmm_gemm_nt: OGMM_PREC_F16X3 (row-major split planes) is served by libogmm_probe.so (ogmm_probe_gemm_nt); "
                 "the product engines take the fragment-major image (OGMM_PREC_F16X3_FRAG)");
#endif
    if (g.pool_k > 0) {
        return g.N <= 64 ? launch<5, 1, 1, 2, true>(g, s) : launch<5, 1, 1, 4, true>(g, s);
    }
    return g.N <= 64 ? launch<2, 1, 2, 2, false>(g, s) : launch<2, 2, 2, 2, false>(g, s);
}

// Would ogmm_gemm_nt take the fused overlap block (ogmm_gemm.ovl_rowpart) for B pairs of N points with D channels?  (host-side planning: 1 / 0)
extern "C" int ogmm_gemm_overlap_fusable(int B, int N, int D) {
    if (B <= 0 || N <= 0 || D <= 0) return 0;
    static float dummy[4];
    ogmm_gemm g = {};
    g.A = dummy; g.lda = D; g.K1 = D; g.M = N; g.N = N; g.batch_outer = B; g.batch_inner = 1; g.precision = OGMM_PREC_F16X3_FRAG;
    g.ldb_h = (D + 63) / 64 * 64; g.B_hi = dummy; g.B_lo = dummy;
    g.ovl_rowpart = dummy; g.ovl_colpart = dummy; g.ovl_orow = dummy; g.ovl_ocol = dummy; g.ovl_ld = 1;
    return ogmm::gemm_f16x3_v10_applicable(g) ? 1 : 0;
}

// Would ogmm_gemm_nt take a fused Cout = 1 head (ogmm_gemm.rd_out) behind an M x N layer with K1 + K2 input channels?  (1 / 0)
extern "C" int ogmm_gemm_rowdot_fusable(int M, int N, int K1, int K2) {
    if (M <= 0 || N <= 0 || K1 <= 0 || K2 < 0) return 0;
    static float dummy[4];
    ogmm_gemm g = {};
    g.A = dummy; g.lda = K1; g.K1 = K1; g.A2 = K2 ? dummy : nullptr; g.lda2 = K2 ? K2 : 0; g.K2 = K2; g.M = M; g.N = N; g.batch_outer = 1; g.batch_inner = 1;
    g.precision = OGMM_PREC_F16X3_FRAG; g.ldb_h = (K1 + 63) / 64 * 64 + (K2 + 63) / 64 * 64; g.B_hi = dummy; g.B_lo = dummy;
    g.rd_out = dummy; g.rd_w = dummy; g.rd_ld = 1;
    return ogmm::gemm_f16x3_v8_applicable(g) ? 1 : 0;
}

// Would ogmm_gemm_nt take the normalisation-backward fusion (ogmm_gemm.nb_*) for an M x N layer with K input channels and groups of group_rows rows?  (1 / 0)
extern "C" int ogmm_gemm_normbwd_fusable(int M, int N, int K, int group_rows) {
    if (M <= 0 || N < 512 || K <= 0 || group_rows <= 0) return 0;
    static float dummy[4];
    static double ddummy[4];
    ogmm_gemm g = {};
    g.A = dummy; g.lda = K; g.K1 = K; g.M = M; g.N = N; g.batch_outer = 1; g.batch_inner = 1; g.precision = OGMM_PREC_F16X3_FRAG;
    g.ldb_h = (K + 63) / 64 * 64; g.B_hi = dummy; g.B_lo = dummy; g.C = dummy; g.Res = dummy; g.col_stats = ddummy; g.group_rows = group_rows;
    g.nb_mean = dummy; g.nb_rstd = dummy; g.nb_scale = dummy; g.nb_shift = dummy; g.nb_act = OGMM_ACT_RELU;
    return ogmm::gemm_f16x3_v10_applicable(g) ? 1 : 0;
}

// Would ogmm_gemm_nt take a transposed A operand (ogmm_gemm.a_trans) for `batch` products of M x N over K with row pitch lda of the [K][M] map?  (1 / 0)
extern "C" int ogmm_gemm_atrans_supported(int M, int N, int K, int64_t lda, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || lda < M || batch <= 0) return 0;
    static float dummy[4];
    ogmm_gemm g = {};
    g.A = dummy; g.lda = lda; g.K1 = K; g.M = M; g.N = N; g.batch_outer = batch; g.batch_inner = 1; g.precision = OGMM_PREC_F16X3_FRAG;
    g.ldb_h = (K + 63) / 64 * 64; g.B_hi = dummy; g.B_lo = dummy; g.C = dummy; g.a_trans = 1;
    return ogmm::gemm_f16x3_v10_applicable(g) ? 1 : 0;
}

// Would ogmm_gemm_nt take gathered A rows (ogmm_gemm.a_gather_ids) for an M x N layer with K input channels over `rows` source rows?  (1 / 0)
extern "C" int ogmm_gemm_gather_fusable(int M, int N, int K, int64_t rows) {
    if (M <= 0 || N < 512 || K <= 0 || rows <= 0) return 0;
    static float dummy[4];
    static int32_t idummy[4];
    ogmm_gemm g = {};
    g.A = dummy; g.lda = K; g.K1 = K; g.M = M; g.N = N; g.batch_outer = 1; g.batch_inner = 1; g.precision = OGMM_PREC_F16X3_FRAG;
    g.ldb_h = (K + 63) / 64 * 64; g.B_hi = dummy; g.B_lo = dummy; g.C = dummy;
    g.a_gather_ids = idummy; g.a_gather_S = 1; g.a_gather_N = 1; g.a_gather_rows = rows;
    return ogmm::gemm_f16x3_v10_applicable(g) ? 1 : 0;
}
