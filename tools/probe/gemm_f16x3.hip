// The split-precision 1x1-convolution engine: fp32-accurate NT GEMM on the binary16 matrix cores.
//
// Every fp32 operand is represented as hi + lo with hi = rn16(x), lo = rn16(x - hi) (x - hi is exact in fp32), i.e.
// 22 significand bits, and each product is evaluated as hi*hi + hi*lo + lo*hi by three v_mfma_f32_32x32x16_f16
// with fp32 accumulation: 3 MFMAs at the 2.5 PFLOP/s rate replace 8 v_mfma_f32_32x32x2_f32 at the 157 TFLOP/s rate
// for the same 32x32x16 block (5.3x fewer matrix-pipe cycles).  The dropped lo*lo term is <= 2^-22 relative.
// SURVEY.md section 7 requires 1e-5 rad end-to-end: a CPU emulation of this arithmetic inside the oracle gives
// R errors of 2e-7..3e-6 rad on the golden cases, the same as the exact-fp32 engine (bf16 x 2 terms: up to 5.8e-6,
// a single fp16 term: 1e-4..4e-4 -- both rejected).
//
// Operands: A (activations, fp32 in HBM) is split while it is staged into LDS; B (weights) is pre-split once at pack
// time (B_hi / B_lo, binary16, optionally scaled by a power of two so that lo stays a normal number).
// LDS image per operand half: [rows][40] binary16 (32 k's + 8 of padding = 80-byte rows: conflict-free ds_read_b128).
// Same tiling, XCD-aware tile map and epilogue as the fp32 engine (gemm.hip / gemm_common.h).
#include "gemm_common.h"

namespace {

using namespace ogmm_gemm_detail;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int BKH = 32;          // k's per LDS tile
constexpr int LDH = BKH + 8;     // padded row length in binary16 elements

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo, bool& ovf) {
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

template <int MT, int NT, int WM, int WN, bool POOL, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_f16x3_kernel(const ogmm_gemm g, const int rows_per_tile,
                                                                     const int m_tiles, const int n_tiles) {
    constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;
    constexpr int A_PIECES = BM * 8, B_PIECES = BN * 4;            // A: float4 pieces (4 k's); B: 16-byte pieces (8 k's) per half
    constexpr int A_P = (A_PIECES + T - 1) / T, B_P = (B_PIECES + T - 1) / T;
    __shared__ __attribute__((aligned(16))) _Float16 smem_h[2 * (BM + BN) * LDH];
    _Float16* Ah = smem_h;
    _Float16* Al = Ah + BM * LDH;
    _Float16* Bh = Al + BM * LDH;
    _Float16* Bl = Bh + BN * LDH;

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
    const _Float16* __restrict__ BH = reinterpret_cast<const _Float16*>(g.B_hi) + zo * g.sB_o + zi * g.sB_i;
    const _Float16* __restrict__ BL = reinterpret_cast<const _Float16*>(g.B_lo) + zo * g.sB_o + zi * g.sB_i;

    const int m0 = tile_m * rows_per_tile, n0 = tile_n * BN;
    const int m_end = min(g.M, m0 + rows_per_tile);
    const int nk1 = (g.K1 + BKH - 1) / BKH, nk2 = (g.K2 + BKH - 1) / BKH, nk = nk1 + nk2;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[A_P];
    f16x8 rbh[B_P], rbl[B_P];
    unsigned ra_ok = 0, rb_ok = 0;     // zero-selects are applied when the registers are consumed, not right after the load
    bool ovf = false;
    auto load_tile = [&](int t) {
        const bool second = t >= nk1;
        const float* Ap = second ? A2 : A;
        const int64_t ld = second ? g.lda2 : g.lda;
        const int kbase = second ? (t - nk1) * BKH : t * BKH;
        const int Kp = second ? g.K2 : g.K1;
        const int kB = second ? g.K1 + kbase : kbase;
        // unconditional loads from clamped (always valid) addresses + select: a branch around a load makes hipcc
        // serialise the loads behind per-load waits
        ra_ok = 0;
        rb_ok = 0;
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int f = tid + i * T, row = f >> 3, kq = (f & 7) * 4;
            const int gm = m0 + row;
            const bool ok = (A_PIECES % T == 0 || f < A_PIECES) && gm < m_end && kbase + kq < Kp;
            const int gmc = min(gm, g.M - 1), kc = ok ? kbase + kq : 0;
            ra[i] = *reinterpret_cast<const f32x4*>(Ap + (int64_t)gmc * ld + kc);
            ra_ok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int f = tid + i * T, row = f >> 2, kq = (f & 3) * 8;
            const int gn = n0 + row;
            const bool ok = (B_PIECES % T == 0 || f < B_PIECES) && gn < g.N && kB + kq < g.ldb_h;
            const int gnc = min(gn, g.N - 1), kc = ok ? kB + kq : 0;
            rbh[i] = *reinterpret_cast<const f16x8*>(BH + (int64_t)gnc * g.ldb_h + kc);
            rbl[i] = *reinterpret_cast<const f16x8*>(BL + (int64_t)gnc * g.ldb_h + kc);
            rb_ok |= (ok ? 1u : 0u) << i;
        }
    };

    load_tile(0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int f = tid + i * T;
            if (A_PIECES % T == 0 || f < A_PIECES) {
                f16x4 hi, lo;
                if (ABL == 2 || ABL == 3) {            // ablation: no conversion work (bit copies keep the loads alive)
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    const f32x2 p0 = {ra[i][0], ra[i][1]}, p1 = {ra[i][2], ra[i][3]};
                    hi = __builtin_bit_cast(f16x4, p0);
                    lo = __builtin_bit_cast(f16x4, p1);
                } else {
                    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                    split4(((ra_ok >> i) & 1u) ? ra[i] : zero4, hi, lo, ovf);
                }
                const int off = (f >> 3) * LDH + (f & 7) * 4;
                *reinterpret_cast<f16x4*>(&Ah[off]) = hi;
                *reinterpret_cast<f16x4*>(&Al[off]) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int f = tid + i * T;
            if (B_PIECES % T == 0 || f < B_PIECES) {
                const int off = (f >> 2) * LDH + (f & 3) * 8;
                const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
                const bool okb = (rb_ok >> i) & 1u;
                *reinterpret_cast<f16x8*>(&Bh[off]) = okb ? rbh[i] : zero8;
                *reinterpret_cast<f16x8*>(&Bl[off]) = okb ? rbl[i] : zero8;
            }
        }
        __syncthreads();
        if (t + 1 < nk && ABL != 1 && ABL != 3) load_tile(t + 1);
#pragma unroll
        for (int s = 0; s < BKH / 16; ++s) {
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int off = ((wm * MT + i) * 32 + lr) * LDH + s * 16 + lh * 8;
                ah[i] = *reinterpret_cast<const f16x8*>(&Ah[off]);
                al[i] = *reinterpret_cast<const f16x8*>(&Al[off]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int off = ((wn * NT + j) * 32 + lr) * LDH + s * 16 + lh * 8;
                bh[j] = *reinterpret_cast<const f16x8*>(&Bh[off]);
                bl[j] = *reinterpret_cast<const f16x8*>(&Bl[off]);
            }
            // term-major order: consecutive MFMAs hit different accumulators (no dependent-accumulator stalls)
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
        }
    }
    if (g.overflow && ovf) atomicOr(g.overflow, 1);
    __syncthreads();    // LDS is reused by the pooling epilogue
    gemm_epilogue<MT, NT, WM, WN, POOL>(g, acc, reinterpret_cast<float*>(smem_h), m0, n0, m_end, zo, zi, g.alpha);
}

template <int MT, int NT, int WM, int WN, bool POOL, int ABL = 0>
int launch_h(const ogmm_gemm& g, hipStream_t stream) {
    constexpr int BM = MT * 32 * WM, BN = NT * 32 * WN, T = WM * WN * 64;
    const int rows_per_tile = POOL ? (BM / g.pool_k) * g.pool_k : BM;
    const int m_tiles = (g.M + rows_per_tile - 1) / rows_per_tile;
    const int n_tiles = (g.N + BN - 1) / BN;
    const int m_tiles8 = (m_tiles + 7) / 8 * 8;
    dim3 grid((unsigned)(m_tiles8 * n_tiles), 1, (unsigned)(g.batch_outer * g.batch_inner));
    hipLaunchKernelGGL((gemm_nt_f16x3_kernel<MT, NT, WM, WN, POOL, ABL>), grid, dim3(T), 0, stream, g, rows_per_tile, m_tiles, n_tiles);
    return ogmm::check_launch("ogmm_gemm_nt(f16x3)");
}

}  // namespace

namespace ogmm {

// called by ogmm_gemm_nt (gemm.hip) after the common argument checks
int gemm_nt_f16x3(const ogmm_gemm& g, hipStream_t s) {
    OGMM_REQUIRE(g.B_hi && g.B_lo && g.ldb_h > 0 && g.ldb_h % 8 == 0 && aligned16(g.B_hi) && aligned16(g.B_lo),
                 "ogmm_gemm_nt(f16x3): needs pre-split B_hi/B_lo (16-byte aligned, ldb_h a multiple of 8)");
    OGMM_REQUIRE(g.sB_o % 8 == 0 && g.sB_i % 8 == 0, "ogmm_gemm_nt(f16x3): B batch strides must be multiples of 8");
    OGMM_REQUIRE(g.K1 + g.K2 <= g.ldb_h, "ogmm_gemm_nt(f16x3): K1+K2=%d exceeds ldb_h=%lld", g.K1 + g.K2, (long long)g.ldb_h);
    if (g.pool_k > 0) return g.N <= 64 ? launch_h<5, 1, 1, 2, true>(g, s) : launch_h<5, 1, 1, 4, true>(g, s);
    switch (g.precision) {      // codes > 10: tile-shape experiments (tools/gemm_bench.py)
        case 12: return launch_h<4, 2, 2, 2, false>(g, s);     // 256 x 128, 4 waves of 128 x 64
        case 13: return launch_h<2, 2, 4, 2, false>(g, s);     // 256 x 128, 8 waves of  64 x 64
        case 14: return launch_h<4, 2, 2, 4, false>(g, s);     // 256 x 256, 8 waves of 128 x 64
        case 15: return launch_h<2, 4, 2, 2, false>(g, s);     // 128 x 256, 4 waves of  64 x 128
        case 16: return launch_h<4, 2, 2, 4, false, 1>(g, s);  // ablation: no global loads after tile 0
        case 17: return launch_h<4, 2, 2, 4, false, 2>(g, s);  // ablation: no split arithmetic
        case 18: return launch_h<4, 2, 2, 4, false, 3>(g, s);  // ablation: both
        default: break;
    }
    return g.N <= 64 ? launch_h<2, 1, 2, 2, false>(g, s) : launch_h<2, 2, 2, 2, false>(g, s);
}

}  // namespace ogmm
