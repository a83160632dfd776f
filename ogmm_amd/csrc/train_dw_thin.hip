// Weight gradient of the THIN dense layers of the training step:  dW[n][k] = sum_r dY[r][n] X[r][k]  with few channels
// (per-edge EdgeConv maps: n <= 256, k <= 128 over 2.6 M rows at B = 64; the 6 -> 64 edge layer; the 1 -> 64 positional layers).
// These are HBM-bound reductions (a few hundred flops per loaded byte at most), so they run in exact fp32 on
// v_mfma_f32_32x32x2_f32, whose operand layout matches a row-major "contraction over rows" product directly:
//   A operand: lane l holds A[m = l % 32][kk = l / 32]  ->  dY[row r0 + l/32][some column of the lane l%32]
//   B operand: lane l holds B[kk = l / 32][n = l % 32]  ->  X [row r0 + l/32][some column of the lane l%32]
// i.e. one MFMA consumes TWO rows.  A lane loads NV consecutive dY columns (and KV consecutive X columns) of its row with one
// vector load; component e of that vector feeds MFMA e, so MFMA (e, f) accumulates the outputs n = NV*j + e, k = KV*j' + f
// (a strided set of 32 x 32 outputs) -- any fixed assignment of output rows to lanes is as good as the contiguous one, the
// accumulators are un-permuted when they are written.  Per two rows a wave issues 2 vector loads and NV*KV MFMAs.
// Every stream (workgroup x row split) reduces a contiguous row range and writes its own partial [n][k]; the host sums them.
#include "ogmm_common.h"
#include <algorithm>

namespace {

using namespace ogmm;
using f32x16t = __attribute__((ext_vector_type(16))) float;

template <int V>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, bool ok, int valid, float (&v)[V]) {
    // `valid` = number of in-range columns starting at p (>= V when the whole vector is inside the matrix)
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.0f;
    if (!ok || valid <= 0) return;
    if (valid >= V) {
        if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else if constexpr (V == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
        else v[0] = p[0];
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) if (e < valid) v[e] = p[e];
    }
}

// one wave: output tile of NV*32 rows (n) x KV*32 columns (k), rows [r_lo, r_hi) of the contraction
template <int NV, int KV>
__global__ __launch_bounds__(256) void dw_thin_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                      int64_t R, int n, int k, int n_tiles, int k_tiles, int row_splits, int64_t rows_per_stream,
                                                      float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles = n_tiles * k_tiles;
    const int tile = wave % tiles, rs = wave / tiles;                  // 4 waves = tiles * row_splits (host guarantees)
    if (rs >= row_splits) return;
    const int tn = tile / k_tiles, tk = tile % k_tiles;
    const int64_t stream = (int64_t)blockIdx.x * row_splits + rs;
    const int64_t r_lo = stream * rows_per_stream, r_hi = min(R, r_lo + rows_per_stream);
    const int j = lane & 31, half = lane >> 5;
    const int n0 = tn * NV * 32 + NV * j, k0 = tk * KV * 32 + KV * j;
    const float* __restrict__ pa = dy + n0;
    const float* __restrict__ pb = x + k0;
    const int a_valid = n - n0, b_valid = k - k0;

    f32x16t acc[NV][KV];
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = 0; f < KV; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.0f;

    constexpr int U = 4;                                               // row pairs in flight
    for (int64_t r = r_lo; r < r_hi; r += 2 * U) {
        float a[U][NV], b[U][KV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r + 2 * u + half;
            const bool ok = row < r_hi;
            load_vec<NV>(pa + row * lddy, ok, a_valid, a[u]);
            load_vec<KV>(pb + row * ldx, ok, b_valid, b[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < NV; ++e)
#pragma unroll
                for (int f = 0; f < KV; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][f], acc[e][f], 0, 0, 0);
    }
    // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* __restrict__ out = part + stream * (int64_t)n * k;
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = 0; f < KV; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int nn = tn * NV * 32 + NV * mrow + e, kk = tk * KV * 32 + KV * j + f;
                if (nn < n && kk < k) out[(int64_t)nn * k + kk] = acc[e][f][r];
            }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same reduction on the fp16x3 arithmetic of the GEMM engines (three v_mfma_f32_32x32x16_f16 per product block: hi*hi + hi*lo + lo*hi,
// fp32 accumulation), for the training step's default precision.  The exact-fp32 form above is MATRIX-bound on the widest per-edge layer
// (256 x 128 outputs over 5.2 M rows: 343 GFLOP at the ~100 TFLOP/s of v_mfma_f32_32x32x2_f32 = 3.4 of its 4.7 ms; its 8 GB of operands are 1.5 ms
// of HBM time).  The f16 instruction contracts 16 rows where the fp32 one contracts 2, and its operand layout still matches a row-major
// "contraction over rows" product: lane l holds, for ITS column (l % 32), the 8 consecutive rows 8 * (l / 32) ... + 7 -- so a lane loads its NV (KV)
// columns of 8 rows (8 vector loads per operand, 16 in flight), splits every value into two binary16 in registers (hi = rn(v), lo = rn(v - hi)) and
// feeds 3 matrix instructions per (e, f) pair and 16 rows: 96 matrix cycles where the fp32 form needs 512.  Values above binary16's range set the
// overflow word (the trainer lowers its power-of-two loss scale and repeats the step, as for the engine's own operands).
using h8t = __attribute__((ext_vector_type(8))) _Float16;

// (a paired form -- one v_cvt_pk_f16_f32 per two values, the residuals from the packed register through v_cvt_f32_f16_sdwa -- has 40 % fewer vector
//  instructions and ran 4.7x SLOWER in this kernel: 8998 against 1885 us on the 256 x 128 layer, same box, same launch.  The instructions themselves are not
//  slow -- tools/sdwa_rate.hip measures all three sequences within 8 % of each other, subnormal results included -- so it is the schedule the compiler
//  made of it at 254 registers; kept per element)
template <int V>
__device__ __forceinline__ void split_rows(const float (&v)[8][V], h8t (&hi)[V], h8t (&lo)[V], float& amax) {
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = v[i][e];
            amax = fmaxf(amax, fabsf(t));
            const _Float16 h = (_Float16)t;
            hi[e][i] = h;
            lo[e][i] = (_Float16)(t - (float)h);
        }
}

template <int NV, int KV>
__global__ __launch_bounds__(256, 2) void dw_thin16_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                        int64_t R, int n, int k, int n_tiles, int k_tiles, int row_splits, int64_t rows_per_stream,
                                                        float* __restrict__ part, int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles = n_tiles * k_tiles;
    const int tile = wave % tiles, rs = wave / tiles;
    if (rs >= row_splits) return;
    const int tn = tile / k_tiles, tk = tile % k_tiles;
    const int64_t stream = (int64_t)blockIdx.x * row_splits + rs;
    const int64_t r_lo = stream * rows_per_stream, r_hi = min(R, r_lo + rows_per_stream);
    const int j = lane & 31, half = lane >> 5;
    const int n0 = tn * NV * 32 + NV * j, k0 = tk * KV * 32 + KV * j;
    const float* __restrict__ pa = dy + n0;
    const float* __restrict__ pb = x + k0;
    const int a_valid = n - n0, b_valid = k - k0;

    f32x16t acc[NV][KV];
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = 0; f < KV; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.0f;
    float amax = 0.0f;
    // (requesting step s+1's rows right after step s has been split -- a software pipeline on the same registers -- makes the compiler spill at the
    //  256-register cap; two waves per SIMD hide the load latency instead)
    for (int64_t r = r_lo; r < r_hi; r += 16) {
        float a[8][NV], b[8][KV];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = r + 8 * half + i;
            const bool ok = row < r_hi;
            load_vec<NV>(pa + row * lddy, ok, a_valid, a[i]);
            load_vec<KV>(pb + row * ldx, ok, b_valid, b[i]);
        }
        h8t ah[NV], al[NV], bh[KV], bl[KV];
        split_rows<NV>(a, ah, al, amax);
        split_rows<KV>(b, bh, bl, amax);
#pragma unroll
        for (int e = 0; e < NV; ++e)
#pragma unroll
            for (int f = 0; f < KV; ++f) {
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[e], bl[f], acc[e][f], 0, 0, 0);
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[e], bh[f], acc[e][f], 0, 0, 0);
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[e], bh[f], acc[e][f], 0, 0, 0);
            }
    }
    // v_max_f32 returns its non-NaN operand, so a NaN operand never shows in amax: NaN (and an Inf whose binary16 hi part met a zero)
    // is read off the accumulators instead -- every input element of the workgroup's slab multiplies into one of them
    float chk = 0.0f;
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = 0; f < KV; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk = fmaf(acc[e][f][r], 0.0f, chk);
    if (overflow && (!(amax <= 65504.0f) || chk != chk)) atomicOr(overflow, 1);          // an operand beyond binary16's range, Inf or NaN
    float* __restrict__ out = part + stream * (int64_t)n * k;
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = 0; f < KV; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int nn = tn * NV * 32 + NV * mrow + e, kk = tk * KV * 32 + KV * j + f;
                if (nn < n && kk < k) out[(int64_t)nn * k + kk] = acc[e][f][r];
            }
}

}  // namespace

// Host side: picks (NV, KV) so that NV*32 >= n or the n axis splits into tiles, ditto k; 4 waves = tiles x row splits.
extern "C" int64_t ogmm_weight_grad_thin_streams(int n, int k) {
    const int NV = n > 64 ? 4 : (n > 32 ? 2 : 1), KV = k > 32 ? 2 : 1;
    const int n_tiles = (n + NV * 32 - 1) / (NV * 32), k_tiles = (k + KV * 32 - 1) / (KV * 32);
    const int tiles = n_tiles * k_tiles;
    if (tiles > 4) return -1;
    const int row_splits = 4 / tiles;
    return (int64_t)1024 * row_splits;                                 // 1024 workgroups
}

static int weight_grad_thin_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t R, int n, int k, float* part, bool f16x3, int* overflow,
                                 void* stream) {
    OGMM_REQUIRE(dy && x && part && R > 0 && n > 0 && k > 0, "ogmm_weight_grad_thin: null pointer or empty input");
    const int NV = n > 64 ? 4 : (n > 32 ? 2 : 1), KV = k > 32 ? 2 : 1;
    OGMM_REQUIRE(lddy % NV == 0 && ldx % KV == 0 && (reinterpret_cast<uintptr_t>(dy) % (4 * NV)) == 0 && (reinterpret_cast<uintptr_t>(x) % (4 * KV)) == 0,
                 "ogmm_weight_grad_thin: rows of dy / x must be aligned to their %d- / %d-float vector loads", NV, KV);
    const int n_tiles = (n + NV * 32 - 1) / (NV * 32), k_tiles = (k + KV * 32 - 1) / (KV * 32);
    const int tiles = n_tiles * k_tiles;
    OGMM_REQUIRE(tiles <= 4, "ogmm_weight_grad_thin: n <= 256 with k <= 64, or n <= 128 with k <= 128, ... (n=%d, k=%d need %d wave tiles > 4)", n, k, tiles);
    const int row_splits = 4 / tiles;
    const int64_t streams = (int64_t)1024 * row_splits;
    int64_t rows_per_stream = (R + streams - 1) / streams;
    rows_per_stream = (rows_per_stream + 15) / 16 * 16;                // whole unrolled iterations (8 rows of the fp32 form, 16 of the fp16x3 form)
    dim3 grid(1024), block(256);
    hipStream_t s = as_stream(stream);
#define OGMM_DW_THIN(NVv, KVv)                                                                                                                            \
    do {                                                                                                                                                  \
        if (f16x3) hipLaunchKernelGGL((dw_thin16_kernel<NVv, KVv>), grid, block, 0, s, dy, lddy, x, ldx, R, n, k, n_tiles, k_tiles, row_splits,           \
                                      rows_per_stream, part, overflow);                                                                                   \
        else hipLaunchKernelGGL((dw_thin_kernel<NVv, KVv>), grid, block, 0, s, dy, lddy, x, ldx, R, n, k, n_tiles, k_tiles, row_splits, rows_per_stream,  \
                                part);                                                                                                                    \
    } while (0)
    if (NV == 4 && KV == 2) OGMM_DW_THIN(4, 2);
    else if (NV == 4) OGMM_DW_THIN(4, 1);
    else if (NV == 2 && KV == 2) OGMM_DW_THIN(2, 2);
    else if (NV == 2) OGMM_DW_THIN(2, 1);
    else if (KV == 2) OGMM_DW_THIN(1, 2);
    else OGMM_DW_THIN(1, 1);
#undef OGMM_DW_THIN
    return check_launch("ogmm_weight_grad_thin");
}

extern "C" int ogmm_weight_grad_thin(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t R, int n, int k, float* part, void* stream) {
    return weight_grad_thin_impl(dy, lddy, x, ldx, R, n, k, part, false, nullptr, stream);
}

extern "C" int ogmm_weight_grad_thin_f16x3(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t R, int n, int k, float* part, int* overflow,
                                           void* stream) {
    return weight_grad_thin_impl(dy, lddy, x, ldx, R, n, k, part, true, overflow, stream);
}
