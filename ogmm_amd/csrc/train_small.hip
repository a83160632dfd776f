// Small batched products of the TRAINING step that have no matrix-core shape: one operand is thin (contraction over 1 ... 128 elements) or both are tiny
// (n, m <= 64 rows).  Round 5 ran them through torch.matmul / torch.bmm, i.e. as hipBLASLt `Cijk_*` kernels (profiles/round5_train_kernel_stats.txt: ~25 calls and
// ~1 ms per step); north_star asks for hand-written kernels only (missing 4 / next 5c).  Exact fp32 fmaf chains in ascending contraction order.
//
//   ogmm_small_bmm_nn   out[b][i][:] = sum_{j < m} S[b][i][j] X[b][j][:] (+ bias)      contraction m <= 1024 (thin is what it is built for), any number of rows, any D; S and X by strides
//       emd.conv1 / pos.conv_dis.0 / pos.conv_ang1.0 forward  y = x W^T       (models/dgcnn.py:121,138; models/attn.py:37-47: S = x [R, K <= 6], X = W^T)
//       their dX                                              dx = dy W       (S = dy [R, 64], X = W)
//       the Cout = 1 heads proj.3 / overlap.6 in training     y = x w + b     (models/gmmreg.py:36-37: S = x [R, 256], X = w^T [256, 1]) and their dx = dy w
//       feature-mean backward  df = gamma (dmu / (pi N + 1e-5))               (lib/utils.py:138-140: S = gamma [C, N, J], X = scaled dmu [C, J, D])
//       soft correspondences   corr = scores mu_t                             (models/dgcnn.py:109)
//       and the S^T dOut / dG X halves of the backward of the two forms
//   ogmm_small_bmm_nt   out[b][i][j] = alpha sum_d A[b][i][d] B[b][j][d]                any n, m (64 x 64 output tiles), any D
//       cluster-feature similarity of the matching  sim = fn_s fn_t^T          (lib/utils.py:222-226, models/dgcnn.py:107)
//       the Gram matrix of the clustering loss      [x; y] [x; y]^T / tau      (lib/loss.py:40-47)
//       and d(scores) = dcorr mu_t^T
#include "ogmm_common.h"
#include <algorithm>

namespace {

using namespace ogmm;

constexpr int NN_X_FLOATS = 12288;          // 48 KiB of X per pass (m x Dc)
constexpr int NN_S_FLOATS = 4096;           // 16 KiB of S rows per pass (RB x m)

// block = 256 threads: X[b][0..m)[d0 .. d0 + Dc) staged once, then row chunks of RB rows: S rows staged, thread (row lane, column quad) accumulates
template <bool VEC4>
__global__ __launch_bounds__(256) void small_bmm_nn_kernel(const float* __restrict__ S, int64_t sS_b, int64_t sS_i, int64_t sS_j, const float* __restrict__ X,
                                                           int64_t sX_b, int64_t sX_j, int64_t sX_d, const float* __restrict__ bias, int64_t rows, int m, int D, int Dc,
                                                           int RB, int64_t rows_per_block, float* __restrict__ out, int64_t sO_b, int64_t ldO) {
    extern __shared__ __attribute__((aligned(16))) float lds_nn[];
    float* xs = lds_nn;                       // [m][Dc_pad]
    const int b = blockIdx.y, d0 = blockIdx.z * Dc, dc = min(Dc, D - d0);
    const int dcp = (Dc + 3) & ~3;
    float* ss = lds_nn + (size_t)m * dcp;      // [RB][m]
    const float* __restrict__ Xb = X + b * sX_b;
    for (int e = threadIdx.x; e < m * dcp; e += 256) {
        const int j = e / dcp, d = e - j * dcp;
        xs[e] = d < dc ? Xb[j * sX_j + (int64_t)(d0 + d) * sX_d] : 0.0f;
    }
    constexpr int CW = VEC4 ? 4 : 1;
    const int cols = (dc + CW - 1) / CW;                      // column groups of this pass
    const int tpr = min(256, cols);                           // threads per row
    const int rpp = 256 / tpr;                                // rows per pass of the block
    const int rl = threadIdx.x / tpr, cl = threadIdx.x - rl * tpr;
    const float* __restrict__ Sb = S + b * sS_b;
    float* __restrict__ Ob = out + b * sO_b;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_block, r_hi = min(rows, r_lo + rows_per_block);
    for (int64_t r0 = r_lo; r0 < r_hi; r0 += RB) {
        const int nr = (int)min((int64_t)RB, r_hi - r0);
        __syncthreads();                                      // (first pass: X staged; later: everybody is done with the previous S rows)
        for (int e = threadIdx.x; e < nr * m; e += 256) {
            const int i = e / m, j = e - i * m;
            ss[e] = Sb[(r0 + i) * sS_i + j * sS_j];
        }
        __syncthreads();
        if (rl < rpp)
            for (int i = rl; i < nr; i += rpp) {
                const float* __restrict__ srow = ss + i * m;
                for (int c = cl; c < cols; c += tpr) {
                    float acc[CW];
#pragma unroll
                    for (int e = 0; e < CW; ++e) acc[e] = 0.0f;
                    for (int j = 0; j < m; ++j) {
                        const float s = srow[j];
                        if (VEC4) {
                            const float4 x = *reinterpret_cast<const float4*>(xs + j * dcp + c * 4);
                            acc[0] = fmaf(s, x.x, acc[0]); acc[1 % CW] = fmaf(s, x.y, acc[1 % CW]); acc[2 % CW] = fmaf(s, x.z, acc[2 % CW]); acc[3 % CW] = fmaf(s, x.w, acc[3 % CW]);
                        } else
                            acc[0] = fmaf(s, xs[j * dcp + c], acc[0]);
                    }
                    float* __restrict__ op = Ob + (r0 + i) * ldO + d0 + c * CW;
                    if (VEC4) {
                        float4 y = make_float4(acc[0], acc[1 % CW], acc[2 % CW], acc[3 % CW]);
                        if (bias) { const float4 bb = *reinterpret_cast<const float4*>(bias + d0 + c * 4); y.x += bb.x; y.y += bb.y; y.z += bb.z; y.w += bb.w; }
                        *reinterpret_cast<float4*>(op) = y;
                    } else
                        op[0] = acc[0] + (bias ? bias[d0 + c] : 0.0f);
                }
            }
    }
}

// one block per batch entry and 64 x 64 output tile: thread (ti, tj) of a 16 x 16 grid owns outputs i in {ti, ti + 16, ...}, j in {tj, tj + 16, ...} (n, m <= 64: a 4 x 4 register
// tile); A and B rows staged in chunks of 32 columns (row pitch 33: the 16 rows a half wave reads fall on distinct banks)
__global__ __launch_bounds__(256) void small_bmm_nt_kernel(const float* __restrict__ A, int64_t sA_b, int64_t ldA, const float* __restrict__ Bm, int64_t sB_b, int64_t ldB,
                                                           int n, int m, int D, float alpha, float* __restrict__ out, int64_t sO_b, int64_t ldO) {
    __shared__ float as[64][33], bs[64][33];
    const int b = blockIdx.x, ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;          // this block's 64 x 64 output tile
    const float* __restrict__ Ab = A + b * sA_b + (int64_t)i0 * ldA;
    const float* __restrict__ Bb = Bm + b * sB_b + (int64_t)j0 * ldB;
    n = min(64, n - i0);
    m = min(64, m - j0);
    float acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0f;
    for (int d0 = 0; d0 < D; d0 += 32) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 32; e += 256) {
            const int r = e >> 5, d = e & 31;
            as[r][d] = (r < n && d0 + d < D) ? Ab[(int64_t)r * ldA + d0 + d] : 0.0f;
            bs[r][d] = (r < m && d0 + d < D) ? Bb[(int64_t)r * ldB + d0 + d] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int d = 0; d < 32; ++d) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = as[ti + 16 * u][d]; bv[u] = bs[tj + 16 * u][d]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(av[u], bv[v], acc[u][v]);
        }
    }
    float* __restrict__ Ob = out + b * sO_b + (int64_t)i0 * ldO + j0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = ti + 16 * u, j = tj + 16 * v;
            if (i < n && j < m) Ob[(int64_t)i * ldO + j] = acc[u][v] * alpha;
        }
}

// out[rows[i]][:] += g[i][:]: the backward of a row gather (lib/utils.py:111-127: the anchors' and nearest points' gradient rows go back into their map).  One wave
// per gathered row, fp32 atomics (FPS picks are distinct per cloud; two clusters may share a nearest point -- then the order of their two additions is not fixed,
// as with the library's index_add_ this replaces)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ out, int64_t ldo, const int64_t* __restrict__ rows, const float* __restrict__ g,
                                                               int64_t ldg, int64_t n, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float* __restrict__ o = out + rows[i] * ldo;
    const float* __restrict__ gi = g + i * ldg;
    for (int d = lane; d < D; d += 64) unsafeAtomicAdd(o + d, gi[d]);
}

}  // namespace

extern "C" int ogmm_scatter_add_rows(float* out, int64_t ldo, int64_t out_rows, const int64_t* rows, const float* g, int64_t ldg, int64_t n, int D, void* stream) {
    OGMM_REQUIRE(out && rows && g && n >= 0 && D > 0 && ldo >= D && ldg >= D && out_rows > 0, "ogmm_scatter_add_rows: null pointer or bad sizes (n=%lld D=%d)", (long long)n, D);
    if (n == 0) return 0;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ogmm::as_stream(stream), out, ldo, rows, g, ldg, n, D);
    return ogmm::check_launch("ogmm_scatter_add_rows");
}

extern "C" int ogmm_small_bmm_nn(const float* S, int64_t sS_b, int64_t sS_i, int64_t sS_j, const float* X, int64_t sX_b, int64_t sX_j, int64_t sX_d, const float* bias,
                                 int batch, int64_t rows, int m, int D, float* out, int64_t sO_b, int64_t ldO, void* stream) {
    OGMM_REQUIRE(S && X && out && batch > 0 && batch <= 65535 && rows > 0 && m > 0 && m <= 1024 && D > 0 && ldO >= D,
                 "ogmm_small_bmm_nn: batch=%d rows=%lld m=%d (1..1024) D=%d ldO=%lld", batch, (long long)rows, m, D, (long long)ldO);
    const bool vec4 = D % 4 == 0 && ldO % 4 == 0 && sO_b % 4 == 0 && ogmm::aligned16(out) && (!bias || ogmm::aligned16(bias));
    int Dc = std::min(D, NN_X_FLOATS / m);
    if (vec4) Dc = std::max(4, Dc & ~3);
    const int d_chunks = (D + Dc - 1) / Dc;
    OGMM_REQUIRE(d_chunks <= 65535, "ogmm_small_bmm_nn: D=%d too wide for m=%d", D, m);
    const int RB = std::max(1, std::min(64, NN_S_FLOATS / m));
    // enough blocks to fill the chip, whole row chunks per block (X is staged once per block: keep >= 4 chunks per block when there are rows to spare)
    const int64_t chunks = (rows + RB - 1) / RB;
    int64_t blocks_x = std::min<int64_t>(chunks, std::max<int64_t>(1, 2048 / ((int64_t)batch * d_chunks)));
    const int64_t rows_per_block = (chunks + blocks_x - 1) / blocks_x * RB;
    blocks_x = (rows + rows_per_block - 1) / rows_per_block;
    const size_t lds = ((size_t)m * ((Dc + 3) & ~3) + (size_t)RB * m) * sizeof(float);
    const dim3 grid((unsigned)blocks_x, (unsigned)batch, (unsigned)d_chunks);
    if (vec4)
        hipLaunchKernelGGL(small_bmm_nn_kernel<true>, grid, dim3(256), lds, ogmm::as_stream(stream), S, sS_b, sS_i, sS_j, X, sX_b, sX_j, sX_d, bias, rows, m, D, Dc, RB,
                           rows_per_block, out, sO_b, ldO);
    else
        hipLaunchKernelGGL(small_bmm_nn_kernel<false>, grid, dim3(256), lds, ogmm::as_stream(stream), S, sS_b, sS_i, sS_j, X, sX_b, sX_j, sX_d, bias, rows, m, D, Dc, RB,
                           rows_per_block, out, sO_b, ldO);
    return ogmm::check_launch("ogmm_small_bmm_nn");
}

extern "C" int ogmm_small_bmm_nt(const float* A, int64_t sA_b, int64_t ldA, const float* Bm, int64_t sB_b, int64_t ldB, int batch, int n, int m, int D, float alpha,
                                 float* out, int64_t sO_b, int64_t ldO, void* stream) {
    OGMM_REQUIRE(A && Bm && out && batch > 0 && n > 0 && n <= 4096 && m > 0 && m <= 4096 && D > 0 && ldA >= D && ldB >= D && ldO >= m,
                 "ogmm_small_bmm_nt: batch=%d n=%d m=%d (1..4096 each) D=%d", batch, n, m, D);
    hipLaunchKernelGGL(small_bmm_nt_kernel, dim3((unsigned)batch, (unsigned)((n + 63) / 64), (unsigned)((m + 63) / 64)), dim3(256), 0, ogmm::as_stream(stream), A, sA_b, ldA, Bm, sB_b, ldB, n, m, D, alpha, out, sO_b, ldO);
    return ogmm::check_launch("ogmm_small_bmm_nt");
}
