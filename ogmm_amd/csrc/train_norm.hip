// Training-mode normalisation and pooling kernels (forward AND backward), all HBM-bound.
//
//   BatchNorm in .train() (models/dgcnn.py:126-130, :21-27; models/attn.py:34-57) and InstanceNorm1d (models/attn.py:24)
//   are the same computation over different row groups of a point-major [rows][cols] map: a group is the src (or tgt)
//   half of the stacked batch for BatchNorm -- the reference calls each shared layer once per cloud set -- and one cloud
//   for InstanceNorm.  Per (group, column): mean / biased variance -> y = act(x * scale + shift) with
//   scale = gamma * rstd, shift = beta - mean * scale.
//   Backward (dz = dy * act'(y)):  dx = scale * (dz - mean_g(dz) - xhat * mean_g(dz * xhat)),
//                                  dgamma = sum dz * xhat, dbeta = sum dz.
//   Column sums accumulate in fp64 (per thread, then one atomic per block and column): gradients of this network are
//   ill-conditioned in fp32 (tests/golden/make_golden_train.py), so the reductions must not add noise of their own.
//
//   Grid: row blocks on x (can exceed 65535), 64-column slabs on y, groups on z.
//   Thread mapping everywhere: 64 consecutive columns on the 64 lanes of a wave (256-byte lines), 4 waves = 4 row lanes.
#include "ogmm_common.h"

namespace {

using namespace ogmm;

constexpr int ROW_CHUNK = 512;       // rows of one group handled by one workgroup of the reduction kernels

__device__ __forceinline__ float act_grad(float y, int act) {      // derivative of the activation, from its OUTPUT
    if (act == OGMM_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (act == OGMM_ACT_LEAKY02) return y > 0.0f ? 1.0f : 0.2f;
    return 1.0f;
}
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == OGMM_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == OGMM_ACT_LEAKY02) return v > 0.0f ? v : 0.2f * v;
    return v;
}

// ---------------------------------------------------------------- column statistics: stats[g][c] = {sum x, sum x^2}
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int64_t ldx, int cols, int64_t group_rows,
                                                       double* __restrict__ stats) {
    __shared__ double red[2][4][64];
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    const int g = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * ROW_CHUNK, r1 = min(r0 + ROW_CHUNK, group_rows);
    const float* __restrict__ base = x + ((int64_t)g * group_rows) * ldx + col;
    double s = 0.0, ss = 0.0;
    if (col < cols)
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const double v = base[r * ldx];
            s += v;
            ss += v * v;
        }
    red[0][rl][ch] = s;
    red[1][rl][ch] = ss;
    __syncthreads();
    if (rl < 2 && col < cols) {
        const double t = red[rl][0][ch] + red[rl][1][ch] + red[rl][2][ch] + red[rl][3][ch];
        atomicAdd(&stats[((int64_t)g * cols + col) * 2 + rl], t);
    }
}

// ---------------------------------------------------------------- y = act(x * scale[g][c] + shift[g][c])
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, int64_t group_rows,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                         float* __restrict__ y, int64_t ldy) {
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    if (col >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = min(r0 + 64, rows);
    for (int64_t r = r0 + rl; r < r1; r += 4) {
        const int64_t g = r / group_rows;
        y[r * ldy + col] = act_fwd(fmaf(x[r * ldx + col], scale[g * cols + col], shift[g * cols + col]), act);
    }
}

// ---------------------------------------------------------------- backward reduction: sums[g][c] = {sum dz, sum dz * xhat}
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ dy, int64_t lddy, int cols, int64_t group_rows,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                              double* __restrict__ sums) {
    __shared__ double red[2][4][64];
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    const int g = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * ROW_CHUNK, r1 = min(r0 + ROW_CHUNK, group_rows);
    const int64_t gr = (int64_t)g * group_rows;
    double s1 = 0.0, s2 = 0.0;
    if (col < cols) {
        const float m = mean[(int64_t)g * cols + col], rs = rstd[(int64_t)g * cols + col];
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const int64_t rr = gr + r;
            const float dz = dy[rr * lddy + col] * act_grad(y[rr * ldy + col], act);
            const float xh = (x[rr * ldx + col] - m) * rs;
            s1 += (double)dz;
            s2 += (double)dz * (double)xh;
        }
    }
    red[0][rl][ch] = s1;
    red[1][rl][ch] = s2;
    __syncthreads();
    if (rl < 2 && col < cols) {
        const double t = red[rl][0][ch] + red[rl][1][ch] + red[rl][2][ch] + red[rl][3][ch];
        atomicAdd(&sums[((int64_t)g * cols + col) * 2 + rl], t);
    }
}

// ---------------------------------------------------------------- dx = scale * (dz - S1/n - xhat * S2/n)
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ dy, int64_t lddy, int64_t rows, int cols, int64_t group_rows,
                                                             const float* __restrict__ scale, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int act, const double* __restrict__ sums,
                                                             float* __restrict__ dx, int64_t lddx) {
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    if (col >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = min(r0 + 64, rows);
    const double inv_n = 1.0 / (double)group_rows;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
        const int64_t g = r / group_rows;
        const int64_t gc = g * cols + col;
        const float m1 = (float)(sums[gc * 2] * inv_n), m2 = (float)(sums[gc * 2 + 1] * inv_n);
        const float dz = dy[r * lddy + col] * act_grad(y[r * ldy + col], act);
        const float xh = (x[r * ldx + col] - mean[gc]) * rstd[gc];
        dx[r * lddx + col] = scale[gc] * (dz - m1 - xh * m2);
    }
}

// ---------------------------------------------------------------- max over the k consecutive rows of a point (+ winning edge), and its backward
__global__ __launch_bounds__(256) void maxpool_k_kernel(const float* __restrict__ h, int64_t ldh, int64_t points, int k, int cols,
                                                        float* __restrict__ out, int64_t ldo, uint8_t* __restrict__ arg) {
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    if (col >= cols) return;
    const int64_t p0 = (int64_t)blockIdx.x * 16, p1 = min(p0 + 16, points);
    for (int64_t p = p0 + rl; p < p1; p += 4) {
        const float* __restrict__ src = h + p * k * ldh + col;
        float best = src[0];
        int bj = 0;
        for (int j = 1; j < k; ++j) {
            const float v = src[(int64_t)j * ldh];
            if (v > best) { best = v; bj = j; }             // first maximum wins, like torch.max
        }
        out[p * ldo + col] = best;
        arg[p * cols + col] = (uint8_t)bj;
    }
}

__global__ __launch_bounds__(256) void maxpool_k_bwd_kernel(const float* __restrict__ dout, int64_t ldo, const uint8_t* __restrict__ arg,
                                                            int64_t points, int k, int cols, float* __restrict__ dh, int64_t ldh) {
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    if (col >= cols) return;
    const int64_t p0 = (int64_t)blockIdx.x * 16, p1 = min(p0 + 16, points);
    for (int64_t p = p0 + rl; p < p1; p += 4) {
        const float g = dout[p * ldo + col];
        const int bj = arg[p * cols + col];
        float* __restrict__ dst = dh + p * k * ldh + col;
        for (int j = 0; j < k; ++j) dst[(int64_t)j * ldh] = j == bj ? g : 0.0f;
    }
}

}  // namespace

extern "C" {

int ogmm_colstats(const float* x, int64_t ldx, int64_t rows, int cols, int64_t group_rows, double* stats, void* stream) {
    OGMM_REQUIRE(rows >= 0 && cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_colstats: rows=%lld must be a multiple of group_rows=%lld",
                 (long long)rows, (long long)group_rows);
    if (rows == 0) return 0;
    const int64_t G = rows / group_rows;
    OGMM_REQUIRE(G <= 65535, "ogmm_colstats: too many groups (%lld)", (long long)G);
    hipMemsetAsync(stats, 0, sizeof(double) * 2 * G * cols, as_stream(stream));
    dim3 grid((unsigned)((group_rows + ROW_CHUNK - 1) / ROW_CHUNK), (cols + 63) / 64, (unsigned)G);
    hipLaunchKernelGGL(colstats_kernel, grid, dim3(256), 0, as_stream(stream), x, ldx, cols, group_rows, stats);
    return check_launch("ogmm_colstats");
}

int ogmm_affine_act(const float* x, int64_t ldx, int64_t rows, int cols, int64_t group_rows, const float* scale, const float* shift, int act,
                    float* y, int64_t ldy, void* stream) {
    OGMM_REQUIRE(cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_affine_act: bad shape");
    OGMM_REQUIRE(act == OGMM_ACT_NONE || act == OGMM_ACT_RELU || act == OGMM_ACT_LEAKY02, "ogmm_affine_act: activation %d not supported", act);
    if (rows == 0) return 0;
    dim3 grid((unsigned)((rows + 63) / 64), (cols + 63) / 64);
    hipLaunchKernelGGL(affine_act_kernel, grid, dim3(256), 0, as_stream(stream), x, ldx, rows, cols, group_rows, scale, shift, act, y, ldy);
    return check_launch("ogmm_affine_act");
}

int ogmm_norm_bwd_reduce(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, int64_t rows, int cols,
                         int64_t group_rows, const float* mean, const float* rstd, int act, double* sums, void* stream) {
    OGMM_REQUIRE(cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_norm_bwd_reduce: bad shape");
    if (rows == 0) return 0;
    const int64_t G = rows / group_rows;
    OGMM_REQUIRE(G <= 65535, "ogmm_norm_bwd_reduce: too many groups");
    hipMemsetAsync(sums, 0, sizeof(double) * 2 * G * cols, as_stream(stream));
    dim3 grid((unsigned)((group_rows + ROW_CHUNK - 1) / ROW_CHUNK), (cols + 63) / 64, (unsigned)G);
    hipLaunchKernelGGL(norm_bwd_reduce_kernel, grid, dim3(256), 0, as_stream(stream), x, ldx, y, ldy, dy, lddy, cols, group_rows, mean, rstd, act, sums);
    return check_launch("ogmm_norm_bwd_reduce");
}

int ogmm_norm_bwd_apply(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, int64_t rows, int cols,
                        int64_t group_rows, const float* scale, const float* mean, const float* rstd, int act, const double* sums,
                        float* dx, int64_t lddx, void* stream) {
    OGMM_REQUIRE(cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_norm_bwd_apply: bad shape");
    if (rows == 0) return 0;
    dim3 grid((unsigned)((rows + 63) / 64), (cols + 63) / 64);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, grid, dim3(256), 0, as_stream(stream), x, ldx, y, ldy, dy, lddy, rows, cols, group_rows, scale, mean,
                       rstd, act, sums, dx, lddx);
    return check_launch("ogmm_norm_bwd_apply");
}

int ogmm_maxpool_k(const float* h, int64_t ldh, int64_t points, int k, int cols, float* out, int64_t ldo, uint8_t* arg, void* stream) {
    OGMM_REQUIRE(k >= 1 && k <= 255 && cols > 0, "ogmm_maxpool_k: k must be 1..255");
    if (points == 0) return 0;
    dim3 grid((unsigned)((points + 15) / 16), (cols + 63) / 64);
    hipLaunchKernelGGL(maxpool_k_kernel, grid, dim3(256), 0, as_stream(stream), h, ldh, points, k, cols, out, ldo, arg);
    return check_launch("ogmm_maxpool_k");
}

int ogmm_maxpool_k_bwd(const float* dout, int64_t ldo, const uint8_t* arg, int64_t points, int k, int cols, float* dh, int64_t ldh, void* stream) {
    OGMM_REQUIRE(k >= 1 && k <= 255 && cols > 0, "ogmm_maxpool_k_bwd: k must be 1..255");
    if (points == 0) return 0;
    dim3 grid((unsigned)((points + 15) / 16), (cols + 63) / 64);
    hipLaunchKernelGGL(maxpool_k_bwd_kernel, grid, dim3(256), 0, as_stream(stream), dout, ldo, arg, points, k, cols, dh, ldh);
    return check_launch("ogmm_maxpool_k_bwd");
}

}  // extern "C"
