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
#include "ogmm_common.h"

namespace {

using namespace ogmm;

constexpr int ROW_CHUNK = 512;       // rows of one group handled by one workgroup of the reduction kernels (smallest value)

// Rows per workgroup of a reduction kernel.  Every workgroup ends with one fp64 atomic per column and statistic on the SAME few
// addresses of its group: on the 5.2 M-row per-edge maps 512-row chunks meant 5120 atomics per address, and the kernels ran at
// 0.9-1.7 TB/s waiting for them.  Grow the chunk until ~2048 workgroups are left (still 8 per CU).
static int pick_row_chunk(int64_t group_rows, int64_t groups, int64_t col_slabs) {
    int64_t chunk = ROW_CHUNK;
    while (chunk < 16384 && ((group_rows + chunk - 1) / chunk) * groups * col_slabs > 2048) chunk *= 2;
    return (int)chunk;
}

__device__ __forceinline__ float act_grad(float y, int act) {      // derivative of the activation, from its output or its pre-activation (same sign)
    if (act == OGMM_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (act == OGMM_ACT_LEAKY02) return y > 0.0f ? 1.0f : 0.2f;
    return 1.0f;
}
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == OGMM_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == OGMM_ACT_LEAKY02) return v > 0.0f ? v : 0.2f * v;
    return v;
}

// All four kernels move float4 per lane: a wave covers 256 consecutive columns of one row (or 64 / 128 columns of 4 / 2 rows),
// a block of 256 threads = 4 waves covers `ROWS_PER_PASS` rows per pass.  cols % 4 == 0 and 16-byte aligned rows are required
// by the vector kernels; the scalar kernels below them take any shape.
template <int CV>   // CV = lanes per row = min(cols / 4, 64), a power of two
struct Map {
    static constexpr int rows_per_wave = 64 / CV;
    static constexpr int rows_per_pass = 4 * rows_per_wave;
};

// ---------------------------------------------------------------- column statistics: stats[g][c] = {sum x, sum x^2}
template <int CV>
__global__ __launch_bounds__(256) void colstats_v4_kernel(const float* __restrict__ x, int64_t ldx, int cols, int64_t group_rows,
                                                          double* __restrict__ stats, int row_chunk) {
    __shared__ double red[4][64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CV, rl = wave * Map<CV>::rows_per_wave + lane / CV;
    const int col = (blockIdx.y * CV + cl) * 4;
    const int g = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * row_chunk, r1 = min(r0 + row_chunk, group_rows);
    const float* __restrict__ base = x + ((int64_t)g * group_rows) * ldx + col;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (col < cols)
#pragma unroll 4
        for (int64_t r = r0 + rl; r < r1; r += Map<CV>::rows_per_pass) {
            const float4 v = *reinterpret_cast<const float4*>(base + r * ldx);
            const double a = v.x, b = v.y, c = v.z, d = v.w;
            s[0] += a; s[1] += b; s[2] += c; s[3] += d;
            ss[0] += a * a; ss[1] += b * b; ss[2] += c * c; ss[3] += d * d;
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][lane][e] = s[e]; red[wave][lane][4 + e] = ss[e]; }
    __syncthreads();
    // lanes (wave 0, lane < CV) gather the partial sums of every row lane that shares their columns
    if (wave == 0 && lane < CV && col < cols) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 4; ++w)
            for (int l = lane; l < 64; l += CV)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] += red[w][l][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e < cols) {
                atomicAdd(&stats[((int64_t)g * cols + col + e) * 2], t[e]);
                atomicAdd(&stats[((int64_t)g * cols + col + e) * 2 + 1], t[4 + e]);
            }
        }
    }
}

// ---------------------------------------------------------------- y = act(x * scale[g][c] + shift[g][c])
template <int CV>
__global__ __launch_bounds__(256) void affine_act_v4_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, int64_t group_rows,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                            float* __restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CV, rl = wave * Map<CV>::rows_per_wave + lane / CV;
    const int col = (blockIdx.y * CV + cl) * 4;
    if (col >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = min(r0 + 64, rows);
    const int64_t g0 = r0 / group_rows;
    const bool one_group = (r1 - 1) / group_rows == g0;          // (group_rows %% 64 == 0: the constants are loaded once per thread, see norm_bwd_apply_v4_kernel)
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    if (one_group) { sc = *reinterpret_cast<const float4*>(scale + g0 * cols + col); sh = *reinterpret_cast<const float4*>(shift + g0 * cols + col); }
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += Map<CV>::rows_per_pass) {
        if (!one_group) {
            const int64_t gc = (r / group_rows) * cols + col;
            sc = *reinterpret_cast<const float4*>(scale + gc); sh = *reinterpret_cast<const float4*>(shift + gc);
        }
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + col);
        float4 o;
        o.x = act_fwd(fmaf(v.x, sc.x, sh.x), act); o.y = act_fwd(fmaf(v.y, sc.y, sh.y), act);
        o.z = act_fwd(fmaf(v.z, sc.z, sh.z), act); o.w = act_fwd(fmaf(v.w, sc.w, sh.w), act);
        *reinterpret_cast<float4*>(y + r * ldy + col) = o;
    }
}

// Upstream gradient of one element of a normalised map: the dense part dy (may be absent) plus, for maps that were max-pooled
// over the k rows of a point (EdgeConv, models/dgcnn.py:139-148), the pooled gradient routed to the winning row (`arg`).
struct Routed {
    const float* dy; int64_t lddy;          // dense upstream gradient or NULL
    const float* dpool; int64_t ldp;        // [points][cols] gradient of the pooled map or NULL
    const uint8_t* arg; int k; int cols;    // winning row per (point, column)
};
__device__ __forceinline__ float4 routed_load4(const Routed& u, int64_t r, int col) {
    float4 v = u.dy ? *reinterpret_cast<const float4*>(u.dy + r * u.lddy + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (u.dpool) {
        const int64_t p = r / u.k;
        const int j = (int)(r - p * u.k);
        const uchar4 a = *reinterpret_cast<const uchar4*>(u.arg + p * u.cols + col);
        const float4 g = *reinterpret_cast<const float4*>(u.dpool + p * u.ldp + col);
        v.x += a.x == j ? g.x : 0.f; v.y += a.y == j ? g.y : 0.f; v.z += a.z == j ? g.z : 0.f; v.w += a.w == j ? g.w : 0.f;
    }
    return v;
}
__device__ __forceinline__ float routed_load1(const Routed& u, int64_t r, int col) {
    float v = u.dy ? u.dy[r * u.lddy + col] : 0.f;
    if (u.dpool) {
        const int64_t p = r / u.k;
        if (u.arg[p * u.cols + col] == (int)(r - p * u.k)) v += u.dpool[p * u.ldp + col];
    }
    return v;
}

// ---------------------------------------------------------------- y = act(x * scale + shift) (optional) and its max over the k rows of a point
// One lane owns (point, 4 columns) and walks the point's k rows: the normalised per-edge map is written only if a later layer
// reads it (the last EdgeConv layer and the positional angle branch only use the pooled map).
template <int CV>
__global__ __launch_bounds__(256) void affine_act_pool_v4_kernel(const float* __restrict__ x, int64_t ldx, int64_t points, int k, int cols,
                                                                 int64_t group_points, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 int act, float* __restrict__ y, int64_t ldy, float* __restrict__ pooled, int64_t ldp,
                                                                 uint8_t* __restrict__ arg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CV, pl = wave * Map<CV>::rows_per_wave + lane / CV;
    const int col = (blockIdx.y * CV + cl) * 4;
    if (col >= cols) return;
    const int64_t p0 = (int64_t)blockIdx.x * 16, p1 = min(p0 + 16, points);
    for (int64_t p = p0 + pl; p < p1; p += Map<CV>::rows_per_pass) {
        const int64_t gc = (p / group_points) * cols + col;
        const float4 sc = *reinterpret_cast<const float4*>(scale + gc), sh = *reinterpret_cast<const float4*>(shift + gc);
        float best[4];
        int bj[4] = {0, 0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const int64_t r = p * k + j;
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + col);
            const float o[4] = {act_fwd(fmaf(v.x, sc.x, sh.x), act), act_fwd(fmaf(v.y, sc.y, sh.y), act),
                                act_fwd(fmaf(v.z, sc.z, sh.z), act), act_fwd(fmaf(v.w, sc.w, sh.w), act)};
            if (y) *reinterpret_cast<float4*>(y + r * ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j == 0 || o[e] > best[e]) { best[e] = o[e]; bj[e] = j; }          // first maximum wins, like torch.max
        }
        *reinterpret_cast<float4*>(pooled + p * ldp + col) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<uchar4*>(arg + p * cols + col) = make_uchar4((uint8_t)bj[0], (uint8_t)bj[1], (uint8_t)bj[2], (uint8_t)bj[3]);
    }
}

// ---------------------------------------------------------------- backward reduction: sums[g][c] = {sum dz, sum dz * xhat}
// The activation derivative comes from the recomputed pre-activation x * scale + shift (the stored output is not re-read).
template <int CV>
__global__ __launch_bounds__(256) void norm_bwd_reduce_v4_kernel(const float* __restrict__ x, int64_t ldx, const Routed up,
                                                                 int cols, int64_t group_rows, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int act, double* __restrict__ sums, int row_chunk) {
    __shared__ double red[4][64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CV, rl = wave * Map<CV>::rows_per_wave + lane / CV;
    const int col = (blockIdx.y * CV + cl) * 4;
    const int g = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * row_chunk, r1 = min(r0 + row_chunk, group_rows);
    const int64_t gr = (int64_t)g * group_rows;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (col < cols) {
        const int64_t gc = (int64_t)g * cols + col;
        const float4 sc = *reinterpret_cast<const float4*>(scale + gc), sh = *reinterpret_cast<const float4*>(shift + gc);
        const float4 m = *reinterpret_cast<const float4*>(mean + gc), rs = *reinterpret_cast<const float4*>(rstd + gc);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, mv[4] = {m.x, m.y, m.z, m.w}, rv[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll 2
        for (int64_t r = r0 + rl; r < r1; r += Map<CV>::rows_per_pass) {
            const float4 xv4 = *reinterpret_cast<const float4*>(x + (gr + r) * ldx + col);
            const float4 dv4 = routed_load4(up, gr + r, col);
            const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dz = dv[e] * act_grad(fmaf(xv[e], scv[e], shv[e]), act);
                const float xh = (xv[e] - mv[e]) * rv[e];
                s1[e] += (double)dz;
                s2[e] += (double)dz * (double)xh;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][lane][e] = s1[e]; red[wave][lane][4 + e] = s2[e]; }
    __syncthreads();
    if (wave == 0 && lane < CV && col < cols) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 4; ++w)
            for (int l = lane; l < 64; l += CV)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] += red[w][l][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e < cols) {
                atomicAdd(&sums[((int64_t)g * cols + col + e) * 2], t[e]);
                atomicAdd(&sums[((int64_t)g * cols + col + e) * 2 + 1], t[4 + e]);
            }
        }
    }
}

// ---------------------------------------------------------------- dx = scale * (dz - S1/n - xhat * S2/n)
template <int CV>
__global__ __launch_bounds__(256) void norm_bwd_apply_v4_kernel(const float* __restrict__ x, int64_t ldx, const Routed up,
                                                                int64_t rows, int cols, int64_t group_rows, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int act, const double* __restrict__ sums,
                                                                float* __restrict__ dx, int64_t lddx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CV, rl = wave * Map<CV>::rows_per_wave + lane / CV;
    const int col = (blockIdx.y * CV + cl) * 4;
    if (col >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = min(r0 + 64, rows);
    const double inv_n = 1.0 / (double)group_rows;
    // the per-(group, column) constants of one row: 4 float4 + 8 doubles = 128 bytes beside the 32 bytes of x and dz -- re-read per row they made the kernel
    // bound by the CU's vector-memory path (3.9 TB/s of HBM traffic); a workgroup's 64 rows lie in one group whenever group_rows %% 64 == 0 (every
    // normalisation of the model), so they are loaded once per thread (round 4; same arithmetic, same results)
    float scv[4], shv[4], mv[4], rv[4], m1[4], m2[4];
    auto load_consts = [&](int64_t gc) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + gc), sh = *reinterpret_cast<const float4*>(shift + gc);
        const float4 m = *reinterpret_cast<const float4*>(mean + gc), rs = *reinterpret_cast<const float4*>(rstd + gc);
        scv[0] = sc.x; scv[1] = sc.y; scv[2] = sc.z; scv[3] = sc.w; shv[0] = sh.x; shv[1] = sh.y; shv[2] = sh.z; shv[3] = sh.w;
        mv[0] = m.x; mv[1] = m.y; mv[2] = m.z; mv[3] = m.w; rv[0] = rs.x; rv[1] = rs.y; rv[2] = rs.z; rv[3] = rs.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) { m1[e] = (float)(sums[(gc + e) * 2] * inv_n); m2[e] = (float)(sums[(gc + e) * 2 + 1] * inv_n); }
    };
    auto row = [&](int64_t r) {
        const float4 xv4 = *reinterpret_cast<const float4*>(x + r * ldx + col);
        const float4 dv4 = routed_load4(up, r, col);
        const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dz = dv[e] * act_grad(fmaf(xv[e], scv[e], shv[e]), act);
            const float xh = (xv[e] - mv[e]) * rv[e];
            o[e] = scv[e] * (dz - m1[e] - xh * m2[e]);
        }
        *reinterpret_cast<float4*>(dx + r * lddx + col) = make_float4(o[0], o[1], o[2], o[3]);
    };
    const int64_t g0 = r0 / group_rows;
    if ((r1 - 1) / group_rows == g0) {
        load_consts(g0 * cols + col);
#pragma unroll 4
        for (int64_t r = r0 + rl; r < r1; r += Map<CV>::rows_per_pass) row(r);
    } else {
        for (int64_t r = r0 + rl; r < r1; r += Map<CV>::rows_per_pass) {
            load_consts((r / group_rows) * cols + col);
            row(r);
        }
    }
}

// ---------------------------------------------------------------- scalar fallbacks (cols % 4 != 0 or unaligned rows)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int64_t ldx, int cols, int64_t group_rows,
                                                       double* __restrict__ stats, int row_chunk) {
    __shared__ double red[2][4][64];
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    const int g = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * row_chunk, r1 = min(r0 + row_chunk, group_rows);
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

__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const float* __restrict__ x, int64_t ldx, const Routed up,
                                                              int cols, int64_t group_rows, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int act, double* __restrict__ sums, int row_chunk) {
    __shared__ double red[2][4][64];
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    const int g = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * row_chunk, r1 = min(r0 + row_chunk, group_rows);
    const int64_t gr = (int64_t)g * group_rows;
    double s1 = 0.0, s2 = 0.0;
    if (col < cols) {
        const int64_t gc = (int64_t)g * cols + col;
        const float m = mean[gc], rs = rstd[gc], sc = scale[gc], sh = shift[gc];
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const int64_t rr = gr + r;
            const float xv = x[rr * ldx + col];
            const float dz = routed_load1(up, rr, col) * act_grad(fmaf(xv, sc, sh), act);
            s1 += (double)dz;
            s2 += (double)dz * (double)((xv - m) * rs);
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

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, int64_t ldx, const Routed up,
                                                             int64_t rows, int cols, int64_t group_rows, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int act, const double* __restrict__ sums,
                                                             float* __restrict__ dx, int64_t lddx) {
    const int ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + ch;
    if (col >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = min(r0 + 64, rows);
    const double inv_n = 1.0 / (double)group_rows;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
        const int64_t gc = (r / group_rows) * cols + col;
        const float m1 = (float)(sums[gc * 2] * inv_n), m2 = (float)(sums[gc * 2 + 1] * inv_n);
        const float xv = x[r * ldx + col];
        const float dz = routed_load1(up, r, col) * act_grad(fmaf(xv, scale[gc], shift[gc]), act);
        const float xh = (xv - mean[gc]) * rstd[gc];
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

// lanes per row for the float4 kernels: cols/4 rounded up to a power of two, at most 64
static int lanes_per_row(int cols) {
    int cv = 1;
    while (cv < 64 && cv * 4 < cols) cv <<= 1;
    return cv;
}
static bool vec_ok(const void* p, int64_t ld, int cols) { return cols % 4 == 0 && ld % 4 == 0 && aligned16(p); }

#define OGMM_DISPATCH_CV(cv, CALL)                 \
    switch (cv) {                                  \
        case 1: { constexpr int CV = 1; CALL; } break;   \
        case 2: { constexpr int CV = 2; CALL; } break;   \
        case 4: { constexpr int CV = 4; CALL; } break;   \
        case 8: { constexpr int CV = 8; CALL; } break;   \
        case 16: { constexpr int CV = 16; CALL; } break; \
        case 32: { constexpr int CV = 32; CALL; } break; \
        default: { constexpr int CV = 64; CALL; } break; \
    }

// stats -> the constants of one normalisation layer, in one launch (round 4: the host-side tensor expressions were ~17 small launches per layer and step):
// mean = s0 / n, var = max(s1 / n - mean^2, 0) (biased), rstd = 1 / sqrt(var + eps), scale = rstd * gamma, shift = beta - mean * scale -- all in fp64, the
// float outputs rounded once.
__global__ __launch_bounds__(256) void norm_finalize_kernel(const double* __restrict__ stats, int64_t total, int cols, double n, double eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale,
                                                            float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ rstd,
                                                            double* __restrict__ mean64, double* __restrict__ var64) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cols);
    const double m = stats[2 * i] / n;
    const double v = fmax(stats[2 * i + 1] / n - m * m, 0.0);
    const double r = 1.0 / sqrt(v + eps);
    const double sc = gamma ? r * (double)gamma[c] : r;
    const double sh = beta ? (double)beta[c] - m * sc : -m * sc;
    scale[i] = (float)sc; shift[i] = (float)sh; mean[i] = (float)m; rstd[i] = (float)r;
    mean64[i] = m; var64[i] = v;
}

int ogmm_norm_finalize(const double* stats, int64_t groups, int cols, int64_t group_rows, double eps, const float* gamma, const float* beta, float* scale,
                       float* shift, float* mean, float* rstd, double* mean64, double* var64, void* stream) {
    OGMM_REQUIRE(stats && scale && shift && mean && rstd && mean64 && var64 && groups > 0 && cols > 0 && group_rows > 0, "ogmm_norm_finalize: null pointer or empty shape");
    const int64_t total = groups * cols;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), stats, total, cols, (double)group_rows, eps,
                       gamma, beta, scale, shift, mean, rstd, mean64, var64);
    return check_launch("ogmm_norm_finalize");
}

// BatchNorm's running statistics after a train-mode forward over G sequential calls of the shared layer (torch.nn.BatchNorm1d semantics, momentum 0.1:
// running = (1 - momentum) running + momentum batch, the variance unbiased by n / (n - 1)), one launch for all groups and both buffers
__global__ __launch_bounds__(256) void bn_update_running_kernel(const double* __restrict__ mean64, const double* __restrict__ var64, int groups, int cols, double unbias,
                                                                float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                int64_t* __restrict__ num_batches) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && num_batches) *num_batches += groups;
    if (c >= cols) return;
    float rm = running_mean[c], rv = running_var[c];
    for (int g = 0; g < groups; ++g) {
        const float m = (float)mean64[(int64_t)g * cols + c], v = (float)(var64[(int64_t)g * cols + c] * unbias);
        rm = add_rn(mul_rn(rm, 1.0f - momentum), mul_rn(momentum, m));
        rv = add_rn(mul_rn(rv, 1.0f - momentum), mul_rn(momentum, v));
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}

int ogmm_bn_update_running(const double* mean64, const double* var64, int groups, int cols, int64_t group_rows, float momentum, float* running_mean,
                           float* running_var, int64_t* num_batches, void* stream) {
    OGMM_REQUIRE(mean64 && var64 && running_mean && running_var && groups > 0 && cols > 0 && group_rows > 0, "ogmm_bn_update_running: null pointer or empty shape");
    const double unbias = (double)group_rows / (double)(group_rows > 1 ? group_rows - 1 : 1);
    hipLaunchKernelGGL(bn_update_running_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, as_stream(stream), mean64, var64, groups, cols, unbias, momentum,
                       running_mean, running_var, num_batches);
    return check_launch("ogmm_bn_update_running");
}

// gradients of a normalisation layer's affine parameters from the backward sums: dgamma[c] = sum_g sums[g][c][1], dbeta[c] = sum_g sums[g][c][0] (fp64, rounded once)
__global__ __launch_bounds__(256) void norm_param_grads_kernel(const double* __restrict__ sums, int groups, int cols, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    double a = 0.0, b = 0.0;
    for (int g = 0; g < groups; ++g) { b += sums[((int64_t)g * cols + c) * 2]; a += sums[((int64_t)g * cols + c) * 2 + 1]; }
    dgamma[c] = (float)a;
    dbeta[c] = (float)b;
}

int ogmm_norm_param_grads(const double* sums, int groups, int cols, float* dgamma, float* dbeta, void* stream) {
    OGMM_REQUIRE(sums && dgamma && dbeta && groups > 0 && cols > 0, "ogmm_norm_param_grads: null pointer or empty shape");
    hipLaunchKernelGGL(norm_param_grads_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, as_stream(stream), sums, groups, cols, dgamma, dbeta);
    return check_launch("ogmm_norm_param_grads");
}

int ogmm_colstats(const float* x, int64_t ldx, int64_t rows, int cols, int64_t group_rows, double* stats, void* stream) {
    OGMM_REQUIRE(rows >= 0 && cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_colstats: rows=%lld must be a multiple of group_rows=%lld",
                 (long long)rows, (long long)group_rows);
    if (rows == 0) return 0;
    const int64_t G = rows / group_rows;
    OGMM_REQUIRE(G <= 65535, "ogmm_colstats: too many groups (%lld)", (long long)G);
    (void)hipMemsetAsync(stats, 0, sizeof(double) * 2 * G * cols, as_stream(stream));
    if (vec_ok(x, ldx, cols)) {
        const int cv = lanes_per_row(cols);
        const int slabs = (cols / 4 + cv - 1) / cv, row_chunk = pick_row_chunk(group_rows, G, slabs);
        dim3 grid((unsigned)((group_rows + row_chunk - 1) / row_chunk), slabs, (unsigned)G);
        OGMM_DISPATCH_CV(cv, hipLaunchKernelGGL(colstats_v4_kernel<CV>, grid, dim3(256), 0, as_stream(stream), x, ldx, cols, group_rows, stats, row_chunk));
    } else {
        const int slabs = (cols + 63) / 64, row_chunk = pick_row_chunk(group_rows, G, slabs);
        hipLaunchKernelGGL(colstats_kernel, dim3((unsigned)((group_rows + row_chunk - 1) / row_chunk), slabs, (unsigned)G), dim3(256), 0, as_stream(stream), x, ldx,
                           cols, group_rows, stats, row_chunk);
    }
    return check_launch("ogmm_colstats");
}

int ogmm_affine_act(const float* x, int64_t ldx, int64_t rows, int cols, int64_t group_rows, const float* scale, const float* shift, int act,
                    float* y, int64_t ldy, void* stream) {
    OGMM_REQUIRE(cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_affine_act: bad shape");
    OGMM_REQUIRE(act == OGMM_ACT_NONE || act == OGMM_ACT_RELU || act == OGMM_ACT_LEAKY02, "ogmm_affine_act: activation %d not supported", act);
    if (rows == 0) return 0;
    const unsigned rblocks = (unsigned)((rows + 63) / 64);
    if (vec_ok(x, ldx, cols) && vec_ok(y, ldy, cols) && aligned16(scale) && aligned16(shift)) {
        const int cv = lanes_per_row(cols);
        dim3 grid(rblocks, (cols / 4 + cv - 1) / cv);
        OGMM_DISPATCH_CV(cv, hipLaunchKernelGGL(affine_act_v4_kernel<CV>, grid, dim3(256), 0, as_stream(stream), x, ldx, rows, cols, group_rows, scale,
                                                shift, act, y, ldy));
    } else {
        hipLaunchKernelGGL(affine_act_kernel, dim3(rblocks, (cols + 63) / 64), dim3(256), 0, as_stream(stream), x, ldx, rows, cols, group_rows, scale,
                           shift, act, y, ldy);
    }
    return check_launch("ogmm_affine_act");
}

static int make_routed(Routed& up, const char* who, const float* dy, int64_t lddy, const float* dpool, int64_t ldp, const uint8_t* arg, int k,
                       int64_t rows, int cols) {
    OGMM_REQUIRE(dy || dpool, "%s: no upstream gradient (dy and dpool both NULL)", who);
    OGMM_REQUIRE(!dpool || (arg && k >= 1 && k <= 255 && rows % k == 0), "%s: pooled gradient needs arg and 1 <= k <= 255 dividing rows", who);
    up = Routed{dy, lddy, dpool, ldp, arg, k > 0 ? k : 1, cols};
    return 0;
}
static bool routed_vec_ok(const Routed& up, int cols) {
    return (!up.dy || vec_ok(up.dy, up.lddy, cols)) && (!up.dpool || (vec_ok(up.dpool, up.ldp, cols) && (reinterpret_cast<uintptr_t>(up.arg) & 3) == 0));
}

int ogmm_norm_bwd_reduce(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* dpool, int64_t ldp, const uint8_t* arg, int k,
                         int64_t rows, int cols, int64_t group_rows,
                         const float* scale, const float* shift, const float* mean, const float* rstd, int act, double* sums, void* stream) {
    OGMM_REQUIRE(cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_norm_bwd_reduce: bad shape");
    if (rows == 0) return 0;
    Routed up;
    if (make_routed(up, "ogmm_norm_bwd_reduce", dy, lddy, dpool, ldp, arg, k, rows, cols)) return 1;
    const int64_t G = rows / group_rows;
    OGMM_REQUIRE(G <= 65535, "ogmm_norm_bwd_reduce: too many groups");
    (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * G * cols, as_stream(stream));
    if (vec_ok(x, ldx, cols) && routed_vec_ok(up, cols) && aligned16(scale) && aligned16(shift) && aligned16(mean) && aligned16(rstd)) {
        const int cv = lanes_per_row(cols);
        const int slabs = (cols / 4 + cv - 1) / cv, row_chunk = pick_row_chunk(group_rows, G, slabs);
        dim3 grid((unsigned)((group_rows + row_chunk - 1) / row_chunk), slabs, (unsigned)G);
        OGMM_DISPATCH_CV(cv, hipLaunchKernelGGL(norm_bwd_reduce_v4_kernel<CV>, grid, dim3(256), 0, as_stream(stream), x, ldx, up, cols, group_rows,
                                                scale, shift, mean, rstd, act, sums, row_chunk));
    } else {
        const int slabs = (cols + 63) / 64, row_chunk = pick_row_chunk(group_rows, G, slabs);
        hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3((unsigned)((group_rows + row_chunk - 1) / row_chunk), slabs, (unsigned)G), dim3(256), 0, as_stream(stream),
                           x, ldx, up, cols, group_rows, scale, shift, mean, rstd, act, sums, row_chunk);
    }
    return check_launch("ogmm_norm_bwd_reduce");
}

int ogmm_norm_bwd_apply(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* dpool, int64_t ldp, const uint8_t* arg, int k,
                        int64_t rows, int cols, int64_t group_rows,
                        const float* scale, const float* shift, const float* mean, const float* rstd, int act, const double* sums,
                        float* dx, int64_t lddx, void* stream) {
    OGMM_REQUIRE(cols > 0 && group_rows > 0 && rows % group_rows == 0, "ogmm_norm_bwd_apply: bad shape");
    if (rows == 0) return 0;
    Routed up;
    if (make_routed(up, "ogmm_norm_bwd_apply", dy, lddy, dpool, ldp, arg, k, rows, cols)) return 1;
    const unsigned rblocks = (unsigned)((rows + 63) / 64);
    if (vec_ok(x, ldx, cols) && routed_vec_ok(up, cols) && vec_ok(dx, lddx, cols) && aligned16(scale) && aligned16(shift) && aligned16(mean) && aligned16(rstd)) {
        const int cv = lanes_per_row(cols);
        dim3 grid(rblocks, (cols / 4 + cv - 1) / cv);
        OGMM_DISPATCH_CV(cv, hipLaunchKernelGGL(norm_bwd_apply_v4_kernel<CV>, grid, dim3(256), 0, as_stream(stream), x, ldx, up, rows, cols,
                                                group_rows, scale, shift, mean, rstd, act, sums, dx, lddx));
    } else {
        hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(rblocks, (cols + 63) / 64), dim3(256), 0, as_stream(stream), x, ldx, up, rows, cols,
                           group_rows, scale, shift, mean, rstd, act, sums, dx, lddx);
    }
    return check_launch("ogmm_norm_bwd_apply");
}

int ogmm_affine_act_pool(const float* x, int64_t ldx, int64_t points, int k, int cols, int64_t group_points, const float* scale, const float* shift,
                         int act, float* y, int64_t ldy, float* pooled, int64_t ldp, uint8_t* arg, void* stream) {
    OGMM_REQUIRE(x && pooled && arg && cols > 0 && k >= 1 && k <= 255 && group_points > 0 && points % group_points == 0, "ogmm_affine_act_pool: bad shape");
    OGMM_REQUIRE(act == OGMM_ACT_NONE || act == OGMM_ACT_RELU || act == OGMM_ACT_LEAKY02, "ogmm_affine_act_pool: activation %d not supported", act);
    OGMM_REQUIRE(vec_ok(x, ldx, cols) && vec_ok(pooled, ldp, cols) && (!y || vec_ok(y, ldy, cols)) && aligned16(scale) && aligned16(shift) &&
                 (reinterpret_cast<uintptr_t>(arg) & 3) == 0, "ogmm_affine_act_pool: cols %% 4 == 0 and 16-byte aligned rows required");
    if (points == 0) return 0;
    const int cv = lanes_per_row(cols);
    dim3 grid((unsigned)((points + 15) / 16), (cols / 4 + cv - 1) / cv);
    OGMM_DISPATCH_CV(cv, hipLaunchKernelGGL(affine_act_pool_v4_kernel<CV>, grid, dim3(256), 0, as_stream(stream), x, ldx, points, k, cols, group_points,
                                            scale, shift, act, y, ldy, pooled, ldp, arg));
    return check_launch("ogmm_affine_act_pool");
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
