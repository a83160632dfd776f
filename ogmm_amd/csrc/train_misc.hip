// Small training-mode kernels: constants of the input that feed trainable thin layers, and the backward of the channel
// L2-normalisation.
//   edge features  [x_j - x_i ; x_i]                      lib/utils.py:47-66 (input of emd.conv1, models/dgcnn.py:137)
//   positional inputs |p - centroid|^2 and cos(angle)     models/attn.py:60-70 (inputs of pos.conv_dis.0 / pos.conv_ang1.0)
//   d/dx of x / max(|x|, 1e-12)                           models/gmmreg.py:74 (F.normalize over channels)
#include "ogmm_common.h"
#include <algorithm>

namespace {

using namespace ogmm;

__global__ __launch_bounds__(256) void edge_features_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int N, int k,
                                                            int64_t edges, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= edges) return;
    const int64_t p = e / k;                         // global point row = cloud * N + i
    const int64_t cloud = p / N;
    const float* __restrict__ ctr = xyz + p * 3;
    const float* __restrict__ nb = xyz + (cloud * N + idx[e]) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
    float* __restrict__ o = out + e * 6;
    o[0] = nb[0] - cx; o[1] = nb[1] - cy; o[2] = nb[2] - cz;
    o[3] = cx; o[4] = cy; o[5] = cz;
}

__global__ __launch_bounds__(256) void pos_features_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int N, int k,
                                                           int64_t points, const float* __restrict__ centroid,
                                                           float* __restrict__ d2, float* __restrict__ alpha) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= points) return;
    const int64_t cloud = p / N;
    const float* __restrict__ me = xyz + p * 3;
    const float* __restrict__ cen = centroid + cloud * 3;
    const float gx = me[0] - cen[0], gy = me[1] - cen[1], gz = me[2] - cen[2];
    const float g2 = gx * gx + gy * gy + gz * gz;
    d2[p] = g2;
    const float gi = 1.0f / fmaxf(sqrtf(g2), 1e-12f);
    for (int j = 0; j < k; ++j) {
        const float* __restrict__ nb = xyz + (cloud * N + idx[p * k + j]) * 3;
        const float lx = nb[0] - me[0], ly = nb[1] - me[1], lz = nb[2] - me[2];
        const float li = 1.0f / fmaxf(sqrtf(lx * lx + ly * ly + lz * lz), 1e-12f);
        alpha[p * k + j] = (lx * li) * (gx * gi) + (ly * li) * (gy * gi) + (lz * li) * (gz * gi);
    }
}

// dx = g / n - x (x.g) / n^3,  n = max(|x|, 1e-12); one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g, int64_t ldg,
                                                              int64_t rows, int D, float* __restrict__ dx, int64_t lddx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* __restrict__ px = x + row * ldx;
    const float* __restrict__ pg = g + row * ldg;
    float ss = 0.0f, dot = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float xv = px[d];
        ss += xv * xv;
        dot += xv * pg[d];
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float nrm = sqrtf(ss);
    float* __restrict__ q = dx + row * lddx;
    if (nrm > 1e-12f) {
        const float inv = 1.0f / nrm, c = dot * inv * inv * inv;
        for (int d = lane; d < D; d += 64) q[d] = pg[d] * inv - px[d] * c;
    } else {                                           // clamped denominator: y = x / 1e-12 is linear in x
        for (int d = lane; d < D; d += 64) q[d] = pg[d] * 1e12f;
    }
}

}  // namespace

extern "C" int ogmm_pow2_scale(const float* W, int64_t count, int top, float* scale_out, float* inv_out, int inv_len, void* stream);

namespace {
// Power-of-two scale of a weight for the binary16 split (ops.split_f16): 2^e with max|W| 2^e in [2^top, 2^(top+1)), e clamped to +-24; one workgroup scans
// the weight (<= a few MB: ~10 us), writes 2^e and fills a vector with 2^-e that the GEMM takes as its per-column scale (struct ogmm_gemm.scale).  No
// host round trip and no cache: the training step re-splits ~120 weights (and derived, permuted / transposed copies of them) every step.
__global__ __launch_bounds__(256) void pow2_scale_kernel(const float* __restrict__ W, int64_t count, int top, float* __restrict__ scale_out,
                                                         float* __restrict__ inv_out, int inv_len) {
    // grid-wide: every workgroup reduces its slice (a single workgroup scanning a 4 MB weight took 80-285 us: 5 ms per training step), the maxima
    // meet in scale_out[2] (bit pattern of a non-negative float: integer order = float order) and the last workgroup to arrive (ticket in
    // scale_out[3]) derives the scale and fills the inverse vector.  scale_out[0..3] must be zero on entry.
    __shared__ float red[4];
    __shared__ int s_last;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(W[i]));
    m = ogmm::wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int* bits = reinterpret_cast<int*>(scale_out + 2);
        int* ticket = reinterpret_cast<int*>(scale_out + 3);
        __hip_atomic_fetch_max(bits, __float_as_int(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // Only thread 0 reads the maximum, and it resets the scratch AFTER having read it: with every thread loading scale_out[2] and thread 0 storing 0 there,
    // a wave that issued its load after wave 0's store saw mx == 0 and filled its share of inv_out with 1.0.  The scale travels to the
    // other waves through LDS behind a barrier.
    __shared__ float s_inv;
    if (threadIdx.x == 0) {
        const float mx = __int_as_float(__hip_atomic_load(reinterpret_cast<int*>(scale_out + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        int e = 0;
        if (mx > 0.0f && mx < __builtin_inff()) e = min(24, max(-24, top - (int)floorf(log2f(mx))));
        scale_out[0] = exp2f((float)e);
        s_inv = exp2f((float)-e);
        // leave the reduction's scratch as it was found (zero): a caller may hand the same four floats to the next call (ogmm_split_weight's slot pool)
        __hip_atomic_store(reinterpret_cast<int*>(scale_out + 2), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<int*>(scale_out + 3), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const float inv = s_inv;
    for (int i = threadIdx.x; i < inv_len; i += 256) inv_out[i] = inv;
}

// The OGMM_PREC_F16X3_FRAG image of sc * W (or of sc * W^T) for a weight that changes every step, sc = *scale a device scalar (a power of two: the product
// is exact).  Image entry ((nb * (Kp / 16) + kb) * 64 + lane) holds B[nb * 32 + (lane & 31)][kb * 16 + (lane >> 5) * 8 + j], j = 0..7, with
// B[n][k] = W[n][src(k)] (TRANSPOSE: W[src(k)][n]); src maps the image's K axis -- piece one [0, k1) at 0, piece two [k1, K) at k1p, both padded with
// zeros to multiples of 64 (struct ogmm_gemm: K1 | K2) -- onto the weight's columns; rows n >= N are zero.
using f16x8w = __attribute__((ext_vector_type(8))) _Float16;
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, int64_t ld, int N, int K, int k1, int k1p, int Kp, int64_t total,
                                                           const float* __restrict__ scale, f16x8w* __restrict__ hi, f16x8w* __restrict__ lo) {
    const int64_t gI = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gI >= total) return;
    const float sc = *scale;
    const int lane = (int)(gI & 63);
    const int64_t blk = gI >> 6;
    const int kb = (int)(blk % (Kp / 16)), nb = (int)(blk / (Kp / 16));
    const int n = nb * 32 + (lane & 31), kk0 = kb * 16 + (lane >> 5) * 8;
    f16x8w h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = kk0 + j;
            const int k = kk < k1p ? (kk < k1 ? kk : -1) : (kk - k1p + k1 < K ? kk - k1p + k1 : -1);
            float v = 0.0f;
            if (k >= 0) v = TRANSPOSE ? W[(int64_t)k * ld + n] : W[(int64_t)n * ld + k];
            const float x = __builtin_amdgcn_fmed3f(v * sc, -65504.0f, 65504.0f);
            const _Float16 hh = (_Float16)x;
            h[j] = hh;
            l[j] = (_Float16)(x - (float)hh);
        }
    }
    hi[gI] = h;
    lo[gI] = l;
}
}  // namespace

// Power-of-two scale (as ogmm_pow2_scale, top = 10) and split fragment image of a per-step weight in two launches.  W [rows][ld] fp32; transpose = 0: the
// image of W (N = rows, K = cols in the two-piece layout k1 | cols - k1, each padded to 64: Kp = ldb_h), 1: the image of W^T (N = cols, K = rows, one
// piece).  n_pad (a multiple of 32) = rows of the image.  scratch4: four floats, ZERO on entry; on exit [0] = the scale, [2], [3] zero again.
extern "C" int ogmm_split_weight(const float* W, int64_t ld, int rows, int cols, int transpose, int k1, float* scratch4, float* inv_out, int inv_len, void* hi,
                                 void* lo, int64_t ldb_h, int n_pad, void* stream) {
    OGMM_REQUIRE(W && scratch4 && inv_out && hi && lo && rows > 0 && cols > 0 && ld >= cols && inv_len > 0 && n_pad % 32 == 0 && ldb_h % 64 == 0,
                 "ogmm_split_weight: null pointer, empty shape, or n_pad %% 32 / ldb_h %% 64 != 0");
    OGMM_REQUIRE(ogmm::aligned16(hi) && ogmm::aligned16(lo), "ogmm_split_weight: images must be 16-byte aligned");
    const int N = transpose ? cols : rows, K = transpose ? rows : cols;
    if (transpose || k1 <= 0 || k1 > K) k1 = K;
    const int k1p = (k1 + 63) / 64 * 64, k2p = (K - k1 + 63) / 64 * 64;
    OGMM_REQUIRE(n_pad >= N && ldb_h == k1p + k2p, "ogmm_split_weight: n_pad < N or ldb_h != padded K (%d + %d)", k1p, k2p);
    OGMM_REQUIRE(ld == cols, "ogmm_split_weight: the scale scan reads W as one contiguous block (ld == cols)");
    int rc = ogmm_pow2_scale(W, (int64_t)rows * cols, 10, scratch4, inv_out, inv_len, stream);
    if (rc) return rc;
    const int64_t total = (int64_t)(n_pad / 32) * (ldb_h / 16) * 64;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (transpose)
        hipLaunchKernelGGL(split_weight_kernel<true>, dim3(blocks), dim3(256), 0, ogmm::as_stream(stream), W, ld, N, K, k1, k1p, (int)ldb_h, total, scratch4,
                           reinterpret_cast<f16x8w*>(hi), reinterpret_cast<f16x8w*>(lo));
    else
        hipLaunchKernelGGL(split_weight_kernel<false>, dim3(blocks), dim3(256), 0, ogmm::as_stream(stream), W, ld, N, K, k1, k1p, (int)ldb_h, total, scratch4,
                           reinterpret_cast<f16x8w*>(hi), reinterpret_cast<f16x8w*>(lo));
    return ogmm::check_launch("ogmm_split_weight");
}

extern "C" int ogmm_pow2_scale(const float* W, int64_t count, int top, float* scale_out, float* inv_out, int inv_len, void* stream) {
    OGMM_REQUIRE(W && scale_out && inv_out && count > 0 && inv_len > 0 && top >= 0 && top <= 14, "ogmm_pow2_scale: bad arguments");
    const unsigned blocks = (unsigned)std::min<int64_t>(256, (count + 4095) / 4096);
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(blocks), dim3(256), 0, ogmm::as_stream(stream), W, count, top, scale_out, inv_out, inv_len);
    return ogmm::check_launch("ogmm_pow2_scale");
}

extern "C" int ogmm_edge_features(const float* xyz, const int32_t* idx, int C, int N, int k, float* out, void* stream) {
    OGMM_REQUIRE(xyz && idx && out && C > 0 && N > 0 && k > 0, "ogmm_edge_features: null pointer or empty input");
    const int64_t edges = (int64_t)C * N * k;
    hipLaunchKernelGGL(edge_features_kernel, dim3((unsigned)((edges + 255) / 256)), dim3(256), 0, as_stream(stream), xyz, idx, N, k, edges, out);
    return check_launch("ogmm_edge_features");
}

extern "C" int ogmm_pos_features(const float* xyz, const int32_t* idx, int C, int N, int k, const float* centroid, float* d2, float* alpha, void* stream) {
    OGMM_REQUIRE(xyz && idx && centroid && d2 && alpha && C > 0 && N > 0 && k > 0, "ogmm_pos_features: null pointer or empty input");
    const int64_t points = (int64_t)C * N;
    hipLaunchKernelGGL(pos_features_kernel, dim3((unsigned)((points + 255) / 256)), dim3(256), 0, as_stream(stream), xyz, idx, N, k, points, centroid, d2, alpha);
    return check_launch("ogmm_pos_features");
}

extern "C" int ogmm_l2norm_rows_bwd(const float* x, int64_t ldx, const float* g, int64_t ldg, int64_t rows, int D, float* dx, int64_t lddx, void* stream) {
    OGMM_REQUIRE(x && g && dx && rows > 0 && D > 0, "ogmm_l2norm_rows_bwd: null pointer or empty input");
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, ldx, g, ldg, rows, D, dx, lddx);
    return check_launch("ogmm_l2norm_rows_bwd");
}

// ---------------------------------------------------------------- out = s_0 + s_1 + ... + s_{n-1}  (n <= 8 row-major maps, own row pitches)
// The gradient of a feature map with several consumers (the residual, the Q projection, the MLP's first piece, the anchor gather, ...: up to
// seven for the cross-attention output of models/gmmreg.py:64-97).  autograd adds them pairwise as they arrive: (n - 1) x (2 reads + 1 write);
// here n reads + 1 write.  Summation order s_0 + s_1 + ... as autograd's.
namespace {
struct AddN { const float* s[8]; int64_t ld[8]; };
template <bool VEC>
__global__ __launch_bounds__(256) void add_n_kernel(AddN a, int n, int64_t rows, int cols, float* __restrict__ out, int64_t ldo) {
    const int per_row = VEC ? cols / 4 : cols;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) * (VEC ? 4 : 1);
        if (VEC) {
            float4 acc = *reinterpret_cast<const float4*>(a.s[0] + r * a.ld[0] + c);
#pragma unroll
            for (int k = 1; k < 8; ++k)
                if (k < n) {
                    const float4 v = *reinterpret_cast<const float4*>(a.s[k] + r * a.ld[k] + c);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            *reinterpret_cast<float4*>(out + r * ldo + c) = acc;
        } else {
            float acc = a.s[0][r * a.ld[0] + c];
#pragma unroll
            for (int k = 1; k < 8; ++k) if (k < n) acc += a.s[k][r * a.ld[k] + c];
            out[r * ldo + c] = acc;
        }
    }
}
}  // namespace

extern "C" int ogmm_add_n(int n, const float* const* srcs, const int64_t* lds, int64_t rows, int cols, float* out, int64_t ldo, void* stream) {
    OGMM_REQUIRE(srcs && lds && out && n >= 1 && n <= 8 && rows > 0 && cols > 0, "ogmm_add_n: 1 <= n <= 8 maps, non-empty");
    AddN a{};
    bool vec = cols % 4 == 0 && ldo % 4 == 0 && aligned16(out);
    for (int k = 0; k < n; ++k) {
        OGMM_REQUIRE(srcs[k], "ogmm_add_n: null map %d", k);
        a.s[k] = srcs[k]; a.ld[k] = lds[k];
        vec = vec && lds[k] % 4 == 0 && aligned16(srcs[k]);
    }
    const int64_t total = rows * (vec ? cols / 4 : cols);
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16);
    if (vec) hipLaunchKernelGGL(add_n_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), a, n, rows, cols, out, ldo);
    else hipLaunchKernelGGL(add_n_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), a, n, rows, cols, out, ldo);
    return check_launch("ogmm_add_n");
}

// ---------------------------------------------------------------- nearest squared distance, brute force (evaluation metrics)
// lib/metric.py:193-194 (`square_distance`, direct form sum (a - b)^2) followed by min over the other cloud (:221-236): the
// [B,Na,Nb] matrix of the reference is never formed.  The other cloud is staged through LDS in tiles; one thread per query.
namespace {
constexpr int MSD_TILE = 1024;
__global__ __launch_bounds__(256) void min_sqdist_kernel(const float* __restrict__ a, const float* __restrict__ b, int Na, int Nb,
                                                         float* __restrict__ out) {
    __shared__ float tile[MSD_TILE * 3];
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* __restrict__ pa = a + ((int64_t)c * Na + (i < Na ? i : 0)) * 3;
    const float x = pa[0], y = pa[1], z = pa[2];
    float best = __builtin_inff();
    for (int j0 = 0; j0 < Nb; j0 += MSD_TILE) {
        const int n = min(MSD_TILE, Nb - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < n * 3; t += 256) tile[t] = b[((int64_t)c * Nb + j0) * 3 + t];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float dx = x - tile[3 * j], dy = y - tile[3 * j + 1], dz = z - tile[3 * j + 2];
            best = fminf(best, ogmm::add_rn(ogmm::add_rn(ogmm::mul_rn(dx, dx), ogmm::mul_rn(dy, dy)), ogmm::mul_rn(dz, dz)));
        }
    }
    if (i < Na) out[(int64_t)c * Na + i] = best;
}
}  // namespace

extern "C" int ogmm_min_sqdist(const float* a, const float* b, int B, int Na, int Nb, float* out, void* stream) {
    OGMM_REQUIRE(a && b && out && B > 0 && Na > 0 && Nb > 0, "ogmm_min_sqdist: null pointer or empty input");
    hipLaunchKernelGGL(min_sqdist_kernel, dim3((Na + 255) / 256, B), dim3(256), 0, ogmm::as_stream(stream), a, b, Na, Nb, out);
    return ogmm::check_launch("ogmm_min_sqdist");
}
