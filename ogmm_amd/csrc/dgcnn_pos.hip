// K2+K3 (layer 1): EdgeConv gather + conv 6->64 + BN + ReLU + max over k   (lib/utils.py:56-64, models/dgcnn.py:137-139)
// K7 (front):     PositionEncoding hidden maps                               (models/attn.py:65-73)
//
// Both are VALU/HBM work (K = 6 resp. K = 1 contractions are not worth an MFMA tile): a thread owns one
// output channel, 64 consecutive threads write one 256-byte row, so the [edges][64] / [points][64]
// outputs stream out fully coalesced; neighbour coordinates are gathered through L2.
#include "ogmm_common.h"

namespace {

using namespace ogmm;

constexpr int PTS_PER_BLOCK = 32;   // edgeconv_first: 4 point slots x 8 rounds

__global__ __launch_bounds__(256) void edgeconv_first_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx,
                                                             int N, int k, int64_t total_pts,
                                                             const float* __restrict__ W, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ h1,
                                                             float* __restrict__ pool_out, int64_t ldp) {
    const int ch = threadIdx.x & 63, slot = threadIdx.x >> 6;
    float w[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = W[ch * 6 + i];
    const float s = scale[ch], t = shift[ch];
    for (int r = 0; r < PTS_PER_BLOCK / 4; ++r) {
        const int64_t p = (int64_t)blockIdx.x * PTS_PER_BLOCK + r * 4 + slot;     // global point row (cloud * N + i)
        if (p >= total_pts) break;
        const int64_t cloud_base = (p / N) * N;
        const float xi = xyz[3 * p], yi = xyz[3 * p + 1], zi = xyz[3 * p + 2];
        // the x_i half of cat(x_j - x_i, x_i) is the same for all k edges
        const float ctr = fmaf(w[5], zi, fmaf(w[4], yi, w[3] * xi));
        float best = 0.0f;   // ReLU output >= 0
        const int32_t* nb = idx + p * k;
        float* out = h1 + p * k * 64 + ch;
        for (int e = 0; e < k; ++e) {
            const int64_t j = cloud_base + nb[e];
            const float dx = xyz[3 * j] - xi, dy = xyz[3 * j + 1] - yi, dz = xyz[3 * j + 2] - zi;
            const float acc = fmaf(w[2], dz, fmaf(w[1], dy, w[0] * dx)) + ctr;
            const float v = fmaxf(fmaf(acc, s, t), 0.0f);
            out[(int64_t)e * 64] = v;
            best = fmaxf(best, v);
        }
        pool_out[p * ldp + ch] = best;
    }
}

__device__ __forceinline__ float leaky02(float v) { return v > 0.0f ? v : 0.2f * v; }

// one block = one tile of 64 points of one cloud; the cloud centroid is recomputed per block (12 KB from L2)
__global__ __launch_bounds__(256) void pos_hidden_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int idx_ld,
                                                         int k_pos, int N, const float* __restrict__ w_dis,
                                                         const float* __restrict__ s_dis, const float* __restrict__ t_dis,
                                                         const float* __restrict__ w_ang, const float* __restrict__ s_ang,
                                                         const float* __restrict__ t_ang, float* __restrict__ hid_dis,
                                                         float* __restrict__ hid_ang) {
    __shared__ double part[4][3];
    const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    double sx = 0, sy = 0, sz = 0;
    for (int j = tid; j < N; j += 256) { sx += cloud[3 * j]; sy += cloud[3 * j + 1]; sz += cloud[3 * j + 2]; }
    sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
    if (lane == 0) { part[wave][0] = sx; part[wave][1] = sy; part[wave][2] = sz; }
    __syncthreads();
    const float fn = (float)N;
    const float cx = (float)(part[0][0] + part[1][0] + part[2][0] + part[3][0]) / fn;
    const float cy = (float)(part[0][1] + part[1][1] + part[2][1] + part[3][1]) / fn;
    const float cz = (float)(part[0][2] + part[1][2] + part[2][2] + part[3][2]) / fn;

    const int ch = lane, slot = wave;
    const float wd = w_dis[ch], sd = s_dis[ch], td = t_dis[ch];
    const float wa = w_ang[ch], sa = s_ang[ch], ta = t_ang[ch];
    for (int r = 0; r < 16; ++r) {
        const int i = blockIdx.x * 64 + r * 4 + slot;
        if (i >= N) break;
        const float xi = cloud[3 * i], yi = cloud[3 * i + 1], zi = cloud[3 * i + 2];
        const float gx = xi - cx, gy = yi - cy, gz = zi - cz;
        const float d2 = (gx * gx + gy * gy) + gz * gz;
        const int64_t row = (int64_t)c * N + i;
        hid_dis[row * 64 + ch] = leaky02(fmaf(wd * d2, sd, td));
        const float gn = fmaxf(sqrtf(d2), 1e-12f);
        const float ux = gx / gn, uy = gy / gn, uz = gz / gn;
        float best = -__builtin_inff();
        const int32_t* nb = idx + row * idx_ld;
        for (int e = 0; e < k_pos; ++e) {
            const int j = nb[e];
            const float lx = cloud[3 * j] - xi, ly = cloud[3 * j + 1] - yi, lz = cloud[3 * j + 2] - zi;
            const float ln = fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-12f);
            const float alpha = ((lx / ln) * ux + (ly / ln) * uy) + (lz / ln) * uz;
            best = fmaxf(best, leaky02(fmaf(wa * alpha, sa, ta)));
        }
        hid_ang[row * 64 + ch] = best;
    }
}

}  // namespace

extern "C" int ogmm_edgeconv_first(const float* xyz, const int32_t* idx, int C, int N, int k, const float* W, const float* scale,
                                   const float* shift, float* h1, float* pool_out, int64_t ldp, void* stream) {
    OGMM_REQUIRE(xyz && idx && W && scale && shift && h1 && pool_out, "ogmm_edgeconv_first: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && k > 0 && ldp >= 64, "ogmm_edgeconv_first: bad sizes C=%d N=%d k=%d ldp=%lld", C, N, k, (long long)ldp);
    const int64_t total = (int64_t)C * N;
    const unsigned blocks = (unsigned)((total + PTS_PER_BLOCK - 1) / PTS_PER_BLOCK);
    hipLaunchKernelGGL(edgeconv_first_kernel, dim3(blocks), dim3(256), 0, ogmm::as_stream(stream), xyz, idx, N, k, total, W, scale,
                       shift, h1, pool_out, ldp);
    return ogmm::check_launch("ogmm_edgeconv_first");
}

extern "C" int ogmm_pos_hidden(const float* xyz, const int32_t* idx, int idx_ld, int k_pos, int C, int N, const float* w_dis,
                               const float* s_dis, const float* t_dis, const float* w_ang, const float* s_ang, const float* t_ang,
                               float* hid_dis, float* hid_ang, void* stream) {
    OGMM_REQUIRE(xyz && idx && w_dis && s_dis && t_dis && w_ang && s_ang && t_ang && hid_dis && hid_ang, "ogmm_pos_hidden: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && k_pos > 0 && k_pos <= idx_ld, "ogmm_pos_hidden: bad sizes C=%d N=%d k_pos=%d idx_ld=%d", C, N, k_pos, idx_ld);
    hipLaunchKernelGGL(pos_hidden_kernel, dim3((N + 63) / 64, C), dim3(256), 0, ogmm::as_stream(stream), xyz, idx, idx_ld, k_pos, N,
                       w_dis, s_dis, t_dis, w_ang, s_ang, t_ang, hid_dis, hid_ang);
    return ogmm::check_launch("ogmm_pos_hidden");
}
