// Operand preparation for the weight-gradient GEMM of the training step:  dW[n][k] = sum_r dY[r][n] * X[r][k]
// (the derivative of y = x W^T in every dense layer of models/dgcnn.py, models/attn.py, models/gmmreg.py).
//
// The GEMM engine contracts over the LAST axis of both operands (C = A B^T), so the contraction over the row axis r is
// fed as A = dY^T (fp32, [n][r]) and B = X^T in the engine's pre-split fragment-major binary16 image ([k/32][r/16][64][8]).
// Both kernels are single-pass HBM-bound relayouts.  The contraction is cut into S chunks of `chunk` rows (split-K over the
// batch dimension of ogmm_gemm_nt; r is zero-padded to S*chunk) and each chunk's operand is stored compactly, chunk-major,
// with a row pitch of chunk + 64 elements: rows of one long [n][r_pad] matrix would sit a power of two apart (512 KB at
// r_pad = 131072), which puts every row of a tile on the same HBM channel.
#include "ogmm_common.h"
#include <algorithm>

namespace {

using namespace ogmm;
using f16x8t = __attribute__((ext_vector_type(8))) _Float16;

// out[s][c][rr] = x[s*chunk + rr][c] (0 beyond `rows`), row pitch `pitch`.  64x64 tiles through LDS, float4 on both sides (a 64-row block lies
// inside one chunk: chunk % 64 == 0).  colsum != NULL: the block also adds its 64-row column sums into colsum[blockIdx.x & slot_mask][c] (fp64
// atomics) -- the bias gradient db = sum_r dY[r][:] of the same layer, which would otherwise be one more pass over dY.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, int64_t r_pad,
                                                            int64_t chunk, int64_t pitch, float* __restrict__ out, double* __restrict__ colsum,
                                                            int slot_mask) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    {
        const int q = tid & 15, c4 = q * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = p * 16 + (tid >> 4);
            const int64_t r = r0 + row;
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (r < rows) {
                if (VEC && c0 + c4 + 3 < cols) {
                    const float4 t = *reinterpret_cast<const float4*>(x + r * ldx + c0 + c4);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c0 + c4 + e < cols) v[e] = x[r * ldx + c0 + c4 + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[row][c4 + e] = v[e];
        }
    }
    __syncthreads();
    if (colsum && tid < 64 && c0 + tid < cols) {
        float s = 0.0f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) s += tile[i][tid];
        atomicAdd(colsum + (int64_t)(blockIdx.x & slot_mask) * cols + c0 + tid, (double)s);
    }
    const int64_t sb = r0 / chunk, rr0 = r0 - sb * chunk;
    if (r0 < r_pad) {
        const int q = tid & 15, r4 = q * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = p * 16 + (tid >> 4);
            if (c0 + c < cols) {
                float* __restrict__ o = out + (sb * cols + c0 + c) * pitch + rr0 + r4;
                if (VEC) {
                    *reinterpret_cast<float4*>(o) = make_float4(tile[r4][c], tile[r4 + 1][c], tile[r4 + 2][c], tile[r4 + 3][c]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = tile[r4 + e][c];
                }
            }
        }
    }
}

// Fragment-major split images of X^T, one per chunk s: entry (((s * n_pad/32 + nb) * (pitch/16) + kb) * 64 + lane) holds the 8
// binary16 values X[s*chunk + kb*16 + (lane>>5)*8 + j][nb*32 + (lane&31)], j = 0..7 (hi and lo planes); zero outside the
// matrix and in the pitch padding (kb*16 >= chunk).
__global__ __launch_bounds__(256) void pack_frag_t_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, int64_t chunk,
                                                          int64_t pitch, int S, int n_pad, f16x8t* __restrict__ hi, f16x8t* __restrict__ lo,
                                                          int* __restrict__ overflow, const float* __restrict__ a_scale,
                                                          const float* __restrict__ a_shift, int a_relu, int64_t group_rows) {
    // grid (k blocks of 16 rows, groups of four 32-column blocks, chunks): no index division per entry (the grid-stride form spent ~160
    // instructions per entry on 64-bit div / mod and ran at 3.5 TB/s); the four waves of a workgroup read 512 contiguous bytes of every row
    const int64_t kblocks = pitch / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t kb = blockIdx.x, sb = blockIdx.z;
    const int nb = blockIdx.y * 4 + wave;
    if (nb >= n_pad / 32) return;
    const int64_t gI = (((int64_t)sb * (n_pad / 32) + nb) * kblocks + kb) * 64 + lane;
    const int n = nb * 32 + (lane & 31);
    const int64_t k0 = sb * chunk + kb * 16 + (lane >> 5) * 8;
    bool clipped = false;
    f16x8t h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < cols && kb * 16 < chunk) {
        // a_scale: the map is the pre-normalisation output, X = relu(x * scale_g + shift_g) as the forward GEMM read it.  One division per
        // 8 rows: they lie in one row group or straddle one boundary.
        float sc0 = 1.0f, sh0 = 0.0f, sc1 = 1.0f, sh1 = 0.0f;
        int64_t first_of_next = rows;
        if (a_scale && k0 < rows) {
            const int64_t g0 = k0 / group_rows;
            first_of_next = (g0 + 1) * group_rows;
            sc0 = a_scale[g0 * cols + n]; sh0 = a_shift[g0 * cols + n];
            if (first_of_next < k0 + 8 && first_of_next < rows) { sc1 = a_scale[(g0 + 1) * cols + n]; sh1 = a_shift[(g0 + 1) * cols + n]; }
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = k0 + j < rows ? x[(k0 + j) * ldx + n] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = k0 + j;
            float t = v[j];
            if (a_scale && r < rows) {
                t = r < first_of_next ? fmaf(t, sc0, sh0) : fmaf(t, sc1, sh1);
                if (a_relu) t = fmaxf(t, 0.0f);
            }
            const float c = __builtin_amdgcn_fmed3f(t, -65504.0f, 65504.0f);
            clipped |= c != t && t == t;
            const _Float16 hh = (_Float16)c;
            h[j] = hh;
            l[j] = (_Float16)(c - (float)hh);
        }
    }
    hi[gI] = h;
    lo[gI] = l;
    if (clipped && overflow) atomicOr(overflow, 1);
}

}  // namespace

extern "C" int ogmm_transpose_pad(const float* x, int64_t ldx, int64_t rows, int cols, int64_t chunk, int64_t pitch, int S, float* out, double* colsum,
                                  int colsum_slots, void* stream) {
    OGMM_REQUIRE(x && out && rows > 0 && cols > 0 && chunk > 0 && chunk % 64 == 0 && S > 0 && (int64_t)S * chunk >= rows && pitch >= chunk,
                 "ogmm_transpose_pad: bad shape (chunk must be a multiple of 64)");
    OGMM_REQUIRE(!colsum || (colsum_slots >= 1 && (colsum_slots & (colsum_slots - 1)) == 0), "ogmm_transpose_pad: colsum_slots must be a power of two");
    const int64_t r_pad = (int64_t)S * chunk;
    dim3 grid((unsigned)((r_pad + 63) / 64), (cols + 63) / 64);
    if (colsum) (void)hipMemsetAsync(colsum, 0, sizeof(double) * (size_t)colsum_slots * cols, as_stream(stream));
    const bool vec = ldx % 4 == 0 && pitch % 4 == 0 && aligned16(x) && aligned16(out);
    if (vec) hipLaunchKernelGGL(transpose_pad_kernel<true>, grid, dim3(256), 0, as_stream(stream), x, ldx, rows, cols, r_pad, chunk, pitch, out, colsum, colsum_slots - 1);
    else hipLaunchKernelGGL(transpose_pad_kernel<false>, grid, dim3(256), 0, as_stream(stream), x, ldx, rows, cols, r_pad, chunk, pitch, out, colsum, colsum_slots - 1);
    return check_launch("ogmm_transpose_pad");
}

extern "C" int ogmm_pack_frag_t(const float* x, int64_t ldx, int64_t rows, int cols, int64_t chunk, int64_t pitch, int S, int n_pad, void* hi, void* lo,
                                int* overflow, const float* a_scale, const float* a_shift, int a_relu, int64_t group_rows, void* stream) {
    OGMM_REQUIRE(!a_scale || (a_shift && group_rows > 0), "ogmm_pack_frag_t: a_scale needs a_shift and group_rows > 0");
    OGMM_REQUIRE(x && hi && lo && rows > 0 && cols > 0 && S > 0 && (int64_t)S * chunk >= rows && chunk % 16 == 0 && pitch % 16 == 0 && pitch >= chunk &&
                 n_pad >= cols && n_pad % 32 == 0 && aligned16(hi) && aligned16(lo),
                 "ogmm_pack_frag_t: chunk, pitch %% 16 == 0, n_pad %% 32 == 0, 16-byte aligned images required");
    OGMM_REQUIRE(pitch / 16 < ((int64_t)1 << 31) && S <= 65535 && (n_pad / 32 + 3) / 4 <= 65535, "ogmm_pack_frag_t: grid too large");
    const dim3 grid((unsigned)(pitch / 16), (unsigned)((n_pad / 32 + 3) / 4), (unsigned)S);
    hipLaunchKernelGGL(pack_frag_t_kernel, grid, dim3(256), 0, as_stream(stream), x, ldx, rows, cols, chunk, pitch, S, n_pad,
                       reinterpret_cast<f16x8t*>(hi), reinterpret_cast<f16x8t*>(lo), overflow, a_scale, a_shift, a_relu, group_rows);
    return check_launch("ogmm_pack_frag_t");
}
