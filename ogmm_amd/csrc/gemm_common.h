// Shared pieces of the GEMM engines (exact-fp32 and fp16x3-split): activation, tile->workgroup map, fused epilogue.
#pragma once
#include "ogmm_common.h"

namespace ogmm_gemm_detail {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case OGMM_ACT_RELU: return fmaxf(v, 0.0f);
        case OGMM_ACT_LEAKY02: return v > 0.0f ? v : 0.2f * v;
        case OGMM_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// Fused epilogue for a workgroup tile whose waves hold MT x NT 32x32 accumulators in the MFMA C/D layout
// (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)):
//   v = act(acc * alpha * scale + shift) (+ Res); optional store; optional max over groups of pool_k rows
// (EdgeConv): per-lane run-combine -> LDS atomicMax on the int pattern (v >= 0 after ReLU) -> one coalesced store.
// `smem` must hold groups * BN ints and be free for reuse (the caller's K loop is finished).
template <int MT, int NT, int WM, int WN, bool POOL>
__device__ __forceinline__ void gemm_epilogue(const ogmm_gemm& g, f32x16 (&acc)[MT][NT], float* smem, int m0, int n0, int m_end,
                                              int zo, int zi, float alpha) {
    constexpr int BN = NT * 32 * WN, T = WM * WN * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    float* __restrict__ Cm = g.C ? g.C + zo * g.sC_o + zi * g.sC_i : nullptr;
    const float* __restrict__ Rm = g.Res ? g.Res + zo * g.sR_o + zi * g.sR_i : nullptr;
    const bool store_c = Cm != nullptr && (!POOL || g.store_c);
    int* pool_s = reinterpret_cast<int*>(smem);
    const int groups = POOL ? (m_end - m0) / g.pool_k : 0;
    if (POOL) {
        __syncthreads();
        for (int i = tid; i < groups * BN; i += T) pool_s[i] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int cl = (wn * NT + j) * 32 + lr;      // column inside the tile
        const int col = n0 + cl;
        const bool col_ok = col < g.N;
        float cs = 1.0f, ct = 0.0f;
        if (!g.row_affine && col_ok) {
            if (g.scale) cs = g.scale[col];
            if (g.shift) ct = g.shift[col];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int cur_group = -1;
            float cur_max = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // row inside the tile
                const int row = m0 + rl;
                if (row < m_end && col_ok) {
                    float s = cs, sh = ct;
                    if (g.row_affine) {
                        s = g.scale ? g.scale[row] : 1.0f;
                        sh = g.shift ? g.shift[row] : 0.0f;
                    }
                    float v = apply_act(fmaf(acc[i][j][r] * alpha, s, sh), g.act);
                    if (Rm) v += Rm[(int64_t)row * g.ldr + col];
                    if (store_c) Cm[(int64_t)row * g.ldc + col] = v;
                    if (POOL) {
                        const int grp = rl / g.pool_k;
                        if (grp != cur_group) {
                            if (cur_group >= 0) atomicMax(&pool_s[cur_group * BN + cl], __float_as_int(cur_max));
                            cur_group = grp;
                            cur_max = v;
                        } else {
                            cur_max = fmaxf(cur_max, v);
                        }
                    }
                }
            }
            if (POOL && cur_group >= 0) atomicMax(&pool_s[cur_group * BN + cl], __float_as_int(cur_max));
        }
    }
    if (POOL) {
        __syncthreads();
        float* __restrict__ Pm = g.pool_out + zo * 0;   // pooled output is not batched
        const int64_t p0 = (int64_t)(m0 / g.pool_k);
        for (int i = tid; i < groups * BN; i += T) {
            const int p = i / BN, c = i % BN;
            if (n0 + c < g.N) Pm[(p0 + p) * g.ldp + n0 + c] = __int_as_float(pool_s[i]);
        }
    }
}

}  // namespace ogmm_gemm_detail
