// K2+K3 fused: the whole DGCNN EdgeConv chain of models/dgcnn.py:135-150 for a tile of points in one workgroup:
//   gather neighbours -> cat(x_j - x_i, x_i) -> conv1(6->64)+BN+ReLU -> conv2(64->64) -> conv3(64->128) -> conv4(128->256),
//   each +BN+ReLU, with the max over the k edges of every point taken after each layer (x1|x2|x3|x4 -> xcat[point][512]).
// The per-edge tensors [C*N*k][64|64|128] (2.7 GB at B=64) never leave the chip: layer l's activated output is written
// -- already split into binary16 hi/lo planes in A-operand order -- into LDS and consumed by layer l+1.
//
// Tile = P = floor(160 / k) points = P*k <= 160 edge rows (5 MFMA row blocks); 8 waves (2 per SIMD).
//   layer 1: VALU (K = 6), thread = (channel, edge slot)
//   layers 2-4: fp16x3 split MFMA (v_mfma_f32_32x32x16_f16, see gemm_f16x3.hip); the (row block, 32-column block) grid of a
//   layer (5 x 2, 5 x 4, 5 x 8) is dealt over the 8 waves; a wave's weight fragments (one column block of a whole layer)
//   are fetched from the fragment-major images (L2 resident, 180 KB) a full layer ahead of their use.
// LDS: region A = h1 planes, later h3 planes (87 KB); region B = h2 planes (46 KB); pool scratch 8 x 256 ints.
#include "ogmm_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int ROWS = 160;
constexpr int LD64 = 64 + 8, LD128 = 128 + 8;       // plane row lengths in halfs (conflict-free ds_read_b128)

__device__ __forceinline__ void split_h(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Weight fragments of one 32-column block for a whole layer (KS k-steps): hi/lo, fetched well before they are needed.
template <int KS>
__device__ __forceinline__ void load_weights(f16x8 (&wb)[KS][2], const void* hi, const void* lo, int nb, int lane) {
    const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(hi);
    const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(lo);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int64_t off = ((int64_t)nb * KS + s) * 64 + lane;
        wb[s][0] = BH[off];
        wb[s][1] = BL[off];
    }
}

// One MFMA layer for this wave: NB row blocks of 32 starting at block rb0 x the 32-column block nb (weights already in registers).
//   in  : A planes (hi at Ain, lo at Ain + ROWS*LDA), K = 16*KS input channels
//   out : relu(acc * scale + shift) -> pooled max per point into pool_s[point][column] (int atomicMax), and, if Aout != null,
//         split into the next layer's A planes.
template <int KS, int NB>
__device__ __forceinline__ void mfma_layer(const _Float16* Ain, int LDA, const f16x8 (&wb)[KS][2], int nb, int rb0, float inv_scale,
                                           const float* __restrict__ scale, const float* __restrict__ shift, _Float16* Aout, int LDO,
                                           int* pool_s, int pool_ld, unsigned inv_k16, int rows_valid, int lane) {
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int APL = ROWS * LDA;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 ah[NB], al[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int off = ((rb0 + i) * 32 + lr) * LDA + s * 16 + lh * 8;
            ah[i] = *reinterpret_cast<const f16x8*>(&Ain[off]);
            al[i] = *reinterpret_cast<const f16x8*>(&Ain[APL + off]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wb[s][0], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wb[s][1], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wb[s][0], acc[i], 0, 0, 0);
    }
    const int OPL = ROWS * LDO;
    const int col = nb * 32 + lr;
    const float sc = scale[col] * inv_scale, sh = shift[col];
    int cur_group = -1;
    float cur_max = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (rb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = fmaxf(fmaf(acc[i][r], sc, sh), 0.0f);
            if (Aout) {
                _Float16 a, b;
                split_h(v, a, b);
                Aout[row * LDO + col] = a;
                Aout[OPL + row * LDO + col] = b;
            }
            if (row < rows_valid) {
                const int grp = (int)(((unsigned)row * inv_k16) >> 16);    // row / k for row < 160, 7 <= k <= 32 (no division, no LDS)
                if (grp != cur_group) {
                    if (cur_group >= 0) atomicMax(&pool_s[cur_group * pool_ld + col], __float_as_int(cur_max));
                    cur_group = grp;
                    cur_max = v;
                } else {
                    cur_max = fmaxf(cur_max, v);
                }
            }
        }
    if (cur_group >= 0) atomicMax(&pool_s[cur_group * pool_ld + col], __float_as_int(cur_max));
}

struct EdgeW {
    const float* W1; const float* s1; const float* t1;                                   // [64][6], [64], [64]
    const void* h2; const void* l2; const float* s2; const float* t2; float inv2;        // fragment images + folded BN
    const void* h3; const void* l3; const float* s3; const float* t3; float inv3;
    const void* h4; const void* l4; const float* s4; const float* t4; float inv4;
};

__global__ __launch_bounds__(512) void edgeconv_fused_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int N, int k,
                                                             int64_t total_pts, int P, const EdgeW w, float* __restrict__ xcat, int64_t ldx) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16* regA = lds;                               // h1 planes [2][160][72]  -> later h3 planes [2][160][136]
    _Float16* regB = lds + 2 * ROWS * LD128;            // h2 planes [2][160][72]
    int* pool_s = reinterpret_cast<int*>(regB + 2 * ROWS * LD64);      // [P <= 40][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * P;
    const int pts = (int)min((int64_t)P, total_pts - p0);
    const int rows_valid = pts * k;
    const unsigned inv_k16 = (65536u + (unsigned)k - 1u) / (unsigned)k;     // (row * inv_k16) >> 16 == row / k on this range

    // weight fragments travel from L2 while the gather and layer 1 run (they do not depend on the activations)
    f16x8 wb2[4][2], wb3[4][2], wb4[8][2];
    load_weights<4>(wb2, w.h2, w.l2, wave & 1, lane);
    load_weights<4>(wb3, w.h3, w.l3, wave & 3, lane);

    // ---- gather: one thread per edge row fetches (x_j - x_i, x_i) once (two dependent global loads per edge, all 160 in flight
    // together) into region B, which is free until layer 2 writes h2
    float* ef = reinterpret_cast<float*>(regB);          // [160][6]
    for (int i = tid; i < P * 256; i += 512) pool_s[i] = 0;
    if (tid < ROWS) {
        const int e = tid;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
        if (e < rows_valid) {
            const int64_t p = p0 + e / k;
            const int64_t j = (p / N) * N + idx[p * k + e % k];
            f3 = xyz[3 * p]; f4 = xyz[3 * p + 1]; f5 = xyz[3 * p + 2];
            f0 = xyz[3 * j] - f3; f1 = xyz[3 * j + 1] - f4; f2 = xyz[3 * j + 2] - f5;
        }
        ef[e * 6 + 0] = f0; ef[e * 6 + 1] = f1; ef[e * 6 + 2] = f2; ef[e * 6 + 3] = f3; ef[e * 6 + 4] = f4; ef[e * 6 + 5] = f5;
    }
    __syncthreads();
    // ---- layer 1 (VALU): thread = (channel, slot); the slot walks edge rows slot, slot+4, ...
    {
        const int ch = lane, slot = wave;
        float wv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = w.W1[ch * 6 + i];
        const float s = w.s1[ch], t = w.t1[ch];
        for (int e = slot; e < ROWS; e += 8) {
            float v = 0.0f;
            if (e < rows_valid) {
                const float ctr = fmaf(wv[5], ef[e * 6 + 5], fmaf(wv[4], ef[e * 6 + 4], wv[3] * ef[e * 6 + 3]));
                const float acc = fmaf(wv[2], ef[e * 6 + 2], fmaf(wv[1], ef[e * 6 + 1], wv[0] * ef[e * 6 + 0])) + ctr;
                v = fmaxf(fmaf(acc, s, t), 0.0f);
                atomicMax(&pool_s[(int)(((unsigned)e * inv_k16) >> 16) * 256 + ch], __float_as_int(v));       // x1 = max over the point's k edges
            }
            _Float16 a, b;
            split_h(v, a, b);
            regA[e * LD64 + ch] = a;
            regA[ROWS * LD64 + e * LD64 + ch] = b;
        }
    }
    __syncthreads();
    for (int i = tid; i < pts * 64; i += 512) {
        const int p = i >> 6, ch = i & 63;
        xcat[(p0 + p) * ldx + ch] = __int_as_float(pool_s[p * 256 + ch]);
    }
    __syncthreads();
    for (int i = tid; i < P * 256; i += 512) pool_s[i] = 0;
    __syncthreads();

    // ---- layer 2: 64 -> 64, waves 0 and 1 own 32 columns each
    load_weights<8>(wb4, w.h4, w.l4, wave, lane);       // needed two layers from now
    // 2 column blocks x 5 row blocks over 8 waves: waves 0-1 take row blocks {0,1}, waves 2-7 one of {2,3,4}
    if (wave < 2) mfma_layer<4, 2>(regA, LD64, wb2, wave & 1, 0, w.inv2, w.s2, w.t2, regB, LD64, pool_s, 256, inv_k16, rows_valid, lane);
    else mfma_layer<4, 1>(regA, LD64, wb2, wave & 1, 1 + (wave >> 1), w.inv2, w.s2, w.t2, regB, LD64, pool_s, 256, inv_k16, rows_valid, lane);
    __syncthreads();
    for (int i = tid; i < pts * 64; i += 512) {
        const int p = i >> 6, ch = i & 63;
        xcat[(p0 + p) * ldx + 64 + ch] = __int_as_float(pool_s[p * 256 + ch]);
    }
    __syncthreads();
    for (int i = tid; i < P * 256; i += 512) pool_s[i] = 0;
    __syncthreads();

    // ---- layer 3: 64 -> 128, waves 0-3 own 32 columns each; output planes overwrite region A (h1 is dead)
    // 4 column blocks x 5 row blocks over 8 waves: waves 0-3 take row blocks {0,1,2}, waves 4-7 {3,4}
    if (wave < 4) mfma_layer<4, 3>(regB, LD64, wb3, wave & 3, 0, w.inv3, w.s3, w.t3, regA, LD128, pool_s, 256, inv_k16, rows_valid, lane);
    else mfma_layer<4, 2>(regB, LD64, wb3, wave & 3, 3, w.inv3, w.s3, w.t3, regA, LD128, pool_s, 256, inv_k16, rows_valid, lane);
    __syncthreads();
    for (int i = tid; i < pts * 128; i += 512) {
        const int p = i >> 7, ch = i & 127;
        xcat[(p0 + p) * ldx + 128 + ch] = __int_as_float(pool_s[p * 256 + ch]);
    }
    __syncthreads();
    for (int i = tid; i < P * 256; i += 512) pool_s[i] = 0;
    __syncthreads();

    // ---- layer 4: 128 -> 256, every wave 32 columns; only the pooled output is needed
    mfma_layer<8, 5>(regA, LD128, wb4, wave, 0, w.inv4, w.s4, w.t4, nullptr, 0, pool_s, 256, inv_k16, rows_valid, lane);
    __syncthreads();
    for (int i = tid; i < pts * 256; i += 512) {
        const int p = i >> 8, ch = i & 255;
        xcat[(p0 + p) * ldx + 256 + ch] = __int_as_float(pool_s[p * 256 + ch]);
    }
}

}  // namespace

extern "C" int ogmm_edgeconv_fused(const float* xyz, const int32_t* idx, int C, int N, int k, const float* W1, const float* s1, const float* t1,
                                   const void* h2, const void* l2, const float* s2, const float* t2, float inv2, const void* h3,
                                   const void* l3, const float* s3, const float* t3, float inv3, const void* h4, const void* l4,
                                   const float* s4, const float* t4, float inv4, float* xcat, int64_t ldx, void* stream) {
    OGMM_REQUIRE(xyz && idx && W1 && s1 && t1 && h2 && l2 && s2 && t2 && h3 && l3 && s3 && t3 && h4 && l4 && s4 && t4 && xcat,
                 "ogmm_edgeconv_fused: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && k >= 4 && k <= 32 && ldx >= 512, "ogmm_edgeconv_fused: bad sizes C=%d N=%d k=%d ldx=%lld", C, N, k, (long long)ldx);
    const int P = ROWS / k;
    const int64_t total = (int64_t)C * N;
    const size_t lds = (size_t)(2 * ROWS * LD128 + 2 * ROWS * LD64) * sizeof(_Float16) + (size_t)P * 256 * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    OGMM_REQUIRE(lds <= 160 * 1024, "ogmm_edgeconv_fused: LDS budget exceeded");
    EdgeW w{W1, s1, t1, h2, l2, s2, t2, inv2, h3, l3, s3, t3, inv3, h4, l4, s4, t4, inv4};
    const unsigned blocks = (unsigned)((total + P - 1) / P);
    hipLaunchKernelGGL(edgeconv_fused_kernel, dim3(blocks), dim3(512), lds, ogmm::as_stream(stream), xyz, idx, N, k, total, P, w, xcat, ldx);
    return ogmm::check_launch("ogmm_edgeconv_fused");
}
