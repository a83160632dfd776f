// K9 fused: anchor attention  softmax(Q K^T / sqrt(dh)) V  for one (cloud, head, 128-query tile) per workgroup.
// Replaces models/attn.py:78-82 (two einsums + softmax) without ever writing the [C,H,N,M] score tensor.
//
// Same arithmetic as the weight-GEMM engine: every fp32 operand (Q, K, V, and the probabilities P) is split into two
// binary16 terms and each product block is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation
// (fp32-class accuracy, see gemm_f16x3.hip); the softmax itself runs in fp32 on the accumulator registers.
//
// Workgroup = 4 waves; wave w owns 32 query rows.
//   1. K_h [M keys][dh] and V_h [M keys][dh] (fp32, head-major channel slabs) are split and staged into LDS as
//      binary16 hi/lo planes: K as [key][dh] (B operand of S = Q K^T: n = key, k = d), V TRANSPOSED as [d][key]
//      (B operand of O = P V: n = d, k = key).
//   2. Q rows are loaded straight into registers in A-fragment order (lane: row l&31, 8 consecutive d's per k-step),
//      split there; S = Q K^T accumulates in MT x (M/32) 32x32 tiles.
//   3. softmax over the M keys of each row: the row lives in one 32-lane half across the M/32 tiles -> per-register
//      max / sum with 5 xor-shuffles each.
//   4. P is written (split) to a per-wave LDS patch in A-operand order -- the patch re-uses the K planes, hence one
//      barrier -- and O = P V accumulates in dh/32 tiles, stored head-major.
#include "ogmm_common.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int DH = 128;                 // head dimension (emb_dims 512 / 4 heads)
constexpr int QT = 128;                 // queries per workgroup

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

template <int MK>      // MK = M / 32 key tiles (1, 2 or 4)
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                        int64_t ldk, const float* __restrict__ v, int64_t ldv, int N, int H,
                                                        float scale, float* __restrict__ out, int64_t ldo) {
    constexpr int M = MK * 32;
    constexpr int LDK = DH + 8;          // halfs per K-plane row   ([key][d])
    constexpr int LDV = M + 8;           // halfs per V^T-plane row ([d][key])
    constexpr int LDP = M + 8;           // halfs per P-patch row   ([query][key])
    constexpr int KPLANE = M * LDK, VPLANE = DH * LDV, PPLANE = 32 * LDP;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    constexpr int OPATCH = 32 * (DH + 4) * 2;                                        // fp32 [32][DH+4] output patch, in halfs
    constexpr int WSTRIDE = (2 * PPLANE > OPATCH) ? 2 * PPLANE : OPATCH;             // per-wave patch (P planes, later O)
    constexpr int KREGION = (2 * KPLANE > 4 * WSTRIDE) ? 2 * KPLANE : 4 * WSTRIDE;   // K planes, later the 4 wave patches
    _Float16* Kh = lds;                  // [2 planes][M][LDK]   (later: 4 waves x [2 planes][32][LDP])
    _Float16* Vt = lds + KREGION;        // [2 planes][DH][LDV]

    const int h = blockIdx.y, c = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const float* __restrict__ kc = k + ((int64_t)c * M) * ldk + h * DH;
    const float* __restrict__ vc = v + ((int64_t)c * M) * ldv + h * DH;

    // ---- 1. stage K (split) and V (split + transposed).  K: coalesced rows in, rows out.  V: consecutive lanes take
    // consecutive KEYS of one d-quad, so the transposed 2-byte stores of a wave land on consecutive halfs of one plane row
    // (the other way round -- consecutive d's per lane -- is a 16-way bank conflict on every store).
    // (loads are issued in batches of 8 before anything consumes them: with one wave per SIMD a load -> use -> load chain
    // would expose the full memory latency on every iteration)
    constexpr int PIECES = M * (DH / 4) / 256;        // float4 per thread and operand: 4, 8 or 16
    constexpr int BATCH = PIECES < 8 ? PIECES : 8;
#pragma unroll
    for (int b0 = 0; b0 < PIECES; b0 += BATCH) {
        f32x4 kv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int f = tid + (b0 + u) * 256;
            const int key = f / (DH / 4), d4 = (f % (DH / 4)) * 4;
            kv[u] = *reinterpret_cast<const f32x4*>(kc + (int64_t)key * ldk + d4);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int f = tid + (b0 + u) * 256;
            const int key = f / (DH / 4), d4 = (f % (DH / 4)) * 4;
            f16x4 khi, klo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, b;
                split1(kv[u][e], a, b);
                khi[e] = a; klo[e] = b;
            }
            *reinterpret_cast<f16x4*>(&Kh[key * LDK + d4]) = khi;
            *reinterpret_cast<f16x4*>(&Kh[KPLANE + key * LDK + d4]) = klo;
        }
    }
#pragma unroll
    for (int b0 = 0; b0 < PIECES; b0 += BATCH) {
        f32x4 vv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int f = tid + (b0 + u) * 256;
            const int key = f % M, d4 = (f / M) * 4;
            vv[u] = *reinterpret_cast<const f32x4*>(vc + (int64_t)key * ldv + d4);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int f = tid + (b0 + u) * 256;
            const int key = f % M, d4 = (f / M) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, b;
                split1(vv[u][e], a, b);
                Vt[(d4 + e) * LDV + key] = a;
                Vt[VPLANE + (d4 + e) * LDV + key] = b;
            }
        }
    }
    const int tile = blockIdx.x;
    // ---- 2. Q fragments into registers (row = query, 8 consecutive d per k-step), overlapping the staging above
    const int q_row = tile * QT + wave * 32 + lr;
    const bool row_ok = q_row < N;
    const float* __restrict__ qp = q + ((int64_t)c * N + min(q_row, N - 1)) * ldq + h * DH;
    f16x8 qh[DH / 16], ql[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(qp + s * 16 + lh * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(qp + s * 16 + lh * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 x, y;
            split1(a[e], x, y); qh[s][e] = x; ql[s][e] = y;
            split1(b[e], x, y); qh[s][4 + e] = x; ql[s][4 + e] = y;
        }
    }
    __syncthreads();

    // ---- S = Q K^T
    f32x16 sacc[MK];
#pragma unroll
    for (int j = 0; j < MK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
        f16x8 bh[MK], bl[MK];
#pragma unroll
        for (int j = 0; j < MK; ++j) {
            const int off = (j * 32 + lr) * LDK + s * 16 + lh * 8;
            bh[j] = *reinterpret_cast<const f16x8*>(&Kh[off]);
            bl[j] = *reinterpret_cast<const f16x8*>(&Kh[KPLANE + off]);
        }
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[s], bh[j], sacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[s], bl[j], sacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[s], bh[j], sacc[j], 0, 0, 0);
    }

    // ---- 3. softmax over keys: register r of every tile holds row (r&3)+8(r>>2)+4*lh, column = lane&31 (+32 j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < MK; ++j) { sacc[j][r] *= scale; m = fmaxf(m, sacc[j][r]); }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < MK; ++j) { sacc[j][r] = expf(sacc[j][r] - m); sum += sacc[j][r]; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j][r] = sacc[j][r] / sum;
    }

    // ---- 4. P -> per-wave patch (A-operand order) in the K planes; all waves must be done reading K first
    __syncthreads();
    _Float16* Ph = lds + wave * WSTRIDE;
#pragma unroll
    for (int j = 0; j < MK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            _Float16 a, b;
            split1(sacc[j][r], a, b);
            const int off = ((r & 3) + 8 * (r >> 2) + 4 * lh) * LDP + j * 32 + lr;
            Ph[off] = a;
            Ph[PPLANE + off] = b;
        }
    // (the patch is private to the wave: LDS operations of one wave complete in order)

    // ---- O = P V
    f32x16 oacc[DH / 32];
#pragma unroll
    for (int j = 0; j < DH / 32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < M / 16; ++s) {
        const int aoff = lr * LDP + s * 16 + lh * 8;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&Ph[aoff]);
        const f16x8 al = *reinterpret_cast<const f16x8*>(&Ph[PPLANE + aoff]);
        f16x8 bh[DH / 32], bl[DH / 32];
#pragma unroll
        for (int j = 0; j < DH / 32; ++j) {
            const int off = (j * 32 + lr) * LDV + s * 16 + lh * 8;
            bh[j] = *reinterpret_cast<const f16x8*>(&Vt[off]);
            bl[j] = *reinterpret_cast<const f16x8*>(&Vt[VPLANE + off]);
        }
#pragma unroll
        for (int j = 0; j < DH / 32; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], oacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DH / 32; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], oacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DH / 32; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], oacc[j], 0, 0, 0);
    }

    // ---- store head-major: out[c*N + query][h*dh + d].  The wave's 32 x 128 tile is transposed through its (now free) P
    // patch so that 16 consecutive lanes write one 512-byte row segment with dwordx4 stores (4x fewer store instructions
    // than one dword per lane in the MFMA layout; the output phase is store-issue bound).
    {
        float* patch = reinterpret_cast<float*>(Ph);              // [32][DH + 4] floats = 16.9 KB <= 2 * PPLANE halfs
#pragma unroll
        for (int j = 0; j < DH / 32; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * (DH + 4) + j * 32 + lr] = oacc[j][r];
#pragma unroll
        for (int qi = 0; qi < 32 * (DH / 4) / 64; ++qi) {
            const int idx = qi * 64 + lane;
            const int rl = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
            const int row = tile * QT + wave * 32 + rl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&patch[rl * (DH + 4) + c4]);
            if (row < N) *reinterpret_cast<f32x4*>(out + ((int64_t)c * N + row) * ldo + h * DH + c4) = v;
        }
    }
    (void)row_ok;
}

// ------------------------------------------------------------------------------------------------------------------
// Second structure: K and V of every (cloud, head) are split and laid out ONCE per call, by attention_pack_kernel, as
// fragment-major images (the B-operand order of v_mfma_f32_32x32x16_f16, like the weight images of gemm_f16x3_v2.hip):
//   Kimg[c][h][plane][key/32][d/16][lane][8]    lane = ((d % 16) / 8) * 32 + key % 32      (S = Q K^T:  n = key, k = d)
//   Vimg[c][h][plane][d/32][key/16][lane][8]    lane = ((key % 16) / 8) * 32 + d % 32      (O = P V:    n = d,   k = key)
// The attention workgroups then read their B fragments as coalesced 1 KiB loads (L2/L1 resident, shared by the 8 query
// tiles of a head) instead of re-splitting and re-transposing 128 KB of K/V per 128 queries; LDS only holds the per-wave
// P patch (70 KB per workgroup -> two workgroups per CU).
// ------------------------------------------------------------------------------------------------------------------
// KEYPERM: the 16 keys of a V k-step are taken in the order in which a 32x32 ACCUMULATOR holds them along its rows (lane half lh,
// element e -> key 4*lh + (e & 3) + 8*(e >> 2)), for the transposed kernel below whose probabilities never leave the registers.
template <int MK, bool KEYPERM>
__global__ __launch_bounds__(256) void attention_pack_kernel(const float* __restrict__ k, int64_t ldk, const float* __restrict__ v,
                                                             int64_t ldv, int H, f16x8* __restrict__ kimg, f16x8* __restrict__ vimg) {
    constexpr int M = MK * 32;
    constexpr int GROUPS = M * DH / 8;                 // 8-element groups per plane and operand
    const int h = blockIdx.y, c = blockIdx.z;
    const float* __restrict__ kc = k + ((int64_t)c * M) * ldk + h * DH;
    const float* __restrict__ vc = v + ((int64_t)c * M) * ldv + h * DH;
    f16x8* __restrict__ ko = kimg + ((int64_t)c * H + h) * 2 * GROUPS;
    f16x8* __restrict__ vo = vimg + ((int64_t)c * H + h) * 2 * GROUPS;
    for (int gI = blockIdx.x * 256 + threadIdx.x; gI < GROUPS; gI += gridDim.x * 256) {
        const int lane = gI & 63;
        {   // K image: [key/32][d/16][lane]
            const int kb = (gI >> 6) % (DH / 16), nb = (gI >> 6) / (DH / 16);
            const int key = nb * 32 + (lane & 31), d0 = kb * 16 + (lane >> 5) * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(kc + (int64_t)key * ldk + d0);
            const f32x4 b = *reinterpret_cast<const f32x4*>(kc + (int64_t)key * ldk + d0 + 4);
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 x, y;
                split1(a[e], x, y); hi[e] = x; lo[e] = y;
                split1(b[e], x, y); hi[4 + e] = x; lo[4 + e] = y;
            }
            ko[gI] = hi;
            ko[GROUPS + gI] = lo;
        }
        {   // V image: [d/32][key/16][lane]
            const int kb = (gI >> 6) % (M / 16), nb = (gI >> 6) / (M / 16);
            const int d = nb * 32 + (lane & 31), key0 = kb * 16 + (lane >> 5) * 8;
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                const int key = KEYPERM ? kb * 16 + (lane >> 5) * 4 + (e & 3) + 8 * (e >> 2) : key0 + e;
                split1(vc[(int64_t)key * ldv + d], x, y);
                hi[e] = x; lo[e] = y;
            }
            vo[gI] = hi;
            vo[GROUPS + gI] = lo;
        }
    }
}

template <int MK>
__global__ __launch_bounds__(256, 2) void attention_frag_kernel(const float* __restrict__ q, int64_t ldq, const f16x8* __restrict__ kimg,
                                                                const f16x8* __restrict__ vimg, int N, int H, float scale,
                                                                float* __restrict__ out, int64_t ldo) {
    constexpr int M = MK * 32;
    constexpr int GROUPS = M * DH / 8;
    constexpr int LDP = M + 8;
    constexpr int PPLANE = 32 * LDP;
    constexpr int OPATCH = 32 * (DH + 4) * 2;
    constexpr int WSTRIDE = (2 * PPLANE > OPATCH) ? 2 * PPLANE : OPATCH;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tile = blockIdx.x, h = blockIdx.y, c = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const f16x8* __restrict__ KH = kimg + ((int64_t)c * H + h) * 2 * GROUPS + lane;
    const f16x8* __restrict__ VH = vimg + ((int64_t)c * H + h) * 2 * GROUPS + lane;

    // Q fragments (row = query, 8 consecutive d per k-step), split in registers
    const int q_row = tile * QT + wave * 32 + lr;
    const float* __restrict__ qp = q + ((int64_t)c * N + min(q_row, N - 1)) * ldq + h * DH;
    f32x4 qa[DH / 16], qb[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
        qa[s] = *reinterpret_cast<const f32x4*>(qp + s * 16 + lh * 8);
        qb[s] = *reinterpret_cast<const f32x4*>(qp + s * 16 + lh * 8 + 4);
    }
    // first K fragments travel together with Q
    f16x8 nbh[MK], nbl[MK];
#pragma unroll
    for (int j = 0; j < MK; ++j) { nbh[j] = KH[(j * (DH / 16)) * 64]; nbl[j] = KH[GROUPS + (j * (DH / 16)) * 64]; }
    f16x8 qh[DH / 16], ql[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 x, y;
            split1(qa[s][e], x, y); qh[s][e] = x; ql[s][e] = y;
            split1(qb[s][e], x, y); qh[s][4 + e] = x; ql[s][4 + e] = y;
        }

    // ---- S = Q K^T, B fragments one k-step ahead
    f32x16 sacc[MK];
#pragma unroll
    for (int j = 0; j < MK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[j][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
        f16x8 bh[MK], bl[MK];
#pragma unroll
        for (int j = 0; j < MK; ++j) { bh[j] = nbh[j]; bl[j] = nbl[j]; }
        if (s + 1 < DH / 16) {
#pragma unroll
            for (int j = 0; j < MK; ++j) { nbh[j] = KH[(j * (DH / 16) + s + 1) * 64]; nbl[j] = KH[GROUPS + (j * (DH / 16) + s + 1) * 64]; }
        } else {
#pragma unroll
            for (int j = 0; j < MK; ++j) { nbh[j] = VH[(j * (M / 16)) * 64]; nbl[j] = VH[GROUPS + (j * (M / 16)) * 64]; }   // first V step (MK <= DH/32)
        }
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[s], bh[j], sacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[s], bl[j], sacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[s], bh[j], sacc[j], 0, 0, 0);
    }

    // ---- softmax over keys (fp32, in the accumulator layout)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < MK; ++j) { sacc[j][r] *= scale; m = fmaxf(m, sacc[j][r]); }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < MK; ++j) { sacc[j][r] = __builtin_amdgcn_exp2f((sacc[j][r] - m) * 1.4426950408889634f); sum += sacc[j][r]; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.0f / sum;       // v_exp_f32 (1 ulp) and one reciprocal per row: the kernel is VALU-bound, not MFMA-bound
#pragma unroll
        for (int j = 0; j < MK; ++j) sacc[j][r] = sacc[j][r] * inv;
    }

    // ---- P -> private per-wave patch in A-operand order (no barrier: nobody else touches it)
    _Float16* Ph = lds + wave * WSTRIDE;
#pragma unroll
    for (int j = 0; j < MK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            _Float16 a, b;
            split1(sacc[j][r], a, b);
            const int off = ((r & 3) + 8 * (r >> 2) + 4 * lh) * LDP + j * 32 + lr;
            Ph[off] = a;
            Ph[PPLANE + off] = b;
        }

    // ---- O = P V: DH/32 = 4 column blocks, V fragments one k-step ahead (the first MK of the first step are already here)
    constexpr int NV = DH / 32;
    f32x16 oacc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;
    f16x8 vh[NV], vl[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (j < MK) { vh[j] = nbh[j]; vl[j] = nbl[j]; }
        else { vh[j] = VH[(j * (M / 16)) * 64]; vl[j] = VH[GROUPS + (j * (M / 16)) * 64]; }
    }
#pragma unroll
    for (int s = 0; s < M / 16; ++s) {
        f16x8 bh[NV], bl[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { bh[j] = vh[j]; bl[j] = vl[j]; }
        if (s + 1 < M / 16) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { vh[j] = VH[(j * (M / 16) + s + 1) * 64]; vl[j] = VH[GROUPS + (j * (M / 16) + s + 1) * 64]; }
        }
        const int aoff = lr * LDP + s * 16 + lh * 8;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&Ph[aoff]);
        const f16x8 al = *reinterpret_cast<const f16x8*>(&Ph[PPLANE + aoff]);
#pragma unroll
        for (int j = 0; j < NV; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], oacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], oacc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], oacc[j], 0, 0, 0);
    }

    // ---- transposed wide store (see attention_kernel)
    float* patch = reinterpret_cast<float*>(Ph);
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * (DH + 4) + j * 32 + lr] = oacc[j][r];
#pragma unroll
    for (int qi = 0; qi < 32 * (DH / 4) / 64; ++qi) {
        const int idx = qi * 64 + lane;
        const int rl = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
        const int row = tile * QT + wave * 32 + rl;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(&patch[rl * (DH + 4) + c4]);
        if (row < N) *reinterpret_cast<f32x4*>(out + ((int64_t)c * N + row) * ldo + h * DH + c4) = v4;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Third structure: the TRANSPOSED product.  S^T = K Q^T puts keys on the accumulator's rows (registers) and queries on its columns
// (lanes): a query's whole score row then lives in ONE lane pair (l, l+32), so
//   * the softmax is 64 in-register max / exp / sum operations and two cross-half exchanges per query, instead of ten shuffles
//     per accumulator register (the second structure's softmax + P patch cost more VALU / LDS time than its MFMAs);
//   * the probabilities are already in B-operand order for O^T = V^T P^T: lane = query, k = keys -- in the accumulator's row
//     order, which is why the V image is packed with KEYPERM -- so P never goes through LDS; it is split to hi/lo in place.
// With LDS free of P patches, a workgroup of 8 waves (256 queries of one cloud and head) stages the K and V fragment images
// there once (128 KB at M = 128) and every wave reads its A fragments with conflict-free ds_read_b128: the second structure
// streamed 128 KB of fragments from L2 per WAVE (2.1 GB per call at B = 64).
// ------------------------------------------------------------------------------------------------------------------
constexpr int QT2 = 256;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

__device__ __forceinline__ void split_pair(float a, float b, f16x2& hi, f16x2& lo) {
    f32x2 x = {__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f)};
    hi = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(hi, f32x2);
    lo = __builtin_convertvector(r, f16x2);
}

// The same split on the instructions the GEMM engines use (gemm_common.h split4_f16): hi = v_cvt_pk_f16_f32, lo = v_fma_mix{lo,hi}_f16 (x - hi is exact
// in fp32 and is rounded straight into the packed destination): 12 vector instructions per 8 values against ~30 of hipcc's lowering of split_pair --
// the softmax and the two splits cost this kernel as much issue time as its MFMAs.  CLAMP: the +-65504 clamp of the packed form (the probabilities
// are in [0, 1] and need none).
template <bool CLAMP>
__device__ __forceinline__ void split8_fast(f32x4 a, f32x4 b, f16x8& hi, f16x8& lo) {
    if (CLAMP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = __builtin_amdgcn_fmed3f(a[e], -65504.0f, 65504.0f); b[e] = __builtin_amdgcn_fmed3f(b[e], -65504.0f, 65504.0f); }
    }
    f16x2 h0, h1, h2, h3, l0, l1, l2, l3;
    asm("v_cvt_pk_f16_f32 %0, %8, %9\n\tv_cvt_pk_f16_f32 %1, %10, %11\n\tv_cvt_pk_f16_f32 %2, %12, %13\n\tv_cvt_pk_f16_f32 %3, %14, %15\n\t"
        "v_fma_mixlo_f16 %4, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %5, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %6, %2, -1.0, %12 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %7, %3, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %4, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %5, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %6, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %7, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    hi = f16x8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
    lo = f16x8{l0[0], l0[1], l1[0], l1[1], l2[0], l2[1], l3[0], l3[1]};
}

// hi part only (rn16 of 8 values): the score product with both operands rounded to binary16 (QK1) needs no lo part
template <bool CLAMP>
__device__ __forceinline__ void round8_fast(f32x4 a, f32x4 b, f16x8& hi) {
    if (CLAMP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = __builtin_amdgcn_fmed3f(a[e], -65504.0f, 65504.0f); b[e] = __builtin_amdgcn_fmed3f(b[e], -65504.0f, 65504.0f); }
    }
    f16x2 h0, h1, h2, h3;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\tv_cvt_pk_f16_f32 %2, %8, %9\n\tv_cvt_pk_f16_f32 %3, %10, %11"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    hi = f16x8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
}

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
    f16x2 h, l;
    split_pair(a[0], a[1], h, l); hi[0] = h[0]; hi[1] = h[1]; lo[0] = l[0]; lo[1] = l[1];
    split_pair(a[2], a[3], h, l); hi[2] = h[0]; hi[3] = h[1]; lo[2] = l[0]; lo[3] = l[1];
    split_pair(b[0], b[1], h, l); hi[4] = h[0]; hi[5] = h[1]; lo[4] = l[0]; lo[5] = l[1];
    split_pair(b[2], b[3], h, l); hi[6] = h[0]; hi[7] = h[1]; lo[6] = l[0]; lo[7] = l[1];
}

// FUSED: no packed images -- the workgroup splits its (cloud, head)'s K and V rows itself while staging them (with one workgroup per (cloud, head),
// as at B = 64, nothing is split twice, and the pack kernel with its workspace round trip disappears: -25 us per call).
// QK1: the score product q k^T with BOTH operands rounded to binary16 (one matrix instruction per block instead of three, no lo part of Q): the
// per-layer term budget's entry for the attention scores (HISTORY.md section 4; measured insensitive like the Q projection itself).  P V keeps three.
template <int MK, bool FUSED, bool QK1 = false>
__global__ __launch_bounds__(512) void attention_t_kernel(const float* __restrict__ q, int64_t ldq, const f16x8* __restrict__ kimg,
                                                          const f16x8* __restrict__ vimg, const float* __restrict__ kraw, int64_t ldk,
                                                          const float* __restrict__ vraw, int64_t ldv, int N, int H, float scale,
                                                          float* __restrict__ out, int64_t ldo) {
    constexpr int M = MK * 32;
    constexpr int GROUPS = M * DH / 8;                   // f16x8 groups per plane and operand
    constexpr int KS = DH / 16;                          // k-steps of S^T (over d)
    constexpr int VS = M / 16;                           // k-steps of O^T (over keys)
    constexpr int NV = DH / 32;                          // row blocks of O^T (over d)
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    f16x8* Ks = reinterpret_cast<f16x8*>(lds);           // [2 planes][MK][KS][64 lanes]
    f16x8* Vs = Ks + 2 * GROUPS;                         // [2 planes][NV][VS][64 lanes]
    const int h = blockIdx.y, c = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const f16x8* __restrict__ KG = kimg + ((int64_t)c * H + h) * 2 * GROUPS;
    const f16x8* __restrict__ VG = vimg + ((int64_t)c * H + h) * 2 * GROUPS;

    // ---- stage both images (all loads in flight together with the Q rows)
    constexpr int PER = FUSED ? 1 : (2 * GROUPS + 511) / 512;        // packed: vectors per thread and image
    constexpr int GP = FUSED ? (GROUPS + 511) / 512 : 1;             // fused: 8-element groups per thread and operand
    f16x8 kst[PER], vst[PER];
    f32x4 ka[GP], kb[GP];
    float vv[GP][8];
    if constexpr (FUSED) {
        const float* __restrict__ kc = kraw + ((int64_t)c * M) * ldk + h * DH;
        const float* __restrict__ vc = vraw + ((int64_t)c * M) * ldv + h * DH;
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int g = min(i * 512 + tid, GROUPS - 1), l = g & 63, blk = g >> 6;
            {   // K image group: key = 32 j + l % 32, d = 16 s + 8 (l / 32) .. + 7
                const int s_ = blk % KS, j_ = blk / KS;
                const float* p = kc + (int64_t)(j_ * 32 + (l & 31)) * ldk + s_ * 16 + (l >> 5) * 8;
                ka[i] = *reinterpret_cast<const f32x4*>(p);
                kb[i] = *reinterpret_cast<const f32x4*>(p + 4);
            }
            {   // V image group (accumulator row order of the keys, see attention_pack_kernel<.., true>): d = 32 jd + l % 32
                const int ks_ = blk % VS, jd_ = blk / VS;
                const float* p = vc + jd_ * 32 + (l & 31);
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[i][e] = p[(int64_t)(ks_ * 16 + (l >> 5) * 4 + (e & 3) + 8 * (e >> 2)) * ldv];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int g = i * 512 + tid;
            if (2 * GROUPS % 512 == 0 || g < 2 * GROUPS) { kst[i] = KG[g]; vst[i] = VG[g]; }
        }
    }
    // Q rows of a tile: lane = query, 8 consecutive d per k-step (B operand of S^T).  The workgroup walks the query tiles
    // blockIdx.x, + gridDim.x, ... of its (cloud, head); the next tile's rows are fetched before the current tile is computed.
    const int n_tiles = (N + QT2 - 1) / QT2;
    f32x4 qa[KS], qb[KS];
    auto load_q = [&](int tile) {
        const int row = min(tile * QT2 + wave * 32 + lr, N - 1);
        const float* __restrict__ qp = q + ((int64_t)c * N + row) * ldq + h * DH;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qa[s] = *reinterpret_cast<const f32x4*>(qp + s * 16 + lh * 8);
            qb[s] = *reinterpret_cast<const f32x4*>(qp + s * 16 + lh * 8 + 4);
        }
    };
    load_q(blockIdx.x);
    if constexpr (FUSED) {
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int g = i * 512 + tid;
            if (GROUPS % 512 == 0 || g < GROUPS) {
                f16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 x, y;
                    split1(ka[i][e], x, y); hi[e] = x; lo[e] = y;
                    split1(kb[i][e], x, y); hi[4 + e] = x; lo[4 + e] = y;
                }
                Ks[g] = hi; Ks[GROUPS + g] = lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 x, y;
                    split1(vv[i][e], x, y); hi[e] = x; lo[e] = y;
                }
                Vs[g] = hi; Vs[GROUPS + g] = lo;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int g = i * 512 + tid;
            if (2 * GROUPS % 512 == 0 || g < 2 * GROUPS) { Ks[g] = kst[i]; Vs[g] = vst[i]; }
        }
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int q_row = tile * QT2 + wave * 32 + lr;
    // opaque per tile: otherwise the (tile-invariant) LDS fragment reads are hoisted out of the loop, into registers that do not exist
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const f16x8* __restrict__ Kl = Ks + lane_t;
    const f16x8* __restrict__ Vl = Vs + lane_t;
    f16x8 qh[KS], ql[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (QK1) round8_fast<true>(qa[s], qb[s], qh[s]);
        else split8_fast<true>(qa[s], qb[s], qh[s], ql[s]);
    }
    if (tile + (int)gridDim.x < n_tiles) load_q(tile + gridDim.x);

    // ---- S^T = K Q^T: row blocks = 32 keys each, A fragments from LDS
    f32x16 sacc[MK];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 ah[MK], al[MK];
#pragma unroll
        for (int j = 0; j < MK; ++j) {
            ah[j] = Kl[(j * KS + s) * 64];
            if (!QK1) al[j] = Kl[GROUPS + (j * KS + s) * 64];
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (QK1) {
#pragma unroll
            for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], qh[s], s == 0 ? zero : sacc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], qh[s], s == 0 ? zero : sacc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], ql[s], sacc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < MK; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], qh[s], sacc[j], 0, 0, 0);
        }
    }

    // ---- softmax over the keys of this lane's query: registers of this lane and of lane ^ 32
    const float sl2 = scale * 1.4426950408889634f;       // exp(x * scale - m) = exp2((x - m') * scale * log2 e), scale > 0
    float m = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < MK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[j][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.0f;
    const float mb = -m * sl2;          // exp2(s sl2 - m sl2): one fma per score
#pragma unroll
    for (int j = 0; j < MK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[j][r] = __builtin_amdgcn_exp2f(fmaf(sacc[j][r], sl2, mb));
            sum += sacc[j][r];
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O^T = V^T P^T: B fragments of k-step (j, t) are the accumulator registers 8t .. 8t+7 of key block j, split in place
    f32x16 oacc[NV];
#pragma unroll
    for (int ks = 0; ks < VS; ++ks) {
        const int j = ks >> 1, t = ks & 1;
        f16x8 ph, pl;
        {
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = sacc[j][8 * t + e] * inv; b[e] = sacc[j][8 * t + 4 + e] * inv; }
            split8_fast<false>(a, b, ph, pl);
        }
        f16x8 vh[NV], vl[NV];
#pragma unroll
        for (int jd = 0; jd < NV; ++jd) {
            vh[jd] = Vl[(jd * VS + ks) * 64];
            vl[jd] = Vl[GROUPS + (jd * VS + ks) * 64];
        }
#pragma unroll
        for (int jd = 0; jd < NV; ++jd) {
            if (ks == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                oacc[jd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[jd], ph, zero, 0, 0, 0);
            } else {
                oacc[jd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[jd], ph, oacc[jd], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jd = 0; jd < NV; ++jd) oacc[jd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[jd], pl, oacc[jd], 0, 0, 0);
#pragma unroll
        for (int jd = 0; jd < NV; ++jd) oacc[jd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[jd], ph, oacc[jd], 0, 0, 0);
    }

    // ---- store: lane = query, register quads = 4 consecutive d; the two lane halves fill 32 contiguous bytes, the four quads of a
    // block one 128-byte line (merged in L2)
    if (q_row < N) {
        float* __restrict__ op = out + ((int64_t)c * N + q_row) * ldo + h * DH + 4 * lh;
#pragma unroll
        for (int jd = 0; jd < NV; ++jd)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v4 = {oacc[jd][4 * g], oacc[jd][4 * g + 1], oacc[jd][4 * g + 2], oacc[jd][4 * g + 3]};
                *reinterpret_cast<f32x4*>(op + jd * 32 + 8 * g) = v4;
            }
    }
    }
}

template <int MK>
int launch_attention_t(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int H,
                       float scale, float* out, int64_t ldo, void* workspace, hipStream_t s, int qk_terms = 0) {
    constexpr int M = MK * 32;
    constexpr int GROUPS = M * DH / 8;
    const size_t lds = (size_t)4 * GROUPS * sizeof(f16x8);
    f16x8* kimg = reinterpret_cast<f16x8*>(workspace);
    f16x8* vimg = kimg + (int64_t)C * H * 2 * GROUPS;
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_t_kernel<MK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_t_kernel<MK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_t_kernel<MK, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    // query tiles per workgroup: as many as leave at least two workgroups per CU (the K / V images are staged once per workgroup)
    const int n_tiles = (N + QT2 - 1) / QT2;
    static const int force_x = [] { const char* e = getenv("OGMM_ATTN_GX"); return e ? atoi(e) : 0; }();
    static const bool packed = [] { const char* e = getenv("OGMM_ATTN_PACKED"); return e && e[0] == '1'; }();      // A/B: separate pack kernel
    int gx = (512 + C * H - 1) / (C * H);
    if (force_x > 0) gx = force_x;
    gx = gx < 1 ? 1 : (gx > n_tiles ? n_tiles : gx);
    if (packed) {
        hipLaunchKernelGGL((attention_pack_kernel<MK, true>), dim3((GROUPS + 255) / 256, H, C), dim3(256), 0, s, k, ldk, v, ldv, H, kimg, vimg);
        hipLaunchKernelGGL((attention_t_kernel<MK, false>), dim3(gx, H, C), dim3(512), lds, s, q, ldq, kimg, vimg, k, ldk, v, ldv, N, H, scale, out, ldo);
    } else if (qk_terms == 1) {
        hipLaunchKernelGGL((attention_t_kernel<MK, true, true>), dim3(gx, H, C), dim3(512), lds, s, q, ldq, kimg, vimg, k, ldk, v, ldv, N, H, scale, out, ldo);
    } else {
        hipLaunchKernelGGL((attention_t_kernel<MK, true>), dim3(gx, H, C), dim3(512), lds, s, q, ldq, kimg, vimg, k, ldk, v, ldv, N, H, scale, out, ldo);
    }
    return ogmm::check_launch("ogmm_attention(transposed)");
}

template <int MK>
int launch_attention_frag(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int H,
                          float scale, float* out, int64_t ldo, void* workspace, hipStream_t s) {
    constexpr int M = MK * 32;
    constexpr int GROUPS = M * DH / 8;
    constexpr int PPL = 2 * 32 * (M + 8), OPL = 32 * (DH + 4) * 2;
    const size_t lds = (size_t)4 * (PPL > OPL ? PPL : OPL) * sizeof(_Float16);
    f16x8* kimg = reinterpret_cast<f16x8*>(workspace);
    f16x8* vimg = kimg + (int64_t)C * H * 2 * GROUPS;
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_frag_kernel<MK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((attention_pack_kernel<MK, false>), dim3((GROUPS + 255) / 256, H, C), dim3(256), 0, s, k, ldk, v, ldv, H, kimg, vimg);
    hipLaunchKernelGGL(attention_frag_kernel<MK>, dim3((N + QT - 1) / QT, H, C), dim3(256), lds, s, q, ldq, kimg, vimg, N, H, scale, out, ldo);
    return ogmm::check_launch("ogmm_attention(frag)");
}

template <int MK>
int launch_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int H, float scale,
                     float* out, int64_t ldo, hipStream_t s) {
    constexpr int M = MK * 32;
    constexpr int KPL = 2 * M * (DH + 8), PPL = 2 * 32 * (M + 8), OPL = 32 * (DH + 4) * 2;
    constexpr int WST = PPL > OPL ? PPL : OPL;
    const size_t lds = (size_t)((KPL > 4 * WST ? KPL : 4 * WST) + 2 * DH * (M + 8)) * sizeof(_Float16);
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<MK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(attention_kernel<MK>, dim3((N + QT - 1) / QT, H, C), dim3(256), lds, s, q, ldq, k, ldk, v, ldv, N, H, scale, out, ldo);
    return ogmm::check_launch("ogmm_attention");
}

}  // namespace

extern "C" int64_t ogmm_attention_workspace_bytes(int C, int M, int H, int dh) {
    return (int64_t)C * H * 2 /*K,V*/ * 2 /*hi,lo*/ * M * dh * (int64_t)sizeof(_Float16);
}

extern "C" int ogmm_attention_terms(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int M,
                                    int H, int dh, float scale, float* out, int64_t ldo, int qk_terms, void* workspace, void* stream);

extern "C" int ogmm_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int M,
                              int H, int dh, float scale, float* out, int64_t ldo, void* workspace, void* stream) {
    return ogmm_attention_terms(q, ldq, k, ldk, v, ldv, C, N, M, H, dh, scale, out, ldo, 0, workspace, stream);
}

// qk_terms: 0 / 3 = the score product in three binary16 terms (fp32-class); 1 = both operands rounded to binary16 (a permission: only the transposed
// kernel with a workspace has the form, the others run three terms)
extern "C" int ogmm_attention_terms(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int C, int N, int M,
                                    int H, int dh, float scale, float* out, int64_t ldo, int qk_terms, void* workspace, void* stream) {
    OGMM_REQUIRE(q && k && v && out && C > 0 && N > 0 && H > 0, "ogmm_attention: null pointer or empty input");
    OGMM_REQUIRE(dh == DH, "ogmm_attention: head dimension %d not supported (built for %d)", dh, DH);
    OGMM_REQUIRE(M == 32 || M == 64 || M == 128, "ogmm_attention: %d anchors not supported (32, 64 or 128)", M);
    OGMM_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && ogmm::aligned16(q) && ogmm::aligned16(k) && ogmm::aligned16(v) && ogmm::aligned16(out),
                 "ogmm_attention: row strides must be multiples of 4 and pointers 16-byte aligned");
    hipStream_t s = ogmm::as_stream(stream);
    if (workspace) {
        OGMM_REQUIRE(ogmm::aligned16(workspace), "ogmm_attention: workspace must be 16-byte aligned");
        static const bool old_frag = [] { const char* e = getenv("OGMM_ATTN_FRAG"); return e && e[0] == '1'; }();     // A/B: the second structure
        if (old_frag) {
            if (M == 32) return launch_attention_frag<1>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, workspace, s);
            if (M == 64) return launch_attention_frag<2>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, workspace, s);
            return launch_attention_frag<4>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, workspace, s);
        }
        if (M == 32) return launch_attention_t<1>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, workspace, s, qk_terms);
        if (M == 64) return launch_attention_t<2>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, workspace, s, qk_terms);
        return launch_attention_t<4>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, workspace, s, qk_terms);
    }
    if (M == 32) return launch_attention<1>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, s);
    if (M == 64) return launch_attention<2>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, s);
    return launch_attention<4>(q, ldq, k, ldk, v, ldv, C, N, H, scale, out, ldo, s);
}
