// K2+K3 fused, PRODUCER / CONSUMER form (k = 20): the EdgeConv chain of models/dgcnn.py:135-150 as a pipeline over 32-row blocks inside one
// persistent 8-wave workgroup per CU.
//
// Why: the phase probe of edgeconv_fused.hip (OGMM_EDGECONV_PROBE=1) shows 24.9 k shader cycles per 160-row tile against 10.6 k cycles of matrix
// instructions -- layer 4 (73 % of the MFMAs) takes 40 % of the time, the rest goes to the VALU / LDS epilogues of layers 1-3 and to five
// workgroup barriers per tile, and in every phase BOTH waves of a SIMD do the same kind of work: matrix pipe and vector ALU take turns.
// A 1x1 convolution is row-wise, so a 32-row block's chain  edge features -> L1 -> L2 -> L3 -> L4  depends on no other block; only the max over
// the 20 edges of a point crosses blocks, and that is an integer atomicMax into a per-tile pool (post-ReLU values are >= 0).  Hence:
//   * waves 0-3 (one per SIMD) are PRODUCERS: producer p takes every 4th block, computes layers 1-3 for it in a wave-private 9 KiB LDS buffer
//     (h1, overwritten by h2) and writes the block's h3 planes (A-operand order, binary16 hi / lo) into ring slot p;
//   * waves 4-7 (one per SIMD) are CONSUMERS: every consumer takes EVERY block, multiplies it with its own two 32-column blocks of layer 4 (weights
//     resident in 128 registers) and pools;
//   so each SIMD always holds one VALU-heavy and one MFMA-heavy wave.  No workgroup barrier after the prologue: ring slots are handed over through
//   LDS sequence counters (full / empty), the pooled maxima of a tile (8 points x 512 channels, double-buffered) are flushed to xcat by the last
//   consumer to finish the tile's fifth block.
// Arithmetic per element is that of edgeconv_fused.hip (same fmaf chains, same split, same MFMA order per accumulator): xcat is bit-identical.
#include "ogmm_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int KE = 20;                                  // edges per point
constexpr int TP = 8, TB = 5;                           // points and 32-row blocks per tile (8 x 20 = 5 x 32 rows)
constexpr int LD64 = 64 + 8, LD128 = 128 + 8;           // plane row lengths in halfs (conflict-free ds_read_b128)
constexpr int H12_HALFS = 2 * 32 * LD64;                // one block's h1 / h2 planes: 9216 B
constexpr int H3_HALFS = 2 * 32 * LD128;                // one block's h3 planes: 17408 B
constexpr int SLOTS = 4;
constexpr int OFF_RING = 0;
constexpr int OFF_PRIV = OFF_RING + SLOTS * H3_HALFS * 2;          // 69632
constexpr int OFF_W2 = OFF_PRIV + 4 * H12_HALFS * 2;               // 106496
constexpr int OFF_POOL = OFF_W2 + 16384;                           // 122880
constexpr int OFF_EF = OFF_POOL + 2 * TP * 512 * 4;                // 155648
constexpr int OFF_FLAGS = OFF_EF + 4 * 2 * 32 * 16;                // 159744
constexpr int LDS_BYTES = OFF_FLAGS + 64;                          // 159808

// split of two values (already clamped to [0, 65504] by the activation): hi = rn16(x) by one v_cvt_pk_f16_f32, lo = rn16(x - hi) by two v_fma_mix*_f16
// (the binary16 source is read in place, x - hi is exact in fp32): three vector instructions instead of eight, the same bits as the
// convert / subtract / convert form of edgeconv_fused.hip
__device__ __forceinline__ void split_h2(float a, float b, f16x2& hi, f16x2& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "s_nop 0\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "s_nop 0\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo) : "v"(a), "v"(b));
}

__device__ __forceinline__ void trade_pair(float v0, float v1, bool odd, float& left, float& right) {
    const float n0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v0), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    const float n1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0xB1, 0xf, 0xf, true));
    left = odd ? n1 : v0;
    right = odd ? v1 : n0;
}

__device__ __forceinline__ int lds_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// (bounded: a protocol error must end as a poisoned result, not as a hung GPU -- `dead` is raised, every later wait of the workgroup falls through)
__device__ __forceinline__ void lds_wait_ge(const int* p, int target, int* dead) {
    int polls = 0;
    while (lds_load(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        // dead[1]: the poll limit (2^20; OGMM_EDGECONV_POLL_LIMIT lowers it for the test of this path)
        if (++polls > dead[1] || ((polls & 1023) == 0 && lds_load(dead))) { __hip_atomic_store(dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// point (0..7) of row r of the tile
constexpr int point_of(int r) { return r / KE; }

// Pooled maximum of one accumulator set over the rows of block B (rows 32 B .. 32 B + 31 of the tile; register r of a lane is row
// (r & 3) + 8 (r >> 2) + 4 lh): every register's point is a compile-time constant per lane half, so the pooling is one v_max per element plus an
// atomicMax where a lane half leaves a point -- the structure of edgeconv_fused.hip's KC > 0 path.
template <int B>
__device__ __forceinline__ void pool_block(const f32x16& acc, int* pool_col /* &pool[0][column] */, int lh) {
    int g_lo = -1, g_hi = -1;
    float cur = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rowc = B * 32 + (r & 3) + 8 * (r >> 2);
        const int n_lo = point_of(rowc), n_hi = point_of(rowc + 4);
        const bool new_lo = n_lo != g_lo, new_hi = n_hi != g_hi;
        const float v = acc[r];
        if (!new_lo && !new_hi) {
            cur = fmaxf(cur, v);
        } else {
            const bool mine_new = lh ? new_hi : new_lo;
            const int prev = lh ? g_hi : g_lo;
            if (mine_new) {
                if (prev >= 0) atomicMax(&pool_col[prev * 512], __float_as_int(cur));
                cur = v;
            } else {
                cur = fmaxf(cur, v);
            }
        }
        g_lo = n_lo;
        g_hi = n_hi;
    }
    atomicMax(&pool_col[(lh ? g_hi : g_lo) * 512], __float_as_int(cur));
}

// relu(acc * sc + sh) in place, then (optionally) the block's planes for the next layer: lane pairs trade one of their two rows so that each lane
// writes a column PAIR of one row with a single ds_write_b32 per plane (as edgeconv_fused.hip).
__device__ __forceinline__ void activate(f32x16& acc, float sc, float sh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_fmed3f(fmaf(acc[r], sc, sh), 0.0f, 65504.0f);          // ReLU + the binary16 range clamp of the split
}
__device__ __forceinline__ void write_planes(const f32x16& acc, _Float16* out, int LDO, int col, int lane) {
    const int lh = lane >> 5;
    const bool odd = lane & 1;
    const int OPL = 32 * LDO;
    _Float16* __restrict__ dst = out + ((4 * lh + (odd ? 1 : 0)) * LDO + (col & ~1));
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float left, right;
        trade_pair(acc[2 * q], acc[2 * q + 1], odd, left, right);
        f16x2 hi2, lo2;
        split_h2(left, right, hi2, lo2);
        const int off = (((2 * q) & 3) + 8 * ((2 * q) >> 2)) * LDO;
        *reinterpret_cast<f16x2*>(&dst[off]) = hi2;
        *reinterpret_cast<f16x2*>(&dst[OPL + off]) = lo2;
    }
}

// phase probe (OGMM_EDGECONV_PROBE=1): shader cycles of lane 0 per role and phase {L1, L2, L3 up to the slot wait, slot wait, L3 epilogue, consumer wait,
// consumer compute, blocks}
__device__ unsigned long long g_pc_probe[8];
// (accumulated in registers, one atomic per wave and slot at the end: an atomic per block and phase queues behind itself and distorts what it measures)
template <bool PROBE>
__device__ __forceinline__ void pc_probe(long long& t, unsigned long long (&tot)[8], int slot) {
    if (PROBE) {
        const long long now = clock64();
        tot[slot] += (unsigned long long)(now - t);
        t = now;
    }
}

struct EdgeW {
    const float* W1; const float* s1; const float* t1;
    const void* h2; const void* l2; const float* s2; const float* t2; float inv2;
    const void* h3; const void* l3; const float* s3; const float* t3; float inv3;
    const void* h4; const void* l4; const float* s4; const float* t4; float inv4;
};

// ---- producer, block B of a tile: layers 1-3 in the wave's private buffer, h3 planes into `slot`, x1 / x2 / x3 maxima into the tile's pool
template <int B, bool PROBE>
__device__ __forceinline__ void produce_block(const float4* efd, const float4* efc, _Float16* priv, _Float16* slot, const _Float16* w2img,
                                              const f16x8 (&wb3)[4][4][2], int* pool, const float (&wv)[6], float s1, float t1,
                                              const float (&sc2)[2], const float (&sh2)[2], const float (&sc3)[4], const float (&sh3)[4], int lane,
                                              const int* empty_flag, int empty_target, int* dead, unsigned long long (&tot)[8]) {
    const int lr = lane & 31, lh = lane >> 5;
    const bool odd = lane & 1;
    long long pt = PROBE ? clock64() : 0;
    // ---- layer 1 (VALU): lane = channel; rows in batches of 8 with their edge vectors loaded up front (the loop is a chain of LDS latency otherwise:
    // measured 6.4 k cycles per block), the centre term once per POINT (a block of 32 rows touches at most three points, known at compile time),
    // two rows per step written as a column pair per lane (see write_planes)
    {
        _Float16* __restrict__ dst = priv + ((odd ? 1 : 0) * LD64 + (lane & ~1));
        constexpr int P0 = point_of(B * 32), P1 = point_of(B * 32 + 31);          // first and last point of the block (P1 - P0 <= 2)
        float ctr[3];
#pragma unroll
        for (int i = 0; i <= P1 - P0; ++i) {
            const int row = (P0 + i) * KE > B * 32 ? (P0 + i) * KE - B * 32 : 0;          // a row of point P0 + i inside this block
            const float4 c = efc[row];
            ctr[i] = fmaf(wv[5], c.z, fmaf(wv[4], c.y, wv[3] * c.x));
        }
        int g = -1;
        float cur = 0.0f;
#pragma unroll
        for (int q8 = 0; q8 < 4; ++q8) {
            float4 a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = efd[q8 * 8 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = q8 * 4 + i;
                const int p0 = point_of(B * 32 + 2 * q), p1 = point_of(B * 32 + 2 * q + 1);
                const float4 a0 = a[2 * i], a1 = a[2 * i + 1];
                const float v0 = fmaxf(fmaf(fmaf(wv[2], a0.z, fmaf(wv[1], a0.y, wv[0] * a0.x)) + ctr[p0 - P0], s1, t1), 0.0f);
                const float v1 = fmaxf(fmaf(fmaf(wv[2], a1.z, fmaf(wv[1], a1.y, wv[0] * a1.x)) + ctr[p1 - P0], s1, t1), 0.0f);
                if (p0 != g) { if (g >= 0) atomicMax(&pool[g * 512 + lane], __float_as_int(cur)); g = p0; cur = v0; } else cur = fmaxf(cur, v0);
                if (p1 != g) { atomicMax(&pool[g * 512 + lane], __float_as_int(cur)); g = p1; cur = v1; } else cur = fmaxf(cur, v1);
                float left, right;
                trade_pair(fminf(v0, 65504.0f), fminf(v1, 65504.0f), odd, left, right);
                f16x2 hi2, lo2;
                split_h2(left, right, hi2, lo2);
                *reinterpret_cast<f16x2*>(&dst[2 * q * LD64]) = hi2;
                *reinterpret_cast<f16x2*>(&dst[32 * LD64 + 2 * q * LD64]) = lo2;
            }
        }
        atomicMax(&pool[g * 512 + lane], __float_as_int(cur));
    }
    pc_probe<PROBE>(pt, tot, 0);
    // ---- layer 2: 64 -> 64 (two column blocks), weights from the LDS image; h2 overwrites h1 once both accumulators are complete
    {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&priv[lr * LD64 + s * 16 + lh * 8]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&priv[32 * LD64 + lr * LD64 + s * 16 + lh * 8]);
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(&w2img[((j * 4 + s) * 64 + lane) * 8]);
                bl[j] = *reinterpret_cast<const f16x8*>(&w2img[4096 + ((j * 4 + s) * 64 + lane) * 8]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            activate(acc[j], sc2[j], sh2[j]);
            write_planes(acc[j], priv, LD64, j * 32 + lr, lane);          // (same wave: its LDS reads above are done before these writes execute)
            pool_block<B>(acc[j], pool + 64 + j * 32 + lr, lh);
        }
    }
    pc_probe<PROBE>(pt, tot, 1);
    // ---- layer 3: 64 -> 128 (four column blocks, weights in registers) -> the ring slot's h3 planes
    {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&priv[lr * LD64 + s * 16 + lh * 8]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&priv[32 * LD64 + lr * LD64 + s * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wb3[j][s][0], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wb3[j][s][1], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wb3[j][s][0], acc[j], 0, 0, 0);
        }
        // the slot is needed only now: the consumers had layers 1-3 of this block's time to finish with its previous contents
        pc_probe<PROBE>(pt, tot, 2);
        lds_wait_ge(empty_flag, empty_target, dead);
        pc_probe<PROBE>(pt, tot, 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            activate(acc[j], sc3[j], sh3[j]);
            write_planes(acc[j], slot, LD128, j * 32 + lr, lane);
            pool_block<B>(acc[j], pool + 128 + j * 32 + lr, lh);
        }
        pc_probe<PROBE>(pt, tot, 4);
    }
}

// ---- consumer, block B: this wave's two 32-column blocks of layer 4 against the slot's h3 planes; x4 maxima into the pool
template <int B>
__device__ __forceinline__ void consume_block(const _Float16* slot, const f16x8 (&wb4)[2][8][2], int* pool, int* empty_flag, int cb0,
                                              const float (&sc4)[2], const float (&sh4)[2], int lane) {
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    f16x8 ah[2], al[2];
    ah[0] = *reinterpret_cast<const f16x8*>(&slot[lr * LD128 + lh * 8]);
    al[0] = *reinterpret_cast<const f16x8*>(&slot[32 * LD128 + lr * LD128 + lh * 8]);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int cur = s & 1;
        if (s + 1 < 8) {
            ah[cur ^ 1] = *reinterpret_cast<const f16x8*>(&slot[lr * LD128 + (s + 1) * 16 + lh * 8]);
            al[cur ^ 1] = *reinterpret_cast<const f16x8*>(&slot[32 * LD128 + lr * LD128 + (s + 1) * 16 + lh * 8]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], wb4[j][s][0], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], wb4[j][s][1], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], wb4[j][s][0], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // every fragment of the slot is in registers (the last reads were consumed by the last MFMAs): hand the slot back
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(empty_flag, 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        activate(acc[j], sc4[j], sh4[j]);
        pool_block<B>(acc[j], pool + 256 + (cb0 + j) * 32 + lr, lh);
    }
}

template <bool PROBE, bool PRIO = false>
__global__ __launch_bounds__(512) void edgeconv_pc_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int N, int64_t total_pts,
                                                          int64_t n_tiles, const EdgeW w, float* __restrict__ xcat, int64_t ldx, int32_t* status,
                                                          int poll_limit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* ring = reinterpret_cast<_Float16*>(smem + OFF_RING);
    _Float16* w2img = reinterpret_cast<_Float16*>(smem + OFF_W2);          // [hi 8 KiB | lo 8 KiB], fragment-major
    int* pools = reinterpret_cast<int*>(smem + OFF_POOL);                  // [2][TP][512]
    int* flags = reinterpret_cast<int*>(smem + OFF_FLAGS);                 // full[4], empty[4], done[2], flushed[2]
    int* full = flags, *empty = flags + 4, *done = flags + 8, *flushed = flags + 10, *dead = flags + 12;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31;
    unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- prologue: flags and pools zeroed, layer 2's weight image staged
    for (int i = tid; i < 2 * TP * 512; i += 512) pools[i] = 0;
    if (tid < 16) flags[tid] = tid == 13 ? poll_limit : 0;
    {
        const int4* __restrict__ sh = reinterpret_cast<const int4*>(w.h2);
        const int4* __restrict__ sl = reinterpret_cast<const int4*>(w.l2);
        int4* dst = reinterpret_cast<int4*>(w2img);
        dst[tid] = sh[tid];                 // 512 x 16 B = 8 KiB per plane
        dst[512 + tid] = sl[tid];
    }
    __syncthreads();

    // this workgroup's blocks: seq = 0, 1, ...: tile = blockIdx.x + (seq / 5) * gridDim.x, block = seq % 5
    const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t n_seq = my_tiles * TB;

    if (wave < 4) {
        // =============================================================== PRODUCER p = wave
        const int p = wave;
        _Float16* priv = reinterpret_cast<_Float16*>(smem + OFF_PRIV) + p * H12_HALFS;
        _Float16* slot = ring + p * H3_HALFS;
        float4* efd = reinterpret_cast<float4*>(smem + OFF_EF) + p * 64;          // [32] x_j - x_i
        float4* efc = efd + 32;                                                   // [32] x_i
        f16x8 wb3[4][4][2];
        {
            const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(w.h3);
            const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(w.l3);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    wb3[j][s][0] = BH[(j * 4 + s) * 64 + lane];
                    wb3[j][s][1] = BL[(j * 4 + s) * 64 + lane];
                }
        }
        float wv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = w.W1[lane * 6 + i];
        const float s1 = w.s1[lane], t1 = w.t1[lane];
        float sc2[2], sh2[2], sc3[4], sh3[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) { sc2[j] = w.s2[j * 32 + lr] * w.inv2; sh2[j] = w.t2[j * 32 + lr]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc3[j] = w.s3[j * 32 + lr] * w.inv3; sh3[j] = w.t3[j * 32 + lr]; }

        // gather of a block's 32 edge rows by lanes 0-31: index two own blocks ahead, coordinates one ahead (two dependent loads in flight
        // across a whole block's work)
        auto row_point = [&](int64_t seq) -> int64_t {          // global point of this lane's row, or -1
            if (lane >= 32 || seq >= n_seq) return -1;
            const int64_t tile = blockIdx.x + (seq / TB) * gridDim.x;
            const int64_t pt = tile * TP + ((seq % TB) * 32 + lane) / KE;
            return pt < total_pts ? pt : -1;
        };
        auto load_index = [&](int64_t seq) -> int {
            const int64_t pt = row_point(seq);
            if (pt < 0) return -1;
            const int e = (int)(((seq % TB) * 32 + lane) % KE);
            return idx[pt * KE + e];
        };
        float f[6];
        auto load_coords = [&](int64_t seq, int jn) {
#pragma unroll
            for (int i = 0; i < 6; ++i) f[i] = 0.0f;
            const int64_t pt = row_point(seq);
            if (pt >= 0 && jn >= 0) {
                const int64_t j = (pt / N) * N + jn;
                f[3] = xyz[3 * pt]; f[4] = xyz[3 * pt + 1]; f[5] = xyz[3 * pt + 2];
                f[0] = xyz[3 * j]; f[1] = xyz[3 * j + 1]; f[2] = xyz[3 * j + 2];
            }
        };
        int j_next = load_index(p);
        load_coords(p, j_next);
        j_next = load_index(p + 4);
        int n_prod = 0;
        for (int64_t seq = p; seq < n_seq; seq += 4, ++n_prod) {
            const int64_t tl = seq / TB;                    // tile index within this workgroup
            const int b = (int)(seq % TB);
            if (lane < 32) {
                efd[lane] = make_float4(f[0] - f[3], f[1] - f[4], f[2] - f[5], 0.0f);
                efc[lane] = make_float4(f[3], f[4], f[5], 0.0f);
            }
            load_coords(seq + 4, j_next);                   // next own block's coordinates (its index arrived during the previous block)
            j_next = load_index(seq + 8);
            // the tile's pool must have been flushed by the tile two before; the slot must have been released by all four consumers
            if (tl >= 2) lds_wait_ge(&flushed[tl & 1], (int)(tl / 2), dead);
            int* pool = pools + (tl & 1) * TP * 512;
            switch (b) {
                case 0: produce_block<0, PROBE>(efd, efc, priv, slot, w2img, wb3, pool, wv, s1, t1, sc2, sh2, sc3, sh3, lane, &empty[p], 4 * n_prod, dead, tot); break;
                case 1: produce_block<1, PROBE>(efd, efc, priv, slot, w2img, wb3, pool, wv, s1, t1, sc2, sh2, sc3, sh3, lane, &empty[p], 4 * n_prod, dead, tot); break;
                case 2: produce_block<2, PROBE>(efd, efc, priv, slot, w2img, wb3, pool, wv, s1, t1, sc2, sh2, sc3, sh3, lane, &empty[p], 4 * n_prod, dead, tot); break;
                case 3: produce_block<3, PROBE>(efd, efc, priv, slot, w2img, wb3, pool, wv, s1, t1, sc2, sh2, sc3, sh3, lane, &empty[p], 4 * n_prod, dead, tot); break;
                default: produce_block<4, PROBE>(efd, efc, priv, slot, w2img, wb3, pool, wv, s1, t1, sc2, sh2, sc3, sh3, lane, &empty[p], 4 * n_prod, dead, tot); break;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // planes and pool atomics done before the slot is announced
            if (lane == 0) __hip_atomic_store(&full[p], n_prod + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        // =============================================================== CONSUMER c = wave - 4: column blocks 2c, 2c + 1 of layer 4
        const int c = wave - 4;
        // the consumers are the pipeline's pole (48 of the 66 matrix instructions per block and SIMD): static priority over the SIMD's producer, whose
        // matrix instructions and epilogues then fill what the consumer leaves (no per-phase flips: MI355X_MICROARCH.md, two waves per SIMD, item 4)
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        f16x8 wb4[2][8][2];
        {
            const f16x8* __restrict__ BH = reinterpret_cast<const f16x8*>(w.h4);
            const f16x8* __restrict__ BL = reinterpret_cast<const f16x8*>(w.l4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    wb4[j][s][0] = BH[((2 * c + j) * 8 + s) * 64 + lane];
                    wb4[j][s][1] = BL[((2 * c + j) * 8 + s) * 64 + lane];
                }
        }
        float sc4[2], sh4[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { sc4[j] = w.s4[(2 * c + j) * 32 + lr] * w.inv4; sh4[j] = w.t4[(2 * c + j) * 32 + lr]; }
        for (int64_t seq = 0; seq < n_seq; ++seq) {
            const int64_t tl = seq / TB;
            const int b = (int)(seq % TB);
            const int sl = (int)(seq & 3);
            long long ct = PROBE ? clock64() : 0;
            lds_wait_ge(&full[sl], (int)(seq >> 2) + 1, dead);
            pc_probe<PROBE>(ct, tot, 5);
            const _Float16* slot = ring + sl * H3_HALFS;
            int* pool = pools + (tl & 1) * TP * 512;
            switch (b) {
                case 0: consume_block<0>(slot, wb4, pool, &empty[sl], 2 * c, sc4, sh4, lane); break;
                case 1: consume_block<1>(slot, wb4, pool, &empty[sl], 2 * c, sc4, sh4, lane); break;
                case 2: consume_block<2>(slot, wb4, pool, &empty[sl], 2 * c, sc4, sh4, lane); break;
                case 3: consume_block<3>(slot, wb4, pool, &empty[sl], 2 * c, sc4, sh4, lane); break;
                default: consume_block<4>(slot, wb4, pool, &empty[sl], 2 * c, sc4, sh4, lane); break;
            }
            pc_probe<PROBE>(ct, tot, 6);
            if (PROBE && c == 0) tot[7] += 1;
            if (b == TB - 1) {
                // the tile's last block: the consumer that finishes it last writes the tile's 8 x 512 pooled maxima to xcat and re-zeroes the pool
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                int last = 0;
                if (lane == 0) last = atomicAdd(&done[tl & 1], 1) == 4 * (int)(tl / 2) + 3;
                last = __builtin_amdgcn_readfirstlane(last);
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const int64_t tile = blockIdx.x + tl * gridDim.x;
                    const int64_t p0 = tile * TP;
                    const int pts = (int)min((int64_t)TP, total_pts - p0);
#pragma unroll
                    for (int it = 0; it < TP * 512 / 4 / 64; ++it) {          // 1024 float4 cells, 16 per lane
                        const int cell = it * 64 + lane, pt = cell >> 7, ch = (cell & 127) * 4;
                        int4* cp = reinterpret_cast<int4*>(&pool[pt * 512 + ch]);
                        const int4 v = *cp;
                        if (pt < pts)
                            *reinterpret_cast<float4*>(&xcat[(p0 + pt) * ldx + ch]) =
                                make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
                        *cp = make_int4(0, 0, 0, 0);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) atomicAdd(&flushed[tl & 1], 1);
                }
            }
        }
    }
    // a wait ran into its limit (a protocol error): reported through the caller's status word (bit OGMM_STATUS_EDGECONV_PROTOCOL; atomically, so that the
    // report cannot be overwritten as an output element could), and the result is made loudly wrong as well -- NaN into this workgroup's first output row
    if (lane == 0 && lds_load(dead)) {
        if (status) atomicOr(status, OGMM_STATUS_EDGECONV_PROTOCOL);
        xcat[(int64_t)blockIdx.x * TP * ldx] = __builtin_nanf("");
    }
    if (PROBE && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (tot[i]) atomicAdd(&g_pc_probe[i], tot[i]);
    }
}

}  // namespace

// Same contract as ogmm_edgeconv_fused (k = 20 only): see include/ogmm_hip.h.
extern "C" int ogmm_edgeconv_pc(const float* xyz, const int32_t* idx, int C, int N, int k, const float* W1, const float* s1, const float* t1,
                                const void* h2, const void* l2, const float* s2, const float* t2, float inv2, const void* h3,
                                const void* l3, const float* s3, const float* t3, float inv3, const void* h4, const void* l4,
                                const float* s4, const float* t4, float inv4, float* xcat, int64_t ldx, int32_t* status, void* stream) {
    OGMM_REQUIRE(xyz && idx && W1 && s1 && t1 && h2 && l2 && s2 && t2 && h3 && l3 && s3 && t3 && h4 && l4 && s4 && t4 && xcat,
                 "ogmm_edgeconv_pc: null pointer");
    const char* pl = getenv("OGMM_EDGECONV_POLL_LIMIT");          // (read per call: a test lowers it to force the protocol-error path)
    const int poll_limit = pl && atoi(pl) > 0 ? atoi(pl) : (1 << 20);
    OGMM_REQUIRE(C > 0 && N > 0 && k == KE && ldx >= 512 && ldx % 4 == 0 && reinterpret_cast<uintptr_t>(xcat) % 16 == 0,
                 "ogmm_edgeconv_pc: k must be 20, ldx >= 512 and a multiple of 4 (C=%d N=%d k=%d ldx=%lld)", C, N, k, (long long)ldx);
    const int64_t total = (int64_t)C * N;
    static ogmm::PerDeviceOnce once;
    static int n_cu_of[64] = {};
    const int dev = once.device();
    if (once.first() || dev < 0 || dev >= 64 || !n_cu_of[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_pc_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_pc_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (dev >= 0 && dev < 64) n_cu_of[dev] = n;
    }
    const int n_cu = (dev >= 0 && dev < 64 && n_cu_of[dev]) ? n_cu_of[dev] : 256;
    EdgeW w{W1, s1, t1, h2, l2, s2, t2, inv2, h3, l3, s3, t3, inv3, h4, l4, s4, t4, inv4};
    const int64_t n_tiles = (total + TP - 1) / TP;
    const unsigned blocks = (unsigned)std::min<int64_t>(n_tiles, n_cu);
    static const bool probe = [] { const char* e = getenv("OGMM_EDGECONV_PROBE"); return e && e[0] == '1'; }();
    static const bool noprio = [] { const char* e = getenv("OGMM_EDGECONV_PRIO"); return e && e[0] == '1'; }();
    if (noprio) {          // (kept for A/B: consumers at s_setprio 2 -- measured 704 against 675 us: the producers then become the pole)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_pc_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((edgeconv_pc_kernel<false, true>), dim3(blocks), dim3(512), LDS_BYTES, ogmm::as_stream(stream), xyz, idx, N, total, n_tiles, w, xcat, ldx, status, poll_limit);
    } else
    if (probe) hipLaunchKernelGGL(edgeconv_pc_kernel<true>, dim3(blocks), dim3(512), LDS_BYTES, ogmm::as_stream(stream), xyz, idx, N, total, n_tiles, w, xcat, ldx, status, poll_limit);
    else hipLaunchKernelGGL(edgeconv_pc_kernel<false>, dim3(blocks), dim3(512), LDS_BYTES, ogmm::as_stream(stream), xyz, idx, N, total, n_tiles, w, xcat, ldx, status, poll_limit);
    return ogmm::check_launch("ogmm_edgeconv_pc");
}

// diagnostic (tools/edgeconv_time.py): read and clear the phase probe of the producer / consumer kernel
extern "C" int ogmm_debug_edgeconv_pc_probe(unsigned long long* host8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_pc_probe), sizeof(z)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pc_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
