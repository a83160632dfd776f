// Backward of the anchor attention  O = softmax(Q K^T * scale) V  (models/attn.py:78-82) for the training step: given dO it
// returns dQ, dK, dV per (cloud, head) without materialising the [C,H,N,M] scores (the library path re-formed them with five
// batched fp32 GEMMs, two softmax passes and six layout copies per attention: 4.3 ms at 128 pairs; this kernel: see DESIGN §7).
//
// Arithmetic: exact fp32 on v_mfma_f32_32x32x2_f32.  Its operand layout (A: lane l holds A[row l%32][k l/32]; B: lane l holds
// B[k l/32][col l%32]: ONE element per lane and instruction) makes every transposition free -- any matrix can be read as either
// operand with plain ds_read / register indexing, and an accumulator tile (lane = column, registers = rows 8i + 4(l/32) + r) is
// directly a B operand of a product that contracts over its rows.  Five products per query tile, 64 MFMAs each:
//   S^T  = K Q^T              (rows = keys, columns = queries)       A = K rows of the wave (registers), B = Q tile (LDS)
//   dP^T = V dO^T                                                     A = V rows of the wave (registers), B = dO tile (LDS)
//   dV  += P^T dO             (contracts over the tile's queries)     A = P^T via a per-wave LDS patch,   B = dO tile (LDS)
//   dK  += dS^T Q                                                     A = dS^T via a per-wave LDS patch,  B = Q tile (LDS)
//   dQ   = dS K               (contracts over all keys)               A = dS tile (LDS, all waves),       B = K columns (registers)
// Workgroup = one (cloud, head), 4 waves, one wave per SIMD; it walks the query tiles of 32 rows.  Wave w owns key block w
// (32 keys): its rows of K and V, its 32 columns of K (for dQ) and its [32 keys x 128] blocks of dK and dV stay in registers for the
// whole cloud.  In S^T / dP^T a lane holds one query and 16 of the wave's keys, so the softmax statistics are in-lane sums plus
// one lane^32 exchange; the four waves' (max, sum, sum e*dP) triples meet in LDS (one barrier), which gives the row maximum,
// the normaliser and delta = sum_key P dP at once.  Three barriers per tile, ~320 MFMAs (20.5 k matrix-pipe cycles) per wave.
#include "ogmm_common.h"

namespace {

using namespace ogmm;
using f32x16b = __attribute__((ext_vector_type(16))) float;
using f32x4b = __attribute__((ext_vector_type(4))) float;
using f32x2b = __attribute__((ext_vector_type(2))) float;

constexpr int BDH = 128;                 // head dimension
constexpr int BM = 128;                  // anchors (keys)
constexpr int TQ = 32;                   // queries per tile
constexpr int PT = 130;                  // floats per row of the Q / dO tiles     (8-byte aligned rows, conflict-free b64 column reads)
constexpr int PX = 132;                  // floats per row of the dS exchange tile (16-byte aligned rows)
constexpr int PP = 34;                   // floats per row of a wave's transposition patch
constexpr int PH = 136;                  // binary16 per row of the split Q / dO tiles (SPLIT form): 16-byte aligned rows, conflict-free b128 reads

using h8b = __attribute__((ext_vector_type(8))) _Float16;
using h4b = __attribute__((ext_vector_type(4))) _Float16;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// hi = rn(v), lo = rn(v - hi) in binary16; amax collects |v| for the range check
// (v_max_f32 returns its non-NaN operand: a NaN never shows in amax, so `probe` sums v * 0 -- NaN as soon as one operand is NaN or Inf)
__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo, float& amax, float& probe) {
    amax = fmaxf(amax, fabsf(v));
    probe = fmaf(v, 0.0f, probe);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// step s (0..63) of a contraction over 128 indices takes index sel(s, lh) from lane half lh: two consecutive steps use two adjacent
// indices, so one 8-byte read feeds both
__device__ __forceinline__ constexpr int sel(int s, int lh) { return 4 * (s >> 1) + 2 * lh + (s & 1); }

// SPLIT (round 5, the fp16x3 training step): the two products that contract over the head dimension -- S^T = K Q^T and dP^T = V dO^T, 128 of the 320
// fp32 matrix instructions of a tile -- run on the engines' arithmetic instead: K and V rows of the wave are held as binary16 hi / lo planes (same
// register count), the staged Q / dO tile is ALSO written as split planes [query][d], and a 16-deep v_mfma_f32_32x32x16_f16 step (lane = key / query,
// 8 consecutive d per lane half) replaces eight 2-deep fp32 steps: 48 matrix instructions of 32 cycles for 128 of 64.  The accumulator layout is the
// same, so everything behind the two blocks is untouched; the three products that contract over queries / keys (P, dS as operands) stay exact fp32:
// dS has no fixed scale, and binary16's range would need a per-tile one.
template <bool SPLIT>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                            const float* __restrict__ v, int64_t ldv, const float* __restrict__ dout,
                                                            int64_t lddo, int N, int H, float scale, float* __restrict__ dq, int64_t lddq,
                                                            float* __restrict__ dk, int64_t lddk, float* __restrict__ dv, int64_t lddv,
                                                            int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) float smem_ab[];
    float* Qs = smem_ab;                               // [TQ][PT]
    float* dOs = Qs + TQ * PT;                         // [TQ][PT]
    float* dSx = dOs + TQ * PT;                        // [TQ][PX]   dS of the tile, all keys
    float* Xs = dSx + TQ * PX;                         // [4 waves][3][32]  (max, sum, sum e*dP) per query
    float* patch = Xs + 4 * 3 * 32;                    // [4 waves][2][32][PP]
    _Float16* Qh = reinterpret_cast<_Float16*>(patch + 4 * 2 * 32 * PP);          // SPLIT: [TQ][PH] x {Q hi, Q lo, dO hi, dO lo}
    _Float16* Ql = Qh + TQ * PH;
    _Float16* Gh = Ql + TQ * PH;
    _Float16* Gl = Gh + TQ * PH;
    float amax = 0.0f, nan_probe = 0.0f;

    const int h = blockIdx.x % H, c = blockIdx.x / H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const float* __restrict__ kc = k + ((int64_t)c * BM) * ldk + h * BDH;
    const float* __restrict__ vc = v + ((int64_t)c * BM) * ldv + h * BDH;
    const float* __restrict__ qc = q + ((int64_t)c * N) * ldq + h * BDH;
    const float* __restrict__ gc = dout + ((int64_t)c * N) * lddo + h * BDH;
    float* Pp = patch + wave * 2 * 32 * PP;            // [key of the wave][query]
    float* Sp = Pp + 32 * PP;

    // ---- the wave's constant slices of K and V
    float krow[SPLIT ? 1 : 64], vrow[SPLIT ? 1 : 64], kcol[64];
    h8b kh[SPLIT ? 8 : 1], kl[SPLIT ? 8 : 1], vh[SPLIT ? 8 : 1], vl[SPLIT ? 8 : 1];          // SPLIT: step u holds d = 16 u + 8 lh ... + 7 of the lane's key
    {
        if constexpr (SPLIT) {
            const float* __restrict__ kr = kc + (int64_t)(32 * wave + lr) * ldk + 8 * lh;
            const float* __restrict__ vr = vc + (int64_t)(32 * wave + lr) * ldv + 8 * lh;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const f32x2b a = *reinterpret_cast<const f32x2b*>(kr + 16 * u + i);
                    const f32x2b b = *reinterpret_cast<const f32x2b*>(vr + 16 * u + i);
                    _Float16 h, l;
                    split1(a[0], h, l, amax, nan_probe); kh[u][i] = h; kl[u][i] = l;
                    split1(a[1], h, l, amax, nan_probe); kh[u][i + 1] = h; kl[u][i + 1] = l;
                    split1(b[0], h, l, amax, nan_probe); vh[u][i] = h; vl[u][i] = l;
                    split1(b[1], h, l, amax, nan_probe); vh[u][i + 1] = h; vl[u][i + 1] = l;
                }
        } else {
            const float* __restrict__ kr = kc + (int64_t)(32 * wave + lr) * ldk + 2 * lh;
            const float* __restrict__ vr = vc + (int64_t)(32 * wave + lr) * ldv + 2 * lh;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const f32x2b a = *reinterpret_cast<const f32x2b*>(kr + 4 * j);
                const f32x2b b = *reinterpret_cast<const f32x2b*>(vr + 4 * j);
                krow[2 * j] = a[0]; krow[2 * j + 1] = a[1];
                vrow[2 * j] = b[0]; vrow[2 * j + 1] = b[1];
            }
        }
#pragma unroll
        for (int s = 0; s < 64; ++s) kcol[s] = kc[(int64_t)sel(s, lh) * ldk + 32 * wave + lr];
    }
    f32x16b dvacc[4], dkacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvacc[t][r] = 0.0f; dkacc[t][r] = 0.0f; }

    // ---- tile staging: thread -> 4 float4 of Q and of dO (row f / 32, columns 4 (f % 32) ...)
    f32x4b qst[4], gst[4];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = tid + 256 * u, row = tile * TQ + (f >> 5), c4 = (f & 31) * 4;
            if (row < N) {
                qst[u] = *reinterpret_cast<const f32x4b*>(qc + (int64_t)row * ldq + c4);
                gst[u] = *reinterpret_cast<const f32x4b*>(gc + (int64_t)row * lddo + c4);
            } else {
                qst[u] = f32x4b{0.f, 0.f, 0.f, 0.f};
                gst[u] = f32x4b{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = tid + 256 * u, row = f >> 5, c4 = (f & 31) * 4;
            f32x2b* qd = reinterpret_cast<f32x2b*>(Qs + row * PT + c4);
            f32x2b* gd = reinterpret_cast<f32x2b*>(dOs + row * PT + c4);
            qd[0] = f32x2b{qst[u][0], qst[u][1]}; qd[1] = f32x2b{qst[u][2], qst[u][3]};
            gd[0] = f32x2b{gst[u][0], gst[u][1]}; gd[1] = f32x2b{gst[u][2], gst[u][3]};
            if constexpr (SPLIT) {
                h4b qh, ql, gh, gl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 h, l;
                    split1(qst[u][e], h, l, amax, nan_probe); qh[e] = h; ql[e] = l;
                    split1(gst[u][e], h, l, amax, nan_probe); gh[e] = h; gl[e] = l;
                }
                *reinterpret_cast<h4b*>(Qh + row * PH + c4) = qh; *reinterpret_cast<h4b*>(Ql + row * PH + c4) = ql;
                *reinterpret_cast<h4b*>(Gh + row * PH + c4) = gh; *reinterpret_cast<h4b*>(Gl + row * PH + c4) = gl;
            }
        }
    };

    const int n_tiles = (N + TQ - 1) / TQ;
    const float sl2 = scale * 1.4426950408889634f;
    load_tile(0);
    store_tile();
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();                                                   // (A) tile visible; dSx / Xs of the previous tile are free
        if (tile + 1 < n_tiles) load_tile(tile + 1);
        // opaque per tile: keeps the tile-invariant LDS addressing from being hoisted into registers that are needed elsewhere
        int lr_t = lr, lh_t = lh;
        asm volatile("" : "+v"(lr_t), "+v"(lh_t));

        // ---- S^T and dP^T blocks of this wave's keys
        f32x16b sacc, pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; pacc[r] = 0.0f; }
        if constexpr (SPLIT) {
            const int off = lr_t * PH + 8 * lh_t;
            // the planes of step u + 1 are requested before the matrix instructions of step u are issued (one wave per SIMD: nobody else hides the LDS latency)
            h8b bqh[2], bql[2], bgh[2], bgl[2];
            bqh[0] = *reinterpret_cast<const h8b*>(Qh + off); bql[0] = *reinterpret_cast<const h8b*>(Ql + off);
            bgh[0] = *reinterpret_cast<const h8b*>(Gh + off); bgl[0] = *reinterpret_cast<const h8b*>(Gl + off);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u + 1 < 8) {
                    bqh[(u + 1) & 1] = *reinterpret_cast<const h8b*>(Qh + off + 16 * (u + 1)); bql[(u + 1) & 1] = *reinterpret_cast<const h8b*>(Ql + off + 16 * (u + 1));
                    bgh[(u + 1) & 1] = *reinterpret_cast<const h8b*>(Gh + off + 16 * (u + 1)); bgl[(u + 1) & 1] = *reinterpret_cast<const h8b*>(Gl + off + 16 * (u + 1));
                }
                sacc = MFMA16(kh[u], bql[u & 1], sacc);
                pacc = MFMA16(vh[u], bgl[u & 1], pacc);
                sacc = MFMA16(kl[u], bqh[u & 1], sacc);
                pacc = MFMA16(vl[u], bgh[u & 1], pacc);
                sacc = MFMA16(kh[u], bqh[u & 1], sacc);
                pacc = MFMA16(vh[u], bgh[u & 1], pacc);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const float* __restrict__ qb = Qs + lr_t * PT + 2 * lh_t;
            const float* __restrict__ gb = dOs + lr_t * PT + 2 * lh_t;
            // operands of step j + 2 are requested before the MFMAs of step j are issued (a read right in front of its use leaves
            // the matrix pipe idle for the LDS latency: one wave per SIMD, nobody else to fill it)
            f32x2b bq[3], bg[3];
            bq[0] = *reinterpret_cast<const f32x2b*>(qb); bg[0] = *reinterpret_cast<const f32x2b*>(gb);
            bq[1] = *reinterpret_cast<const f32x2b*>(qb + 4); bg[1] = *reinterpret_cast<const f32x2b*>(gb + 4);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j + 2 < 32) {
                    bq[(j + 2) % 3] = *reinterpret_cast<const f32x2b*>(qb + 4 * (j + 2));
                    bg[(j + 2) % 3] = *reinterpret_cast<const f32x2b*>(gb + 4 * (j + 2));
                }
                sacc = MFMA32(krow[2 * j], bq[j % 3][0], sacc);
                pacc = MFMA32(vrow[2 * j], bg[j % 3][0], pacc);
                sacc = MFMA32(krow[2 * j + 1], bq[j % 3][1], sacc);
                pacc = MFMA32(vrow[2 * j + 1], bg[j % 3][1], pacc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- softmax pieces of this wave's keys for the lane's query: local maximum, e = exp2(s - max), sum e, sum e*dP
        float mw = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] *= sl2; mw = fmaxf(mw, sacc[r]); }
        mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
        float sw = 0.0f, ew = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - mw);
            sw += sacc[r];
            ew = fmaf(sacc[r], pacc[r], ew);
        }
        sw += __shfl_xor(sw, 32, 64);
        ew += __shfl_xor(ew, 32, 64);
        if (lh_t == 0) {
            Xs[(wave * 3 + 0) * 32 + lr_t] = mw;
            Xs[(wave * 3 + 1) * 32 + lr_t] = sw;
            Xs[(wave * 3 + 2) * 32 + lr_t] = ew;
        }
        __syncthreads();                                                   // (X) the four waves' triples
        float mall = -__builtin_inff();
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) mall = fmaxf(mall, Xs[(w2 * 3 + 0) * 32 + lr_t]);
        float tot = 0.0f, dlt = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const float f = __builtin_amdgcn_exp2f(Xs[(w2 * 3 + 0) * 32 + lr_t] - mall);
            tot = fmaf(Xs[(w2 * 3 + 1) * 32 + lr_t], f, tot);
            dlt = fmaf(Xs[(w2 * 3 + 2) * 32 + lr_t], f, dlt);
        }
        const float inv = 1.0f / tot;
        dlt *= inv;
        const float pf = __builtin_amdgcn_exp2f(mw - mall) * inv;
        // P = e * pf;  dS = P (dP - delta) * scale      (a query row past N has Q = dO = 0: P uniform, dP = delta = 0, dS = 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] *= pf;
            pacc[r] = sacc[r] * (pacc[r] - dlt) * scale;
        }

        // ---- dS -> exchange tile [query][key]; P^T, dS^T -> the wave's patches [key][query]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4b*>(dSx + lr_t * PX + 32 * wave + 8 * i + 4 * lh_t) =
                f32x4b{pacc[4 * i], pacc[4 * i + 1], pacc[4 * i + 2], pacc[4 * i + 3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Pp[(8 * i + 4 * lh_t + r) * PP + lr_t] = sacc[4 * i + r];
                Sp[(8 * i + 4 * lh_t + r) * PP + lr_t] = pacc[4 * i + r];
            }
        }
        // ---- dV += P^T dO, dK += dS^T Q over the 32 queries of the tile (16 steps of two queries)
        {
            const float* __restrict__ pa = Pp + lr_t * PP + 2 * lh_t;
            const float* __restrict__ sa = Sp + lr_t * PP + 2 * lh_t;
            // step u = 2 j + e takes query 4 j + e (+ 2 lh); its 8 B values are requested one step ahead, the A pairs two steps ahead
            const float* __restrict__ gb0 = dOs + 2 * lh_t * PT + lr_t;
            const float* __restrict__ qb0 = Qs + 2 * lh_t * PT + lr_t;
            f32x2b ap[2], as[2];
            float bg[2][4], bq[2][4];
            ap[0] = *reinterpret_cast<const f32x2b*>(pa); as[0] = *reinterpret_cast<const f32x2b*>(sa);
#pragma unroll
            for (int t = 0; t < 4; ++t) { bg[0][t] = gb0[32 * t]; bq[0][t] = qb0[32 * t]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = u >> 1, e = u & 1;
                if (u + 1 < 16) {
                    const int qn = 4 * ((u + 1) >> 1) + ((u + 1) & 1);
                    if (e == 1) { ap[(j + 1) & 1] = *reinterpret_cast<const f32x2b*>(pa + 4 * (j + 1)); as[(j + 1) & 1] = *reinterpret_cast<const f32x2b*>(sa + 4 * (j + 1)); }
#pragma unroll
                    for (int t = 0; t < 4; ++t) { bg[(u + 1) & 1][t] = gb0[qn * PT + 32 * t]; bq[(u + 1) & 1][t] = qb0[qn * PT + 32 * t]; }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dvacc[t] = MFMA32(ap[j & 1][e], bg[u & 1][t], dvacc[t]);
                    dkacc[t] = MFMA32(as[j & 1][e], bq[u & 1][t], dkacc[t]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                                   // (B) dS of all keys; every wave is done with Qs / dOs

        // ---- dQ block: 32 queries x the wave's 32 columns, over all 128 keys
        f32x16b qacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) qacc[r] = 0.0f;
        {
            const float* __restrict__ sb = dSx + lr_t * PX + 2 * lh_t;
            f32x2b a[3];
            a[0] = *reinterpret_cast<const f32x2b*>(sb);
            a[1] = *reinterpret_cast<const f32x2b*>(sb + 4);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j + 2 < 32) a[(j + 2) % 3] = *reinterpret_cast<const f32x2b*>(sb + 4 * (j + 2));
                qacc = MFMA32(a[j % 3][0], kcol[2 * j], qacc);
                qacc = MFMA32(a[j % 3][1], kcol[2 * j + 1], qacc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tile + 1 < n_tiles) store_tile();                              // the next tile's rows (nobody reads Qs / dOs before (A))
        {
            float* __restrict__ dqc = dq + ((int64_t)c * N + tile * TQ) * lddq + h * BDH + 32 * wave + lr_t;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 8 * i + 4 * lh_t + r;
                    if (tile * TQ + row < N) dqc[(int64_t)row * lddq] = qacc[4 * i + r];
                }
        }
    }

    if constexpr (SPLIT) {
        if (overflow && (!(amax <= 65504.0f) || nan_probe != nan_probe)) atomicOr(overflow, 1);          // an operand beyond binary16's range, NaN or Inf: the trainer lowers its loss scale
    }
    // ---- dK, dV blocks of the wave: lane = column d, registers = keys
    {
        float* __restrict__ dkc = dk + ((int64_t)c * BM + 32 * wave) * lddk + h * BDH + lr;
        float* __restrict__ dvc = dv + ((int64_t)c * BM + 32 * wave) * lddv + h * BDH + lr;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 8 * i + 4 * lh + r;
                    dkc[(int64_t)key * lddk + 32 * t] = dkacc[t][4 * i + r];
                    dvc[(int64_t)key * lddv + 32 * t] = dvacc[t][4 * i + r];
                }
    }
}

constexpr int BWD_LDS_BYTES = (2 * TQ * PT + TQ * PX + 4 * 3 * 32 + 4 * 2 * 32 * PP) * 4;
constexpr int BWD_LDS_BYTES_SPLIT = BWD_LDS_BYTES + 4 * TQ * PH * 2;
PerDeviceOnce g_bwd_once, g_bwd_once_split;

}  // namespace

extern "C" int ogmm_attention_bwd_supported(int M, int dh) { return (M == BM && dh == BDH) ? 1 : 0; }

static int attention_bwd_impl(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                              int64_t lddo, int C, int N, int M, int H, int dh, float scale, float* dq, int64_t lddq, float* dk,
                              int64_t lddk, float* dv, int64_t lddv, bool split, int* overflow, void* stream) {
    OGMM_REQUIRE(q && k && v && dout && dq && dk && dv, "ogmm_attention_bwd: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && H > 0, "ogmm_attention_bwd: empty problem (C=%d N=%d H=%d)", C, N, H);
    OGMM_REQUIRE(M == BM && dh == BDH, "ogmm_attention_bwd: built for M = %d anchors and dh = %d (got M=%d dh=%d)", BM, BDH, M, dh);
    OGMM_REQUIRE(ldq % 4 == 0 && lddo % 4 == 0 && ldk % 2 == 0 && ldv % 2 == 0 && aligned16(q) && aligned16(dout) && aligned16(k) && aligned16(v),
                 "ogmm_attention_bwd: q / dout rows must be 16-byte aligned, k / v rows 8-byte aligned");
    OGMM_REQUIRE((int64_t)C * H < (int64_t)1 << 31, "ogmm_attention_bwd: C * H exceeds the grid limit");
    if (split) {
        OGMM_REQUIRE(ldk % 2 == 0 && ldv % 2 == 0, "ogmm_attention_bwd_f16x3: k / v rows must be 8-byte aligned");
        if (g_bwd_once_split.first())
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS_BYTES_SPLIT);
        hipLaunchKernelGGL(attention_bwd_kernel<true>, dim3(C * H), dim3(256), BWD_LDS_BYTES_SPLIT, as_stream(stream), q, ldq, k, ldk, v, ldv, dout, lddo, N, H,
                           scale, dq, lddq, dk, lddk, dv, lddv, overflow);
        return check_launch("ogmm_attention_bwd_f16x3");
    }
    if (g_bwd_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS_BYTES);
    hipLaunchKernelGGL(attention_bwd_kernel<false>, dim3(C * H), dim3(256), BWD_LDS_BYTES, as_stream(stream), q, ldq, k, ldk, v, ldv, dout, lddo, N, H,
                       scale, dq, lddq, dk, lddk, dv, lddv, (int*)nullptr);
    return check_launch("ogmm_attention_bwd");
}

extern "C" int ogmm_attention_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                                  int64_t lddo, int C, int N, int M, int H, int dh, float scale, float* dq, int64_t lddq, float* dk,
                                  int64_t lddk, float* dv, int64_t lddv, void* stream) {
    return attention_bwd_impl(q, ldq, k, ldk, v, ldv, dout, lddo, C, N, M, H, dh, scale, dq, lddq, dk, lddk, dv, lddv, false, nullptr, stream);
}

int ogmm_attention_bwd16_launch(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout, int64_t lddo,
                                int C, int N, int H, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                                int* overflow, void* stream);          // train_attn_bwd16.hip

extern "C" int ogmm_attention_bwd_f16x3(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout,
                                        int64_t lddo, int C, int N, int M, int H, int dh, float scale, float* dq, int64_t lddq, float* dk,
                                        int64_t lddk, float* dv, int64_t lddv, int all_products, int* overflow, void* stream) {
    if (!all_products)
        return attention_bwd_impl(q, ldq, k, ldk, v, ldv, dout, lddo, C, N, M, H, dh, scale, dq, lddq, dk, lddk, dv, lddv, true, overflow, stream);
    OGMM_REQUIRE(q && k && v && dout && dq && dk && dv, "ogmm_attention_bwd_f16x3: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && H > 0, "ogmm_attention_bwd_f16x3: empty problem (C=%d N=%d H=%d)", C, N, H);
    OGMM_REQUIRE(M == BM && dh == BDH, "ogmm_attention_bwd_f16x3: built for M = %d anchors and dh = %d (got M=%d dh=%d)", BM, BDH, M, dh);
    OGMM_REQUIRE(ldq % 4 == 0 && lddo % 4 == 0 && ldk % 2 == 0 && ldv % 2 == 0 && aligned16(q) && aligned16(dout) && aligned16(k) && aligned16(v),
                 "ogmm_attention_bwd_f16x3: q / dout rows must be 16-byte aligned, k / v rows 8-byte aligned");
    OGMM_REQUIRE((int64_t)C * H < (int64_t)1 << 31, "ogmm_attention_bwd_f16x3: C * H exceeds the grid limit");
    return ogmm_attention_bwd16_launch(q, ldq, k, ldk, v, ldv, dout, lddo, C, N, H, scale, dq, lddq, dk, lddk, dv, lddv, overflow, stream);
}
