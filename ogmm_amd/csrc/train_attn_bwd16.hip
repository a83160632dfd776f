// Backward of the anchor attention (models/attn.py:78-82), ALL FIVE products on the engines' fp16x3 arithmetic (round 5; the fp16x3 training step).
//
// Same decomposition as train_attn_bwd.hip (workgroup = one (cloud, head), wave w owns keys 32 w ... 32 w + 31 and their dK / dV blocks, 32-query
// tiles, scores re-formed on chip), but every product is three v_mfma_f32_32x32x16_f16 per 16-deep step (hi*lo + lo*hi + hi*hi, fp32 accumulation):
// 120 matrix instructions of 32 cycles per tile and wave where the fp32 kernel issues 320 of 64.  That instruction wants 8 CONSECUTIVE contraction
// indices per lane, so the operands are laid out for it:
//   S^T  = K Q^T   (over d)      A = K rows of the wave: registers, binary16 hi / lo          B = Q tile, row-major planes [query][d] in LDS
//   dP^T = V dO^T  (over d)      A = V rows: registers                                         B = dO tile, row-major planes
//   dV  += P^T dO  (over queries) A = P^T, the wave's patch [key][query] (x 2^10)              B = dO tile TRANSPOSED planes [d][query]
//   dK  += dS^T Q  (over queries) A = dS^T, the wave's patch (x 2^e)                           B = Q tile transposed planes
//   dQ   = dS K    (over keys)    A = dS exchange planes [query][key] (x 2^e, all waves)       B = K columns of the wave: registers
// Ranges: Q, K, V are activations, dO carries the trainer's power-of-two loss scale -- split as they are, values beyond binary16 set the overflow
// word (as for every operand of the engines).  P <= 1 is split as P * 2^10 (so that its low part stays a normal binary16).  dS has no a-priori
// scale: every tile takes the power of two 2^e that puts the tile's largest |dS| just below 2^14 (a workgroup-wide maximum: one more barrier per
// tile); dQ of the tile is un-scaled when it is written, the running dK accumulator is kept in units of the current tile's 2^e (re-scaled by an exact
// power of two whenever e changes), dV by 2^-10 at the end.
// The transposed planes are written by 8-byte groups of four queries with the group index XOR-ed by the writer's (d / 32): rows of 72 bytes, writes
// and reads both conflict-free (HISTORY.md section 7 has the bank arithmetic).
#include "ogmm_common.h"

namespace {

using namespace ogmm;
using f32x16c = __attribute__((ext_vector_type(16))) float;
using f32x4c = __attribute__((ext_vector_type(4))) float;
using f32x2c = __attribute__((ext_vector_type(2))) float;
using h8c = __attribute__((ext_vector_type(8))) _Float16;
using h4c = __attribute__((ext_vector_type(4))) _Float16;

constexpr int CDH = 128;                 // head dimension
constexpr int CM = 128;                  // anchors (keys)
constexpr int CTQ = 32;                  // queries per tile
constexpr int CPH = 136;                 // binary16 per row of the row-major planes (Q, dO: [query][d]; dS exchange: [query][key]): 272-byte rows
constexpr int CPT = 36;                  // binary16 per row of the transposed planes [d][query]: 72-byte rows, 8-byte groups swizzled
constexpr int CPP = 40;                  // binary16 per row of a wave's patches [key][query]: 80-byte rows
constexpr float P_SCALE = 1024.0f;

#define MFMA16C(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split1c(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ h8c join8(h4c a, h4c b) { return h8c{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

__global__ __launch_bounds__(256) void attention_bwd16_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                              const float* __restrict__ v, int64_t ldv, const float* __restrict__ dout,
                                                              int64_t lddo, int N, int H, float scale, float* __restrict__ dq, int64_t lddq,
                                                              float* __restrict__ dk, int64_t lddk, float* __restrict__ dv, int64_t lddv,
                                                              int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* Qh = smem_h;                             // [CTQ][CPH]
    _Float16* Ql = Qh + CTQ * CPH;
    _Float16* Gh = Ql + CTQ * CPH;
    _Float16* Gl = Gh + CTQ * CPH;
    _Float16* QTh = Gl + CTQ * CPH;                    // [CDH][CPT]
    _Float16* QTl = QTh + CDH * CPT;
    _Float16* GTh = QTl + CDH * CPT;
    _Float16* GTl = GTh + CDH * CPT;
    _Float16* Sxh = GTl + CDH * CPT;                   // [CTQ][CPH]  dS * 2^e of the tile, all keys
    _Float16* Sxl = Sxh + CTQ * CPH;
    _Float16* patch = Sxl + CTQ * CPH;                 // [4 waves][P^T hi, P^T lo, dS^T hi, dS^T lo][32 keys][CPP]
    float* Xs = reinterpret_cast<float*>(patch + 4 * 4 * 32 * CPP);          // [4 waves][3][32] (max, sum, sum e*dP) per query, then [4] wave maxima of |dS|
    float* Xm = Xs + 4 * 3 * 32;

    const int h = blockIdx.x % H, c = blockIdx.x / H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const float* __restrict__ kc = k + ((int64_t)c * CM) * ldk + h * CDH;
    const float* __restrict__ vc = v + ((int64_t)c * CM) * ldv + h * CDH;
    const float* __restrict__ qc = q + ((int64_t)c * N) * ldq + h * CDH;
    const float* __restrict__ gc = dout + ((int64_t)c * N) * lddo + h * CDH;
    _Float16* PTh = patch + wave * 4 * 32 * CPP;
    _Float16* PTl = PTh + 32 * CPP;
    _Float16* STh = PTl + 32 * CPP;
    _Float16* STl = STh + 32 * CPP;
    float amax = 0.0f, nan_probe = 0.0f;          // nan_probe: x * 0 summed over every operand element -- NaN as soon as one of them is NaN (or Inf)

    // ---- the wave's constant slices: K and V rows (A operands over d), K columns (B operand over keys)
    h8c kh[8], kl[8], vh[8], vl[8], kch[8], kcl[8];
    {
        const float* __restrict__ kr = kc + (int64_t)(32 * wave + lr) * ldk + 8 * lh;
        const float* __restrict__ vr = vc + (int64_t)(32 * wave + lr) * ldv + 8 * lh;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const f32x2c a = *reinterpret_cast<const f32x2c*>(kr + 16 * u + i);
                const f32x2c b = *reinterpret_cast<const f32x2c*>(vr + 16 * u + i);
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(b[0]), fabsf(b[1]))));
                nan_probe = fmaf(a[0], 0.0f, fmaf(a[1], 0.0f, fmaf(b[0], 0.0f, fmaf(b[1], 0.0f, nan_probe))));
                _Float16 hh, ll;
                split1c(a[0], hh, ll); kh[u][i] = hh; kl[u][i] = ll;
                split1c(a[1], hh, ll); kh[u][i + 1] = hh; kl[u][i + 1] = ll;
                split1c(b[0], hh, ll); vh[u][i] = hh; vl[u][i] = ll;
                split1c(b[1], hh, ll); vh[u][i + 1] = hh; vl[u][i + 1] = ll;
            }
        // step s of the contraction over keys takes keys 16 s + 8 lh ... + 7; the lane's column is d = 32 wave + lr
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t = kc[(int64_t)(16 * s + 8 * lh + i) * ldk + 32 * wave + lr];
                _Float16 hh, ll;
                split1c(t, hh, ll); kch[s][i] = hh; kcl[s][i] = ll;
            }
    }
    f32x16c dvacc[4], dkacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvacc[t][r] = 0.0f; dkacc[t][r] = 0.0f; }
    int e_cur = 0;                                      // dkacc is held in units of 2^-e_cur ... i.e. it accumulates (dS * 2^e_cur)^T Q

    // ---- tile staging: thread (g = tid / 32, L = tid % 32) -> rows 4 g + u (u = 0..3), columns 4 L ... 4 L + 3 of Q and of dO
    const int sg = tid >> 5, sL = tid & 31;
    f32x4c qst[4], gst[4];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = tile * CTQ + 4 * sg + u;
            if (row < N) {
                qst[u] = *reinterpret_cast<const f32x4c*>(qc + (int64_t)row * ldq + 4 * sL);
                gst[u] = *reinterpret_cast<const f32x4c*>(gc + (int64_t)row * lddo + 4 * sL);
            } else {
                qst[u] = f32x4c{0.f, 0.f, 0.f, 0.f};
                gst[u] = f32x4c{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tile = [&]() {
        _Float16 qh_[4][4], ql_[4][4], gh_[4][4], gl_[4][4];          // [u][e]
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                amax = fmaxf(amax, fmaxf(fabsf(qst[u][e]), fabsf(gst[u][e])));
                nan_probe = fmaf(qst[u][e], 0.0f, fmaf(gst[u][e], 0.0f, nan_probe));
                split1c(qst[u][e], qh_[u][e], ql_[u][e]);
                split1c(gst[u][e], gh_[u][e], gl_[u][e]);
            }
        // row-major planes: one 8-byte write per row and plane
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = (4 * sg + u) * CPH + 4 * sL;
            *reinterpret_cast<h4c*>(Qh + o) = h4c{qh_[u][0], qh_[u][1], qh_[u][2], qh_[u][3]};
            *reinterpret_cast<h4c*>(Ql + o) = h4c{ql_[u][0], ql_[u][1], ql_[u][2], ql_[u][3]};
            *reinterpret_cast<h4c*>(Gh + o) = h4c{gh_[u][0], gh_[u][1], gh_[u][2], gh_[u][3]};
            *reinterpret_cast<h4c*>(Gl + o) = h4c{gl_[u][0], gl_[u][1], gl_[u][2], gl_[u][3]};
        }
        // transposed planes: row d = 4 L + e, the four queries 4 g ... 4 g + 3 as one 8-byte group at position g ^ (d / 32)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = 4 * sL + e;
            const int o = d * CPT + 4 * (sg ^ ((d >> 5) & 3));
            *reinterpret_cast<h4c*>(QTh + o) = h4c{qh_[0][e], qh_[1][e], qh_[2][e], qh_[3][e]};
            *reinterpret_cast<h4c*>(QTl + o) = h4c{ql_[0][e], ql_[1][e], ql_[2][e], ql_[3][e]};
            *reinterpret_cast<h4c*>(GTh + o) = h4c{gh_[0][e], gh_[1][e], gh_[2][e], gh_[3][e]};
            *reinterpret_cast<h4c*>(GTl + o) = h4c{gl_[0][e], gl_[1][e], gl_[2][e], gl_[3][e]};
        }
    };

    const int n_tiles = (N + CTQ - 1) / CTQ;
    const float sl2 = scale * 1.4426950408889634f;
    load_tile(0);
    store_tile();
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();                                                   // (A) tile visible; exchange planes / Xs of the previous tile are free
        int lr_t = lr, lh_t = lh;
        asm volatile("" : "+v"(lr_t), "+v"(lh_t));                         // (keeps the tile-invariant LDS addressing out of long-lived registers)

        // ---- S^T and dP^T blocks of this wave's keys (lane = query, registers = 16 of the wave's keys)
        f32x16c sacc, pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; pacc[r] = 0.0f; }
        {
            const int off = lr_t * CPH + 8 * lh_t;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const h8c bqh = *reinterpret_cast<const h8c*>(Qh + off + 16 * u), bql = *reinterpret_cast<const h8c*>(Ql + off + 16 * u);
                const h8c bgh = *reinterpret_cast<const h8c*>(Gh + off + 16 * u), bgl = *reinterpret_cast<const h8c*>(Gl + off + 16 * u);
                sacc = MFMA16C(kh[u], bql, sacc);
                pacc = MFMA16C(vh[u], bgl, pacc);
                sacc = MFMA16C(kl[u], bqh, sacc);
                pacc = MFMA16C(vl[u], bgh, pacc);
                sacc = MFMA16C(kh[u], bqh, sacc);
                pacc = MFMA16C(vh[u], bgh, pacc);
            }
        }

        // ---- softmax pieces (as in the fp32 kernel): local maximum, e = exp2(s - max), sum e, sum e * dP; the four waves' triples meet in LDS
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
        float dmax = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] *= pf;
            pacc[r] = sacc[r] * (pacc[r] - dlt) * scale;
            dmax = fmaxf(dmax, fabsf(pacc[r]));
        }
        // ---- the tile's power of two for dS: workgroup-wide maximum of |dS|
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
        if (lane == 0) Xm[wave] = dmax;
        __syncthreads();                                                   // (X2)
        if (tile + 1 < n_tiles) load_tile(tile + 1);                       // (requested here, not at the top: 32 fewer live registers through the softmax; the three products below cover the latency)
        const float tmax = fmaxf(fmaxf(Xm[0], Xm[1]), fmaxf(Xm[2], Xm[3]));
        int e_t = e_cur;
        if (tmax > 0.0f && tmax < __builtin_inff()) {
            const int ex = (int)((__float_as_uint(tmax) >> 23) & 255u) - 126;          // tmax = m * 2^ex, m in [0.5, 1)  (a subnormal tmax: ex = -126, still fine)
            e_t = min(max(14 - ex, -30), 50);                              // (clamped: |dS| beyond 2^44 would leave binary16 -- flagged below; below 2^-36 it only loses low bits)
            if (tmax * __uint_as_float((uint32_t)(e_t + 127) << 23) > 65504.0f) amax = __builtin_inff();
        }
        const float f_t = __uint_as_float((uint32_t)(e_t + 127) << 23);
        if (e_t != e_cur) {                                                // (uniform over the workgroup)  exact re-scaling of the running dK by a power of two
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dkacc[t][r] = __builtin_ldexpf(dkacc[t][r], e_t - e_cur);
            e_cur = e_t;
        }

        // ---- P * 2^10 -> the wave's P^T patch; dS * 2^e -> the exchange planes [query][key] and the wave's dS^T patch
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h4c sh, sl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 hh, ll;
                split1c(sacc[4 * i + r] * P_SCALE, hh, ll);
                PTh[(8 * i + 4 * lh_t + r) * CPP + lr_t] = hh;
                PTl[(8 * i + 4 * lh_t + r) * CPP + lr_t] = ll;
                split1c(pacc[4 * i + r] * f_t, hh, ll);
                STh[(8 * i + 4 * lh_t + r) * CPP + lr_t] = hh;
                STl[(8 * i + 4 * lh_t + r) * CPP + lr_t] = ll;
                sh[r] = hh; sl[r] = ll;
            }
            *reinterpret_cast<h4c*>(Sxh + lr_t * CPH + 32 * wave + 8 * i + 4 * lh_t) = sh;
            *reinterpret_cast<h4c*>(Sxl + lr_t * CPH + 32 * wave + 8 * i + 4 * lh_t) = sl;
        }
        // ---- dV += P^T dO, dK += dS^T Q over the 32 queries of the tile: two 16-deep steps, four column blocks of 32
        {
            const int po = lr_t * CPP + 8 * lh_t;
            h8c ap[2], apl[2], as[2], asl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ap[s] = *reinterpret_cast<const h8c*>(PTh + po + 16 * s); apl[s] = *reinterpret_cast<const h8c*>(PTl + po + 16 * s);
                as[s] = *reinterpret_cast<const h8c*>(STh + po + 16 * s); asl[s] = *reinterpret_cast<const h8c*>(STl + po + 16 * s);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int d = 32 * t + lr_t;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int g0 = 4 * s + 2 * lh_t;
                    const int o0 = d * CPT + 4 * (g0 ^ t), o1 = d * CPT + 4 * ((g0 + 1) ^ t);
                    const h8c bgh = join8(*reinterpret_cast<const h4c*>(GTh + o0), *reinterpret_cast<const h4c*>(GTh + o1));
                    const h8c bgl = join8(*reinterpret_cast<const h4c*>(GTl + o0), *reinterpret_cast<const h4c*>(GTl + o1));
                    const h8c bqh = join8(*reinterpret_cast<const h4c*>(QTh + o0), *reinterpret_cast<const h4c*>(QTh + o1));
                    const h8c bql = join8(*reinterpret_cast<const h4c*>(QTl + o0), *reinterpret_cast<const h4c*>(QTl + o1));
                    dvacc[t] = MFMA16C(ap[s], bgl, dvacc[t]);
                    dkacc[t] = MFMA16C(as[s], bql, dkacc[t]);
                    dvacc[t] = MFMA16C(apl[s], bgh, dvacc[t]);
                    dkacc[t] = MFMA16C(asl[s], bqh, dkacc[t]);
                    dvacc[t] = MFMA16C(ap[s], bgh, dvacc[t]);
                    dkacc[t] = MFMA16C(as[s], bqh, dkacc[t]);
                }
            }
        }
        __syncthreads();                                                   // (B) dS of all keys visible; every wave is done with the Q / dO planes

        // ---- dQ block: 32 queries x the wave's 32 columns, over all 128 keys (8 steps of 16)
        f32x16c qacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) qacc[r] = 0.0f;
        {
            const int so = lr_t * CPH + 8 * lh_t;
            h8c ah[2], al[2];
            ah[0] = *reinterpret_cast<const h8c*>(Sxh + so); al[0] = *reinterpret_cast<const h8c*>(Sxl + so);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8) { ah[(s + 1) & 1] = *reinterpret_cast<const h8c*>(Sxh + so + 16 * (s + 1)); al[(s + 1) & 1] = *reinterpret_cast<const h8c*>(Sxl + so + 16 * (s + 1)); }
                qacc = MFMA16C(ah[s & 1], kcl[s], qacc);
                qacc = MFMA16C(al[s & 1], kch[s], qacc);
                qacc = MFMA16C(ah[s & 1], kch[s], qacc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tile + 1 < n_tiles) store_tile();                              // the next tile's planes (nobody reads them before (A))
        {
            const float un = __uint_as_float((uint32_t)(127 - e_t) << 23);
            float* __restrict__ dqc = dq + ((int64_t)c * N + tile * CTQ) * lddq + h * CDH + 32 * wave + lr_t;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 8 * i + 4 * lh_t + r;
                    if (tile * CTQ + row < N) dqc[(int64_t)row * lddq] = qacc[4 * i + r] * un;
                }
        }
    }
    // v_max_f32 returns its non-NaN operand, so a NaN in q / k / v / dO never shows in amax, and the range clamps of the split launder it before the accumulators
    // could tell: nan_probe says so
    if (overflow && (!(amax <= 65504.0f) || nan_probe != nan_probe)) atomicOr(overflow, 1);          // an operand beyond binary16's range, Inf or NaN: the trainer lowers its loss scale

    // ---- dK, dV blocks of the wave: lane = column d, registers = keys
    {
        const float unk = __uint_as_float((uint32_t)(127 - e_cur) << 23), unv = 1.0f / P_SCALE;
        float* __restrict__ dkc = dk + ((int64_t)c * CM + 32 * wave) * lddk + h * CDH + lr;
        float* __restrict__ dvc = dv + ((int64_t)c * CM + 32 * wave) * lddv + h * CDH + lr;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 8 * i + 4 * lh + r;
                    dkc[(int64_t)key * lddk + 32 * t] = dkacc[t][4 * i + r] * unk;
                    dvc[(int64_t)key * lddv + 32 * t] = dvacc[t][4 * i + r] * unv;
                }
    }
}

constexpr int BWD16_LDS_BYTES = (4 * CTQ * CPH + 4 * CDH * CPT + 2 * CTQ * CPH + 4 * 4 * 32 * CPP) * 2 + (4 * 3 * 32 + 4) * 4;
PerDeviceOnce g_bwd16_once;

}  // namespace

// called by ogmm_attention_bwd_f16x3 (train_attn_bwd.hip) for level 2
int ogmm_attention_bwd16_launch(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* dout, int64_t lddo,
                                int C, int N, int H, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                                int* overflow, void* stream) {
    if (g_bwd16_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BWD16_LDS_BYTES);
    hipLaunchKernelGGL(attention_bwd16_kernel, dim3(C * H), dim3(256), BWD16_LDS_BYTES, ogmm::as_stream(stream), q, ldq, k, ldk, v, ldv, dout, lddo, N, H,
                       scale, dq, lddq, dk, lddk, dv, lddv, overflow);
    return ogmm::check_launch("ogmm_attention_bwd_f16x3");
}
