// K2+K3 fused: the whole DGCNN EdgeConv chain of models/dgcnn.py:135-150 for a tile of points in one workgroup:
//   gather neighbours -> cat(x_j - x_i, x_i) -> conv1(6->64)+BN+ReLU -> conv2(64->64) -> conv3(64->128) -> conv4(128->256),
//   each +BN+ReLU, with the max over the k edges of every point taken after each layer (x1|x2|x3|x4 -> xcat[point][512]).
// The per-edge tensors [C*N*k][64|64|128] (2.7 GB at B=64) never leave the chip: layer l's activated output is written
// -- already split into binary16 hi/lo planes in A-operand order -- into LDS and consumed by layer l+1.
//
// PERSISTENT: one 8-wave workgroup per CU (the planes fill its LDS) walks tiles blockIdx.x, + gridDim.x, ...  Everything
// that does not depend on the tile stays in registers across tiles -- the fragment-major weight images of layers 2-4
// (128 VGPRs), the folded BatchNorm scale / shift of the wave's columns, layer 1's weight row -- and the neighbour gather of
// tile t+1 (two dependent global loads) is in flight while tile t computes.  A one-tile-per-workgroup version of this kernel
// spent 25 % of its time waiting on those loads at the top of every tile, and 45 % outside the MFMA layers and their epilogues.
//
// Tile = P = floor(160 / k) points = P*k <= 160 edge rows (5 MFMA row blocks).
//   layer 1: VALU (K = 6); a wave owns a point at a time, lane = channel, running max over the point's k edges in a register
//   layers 2-4: fp16x3 split MFMA (v_mfma_f32_32x32x16_f16, see gemm_f16x3.hip); the (row block, 32-column block) grid of a
//   layer (5 x 2, 5 x 4, 5 x 8) is dealt over the 8 waves; the pooled maxima are collected per (point, column) with LDS integer
//   atomicMax (post-ReLU values are >= 0) in one of two scratch buffers, so that writing a layer's pooled map to HBM needs no barrier
//   of its own.  Five barriers per tile.
// LDS: region A = h1 planes, later h3 planes (87 KB); region B = h2 planes (46 KB); edge features 4 KB; pool scratch 2 x P x 256 ints
// (one buffer and two more barriers per tile when two do not fit: k < 13).
#include "ogmm_common.h"
#include <algorithm>
#include <cstdlib>

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

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

// split of two values at once: packed conversions (v_cvt_pk_f16_f32) and a packed subtract instead of five scalar ops per value
__device__ __forceinline__ void split_h2(float a, float b, f16x2& hi, f16x2& lo) {
    f32x2 x = {__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f)};
    hi = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(hi, f32x2);
    lo = __builtin_convertvector(r, f16x2);
}

// lanes 2j and 2j+1 hold (v0, v1) of columns c and c+1: afterwards the even lane has (v0 of c, v0 of c+1), the odd lane (v1 of c, v1 of c+1)
__device__ __forceinline__ void trade_pair(float v0, float v1, bool odd, float& left, float& right) {
    const float n0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v0), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    const float n1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), 0xB1, 0xf, 0xf, true));
    left = odd ? n1 : v0;
    right = odd ? v1 : n0;
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
// KC > 0: k is the compile-time constant KC and rb0 == RB0; for a full tile every accumulator register's point (row / KC) is then known
// when the epilogue is unrolled, and the pooling costs one v_max per element plus an atomic where a lane half crosses into the next
// point, instead of a multiply / shift / compare / branch per element.
template <int KS, int NB, int KC = 0, int RB0 = 0>
__device__ __forceinline__ void mfma_layer(const _Float16* Ain, int LDA, const f16x8 (&wb)[KS][2], int nb, int rb0, float sc, float sh,
                                           _Float16* Aout, int LDO,
                                           int* pool_s, int pool_ld, unsigned inv_k16, int rows_valid, int lane) {
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int APL = ROWS * LDA;
    // A fragments one k-step ahead of their MFMAs; the scheduling barriers keep the compiler from hoisting every step's LDS reads to the
    // top (the weight images already hold 128 of the 256 registers)
    f16x8 ah[2][NB], al[2][NB];
    auto load_a = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int off = ((rb0 + i) * 32 + lr) * LDA + s * 16 + lh * 8;
            ah[buf][i] = *reinterpret_cast<const f16x8*>(&Ain[off]);
            al[buf][i] = *reinterpret_cast<const f16x8*>(&Ain[APL + off]);
        }
    };
    load_a(0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int cur = s & 1;
        if (s + 1 < KS) load_a(s + 1, cur ^ 1);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], wb[s][0], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], wb[s][1], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], wb[s][0], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int col = nb * 32 + lr;
    // ---- activation in place
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(fmaf(acc[i][r], sc, sh), 0.0f);
    // ---- next layer's A planes.  A lane holds one column of rows (r, r+1); neighbouring lanes trade one of the two (DPP quad swap) so
    // that the even lane owns row r and the odd lane row r+1 of the column pair, and each writes both halfs with one ds_write_b32 per
    // plane: half the LDS write instructions of a b16 store per element, which were the bottleneck of this epilogue.
    if (Aout) {
        const int OPL = ROWS * LDO;
        const bool odd = lane & 1;
        _Float16* __restrict__ dst = Aout + ((rb0 * 32 + 4 * lh + (odd ? 1 : 0)) * LDO + (col & ~1));
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float left, right;
                trade_pair(acc[i][2 * q], acc[i][2 * q + 1], odd, left, right);           // columns (col & ~1), (col & ~1) + 1
                f16x2 hi2, lo2;
                split_h2(left, right, hi2, lo2);
                const int off = (i * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2)) * LDO;      // row of register 2q, relative to the lane's base row
                *reinterpret_cast<f16x2*>(&dst[off]) = hi2;
                *reinterpret_cast<f16x2*>(&dst[OPL + off]) = lo2;
            }
    }
    // ---- pooling
    if (KC > 0 && rows_valid == (ROWS / (KC > 0 ? KC : 1)) * KC) {
        constexpr int KD = KC > 0 ? KC : 1, FULL = (ROWS / KD) * KD;
        int g_lo = -1, g_hi = -1;                         // current point of the lower / upper lane half: constants once unrolled
        float cur = 0.0f;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowc = (RB0 + i) * 32 + (r & 3) + 8 * (r >> 2);          // row of the lower half; the upper half is 4 further
                const float v = acc[i][r];
                const int n_lo = rowc < FULL ? rowc / KD : -2, n_hi = rowc + 4 < FULL ? (rowc + 4) / KD : -2;   // -2: padding row
                const bool new_lo = n_lo != g_lo, new_hi = n_hi != g_hi;
                if (!new_lo && !new_hi) {
                    if (n_lo >= 0 || n_hi >= 0) cur = fmaxf(cur, v);              // (both halves inside their current point, or padding)
                } else {
                    const bool mine_new = lh ? new_hi : new_lo;
                    const int prev = lh ? g_hi : g_lo, next = lh ? n_hi : n_lo;
                    if (mine_new) {
                        if (prev >= 0) atomicMax(&pool_s[prev * pool_ld + col], __float_as_int(cur));
                        cur = v;
                    } else if (next >= 0) {
                        cur = fmaxf(cur, v);
                    }
                }
                g_lo = n_lo;
                g_hi = n_hi;
            }
        const int last = lh ? g_hi : g_lo;
        if (last >= 0) atomicMax(&pool_s[last * pool_ld + col], __float_as_int(cur));
        return;
    }
    int cur_group = -1;
    float cur_max = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (rb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = acc[i][r];
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

// A use the compiler cannot see through: makes it place the s_waitcnt for a load HERE.  The kernel's loads are waited for at points
// where everything older has long arrived; left to itself the compiler waits at first use, after younger conditional stores have been
// issued, where the only safe count is vmcnt(0) -- the HBM round trip of a store that was issued a moment ago.
template <typename T>
__device__ __forceinline__ void touch(const T& x) { asm volatile("" ::"v"(x) : "memory"); }
template <int KS>
__device__ __forceinline__ void touch_weights(const f16x8 (&wb)[KS][2]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) { touch(wb[s][0]); touch(wb[s][1]); }
}

struct EdgeW {
    const float* W1; const float* s1; const float* t1;                                   // [64][6], [64], [64]
    const void* h2; const void* l2; const float* s2; const float* t2; float inv2;        // fragment images + folded BN
    const void* h3; const void* l3; const float* s3; const float* t3; float inv3;
    const void* h4; const void* l4; const float* s4; const float* t4; float inv4;
};

// phase probe (OGMM_EDGECONV_PROBE=1, tools/edgeconv_time.py): thread 0 of every workgroup adds the shader cycles between the tile's barriers
// {setup -> (1), layer 1 -> (2), layer 2 -> (3), layer 3 -> (4), layer 4 -> (5), flush + loop end} and the number of tiles to a device array
__device__ unsigned long long g_edgeconv_probe[8];

template <int KC, bool PROBE = false>
__global__ __launch_bounds__(512) void edgeconv_fused_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int N, int k,
                                                             int64_t total_pts, int P, int64_t n_tiles, int two_pools, const EdgeW w,
                                                             float* __restrict__ xcat, int64_t ldx) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16* regA = lds;                               // h1 planes [2][160][72]  -> later h3 planes [2][160][136]
    _Float16* regB = lds + 2 * ROWS * LD128;            // h2 planes [2][160][72]
    float4* ef4 = reinterpret_cast<float4*>(regB + 2 * ROWS * LD64);   // [160] (x_j - x_i, 0) per edge row, then [P <= 24] centres x_i
    float4* ctr4 = ef4 + ROWS;
    int* pool0 = reinterpret_cast<int*>(ef4 + ROWS + 24);              // [P][256]
    int* pool1 = two_pools ? pool0 + P * 256 : pool0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned inv_k16 = (65536u + (unsigned)k - 1u) / (unsigned)k;     // (row * inv_k16) >> 16 == row / k on this range

    // ---- tile-independent state, loaded once
    f16x8 wb4[8][2];                                     // layer 4's image (64 VGPRs) stays; layers 2 and 3 (32 each) are re-fetched from L2
    load_weights<8>(wb4, w.h4, w.l4, wave, lane);        // per tile, a whole layer ahead of their use -- all three resident would spill
    const int lr = lane & 31;
    const float sc2 = w.s2[(wave & 1) * 32 + lr] * w.inv2, sh2 = w.t2[(wave & 1) * 32 + lr];
    const float sc3 = w.s3[(wave & 3) * 32 + lr] * w.inv3, sh3 = w.t3[(wave & 3) * 32 + lr];
    const float sc4 = w.s4[wave * 32 + lr] * w.inv4, sh4 = w.t4[wave * 32 + lr];
    float wv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wv[i] = w.W1[lane * 6 + i];
    const float s1 = w.s1[lane], t1 = w.t1[lane];
    for (int i = tid; i < (two_pools ? 2 : 1) * P * 256; i += 512) pool0[i] = 0;

    // ---- the gather of an edge row: thread e < 160 fetches (x_j - x_i, x_i); two dependent loads, issued a tile ahead
    const int e_pt = tid / k, e_nb = tid % k;            // this thread's (point of the tile, neighbour slot) as an edge row
    // (both lambdas only ISSUE loads: the loaded values are first used -- index arithmetic, x_j - x_i -- a phase later, so no wait lands
    // right behind the load)
    auto edge_index = [&](int64_t tile) -> int {          // neighbour slot's point index within its cloud, or -1 for a padding row
        const int64_t p = tile * P + e_pt;
        if (tid >= ROWS || e_pt >= P || tile >= n_tiles || p >= total_pts) return -1;
        return idx[p * k + e_nb];
    };
    float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // raw x_j, x_i of this thread's edge row
    auto edge_fetch = [&](int64_t tile, int jn) {
#pragma unroll
        for (int i = 0; i < 6; ++i) f[i] = 0.0f;
        if (jn >= 0) {
            const int64_t p = tile * P + e_pt;
            const int64_t j = (p / N) * N + jn;
            f[3] = xyz[3 * p]; f[4] = xyz[3 * p + 1]; f[5] = xyz[3 * p + 2];
            f[0] = xyz[3 * j]; f[1] = xyz[3 * j + 1]; f[2] = xyz[3 * j + 2];
        }
    };
    edge_fetch(blockIdx.x, edge_index(blockIdx.x));
    touch_weights<8>(wb4);
    touch(sc2); touch(sh2); touch(sc3); touch(sh3); touch(sc4); touch(sh4); touch(s1); touch(t1);
#pragma unroll
    for (int i = 0; i < 6; ++i) { touch(wv[i]); touch(f[i]); }

    // Pooled maxima of a layer -> xcat, cells re-zeroed.  With k fixed at compile time every thread does this at most once: a loop
    // with a run-time trip count would hide the number of stores in flight from the compiler's s_waitcnt bookkeeping, and the next
    // wait on a LOAD (weights, gather) would degrade to vmcnt(0), i.e. sit out the HBM round trip of these stores.
    constexpr bool ONE_PASS = KC > 0 && (ROWS / (KC > 0 ? KC : 1)) <= 8;
    auto flush = [&](int* pool, int shift /* log2(columns / 4) */, int base, int pts, int64_t p0) {
        auto cell_out = [&](int i) {
            const int p = i >> shift, ch = (i & ((1 << shift) - 1)) * 4;
            int4* cell = reinterpret_cast<int4*>(&pool[p * 256 + ch]);
            const int4 v = *cell;
            *reinterpret_cast<float4*>(&xcat[(p0 + p) * ldx + base + ch]) =
                make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
            *cell = make_int4(0, 0, 0, 0);
        };
        if constexpr (ONE_PASS) {
            if (tid < (pts << shift)) cell_out(tid);
        } else {
            for (int i = tid; i < (pts << shift); i += 512) cell_out(i);
        }
    };

    long long pc = 0;
    auto probe = [&](int slot) {
        if (PROBE && tid == 0) {
            const long long now = clock64();
            atomicAdd(&g_edgeconv_probe[slot], (unsigned long long)(now - pc));
            pc = now;
        }
    };
    if (PROBE && tid == 0) pc = clock64();
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t p0 = tile * P;
        const int pts = (int)min((int64_t)P, total_pts - p0);
        const int rows_valid = pts * k;
        // opaque copy: otherwise the row -> point map of every accumulator register (tile-invariant) is hoisted out of the tile loop
        // into ~200 registers and the kernel spills
        unsigned ik16 = inv_k16;
        asm volatile("" : "+s"(ik16));
        int lane_t = lane;                                               // same for the per-register LDS addresses
        asm volatile("" : "+v"(lane_t));
        if (tid < ROWS) {
            ef4[tid] = make_float4(f[0] - f[3], f[1] - f[4], f[2] - f[5], 0.0f);
            if (e_nb == 0 && e_pt < P) ctr4[e_pt] = make_float4(f[3], f[4], f[5], 0.0f);
        }
        const int j_next = edge_index(tile + gridDim.x);                // the index load travels during layer 1
        f16x8 wb2[4][2], wb3[4][2];
        load_weights<4>(wb2, w.h2, w.l2, wave & 1, lane);
        load_weights<4>(wb3, w.h3, w.l3, wave & 3, lane);
        __syncthreads();                                                 // (1) edge features visible; the previous tile is finished
        probe(0);

        // ---- layer 1 (VALU): wave -> point, lane -> channel; the centre term is constant over a point's edges
        // two edges per step: neighbouring lanes trade one of the two values so that each writes a (column pair) of one row with a
        // single ds_write_b32 per plane, as in the MFMA layers' epilogue
        const bool odd = lane & 1;
        const int kk = KC > 0 ? KC : k;
        float mx1 = 0.0f;
        for (int pt = wave; pt < pts; pt += 8) {                        // (a single pass when ONE_PASS: P <= 8)
            const int e0 = pt * kk;
            const float4 c = ctr4[pt];
            const float ctr = fmaf(wv[5], c.z, fmaf(wv[4], c.y, wv[3] * c.x));
            float mx = 0.0f;
            _Float16* __restrict__ dst = regA + (e0 + (odd ? 1 : 0)) * LD64 + (lane & ~1);
#pragma unroll
            for (int q = 0; q < (KC > 0 ? (KC + 1) / 2 : 16); ++q) {
                if (KC == 0 && 2 * q >= kk) break;
                const bool second = 2 * q + 1 < kk;                      // an odd k leaves the last pair half empty
                const float4 a0 = ef4[e0 + 2 * q], a1 = ef4[e0 + 2 * q + (second ? 1 : 0)];
                const float v0 = fmaxf(fmaf(fmaf(wv[2], a0.z, fmaf(wv[1], a0.y, wv[0] * a0.x)) + ctr, s1, t1), 0.0f);
                const float v1 = second ? fmaxf(fmaf(fmaf(wv[2], a1.z, fmaf(wv[1], a1.y, wv[0] * a1.x)) + ctr, s1, t1), 0.0f) : 0.0f;
                mx = fmaxf(mx, fmaxf(v0, v1));
                float left, right;
                trade_pair(v0, v1, odd, left, right);
                f16x2 hi2, lo2;
                split_h2(left, right, hi2, lo2);
                if (!odd || second) {
                    *reinterpret_cast<f16x2*>(&dst[2 * q * LD64]) = hi2;
                    *reinterpret_cast<f16x2*>(&dst[ROWS * LD64 + 2 * q * LD64]) = lo2;
                }
            }
            if constexpr (ONE_PASS) {
                mx1 = mx;
                break;
            }
            xcat[(p0 + pt) * ldx + lane] = mx;                           // x1 = max over the point's k edges
        }
        touch_weights<4>(wb2);                                           // fetched at the top of the tile: arrived during layer 1
        touch_weights<4>(wb3);
        if constexpr (ONE_PASS) {
            if (wave < pts) xcat[(p0 + wave) * ldx + lane] = mx1;
        }
        edge_fetch(tile + gridDim.x, j_next);                            // next tile's coordinates travel during layers 2-4
        __syncthreads();                                                 // (2) h1 planes complete
        probe(1);

        // ---- layer 2: 64 -> 64; 2 column blocks x 5 row blocks over 8 waves: waves 0-1 take row blocks {0,1}, waves 2-7 one of {2,3,4}
        if (wave < 2) mfma_layer<4, 2, KC, 0>(regA, LD64, wb2, wave & 1, 0, sc2, sh2, regB, LD64, pool0, 256, ik16, rows_valid, lane_t);
        else if (wave < 4) mfma_layer<4, 1, KC, 2>(regA, LD64, wb2, wave & 1, 2, sc2, sh2, regB, LD64, pool0, 256, ik16, rows_valid, lane_t);
        else if (wave < 6) mfma_layer<4, 1, KC, 3>(regA, LD64, wb2, wave & 1, 3, sc2, sh2, regB, LD64, pool0, 256, ik16, rows_valid, lane_t);
        else mfma_layer<4, 1, KC, 4>(regA, LD64, wb2, wave & 1, 4, sc2, sh2, regB, LD64, pool0, 256, ik16, rows_valid, lane_t);
        __syncthreads();                                                 // (3) h2 planes and x2 maxima complete
        probe(2);
        flush(pool0, 4, 64, pts, p0);
        if (!two_pools) __syncthreads();

        // ---- layer 3: 64 -> 128; 4 column blocks x 5 row blocks: waves 0-3 take row blocks {0,1,2}, waves 4-7 {3,4}; h3 planes
        // overwrite region A (h1 is dead)
        if (wave < 4) mfma_layer<4, 3, KC, 0>(regB, LD64, wb3, wave & 3, 0, sc3, sh3, regA, LD128, pool1, 256, ik16, rows_valid, lane_t);
        else mfma_layer<4, 2, KC, 3>(regB, LD64, wb3, wave & 3, 3, sc3, sh3, regA, LD128, pool1, 256, ik16, rows_valid, lane_t);
        __syncthreads();                                                 // (4)
        probe(3);
        flush(pool1, 5, 128, pts, p0);
        if (!two_pools) __syncthreads();

        // ---- layer 4: 128 -> 256, every wave 32 columns x 5 row blocks; only the pooled output is needed
        // (two passes over the row blocks: 5 accumulator tiles next to the 128 weight registers would spill)
        mfma_layer<8, 3, KC, 0>(regA, LD128, wb4, wave, 0, sc4, sh4, nullptr, 0, pool0, 256, ik16, rows_valid, lane_t);
        mfma_layer<8, 2, KC, 3>(regA, LD128, wb4, wave, 3, sc4, sh4, nullptr, 0, pool0, 256, ik16, rows_valid, lane_t);
        __syncthreads();                                                 // (5)
        probe(4);
#pragma unroll
        for (int i = 0; i < 6; ++i) touch(f[i]);                         // next tile's edge features: in flight since layer 1
        flush(pool0, 6, 256, pts, p0);
        // pool0 is next touched by layer 2 of the following tile, two barriers from here
        if (PROBE && tid == 0) atomicAdd(&g_edgeconv_probe[6], 1ull);
    }
}

}  // namespace

extern "C" int ogmm_edgeconv_fused(const float* xyz, const int32_t* idx, int C, int N, int k, const float* W1, const float* s1, const float* t1,
                                   const void* h2, const void* l2, const float* s2, const float* t2, float inv2, const void* h3,
                                   const void* l3, const float* s3, const float* t3, float inv3, const void* h4, const void* l4,
                                   const float* s4, const float* t4, float inv4, float* xcat, int64_t ldx, void* stream) {
    OGMM_REQUIRE(xyz && idx && W1 && s1 && t1 && h2 && l2 && s2 && t2 && h3 && l3 && s3 && t3 && h4 && l4 && s4 && t4 && xcat,
                 "ogmm_edgeconv_fused: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && k >= 7 && k <= 32 && ldx >= 512 && ldx % 4 == 0 && reinterpret_cast<uintptr_t>(xcat) % 16 == 0, "ogmm_edgeconv_fused: bad sizes C=%d N=%d k=%d ldx=%lld", C, N, k, (long long)ldx);
    const int64_t total = (int64_t)C * N;
    const int P = ROWS / k;
    const size_t fixed = (size_t)(2 * ROWS * LD128 + 2 * ROWS * LD64) * sizeof(_Float16) + (size_t)(ROWS + 24) * sizeof(float4);
    const size_t pool = (size_t)P * 256 * sizeof(int);
    const int two_pools = fixed + 2 * pool <= 160 * 1024;
    const size_t lds = fixed + (two_pools ? 2 : 1) * pool;
    OGMM_REQUIRE(lds <= 160 * 1024, "ogmm_edgeconv_fused: LDS budget exceeded");
    static ogmm::PerDeviceOnce once;          // attributes and the CU count are per device
    static int n_cu_of[64] = {};
    const int dev = once.device();
    if (once.first() || dev < 0 || dev >= 64 || !n_cu_of[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_fused_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_fused_kernel<20>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (dev >= 0 && dev < 64) n_cu_of[dev] = n;
    }
    const int n_cu = (dev >= 0 && dev < 64 && n_cu_of[dev]) ? n_cu_of[dev] : 256;
    EdgeW w{W1, s1, t1, h2, l2, s2, t2, inv2, h3, l3, s3, t3, inv3, h4, l4, s4, t4, inv4};
    const int64_t n_tiles = (total + P - 1) / P;
    const unsigned blocks = (unsigned)std::min<int64_t>(n_tiles, n_cu);          // one persistent workgroup per CU
    static const bool probe = [] { const char* e = getenv("OGMM_EDGECONV_PROBE"); return e && e[0] == '1'; }();
    if (probe && k == 20) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edgeconv_fused_kernel<20, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((edgeconv_fused_kernel<20, true>), dim3(blocks), dim3(512), lds, ogmm::as_stream(stream), xyz, idx, N, k, total, P, n_tiles, two_pools, w, xcat, ldx);
    } else
    if (k == 20)          // the reference's gnn_k: pooling specialised at compile time
        hipLaunchKernelGGL(edgeconv_fused_kernel<20>, dim3(blocks), dim3(512), lds, ogmm::as_stream(stream), xyz, idx, N, k, total, P, n_tiles, two_pools,
                           w, xcat, ldx);
    else
        hipLaunchKernelGGL(edgeconv_fused_kernel<0>, dim3(blocks), dim3(512), lds, ogmm::as_stream(stream), xyz, idx, N, k, total, P, n_tiles, two_pools,
                           w, xcat, ldx);
    return ogmm::check_launch("ogmm_edgeconv_fused");
}

// diagnostic (tools/edgeconv_time.py): read and clear the phase probe
extern "C" int ogmm_debug_edgeconv_probe(unsigned long long* host8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_edgeconv_probe), sizeof(z)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_edgeconv_probe), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
