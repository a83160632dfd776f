// K1 kNN graph, K5 farthest-point sampling, K6 row gather.
//
// Both selection kernels reproduce the reference's fp32 distance VALUES bit-for-bit (see the oracle
// header for the probed rounding sequences), because 1-ulp differences flip neighbour sets / FPS chains
// and every later feature depends on them (SURVEY.md section 7 "hard parts").
#include <cstdlib>
#include "ogmm_common.h"
#include "torch_topk_select.h"

#ifndef OGMM_KNN_CHUNK
#define OGMM_KNN_CHUNK 32
#endif

namespace {

using namespace ogmm;

// ------------------------------------------------------------------------------------------------
// kNN: one thread per query point, the whole cloud (x,y,z,|p|^2) resident in LDS, candidates are
// visited in index order and kept in a sorted register list (strict '<' => ties keep the lower index); rows whose
// k-th and (k+1)-th distances tie exactly are re-done with torch.topk's own (libstdc++) selection, see below.
// HBM traffic is the compulsory 12 B/point in + 4k B/point out; the N x N distance matrix of
// lib/utils.py:28-33 is never materialised.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float knn_dist(const float4 pq, const float4 pj) {
    // torch.matmul with K=3: fma(z,z', fma(y,y', x*x'));  then -2*., + |q|^2, + |p_j|^2, clamp(1e-12)
    const float dot = __fmaf_rn(pq.z, pj.z, __fmaf_rn(pq.y, pj.y, mul_rn(pq.x, pj.x)));
    return fmaxf(add_rn(add_rn(mul_rn(-2.0f, dot), pq.w), pj.w), 1e-12f);
}

// KL = list length = largest supported k + 1: the extra slot holds the (k+1)-th smallest distance, which tells
// whether rank k is an exact tie (then the row is marked by a negative first index and re-done by
// knn_resolve_ties_kernel with torch.topk's own selection algorithm).
template <int KL>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz, int N, int k, int32_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float4 pts[];   // [N]
    const int c = blockIdx.y;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const float x = cloud[3 * j], y = cloud[3 * j + 1], z = cloud[3 * j + 2];
        pts[j] = make_float4(x, y, z, sqnorm3(x, y, z));
    }
    __syncthreads();
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const float4 pq = pts[q];
    float dk[KL];
    int ik[KL];
#pragma unroll
    for (int p = 0; p < KL; ++p) { dk[p] = __builtin_inff(); ik[p] = 0; }
    // branch-free stable insertion (selects only; hipcc turns the if/else-if ladder into ~20 branches per insertion);
    // the only branch left is the wave-level "does any lane insert" test
    auto insert = [&](float d, int j) {
        if (d < dk[KL - 1]) {
#pragma unroll
            for (int p = KL - 1; p > 0; --p) {
                const bool shift = d < dk[p - 1];
                const bool here = !shift && d < dk[p];
                dk[p] = shift ? dk[p - 1] : (here ? d : dk[p]);
                ik[p] = shift ? ik[p - 1] : (here ? j : ik[p]);
            }
            const bool first = d < dk[0];
            dk[0] = first ? d : dk[0];
            ik[0] = first ? j : ik[0];
        }
    };
    // Candidates in chunks of 32.  Pass 1 only tests each distance against the list's current worst entry and records a bit; pass 2
    // drains the bits: every lane takes its lowest marked candidate (so a lane's insertions stay in index order), recomputes the
    // distance (same operations, same value) and runs the insertion ladder.  The ladder (~6 selects per list slot) is executed by
    // the whole wave whenever ANY lane inserts; per candidate that is almost always the case (64 lanes x ~k ln(N/k) / N insertions
    // each), per drain round it happens max-over-lanes(marks) times per chunk: ~4x fewer ladder passes at N = 1024, k = 20.
    int j = 0;
    constexpr int CHK = OGMM_KNN_CHUNK;
    for (; j + CHK <= N; j += CHK) {
        const float worst = dk[KL - 1];
        unsigned mask = 0u;
#pragma unroll
        for (int t = 0; t < CHK; t += 4) {
            const float4 p0 = pts[j + t], p1 = pts[j + t + 1], p2 = pts[j + t + 2], p3 = pts[j + t + 3];
            mask |= (knn_dist(pq, p0) < worst ? 1u : 0u) << t;
            mask |= (knn_dist(pq, p1) < worst ? 1u : 0u) << (t + 1);
            mask |= (knn_dist(pq, p2) < worst ? 1u : 0u) << (t + 2);
            mask |= (knn_dist(pq, p3) < worst ? 1u : 0u) << (t + 3);
        }
        while (__any(mask != 0u)) {
            const bool mine = mask != 0u;
            const int b = mine ? __ffs(mask) - 1 : 0;
            const float d = mine ? knn_dist(pq, pts[j + b]) : __builtin_inff();
            insert(d, j + b);
            mask &= mask - 1u;
        }
    }
    for (; j < N; ++j) insert(knn_dist(pq, pts[j]), j);
    float d_last = 0.0f, d_next = -1.0f;
#pragma unroll
    for (int p = 0; p < KL; ++p) {
        if (p == k - 1) d_last = dk[p];
        if (p == k) d_next = dk[p];
    }
    const bool boundary_tie = k < N && d_last == d_next;
    int32_t* out = idx + ((int64_t)c * N + q) * k;
#pragma unroll
    for (int p = 0; p < KL - 1; ++p)
        if (p < k) out[p] = (p == 0 && boundary_tie) ? ~ik[0] : ik[p];
}

// The same result in two scans (the default while the candidate buffers fit next to the cloud): the insertion ladder above costs ~6 selects per
// list slot because the indices travel with the distances, and the whole wave runs it max-over-lanes(marks) times per chunk -- ~220 times per
// query wave at N = 1024, k = 20, 60 % of the kernel's instructions.
//   scan A keeps only the k+1 smallest DISTANCES, sorted: inserting d is one v_med3_f32 per slot (new d[p] = median(d[p-1], d, d[p]));
//   scan B knows tau = the k-th smallest distance and appends, in index order, the candidates with d < tau (at most k-1) and the first k
//          with d == tau to a per-thread list in LDS; only those go through the full ladder -- k to k+few passes per wave instead of ~220.
// The first k entries are the lexicographically smallest (distance, index) pairs in both forms; the rank-k tie flag comes from scan A.
// The scans are bound by vector instructions per candidate, so they use the shortest exact forms: -2 dot + |q|^2 as ONE fma (the product by 2 is
// exact, so fma(-2, dot, |q|^2) rounds like the reference's mul-then-add), no clamp where only the comparison matters (a clamped distance is
// 1e-12 <= any threshold), and the per-lane bit mask of a chunk is built by v_cmp + v_addc (mask = 2 mask + (d < t)): candidate t of the chunk
// ends up in bit 31 - t, so the lowest index is the highest bit.
__device__ __forceinline__ float knn_dist_raw(const float4 pq, const float4 pj) {
    const float dot = __fmaf_rn(pq.z, pj.z, __fmaf_rn(pq.y, pj.y, mul_rn(pq.x, pj.x)));
    return add_rn(__fmaf_rn(-2.0f, dot, pq.w), pj.w);
}
__device__ __forceinline__ void mark_lt(unsigned& m, float d, float thr) {
    asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d), "v"(thr) : "vcc");
}
__device__ __forceinline__ void mark_le(unsigned& m, float d, float thr) {
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d), "v"(thr) : "vcc");
}

// rank-k ties of this workgroup's rows, resolved in place (defined behind block_introselect)
__device__ void resolve_ties_in_block(const float4* pts, int N, int k, int q0, int32_t* idx_cloud, void* scratch, bool have_lists);

template <int KL>
__global__ __launch_bounds__(256) void knn2_kernel(const float* __restrict__ xyz, int N, int k, int32_t* __restrict__ idx, int fold_ties) {
    static_assert(OGMM_KNN_CHUNK == 32, "the chunk is one 32-bit mask");
    extern __shared__ __attribute__((aligned(16))) float4 pts[];   // [N], then the candidate lists [2 (KL-1)][256] int16
    short* __restrict__ buf = reinterpret_cast<short*>(pts + N) + threadIdx.x;          // entry i of this thread at buf[i * 256]
    const int c = blockIdx.y;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const float x = cloud[3 * j], y = cloud[3 * j + 1], z = cloud[3 * j + 2];
        pts[j] = make_float4(x, y, z, sqnorm3(x, y, z));
    }
    __syncthreads();
    const int q_raw = blockIdx.x * blockDim.x + threadIdx.x;
    if (q_raw >= N && !fold_ties) return;
    const bool live = q_raw < N;          // (with the tie resolution folded in, every thread stays for the workgroup barriers behind the scans)
    const int q = live ? q_raw : N - 1;
    const float4 pq = pts[q];
    constexpr int CHK = 32;
    // ---- scan A: the k+1 smallest distances
    float dk[KL];
#pragma unroll
    for (int p = 0; p < KL; ++p) dk[p] = __builtin_inff();
    auto insert_d = [&](float d) {
#pragma unroll
        for (int p = KL - 1; p > 0; --p) dk[p] = __builtin_amdgcn_fmed3f(dk[p - 1], d, dk[p]);
        dk[0] = fminf(dk[0], d);
    };
    // The chunk's 32 candidates are in registers before they are used: the next chunk's broadcast reads (one ds_read_b128 per candidate, the same
    // address in every lane) are issued while this chunk is processed -- with two waves per SIMD nothing else hides the LDS latency (the compiler's
    // own schedule kept 2-4 reads in flight and waited for each: 160 -> 146 us came from fewer instructions, the rest is this).
    float4 ca[16], cb[16];          // the two halves of a chunk, each requested while the other one is processed
    auto fetch = [&](float4 (&dst)[16], int j0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) dst[t] = pts[j0 + t];          // (past the cloud: the candidate lists' bytes, never used as candidates)
    };
    int j = 0;
    fetch(ca, 0);
    for (; j + CHK <= N; j += CHK) {
        const float worst = dk[KL - 1];
        unsigned mask = 0u;
        fetch(cb, j + 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) mark_lt(mask, knn_dist_raw(pq, ca[t]), worst);
        fetch(ca, j + CHK);
#pragma unroll
        for (int t = 0; t < 16; ++t) mark_lt(mask, knn_dist_raw(pq, cb[t]), worst);
        while (__any(mask != 0u)) {
            const bool mine = mask != 0u;
            const int t = mine ? __clz(mask) : 0;
            insert_d(mine ? knn_dist(pq, pts[j + t]) : __builtin_inff());
            mask &= ~(0x80000000u >> t);
        }
    }
    for (; j < N; ++j) insert_d(knn_dist(pq, pts[j]));
    float tau = 0.0f, d_next = -1.0f;
#pragma unroll
    for (int p = 0; p < KL; ++p) {
        if (p == k - 1) tau = dk[p];
        if (p == k) d_next = dk[p];
    }
    const bool boundary_tie = k < N && tau == d_next;
    // ---- scan B: the candidates that can be among the first k, in index order (tau >= 1e-12, so the unclamped distance compares like the
    // clamped one against it, except that everything below the clamp is EQUAL to tau = 1e-12: both cases below use the clamped value)
    int cnt = 0, ties = 0;
    auto append = [&](int jj, bool less) {
        if (less || ties < k) {
            buf[cnt * 256] = (short)jj;
            ++cnt;
            ties += less ? 0 : 1;
        }
    };
    fetch(ca, 0);
    for (j = 0; j + CHK <= N; j += CHK) {
        unsigned lt = 0u, le = 0u;
        fetch(cb, j + 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float d = knn_dist(pq, ca[t]);
            mark_lt(lt, d, tau);
            mark_le(le, d, tau);
        }
        fetch(ca, j + CHK);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float d = knn_dist(pq, cb[t]);
            mark_lt(lt, d, tau);
            mark_le(le, d, tau);
        }
        while (le != 0u) {          // (per-lane loop: a handful of iterations per chunk, nothing wave-wide inside)
            const int t = __clz(le);
            const unsigned bit = 0x80000000u >> t;
            append(j + t, (lt & bit) != 0u);
            le &= ~bit;
        }
    }
    for (; j < N; ++j) {
        const float d = knn_dist(pq, pts[j]);
        if (d <= tau) append(j, d < tau);
    }
    // ---- the full ladder over the short list (strict '<': equal distances keep their index order)
    float dl[KL];
    int ik[KL];
#pragma unroll
    for (int p = 0; p < KL; ++p) { dl[p] = __builtin_inff(); ik[p] = 0; }
    for (int i = 0; __any(i < cnt); ++i) {
        const bool mine = i < cnt;
        const int jj = mine ? (int)buf[i * 256] : 0;
        const float d = mine ? knn_dist(pq, pts[jj]) : __builtin_inff();
        if (d < dl[KL - 1]) {
#pragma unroll
            for (int p = KL - 1; p > 0; --p) {
                const bool shift = d < dl[p - 1];
                const bool here = !shift && d < dl[p];
                dl[p] = shift ? dl[p - 1] : (here ? d : dl[p]);
                ik[p] = shift ? ik[p - 1] : (here ? jj : ik[p]);
            }
            const bool first = d < dl[0];
            dl[0] = first ? d : dl[0];
            ik[0] = first ? jj : ik[0];
        }
    }
    int32_t* out = idx + ((int64_t)c * N + q) * k;
    if (live) {
#pragma unroll
        for (int p = 0; p < KL - 1; ++p)
            if (p < k) out[p] = (p == 0 && boundary_tie) ? ~ik[0] : ik[p];
    }
    if (!fold_ties) return;
    // ---- rows with an exact tie at rank k (~6e-5 of them): torch.topk's own selection, by this workgroup, in the LDS the candidate lists no longer need
    // (a separate launch for them sat on the forward's critical path in front of the EdgeConv kernel: 47 us + a launch gap for eight rows)
    if (!__syncthreads_or(live && boundary_tie)) return;
    __shared__ int tie_rows[256];
    __shared__ int n_tie;
    if (threadIdx.x == 0) n_tie = 0;
    __syncthreads();
    if (live && boundary_tie) tie_rows[atomicAdd(&n_tie, 1)] = q;
    __syncthreads();
    // (ascending row order: the order the separate kernel processes them in is irrelevant -- rows are independent -- but keep it deterministic)
    const int nt = n_tie;
    int prev = -1;
    for (int f = 0; f < nt; ++f) {
        int row = 0x7fffffff;
        for (int g = 0; g < nt; ++g) { const int r = tie_rows[g]; if (r > prev && r < row) row = r; }          // the next flagged row (same in every thread)
        prev = row;
        resolve_ties_in_block(pts, N, k, row, idx + (int64_t)c * N * k, reinterpret_cast<void*>(pts + N), fold_ties == 2);
    }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the head of the forward as ONE kernel (knn4_kernel).
//  (1) Scan B without a second pass over the cloud.  knn2_kernel's scan B recomputes all N distances to find the <= 2k - 1 candidates with d <= tau:
//      36 % of the kernel's vector instructions.  Scan A already knows which candidates can matter: a candidate of the final set has d <= tau <= the
//      list's worst entry at the time its chunk was scanned, i.e. it was MARKED (raw d < worst; the one exception -- d == worst == tau -- has k + 1
//      earlier candidates at d <= tau in front of it and is never among the first k by (distance, index)).  So scan A keeps its 32-bit mark word per
//      chunk (a coalesced store to a global bitmap, [chunk][thread]: 4 N bytes per query, L2-resident) and scan B walks the ~120 marked candidates
//      of its query in index order instead of all N.  Same lists, same ladder, same flags: the neighbour sets are identical.
//  (2) HEAD: the 5-NN graph of the positional encoding (lib/utils.py:52 <- models/attn.py:69: its own topk call in the reference) is the first five
//      entries of the sorted 20-NN list -- with its OWN rank-5 tie flag (d[4] == d[5]) resolved by torch.topk(k = 5)'s selection, as knn_kernel<9> +
//      the tie pass did -- and the positional front end (models/attn.py:65-73, pos_hidden_kernel) needs nothing but the cloud, its centroid and that
//      graph: both are computed here, from the cloud this workgroup already holds in LDS.  pos_hidden_kernel evaluated a point's geometry (three
//      divisions and a square root per neighbour) once per CHANNEL, 64-fold; here once per point, then the 64 channels expand the six scalars.  The
//      centroid is summed in pos_hidden_kernel's order (thread t: points t, t + 256, ...; wave sums; four partials) and every expression is the same,
//      so hid_dis / hid_ang are bit-identical.  Two kernels (85-165 us and 190 us alone) leave the forward's head, where they competed with this one
//      for the chip in front of the persistent EdgeConv kernel.
struct knn_pos_args {          // HEAD outputs / constants (nullptr idx5 = plain kNN)
    int32_t* idx5;             // [C][N][5]
    const float *w_dis, *s_dis, *t_dis, *w_ang, *s_ang, *t_ang;      // [64] each (models/attn.py:34-57 hidden layers, BatchNorm folded)
    float *hid_dis, *hid_ang;  // [C*N][64]
};

template <int KL, bool HEAD>
__global__ __launch_bounds__(256) void knn4_kernel(const float* __restrict__ xyz, int N, int k, int32_t* __restrict__ idx, unsigned* __restrict__ bitmap,
                                                   int fold_ties, const knn_pos_args pa) {
    extern __shared__ __attribute__((aligned(16))) float4 pts[];   // [N], then the candidate lists [2 (KL-1)][256] int16
    short* __restrict__ buf = reinterpret_cast<short*>(pts + N) + threadIdx.x;          // entry i of this thread at buf[i * 256]
    const int c = blockIdx.y;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const float x = cloud[3 * j], y = cloud[3 * j + 1], z = cloud[3 * j + 2];
        pts[j] = make_float4(x, y, z, sqnorm3(x, y, z));
    }
    __syncthreads();
    const int q_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q_raw < N;          // (every thread stays for the workgroup barriers behind the scans)
    const int q = live ? q_raw : N - 1;
    const float4 pq = pts[q];
    constexpr int CHK = 32;
    const int n_chunks = N / CHK;
    unsigned* __restrict__ bm = bitmap + ((int64_t)c * gridDim.x + blockIdx.x) * (int64_t)n_chunks * 256 + threadIdx.x;          // word ch of this thread at bm[ch * 256]
    // ---- scan A: the k+1 smallest distances; every chunk's mark word is kept
    float dk[KL];
#pragma unroll
    for (int p = 0; p < KL; ++p) dk[p] = __builtin_inff();
    auto insert_d = [&](float d) {
#pragma unroll
        for (int p = KL - 1; p > 0; --p) dk[p] = __builtin_amdgcn_fmed3f(dk[p - 1], d, dk[p]);
        dk[0] = fminf(dk[0], d);
    };
    float4 ca[16], cb[16];          // the two halves of a chunk, each requested while the other one is processed (knn2_kernel)
    auto fetch = [&](float4 (&dst)[16], int j0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) dst[t] = pts[j0 + t];          // (past the cloud: the candidate lists' bytes, never used as candidates)
    };
    int j = 0;
    fetch(ca, 0);
    for (int ch = 0; ch < n_chunks; ++ch, j += CHK) {
        const float worst = dk[KL - 1];
        unsigned mask = 0u;
        fetch(cb, j + 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) mark_lt(mask, knn_dist_raw(pq, ca[t]), worst);
        fetch(ca, j + CHK);
#pragma unroll
        for (int t = 0; t < 16; ++t) mark_lt(mask, knn_dist_raw(pq, cb[t]), worst);
        bm[ch * 256] = mask;
        while (__any(mask != 0u)) {
            const bool mine = mask != 0u;
            const int t = mine ? __clz(mask) : 0;
            insert_d(mine ? knn_dist(pq, pts[j + t]) : __builtin_inff());
            mask &= ~(0x80000000u >> t);
        }
    }
    for (; j < N; ++j) insert_d(knn_dist(pq, pts[j]));
    float tau = 0.0f, d_next = -1.0f;
#pragma unroll
    for (int p = 0; p < KL; ++p) {
        if (p == k - 1) tau = dk[p];
        if (p == k) d_next = dk[p];
    }
    const bool boundary_tie = k < N && tau == d_next;
    // ---- scan B over the marked candidates only, in index order
    int cnt = 0, ties = 0;
    auto append = [&](int jj, bool less) {
        if (less || ties < k) {
            buf[cnt * 256] = (short)jj;
            ++cnt;
            ties += less ? 0 : 1;
        }
    };
    {
        unsigned word = n_chunks > 0 ? bm[0] : 0u;          // (this thread's own stores: program order makes them visible to it)
        for (int ch = 0; ch < n_chunks; ++ch) {
            const unsigned next = ch + 1 < n_chunks ? bm[(ch + 1) * 256] : 0u;
            while (word != 0u) {          // (per-lane loop over this lane's marks)
                const int t = __clz(word);
                const int jj = ch * CHK + t;
                const float d = knn_dist(pq, pts[jj]);
                if (d <= tau) append(jj, d < tau);
                word &= ~(0x80000000u >> t);
            }
            word = next;
        }
        for (j = n_chunks * CHK; j < N; ++j) {
            const float d = knn_dist(pq, pts[j]);
            if (d <= tau) append(j, d < tau);
        }
    }
    // ---- the full ladder over the short list (strict '<': equal distances keep their index order)
    float dl[KL];
    int ik[KL];
#pragma unroll
    for (int p = 0; p < KL; ++p) { dl[p] = __builtin_inff(); ik[p] = 0; }
    for (int i = 0; __any(i < cnt); ++i) {
        const bool mine = i < cnt;
        const int jj = mine ? (int)buf[i * 256] : 0;
        const float d = mine ? knn_dist(pq, pts[jj]) : __builtin_inff();
        if (d < dl[KL - 1]) {
#pragma unroll
            for (int p = KL - 1; p > 0; --p) {
                const bool shift = d < dl[p - 1];
                const bool here = !shift && d < dl[p];
                dl[p] = shift ? dl[p - 1] : (here ? d : dl[p]);
                ik[p] = shift ? ik[p - 1] : (here ? jj : ik[p]);
            }
            const bool first = d < dl[0];
            dl[0] = first ? d : dl[0];
            ik[0] = first ? jj : ik[0];
        }
    }
    int32_t* out = idx + ((int64_t)c * N + q) * k;
    if (live) {
#pragma unroll
        for (int p = 0; p < KL - 1; ++p)
            if (p < k) out[p] = (p == 0 && boundary_tie) ? ~ik[0] : ik[p];
    }
    bool tie5 = false;
    if constexpr (HEAD) {
        tie5 = 5 < N && dl[4] == dl[5];          // torch.topk(k = 5)'s own boundary
        if (live) {
            int32_t* o5 = pa.idx5 + ((int64_t)c * N + q) * 5;
#pragma unroll
            for (int p = 0; p < 5; ++p) o5[p] = ik[p];
        }
    }
    // ---- rows with an exact tie at rank k (and, HEAD, at rank 5): torch.topk's own selection, by this workgroup, in the LDS the lists no longer need
    __shared__ int tie_rows[256];
    __shared__ int n_tie;
    auto resolve = [&](bool flagged, int kk, int32_t* base, bool lists) {
        if (!__syncthreads_or(live && flagged)) return;
        if (threadIdx.x == 0) n_tie = 0;
        __syncthreads();
        if (live && flagged) tie_rows[atomicAdd(&n_tie, 1)] = q;
        __syncthreads();
        const int nt = n_tie;
        int prev = -1;
        for (int f = 0; f < nt; ++f) {          // ascending row order (rows are independent; deterministic all the same)
            int row = 0x7fffffff;
            for (int g = 0; g < nt; ++g) { const int r = tie_rows[g]; if (r > prev && r < row) row = r; }
            prev = row;
            resolve_ties_in_block(pts, N, kk, row, base, reinterpret_cast<void*>(pts + N), lists);
        }
        __syncthreads();
    };
    if (fold_ties) resolve(boundary_tie, k, idx + (int64_t)c * N * k, fold_ties == 2);
    if constexpr (HEAD) {
        resolve(tie5, 5, pa.idx5 + (int64_t)c * N * 5, false);
        // ---- positional front end (pos_hidden_kernel's arithmetic, once per point)
        __shared__ double part[4][3];
        __shared__ float geo[256][6];          // d2, alpha[0..4] of this workgroup's points
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        double sx = 0, sy = 0, sz = 0;
        for (int jj = tid; jj < N; jj += 256) { const float4 p = pts[jj]; sx += p.x; sy += p.y; sz += p.z; }
        sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
        if (lane == 0) { part[wave][0] = sx; part[wave][1] = sy; part[wave][2] = sz; }
        __syncthreads();
        const float fn = (float)N;
        const float cx = (float)(part[0][0] + part[1][0] + part[2][0] + part[3][0]) / fn;
        const float cy = (float)(part[0][1] + part[1][1] + part[2][1] + part[3][1]) / fn;
        const float cz = (float)(part[0][2] + part[1][2] + part[2][2] + part[3][2]) / fn;
        if (live) {
            const float xi = pq.x, yi = pq.y, zi = pq.z;
            const float gx = xi - cx, gy = yi - cy, gz = zi - cz;
            const float d2 = (gx * gx + gy * gy) + gz * gz;
            const float gn = fmaxf(sqrtf(d2), 1e-12f);
            const float ux = gx / gn, uy = gy / gn, uz = gz / gn;
            geo[tid][0] = d2;
            const int32_t* nb = pa.idx5 + ((int64_t)c * N + q) * 5;          // (a resolved row differs from the registers: read what the graph says)
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                const float4 pj = pts[tie5 ? nb[e] : ik[e]];
                const float lx = pj.x - xi, ly = pj.y - yi, lz = pj.z - zi;
                const float ln = fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-12f);
                geo[tid][1 + e] = ((lx / ln) * ux + (ly / ln) * uy) + (lz / ln) * uz;
            }
        }
        __syncthreads();
        // a lane takes four channels of one point, a wave four points per trip: 1 KiB per store instruction instead of 256 B (the two maps are 67 MB)
        const int ch = (lane & 15) * 4, sub = lane >> 4;
        const float4 wd = *reinterpret_cast<const float4*>(pa.w_dis + ch), sd = *reinterpret_cast<const float4*>(pa.s_dis + ch), td = *reinterpret_cast<const float4*>(pa.t_dis + ch);
        const float4 wa = *reinterpret_cast<const float4*>(pa.w_ang + ch), sa = *reinterpret_cast<const float4*>(pa.s_ang + ch), ta = *reinterpret_cast<const float4*>(pa.t_ang + ch);
        auto leaky = [](float v) { return v > 0.0f ? v : 0.2f * v; };
        for (int r = 0; r < 16; ++r) {
            const int pl = wave * 64 + r * 4 + sub, i = blockIdx.x * 256 + pl;
            if (i >= N) continue;
            const int64_t row = (int64_t)c * N + i;
            const float d2 = geo[pl][0];
            *reinterpret_cast<float4*>(pa.hid_dis + row * 64 + ch) =
                make_float4(leaky(fmaf(wd.x * d2, sd.x, td.x)), leaky(fmaf(wd.y * d2, sd.y, td.y)), leaky(fmaf(wd.z * d2, sd.z, td.z)), leaky(fmaf(wd.w * d2, sd.w, td.w)));
            float4 best = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                const float al = geo[pl][1 + e];
                best.x = fmaxf(best.x, leaky(fmaf(wa.x * al, sa.x, ta.x))); best.y = fmaxf(best.y, leaky(fmaf(wa.y * al, sa.y, ta.y)));
                best.z = fmaxf(best.z, leaky(fmaf(wa.z * al, sa.z, ta.z))); best.w = fmaxf(best.w, leaky(fmaf(wa.w * al, sa.w, ta.w)));
            }
            *reinterpret_cast<float4*>(pa.hid_ang + row * 64 + ch) = best;
        }
    }
}

// src, tgt [B][3][N] (the model's input layout, models/gmmreg.py:50) -> xyz [2B][N][3] (src clouds, then tgt clouds): what torch.cat + transpose +
// contiguous made in two launches
__global__ __launch_bounds__(256) void pack_clouds_kernel(const float* __restrict__ src, const float* __restrict__ tgt, int B, int N, float* __restrict__ xyz) {
    const int c = blockIdx.y;
    const float* __restrict__ in = (c < B ? src + (int64_t)c * 3 * N : tgt + (int64_t)(c - B) * 3 * N);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    float* __restrict__ o = xyz + ((int64_t)c * N + j) * 3;
    o[0] = in[j]; o[1] = in[N + j]; o[2] = in[2 * N + j];
}

// ---- std::nth_element's partition, by the whole workgroup, with the element moves of the sequential loop
//     for (;;) { while (q[first] < pivot) ++first;  --last;  while (pivot < q[last]) --last;
//                if (!(first < last)) return first;  swap(q[first], q[last]);  ++first; }
// Until the pointers cross, `first` only ever looks at elements that have not been moved yet, and so does `last`; hence the t-th swap
// exchanges the t-th position from the left holding a value >= pivot (L_t) with the t-th position from the right holding a value <=
// pivot (R_t), for as long as L_t < R_t.  With T such swaps the loop returns the next stop of `first`: L_T, unless it runs into the
// element swap T-1 put at R_{T-1} first.  Ranks come from two workgroup prefix sums, the swaps are independent.  The serial loop on one
// thread (LDS round trip per element) made a flagged row cost ~180 us, on the critical path of the forward.
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    __syncthreads();                                     // wave_tot may still be read from the previous scan
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = wave_tot[w];
        if (w < wave) off += t;
        total += t;
    }
    return off + inc - v;
}

// partition q[first, last) around the value of q[pivot] (pivot outside the range); every thread returns the cut
__device__ int block_partition_around(ogmm_select::Cand* q, int first, int last, int pivot, int* Lpos, int* Rpos, int* wave_tot) {
    const int tid = threadIdx.x;
    const float pv = q[pivot].v;
    const int len = last - first;
    const int per = (len + 255) / 256;                   // consecutive positions per thread
    const int lo = first + tid * per, hi = min(last, lo + per);
    int cl = 0, cr = 0;
    for (int x = lo; x < hi; ++x) {
        const float a = q[x].v;
        cl += !(a < pv);
        cr += !(pv < a);
    }
    int nL, nR;
    int rl = block_exclusive_scan(cl, wave_tot, nL);
    int rr = block_exclusive_scan(cr, wave_tot, nR);
    for (int x = lo; x < hi; ++x) {
        const float a = q[x].v;
        if (!(a < pv)) Lpos[rl++] = x;
        if (!(pv < a)) Rpos[nR - 1 - (rr++)] = x;          // rank from the right
    }
    __syncthreads();
    // T = number of t with L_t < R_t (monotone in t)
    const int m = min(nL, nR);
    int mine = 0;
    for (int t = tid; t < m; t += 256) mine += Lpos[t] < Rpos[t];
    int T;
    (void)block_exclusive_scan(mine, wave_tot, T);
    const int cut = (T < nL && (T == 0 || Lpos[T] < Rpos[T - 1])) ? Lpos[T] : Rpos[T - 1];
    for (int t = tid; t < T; t += 256) {
        const ogmm_select::Cand a = q[Lpos[t]], b = q[Rpos[t]];
        q[Lpos[t]] = b;
        q[Rpos[t]] = a;
    }
    __syncthreads();
    return cut;
}

// std::nth_element(q, q + nth, q + n) as in torch_topk_select.h's introselect, partitions of more than 64 elements by the workgroup
__device__ void block_introselect(ogmm_select::Cand* q, int nth, int n, int* Lpos, int* Rpos, int* wave_tot, int* ctl) {
    const int tid = threadIdx.x;
    int first = 0, last = n;
    int lg = 0;
    for (int m = n; m > 1; m >>= 1) ++lg;
    int depth = 2 * lg;
    while (last - first > 64 && depth > 0) {
        --depth;
        if (tid == 0) ogmm_select::median_to_first(q, first, first + 1, first + (last - first) / 2, last - 1);
        __syncthreads();
        const int cut = block_partition_around(q, first + 1, last, first, Lpos, Rpos, wave_tot);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (tid == 0) ogmm_select::introselect_range(q, nth, first, last, depth);
    __syncthreads();
    (void)ctl;
}

// One flagged row, by the whole workgroup, from the cloud already in LDS (knn2_kernel's tail): the row's N candidates in index order, torch.topk's
// selection (torch_topk_select.h), the kept set written in (distance, index) order.  scratch: N Cand (+ 2 N ints when have_lists).
__device__ void resolve_ties_in_block(const float4* pts, int N, int k, int q, int32_t* idx_cloud, void* scratch, bool have_lists) {
    __shared__ int wave_tot_f[4];
    ogmm_select::Cand* cand = reinterpret_cast<ogmm_select::Cand*>(scratch);
    int* Lpos = reinterpret_cast<int*>(cand + N);
    int* Rpos = Lpos + N;
    const int tid = threadIdx.x;
    const float4 pq = pts[q];
    for (int j = tid; j < N; j += 256) {
        cand[j].v = knn_dist(pq, pts[j]);
        cand[j].i = j;
    }
    __syncthreads();
    if ((long long)k * 64 <= N) {              // torch's heap-select branch: one thread
        if (tid == 0) ogmm_select::heap_select(cand, k, N);
    } else if (have_lists) {
        block_introselect(cand, k - 1, N, Lpos, Rpos, wave_tot_f, nullptr);
    } else {
        if (tid == 0) ogmm_select::introselect(cand, k - 1, N);
    }
    __syncthreads();
    if (tid == 0) {
        for (int a = 1; a < k; ++a) {       // order the kept set by (distance, index)
            const ogmm_select::Cand v = cand[a];
            int b = a;
            while (b > 0 && (v.v < cand[b - 1].v || (v.v == cand[b - 1].v && v.i < cand[b - 1].i))) { cand[b] = cand[b - 1]; --b; }
            cand[b] = v;
        }
        for (int a = 0; a < k; ++a) idx_cloud[(int64_t)q * k + a] = cand[a].i;
    }
    __syncthreads();
}

// Rows marked by knn_kernel: rebuild the row's N candidates in index order and keep what torch.topk keeps
// (torch_topk_select.h); the kept set is then written in (distance, index) order.  ~6e-5 of rows take this path.
__global__ __launch_bounds__(256) void knn_resolve_ties_kernel(const float* __restrict__ xyz, int N, int k, int64_t total_rows,
                                                               int32_t* __restrict__ idx, int have_lists) {
    extern __shared__ __attribute__((aligned(16))) ogmm_select::Cand cand[];   // [N], then int Lpos[N], Rpos[N]
    int* Lpos = reinterpret_cast<int*>(cand + N);
    int* Rpos = Lpos + N;
    __shared__ int flagged[256];
    __shared__ int n_flagged;
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x;
    if (tid == 0) n_flagged = 0;
    __syncthreads();
    const int64_t my_row = (int64_t)blockIdx.x * 256 + tid;
    if (my_row < total_rows && idx[my_row * k] < 0) flagged[atomicAdd(&n_flagged, 1)] = tid;
    __syncthreads();
    const int nf = n_flagged;
    for (int f = 0; f < nf; ++f) {
        const int64_t row = (int64_t)blockIdx.x * 256 + flagged[f];
        const int64_t c = row / N;
        const int q = (int)(row % N);
        const float* __restrict__ cloud = xyz + c * N * 3;
        const float qx = cloud[3 * q], qy = cloud[3 * q + 1], qz = cloud[3 * q + 2];
        const float4 pq = make_float4(qx, qy, qz, sqnorm3(qx, qy, qz));
        for (int j = tid; j < N; j += 256) {
            const float x = cloud[3 * j], y = cloud[3 * j + 1], z = cloud[3 * j + 2];
            cand[j].v = knn_dist(pq, make_float4(x, y, z, sqnorm3(x, y, z)));
            cand[j].i = j;
        }
        __syncthreads();
        if ((long long)k * 64 <= N) {              // torch's heap-select branch: one thread
            if (tid == 0) ogmm_select::heap_select(cand, k, N);
        } else if (have_lists) {
            block_introselect(cand, k - 1, N, Lpos, Rpos, wave_tot, nullptr);
        } else {                                     // clouds so large that the position lists do not fit next to the candidates
            if (tid == 0) ogmm_select::introselect(cand, k - 1, N);
        }
        __syncthreads();
        if (tid == 0) {
            for (int a = 1; a < k; ++a) {       // order the kept set by (distance, index)
                const ogmm_select::Cand v = cand[a];
                int b = a;
                while (b > 0 && (v.v < cand[b - 1].v || (v.v == cand[b - 1].v && v.i < cand[b - 1].i))) { cand[b] = cand[b - 1]; --b; }
                cand[b] = v;
            }
            for (int a = 0; a < k; ++a) idx[row * k + a] = cand[a].i;
        }
        __syncthreads();
    }
}

// torch.topk(v, k, dim = -1, largest) of a [rows][n] map with the CPU kernel's choice among TIED values (torch_topk_select.h: the selection algorithms of
// ATen's TopKImpl.h on GNU libstdc++, move for move) -- the Welsch term of the training loss takes the top_k points of the 0 / 1 ground-truth overlap
// labels (lib/loss.py:92, :95): with more than top_k ones the kept SET is decided by those moves, and the loss's value with it.  One workgroup per row;
// "largest" runs the smallest-selection on the negated values (the comparator a > b is -a < -b; a NaN counts as the largest value, as in ATen).  The kept
// set is written in (value, index) order (torch leaves the order among equal values unspecified).
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ v, int64_t ldv, int n, int k, int largest, int32_t* __restrict__ idx,
                                                        int have_lists) {
    extern __shared__ __attribute__((aligned(16))) ogmm_select::Cand cand_t[];   // [n], then int Lpos[n], Rpos[n]
    int* Lpos = reinterpret_cast<int*>(cand_t + n);
    int* Rpos = Lpos + n;
    __shared__ int wave_tot_t[4];
    const int tid = threadIdx.x;
    const float* __restrict__ row = v + (int64_t)blockIdx.x * ldv;
    for (int j = tid; j < n; j += 256) {
        float x = row[j];
        if (largest) x = x != x ? -__builtin_inff() : -x;
        else if (x != x) x = __builtin_inff();
        cand_t[j].v = x;
        cand_t[j].i = j;
    }
    __syncthreads();
    if ((long long)k * 64 <= n) {
        if (tid == 0) ogmm_select::heap_select(cand_t, k, n);
    } else if (have_lists) {
        block_introselect(cand_t, k - 1, n, Lpos, Rpos, wave_tot_t, nullptr);
    } else {
        if (tid == 0) ogmm_select::introselect(cand_t, k - 1, n);
    }
    __syncthreads();
    // order the kept set by (value, index): rank of every element among the k (k <= a few thousand: k^2 / 256 comparisons per thread)
    for (int a = tid; a < k; a += 256) {
        const ogmm_select::Cand c = cand_t[a];
        int rank = 0;
        for (int b = 0; b < k; ++b) {
            const ogmm_select::Cand o = cand_t[b];
            rank += (o.v < c.v || (o.v == c.v && o.i < c.i)) ? 1 : 0;
        }
        idx[(int64_t)blockIdx.x * k + rank] = c.i;
    }
}

// ------------------------------------------------------------------------------------------------
// FPS: one workgroup per (sampling, cloud); points and the running-min array live in registers
// (PPT points per thread, interleaved so loads coalesce), a copy of the cloud in LDS serves the
// centroid fetch; per pick: PPT distance updates, a wave argmax by shuffles, one LDS hop across waves.
// ------------------------------------------------------------------------------------------------
struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) {   // torch.max: first maximal index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

template <int PPT>
__global__ __launch_bounds__(256) void fps_kernel(const float* __restrict__ xyz, int C, int N, int npoint,
                                                  const int32_t* __restrict__ start, int32_t* __restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // xyz copy [3N] + reduction scratch
    float* cloud_s = lds;
    __shared__ Best wave_best[4];
    __shared__ double csum[4][3];
    const int c = blockIdx.x, set = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    for (int j = tid; j < 3 * N; j += 256) cloud_s[j] = cloud[j];
    __syncthreads();
    float px[PPT], py[PPT], pz[PPT], run[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * 256;
        const bool ok = p < N;
        px[i] = ok ? cloud_s[3 * p] : 0.f;
        py[i] = ok ? cloud_s[3 * p + 1] : 0.f;
        pz[i] = ok ? cloud_s[3 * p + 2] : 0.f;
        run[i] = 1e10f;
    }
    auto block_argmax = [&]() -> int {
        Best b = {-1.0f, 0x7fffffff};
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * 256;
            if (p < N) b = better(b, Best{run[i], p});
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Best other = {__shfl_xor(b.v, o, 64), __shfl_xor(b.i, o, 64)};
            b = better(b, other);
        }
        __syncthreads();            // previous readers of wave_best are done
        if (lane == 0) wave_best[wave] = b;
        __syncthreads();
        Best r = wave_best[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) r = better(r, wave_best[w]);
        return r.i;
    };
    auto update = [&](float cx, float cy, float cz) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float d = sqdist3_direct(px[i], py[i], pz[i], cx, cy, cz);
            if (d < run[i]) run[i] = d;
        }
    };

    int far;
    if (start == nullptr) {
        // is_center=True: centroid = mean over the N points (fp64 accumulation, rounded once)
        double sx = 0, sy = 0, sz = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (tid + i * 256 < N) { sx += px[i]; sy += py[i]; sz += pz[i]; }
        sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
        if (lane == 0) { csum[wave][0] = sx; csum[wave][1] = sy; csum[wave][2] = sz; }
        __syncthreads();
        const float fn = (float)N;
        const float cx = (float)(csum[0][0] + csum[1][0] + csum[2][0] + csum[3][0]) / fn;
        const float cy = (float)(csum[0][1] + csum[1][1] + csum[2][1] + csum[3][1]) / fn;
        const float cz = (float)(csum[0][2] + csum[1][2] + csum[2][2] + csum[3][2]) / fn;
        update(cx, cy, cz);
        far = block_argmax();
    } else {
        far = start[(int64_t)set * C + c];
    }
    int32_t* out = ids + ((int64_t)set * C + c) * npoint;
    for (int s = 0; s < npoint; ++s) {
        if (tid == 0) out[s] = far;
        update(cloud_s[3 * far], cloud_s[3 * far + 1], cloud_s[3 * far + 2]);
        if (s + 1 < npoint) far = block_argmax();
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ feats, int64_t ld, int N, int D,
                                                          const int32_t* __restrict__ ids, int S,
                                                          const int32_t* __restrict__ cloud_map, float* __restrict__ out) {
    const int c = blockIdx.y, s = blockIdx.x;
    const int sc = cloud_map ? cloud_map[c] : c;
    const int row = ids[(int64_t)sc * S + s];
    const float* __restrict__ src = feats + ((int64_t)sc * N + row) * ld;
    float* __restrict__ dst = out + ((int64_t)c * S + s) * D;
    for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4)
        *reinterpret_cast<float4*>(dst + d) = *reinterpret_cast<const float4*>(src + d);
}

}  // namespace

extern "C" int ogmm_topk_rows(const float* v, int64_t ldv, int rows, int n, int k, int largest, int32_t* idx, void* stream) {
    OGMM_REQUIRE(v && idx && rows > 0 && n > 0 && k >= 1 && k <= n && ldv >= n, "ogmm_topk_rows: null pointer or bad shape (rows=%d n=%d k=%d)", rows, n, k);
    const size_t with_lists = (size_t)n * (sizeof(ogmm_select::Cand) + 2 * sizeof(int)), without = (size_t)n * sizeof(ogmm_select::Cand);
    OGMM_REQUIRE(without <= 158 * 1024, "ogmm_topk_rows: n=%d does not fit the LDS-resident candidate list (max 20224)", n);
    const int have_lists = with_lists <= 158 * 1024 ? 1 : 0;
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(256), have_lists ? with_lists : without, ogmm::as_stream(stream), v, ldv, n, k, largest, idx,
                       have_lists);
    return ogmm::check_launch("ogmm_topk_rows");
}

extern "C" int ogmm_knn(const float* xyz, int C, int N, int k, int32_t* idx, void* stream) {
    OGMM_REQUIRE(xyz && idx && C > 0 && N > 0, "ogmm_knn: null pointer or empty input");
    OGMM_REQUIRE(k >= 1 && k <= 32 && k <= N, "ogmm_knn: need 1 <= k <= min(32, N), got k=%d N=%d", k, N);
    OGMM_REQUIRE((size_t)N * 16 <= 160 * 1024, "ogmm_knn: N=%d does not fit the LDS-resident cloud (max 10240)", N);
    dim3 grid((N + 255) / 256, C);
    const size_t lds = (size_t)N * sizeof(float4);
    hipStream_t s = ogmm::as_stream(stream);
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {          // clouds beyond 4096 points need more than the default 64 KB of dynamic LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel<21>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel<33>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_resolve_ties_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    }
    // the two-scan form while its candidate lists fit beside the cloud in 64 KiB (several workgroups per CU stay resident); else one scan
    const int KLsel = k <= 8 ? 9 : (k <= 20 ? 21 : 33);
    const size_t lds2 = lds + (size_t)2 * (KLsel - 1) * 256 * sizeof(short);
    static const int two_scans = [] { const char* e = getenv("OGMM_KNN_TWO_SCANS"); return e ? atoi(e) : 1; }();
    // (for k <= 8 the single scan stays: its ladder is short, 85 against 88 us at N = 1024, k = 5; k = 20: 175 -> 140 us)
    // tie resolution folded into knn2_kernel's tail when its scratch (N candidates of 8 B, + 2 N position ints on the nth_element branch) fits into the
    // LDS of the candidate lists, which are dead by then; 0 = separate launch, 1 = folded (heap / serial branch), 2 = folded with position lists
    static const int fold_env = [] { const char* e = getenv("OGMM_KNN_FOLD_TIES"); return e ? atoi(e) : 1; }();
    int fold = 0;
    if (two_scans && k > 8 && lds2 <= 64 * 1024 && fold_env) {
        const size_t lists_bytes = lds2 - lds;
        const bool heap = (long long)k * 64 <= N;
        if (!heap && (size_t)N * (sizeof(ogmm_select::Cand) + 2 * sizeof(int)) <= lists_bytes) fold = 2;
        else if (heap && (size_t)N * sizeof(ogmm_select::Cand) <= lists_bytes) fold = 1;
    }
    if (two_scans && k > 8 && lds2 <= 64 * 1024) {
        if (k <= 20) hipLaunchKernelGGL(knn2_kernel<21>, grid, dim3(256), lds2, s, xyz, N, k, idx, fold);
        else hipLaunchKernelGGL(knn2_kernel<33>, grid, dim3(256), lds2, s, xyz, N, k, idx, fold);
        if (fold) return ogmm::check_launch("ogmm_knn");
    } else if (k <= 8) hipLaunchKernelGGL(knn_kernel<9>, grid, dim3(256), lds, s, xyz, N, k, idx);
    else if (k <= 20) hipLaunchKernelGGL(knn_kernel<21>, grid, dim3(256), lds, s, xyz, N, k, idx);
    else hipLaunchKernelGGL(knn_kernel<33>, grid, dim3(256), lds, s, xyz, N, k, idx);
    if (int rc = ogmm::check_launch("ogmm_knn")) return rc;
    const int64_t rows = (int64_t)C * N;
    // the position lists of the workgroup partition only exist on the nth_element branch: the heap branch (k * 64 <= N) keeps its 8 B per
    // candidate, so that its workgroups still fit next to a kernel that holds most of a CU's LDS (the persistent EdgeConv kernel)
    const bool lists = (long long)k * 64 > N && (size_t)N * (sizeof(ogmm_select::Cand) + 2 * sizeof(int)) <= 156 * 1024;
    const size_t ties_lds = (size_t)N * (sizeof(ogmm_select::Cand) + (lists ? 2 * sizeof(int) : 0));
    hipLaunchKernelGGL(knn_resolve_ties_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), ties_lds, s, xyz, N, k, rows, idx, lists ? 1 : 0);
    return ogmm::check_launch("ogmm_knn(resolve ties)");
}

extern "C" int ogmm_pack_clouds(const float* src, const float* tgt, int B, int N, float* xyz, void* stream) {
    OGMM_REQUIRE(src && tgt && xyz && B > 0 && N > 0, "ogmm_pack_clouds: null pointer or empty input");
    hipLaunchKernelGGL(pack_clouds_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)(2 * B)), dim3(256), 0, ogmm::as_stream(stream), src, tgt, B, N, xyz);
    return ogmm::check_launch("ogmm_pack_clouds");
}

constexpr size_t KNN_HEAD_STATIC_LDS = 8 * 1024;          // >= knn4_kernel's group_segment_fixed_size (tests/test_abi_and_host.py reads it off the code object)
static bool knn_head_fits(int N, int k, int* fold_out) {
    if (N <= 0 || k <= 8 || k > 32 || k > N || N < 6) return false;
    const int KLsel = k <= 20 ? 21 : 33;
    const size_t lds = (size_t)N * sizeof(float4), lds2 = lds + (size_t)2 * (KLsel - 1) * 256 * sizeof(short);
    // the workgroup's budget is 64 KiB INCLUDING knn4_kernel's static arrays (tie_rows, geo, the tie resolution's scratch: ~8.3 KiB; ADVICE.md round 5 -- the
    // check used to bound the dynamic part alone, so N = 2369 ... 2816 answered "supported" for a 72 KiB workgroup nothing had ever launched).  Beyond it the
    // forward takes the three-kernel head (ogmm_knn + ogmm_knn(5) + ogmm_pos_hidden).
    if (lds2 + KNN_HEAD_STATIC_LDS > 64 * 1024) return false;
    // both tie resolutions (rank k, rank 5) run in the LDS of the candidate lists: the scratch of ogmm_knn's folded form
    const size_t lists_bytes = lds2 - lds;
    const bool heap = (long long)k * 64 <= N;
    int fold = 0;
    if (!heap && (size_t)N * (sizeof(ogmm_select::Cand) + 2 * sizeof(int)) <= lists_bytes) fold = 2;
    else if (heap && (size_t)N * sizeof(ogmm_select::Cand) <= lists_bytes) fold = 1;
    else if (!heap && (size_t)N * sizeof(ogmm_select::Cand) <= lists_bytes) fold = 1;          // (introselect by one thread: correct, slower on a flagged row)
    if (fold_out) *fold_out = fold;
    return fold != 0 && (size_t)N * sizeof(ogmm_select::Cand) <= lists_bytes;          // rank 5 takes the heap branch whenever 5 * 64 <= N, else one thread's introselect
}

extern "C" int ogmm_knn_pos_head_supported(int N, int k) { return knn_head_fits(N, k, nullptr) ? 1 : 0; }

extern "C" int64_t ogmm_knn_pos_head_workspace_bytes(int C, int N) {
    return (int64_t)C * ((N + 255) / 256) * (N / 32) * 256 * (int64_t)sizeof(unsigned) + 256;
}

extern "C" int ogmm_knn_pos_head(const float* xyz, int C, int N, int k, int32_t* idx, int32_t* idx5, const float* w_dis, const float* s_dis, const float* t_dis,
                                 const float* w_ang, const float* s_ang, const float* t_ang, float* hid_dis, float* hid_ang, void* workspace, void* stream) {
    OGMM_REQUIRE(xyz && idx && workspace && C > 0 && N > 0, "ogmm_knn_pos_head: null pointer or empty input");
    int fold = 0;
    OGMM_REQUIRE(knn_head_fits(N, k, &fold), "ogmm_knn_pos_head: N=%d, k=%d is outside this kernel's range (ogmm_knn_pos_head_supported); use ogmm_knn + ogmm_pos_hidden", N, k);
    const bool head = idx5 != nullptr;
    OGMM_REQUIRE(!head || (w_dis && s_dis && t_dis && w_ang && s_ang && t_ang && hid_dis && hid_ang), "ogmm_knn_pos_head: the positional front end needs all six constants and both outputs");
    OGMM_REQUIRE(!head || (ogmm::aligned16(w_dis) && ogmm::aligned16(s_dis) && ogmm::aligned16(t_dis) && ogmm::aligned16(w_ang) && ogmm::aligned16(s_ang) && ogmm::aligned16(t_ang) &&
                           ogmm::aligned16(hid_dis) && ogmm::aligned16(hid_ang)), "ogmm_knn_pos_head: constants and hidden maps must be 16-byte aligned");
    const int KLsel = k <= 20 ? 21 : 33;
    const size_t lds2 = (size_t)N * sizeof(float4) + (size_t)2 * (KLsel - 1) * 256 * sizeof(short);
    knn_pos_args pa = {idx5, w_dis, s_dis, t_dis, w_ang, s_ang, t_ang, hid_dis, hid_ang};
    hipStream_t s = ogmm::as_stream(stream);
    dim3 grid((N + 255) / 256, C);
    unsigned* bm = reinterpret_cast<unsigned*>(workspace);
    if (KLsel == 21) {
        if (head) hipLaunchKernelGGL((knn4_kernel<21, true>), grid, dim3(256), lds2, s, xyz, N, k, idx, bm, fold, pa);
        else hipLaunchKernelGGL((knn4_kernel<21, false>), grid, dim3(256), lds2, s, xyz, N, k, idx, bm, fold, pa);
    } else {
        if (head) hipLaunchKernelGGL((knn4_kernel<33, true>), grid, dim3(256), lds2, s, xyz, N, k, idx, bm, fold, pa);
        else hipLaunchKernelGGL((knn4_kernel<33, false>), grid, dim3(256), lds2, s, xyz, N, k, idx, bm, fold, pa);
    }
    return ogmm::check_launch("ogmm_knn_pos_head");
}

extern "C" int ogmm_fps(const float* xyz, int C, int N, int npoint, int n_sets, const int32_t* start, int32_t* ids, void* stream) {
    OGMM_REQUIRE(xyz && ids && C > 0 && N > 0 && npoint > 0 && n_sets > 0, "ogmm_fps: null pointer or empty input");
    OGMM_REQUIRE(npoint <= N, "ogmm_fps: npoint=%d > N=%d", npoint, N);
    OGMM_REQUIRE(N <= 4096, "ogmm_fps: N=%d > 4096 points per cloud not supported", N);
    OGMM_REQUIRE(start != nullptr || n_sets == 1, "ogmm_fps: centre start supports one sampling per call");
    dim3 grid(C, n_sets);
    const size_t lds = (size_t)N * 3 * sizeof(float);
    hipStream_t s = ogmm::as_stream(stream);
    const int ppt = (N + 255) / 256;
    if (ppt <= 1) hipLaunchKernelGGL(fps_kernel<1>, grid, dim3(256), lds, s, xyz, C, N, npoint, start, ids);
    else if (ppt <= 2) hipLaunchKernelGGL(fps_kernel<2>, grid, dim3(256), lds, s, xyz, C, N, npoint, start, ids);
    else if (ppt <= 4) hipLaunchKernelGGL(fps_kernel<4>, grid, dim3(256), lds, s, xyz, C, N, npoint, start, ids);
    else if (ppt <= 8) hipLaunchKernelGGL(fps_kernel<8>, grid, dim3(256), lds, s, xyz, C, N, npoint, start, ids);
    else hipLaunchKernelGGL(fps_kernel<16>, grid, dim3(256), lds, s, xyz, C, N, npoint, start, ids);
    return ogmm::check_launch("ogmm_fps");
}

extern "C" int ogmm_gather_rows(const float* feats, int64_t ld, int C, int N, int D, const int32_t* ids, int S,
                                const int32_t* cloud_map, float* out, void* stream) {
    OGMM_REQUIRE(feats && ids && out && C > 0 && N > 0 && S > 0, "ogmm_gather_rows: null pointer or empty input");
    OGMM_REQUIRE(D > 0 && D % 4 == 0 && ld % 4 == 0 && ogmm::aligned16(feats) && ogmm::aligned16(out),
                 "ogmm_gather_rows: D and ld must be multiples of 4 and pointers 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(S, C), dim3(128), 0, ogmm::as_stream(stream), feats, ld, N, D, ids, S, cloud_map, out);
    return ogmm::check_launch("ogmm_gather_rows");
}
