// K15 for shapes whose N x J cost matrix does not fit into one CU's LDS (e.g. N = 2048, J = 64: 512 KB): the same E/M loop
// (lib/utils.py:269-288 with :69-108 and :130-140) as a fixed sequence of small grid-wide kernels over a cost matrix that lives
// in a workspace (L2 / Infinity-Cache resident: 64 MB for 128 clouds).  One cloud on one workgroup recomputes every distance
// in every pass and uses half of the chip (128 workgroups on 256 CUs): 15.7 ms at B = 64, N = 2048, J = 64, the long pole of the
// forward.  Here every pass spreads over all CUs:
//   init   : log p, first centres                                         (1 launch)
//   per outer iteration (10):  cost  ->  [u-pass, v-pass] x sk_iters  ->  gamma  ->  M-step        (2 sk_iters + 3 launches)
// All launches are issued by ONE call (no host synchronisation; the reference needs ~3000 launches and 200 .item() syncs).
// The arithmetic of every entry is that of gmm_em_cached_kernel; column reductions differ in summation order only.
#include "ogmm_common.h"
#include "gmm_exit.h"
#include <cstdlib>

namespace {

using namespace ogmm;

__device__ __forceinline__ float cdist_mm2(float x, float y, float z, float xn, float mx, float my, float mz, float mn) {
    float acc = mul_rn(-2.0f * x, mx);
    acc = __fmaf_rn(-2.0f * y, my, acc);
    acc = __fmaf_rn(-2.0f * z, mz, acc);
    acc = add_rn(acc, xn);
    acc = add_rn(acc, mn);
    return sqrtf(fmaxf(acc, 0.0f));
}

struct EmWs {
    float* cost;     // [C][J][N]  cost, later unnormalised gamma
    float* u;        // [C][N]
    float* v;        // [C][J]
    float* logp;     // [C][N]
    float* rclip;    // [C][N]
    float4* mu;      // [C][J]  x, y, z, |mu|^2
    float* vbuf;     // [2][C][J]           v of the last two sweeps (fused sweeps)
    float* pbuf;     // [2][C][chunks][J][2] per-chunk (max, exp-sum) of the pending v-update
    double* mpart;   // [C][chunks][J][4]   per-chunk M-step sums (resident kernel)
    int* sync;       // [0] ticket counter, [1] error flag, [2 + c] arrivals at cloud c's barrier (resident kernel)
};

// one workgroup per cloud: p = o / max(sum o, 1e-4), log(p + 1e-8); centres = xyz[ids0]
__global__ __launch_bounds__(256) void em_init_kernel(const float* __restrict__ xyz, const float* __restrict__ o, const int32_t* __restrict__ ids0,
                                                      int N, int J, EmWs w) {
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ oc = o + (int64_t)c * N;
    float part = 0.0f;
    for (int n = tid; n < N; n += 256) part += oc[n];
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float osum = fmaxf((red[0] + red[1]) + (red[2] + red[3]), 1e-4f);
    for (int n = tid; n < N; n += 256) w.logp[(int64_t)c * N + n] = logf(oc[n] / osum + 1e-8f);
    for (int j = tid; j < J; j += 256) {
        const float* p = xyz + ((int64_t)c * N + ids0[(int64_t)c * J + j]) * 3;
        w.mu[(int64_t)c * J + j] = make_float4(p[0], p[1], p[2], sqnorm3(p[0], p[1], p[2]));
    }
}

// cost[c][j][n] = cdist(xyz_n, mu_j) / tau;  u = 0, v = 0.   grid (N/256, C)
__global__ __launch_bounds__(256) void em_cost_kernel(const float* __restrict__ xyz, int N, int J, float inv_tau, EmWs w) {
    const int c = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) for (int j = threadIdx.x; j < J; j += 256) w.v[(int64_t)c * J + j] = 0.0f;
    if (n >= N) return;
    const float* p = xyz + ((int64_t)c * N + n) * 3;
    const float x = p[0], y = p[1], z = p[2], xn = sqnorm3(x, y, z);
    w.u[(int64_t)c * N + n] = 0.0f;
    const float4* __restrict__ mu = w.mu + (int64_t)c * J;
    float* __restrict__ Cc = w.cost + (int64_t)c * J * N + n;
    for (int j = 0; j < J; ++j) {
        const float4 m = mu[j];
        Cc[(int64_t)j * N] = cdist_mm2(x, y, z, xn, m.x, m.y, m.z, m.w) * inv_tau;
    }
}

// Early exit (gmm_exit.h) in the two-launch sweeps: NOT lagged -- the v kernel of sweep m completes the sweep, its last column workgroup per cloud
// sums the cloud's residual (the u kernel's per-chunk sums + the columns' |dv|, in index order), publishes it, and the last cloud of the call group
// takes the decision; every later launch of the E-step starts by reading the group's stop flag.

// u^{l+1}: one thread per row, the row's J exponents held in registers between the max and the exp-sum pass.  grid (N/256, C)
template <int JMAX>
__global__ __launch_bounds__(256) void em_u_kernel(int N, int J, float inv_eps, float eps, EmWs w, int it, EmExit x) {
    extern __shared__ float vs[];                      // [J], then 4 floats for the residual
    const int c = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (x.on && x.kstop[(c / x.G) * x.iters + it]) return;
    for (int j = threadIdx.x; j < J; j += 256) vs[j] = w.v[(int64_t)c * J + j];
    __syncthreads();
    const bool valid = n < N;
    float du = 0.0f;
    if (valid) {
        const float* __restrict__ Cc = w.cost + (int64_t)c * J * N + n;
        const float un = w.u[(int64_t)c * N + n];
        float t[JMAX];
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            t[j] = j < J ? ((-Cc[(int64_t)j * N] + un) + vs[j]) * inv_eps : -__builtin_inff();
            mx = fmaxf(mx, t[j]);
        }
        double se = 0.0;          // sums of exponentials in fp64: the E/M is ill-conditioned for J close to N (5e-6 on mu at N=717, J=128)
#pragma unroll
        for (int j = 0; j < JMAX; ++j) se += j < J ? (double)expf(t[j] - mx) : 0.0;
        const float un1 = eps * (w.logp[(int64_t)c * N + n] - (mx + logf((float)se))) + un;
        w.u[(int64_t)c * N + n] = un1;
        du = fabsf(un1 - un);
    }
    if (x.on || x.resid) {                              // this chunk's sum |du|
        float* sred = vs + J;
        du = wave_sum(du);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = du;
        __syncthreads();
        if (threadIdx.x == 0) x.dupart[(int64_t)c * gridDim.x + blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    }
}

// v^{l+1}: one workgroup per (cluster j, cloud).  grid (J, C).  m = the sweep (1-based).
__global__ __launch_bounds__(256) void em_v_kernel(int N, int J, float inv_eps, float eps, float logq, EmWs w, int it, int m, int n_chunks, EmExit x) {
    __shared__ float red[4];
    __shared__ double redd[4];
    const int j = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (x.on && x.kstop[(c / x.G) * x.iters + it]) return;
    const float* __restrict__ Cj = w.cost + ((int64_t)c * J + j) * N;
    const float* __restrict__ u = w.u + (int64_t)c * N;
    const float vj = w.v[(int64_t)c * J + j];
    constexpr int VR = 16;                     // exponents of up to 16 rows per thread stay in registers (N <= 4096), the rest is recomputed
    float t[VR];
    float mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < VR; ++i) {
        const int n = tid + i * 256;
        t[i] = n < N ? ((-Cj[n] + u[n]) + vj) * inv_eps : -__builtin_inff();
        mx = fmaxf(mx, t[i]);
    }
    for (int n = tid + VR * 256; n < N; n += 256) mx = fmaxf(mx, ((-Cj[n] + u[n]) + vj) * inv_eps);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    double se = 0.0;
#pragma unroll
    for (int i = 0; i < VR; ++i) se += tid + i * 256 < N ? (double)expf(t[i] - mx) : 0.0;
    for (int n = tid + VR * 256; n < N; n += 256) se += (double)expf(((-Cj[n] + u[n]) + vj) * inv_eps - mx);
    se = wave_sum_d(se);
    if (lane == 0) redd[wave] = se;
    __syncthreads();
    if (tid == 0) {
        const float vj1 = eps * (logq - (mx + logf((float)((redd[0] + redd[1]) + (redd[2] + redd[3]))))) + vj;
        w.v[(int64_t)c * J + j] = vj1;
        if (x.on || x.resid) {
            em_st_agent(w.vbuf + (int64_t)c * J + j, fabsf(vj1 - vj));          // (vbuf is free in the two-launch form)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int prev = __hip_atomic_fetch_add(x.ccount + ((int64_t)it * x.sk + (m - 1)) * x.C + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == J - 1) {                       // the cloud's last column: its residual of this sweep
                float du = 0.0f, dv = 0.0f;
                for (int ch = 0; ch < n_chunks; ++ch) du += x.dupart[(int64_t)c * n_chunks + ch];
                for (int jj = 0; jj < J; ++jj) dv += em_ld_agent(w.vbuf + (int64_t)c * J + jj);
                const float r = du + dv;
                if (x.resid) x.resid[((int64_t)c * x.iters + it) * x.sk + (m - 1)] = r;
                if (x.on && m < x.sk) em_exit_publish(x, c, it, m - 1, r);
            }
        }
    }
}

// ---- fused sweeps (J <= 64): ONE launch and ONE read of the cost matrix per Sinkhorn sweep instead of two.
// The v-update needs a column log-sum-exp over ALL rows of the cloud, i.e. a grid-wide dependency; but a row chunk can contribute its
// (max, exp-sum) per column right after it has updated its own u, and the next launch starts by merging the chunks' partials:
//   launch k:  v_{k-1} = eps (log q - LSE(partials_{k-1})) + v_{k-2}     (every workgroup, redundantly; workgroup 0 stores it)
//              u_k     = eps (log p - LSE_j((-C + u_{k-1} + v_{k-1}) / eps)) + u_{k-1}            (thread = row, costs in registers)
//              partials_k[chunk][j] = (max, sum exp) over the chunk's rows of (-C + u_k + v_{k-1}) / eps
// The column reduction inside a wave goes through a private 64 x 33 LDS tile (32 columns at a time): lane (column, row half) reads its
// column's 32 rows.  exp(x - max) on v_exp_f32 as in the on-chip kernel.  N = 2048, J = 64, 128 clouds: 2 x 45 us -> ~20 us per sweep.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// v of the sweep whose partials are in `part` (or 0 for `first`), into LDS vs[J]; optionally stored to v_out
// returns this thread's share of sum_j |v - v_in| (the early exit's residual)
__device__ __forceinline__ float em_finish_v(int c, int J, int n_chunks, bool first, float eps, float logq, const float* __restrict__ v_in,
                                             const float* __restrict__ part, float* __restrict__ v_out, float* vs) {
    float dv = 0.0f;
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        float vcur = 0.0f;
        if (!first) {
            const float* __restrict__ pj = part + ((int64_t)c * n_chunks * J + j) * 2;
            float M = -__builtin_inff();
            for (int ch = 0; ch < n_chunks; ++ch) M = fmaxf(M, pj[(int64_t)ch * J * 2]);
            float S = 0.0f;
            for (int ch = 0; ch < n_chunks; ++ch) S = fmaf(pj[(int64_t)ch * J * 2 + 1], expf(pj[(int64_t)ch * J * 2] - M), S);
            const float vold = v_in[(int64_t)c * J + j];
            vcur = eps * (logq - (M + logf(S))) + vold;
            dv += fabsf(vcur - vold);
        }
        vs[j] = vcur;
        if (v_out) v_out[(int64_t)c * J + j] = vcur;
    }
    return dv;
}

// Early exit (gmm_exit.h) in the fused sweeps: launch m finishes v_{m-1}, so the residual of sweep m - 1 is complete only there (chunk 0 of every
// cloud: the chunks' sum |du| of launch m - 1 + sum |dv|); its last publisher per call group takes the decision, and launch m + 1 -- the first to
// see it -- and all later launches of the E-step return at once.  Sweep m itself has then been applied already: u is double-buffered by the
// sweep's parity (w.u / x.u2) and v lives in vbuf by parity anyway, so the gamma pass simply picks up (u, v) of the sweep that ended the E-step.
__device__ __forceinline__ void em_fused_residual(const EmExit& x, int c, int it, int m, int n_chunks, float dv_part, bool publish) {
    // wave 0 of the chunk-0 workgroup of cloud c, at the start of launch m >= 2: residual of sweep m - 1 (J <= 64: all of dv_part sits in wave 0)
    const float dv = wave_sum(dv_part);
    float r = 0.0f;
    if ((threadIdx.x & 63) == 0) {
        float du = 0.0f;
        const float* __restrict__ dp = x.dupart + ((int64_t)((m - 1) & 1) * x.C + c) * n_chunks;
        for (int ch = 0; ch < n_chunks; ++ch) du += em_ld_agent(dp + ch);
        r = du + dv;
        if (x.resid) x.resid[((int64_t)c * x.iters + it) * x.sk + (m - 2)] = r;
    }
    if (publish) em_exit_publish_wave(x, c, it, m - 2, r);          // (the whole wave: a last arriver sums its group's residuals in parallel)
}

// The costs themselves are NOT read: thread = row recomputes its J distances from the point and the cloud's centres (LDS) with the
// arithmetic of em_cost_kernel -- bit-identical values, 10 VALU instructions instead of a 4-byte load per entry.  The cost matrix of
// 128 clouds (64 MB) came from the Infinity Cache at ~2 TB/s: 31-55 us per sweep, 100 sweeps per forward; recomputed: see DESIGN.
// FULL: J == JMAX, every `j < J` test is compile-time true (a quarter of the kernel's instructions were those tests and their selects)
template <int JMAX, bool FULL>
__global__ __launch_bounds__(256, 4) void em_sweep_kernel(const float* __restrict__ xyz, float inv_tau, int N, int J, float inv_eps, float eps,
                                                          float logq, int it, int m, EmWs w, EmExit x) {
    __shared__ float vs[JMAX];
    __shared__ float4 mus[JMAX];
    __shared__ float tile[4][64][33];
    __shared__ float wp[4][JMAX][2];
    __shared__ float sred[4];
    const int c = blockIdx.y, chunk = blockIdx.x, n_chunks = gridDim.x, C = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = chunk * 256 + tid;
    const int first = m == 1, parity = m & 1;
    const bool track = x.on || x.resid != nullptr;
    if (x.on) {          // sweep ks ended this E-step for the cloud's call group: known from launch ks + 2 on (launch ks + 1 is where it is decided)
        const int ks = x.kstop[(c / x.G) * x.iters + it];
        if (ks && m >= ks + 2) return;
    }
    const int64_t vsz = (int64_t)C * J, psz = (int64_t)C * n_chunks * J * 2;
    // launch m (parity = m & 1): reads v_{m-2} from vbuf[m & 1] and partials_{m-1} from pbuf[(m-1) & 1]; writes v_{m-1} to vbuf[(m+1) & 1]
    // and partials_m to pbuf[m & 1]
    const float dv_part = em_finish_v(c, J, n_chunks, first != 0, eps, logq, w.vbuf + parity * vsz, w.pbuf + (parity ^ 1) * psz,
                                      chunk == 0 ? w.vbuf + (parity ^ 1) * vsz : nullptr, vs);
    if (track && chunk == 0 && m >= 2 && wave == 0) em_fused_residual(x, c, it, m, n_chunks, dv_part, x.on != 0);
    for (int j = tid; j < J; j += 256) mus[j] = w.mu[(int64_t)c * J + j];
    __syncthreads();
    const float* __restrict__ u_in = (x.on && ((m - 1) & 1)) ? x.u2 : w.u;          // u_{m-1}; with the exit on u_m goes to the other buffer
    float* __restrict__ u_out = (x.on && (m & 1)) ? x.u2 : w.u;
    const bool valid = n < N;
    const float* __restrict__ pt = xyz + ((int64_t)c * N + (valid ? n : 0)) * 3;
    const float px = pt[0], py = pt[1], pz = pt[2], pn = sqnorm3(px, py, pz);
    float cst[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const float4 m = mus[(FULL || j < J) ? j : 0];
        cst[j] = (FULL || j < J) ? cdist_mm2(px, py, pz, pn, m.x, m.y, m.z, m.w) * inv_tau : 0.0f;
    }
    const float un = (valid && !first) ? u_in[(int64_t)c * N + n] : 0.0f;          // u = v = 0 at the start of every outer iteration
    float mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
        if (FULL || j < J) mx = fmaxf(mx, ((-cst[j] + un) + vs[j]) * inv_eps);
    float se = 0.0f;
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
        if (FULL || j < J) se += fexp(((-cst[j] + un) + vs[j]) * inv_eps - mx);
    const float unew = valid ? eps * (w.logp[(int64_t)c * N + n] - (mx + logf(se))) + un : 0.0f;
    if (valid) u_out[(int64_t)c * N + n] = unew;
    if (track) {                                                                   // this chunk's sum |du| of sweep m
        const float du = wave_sum(valid ? fabsf(unew - un) : 0.0f);
        if (lane == 0) sred[wave] = du;
    }
    // column partials of the pending v-update, 32 columns at a time through this wave's tile
#pragma unroll
    for (int h = 0; h < JMAX / 32; ++h) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = 32 * h + i;
            tile[wave][lane][i] = (valid && (FULL || j < J)) ? ((-cst[j] + unew) + vs[j]) * inv_eps : -__builtin_inff();
        }
        // (same wave: LDS operations complete in order)
        const int col = lane & 31, rh = lane >> 5;
        float cm = -__builtin_inff();
        float y[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) { y[r] = tile[wave][rh * 32 + r][col]; cm = fmaxf(cm, y[r]); }
        float cs = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) cs += fexp(y[r] - cm);            // all -inf (column past J, rows past N): exp(nan) -- masked below
        if (!(cm > -__builtin_inff())) cs = 0.0f;
        const float om = __shfl_xor(cm, 32, 64), os = __shfl_xor(cs, 32, 64);
        const float M = fmaxf(cm, om);
        const float S = (cm > -__builtin_inff() ? cs * fexp(cm - M) : 0.0f) + (om > -__builtin_inff() ? os * fexp(om - M) : 0.0f);
        if (lane < 32) { wp[wave][32 * h + col][0] = M; wp[wave][32 * h + col][1] = S; }
    }
    __syncthreads();
    if (tid < J) {
        float M = fmaxf(fmaxf(wp[0][tid][0], wp[1][tid][0]), fmaxf(wp[2][tid][0], wp[3][tid][0]));
        float S = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (wp[q][tid][0] > -__builtin_inff()) S = fmaf(wp[q][tid][1], fexp(wp[q][tid][0] - M), S);
        float* __restrict__ po = w.pbuf + parity * psz + (((int64_t)c * n_chunks + chunk) * J + tid) * 2;
        po[0] = M; po[1] = S;
    }
    if (track && tid == 0) x.dupart[((int64_t)parity * C + c) * n_chunks + chunk] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// ---- resident form (J <= 64): the WHOLE E/M loop in one launch.  A cloud is worked on by its n_chunks workgroups of 256 rows, which stay
// on the chip for all outer iterations and meet at a per-cloud barrier in global memory after every sweep (column partials) and every M-step
// (partial sums): 110 barriers of 2-3 us instead of 120 launches whose drain + dispatch costs ~8 us each, and the 64 distances of a row
// are computed once per outer iteration (registers) instead of once per sweep -- a third of a sweep's vector instructions.
//   * Who works on what is decided by a ticket taken at workgroup start: ticket t -> cloud t / n_chunks, chunk t % n_chunks.  Tickets are
//     handed out in order to workgroups that are running, so the lowest unfinished cloud always has all its chunks resident and makes
//     progress whatever else shares the chip (the GEMM stream next to it, grids larger than the chip): no co-residency assumption.
//   * Data that crosses workgroups (column partials, M-step partial sums) is written and read with agent-scope atomic accesses (sc1:
//     coherent across the XCDs' L2s without fences); the barrier is: stores acknowledged (an explicit s_waitcnt vmcnt(0): hipcc's
//     __syncthreads() waits for the LDS counter only on this target -- without the wait 4-5 of 100 clouds read stale partials) ->
//     workgroup barrier -> one relaxed atomic add per workgroup -> poll until all chunks have arrived.  A poll limit turns a lost workgroup into an error flag
//     (ws.sync[1]) instead of a hang.
// Arithmetic per entry is that of the launch sequence; the M-step sums the chunks' fp64 partials in chunk order.
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void cloud_barrier(int* arrivals, int target, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's exchanged values have been written through
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++polls > (1 << 26)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

template <int JMAX, bool FULL>
__global__ __launch_bounds__(256, 3) void em_resident_kernel(const float* __restrict__ xyz, int C, int N, int J, int n_chunks, int iters, int sk_iters,
                                                              float inv_tau, float inv_eps, float eps, float logq, EmWs w,
                                                              float* __restrict__ gamma_out, float* __restrict__ pi_out, float* __restrict__ mu_out, EmExit x) {
    __shared__ float vs[JMAX];
    __shared__ float vsp[JMAX];                      // v of the sweep before (early exit: gmm_exit.h)
    __shared__ float4 mus[JMAX];
    __shared__ float tile[4][64][33];
    __shared__ float wp[4][JMAX][2];
    __shared__ float pxyz[4][64][3];
    __shared__ float sred[4];
    __shared__ int s_stop;
    __shared__ int s_ticket;
    const bool track = x.on || x.resid != nullptr;
    double (*wd)[JMAX][4] = reinterpret_cast<double (*)[JMAX][4]>(&tile[0][0][0]);      // [4][JMAX][4], over the tiles once every wave is done with them (keeps the kernel at 40 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(w.sync, 1);
    __syncthreads();
    const int c = s_ticket / n_chunks, chunk = s_ticket % n_chunks;
    if (c >= C) return;
    int* arrivals = w.sync + 2 + c;
    int epoch = 0;
    const int n = chunk * 256 + tid;
    const bool valid = n < N;
    const float* __restrict__ pt = xyz + ((int64_t)c * N + (valid ? n : 0)) * 3;
    const float px = pt[0], py = pt[1], pz = pt[2], pn = sqnorm3(px, py, pz);
    const float logp = valid ? w.logp[(int64_t)c * N + n] : 0.0f;
    pxyz[wave][lane][0] = px; pxyz[wave][lane][1] = py; pxyz[wave][lane][2] = pz;
    for (int j = tid; j < J; j += 256) mus[j] = w.mu[(int64_t)c * J + j];          // em_init_kernel's centres (an earlier launch)
    const int64_t psz = (int64_t)C * n_chunks * J * 2;
    const int col = lane & 31, rh = lane >> 5;

    for (int it = 0; it < iters; ++it) {
        __syncthreads();                                                           // mus complete
        float cst[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const float4 m = mus[(FULL || j < J) ? j : 0];
            cst[j] = (FULL || j < J) ? cdist_mm2(px, py, pz, pn, m.x, m.y, m.z, m.w) * inv_tau : 0.0f;
        }
        float un = 0.0f, un_prev = 0.0f;
        for (int j = tid; j < J; j += 256) vs[j] = 0.0f;
        // sweep k = 1 .. sk_iters, then the gamma pass as sweep sk_iters + 1: each starts by finishing the pending v-update from the previous
        // sweep's column partials (buffer parity (k - 1) & 1)
        for (int k = 1; k <= sk_iters + 1; ++k) {
            if (k > 1) {
                const float* __restrict__ part = w.pbuf + ((k - 1) & 1) * psz;
                float dv_part = 0.0f;
                for (int j = tid; j < J; j += 256) {
                    const float* __restrict__ pj = part + ((int64_t)c * n_chunks * J + j) * 2;
                    float M = -__builtin_inff();
                    for (int ch = 0; ch < n_chunks; ++ch) M = fmaxf(M, ld_agent(pj + (int64_t)ch * J * 2));
                    float S = 0.0f;
                    for (int ch = 0; ch < n_chunks; ++ch) S = fmaf(ld_agent(pj + (int64_t)ch * J * 2 + 1), expf(ld_agent(pj + (int64_t)ch * J * 2) - M), S);
                    const float vold = vs[j], vnew = eps * (logq - (M + logf(S))) + vold;
                    vsp[j] = vold;
                    vs[j] = vnew;
                    dv_part += fabsf(vnew - vold);
                }
                // Early exit: the residual of sweep k - 1 is complete here (chunk 0 publishes it); the group's decision about sweep k - 2 was taken
                // while this cloud ran sweep k - 1 -- if it ended the E-step, (u_{k-2}, v_{k-2}) = (un_prev, vsp) is the state.
                if (track && wave == 0) {
                    if (chunk == 0) {
                        const float dv = wave_sum(dv_part);          // (J <= 64: all of it sits in wave 0)
                        float r = 0.0f;
                        if (lane == 0) {
                            float du = 0.0f;
                            const float* __restrict__ dp = x.dupart + ((int64_t)((k - 1) & 1) * C + c) * n_chunks;
                            for (int ch = 0; ch < n_chunks; ++ch) du += em_ld_agent(dp + ch);
                            r = du + dv;
                            if (x.resid) x.resid[((int64_t)c * x.iters + it) * x.sk + (k - 2)] = r;
                        }
                        if (x.on && k - 1 < sk_iters) em_exit_publish_wave(x, c, it, k - 2, r);
                    }
                    if (x.on && k >= 3 && lane == 0) s_stop = em_exit_wait(x, c, it, k - 3) ? 1 : 0;
                }
            }
            __syncthreads();
            if (x.on && k >= 3 && s_stop) {
                un = un_prev;
                for (int j = tid; j < J; j += 256) vs[j] = vsp[j];
                if (x.resid && chunk == 0 && tid == 0) x.resid[((int64_t)c * x.iters + it) * x.sk + (k - 2)] = __builtin_nanf("");          // (the discarded sweep)
                __syncthreads();
                break;
            }
            if (k == sk_iters + 1) break;
            float mx = -__builtin_inff();
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                if (FULL || j < J) mx = fmaxf(mx, ((-cst[j] + un) + vs[j]) * inv_eps);
            float se = 0.0f;
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                if (FULL || j < J) se += fexp(((-cst[j] + un) + vs[j]) * inv_eps - mx);
            const float unew = valid ? eps * (logp - (mx + logf(se))) + un : 0.0f;
            if (track) {                                                           // this chunk's sum |du| of sweep k
                const float du = wave_sum(valid ? fabsf(unew - un) : 0.0f);
                if (lane == 0) sred[wave] = du;
            }
            un_prev = un;
            un = unew;
#pragma unroll
            for (int h = 0; h < JMAX / 32; ++h) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int j = 32 * h + i;
                    tile[wave][lane][i] = (valid && (FULL || j < J)) ? ((-cst[j] + unew) + vs[j]) * inv_eps : -__builtin_inff();
                }
                float cm = -__builtin_inff();
                float y[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) { y[r] = tile[wave][rh * 32 + r][col]; cm = fmaxf(cm, y[r]); }
                float cs = 0.0f;
#pragma unroll
                for (int r = 0; r < 32; ++r) cs += fexp(y[r] - cm);
                if (!(cm > -__builtin_inff())) cs = 0.0f;
                const float om = __shfl_xor(cm, 32, 64), os = __shfl_xor(cs, 32, 64);
                const float M = fmaxf(cm, om);
                const float S = (cm > -__builtin_inff() ? cs * fexp(cm - M) : 0.0f) + (om > -__builtin_inff() ? os * fexp(om - M) : 0.0f);
                if (lane < 32) { wp[wave][32 * h + col][0] = M; wp[wave][32 * h + col][1] = S; }
            }
            __syncthreads();
            if (tid < J) {
                float M = fmaxf(fmaxf(wp[0][tid][0], wp[1][tid][0]), fmaxf(wp[2][tid][0], wp[3][tid][0]));
                float S = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (wp[q][tid][0] > -__builtin_inff()) S = fmaf(wp[q][tid][1], fexp(wp[q][tid][0] - M), S);
                float* __restrict__ po = w.pbuf + (k & 1) * psz + (((int64_t)c * n_chunks + chunk) * J + tid) * 2;
                st_agent(po, M); st_agent(po + 1, S);
            }
            if (track && tid == 0) em_st_agent(x.dupart + ((int64_t)(k & 1) * C + c) * n_chunks + chunk, (sred[0] + sred[1]) + (sred[2] + sred[3]));
            cloud_barrier(arrivals, (++epoch) * n_chunks, w.sync + 1);
        }
        // ---- gamma (exp(K), nan -> 0, inf -> FLT_MAX), row clip, and the chunk's M-step sums per column in fp64
        double rs = 0.0;
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            if (FULL || j < J) {
                float g = expf(((-cst[j] + un) + vs[j]) * inv_eps);
                g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                rs += (double)g;
            }
        const float rc = fmaxf((float)rs, 1e-3f);
        const bool last = it + 1 == iters;
        float* __restrict__ grow = gamma_out + ((int64_t)c * N + (valid ? n : 0)) * J;
        double msum[JMAX / 32][4];
#pragma unroll
        for (int h = 0; h < JMAX / 32; ++h) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int j = 32 * h + i;
                float g = 0.0f;
                if (valid && (FULL || j < J)) {
                    g = expf(((-cst[j] + un) + vs[j]) * inv_eps);
                    g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                    g = g / rc;
                    if (last) grow[j] = g;
                }
                tile[wave][lane][i] = g;
            }
            double sg = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const float g = tile[wave][rh * 32 + r][col];
                sg += g;
                sx += (double)g * pxyz[wave][rh * 32 + r][0]; sy += (double)g * pxyz[wave][rh * 32 + r][1]; sz += (double)g * pxyz[wave][rh * 32 + r][2];
            }
            sg += __shfl_xor(sg, 32, 64); sx += __shfl_xor(sx, 32, 64); sy += __shfl_xor(sy, 32, 64); sz += __shfl_xor(sz, 32, 64);
            msum[h][0] = sg; msum[h][1] = sx; msum[h][2] = sy; msum[h][3] = sz;
        }
        __syncthreads();                                                           // every wave is done with its tile: the sums go on top
        if (lane < 32) {
#pragma unroll
            for (int h = 0; h < JMAX / 32; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) wd[wave][32 * h + col][i] = msum[h][i];
        }
        __syncthreads();
        if (tid < J) {
            double* __restrict__ mo = w.mpart + (((int64_t)c * n_chunks + chunk) * J + tid) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) st_agent(mo + i, (wd[0][tid][i] + wd[1][tid][i]) + (wd[2][tid][i] + wd[3][tid][i]));
        }
        cloud_barrier(arrivals, (++epoch) * n_chunks, w.sync + 1);
        if (tid < J) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            for (int ch = 0; ch < n_chunks; ++ch) {
                const double* __restrict__ mi = w.mpart + (((int64_t)c * n_chunks + ch) * J + tid) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] += ld_agent(mi + i);
            }
            const float pj = (float)t[0] / (float)N;
            const float npi = pj * (float)N + 1e-5f;
            const float nx = (float)t[1] / npi, ny = (float)t[2] / npi, nz = (float)t[3] / npi;
            mus[tid] = make_float4(nx, ny, nz, sqnorm3(nx, ny, nz));
            if (last && chunk == 0) {
                pi_out[(int64_t)c * J + tid] = pj;
                float* mo = mu_out + ((int64_t)c * J + tid) * 3;
                mo[0] = nx; mo[1] = ny; mo[2] = nz;
            }
        }
        // (the next M-step's partials are written only after 1 + sk_iters more barriers: nobody still reads this iteration's)
    }
    // a poll ran into its limit somewhere (a lost workgroup): make the result loudly wrong instead of silently so
    if (chunk == 0 && tid < J && (em_ld_agent(w.sync + 1) != 0 || (x.on && em_ld_agent(x.err) != 0))) {
        pi_out[(int64_t)c * J + tid] = __builtin_nanf("");
        mu_out[((int64_t)c * J + tid) * 3] = __builtin_nanf("");
    }
}

// ---- gamma + M-step of the fused launch sequence in one pass (J <= 64): the gamma kernel below writes the unnormalised gamma of every entry to
// the workspace (64 MB at 128 clouds of 2048 x 64) for a second kernel, one workgroup per (cluster, cloud), to read it back column-wise.
// Here a row chunk forms gamma / rowclip, writes it out only in the last outer iteration, and leaves its fp64 column sums {gamma, gamma x, y, z}
// per cluster (through the wave's LDS tile, as the sweeps do for their column partials); em_mu_kernel adds the chunks up in chunk order.
// Same arithmetic and summation order as em_resident_kernel.
template <int JMAX>
__global__ __launch_bounds__(256, 3) void em_gamma_mstep_kernel(const float* __restrict__ xyz, float inv_tau, int N, int J, float inv_eps, float eps,
                                                                 float logq, int it, int sk_iters, EmWs w, float* __restrict__ gamma_out, EmExit x) {
    __shared__ float vs[JMAX];
    __shared__ float4 mus[JMAX];
    __shared__ float tile[4][64][33];
    __shared__ float pxyz[4][64][3];
    double (*wd)[JMAX][4] = reinterpret_cast<double (*)[JMAX][4]>(&tile[0][0][0]);
    const int c = blockIdx.y, chunk = blockIdx.x, n_chunks = gridDim.x, C = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = chunk * 256 + tid;
    const int64_t vsz = (int64_t)C * J, psz = (int64_t)C * n_chunks * J * 2;
    // plays launch sk_iters + 1 for the pending v-update -- unless sweep ks < sk_iters ended the E-step (gmm_exit.h): then (u_ks, v_ks), both still
    // in their parity buffers, are the state
    const int first = sk_iters == 0, parity = (sk_iters + 1) & 1;
    const int ks = x.on ? x.kstop[(c / x.G) * x.iters + it] : 0;
    if (ks) {
        for (int j = tid; j < J; j += 256) vs[j] = w.vbuf[(ks & 1) * vsz + (int64_t)c * J + j];
    } else {
        const float dv_part = em_finish_v(c, J, n_chunks, first != 0, eps, logq, w.vbuf + parity * vsz, w.pbuf + (parity ^ 1) * psz, nullptr, vs);
        if (x.resid && chunk == 0 && sk_iters >= 1 && wave == 0) em_fused_residual(x, c, it, sk_iters + 1, n_chunks, dv_part, false);
    }
    const int u_par = ks ? (ks & 1) : (sk_iters & 1);
    const float* __restrict__ u_in = (x.on && u_par) ? x.u2 : w.u;
    for (int j = tid; j < J; j += 256) mus[j] = w.mu[(int64_t)c * J + j];
    const bool valid = n < N;
    const float* __restrict__ pt = xyz + ((int64_t)c * N + (valid ? n : 0)) * 3;
    const float px = pt[0], py = pt[1], pz = pt[2], pn = sqnorm3(px, py, pz);
    pxyz[wave][lane][0] = px; pxyz[wave][lane][1] = py; pxyz[wave][lane][2] = pz;
    __syncthreads();
    float cst[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const float4 m = mus[j < J ? j : 0];
        cst[j] = j < J ? cdist_mm2(px, py, pz, pn, m.x, m.y, m.z, m.w) * inv_tau : 0.0f;
    }
    const float un = (valid && !first) ? u_in[(int64_t)c * N + n] : 0.0f;
    double rs = 0.0;
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
        if (j < J) {
            float g = expf(((-cst[j] + un) + vs[j]) * inv_eps);
            g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
            rs += (double)g;
        }
    const float rc = fmaxf((float)rs, 1e-3f);
    float* __restrict__ grow = gamma_out ? gamma_out + ((int64_t)c * N + (valid ? n : 0)) * J : nullptr;
    const int col = lane & 31, rh = lane >> 5;
    double msum[JMAX / 32][4];
#pragma unroll
    for (int h = 0; h < JMAX / 32; ++h) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = 32 * h + i;
            float g = 0.0f;
            if (valid && j < J) {
                g = expf(((-cst[j] + un) + vs[j]) * inv_eps);
                g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                g = g / rc;
                if (grow) grow[j] = g;
            }
            tile[wave][lane][i] = g;
        }
        double sg = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const float g = tile[wave][rh * 32 + r][col];
            sg += g;
            sx += (double)g * pxyz[wave][rh * 32 + r][0]; sy += (double)g * pxyz[wave][rh * 32 + r][1]; sz += (double)g * pxyz[wave][rh * 32 + r][2];
        }
        sg += __shfl_xor(sg, 32, 64); sx += __shfl_xor(sx, 32, 64); sy += __shfl_xor(sy, 32, 64); sz += __shfl_xor(sz, 32, 64);
        msum[h][0] = sg; msum[h][1] = sx; msum[h][2] = sy; msum[h][3] = sz;
    }
    __syncthreads();                                                           // every wave is done with its tile: the sums go on top
    if (lane < 32) {
#pragma unroll
        for (int h = 0; h < JMAX / 32; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) wd[wave][32 * h + col][i] = msum[h][i];
    }
    __syncthreads();
    if (tid < J) {
        double* __restrict__ mo = w.mpart + (((int64_t)c * n_chunks + chunk) * J + tid) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) mo[i] = (wd[0][tid][i] + wd[1][tid][i]) + (wd[2][tid][i] + wd[3][tid][i]);
    }
}

// pi_j, mu_j from the chunks' partial sums (chunk order).  grid (C), 64 threads
__global__ __launch_bounds__(64) void em_mu_kernel(int N, int J, int n_chunks, EmWs w, float* __restrict__ pi_out, float* __restrict__ mu_out) {
    const int c = blockIdx.x, j = threadIdx.x;
    if (j >= J) return;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int ch = 0; ch < n_chunks; ++ch) {
        const double* __restrict__ mi = w.mpart + (((int64_t)c * n_chunks + ch) * J + j) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] += mi[i];
    }
    const float pj = (float)t[0] / (float)N;
    const float npi = pj * (float)N + 1e-5f;
    const float nx = (float)t[1] / npi, ny = (float)t[2] / npi, nz = (float)t[3] / npi;
    w.mu[(int64_t)c * J + j] = make_float4(nx, ny, nz, sqnorm3(nx, ny, nz));
    if (pi_out) {
        pi_out[(int64_t)c * J + j] = pj;
        float* mo = mu_out + ((int64_t)c * J + j) * 3;
        mo[0] = nx; mo[1] = ny; mo[2] = nz;
    }
}

// gamma = exp(K) (nan -> 0, inf -> FLT_MAX) in place; rclip = max(rowsum, 1e-3); the last iteration also writes gamma / rclip.  grid (N/256, C)
// (two-launch sweeps only: u, v are complete in the workspace, also after an early exit)
__global__ __launch_bounds__(256) void em_gamma_kernel(int N, int J, float inv_eps, EmWs w, float* __restrict__ gamma_out) {
    extern __shared__ float vs[];                    // [J]
    const int c = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    for (int j = threadIdx.x; j < J; j += 256) vs[j] = w.v[(int64_t)c * J + j];
    __syncthreads();
    if (n >= N) return;
    float* __restrict__ Cc = w.cost + (int64_t)c * J * N + n;
    const float un = w.u[(int64_t)c * N + n];
    double rs = 0.0;
    for (int j = 0; j < J; ++j) {
        float g = expf(((-Cc[(int64_t)j * N] + un) + vs[j]) * inv_eps);
        g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
        Cc[(int64_t)j * N] = g;
        rs += (double)g;
    }
    const float rc = fmaxf((float)rs, 1e-3f);
    w.rclip[(int64_t)c * N + n] = rc;
    if (gamma_out) {
        float* __restrict__ grow = gamma_out + ((int64_t)c * N + n) * J;
        for (int j = 0; j < J; ++j) grow[j] = Cc[(int64_t)j * N] / rc;
    }
}

// M-step: pi_j = mean_n gamma, mu_j = gamma^T xyz / (N pi + 1e-5), fp64 column sums.  grid (J, C)
__global__ __launch_bounds__(256) void em_mstep_kernel(const float* __restrict__ xyz, int N, int J, EmWs w, float* __restrict__ pi_out,
                                                       float* __restrict__ mu_out) {
    __shared__ double red[4][4];
    const int j = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ Gj = w.cost + ((int64_t)c * J + j) * N;
    const float* __restrict__ rc = w.rclip + (int64_t)c * N;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    double sg = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    for (int n = tid; n < N; n += 256) {
        const float g = Gj[n] / rc[n];
        sg += g;
        sx += (double)g * cloud[3 * n]; sy += (double)g * cloud[3 * n + 1]; sz += (double)g * cloud[3 * n + 2];
    }
    sg = wave_sum_d(sg); sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
    if (lane == 0) { red[wave][0] = sg; red[wave][1] = sx; red[wave][2] = sy; red[wave][3] = sz; }
    __syncthreads();
    if (tid == 0) {
        double t[4];
        for (int i = 0; i < 4; ++i) t[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        const float pj = (float)t[0] / (float)N;
        const float npi = pj * (float)N + 1e-5f;
        const float nx = (float)t[1] / npi, ny = (float)t[2] / npi, nz = (float)t[3] / npi;
        w.mu[(int64_t)c * J + j] = make_float4(nx, ny, nz, sqnorm3(nx, ny, nz));
        if (pi_out) {
            pi_out[(int64_t)c * J + j] = pj;
            float* mo = mu_out + ((int64_t)c * J + j) * 3;
            mo[0] = nx; mo[1] = ny; mo[2] = nz;
        }
    }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" int64_t ogmm_gmm_em_workspace_bytes(int C, int N, int J) {
    const size_t chunks = (size_t)(N + 255) / 256;
    return (int64_t)(align256((size_t)C * J * N * 4) + 3 * align256((size_t)C * N * 4) + align256((size_t)C * J * 4) + align256((size_t)C * J * 16) +
                     align256((size_t)2 * C * J * 4) + align256((size_t)2 * C * chunks * J * 8) + align256((size_t)C * chunks * J * 32) +
                     align256((size_t)(C + 2) * 4));
}

extern "C" int ogmm_gmm_em_multi(const float* xyz, const float* o, const int32_t* ids0, int C, int N, int J, int iters, int sk_iters,
                                 float epsilon, float tau, double thresh, int group_size, float* gamma, float* pi, float* mu, float* resid,
                                 int32_t* sweeps, void* exit_ws, void* workspace, void* stream) {
    using namespace ogmm;
    OGMM_REQUIRE(xyz && o && ids0 && gamma && pi && mu && workspace, "ogmm_gmm_em_multi: null pointer");
    OGMM_REQUIRE(J <= 128, "ogmm_gmm_em_multi: at most 128 clusters (a row of exponents lives in registers), got %d", J);
    OGMM_REQUIRE(C > 0 && C <= 65535 && N > 0 && J > 0 && J <= N && J <= 65535 && iters > 0 && sk_iters >= 0 && epsilon > 0 && tau > 0,
                 "ogmm_gmm_em_multi: bad sizes C=%d N=%d J=%d", C, N, J);
    OGMM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "ogmm_gmm_em_multi: workspace must be 256-byte aligned");
    hipStream_t s = as_stream(stream);
    EmExit x;
    if (int rc = em_exit_setup(x, thresh, group_size, C, N, iters, sk_iters, resid, sweeps, exit_ws, s)) return rc;
    char* p = static_cast<char*>(workspace);
    EmWs w;
    w.cost = reinterpret_cast<float*>(p);  p += align256((size_t)C * J * N * 4);
    w.u = reinterpret_cast<float*>(p);     p += align256((size_t)C * N * 4);
    w.logp = reinterpret_cast<float*>(p);  p += align256((size_t)C * N * 4);
    w.rclip = reinterpret_cast<float*>(p); p += align256((size_t)C * N * 4);
    w.v = reinterpret_cast<float*>(p);     p += align256((size_t)C * J * 4);
    w.mu = reinterpret_cast<float4*>(p);   p += align256((size_t)C * J * 16);
    const size_t n_chunks_ws = (size_t)(N + 255) / 256;
    w.vbuf = reinterpret_cast<float*>(p);  p += align256((size_t)2 * C * J * 4);
    w.pbuf = reinterpret_cast<float*>(p);  p += align256((size_t)2 * C * n_chunks_ws * J * 8);
    w.mpart = reinterpret_cast<double*>(p); p += align256((size_t)C * n_chunks_ws * J * 32);
    w.sync = reinterpret_cast<int*>(p);
    const float inv_eps = (float)(1.0 / (double)epsilon), inv_tau = (float)(1.0 / (double)tau);
    const float logq = logf((float)(1.0 / (double)J) + 1e-8f);
    const dim3 rows((N + 255) / 256, C), cols(J, C), blk(256);
    const size_t vs = (size_t)(J + 4) * sizeof(float);
    hipLaunchKernelGGL(em_init_kernel, dim3(C), blk, 0, s, xyz, o, ids0, N, J, w);
    // Measured (N = 2048, J = 64; resident / launch sequence): 4 clouds 1.15 / 2.7 ms, 96 clouds 3.0 / 3.25 (3 workgroups per CU, all resident),
    // 128 clouds 4.0 / 3.7 (4 per CU: spills) or 5.1 (3 per CU: a third of the clouds wait for a second round).  A full chip is bound by the
    // sweeps' vector instructions either way, so the resident form serves the grids that fit one resident round -- most of all the small
    // batches, where the chain of 120 launches is pure latency.  OGMM_EM_RESIDENT=0 / =1 force the launch sequence / the resident kernel.
    const char* res_env = getenv("OGMM_EM_RESIDENT");                      // (read per call: the tests switch it)
    const int resident = res_env ? (res_env[0] == '0' ? 0 : 2) : 1;
    int cus = 256, dev_id = 0;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    // one resident round: three workgroups per CU (FULL form, J == 64: 48 / 64 / 96 clouds 2.09 / 2.55 / 3.00 ms against 2.86 / 2.87 / 3.25).
    // With the early exit on, the clouds of a call group wait for each other: a whole group must fit one resident round (tickets are handed
    // out in order, so the lowest unfinished group is then always completely on the chip), otherwise the launch sequence runs.
    const bool group_fits = !x.on || (int64_t)x.G * (int64_t)n_chunks_ws <= 3 * (int64_t)cus;
    if (J <= 64 && group_fits && (resident == 2 || (resident == 1 && (int64_t)C * (int64_t)n_chunks_ws <= 3 * (int64_t)cus))) {
        (void)hipMemsetAsync(w.sync, 0, (size_t)(C + 2) * 4, s);
        const dim3 grid((unsigned)(C * n_chunks_ws));
        #define OGMM_EM_RES(JM, FL) hipLaunchKernelGGL((em_resident_kernel<JM, FL>), grid, blk, 0, s, xyz, C, N, J, (int)n_chunks_ws, iters, sk_iters, inv_tau, inv_eps, epsilon, logq, w, gamma, pi, mu, x)
        if (J == 32) OGMM_EM_RES(32, true); else if (J < 32) OGMM_EM_RES(32, false); else if (J == 64) OGMM_EM_RES(64, true); else OGMM_EM_RES(64, false);
#undef OGMM_EM_RES
        return check_launch("ogmm_gmm_em_multi");
    }
    for (int it = 0; it < iters; ++it) {
        const bool last = it + 1 == iters;
        static const bool two_launch = [] { const char* e = getenv("OGMM_EM_MULTI_UNFUSED"); return e && e[0] == '1'; }();      // A/B: u and v kernels
        const bool fused = J <= 64 && !two_launch;
        if (fused) {
            // launch m = 1 .. sk_iters (the fused sweeps recompute the costs); the gamma kernel plays launch sk_iters + 1 for the pending v-update
            for (int m = 1; m <= sk_iters; ++m) {
#define OGMM_EM_SWEEP(JM, FL) hipLaunchKernelGGL((em_sweep_kernel<JM, FL>), rows, blk, 0, s, xyz, inv_tau, N, J, inv_eps, epsilon, logq, it, m, w, x)
                // (the FULL form of this kernel keeps all 64 exponents of a row between its passes: 171 spills under the 128-register cap of four
                // workgroups per CU, 6.9 against 3.7 ms -- the run-time tests stay here; the resident kernel, 168 registers, takes it: 1.15 against 1.6 ms)
                if (J <= 32) OGMM_EM_SWEEP(32, false); else OGMM_EM_SWEEP(64, false);
#undef OGMM_EM_SWEEP
            }
            if (J <= 32) hipLaunchKernelGGL(em_gamma_mstep_kernel<32>, rows, blk, 0, s, xyz, inv_tau, N, J, inv_eps, epsilon, logq, it, sk_iters, w, last ? gamma : (float*)nullptr, x);
            else hipLaunchKernelGGL(em_gamma_mstep_kernel<64>, rows, blk, 0, s, xyz, inv_tau, N, J, inv_eps, epsilon, logq, it, sk_iters, w, last ? gamma : (float*)nullptr, x);
            hipLaunchKernelGGL(em_mu_kernel, dim3(C), dim3(64), 0, s, N, J, (int)n_chunks_ws, w, last ? pi : (float*)nullptr, mu);
            continue;
        }
        hipLaunchKernelGGL(em_cost_kernel, rows, blk, 0, s, xyz, N, J, inv_tau, w);
        for (int m = 1; m <= sk_iters; ++m) {
            if (J <= 16) hipLaunchKernelGGL(em_u_kernel<16>, rows, blk, vs, s, N, J, inv_eps, epsilon, w, it, x);
            else if (J <= 64) hipLaunchKernelGGL(em_u_kernel<64>, rows, blk, vs, s, N, J, inv_eps, epsilon, w, it, x);
            else hipLaunchKernelGGL(em_u_kernel<128>, rows, blk, vs, s, N, J, inv_eps, epsilon, w, it, x);
            hipLaunchKernelGGL(em_v_kernel, cols, blk, 0, s, N, J, inv_eps, epsilon, logq, w, it, m, (int)n_chunks_ws, x);
        }
        hipLaunchKernelGGL(em_gamma_kernel, rows, blk, vs, s, N, J, inv_eps, w, last ? gamma : (float*)nullptr);
        hipLaunchKernelGGL(em_mstep_kernel, cols, blk, 0, s, xyz, N, J, w, last ? pi : (float*)nullptr, mu);
    }
    return check_launch("ogmm_gmm_em_multi");
}
