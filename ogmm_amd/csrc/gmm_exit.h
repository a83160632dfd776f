// Sinkhorn early exit of the reference (lib/utils.py:99-102) for the E/M kernels of gmm.hip and gmm_em_multi.hip.
//
// Reference semantics: `sinkhorn` is called once per E-step for ALL clouds of one `wkeans_plus` call (the src call and the tgt call of
// models/gmmreg.py:100-101 are separate batches of B clouds each); after every sweep it forms diff_c = sum |u - u0| + sum |v - v0| per cloud,
// takes the mean over the clouds of the call and leaves the sweeps when that mean is < thresh (the sweep that produced it stays applied).
// So the clouds of one call ("group") are coupled: they all run the same number of sweeps in a given E-step.
//
// Here a cloud is one workgroup (gmm.hip), several workgroups of one launch (em_resident_kernel) or several workgroups of a launch
// sequence (gmm_em_multi.hip); the coupling goes through global memory and is LAGGED by one sweep so that nobody waits in the common
// case: the decision about sweep k is taken at the end of sweep k + 1 (by then every cloud of the group has normally published its
// residual of sweep k), and a positive decision rolls the state back to (u_k, v_k), of which every kernel keeps a copy.  The result is
// exactly the reference's: sweeps 1..k applied, sweep k + 1 discarded.
//   * on-chip / resident kernels: residuals are published with agent-scope stores, arrivals counted per (group, E-step, sweep); the LAST
//     arriver sums the group's residuals in cloud order (deterministic), compares the mean and publishes the decision word the others poll.
//     Clouds are handed out by ticket in order of workgroup start, so the lowest unfinished group is always completely resident provided
//     one resident round holds a whole group (checked on the host) -- no assumption about dispatch order.
//   * launch sequences: kernel boundaries order everything; launch k + 2 reads the residuals of sweep k.
#pragma once
#include "ogmm_common.h"

namespace ogmm {

struct EmExit {
    double thresh;      // the reference's `thresh` (python float); <= 0: exit off, every sweep runs
    int on;             // thresh > 0
    int G, n_groups;    // clouds per call group, number of groups (C = G * n_groups)
    int C, iters, sk;
    int* ticket;        // [1]  on-chip kernels: next cloud to hand out
    int* err;           // [1]  set when a poll ran into its limit (a lost workgroup): the outputs are NaN-poisoned; word 1 of the exit workspace, which the
                        //      host reads behind the call (ogmm_amd/ops.py gmm_em: `protocol_error`)
    int poll_limit;     // polls before a wait gives up (2^24; OGMM_EM_POLL_LIMIT lowers it for the test of this path)
    int lose_cloud;     // debug knob OGMM_EM_DEBUG_LOSE_CLOUD: this cloud never publishes its residuals, as a lost workgroup would not (-1: off)
    int* gcount;        // [n_groups][iters][sk]  arrivals
    int* decision;      // [n_groups][iters][sk]  0 pending, 1 go on, 2 stop after this sweep
    int* kstop;         // [n_groups][iters]      1-based sweep after which the E-step stopped (0: it did not); the launch sequences' flag
    int* ccount;        // [iters][sk][C]         unfused launch sequence: arrivals of a cloud's J column workgroups
    float* rc;          // [iters][sk][C]         per-cloud residual of every sweep
    float* dupart;      // [2][C][chunks]         launch sequences: per-chunk sum |du| of the last two sweeps
    float* u2;          // [C][N]                 launch sequences: the second u buffer (u_k and u_{k+1} both live)
    int32_t* sweeps;    // [n_groups][iters] or NULL: sweeps the E-step ran (the reference's iteration count), pre-filled with sk
    float* resid;       // [C][iters][sk] or NULL: every sweep's residual per cloud (NaN: sweep not run)
};

__device__ __forceinline__ float em_ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void em_st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int em_ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void em_st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the group's batch mean of sweep k (0-based) against the threshold, clouds summed in index order
__device__ __forceinline__ bool em_exit_mean_below(const EmExit& x, int g, int it, int k, bool agent) {
    const float* __restrict__ rc = x.rc + ((int64_t)it * x.sk + k) * x.C + (int64_t)g * x.G;
    float s = 0.0f;
    for (int i = 0; i < x.G; ++i) s += agent ? em_ld_agent(rc + i) : rc[i];
    const float mean = s / (float)x.G;
    return (double)mean < x.thresh;          // `mean_diff.item() < thresh`: a python double comparison of the fp32 mean
}

// One lane per cloud and sweep: publish this cloud's residual of sweep k (0-based) of E-step `it`.  The last cloud of the group to arrive takes the decision.
__device__ __forceinline__ void em_exit_publish(const EmExit& x, int c, int it, int k, float r) {
    if (c == x.lose_cloud) return;
    const int g = c / x.G;
    em_st_agent(x.rc + ((int64_t)it * x.sk + k) * x.C + c, r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // written through before the arrival is counted
    const int slot = (g * x.iters + it) * x.sk + k;
    const int prev = __hip_atomic_fetch_add(x.gcount + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == x.G - 1) {
        const bool stop = em_exit_mean_below(x, g, it, k, true);
        // With the lag, the sweep that a stop discards has usually been published as well (and is below the threshold too): the FIRST stop of an
        // E-step counts.  Its writer has stored kstop before it arrives at any later sweep's counter, so the later decision maker sees it.
        if (stop && em_ld_agent(x.kstop + g * x.iters + it) == 0) {
            em_st_agent(x.kstop + g * x.iters + it, k + 1);
            if (x.sweeps) x.sweeps[g * x.iters + it] = k + 1;
        }
        em_st_agent(x.decision + slot, stop ? 2 : 1);
    }
}

// On-chip kernels, all 64 lanes of one wave: the group's decision about sweep k (0-based) straight from the published residuals -- no arrival counter
// and no decision word, so a decision costs one store and one (polled) load instead of five dependent trips to memory.  Every cloud of the group reads the
// group's G residuals (one per lane and round of 64), polls until none is the "not yet" pattern em_exit_setup filled in (all ones: a published residual
// has its sign bit clear, em_exit_publish_value), and forms the same sum in the same order (lane-strided partials, xor butterfly) -- so all clouds take
// the same decision.  true = the E-step's sweeps end with sweep k.
__device__ __forceinline__ float em_exit_publish_value(float r) { return r == r ? fabsf(r) : __builtin_inff(); }          // NaN -> inf: "not below" either way
__device__ __forceinline__ bool em_exit_decide_wave(const EmExit& x, int c, int it, int k) {
    const int lane = threadIdx.x & 63, g = c / x.G;
    const float* __restrict__ rc = x.rc + ((int64_t)it * x.sk + k) * x.C + (int64_t)g * x.G;
    float s = 0.0f;
    for (int base = 0; base < x.G; base += 64) {
        const int i = base + lane;
        float r = 0.0f;
        if (i < x.G) {
            int polls = 0;
            while (__float_as_uint(r = em_ld_agent(rc + i)) == 0xFFFFFFFFu) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > x.poll_limit) { em_st_agent(x.err, 1); r = 0.0f; break; }
            }
        }
        s += r;
    }
    s = wave_sum(s);
    const float mean = s / (float)x.G;
    return (double)mean < x.thresh;
}

// The same publication by ALL 64 lanes of one wave (r needs to be valid in lane 0 only): the last arriver's sum over the group's residuals is spread over
// the lanes (lane-strided partials, xor butterfly) -- one lane summing a 256-cloud group with dependent agent-scope loads kept the launch's last
// workgroup busy for ~0.25 ms per sweep (the launch sequence of a 256-pair batch: 16.5 against 12.9 ms without the exit).
__device__ __forceinline__ void em_exit_publish_wave(const EmExit& x, int c, int it, int k, float r) {
    if (c == x.lose_cloud) return;
    const int lane = threadIdx.x & 63, g = c / x.G;
    const int slot = (g * x.iters + it) * x.sk + k;
    int prev = 0;
    if (lane == 0) {
        em_st_agent(x.rc + ((int64_t)it * x.sk + k) * x.C + c, r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // written through before the arrival is counted
        prev = __hip_atomic_fetch_add(x.gcount + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    prev = __shfl(prev, 0, 64);
    if (prev != x.G - 1) return;
    const float* __restrict__ rc = x.rc + ((int64_t)it * x.sk + k) * x.C + (int64_t)g * x.G;
    float s = 0.0f;
    for (int i = lane; i < x.G; i += 64) s += em_ld_agent(rc + i);
    s = wave_sum(s);
    if (lane == 0) {
        const bool stop = (double)(s / (float)x.G) < x.thresh;
        if (stop && em_ld_agent(x.kstop + g * x.iters + it) == 0) {          // the FIRST stop of an E-step counts (em_exit_publish)
            em_st_agent(x.kstop + g * x.iters + it, k + 1);
            if (x.sweeps) x.sweeps[g * x.iters + it] = k + 1;
        }
        em_st_agent(x.decision + slot, stop ? 2 : 1);
    }
}

// One lane: the group's decision about sweep k (0-based): true = the E-step's sweeps end with sweep k.
__device__ __forceinline__ bool em_exit_wait(const EmExit& x, int c, int it, int k) {
    const int slot = ((c / x.G) * x.iters + it) * x.sk + k;
    int d, polls = 0;
    while ((d = em_ld_agent(x.decision + slot)) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > 4 * (int64_t)x.poll_limit) { em_st_agent(x.err, 1); return false; }
    }
    return d == 2;
}

}  // namespace ogmm

// ---------------------------------------------------------------------------------------------- host side
namespace ogmm {

static inline size_t em_exit_align(size_t x) { return (x + 255) / 256 * 256; }

static inline size_t em_exit_bytes(int C, int N, int iters, int sk, int group_size) {
    const size_t G = group_size > 0 ? (size_t)group_size : (size_t)C, ng = ((size_t)C + G - 1) / G, chunks = ((size_t)N + 255) / 256;
    const size_t ints = 64 + 2 * ng * (size_t)iters * (size_t)sk + ng * (size_t)iters + (size_t)iters * (size_t)sk * (size_t)C;
    return em_exit_align(ints * 4) + em_exit_align((size_t)iters * (size_t)sk * (size_t)C * 4) + em_exit_align(2 * (size_t)C * chunks * 4) +
           em_exit_align((size_t)C * (size_t)N * 4);
}

// Carves `ws` (em_exit_bytes, 256-byte aligned; may be NULL when thresh <= 0), zeroes the counters, pre-fills the two optional outputs.  0 / error code.
static inline int em_exit_setup(EmExit& x, double thresh, int group_size, int C, int N, int iters, int sk, float* resid, int32_t* sweeps, void* ws,
                                hipStream_t s) {
    x = EmExit{};
    x.thresh = thresh;
    x.on = thresh > 0.0 && sk > 1 ? 1 : 0;          // with one sweep there is nothing to leave early
    x.G = group_size > 0 ? group_size : C;
    OGMM_REQUIRE(C % x.G == 0, "ogmm_gmm_em: the %d clouds are not whole call groups of %d", C, x.G);
    x.n_groups = C / x.G;
    x.C = C; x.iters = iters; x.sk = sk;
    x.sweeps = sweeps; x.resid = resid;
    const char* pl = getenv("OGMM_EM_POLL_LIMIT");          // (read per call: tests of the timeout path set and clear them)
    x.poll_limit = pl && atoi(pl) > 0 ? atoi(pl) : (1 << 24);
    const char* lc = getenv("OGMM_EM_DEBUG_LOSE_CLOUD");
    x.lose_cloud = lc && lc[0] ? atoi(lc) : -1;
    if (resid) (void)hipMemsetAsync(resid, 0xFF, (size_t)C * iters * sk * sizeof(float), s);          // NaN: sweep not run
    if (sweeps) (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sweeps), sk, (size_t)x.n_groups * iters, s);
    if (!x.on && !resid) return 0;
    OGMM_REQUIRE(ws != nullptr, "ogmm_gmm_em: thresh > 0 (or resid) needs the exit workspace (ogmm_gmm_em_exit_workspace_bytes)");
    OGMM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "ogmm_gmm_em: exit workspace must be 256-byte aligned");
    const size_t ng = (size_t)x.n_groups, chunks = ((size_t)N + 255) / 256;
    const size_t ints = 64 + 2 * ng * (size_t)iters * (size_t)sk + ng * (size_t)iters + (size_t)iters * (size_t)sk * (size_t)C;
    char* p = static_cast<char*>(ws);
    int* ip = reinterpret_cast<int*>(p);
    x.ticket = ip; x.err = ip + 1;
    x.gcount = ip + 64;
    x.decision = x.gcount + ng * iters * sk;
    x.kstop = x.decision + ng * iters * sk;
    x.ccount = x.kstop + ng * iters;
    (void)hipMemsetAsync(ip, 0, ints * 4, s);
    p += em_exit_align(ints * 4);
    x.rc = reinterpret_cast<float*>(p);
    (void)hipMemsetAsync(x.rc, 0xFF, (size_t)iters * sk * C * 4, s);          // "not yet published" (em_exit_decide_wave)
    p += em_exit_align((size_t)iters * sk * C * 4);
    x.dupart = reinterpret_cast<float*>(p); p += em_exit_align(2 * (size_t)C * chunks * 4);
    x.u2 = reinterpret_cast<float*>(p);
    return 0;
}

}  // namespace ogmm
