// K15 overlap-weighted Sinkhorn k-means ("GMM E/M"), K16 cluster feature means, K17 cluster matching,
// K18 weighted Kabsch with an in-register 3x3 SVD, K19 CluLoss (nearest-point anchors + InfoNCE).
//
// The reference runs ~3000 tiny launches and 200 host syncs per forward here (lib/utils.py:269-291 calling
// :69-108 ten times, each with ten sweeps and an .item() per sweep) plus a device->host->device SVD
// (lib/se3.py:276).  Here the whole E/M loop of one cloud runs inside one workgroup with the state in
// LDS/registers, and the rigid solve runs one pair per wavefront without leaving the GPU.
#include "ogmm_common.h"
#include "gmm_exit.h"
#include <cstdlib>

namespace {

using namespace ogmm;

// ================================================================================================
// K15.  One workgroup (EM_T threads) per cloud.  Nothing of size N x J is stored: cost and kernel
// entries are recomputed from xyz (LDS), mu (LDS), u (LDS), v (LDS) whenever needed.
//   u-step: rows -> threads (row n = tid + i*EM_T), reduction over j is thread-local;
//   v-step and M-step: columns -> waves (j = wave + i*n_waves), lanes stride over rows, wave reduce.
// log-sum-exp is evaluated online (running max + rescaled sum) so each entry costs one exp.
// ================================================================================================
constexpr int EM_T = 1024;

struct LSE { float m, s; };
__device__ __forceinline__ void lse_push(LSE& a, float x) {
    if (x > a.m) { a.s = a.s * expf(a.m - x) + 1.0f; a.m = x; }
    else a.s += expf(x - a.m);
}
__device__ __forceinline__ LSE lse_merge(LSE a, LSE b) {
    const float m = fmaxf(a.m, b.m);
    if (m == -__builtin_inff()) return LSE{m, 0.0f};
    return LSE{m, a.s * expf(a.m - m) + b.s * expf(b.m - m)};
}

// torch.cdist (matmul form, lib/utils.py:280): sqrt(clamp(chain([-2x, |x|^2, 1] . [mu, 1, |mu|^2]), 0))
__device__ __forceinline__ float cdist_mm(float x, float y, float z, float xn, float mx, float my, float mz, float mn) {
    float acc = mul_rn(-2.0f * x, mx);
    acc = __fmaf_rn(-2.0f * y, my, acc);
    acc = __fmaf_rn(-2.0f * z, mz, acc);
    acc = add_rn(acc, xn);
    acc = add_rn(acc, mn);
    return sqrtf(fmaxf(acc, 0.0f));
}


// ---- Sinkhorn early exit (gmm_exit.h) in the on-chip kernels.
// Which cloud this workgroup works on: with the exit on, clouds are handed out by ticket in order of workgroup start (the lowest unfinished call
// group is then always completely on the chip: its clouds wait for each other's residuals); otherwise the block index.
__device__ __forceinline__ int em_chip_cloud(const EmExit& x, float* red) {
    if (!x.on) return blockIdx.x;
    if (threadIdx.x == 0) red[0] = __int_as_float(atomicAdd(x.ticket, 1));
    __syncthreads();
    const int c = __float_as_int(red[0]);
    __syncthreads();
    return c;
}

// End of sweep `sk` (0-based) of E-step `it`, called by the whole workgroup behind the barrier that follows the v update: u / v hold this sweep's
// result, uprev / vprev the previous sweep's, red[16 + 16 * (sk & 1) + w] wave w's share of the cloud's sum |du| + sum |dv| of this sweep (summed here in
// wave order: deterministic).  Wave 0 publishes the residual, then forms the group's decision about sweep sk - 1 (one sweep of lag: normally the
// residuals are all there); true = that sweep ended the E-step and u / v are rolled back to it.
__device__ __forceinline__ bool em_chip_sweep_end(const EmExit& x, float* red, int c, int it, int sk, int sk_iters, float* u, const float* uprev,
                                                  float* v, const float* vprev, int N, int J) {
    if (threadIdx.x < 64) {
        const float* part = red + 16 + 16 * (sk & 1);
        float r = 0.0f;
#pragma unroll
        for (int w = 0; w < EM_T / 64; ++w) r += part[w];
        int stop = 0;
        if (x.on) {
            if (sk + 1 < sk_iters && threadIdx.x == 0)          // (nobody asks about the last sweep)
                if (c != x.lose_cloud) em_st_agent(x.rc + ((int64_t)it * x.sk + sk) * x.C + c, em_exit_publish_value(r));
            if (sk >= 1) {
                stop = em_exit_decide_wave(x, c, it, sk - 1) ? 1 : 0;
                if (stop && threadIdx.x == 0 && c % x.G == 0) {          // the group's first cloud reports the count
                    const int g = c / x.G;
                    em_st_agent(x.kstop + g * x.iters + it, sk);
                    if (x.sweeps) x.sweeps[g * x.iters + it] = sk;
                }
            }
        }
        if (threadIdx.x == 0) {
            if (x.resid && !stop) x.resid[((int64_t)c * x.iters + it) * x.sk + sk] = r;          // (a discarded sweep stays NaN)
            red[2] = __int_as_float(stop);
        }
    }
    if (!x.on || sk == 0) return false;
    __syncthreads();
    const bool stop = __float_as_int(red[2]) != 0;
    if (stop) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) u[n] = uprev[n];
        for (int j = threadIdx.x; j < J; j += blockDim.x) v[j] = vprev[j];
    }
    __syncthreads();
    return stop;
}

// a poll ran into its limit somewhere (a lost workgroup): make the result loudly wrong instead of silently so
__device__ __forceinline__ void em_chip_poison(const EmExit& x, int c, int J, float* pi_out, float* mu_out) {
    if (!x.on || em_ld_agent(x.err) == 0) return;
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        pi_out[(int64_t)c * J + j] = __builtin_nanf("");
        mu_out[((int64_t)c * J + j) * 3] = __builtin_nanf("");
    }
}

__global__ __launch_bounds__(EM_T) void gmm_em_kernel(const float* __restrict__ xyz, const float* __restrict__ o,
                                                      const int32_t* __restrict__ ids0, int N, int J, int iters, int sk_iters,
                                                      float inv_eps, float eps, float inv_tau, float* __restrict__ gamma,
                                                      float* __restrict__ pi_out, float* __restrict__ mu_out, EmExit x) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* pts = reinterpret_cast<float4*>(lds);        // [N]  x, y, z, |p|^2
    const int Npad = (N + 3) / 4 * 4;                     // keeps mu 16-byte aligned
    float* u = lds + 4 * (size_t)N;                       // [N]
    float* logp = u + Npad;                               // [N]
    float* rclip = logp + Npad;                           // [N]  max(rowsum, 1e-3)
    float4* mu = reinterpret_cast<float4*>(rclip + Npad); // [J]  mx, my, mz, |mu|^2
    float* v = reinterpret_cast<float*>(mu + J);          // [J]
    float* red = v + J;                                   // [48]: scratch, then per-wave residual shares of the even / odd sweeps
    float* uprev = red + 48;                              // [N]  u, v of the previous sweep (early exit: gmm_exit.h)
    float* vprev = uprev + Npad;                          // [J]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = em_chip_cloud(x, red);
    if (c >= x.C) return;
    const bool track = x.on || x.resid != nullptr;
    constexpr int NW = EM_T / 64;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    const float* __restrict__ oc = o + (int64_t)c * N;

    // p = o / max(sum o, 1e-4)   (lib/utils.py:275-276)
    float part = 0.0f;
    for (int n = tid; n < N; n += EM_T) {
        const float x = cloud[3 * n], y = cloud[3 * n + 1], z = cloud[3 * n + 2];
        pts[n] = make_float4(x, y, z, sqnorm3(x, y, z));
        part += oc[n];
    }
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float osum = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) osum += red[w];
    osum = fmaxf(osum, 1e-4f);
    for (int n = tid; n < N; n += EM_T) logp[n] = logf(oc[n] / osum + 1e-8f);
    for (int j = tid; j < J; j += EM_T) {
        const float4 p = pts[ids0[(int64_t)c * J + j]];
        mu[j] = p;
    }
    const float logq = logf((float)(1.0 / (double)J) + 1e-8f);
    __syncthreads();

    for (int it = 0; it < iters; ++it) {
        for (int n = tid; n < N; n += EM_T) u[n] = 0.0f;
        for (int j = tid; j < J; j += EM_T) v[j] = 0.0f;
        __syncthreads();
        for (int sk = 0; sk < sk_iters; ++sk) {
            // u^{l+1}: rows on threads
            float du = 0.0f, dv = 0.0f;          // this thread's share of the sweep's residual
            for (int n = tid; n < N; n += EM_T) {
                const float4 p = pts[n];
                const float un = u[n];
                LSE a = {-__builtin_inff(), 0.0f};
                for (int j = 0; j < J; ++j) {
                    const float4 m = mu[j];
                    const float cst = cdist_mm(p.x, p.y, p.z, p.w, m.x, m.y, m.z, m.w) * inv_tau;
                    lse_push(a, ((-cst + un) + v[j]) * inv_eps);
                }
                const float un1 = eps * (logp[n] - (a.m + logf(a.s))) + un;
                u[n] = un1;
                if (track) { uprev[n] = un; du += fabsf(un1 - un); }
            }
            if (track) { du = wave_sum(du); if (lane == 0) red[16 + 16 * (sk & 1) + wave] = du; }
            __syncthreads();
            // v^{l+1}: columns on waves
            for (int j = wave; j < J; j += NW) {
                const float4 m = mu[j];
                const float vj = v[j];
                LSE a = {-__builtin_inff(), 0.0f};
                for (int n = lane; n < N; n += 64) {
                    const float4 p = pts[n];
                    const float cst = cdist_mm(p.x, p.y, p.z, p.w, m.x, m.y, m.z, m.w) * inv_tau;
                    lse_push(a, ((-cst + u[n]) + vj) * inv_eps);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    LSE b = {__shfl_xor(a.m, off, 64), __shfl_xor(a.s, off, 64)};
                    a = lse_merge(a, b);
                }
                if (lane == 0) {
                    const float vj1 = eps * (logq - (a.m + logf(a.s))) + vj;
                    v[j] = vj1;
                    if (track) { vprev[j] = vj; dv += fabsf(vj1 - vj); }
                }
            }
            if (track && lane == 0) red[16 + 16 * (sk & 1) + wave] += dv;
            __syncthreads();
            if (track && em_chip_sweep_end(x, red, c, it, sk, sk_iters, u, uprev, v, vprev, N, J)) break;
        }
        // gamma = exp(K); nan -> 0 (inf -> FLT_MAX); row scale 1 / max(rowsum, 1e-3)   (lib/utils.py:281-287)
        const bool last = it + 1 == iters;
        for (int n = tid; n < N; n += EM_T) {
            const float4 p = pts[n];
            const float un = u[n];
            float rs = 0.0f;
            for (int j = 0; j < J; ++j) {
                const float4 m = mu[j];
                const float cst = cdist_mm(p.x, p.y, p.z, p.w, m.x, m.y, m.z, m.w) * inv_tau;
                float g = expf(((-cst + un) + v[j]) * inv_eps);
                g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                rs += g;
            }
            const float rc = fmaxf(rs, 1e-3f);
            rclip[n] = rc;
            if (last) {
                float* __restrict__ grow = gamma + ((int64_t)c * N + n) * J;
                for (int j = 0; j < J; ++j) {
                    const float4 m = mu[j];
                    const float cst = cdist_mm(p.x, p.y, p.z, p.w, m.x, m.y, m.z, m.w) * inv_tau;
                    float g = expf(((-cst + un) + v[j]) * inv_eps);
                    g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                    grow[j] = g / rc;
                }
            }
        }
        __syncthreads();
        // pi = mean_n gamma; mu = gamma^T xyz / (N pi + 1e-5)   (lib/utils.py:130-140), fp64 column sums
        for (int j = wave; j < J; j += NW) {
            const float4 m = mu[j];
            const float vj = v[j];
            double sg = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
            for (int n = lane; n < N; n += 64) {
                const float4 p = pts[n];
                const float cst = cdist_mm(p.x, p.y, p.z, p.w, m.x, m.y, m.z, m.w) * inv_tau;
                float g = expf(((-cst + u[n]) + vj) * inv_eps);
                g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                g = g / rclip[n];
                sg += g;
                sx += (double)g * p.x; sy += (double)g * p.y; sz += (double)g * p.z;
            }
            sg = wave_sum_d(sg); sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
            if (lane == 0) {
                const float pj = (float)sg / (float)N;
                const float npi = pj * (float)N + 1e-5f;
                const float nx = (float)sx / npi, ny = (float)sy / npi, nz = (float)sz / npi;
                mu[j] = make_float4(nx, ny, nz, sqnorm3(nx, ny, nz));
                if (last) {
                    pi_out[(int64_t)c * J + j] = pj;
                    float* mo = mu_out + ((int64_t)c * J + j) * 3;
                    mo[0] = nx; mo[1] = ny; mo[2] = nz;
                }
            }
        }
        __syncthreads();
    }
    em_chip_poison(x, c, J, pi_out, mu_out);
}

// K15, cached variant (used when the N x J cost matrix fits in LDS, e.g. 64 KB at N=1024, J=16): the cost matrix of an
// outer iteration is computed once into LDS as [J][N] (both access patterns -- rows on threads, columns on waves -- then
// touch consecutive words), every Sinkhorn sweep is two passes over it (max, then sum of exp: one exp per entry, no
// branches), and the unnormalised gamma overwrites it for the M-step.  ~3x fewer instructions per sweep than recomputing
// sqrt/exp-merge chains; results agree with gmm_em_kernel to rounding.
// JT > 0: J == JT and N <= 1024 at compile time -- the Sinkhorn sweeps then keep a row's JT (a column's 16 per lane) exponents in
// registers between the max and the exp-sum pass instead of reading LDS and recomputing them (same operations in the same order: the
// results are bit-identical to the JT = 0 form).  FAST: the sweeps' exp(x - max) on v_exp_f32 (x - max <= 0, and only terms with
// x - max near 0 carry weight, where the 2^(x log2 e) form is as accurate as the range-reduced one).
template <int JT, bool FAST>
__global__ __launch_bounds__(EM_T) void gmm_em_cached_kernel(const float* __restrict__ xyz, const float* __restrict__ o,
                                                             const int32_t* __restrict__ ids0, int N, int J, int iters, int sk_iters,
                                                             float inv_eps, float eps, float inv_tau, float* __restrict__ gamma,
                                                             float* __restrict__ pi_out, float* __restrict__ mu_out, EmExit x) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* pts = reinterpret_cast<float4*>(lds);
    const int Npad = (N + 3) / 4 * 4;
    float* u = lds + 4 * (size_t)N;
    float* logp = u + Npad;
    float* rclip = logp + Npad;
    float4* mu = reinterpret_cast<float4*>(rclip + Npad);
    float* v = reinterpret_cast<float*>(mu + J);
    float* red = v + J;                                   // [48]: scratch, then per-wave residual shares of the even / odd sweeps
    float* uprev = red + 48;                              // [N]  u, v of the previous sweep (early exit: gmm_exit.h)
    float* vprev = uprev + Npad;                          // [J]
    float* Cs = vprev + (J + 3) / 4 * 4;                  // [J][N] cost, later unnormalised gamma
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = em_chip_cloud(x, red);
    if (c >= x.C) return;
    const bool track = x.on || x.resid != nullptr;
    constexpr int NW = EM_T / 64;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    const float* __restrict__ oc = o + (int64_t)c * N;

    float part = 0.0f;
    for (int n = tid; n < N; n += EM_T) {
        const float x = cloud[3 * n], y = cloud[3 * n + 1], z = cloud[3 * n + 2];
        pts[n] = make_float4(x, y, z, sqnorm3(x, y, z));
        part += oc[n];
    }
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float osum = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) osum += red[w];
    osum = fmaxf(osum, 1e-4f);
    for (int n = tid; n < N; n += EM_T) logp[n] = logf(oc[n] / osum + 1e-8f);
    for (int j = tid; j < J; j += EM_T) mu[j] = pts[ids0[(int64_t)c * J + j]];
    const float logq = logf((float)(1.0 / (double)J) + 1e-8f);
    __syncthreads();
    // track: sum |u - u0| + sum |v - v0| of every Sinkhorn sweep, the quantity whose batch mean the reference tests against `thresh` for its
    // early exit (lib/utils.py:99-102; em_chip_sweep_end).  Summed per wave into red[16 + 16 * (sk & 1) + wave], then in wave order.

    for (int it = 0; it < iters; ++it) {
        for (int n = tid; n < N; n += EM_T) {
            const float4 p = pts[n];
            u[n] = 0.0f;
            for (int j = 0; j < J; ++j) {
                const float4 m = mu[j];
                Cs[j * N + n] = cdist_mm(p.x, p.y, p.z, p.w, m.x, m.y, m.z, m.w) * inv_tau;
            }
        }
        for (int j = tid; j < J; j += EM_T) v[j] = 0.0f;
        __syncthreads();
        for (int sk = 0; sk < sk_iters; ++sk) {
            auto em_exp = [](float x) { return FAST ? __builtin_amdgcn_exp2f(x * 1.4426950408889634f) : expf(x); };
            float du = 0.0f, dv = 0.0f;          // this thread's share of the sweep's residual
            if constexpr (JT > 0) {
                if (tid < N) {                                            // u^{l+1}: one row per thread (N <= EM_T)
                    const int n = tid;
                    const float un = u[n];
                    float t[JT];
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int j = 0; j < JT; ++j) { t[j] = ((-Cs[j * N + n] + un) + v[j]) * inv_eps; mx = fmaxf(mx, t[j]); }
                    float se = 0.0f;
#pragma unroll
                    for (int j = 0; j < JT; ++j) se += em_exp(t[j] - mx);
                    const float un1 = eps * (logp[n] - (mx + logf(se))) + un;
                    u[n] = un1;
                    if (track) { uprev[n] = un; du = fabsf(un1 - un); }
                }
                if (track) { du = wave_sum(du); if (lane == 0) red[16 + 16 * (sk & 1) + wave] = du; }
                __syncthreads();
                for (int j = wave; j < JT; j += NW) {                     // v^{l+1}: columns on waves, 16 rows per lane
                    const float vj = v[j];
                    const float* __restrict__ Cj = Cs + j * N;
                    float t[16];
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int n = lane + 64 * i;
                        t[i] = n < N ? ((-Cj[n] + u[n]) + vj) * inv_eps : -__builtin_inff();
                        mx = fmaxf(mx, t[i]);
                    }
                    mx = wave_max(mx);
                    float se = 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (lane + 64 * i < N) se += em_exp(t[i] - mx);
                    se = wave_sum(se);
                    if (lane == 0) {
                        const float vj1 = eps * (logq - (mx + logf(se))) + vj;
                        v[j] = vj1;
                        if (track) { vprev[j] = vj; dv += fabsf(vj1 - vj); }
                    }
                }
                if (track && lane == 0) red[16 + 16 * (sk & 1) + wave] += dv;
                __syncthreads();
                if (track && em_chip_sweep_end(x, red, c, it, sk, sk_iters, u, uprev, v, vprev, N, J)) break;
                continue;
            }
            for (int n = tid; n < N; n += EM_T) {                     // u^{l+1}: rows on threads
                const float un = u[n];
                float mx = -__builtin_inff();
                for (int j = 0; j < J; ++j) mx = fmaxf(mx, ((-Cs[j * N + n] + un) + v[j]) * inv_eps);
                float se = 0.0f;
                for (int j = 0; j < J; ++j) se += expf(((-Cs[j * N + n] + un) + v[j]) * inv_eps - mx);
                const float un1 = eps * (logp[n] - (mx + logf(se))) + un;
                u[n] = un1;
                if (track) { uprev[n] = un; du += fabsf(un1 - un); }
            }
            if (track) { du = wave_sum(du); if (lane == 0) red[16 + 16 * (sk & 1) + wave] = du; }
            __syncthreads();
            for (int j = wave; j < J; j += NW) {                      // v^{l+1}: columns on waves
                const float vj = v[j];
                const float* __restrict__ Cj = Cs + j * N;
                float mx = -__builtin_inff();
                for (int n = lane; n < N; n += 64) mx = fmaxf(mx, ((-Cj[n] + u[n]) + vj) * inv_eps);
                mx = wave_max(mx);
                float se = 0.0f;
                for (int n = lane; n < N; n += 64) se += expf(((-Cj[n] + u[n]) + vj) * inv_eps - mx);
                se = wave_sum(se);
                if (lane == 0) {
                    const float vj1 = eps * (logq - (mx + logf(se))) + vj;
                    v[j] = vj1;
                    if (track) { vprev[j] = vj; dv += fabsf(vj1 - vj); }
                }
            }
            if (track && lane == 0) red[16 + 16 * (sk & 1) + wave] += dv;
            __syncthreads();
            if (track && em_chip_sweep_end(x, red, c, it, sk, sk_iters, u, uprev, v, vprev, N, J)) break;
        }
        const bool last = it + 1 == iters;
        for (int n = tid; n < N; n += EM_T) {                         // gamma = exp(K) (in place), row sums
            const float un = u[n];
            float rs = 0.0f;
            for (int j = 0; j < J; ++j) {
                float g = expf(((-Cs[j * N + n] + un) + v[j]) * inv_eps);
                g = (g != g) ? 0.0f : fminf(g, 3.4028234663852886e38f);
                Cs[j * N + n] = g;
                rs += g;
            }
            const float rc = fmaxf(rs, 1e-3f);
            rclip[n] = rc;
            if (last) {
                float* __restrict__ grow = gamma + ((int64_t)c * N + n) * J;
                for (int j = 0; j < J; ++j) grow[j] = Cs[j * N + n] / rc;
            }
        }
        __syncthreads();
        for (int j = wave; j < J; j += NW) {                          // M-step, fp64 column sums
            const float* __restrict__ Gj = Cs + j * N;
            double sg = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
            for (int n = lane; n < N; n += 64) {
                const float4 p = pts[n];
                const float g = Gj[n] / rclip[n];
                sg += g;
                sx += (double)g * p.x; sy += (double)g * p.y; sz += (double)g * p.z;
            }
            sg = wave_sum_d(sg); sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
            if (lane == 0) {
                const float pj = (float)sg / (float)N;
                const float npi = pj * (float)N + 1e-5f;
                const float nx = (float)sx / npi, ny = (float)sy / npi, nz = (float)sz / npi;
                mu[j] = make_float4(nx, ny, nz, sqnorm3(nx, ny, nz));
                if (last) {
                    pi_out[(int64_t)c * J + j] = pj;
                    float* mo = mu_out + ((int64_t)c * J + j) * 3;
                    mo[0] = nx; mo[1] = ny; mo[2] = nz;
                }
            }
        }
        __syncthreads();
    }
    em_chip_poison(x, c, J, pi_out, mu_out);
}

// ================================================================================================
// K16.  mu_feat[c][j][d] = sum_n gamma[c][n][j] * feats[c][n][d] / (N pi[c][j] + 1e-5)
// block = (cloud, 64-channel slab, 16-cluster slab): 64 channels x 4 row lanes, 16 accumulators each.
// ================================================================================================
template <int JS>      // clusters per block: 16, or 64 (feats are then read once instead of J/16 times: 0.87 -> 0.3 ms at J = 64)
__global__ __launch_bounds__(256) void gmm_feat_mean_kernel(const float* __restrict__ gamma, const float* __restrict__ pi,
                                                            const float* __restrict__ feats, int64_t ld, int N, int J, int D,
                                                            float* __restrict__ mu_feat) {
    __shared__ float gs[64][JS + 1];
    __shared__ float red[4][16][64];
    const int c = blockIdx.z, ch = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + ch, j0 = blockIdx.y * JS;
    const float* __restrict__ F = feats + (int64_t)c * N * ld;
    const float* __restrict__ G = gamma + (int64_t)c * N * J;
    float acc[JS];
#pragma unroll
    for (int j = 0; j < JS; ++j) acc[j] = 0.0f;
    for (int n0 = 0; n0 < N; n0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * JS; i += 256) {
            const int r = i / JS, j = i % JS;
            gs[r][j] = (n0 + r < N && j0 + j < J) ? G[(int64_t)(n0 + r) * J + j0 + j] : 0.0f;
        }
        __syncthreads();
        for (int r = rl; r < 64; r += 4) {
            if (n0 + r >= N) break;
            const float f = d < D ? F[(int64_t)(n0 + r) * ld + d] : 0.0f;
#pragma unroll
            for (int j = 0; j < JS; ++j) acc[j] = fmaf(gs[r][j], f, acc[j]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < JS; jb += 16) {          // cross-row-lane reduction, 16 clusters at a time
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) red[rl][j][ch] = acc[jb + j];
        __syncthreads();
        if (rl == 0 && d < D) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j0 + jb + j >= J) break;
                const float sum = (red[0][j][ch] + red[1][j][ch]) + (red[2][j][ch] + red[3][j][ch]);
                const float npi = pi[(int64_t)c * J + j0 + jb + j] * (float)N + 1e-5f;
                mu_feat[((int64_t)c * J + j0 + jb + j) * D + d] = sum / npi;
            }
        }
    }
}

// K16 for 16 < J <= 64 (BASELINE configs[2], [3]: J = 64, N = 2048): the same sums on v_mfma_f32_32x32x2_f32 -- exact fp32 products, and the
// operand layout of that instruction IS a contraction over rows: lane l supplies A[j = l % 32][n = l / 32] = gamma[n][j] and
// B[n = l / 32][d = l % 32] = feats[n][d], both read straight from their row-major maps with the 32 lanes of a half wave on consecutive
// addresses.  Workgroup = (cloud, half of the channels): 4 waves x (2 cluster blocks x 2 channel blocks) accumulators, two rows per MFMA
// step, eight steps' loads in flight.  The VALU form above spends 64 fma per loaded feature: 0.70 ms at 128 clouds of 2048 x 512 (see DESIGN).
using f32x16m = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void gmm_feat_mean_mfma_kernel(const float* __restrict__ gamma, const float* __restrict__ pi,
                                                                 const float* __restrict__ feats, int64_t ld, int N, int J, int D,
                                                                 float* __restrict__ mu_feat) {
    const int c = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int d0 = (blockIdx.x * 4 + wave) * 64;                    // this wave's 64 channels
    const float* __restrict__ F = feats + (int64_t)c * N * ld;
    const float* __restrict__ G = gamma + (int64_t)c * N * J;
    const bool j1 = lr + 32 < J, j0v = lr < J;                      // (J > 16: the first cluster block may still be partial for J < 32)
    const bool da = d0 + lr < D, db = d0 + 32 + lr < D;
    f32x16m acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    constexpr int U = 8;                                             // MFMA steps (2 rows each) per batch of loads
    for (int n0 = 0; n0 < N; n0 += 2 * U) {
        float ga[U][2], fb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + 2 * u + lh;
            const bool ok = n < N;
            ga[u][0] = (ok && j0v) ? G[(int64_t)n * J + lr] : 0.0f;
            ga[u][1] = (ok && j1) ? G[(int64_t)n * J + 32 + lr] : 0.0f;
            fb[u][0] = (ok && da) ? F[(int64_t)n * ld + d0 + lr] : 0.0f;
            fb[u][1] = (ok && db) ? F[(int64_t)n * ld + d0 + 32 + lr] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[u][a], fb[u][b], acc[a][b], 0, 0, 0);
    }
    // accumulator (a, b): lane = channel d0 + 32 b + lr, register v = cluster 32 a + 8 (v / 4) + 4 lh + v % 4
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int d = d0 + 32 * b + lr;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int j = 32 * a + 8 * (v >> 2) + 4 * lh + (v & 3);
                if (j < J && d < D) {
                    const float npi = pi[(int64_t)c * J + j] * (float)N + 1e-5f;
                    mu_feat[((int64_t)c * J + j) * D + d] = acc[a][b][v] / npi;
                }
            }
        }
}

// ================================================================================================
// K18.  Weighted Kabsch (lib/se3.py:256-289) for one pair, executed by one lane in fp64.
//   cov = sum_n w_n (s_n - cs)(c_n - cc)^T (+1e-5 I);  cov = U S V^T;  R = V diag(1,1,det) U^T
// With (v1,v2) the two leading right singular vectors and u_i = cov v_i / sigma_i, the proper rotation is
//   R = v1 u1^T + v2 u2^T + (v1 x v2)(u1 x u2)^T
// which equals the reference's V U^T with V[:,2] negated when det(V U^T) <= 0, independent of the sign
// ambiguities of the SVD.  (v1,v2) come from a cyclic Jacobi eigen-decomposition of cov^T cov.
// ================================================================================================
__device__ void jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (fabs(A[p][q]) < 1e-300) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
            for (int k = 0; k < 3; ++k) {   // A <- A J
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = cs * akp - sn * akq;
                A[k][q] = sn * akp + cs * akq;
            }
            for (int k = 0; k < 3; ++k) {   // A <- J^T A
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = cs * apk - sn * aqk;
                A[q][k] = sn * apk + cs * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = cs * vkp - sn * vkq;
                V[k][q] = sn * vkp + cs * vkq;
            }
        }
    }
}

__device__ void rotation_from_cov(const double cov[3][3], double R[3][3]) {
    double AtA[3][3], V[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += cov[k][i] * cov[k][j];
        AtA[i][j] = s;
    }
    jacobi_eig3(AtA, V);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b)
        if (AtA[ord[b]][ord[b]] > AtA[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double v[2][3], uu[2][3];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 3; ++k) v[i][k] = V[k][ord[i]];
    for (int i = 0; i < 2; ++i) {
        for (int k = 0; k < 3; ++k) uu[i][k] = cov[k][0] * v[i][0] + cov[k][1] * v[i][1] + cov[k][2] * v[i][2];
        if (i == 1) {   // Gram-Schmidt against u1
            const double d = uu[1][0] * uu[0][0] + uu[1][1] * uu[0][1] + uu[1][2] * uu[0][2];
            for (int k = 0; k < 3; ++k) uu[1][k] -= d * uu[0][k];
        }
        const double nrm = sqrt(uu[i][0] * uu[i][0] + uu[i][1] * uu[i][1] + uu[i][2] * uu[i][2]);
        const double inv = nrm > 0 ? 1.0 / nrm : 0.0;
        for (int k = 0; k < 3; ++k) uu[i][k] *= inv;
    }
    const double v3[3] = {v[0][1] * v[1][2] - v[0][2] * v[1][1], v[0][2] * v[1][0] - v[0][0] * v[1][2], v[0][0] * v[1][1] - v[0][1] * v[1][0]};
    const double u3[3] = {uu[0][1] * uu[1][2] - uu[0][2] * uu[1][1], uu[0][2] * uu[1][0] - uu[0][0] * uu[1][2],
                          uu[0][0] * uu[1][1] - uu[0][1] * uu[1][0]};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R[a][b] = v[0][a] * uu[0][b] + v[1][a] * uu[1][b] + v3[a] * u3[b];
}

// src/corr given as callable accessors (a = axis, n = cluster)
template <class FS, class FC, class FW>
__device__ void kabsch_solve(int J, FS src, FC corr, FW wgt, float* __restrict__ R_out, float* __restrict__ t_out) {
    double ws = 0.0, cs[3] = {0, 0, 0}, cc[3] = {0, 0, 0};
    for (int n = 0; n < J; ++n) {
        const double w = wgt(n);
        ws += w;
        for (int a = 0; a < 3; ++a) { cs[a] += w * src(a, n); cc[a] += w * corr(a, n); }
    }
    for (int a = 0; a < 3; ++a) { cs[a] /= ws; cc[a] /= ws; }
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int n = 0; n < J; ++n) {
        const double w = wgt(n);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a][b] += (src(a, n) - cs[a]) * w * (corr(b, n) - cc[b]);
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        if (cov[a][b] != cov[a][b]) cov[a][b] = 0.0;      // nan_to_num(nan=0)
        if (a == b) cov[a][b] += 1e-5;
    }
    double R[3][3];
    rotation_from_cov(cov, R);
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) R_out[a * 3 + b] = (float)R[a][b];
        t_out[a] = (float)(-(R[a][0] * cs[0] + R[a][1] * cs[1] + R[a][2] * cs[2]) + cc[a]);
    }
}

// The same solve by one whole wave: lane n takes clusters n, n + 64, ...; the 7 + 9 sums are wave reductions in fp64.  On one lane the two loops
// over J are chains of dependent global-memory round trips (one wave per SIMD: nothing hides them): ~90 us at J = 64, the longest part of
// match_kabsch_kernel.  (Summation order differs from the sequential loops at the 1e-16 level.)
template <class FS, class FC, class FW>
__device__ void kabsch_solve_wave(int J, FS src, FC corr, FW wgt, float* __restrict__ R_out, float* __restrict__ t_out) {
    const int lane = threadIdx.x & 63;
    double ws = 0.0, cs[3] = {0, 0, 0}, cc[3] = {0, 0, 0};
    for (int n = lane; n < J; n += 64) {
        const double w = wgt(n);
        ws += w;
        for (int a = 0; a < 3; ++a) { cs[a] += w * src(a, n); cc[a] += w * corr(a, n); }
    }
    ws = ogmm::wave_sum_d(ws);
    for (int a = 0; a < 3; ++a) { cs[a] = ogmm::wave_sum_d(cs[a]) / ws; cc[a] = ogmm::wave_sum_d(cc[a]) / ws; }
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int n = lane; n < J; n += 64) {
        const double w = wgt(n);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a][b] += (src(a, n) - cs[a]) * w * (corr(b, n) - cc[b]);
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a][b] = ogmm::wave_sum_d(cov[a][b]);
    if (lane != 0) return;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        if (cov[a][b] != cov[a][b]) cov[a][b] = 0.0;      // nan_to_num(nan=0)
        if (a == b) cov[a][b] += 1e-5;
    }
    double R[3][3];
    rotation_from_cov(cov, R);
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) R_out[a * 3 + b] = (float)R[a][b];
        t_out[a] = (float)(-(R[a][0] * cs[0] + R[a][1] * cs[1] + R[a][2] * cs[2]) + cc[a]);
    }
}

__global__ void kabsch_kernel(const float* __restrict__ src, const float* __restrict__ corr, const float* __restrict__ w, int B, int J,
                              float* __restrict__ R, float* __restrict__ t) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* s = src + (int64_t)b * 3 * J;
    const float* cr = corr + (int64_t)b * 3 * J;
    const float* ww = w + (int64_t)b * J;
    kabsch_solve(J, [&](int a, int n) { return (double)s[a * J + n]; }, [&](int a, int n) { return (double)cr[a * J + n]; },
                 [&](int n) { return (double)ww[n]; }, R + (int64_t)b * 9, t + (int64_t)b * 3);
}

// proper rotation of a given 3x3 cross-covariance (DeepGMR's gmm_register, baseline/deepgmr.py:28-34): R = V diag(1,1,det(V U^T)) U^T
__global__ void rotation_from_cov_kernel(const float* __restrict__ M, int B, float* __restrict__ R) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double cov[3][3], Rd[3][3];
    for (int i = 0; i < 9; ++i) { const float v = M[(int64_t)b * 9 + i]; cov[i / 3][i % 3] = (v != v) ? 0.0 : (double)v; }
    rotation_from_cov(cov, Rd);
    for (int i = 0; i < 9; ++i) R[(int64_t)b * 9 + i] = (float)Rd[i / 3][i % 3];
}

// ================================================================================================
// Backward of K18 (training): gradients of a loss w.r.t. src, corr, w given dL/dR, dL/dt.  One lane per pair, fp64.
//   H = cov = U S V^T,  R = V D U^T,  D = diag(1,1,d),  t = cc - R cs.
//   With A = U^T dH V:  U^T dU = wU, V^T dV = wV skew,  A_ij = wU_ij s_j - s_i wV_ij  =>  for i != j
//     wU_ij = (s_j A_ij + s_i A_ji) / (s_j^2 - s_i^2),   wV_ij = (s_i A_ij + s_j A_ji) / (s_j^2 - s_i^2),
//   and V^T dR U = wV D - D wU.  Contracting with G = V^T (dL/dR) U gives dL/dA; where d_i = d_j the singular
//   denominators cancel:  dL/dA_ij = (G_ji - G_ij) / (s_i + s_j)  (no blow-up for equal singular values);
//   for the reflected pair (d_i = 1, d_j = -1):  dL/dA_ij = dL/dA_ji = -(G_ij + G_ji) / (s_j - s_i).
//   dL/dH = U (dL/dA) V^T.  This is torch's svd_backward specialised to the product V D U^T (lib/se3.py:276-289).
// ================================================================================================
__global__ void kabsch_bwd_kernel(const float* __restrict__ src, const float* __restrict__ corr, const float* __restrict__ w, int B, int J,
                                  const float* __restrict__ gR_in, const float* __restrict__ gt_in,
                                  float* __restrict__ g_src, float* __restrict__ g_corr, float* __restrict__ g_w) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* s = src + (int64_t)b * 3 * J;
    const float* q = corr + (int64_t)b * 3 * J;
    const float* ww = w + (int64_t)b * J;
    double ws = 0.0, cs[3] = {0, 0, 0}, cc[3] = {0, 0, 0};
    for (int n = 0; n < J; ++n) {
        const double wn = ww[n];
        ws += wn;
        for (int a = 0; a < 3; ++a) { cs[a] += wn * s[a * J + n]; cc[a] += wn * q[a * J + n]; }
    }
    for (int a = 0; a < 3; ++a) { cs[a] /= ws; cc[a] /= ws; }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int n = 0; n < J; ++n) {
        const double wn = ww[n];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) H[a][c] += (s[a * J + n] - cs[a]) * wn * (q[c * J + n] - cc[c]);
    }
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) {
        if (H[a][c] != H[a][c]) H[a][c] = 0.0;
        if (a == c) H[a][c] += 1e-5;
    }
    // SVD pieces exactly as the forward builds them
    double AtA[3][3], Vj[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double t_ = 0.0;
        for (int k = 0; k < 3; ++k) t_ += H[k][i] * H[k][j];
        AtA[i][j] = t_;
    }
    jacobi_eig3(AtA, Vj);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int c = a + 1; c < 3; ++c)
        if (AtA[ord[c]][ord[c]] > AtA[ord[a]][ord[a]]) { const int t_ = ord[a]; ord[a] = ord[c]; ord[c] = t_; }
    double V[3][3], U[3][3], sv[3], D[3] = {1.0, 1.0, 1.0};      // columns: V[k][i] = v_i[k]
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 3; ++k) V[k][i] = Vj[k][ord[i]];
    V[0][2] = V[1][0] * V[2][1] - V[2][0] * V[1][1];
    V[1][2] = V[2][0] * V[0][1] - V[0][0] * V[2][1];
    V[2][2] = V[0][0] * V[1][1] - V[1][0] * V[0][1];
    double hv[3][3];                                              // hv[i] = H v_i
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) hv[i][k] = H[k][0] * V[0][i] + H[k][1] * V[1][i] + H[k][2] * V[2][i];
    for (int i = 0; i < 2; ++i) {
        if (i == 1) {
            const double dd = hv[1][0] * U[0][0] + hv[1][1] * U[1][0] + hv[1][2] * U[2][0];
            for (int k = 0; k < 3; ++k) hv[1][k] -= dd * U[k][0];
        }
        const double nrm = sqrt(hv[i][0] * hv[i][0] + hv[i][1] * hv[i][1] + hv[i][2] * hv[i][2]);
        sv[i] = nrm;
        const double inv = nrm > 0 ? 1.0 / nrm : 0.0;
        for (int k = 0; k < 3; ++k) U[k][i] = hv[i][k] * inv;
    }
    const double u3p[3] = {U[1][0] * U[2][1] - U[2][0] * U[1][1], U[2][0] * U[0][1] - U[0][0] * U[2][1], U[0][0] * U[1][1] - U[1][0] * U[0][1]};
    const double proj = u3p[0] * hv[2][0] + u3p[1] * hv[2][1] + u3p[2] * hv[2][2];        // = d * s_3
    D[2] = proj >= 0 ? 1.0 : -1.0;
    sv[2] = fabs(proj);
    for (int k = 0; k < 3; ++k) U[k][2] = D[2] * u3p[k];
    double R[3][3];
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) R[a][c] = V[a][0] * U[c][0] + V[a][1] * U[c][1] + D[2] * V[a][2] * U[c][2];
    // upstream gradients; t = cc - R cs
    double gR[3][3], gt[3], gcs[3], G[3][3], gA[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gH[3][3];
    for (int a = 0; a < 3; ++a) gt[a] = gt_in ? (double)gt_in[b * 3 + a] : 0.0;
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) gR[a][c] = (gR_in ? (double)gR_in[b * 9 + a * 3 + c] : 0.0) - gt[a] * cs[c];
    for (int c = 0; c < 3; ++c) gcs[c] = -(R[0][c] * gt[0] + R[1][c] * gt[1] + R[2][c] * gt[2]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {      // G = V^T gR U
        double t_ = 0.0;
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) t_ += V[a][i] * gR[a][c] * U[c][j];
        G[i][j] = t_;
    }
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) {
        if (D[i] == D[j]) {
            const double den = sv[i] + sv[j];
            const double inv = den > 1e-300 ? 1.0 / den : 0.0;
            gA[i][j] = (G[j][i] - G[i][j]) * inv;
            gA[j][i] = -gA[i][j];
        } else {
            const double den = sv[j] - sv[i];
            const double inv = fabs(den) > 1e-300 ? 1.0 / den : 0.0;
            gA[i][j] = gA[j][i] = -(G[i][j] + G[j][i]) * inv;
        }
    }
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) {      // gH = U gA V^T
        double t_ = 0.0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t_ += U[a][i] * gA[i][j] * V[c][j];
        gH[a][c] = t_;
    }
    for (int n = 0; n < J; ++n) {
        const double wn = ww[n];
        double sc[3], qc[3], hq[3], hs[3];
        for (int a = 0; a < 3; ++a) { sc[a] = s[a * J + n] - cs[a]; qc[a] = q[a * J + n] - cc[a]; }
        for (int a = 0; a < 3; ++a) {
            hq[a] = gH[a][0] * qc[0] + gH[a][1] * qc[1] + gH[a][2] * qc[2];      // gH qc
            hs[a] = gH[0][a] * sc[0] + gH[1][a] * sc[1] + gH[2][a] * sc[2];      // gH^T sc
        }
        double gw = 0.0;
        for (int a = 0; a < 3; ++a) {
            if (g_src) g_src[(int64_t)b * 3 * J + a * J + n] = (float)(wn * hq[a] + wn / ws * gcs[a]);
            if (g_corr) g_corr[(int64_t)b * 3 * J + a * J + n] = (float)(wn * hs[a] + wn / ws * gt[a]);
            gw += sc[a] * hq[a] + (gcs[a] * sc[a] + gt[a] * qc[a]) / ws;
        }
        if (g_w) g_w[(int64_t)b * J + n] = (float)gw;
    }
}

// Cosine similarities of match_kabsch: the normalised rows go through LDS in chunks of MK_DC channels, channel-major ([d][cluster],
// pitch J+4), and every thread accumulates a BS x BS block of the J x J matrix (BS >= ceil(J/16)): J*J*D/256 FMAs per thread
// instead of one 2 KB row pair fetched from L2 per entry and wave (1.8 ms -> 0.35 ms for the whole kernel at J = 64).
template <int BS>
__device__ __forceinline__ void cosine_block(const float* __restrict__ Fs, const float* __restrict__ Ft, const float* __restrict__ ns,
                                             const float* __restrict__ nt, int J, int D, int tid, float* __restrict__ sim, float* __restrict__ xs) {
    constexpr int MK_DC = 64;
    const int pitch = J + 4;
    float* xt = xs + MK_DC * pitch;
    const int n0 = (tid >> 4) * BS, m0 = (tid & 15) * BS;
    double acc[BS][BS];      // fp64 accumulation: the softmax behind it has temperature 0.05 and the solve is ill-conditioned for
                             // near-degenerate cluster layouts (the N=717 / J=128 fixture): keep this stage at the fp32 rounding of its result
#pragma unroll
    for (int a = 0; a < BS; ++a)
#pragma unroll
        for (int c = 0; c < BS; ++c) acc[a][c] = 0.0;
    for (int d0 = 0; d0 < D; d0 += MK_DC) {
        __syncthreads();
        for (int i = tid; i < 2 * J * MK_DC; i += 256) {
            const int d = i % MK_DC, r = i / MK_DC;              // consecutive threads read consecutive channels of one row
            const bool src_row = r < J;
            const int row = src_row ? r : r - J;
            const float v = d0 + d < D ? (src_row ? Fs : Ft)[(int64_t)row * D + d0 + d] / (src_row ? ns[row] : nt[row]) : 0.0f;
            (src_row ? xs : xt)[d * pitch + row] = v;
        }
        __syncthreads();
        if (n0 < J && m0 < J) {
            for (int d = 0; d < MK_DC; ++d) {
                float a_[BS], b_[BS];
#pragma unroll
                for (int a = 0; a < BS; ++a) { a_[a] = n0 + a < J ? xs[d * pitch + n0 + a] : 0.0f; b_[a] = m0 + a < J ? xt[d * pitch + m0 + a] : 0.0f; }
#pragma unroll
                for (int a = 0; a < BS; ++a)
#pragma unroll
                    for (int c = 0; c < BS; ++c) acc[a][c] = fma((double)a_[a], (double)b_[c], acc[a][c]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < BS; ++a)
#pragma unroll
        for (int c = 0; c < BS; ++c)
            if (n0 + a < J && m0 + c < J) sim[(n0 + a) * J + m0 + c] = (float)acc[a][c];
}

// ================================================================================================
// K17+K18.  One workgroup of 256 threads per pair: cosine similarity J x J (each wave reduces one entry
// at a time over D with coalesced 16-byte lane loads), softmax(sim / T) over target clusters, soft
// correspondences, then the rigid solve on lane 0.   models/dgcnn.py:96-115.
// ================================================================================================
__global__ __launch_bounds__(256) void match_kabsch_kernel(const float* __restrict__ mu_s, const float* __restrict__ mu_t,
                                                           const float* __restrict__ f_s, const float* __restrict__ f_t, int J, int D,
                                                           float inv_temp, float* __restrict__ R, float* __restrict__ t,
                                                           float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sim = lds;                 // [J][J]
    float* ns = sim + J * J;          // [J] norms of f_s rows
    float* nt = ns + J;               // [J]
    float* corr = nt + J;             // [3][J]
    float* wsum = corr + 3 * J;       // [J]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ Fs = f_s + (int64_t)b * J * D;
    const float* __restrict__ Ft = f_t + (int64_t)b * J * D;
    for (int r0 = wave; r0 < 2 * J; r0 += 16) {             // four rows per pass of a wave: their loads travel together
        double ss[4] = {0.0, 0.0, 0.0, 0.0};
        const float* __restrict__ p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = min(r0 + 4 * q, 2 * J - 1);
            p[q] = r < J ? Fs + (int64_t)r * D : Ft + (int64_t)(r - J) * D;
        }
        for (int d = lane; d < D; d += 64) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = p[q][d];
#pragma unroll
            for (int q = 0; q < 4; ++q) ss[q] = fma((double)v[q], (double)v[q], ss[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 4 * q;
            const double tot = wave_sum_d(ss[q]);
            if (lane == 0 && r < 2 * J) (r < J ? ns[r] : nt[r - J]) = fmaxf((float)sqrt(tot), 1e-12f);
        }
    }
    __syncthreads();
    {
        const int BSr = (J + 15) / 16;
        float* xs_ = wsum + J;
        if (BSr <= 1) cosine_block<1>(Fs, Ft, ns, nt, J, D, tid, sim, xs_);
        else if (BSr <= 2) cosine_block<2>(Fs, Ft, ns, nt, J, D, tid, sim, xs_);
        else if (BSr <= 4) cosine_block<4>(Fs, Ft, ns, nt, J, D, tid, sim, xs_);
        else cosine_block<8>(Fs, Ft, ns, nt, J, D, tid, sim, xs_);
    }
    __syncthreads();
    if (J <= 64) {
        // four neighbouring lanes per source cluster, each a quarter of the target clusters: every row at once, and the row reductions are two
        // lane exchanges inside a quad instead of six across the wave (one wave per row, rows in turn, took 2 us per row)
        const int n = tid >> 2, part = tid & 3;
        const int m_lo = part * ((J + 3) / 4), m_hi = min(J, m_lo + (J + 3) / 4);
        float mx = -__builtin_inff();
        if (n < J) for (int m = m_lo; m < m_hi; ++m) mx = fmaxf(mx, sim[n * J + m] * inv_temp);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float se = 0.0f;
        if (n < J) for (int m = m_lo; m < m_hi; ++m) se += expf(sim[n * J + m] * inv_temp - mx);
        se += __shfl_xor(se, 1, 64); se += __shfl_xor(se, 2, 64);
        float cx = 0.0f, cy = 0.0f, cz = 0.0f, ws = 0.0f;
        if (n < J) for (int m = m_lo; m < m_hi; ++m) {
            const float sc = expf(sim[n * J + m] * inv_temp - mx) / se;
            if (scores) scores[((int64_t)b * J + n) * J + m] = sc;
            const float* __restrict__ mt = mu_t + ((int64_t)b * J + m) * 3;
            cx = fmaf(mt[0], sc, cx); cy = fmaf(mt[1], sc, cy); cz = fmaf(mt[2], sc, cz);
            ws += sc;
        }
        cx += __shfl_xor(cx, 1, 64); cx += __shfl_xor(cx, 2, 64);
        cy += __shfl_xor(cy, 1, 64); cy += __shfl_xor(cy, 2, 64);
        cz += __shfl_xor(cz, 1, 64); cz += __shfl_xor(cz, 2, 64);
        ws += __shfl_xor(ws, 1, 64); ws += __shfl_xor(ws, 2, 64);
        if (part == 0 && n < J) { corr[n] = cx; corr[J + n] = cy; corr[2 * J + n] = cz; wsum[n] = ws; }
    } else
    for (int n = wave; n < J; n += 4) {     // softmax over m, one wave per source cluster
        float mx = -__builtin_inff();
        for (int m = lane; m < J; m += 64) mx = fmaxf(mx, sim[n * J + m] * inv_temp);
        mx = wave_max(mx);
        float se = 0.0f;
        for (int m = lane; m < J; m += 64) se += expf(sim[n * J + m] * inv_temp - mx);
        se = wave_sum(se);
        float cx = 0.0f, cy = 0.0f, cz = 0.0f, ws = 0.0f;
        for (int m = lane; m < J; m += 64) {
            const float sc = expf(sim[n * J + m] * inv_temp - mx) / se;
            if (scores) scores[((int64_t)b * J + n) * J + m] = sc;
            const float* __restrict__ mt = mu_t + ((int64_t)b * J + m) * 3;
            cx = fmaf(mt[0], sc, cx); cy = fmaf(mt[1], sc, cy); cz = fmaf(mt[2], sc, cz);
            ws += sc;
        }
        cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz); ws = wave_sum(ws);
        if (lane == 0) { corr[n] = cx; corr[J + n] = cy; corr[2 * J + n] = cz; wsum[n] = ws; }
    }
    __syncthreads();
    if (wave == 0) {
        const float* __restrict__ ms = mu_s + (int64_t)b * J * 3;
        kabsch_solve_wave(J, [&](int a, int n) { return (double)ms[n * 3 + a]; }, [&](int a, int n) { return (double)corr[a * J + n]; },
                          [&](int n) { return (double)wsum[n]; }, R + (int64_t)b * 9, t + (int64_t)b * 3);
    }
}

// ================================================================================================
// K19.  (a) nearest point to each cluster centre (lib/utils.py:244-254, cdist in matmul form + top-1);
//       (b) InfoNCE rows (lib/loss.py:22-57): block = (cluster m, cloud); wave w computes one family of
//           J cosine scores  w=0: x_m.x_n  1: x_m.y_n  2: y_m.x_n  3: y_m.y_n   (x = anchors, y = positives).
// ================================================================================================
__global__ __launch_bounds__(256) void nearest_point_kernel(const float* __restrict__ xyz, const float* __restrict__ mu, int N, int J,
                                                            int32_t* __restrict__ near) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int j = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ cloud = xyz + (int64_t)c * N * 3;
    const float* __restrict__ m = mu + ((int64_t)c * J + j) * 3;
    const float mx = m[0], my = m[1], mz = m[2], mn = sqnorm3(mx, my, mz);
    float best = __builtin_inff();
    int bidx = 0x7fffffff;
    for (int n = tid; n < N; n += 256) {
        const float x = cloud[3 * n], y = cloud[3 * n + 1], z = cloud[3 * n + 2];
        // cdist(mu, xyz): the roles of x1/x2 are swapped relative to cdist_mm: chain([-2mu,|mu|^2,1].[x,1,|x|^2])
        float acc = mul_rn(-2.0f * mx, x);
        acc = __fmaf_rn(-2.0f * my, y, acc);
        acc = __fmaf_rn(-2.0f * mz, z, acc);
        acc = add_rn(acc, mn);
        acc = add_rn(acc, sqnorm3(x, y, z));
        const float d = sqrtf(fmaxf(acc, 0.0f));
        if (d < best) { best = d; bidx = n; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (ov < best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] < best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
        near[(int64_t)c * J + j] = bidx;
    }
}

__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ feats, int64_t ld, const float* __restrict__ mu_feat,
                                                           const int32_t* __restrict__ near, int N, int J, int D, float inv_tau,
                                                           float* __restrict__ row_loss) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [4][J]
    const int m = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ F = feats + (int64_t)c * N * ld;
    const float* __restrict__ Y = mu_feat + (int64_t)c * J * D;
    const int32_t* __restrict__ nr = near + (int64_t)c * J;
    const bool a_is_x = wave < 2, b_is_x = (wave & 1) == 0;
    const float* __restrict__ pa = a_is_x ? F + (int64_t)nr[m] * ld : Y + (int64_t)m * D;
    // the wave's own row, normalised, stays in registers (D <= 64 * INF_R); every other row is then read ONCE: its norm and
    // its dot product with the own row come out of the same pass (one division per score instead of two per element)
    constexpr int INF_R = 16;
    float an[INF_R];
    float na = 0.0f;
#pragma unroll
    for (int i = 0; i < INF_R; ++i) { const int d = lane + 64 * i; an[i] = d < D ? pa[d] : 0.0f; na = fmaf(an[i], an[i], na); }
    na = fmaxf(sqrtf(wave_sum(na)), 1e-12f);
#pragma unroll
    for (int i = 0; i < INF_R; ++i) an[i] = an[i] / na;
    for (int n = 0; n < J; ++n) {
        const float* __restrict__ pb = b_is_x ? F + (int64_t)nr[n] * ld : Y + (int64_t)n * D;
        float nb = 0.0f, dot = 0.0f;
#pragma unroll
        for (int i = 0; i < INF_R; ++i) {
            const int d = lane + 64 * i;
            const float v = d < D ? pb[d] : 0.0f;
            nb = fmaf(v, v, nb);
            dot = fmaf(an[i], v, dot);
        }
        nb = fmaxf(sqrtf(wave_sum(nb)), 1e-12f);
        dot = wave_sum(dot) / nb;
        if (lane == 0) sc[wave * J + n] = dot * inv_tau;
    }
    __syncthreads();
    // x-row m: logits [xy[m][m] | xx[m][n!=m] | xy[m][n!=m]];  y-row m: [yx[m][m] | yx[m][n!=m] | yy[m][n!=m]]
    if (wave < 2) {
        const float* s_own = wave == 0 ? sc + 0 * J : sc + 2 * J;     // negatives, first family (xx | yx)
        const float* s_oth = wave == 0 ? sc + 1 * J : sc + 3 * J;     // second family (xy | yy)
        const float pos = wave == 0 ? sc[1 * J + m] : sc[2 * J + m];
        float mx = pos;
        for (int n = lane; n < J; n += 64)
            if (n != m) mx = fmaxf(mx, fmaxf(s_own[n], s_oth[n]));
        mx = wave_max(mx);
        float se = 0.0f;
        for (int n = lane; n < J; n += 64)
            if (n != m) se += expf(s_own[n] - mx) + expf(s_oth[n] - mx);
        se = wave_sum(se) + expf(pos - mx);
        if (lane == 0) row_loss[((int64_t)c * 2 + wave) * J + m] = (mx + logf(se)) - pos;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------- entry points
namespace {
size_t em_chip_lds(int N, int J, bool cached) {
    const size_t Npad = (size_t)(N + 3) / 4 * 4, Jp = (size_t)(J + 3) / 4 * 4;   // Npad keeps the float4 mu array 16-byte aligned behind the per-point floats
    return ((size_t)4 * N + 4 * Npad + 4 * (size_t)J + J + 48 + Jp + (cached ? (size_t)N * J : 0)) * sizeof(float);
}
}  // namespace

extern "C" int64_t ogmm_gmm_em_exit_workspace_bytes(int C, int N, int iters, int sk_iters, int group_size) {
    return (int64_t)ogmm::em_exit_bytes(C, N, iters, sk_iters, group_size);
}

// Largest call group the on-chip kernels take with the early exit on: the clouds of a group wait for each other, so one resident round of
// workgroups must hold a whole group (one 1024-thread workgroup per CU at these LDS sizes).  0: the shape does not run on chip.
extern "C" int ogmm_gmm_em_chip_cached(int N, int J) { return N > 0 && J > 0 && em_chip_lds(N, J, true) <= 128 * 1024 ? 1 : 0; }

extern "C" int ogmm_gmm_em_chip_max_group(int N, int J) {
    if (N <= 0 || J <= 0 || em_chip_lds(N, J, false) > 160 * 1024) return 0;
    int cus = 256, dev_id = 0;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    // Half of the CUs, not all of them: the clouds of a group wait for each other, so a group must be able to become completely resident.  With the cap at
    // the full CU count, two exit-on launches running side by side (two models / streams / processes) could each hold a partial group on the chip and
    // stall each other until the poll limit; with half, two such launches fit, and larger groups take the launch sequence
    // (ogmm_gmm_em_multi), which has no cross-workgroup wait at all.
    return cus / 2;
}

// resid [C][iters][sk_iters] (may be NULL) receives every Sinkhorn sweep's sum |u - u0| + sum |v - v0| per cloud (lib/utils.py:99-101), NaN for
// sweeps that did not run; sweeps [C / group_size][iters] (may be NULL) the number of sweeps each E-step ran for each call group.
extern "C" int ogmm_gmm_em(const float* xyz, const float* o, const int32_t* ids0, int C, int N, int J, int iters, int sk_iters,
                           float epsilon, float tau, double thresh, int group_size, float* gamma, float* pi, float* mu, float* resid,
                           int32_t* sweeps, void* exit_ws, void* stream) {
    OGMM_REQUIRE(xyz && o && ids0 && gamma && pi && mu, "ogmm_gmm_em: null pointer");
    OGMM_REQUIRE(C > 0 && N > 0 && J > 0 && J <= N && iters > 0 && sk_iters >= 0 && epsilon > 0 && tau > 0, "ogmm_gmm_em: bad sizes C=%d N=%d J=%d", C, N, J);
    const size_t lds = em_chip_lds(N, J, false);
    OGMM_REQUIRE(lds <= 160 * 1024, "ogmm_gmm_em: N=%d, J=%d needs %zu B of LDS (> 160 KiB)", N, J, lds);
    ogmm::EmExit x;
    if (int rc = ogmm::em_exit_setup(x, thresh, group_size, C, N, iters, sk_iters, resid, sweeps, exit_ws, ogmm::as_stream(stream))) return rc;
    if (x.on) {
        const int cap = ogmm_gmm_em_chip_max_group(N, J);
        OGMM_REQUIRE(x.G <= cap, "ogmm_gmm_em: a call group of %d clouds does not fit one resident round (%d) with the early exit on: use ogmm_gmm_em_multi", x.G, cap);
    }
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_em_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_em_cached_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_em_cached_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_em_cached_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const float inv_eps = (float)(1.0 / (double)epsilon);   // torch divides by a python scalar as multiply-by-reciprocal
    const float inv_tau = (float)(1.0 / (double)tau);
    const size_t lds_cached = em_chip_lds(N, J, true);
    if (lds_cached <= 128 * 1024) {       // cost matrix resident in LDS
        static const int em_mode = [] { const char* e = getenv("OGMM_EM_MODE"); return e ? atoi(e) : 2; }();      // 0 generic, 1 registers (bit-identical to 0), 2 + v_exp_f32 (default)
        if (J == 16 && N <= EM_T && em_mode == 2)
            hipLaunchKernelGGL((gmm_em_cached_kernel<16, true>), dim3(C), dim3(EM_T), lds_cached, ogmm::as_stream(stream), xyz, o, ids0, N, J, iters,
                               sk_iters, inv_eps, epsilon, inv_tau, gamma, pi, mu, x);
        else if (J == 16 && N <= EM_T && em_mode == 1)
            hipLaunchKernelGGL((gmm_em_cached_kernel<16, false>), dim3(C), dim3(EM_T), lds_cached, ogmm::as_stream(stream), xyz, o, ids0, N, J, iters,
                               sk_iters, inv_eps, epsilon, inv_tau, gamma, pi, mu, x);
        else
            hipLaunchKernelGGL((gmm_em_cached_kernel<0, false>), dim3(C), dim3(EM_T), lds_cached, ogmm::as_stream(stream), xyz, o, ids0, N, J, iters,
                               sk_iters, inv_eps, epsilon, inv_tau, gamma, pi, mu, x);
        return ogmm::check_launch("ogmm_gmm_em(cached)");
    }
    hipLaunchKernelGGL(gmm_em_kernel, dim3(C), dim3(EM_T), lds, ogmm::as_stream(stream), xyz, o, ids0, N, J, iters, sk_iters, inv_eps,
                       epsilon, inv_tau, gamma, pi, mu, x);
    return ogmm::check_launch("ogmm_gmm_em");
}

namespace {
// J <= 16: two channels per thread (float2 loads: 512 B per wave and row), a row's 16 weights read from LDS as four ds_read_b128,
// eight rows' loads in flight per thread.  Rows are dealt to the 4 row lanes exactly as in gmm_feat_mean_kernel<16> (r = lane, +4, ...)
// and summed in the same order, so the result is bit-identical; the older kernel issued 16 scalar LDS reads per loaded dword and ran at
// a third of the HBM rate.
__global__ __launch_bounds__(256) void gmm_feat_mean16_kernel(const float* __restrict__ gamma, const float* __restrict__ pi,
                                                              const float* __restrict__ feats, int64_t ld, int N, int J, int D,
                                                              float* __restrict__ mu_feat) {
    __shared__ __attribute__((aligned(16))) float gs[128][16];
    __shared__ float red[3][16][128];
    const int c = blockIdx.z, cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int d = blockIdx.x * 128 + 2 * cl;
    const bool ok = d + 1 < D;                            // D even (checked on the host)
    const float* __restrict__ F = feats + (int64_t)c * N * ld + (ok ? d : 0);
    const float* __restrict__ G = gamma + (int64_t)c * N * J;
    float a0[16], a1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { a0[j] = 0.0f; a1[j] = 0.0f; }
    for (int n0 = 0; n0 < N; n0 += 128) {
        __syncthreads();
        for (int i = threadIdx.x; i < 128 * 16; i += 256) {
            const int r = i >> 4, j = i & 15;
            gs[r][j] = (n0 + r < N && j < J) ? G[(int64_t)(n0 + r) * J + j] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int r0 = 0; r0 < 128; r0 += 64) {           // 16 rows of this row lane per trip (round 5: 8 in flight per thread left the kernel at 3.3 TB/s)
            float2 f[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // unconditional loads (rows past the end are clamped: their weights in gs are zero), so that all sixteen are in flight
                const int r = min(n0 + r0 + rl + 4 * u, N - 1);
                f[u] = *reinterpret_cast<const float2*>(F + (int64_t)r * ld);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4* g4 = reinterpret_cast<const float4*>(gs[r0 + rl + 4 * u]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 g = g4[q];
                    a0[4 * q] = fmaf(g.x, f[u].x, a0[4 * q]);         a1[4 * q] = fmaf(g.x, f[u].y, a1[4 * q]);
                    a0[4 * q + 1] = fmaf(g.y, f[u].x, a0[4 * q + 1]); a1[4 * q + 1] = fmaf(g.y, f[u].y, a1[4 * q + 1]);
                    a0[4 * q + 2] = fmaf(g.z, f[u].x, a0[4 * q + 2]); a1[4 * q + 2] = fmaf(g.z, f[u].y, a1[4 * q + 2]);
                    a0[4 * q + 3] = fmaf(g.w, f[u].x, a0[4 * q + 3]); a1[4 * q + 3] = fmaf(g.w, f[u].y, a1[4 * q + 3]);
                }
            }
        }
    }
    // (red[0] + red[1]) + (red[2] + red[3]) as in the older kernel; row lane 0 keeps its own partial in registers
    __syncthreads();
    if (rl > 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { red[rl - 1][j][2 * cl] = a0[j]; red[rl - 1][j][2 * cl + 1] = a1[j]; }
    }
    __syncthreads();
    if (rl == 0 && ok) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j >= J) break;
            const float npi = pi[(int64_t)c * J + j] * (float)N + 1e-5f;
            const float s0 = (a0[j] + red[0][j][2 * cl]) + (red[1][j][2 * cl] + red[2][j][2 * cl]);
            const float s1 = (a1[j] + red[0][j][2 * cl + 1]) + (red[1][j][2 * cl + 1] + red[2][j][2 * cl + 1]);
            *reinterpret_cast<float2*>(&mu_feat[((int64_t)c * J + j) * D + d]) = make_float2(s0 / npi, s1 / npi);
        }
    }
}
}  // namespace

extern "C" int ogmm_gmm_feat_mean(const float* gamma, const float* pi, const float* feats, int64_t ld, int C, int N, int J, int D,
                                  float* mu_feat, void* stream) {
    OGMM_REQUIRE(gamma && pi && feats && mu_feat && C > 0 && N > 0 && J > 0 && D > 0 && ld >= D, "ogmm_gmm_feat_mean: null pointer or bad sizes");
    static const bool valu64 = [] { const char* e = getenv("OGMM_FEAT_MEAN_VALU"); return e && e[0] == '1'; }();      // A/B: the VALU form
    if (J > 16 && J <= 64 && !valu64)
        hipLaunchKernelGGL(gmm_feat_mean_mfma_kernel, dim3((D + 255) / 256, C), dim3(256), 0, ogmm::as_stream(stream), gamma, pi, feats, ld, N, J, D,
                           mu_feat);
    else if (J > 16)
        hipLaunchKernelGGL(gmm_feat_mean_kernel<64>, dim3((D + 63) / 64, (J + 63) / 64, C), dim3(256), 0, ogmm::as_stream(stream), gamma, pi, feats,
                           ld, N, J, D, mu_feat);
    else if (D % 2 == 0 && ld % 2 == 0 && reinterpret_cast<uintptr_t>(feats) % 8 == 0 && reinterpret_cast<uintptr_t>(mu_feat) % 8 == 0)
        hipLaunchKernelGGL(gmm_feat_mean16_kernel, dim3((D + 127) / 128, 1, C), dim3(256), 0, ogmm::as_stream(stream), gamma, pi, feats, ld, N, J, D,
                           mu_feat);
    else
        hipLaunchKernelGGL(gmm_feat_mean_kernel<16>, dim3((D + 63) / 64, 1, C), dim3(256), 0, ogmm::as_stream(stream), gamma, pi, feats,
                           ld, N, J, D, mu_feat);
    return ogmm::check_launch("ogmm_gmm_feat_mean");
}

extern "C" int ogmm_match_kabsch(const float* mu_s, const float* mu_t, const float* f_s, const float* f_t, int B, int J, int D,
                                 float temperature, float* R, float* t, float* scores, void* stream) {
    OGMM_REQUIRE(mu_s && mu_t && f_s && f_t && R && t && B > 0 && J > 0 && D > 0 && temperature > 0, "ogmm_match_kabsch: null pointer or bad sizes");
    OGMM_REQUIRE(J <= 128, "ogmm_match_kabsch: at most 128 clusters per cloud (8 x 8 similarity block per thread), got %d", J);
    const size_t lds = ((size_t)J * J + 6 * (size_t)J + 2 * 64 * (size_t)(J + 4)) * sizeof(float);
    OGMM_REQUIRE(lds <= 160 * 1024, "ogmm_match_kabsch: J=%d too large for LDS", J);
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(match_kabsch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(match_kabsch_kernel, dim3(B), dim3(256), lds, ogmm::as_stream(stream), mu_s, mu_t, f_s, f_t, J, D,
                       (float)(1.0 / (double)temperature), R, t, scores);
    return ogmm::check_launch("ogmm_match_kabsch");
}

extern "C" int ogmm_kabsch(const float* src, const float* corr, const float* w, int B, int J, float* R, float* t, void* stream) {
    OGMM_REQUIRE(src && corr && w && R && t && B > 0 && J > 0, "ogmm_kabsch: null pointer or bad sizes");
    hipLaunchKernelGGL(kabsch_kernel, dim3((B + 63) / 64), dim3(64), 0, ogmm::as_stream(stream), src, corr, w, B, J, R, t);
    return ogmm::check_launch("ogmm_kabsch");
}

extern "C" int ogmm_clu_infonce(const float* xyz, const float* mu, const float* feats, int64_t ld, const float* mu_feat, int C, int N,
                                int J, int D, float tau, float* row_loss, int32_t* near, void* stream) {
    OGMM_REQUIRE(xyz && mu && feats && mu_feat && row_loss && near && C > 0 && N > 0 && J > 1 && D > 0 && tau > 0,
                 "ogmm_clu_infonce: null pointer or bad sizes (J must be >= 2)");
    OGMM_REQUIRE(D <= 1024, "ogmm_clu_infonce: at most 1024 feature channels (a row lives in 16 registers per lane), got %d", D);
    hipStream_t s = ogmm::as_stream(stream);
    hipLaunchKernelGGL(nearest_point_kernel, dim3(J, C), dim3(256), 0, s, xyz, mu, N, J, near);
    hipLaunchKernelGGL(infonce_rows_kernel, dim3(J, C), dim3(256), 4 * (size_t)J * sizeof(float), s, feats, ld, mu_feat, near, N, J, D,
                       (float)(1.0 / (double)tau), row_loss);
    return ogmm::check_launch("ogmm_clu_infonce");
}

extern "C" int ogmm_kabsch_bwd(const float* src, const float* corr, const float* w, int B, int J, const float* gR, const float* gt,
                               float* g_src, float* g_corr, float* g_w, void* stream) {
    OGMM_REQUIRE(src && corr && w && B > 0 && J > 0, "ogmm_kabsch_bwd: null pointer or empty input");
    hipLaunchKernelGGL(kabsch_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, ogmm::as_stream(stream), src, corr, w, B, J, gR, gt, g_src, g_corr, g_w);
    return ogmm::check_launch("ogmm_kabsch_bwd");
}

extern "C" int ogmm_nearest_point(const float* xyz, const float* mu, int C, int N, int J, int32_t* near, void* stream) {
    OGMM_REQUIRE(xyz && mu && near && C > 0 && N > 0 && J > 0, "ogmm_nearest_point: null pointer or empty input");
    hipLaunchKernelGGL(nearest_point_kernel, dim3(J, C), dim3(256), 0, ogmm::as_stream(stream), xyz, mu, N, J, near);
    return ogmm::check_launch("ogmm_nearest_point");
}

extern "C" int ogmm_rotation_from_cov(const float* M, int B, float* R, void* stream) {
    OGMM_REQUIRE(M && R && B > 0, "ogmm_rotation_from_cov: null pointer or empty input");
    hipLaunchKernelGGL(rotation_from_cov_kernel, dim3((B + 63) / 64), dim3(64), 0, ogmm::as_stream(stream), M, B, R);
    return ogmm::check_launch("ogmm_rotation_from_cov");
}
