// Point-to-point ICP refinement of `GMMReg.forward(is_test=True)`: lib/o3dutils.py:172-214 (`reg_solver` ->
// `refine_registration` -> open3d `registration_icp(..., TransformationEstimationPointToPoint())`, called at
// models/gmmreg.py:115-117 with max correspondence distance 2 * overlap_radius and the network's (R, t) as the initial guess).
//
// open3d itself is a third-party dependency that is absent from the reference tree (version unpinned, README.md:35), so this
// follows its published algorithm (RegistrationICP, default ICPConvergenceCriteria: relative_fitness = relative_rmse = 1e-6,
// max_iteration = 30):
//     result = evaluate(T)            nearest target point within the radius for every source point -> correspondence set,
//                                     fitness = |set| / N_src, inlier_rmse = sqrt(mean squared distance)
//     repeat max_iteration times:     update = Umeyama(without scale) on the set;  T = update * T;  new = evaluate(T);
//                                     stop when |fitness - new.fitness| and |rmse - new.rmse| are both below 1e-6
// Everything is fp64 like open3d.  One workgroup per pair, the whole loop on chip: target points are cached in LDS
// (broadcast reads: every lane compares its own source point with the same target point), the 17 correspondence sums
// (count, sum d^2, sum p, sum q, sum p q^T) are reduced with wave shuffles, lane 0 solves the 3x3 problem (Jacobi, shared
// with the Kabsch kernel).  The reference runs this per pair on the CPU and copies back (a device sync per batch).
#include "ogmm_common.h"

namespace ogmm_icp {

using namespace ogmm;

constexpr int T = 512;                      // threads per pair
constexpr int MAX_TGT = 8192;               // target points cached in LDS (96 KB)

__device__ void jacobi3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        if (fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]) < 1e-300) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (fabs(A[p][q]) < 1e-300) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
            for (int k = 0; k < 3; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = cs * a - sn * b; A[k][q] = sn * a + cs * b; }
            for (int k = 0; k < 3; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = cs * a - sn * b; A[q][k] = sn * a + cs * b; }
            for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
        }
    }
}

// proper rotation R maximising tr(R^T ... ) for cov[a][b] = sum (p_a - pbar_a)(q_b - qbar_b)  (R p ~ q), as Umeyama without scale
__device__ void rotation_from_cov(const double cov[3][3], double R[3][3]) {
    double AtA[3][3], V[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) AtA[i][j] = cov[0][i] * cov[0][j] + cov[1][i] * cov[1][j] + cov[2][i] * cov[2][j];
    jacobi3(AtA, V);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b)
        if (AtA[ord[b]][ord[b]] > AtA[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double v[2][3], u[2][3];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 3; ++k) v[i][k] = V[k][ord[i]];
    for (int i = 0; i < 2; ++i) {
        for (int k = 0; k < 3; ++k) u[i][k] = cov[k][0] * v[i][0] + cov[k][1] * v[i][1] + cov[k][2] * v[i][2];
        if (i == 1) {
            const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
            for (int k = 0; k < 3; ++k) u[1][k] -= d * u[0][k];
        }
        const double nrm = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
        const double inv = nrm > 0 ? 1.0 / nrm : 0.0;
        for (int k = 0; k < 3; ++k) u[i][k] *= inv;
    }
    const double v3[3] = {v[0][1] * v[1][2] - v[0][2] * v[1][1], v[0][2] * v[1][0] - v[0][0] * v[1][2], v[0][0] * v[1][1] - v[0][1] * v[1][0]};
    const double u3[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2], u[0][0] * u[1][1] - u[0][1] * u[1][0]};
    // cov = U S V^T in the (p, q) ordering above; the rotation taking p to q is V U^T = sum_i v_i u_i^T with u_i = cov v_i / s_i
    // being the LEFT vectors: here u_i (a combination of cov's columns) lives in p-space and v_i in q-space.
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R[a][b] = v[0][a] * u[0][b] + v[1][a] * u[1][b] + v3[a] * u3[b];
}

__global__ __launch_bounds__(T) void icp_kernel(const float* __restrict__ src, const float* __restrict__ tgt, int N, int Nt,
                                                const float* __restrict__ R0, const float* __restrict__ t0, double max_d2, int max_iter,
                                                double eps_fit, double eps_rmse, float* __restrict__ R_out, float* __restrict__ t_out,
                                                float* __restrict__ fit_out, float* __restrict__ rmse_out, int* __restrict__ iters_out) {
    extern __shared__ float s_tgt[];                      // [Nt][3]
    __shared__ double s_red[T / 64][17];
    __shared__ double s_T[12];                            // current R (row-major 9) and t (3)
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ ps = src + (int64_t)b * N * 3;
    const float* __restrict__ pt = tgt + (int64_t)b * Nt * 3;
    for (int i = tid; i < Nt * 3; i += T) s_tgt[i] = pt[i];
    if (tid < 9) s_T[tid] = R0 ? (double)R0[b * 9 + tid] : (tid % 4 == 0 ? 1.0 : 0.0);
    if (tid < 3) s_T[9 + tid] = t0 ? (double)t0[b * 3 + tid] : 0.0;
    __syncthreads();

    double prev_fit = 0.0, prev_rmse = 0.0, fit = 0.0, rmse = 0.0;
    int it = 0;
    for (;; ++it) {
        // ---- evaluate the current transformation
        double acc[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) acc[i] = 0.0;
        const double r00 = s_T[0], r01 = s_T[1], r02 = s_T[2], r10 = s_T[3], r11 = s_T[4], r12 = s_T[5], r20 = s_T[6], r21 = s_T[7], r22 = s_T[8];
        const double tx = s_T[9], ty = s_T[10], tz = s_T[11];
        for (int n = tid; n < N; n += T) {
            const double x = ps[3 * n], y = ps[3 * n + 1], z = ps[3 * n + 2];
            const double px = r00 * x + r01 * y + r02 * z + tx, py = r10 * x + r11 * y + r12 * z + ty, pz = r20 * x + r21 * y + r22 * z + tz;
            double best = 1e300;
            int bj = -1;
            for (int j = 0; j < Nt; ++j) {
                const double dx = px - (double)s_tgt[3 * j], dy = py - (double)s_tgt[3 * j + 1], dz = pz - (double)s_tgt[3 * j + 2];
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < best) { best = d2; bj = j; }
            }
            if (bj >= 0 && best <= max_d2) {
                const double qx = s_tgt[3 * bj], qy = s_tgt[3 * bj + 1], qz = s_tgt[3 * bj + 2];
                acc[0] += 1.0; acc[1] += best;
                acc[2] += px; acc[3] += py; acc[4] += pz;
                acc[5] += qx; acc[6] += qy; acc[7] += qz;
                acc[8] += px * qx; acc[9] += px * qy; acc[10] += px * qz;
                acc[11] += py * qx; acc[12] += py * qy; acc[13] += py * qz;
                acc[14] += pz * qx; acc[15] += pz * qy; acc[16] += pz * qz;
            }
        }
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const double v = wave_sum_d(acc[i]);
            if (lane == 0) s_red[wave][i] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double s[17];
            for (int i = 0; i < 17; ++i) {
                s[i] = 0.0;
                for (int w = 0; w < T / 64; ++w) s[i] += s_red[w][i];
            }
            const double cnt = s[0];
            fit = cnt / (double)N;
            rmse = cnt > 0 ? sqrt(s[1] / cnt) : 0.0;
            int stop = 0;
            if (it > 0 && fabs(prev_fit - fit) < eps_fit && fabs(prev_rmse - rmse) < eps_rmse) stop = 1;
            if (it >= max_iter) stop = 1;
            if (!stop && cnt > 0) {
                // Umeyama without scale on the correspondence set: p (moved source) -> q (target)
                double pb[3] = {s[2] / cnt, s[3] / cnt, s[4] / cnt}, qb[3] = {s[5] / cnt, s[6] / cnt, s[7] / cnt};
                double cov[3][3], Ru[3][3];
                for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) cov[a][c] = s[8 + a * 3 + c] / cnt - pb[a] * qb[c];
                rotation_from_cov(cov, Ru);
                double tu[3];
                for (int a = 0; a < 3; ++a) tu[a] = qb[a] - (Ru[a][0] * pb[0] + Ru[a][1] * pb[1] + Ru[a][2] * pb[2]);
                double Rn[9], tn[3];
                for (int a = 0; a < 3; ++a) {
                    for (int c = 0; c < 3; ++c) Rn[a * 3 + c] = Ru[a][0] * s_T[c] + Ru[a][1] * s_T[3 + c] + Ru[a][2] * s_T[6 + c];
                    tn[a] = Ru[a][0] * s_T[9] + Ru[a][1] * s_T[10] + Ru[a][2] * s_T[11] + tu[a];
                }
                for (int i = 0; i < 9; ++i) s_T[i] = Rn[i];
                for (int i = 0; i < 3; ++i) s_T[9 + i] = tn[i];
            }
            prev_fit = fit;
            prev_rmse = rmse;
            s_flag = stop;
        }
        __syncthreads();
        if (s_flag) break;
    }
    if (tid == 0) {
        for (int i = 0; i < 9; ++i) R_out[b * 9 + i] = (float)s_T[i];
        for (int i = 0; i < 3; ++i) t_out[b * 3 + i] = (float)s_T[9 + i];
        if (fit_out) fit_out[b] = (float)fit;
        if (rmse_out) rmse_out[b] = (float)rmse;
        if (iters_out) iters_out[b] = it;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Grid-wide variant for small batches: with one workgroup per pair a batch of 64 pairs uses a quarter of the chip.  Here every ICP
// iteration is two launches -- `icp_eval_kernel` (SPLIT workgroups per pair, each reducing its share of the source points against the
// LDS-cached target cloud into 17 partial sums) and `icp_update_kernel` (one lane per pair: sums the partials in a fixed order,
// convergence test, Umeyama update) -- all enqueued by one call; finished pairs are skipped through a per-pair flag.  Same arithmetic
// as icp_kernel up to the summation order of the 17 sums.
struct IcpState {            // per pair, in the workspace
    double T[12];
    double prev_fit, prev_rmse, fit, rmse;
    int done, iters;
};

__global__ __launch_bounds__(256) void icp_eval_kernel(const float* __restrict__ src, const float* __restrict__ tgt, int N, int Nt, int split,
                                                       double max_d2, const IcpState* __restrict__ st, double* __restrict__ part) {
    extern __shared__ float s_tgt[];
    __shared__ double s_red[4][17];
    const int b = blockIdx.y, p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (st[b].done) return;
    const float* __restrict__ ps = src + (int64_t)b * N * 3;
    const float* __restrict__ pt = tgt + (int64_t)b * Nt * 3;
    for (int i = tid; i < Nt * 3; i += 256) s_tgt[i] = pt[i];
    const double* T = st[b].T;
    const double r00 = T[0], r01 = T[1], r02 = T[2], r10 = T[3], r11 = T[4], r12 = T[5], r20 = T[6], r21 = T[7], r22 = T[8];
    const double tx = T[9], ty = T[10], tz = T[11];
    __syncthreads();
    double acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.0;
    const int per = (N + split - 1) / split;
    const int n_hi = min(N, (p + 1) * per);
    for (int n = p * per + tid; n < n_hi; n += 256) {
        const double x = ps[3 * n], y = ps[3 * n + 1], z = ps[3 * n + 2];
        const double px = r00 * x + r01 * y + r02 * z + tx, py = r10 * x + r11 * y + r12 * z + ty, pz = r20 * x + r21 * y + r22 * z + tz;
        double best = 1e300;
        int bj = -1;
        for (int j = 0; j < Nt; ++j) {
            const double dx = px - (double)s_tgt[3 * j], dy = py - (double)s_tgt[3 * j + 1], dz = pz - (double)s_tgt[3 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bj = j; }
        }
        if (bj >= 0 && best <= max_d2) {
            const double qx = s_tgt[3 * bj], qy = s_tgt[3 * bj + 1], qz = s_tgt[3 * bj + 2];
            acc[0] += 1.0; acc[1] += best;
            acc[2] += px; acc[3] += py; acc[4] += pz;
            acc[5] += qx; acc[6] += qy; acc[7] += qz;
            acc[8] += px * qx; acc[9] += px * qy; acc[10] += px * qz;
            acc[11] += py * qx; acc[12] += py * qy; acc[13] += py * qz;
            acc[14] += pz * qx; acc[15] += pz * qy; acc[16] += pz * qz;
        }
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const double v = wave_sum_d(acc[i]);
        if (lane == 0) s_red[wave][i] = v;
    }
    __syncthreads();
    if (tid < 17) part[((int64_t)b * split + p) * 17 + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

__global__ void icp_update_kernel(int B, int N, int split, int it, int max_iter, double eps_fit, double eps_rmse, IcpState* __restrict__ st,
                                  const double* __restrict__ part, float* __restrict__ R_out, float* __restrict__ t_out,
                                  float* __restrict__ fit_out, float* __restrict__ rmse_out, int* __restrict__ iters_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    IcpState& S = st[b];
    if (S.done) return;
    double s[17];
    for (int i = 0; i < 17; ++i) {
        s[i] = 0.0;
        for (int p = 0; p < split; ++p) s[i] += part[((int64_t)b * split + p) * 17 + i];
    }
    const double cnt = s[0];
    const double fit = cnt / (double)N, rmse = cnt > 0 ? sqrt(s[1] / cnt) : 0.0;
    bool stop = it > 0 && fabs(S.prev_fit - fit) < eps_fit && fabs(S.prev_rmse - rmse) < eps_rmse;
    if (it >= max_iter) stop = true;
    if (!stop && cnt > 0) {
        const double pb[3] = {s[2] / cnt, s[3] / cnt, s[4] / cnt}, qb[3] = {s[5] / cnt, s[6] / cnt, s[7] / cnt};
        double cov[3][3], Ru[3][3], tu[3], Rn[9], tn[3];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) cov[a][c] = s[8 + a * 3 + c] / cnt - pb[a] * qb[c];
        rotation_from_cov(cov, Ru);
        for (int a = 0; a < 3; ++a) tu[a] = qb[a] - (Ru[a][0] * pb[0] + Ru[a][1] * pb[1] + Ru[a][2] * pb[2]);
        for (int a = 0; a < 3; ++a) {
            for (int c = 0; c < 3; ++c) Rn[a * 3 + c] = Ru[a][0] * S.T[c] + Ru[a][1] * S.T[3 + c] + Ru[a][2] * S.T[6 + c];
            tn[a] = Ru[a][0] * S.T[9] + Ru[a][1] * S.T[10] + Ru[a][2] * S.T[11] + tu[a];
        }
        for (int i = 0; i < 9; ++i) S.T[i] = Rn[i];
        for (int i = 0; i < 3; ++i) S.T[9 + i] = tn[i];
    }
    S.prev_fit = fit;
    S.prev_rmse = rmse;
    if (stop) {
        S.done = 1;
        for (int i = 0; i < 9; ++i) R_out[b * 9 + i] = (float)S.T[i];
        for (int i = 0; i < 3; ++i) t_out[b * 3 + i] = (float)S.T[9 + i];
        if (fit_out) fit_out[b] = (float)fit;
        if (rmse_out) rmse_out[b] = (float)rmse;
        if (iters_out) iters_out[b] = it;
    }
}

__global__ void icp_init_kernel(int B, const float* __restrict__ R0, const float* __restrict__ t0, IcpState* __restrict__ st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int i = 0; i < 9; ++i) st[b].T[i] = R0 ? (double)R0[b * 9 + i] : (i % 4 == 0 ? 1.0 : 0.0);
    for (int i = 0; i < 3; ++i) st[b].T[9 + i] = t0 ? (double)t0[b * 3 + i] : 0.0;
    st[b].prev_fit = st[b].prev_rmse = 0.0;
    st[b].done = 0;
    st[b].iters = 0;
}

}  // namespace ogmm_icp

static int icp_split(int B, int N) {             // workgroups per pair of the grid-wide variant: fill ~1024 workgroup slots, >= 64 source points each
    int split = 1024 / (B > 0 ? B : 1);
    split = split < 1 ? 1 : (split > 16 ? 16 : split);
    while (split > 1 && (N + split - 1) / split < 64) --split;
    return split;
}

extern "C" int64_t ogmm_icp_workspace_bytes(int B, int N) {
    return (int64_t)B * (int64_t)sizeof(ogmm_icp::IcpState) + (int64_t)B * icp_split(B, N) * 17 * (int64_t)sizeof(double) + 256;
}

extern "C" int ogmm_icp_point_to_point_ws(const float* src, const float* tgt, int B, int N, int Nt, const float* R0, const float* t0,
                                          float max_corr_dist, int max_iter, double rel_fitness, double rel_rmse,
                                          float* R, float* t, float* fitness, float* rmse, int* iters, void* workspace, void* stream) {
    using namespace ogmm;
    OGMM_REQUIRE(src && tgt && R && t && workspace && B > 0 && N > 0 && Nt > 0 && max_iter >= 0 && max_corr_dist > 0, "ogmm_icp_point_to_point_ws: null pointer or bad sizes");
    OGMM_REQUIRE(Nt <= ogmm_icp::MAX_TGT, "ogmm_icp_point_to_point_ws: at most %d target points per cloud (LDS cache), got %d", ogmm_icp::MAX_TGT, Nt);
    OGMM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "ogmm_icp_point_to_point_ws: workspace must be 16-byte aligned");
    const int split = icp_split(B, N);
    auto* st = static_cast<ogmm_icp::IcpState*>(workspace);
    double* part = reinterpret_cast<double*>(static_cast<char*>(workspace) + ((size_t)B * sizeof(ogmm_icp::IcpState) + 15) / 16 * 16);
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ogmm_icp::icp_eval_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                ogmm_icp::MAX_TGT * 3 * sizeof(float)) != hipSuccess)
            return fail("ogmm_icp_point_to_point_ws: cannot raise the dynamic LDS limit");
    }
    hipStream_t s = as_stream(stream);
    const double md = (double)max_corr_dist;
    const size_t lds = (size_t)Nt * 3 * sizeof(float);
    const dim3 pairs((B + 63) / 64);
    hipLaunchKernelGGL(ogmm_icp::icp_init_kernel, pairs, dim3(64), 0, s, B, R0, t0, st);
    for (int it = 0; it <= max_iter; ++it) {
        hipLaunchKernelGGL(ogmm_icp::icp_eval_kernel, dim3(split, B), dim3(256), lds, s, src, tgt, N, Nt, split, md * md, st, part);
        hipLaunchKernelGGL(ogmm_icp::icp_update_kernel, pairs, dim3(64), 0, s, B, N, split, it, max_iter, rel_fitness, rel_rmse, st, part, R, t,
                           fitness, rmse, iters);
    }
    return check_launch("ogmm_icp_point_to_point_ws");
}

extern "C" int ogmm_icp_point_to_point(const float* src, const float* tgt, int B, int N, int Nt, const float* R0, const float* t0,
                                       float max_corr_dist, int max_iter, double rel_fitness, double rel_rmse,
                                       float* R, float* t, float* fitness, float* rmse, int* iters, void* stream) {
    using namespace ogmm;
    OGMM_REQUIRE(src && tgt && R && t && B > 0 && N > 0 && Nt > 0 && max_iter >= 0 && max_corr_dist > 0, "ogmm_icp_point_to_point: null pointer or bad sizes");
    OGMM_REQUIRE(Nt <= ogmm_icp::MAX_TGT, "ogmm_icp_point_to_point: at most %d target points per cloud (LDS cache), got %d", ogmm_icp::MAX_TGT, Nt);
    const size_t lds = (size_t)Nt * 3 * sizeof(float);
    static ogmm::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ogmm_icp::icp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                ogmm_icp::MAX_TGT * 3 * sizeof(float)) != hipSuccess)
            return fail("ogmm_icp_point_to_point: cannot raise the dynamic LDS limit");
    }
    const double md = (double)max_corr_dist;
    hipLaunchKernelGGL(ogmm_icp::icp_kernel, dim3(B), dim3(ogmm_icp::T), lds, as_stream(stream), src, tgt, N, Nt, R0, t0, md * md, max_iter,
                       rel_fitness, rel_rmse, R, t, fitness, rmse, iters);
    return check_launch("ogmm_icp_point_to_point");
}
