"""CPU ORACLE for the ICP refinement of `GMMReg.forward(is_test=True)` -- TEST INFRASTRUCTURE ONLY.

The reference delegates this step to open3d (`lib/o3dutils.py:172-214`: `reg_solver` -> `refine_registration` ->
`o3d.pipelines.registration.registration_icp(source, target, 2 * voxel_size, init, TransformationEstimationPointToPoint())`),
a third-party dependency that is absent from /root/reference and not installable here (README.md:35 names it without a
version).  **Parity unpinned**: this file restates open3d's published RegistrationICP algorithm in numpy fp64 -- default
ICPConvergenceCriteria (relative_fitness 1e-6, relative_rmse 1e-6, max_iteration 30), nearest neighbour within the radius,
Umeyama-without-scale update, `T = update @ T` -- and is itself validated only by convergence to the ground-truth motion.
"""
import numpy as np


def _evaluate(src_moved, tgt, max_dist):
    d2 = ((src_moved[:, None, :] - tgt[None, :, :]) ** 2).sum(-1)
    j = d2.argmin(1)
    best = d2[np.arange(len(src_moved)), j]
    keep = best <= max_dist * max_dist
    n = int(keep.sum())
    fitness = n / len(src_moved)
    rmse = float(np.sqrt(best[keep].sum() / n)) if n else 0.0
    return np.nonzero(keep)[0], j[keep], fitness, rmse


def _umeyama_no_scale(p, q):
    """proper rotation R and t minimising sum |R p + t - q|^2 (Eigen::umeyama(..., with_scaling=false))"""
    pb, qb = p.mean(0), q.mean(0)
    cov = (q - qb).T @ (p - pb) / len(p)
    U, _, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    return R, qb - R @ pb


def icp_point_to_point(src, tgt, T_init, max_dist, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """src [N,3], tgt [M,3], T_init [4,4] -> (T [4,4], fitness, inlier_rmse, iterations)"""
    src, tgt, T = np.asarray(src, np.float64), np.asarray(tgt, np.float64), np.array(T_init, np.float64)
    moved = src @ T[:3, :3].T + T[:3, 3]
    si, ti, fit, rmse = _evaluate(moved, tgt, max_dist)
    it = 0
    for it in range(1, max_iter + 1):
        if len(si):
            R, t = _umeyama_no_scale(moved[si], tgt[ti])
        else:
            R, t = np.eye(3), np.zeros(3)
        U = np.eye(4)
        U[:3, :3], U[:3, 3] = R, t
        T = U @ T
        moved = moved @ R.T + t
        si, ti, nfit, nrmse = _evaluate(moved, tgt, max_dist)
        done = abs(fit - nfit) < rel_fitness and abs(rmse - nrmse) < rel_rmse
        fit, rmse = nfit, nrmse
        if done:
            break
    return T, fit, rmse, it
