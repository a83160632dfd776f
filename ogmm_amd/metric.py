"""Parity metrics of the registration path.

`rotation_error` / `translation_error` follow /root/reference/lib/metric.py:85-93 (degrees via fp32 acos of the
Frobenius inner product; Euclidean norm of the translation difference).  The fp32 acos cannot resolve angles below
about 3.5e-4 rad, so the 1e-5 rad parity bar is measured with `rotation_error_rad`, an fp64 chordal formula that is
exact for small angles: angle = 2*asin(||R1-R2||_F / (2*sqrt(2))).
"""
import math

import torch


def rotation_error(rot1, rot2):
    """degrees, [B]; same arithmetic as the reference (lib/metric.py:85-88)"""
    if rot1.shape != rot2.shape:
        raise ValueError("rotation_error: shape mismatch %s vs %s" % (tuple(rot1.shape), tuple(rot2.shape)))
    if rot1.is_cuda:
        # nine products per pair: an elementwise product and a row sum.  (torch lowers the reference's einsum to a batched [1 x 9][9 x 1] GEMM -- the one vendor
        # GEMM kernel the training step still launched in round 6's first trace; the fp32 acos below resolves 3.5e-4 rad at best, so the order of nine additions
        # is far below what this diagnostic can show)
        inner = (rot1 * rot2).flatten(1).sum(dim=1)
    else:
        inner = torch.einsum('bij,bij->b', rot1, rot2)      # same contraction (and summation order) as the reference
    return torch.arccos(torch.clamp((inner - 1) / 2, -1.0, 1.0)) * 180 / math.pi


def translation_error(t1, t2):
    """[B]; lib/metric.py:91-93"""
    if t1.shape != t2.shape:
        raise ValueError("translation_error: shape mismatch %s vs %s" % (tuple(t1.shape), tuple(t2.shape)))
    return torch.norm(t1 - t2, dim=1)


def rotation_error_rad(rot1, rot2):
    """radians, [B], fp64; resolves down to ~1e-8 rad"""
    d = (rot1.double() - rot2.double()).flatten(1).norm(dim=1)
    return 2.0 * torch.asin(torch.clamp(d / (2.0 * math.sqrt(2.0)), max=1.0))


# ---------------------------------------------------------------------------------------------- evaluation metrics
def euler_zyx_deg(R):
    """scipy `Rotation.from_matrix(R).as_euler('zyx', degrees=True)` (lib/metric.py:166-171) in closed form, on the tensor's
    device: extrinsic z-y-x, i.e. R = Rx(c) Ry(b) Rz(a) -> [a, b, c] with b = asin(R02), a = atan2(-R01, R00),
    c = atan2(-R12, R22).  (At gimbal lock, |R02| = 1, scipy sets the third angle to zero; that measure-zero case is not
    reproduced.)"""
    b = torch.asin(torch.clamp(R[:, 0, 2], -1.0, 1.0))
    a = torch.atan2(-R[:, 0, 1], R[:, 0, 0])
    c = torch.atan2(-R[:, 1, 2], R[:, 2, 2])
    return torch.stack([a, b, c], dim=1) * (180.0 / math.pi)


def dcp_metrics(src, tgt, rot_gt, transl_gt, rot_pre, transl_pre, r_th=1.0, t_th=0.1):
    """lib/metric.py:197-245 without leaving the device (the reference round-trips through numpy / scipy and builds three
    [B,N,N] matrices).  src, tgt [B,N,3]; rot_* [B,3,3]; transl_* [B,3].  Returns the reference's dict with torch tensors."""
    from . import ops
    rot_pre, transl_pre, rot_gt, transl_gt = rot_pre.detach(), transl_pre.detach(), rot_gt.detach(), transl_gt.detach()
    e_pre, e_gt = euler_zyx_deg(rot_pre), euler_zyx_deg(rot_gt)
    r_mse, r_mae = ((e_gt - e_pre) ** 2).mean(dim=1), (e_gt - e_pre).abs().mean(dim=1)
    t_mse, t_mae = ((transl_gt - transl_pre) ** 2).mean(dim=1), (transl_gt - transl_pre).abs().mean(dim=1)
    # residual motion gt^-1 * pred (lib/metric.py:207-213, :20-45, :174-190)
    rel = torch.bmm(rot_gt.transpose(1, 2), rot_pre)
    trace = rel[:, 0, 0] + rel[:, 1, 1] + rel[:, 2, 2]
    err_r = torch.acos(torch.clamp(0.5 * (trace - 1), min=-1.0, max=1.0)) * 180.0 / math.pi
    err_t = torch.bmm(rot_gt.transpose(1, 2), (transl_pre - transl_gt)[:, :, None])[:, :, 0].norm(dim=-1)
    src_pre = torch.baddbmm(transl_pre[:, None, :], src, rot_pre.transpose(1, 2))        # datasets/datautils.py `transform`
    src_gt = torch.baddbmm(transl_gt[:, None, :], src, rot_gt.transpose(1, 2))
    d_st, d_ts = ops.min_sqdist(src_pre, tgt), ops.min_sqdist(tgt, src_pre)
    clip = 0.1
    return {"r_mse": r_mse, "r_mae": r_mae, "t_mse": t_mse, "t_mae": t_mae, "err_r_deg": err_r, "err_t": err_t,
            "chamfer_dist": d_st.mean(dim=1) + d_ts.mean(dim=1),
            "pcab_dist": ops.min_sqdist(src_pre, src_gt).mean(dim=1),
            "clip_chamfer_dist": torch.sqrt(d_st).clamp(max=clip).mean(dim=1) + torch.sqrt(d_ts).clamp(max=clip).mean(dim=1),
            "n_correct": ((r_mae < r_th) & (t_mae < t_th)).float(),
            "pre_transform": torch.cat([rot_pre, transl_pre[:, :, None]], dim=2),
            "gt_transform": torch.cat([rot_gt, transl_gt[:, :, None]], dim=2)}


def summarize_metrics(metrics):
    """lib/metric.py:248-264: means over all instances (rmse for *mse keys; mean and rmse for err_* keys)"""
    out = {}
    for k, v in metrics.items():
        v = v.double() if torch.is_tensor(v) else torch.as_tensor(v, dtype=torch.float64)
        if k.endswith("mse"):
            out[k[:-3] + "rmse"] = float(torch.sqrt(v.mean()))
        elif k.startswith("err"):
            out[k + "_mean"] = float(v.mean())
            out[k + "_rmse"] = float(torch.sqrt((v ** 2).mean()))
        elif k.endswith("nomean"):
            out[k] = v
        else:
            out[k] = float(v.mean())
    return out
