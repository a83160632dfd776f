"""SE(3) pack/unpack helpers of the registration path.

Mirrors the call surface of the reference's `decompose_trans` / `integrate_trans`
(/root/reference/lib/se3.py:14-52): torch tensors or numpy arrays, single or batched.
"""
import numpy as np
import torch


def decompose_trans(trans):
    """[4,4] or [B,4,4] -> (R [.,3,3], t [.,3,1]); views of the input like the reference (lib/se3.py:14-26)."""
    if trans.ndim not in (2, 3) or tuple(trans.shape[-2:]) != (4, 4):
        raise ValueError("decompose_trans expects [4,4] or [B,4,4], got %s" % (tuple(trans.shape),))
    return trans[..., :3, :3], trans[..., :3, 3:4]


def integrate_trans(R, t):
    """(R [.,3,3], t [.,3,1] or [.,3]) -> homogeneous [.,4,4] (lib/se3.py:29-52).

    The reference builds the batched numpy result from a broadcast [1,4,4] identity, which only works for B=1;
    here every batch size works.  dtype follows the reference: float32 for tensors, float64 for numpy.
    """
    batched = R.ndim == 3
    if isinstance(R, torch.Tensor):
        out = torch.eye(4, device=R.device).repeat(R.shape[0], 1, 1) if batched else torch.eye(4, device=R.device)
        out[..., :3, :3] = R
        out[..., :3, 3:4] = t.reshape(-1, 3, 1) if batched else t.reshape(3, 1)
        return out
    out = np.tile(np.eye(4), (R.shape[0], 1, 1)) if batched else np.eye(4)
    out[..., :3, :3] = R
    out[..., :3, 3:4] = np.reshape(t, (-1, 3, 1)) if batched else np.reshape(t, (3, 1))
    return out
