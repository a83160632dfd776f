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
