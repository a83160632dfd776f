"""Generates tests/golden/metrics_b4_n300.npz by running the reference's `dcp_metrics` + `summarize_metrics`
(lib/metric.py:197-264) on CPU.  The function hard-codes `.cuda()` (lib/metric.py:227); for this run `Tensor.cuda` is
patched to the identity.  Inputs: synthetic partial pairs, ground-truth motion, and a perturbed motion as the prediction."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_harness import import_reference  # noqa: E402
from ogmm_amd import synth                       # noqa: E402


def main():
    import_reference()
    import lib.metric as ref_metric
    B, N = 4, 300
    src, tgt, R, t = synth.make_batch(60, B, N, "partial")
    g = torch.Generator().manual_seed(9)
    ang = torch.tensor([0.002, 0.01, 0.05, 0.3])
    ax = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    K = torch.zeros(B, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    dR = torch.eye(3)[None] + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K)
    R_pre = dR @ R
    t_pre = t + torch.tensor([0.001, 0.01, 0.05, 0.2])[:, None] * torch.randn(B, 3, generator=g)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        m = ref_metric.dcp_metrics(src.transpose(1, 2), tgt.transpose(1, 2), R, t, R_pre, t_pre)
    finally:
        torch.Tensor.cuda = real_cuda
    summ = ref_metric.summarize_metrics({k: v for k, v in m.items()})
    fx = dict(src=src.numpy(), tgt=tgt.numpy(), R_gt=R.numpy(), t_gt=t.numpy(), R_pre=R_pre.numpy(), t_pre=t_pre.numpy())
    for k, v in m.items():
        fx["m/" + k] = np.asarray(v, dtype=np.float64)
    for k, v in summ.items():
        fx["s/" + k] = np.asarray(v, dtype=np.float64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_b4_n300.npz")
    np.savez_compressed(path, **fx)
    print(path, {k: np.round(np.asarray(v, dtype=np.float64).reshape(-1)[:4], 5).tolist() for k, v in m.items() if "transform" not in k})


if __name__ == "__main__":
    main()
