"""Device-side evaluation metrics against the reference's `dcp_metrics` run on CPU
(tests/golden/make_golden_metrics.py)."""
import os

import numpy as np
import pytest
import torch

from ogmm_amd import metric, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_dcp_metrics_match_reference():
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_b4_n300.npz"))
    t_ = lambda k: torch.from_numpy(fx[k]).to(DEV)                    # noqa: E731
    got = metric.dcp_metrics(t_("src").transpose(1, 2).contiguous(), t_("tgt").transpose(1, 2).contiguous(), t_("R_gt"), t_("t_gt"), t_("R_pre"), t_("t_pre"))
    for k in (f[2:] for f in fx.files if f.startswith("m/")):
        want = np.asarray(fx["m/" + k], dtype=np.float64)
        have = got[k].cpu().double().numpy()
        tol = 2e-3 if k in ("r_mse", "r_mae", "err_r_deg") else 2e-6     # degrees from fp32 acos / atan2 of fp32 matrices
        assert np.abs(have - want).max() <= tol * max(1.0, np.abs(want).max()), (k, have, want)
    s = metric.summarize_metrics(got)
    for k in (f[2:] for f in fx.files if f.startswith("s/")):
        assert abs(s[k] - float(fx["s/" + k])) <= 2e-3 * max(1.0, abs(float(fx["s/" + k]))), k


@pytest.mark.parametrize("Na,Nb", [(300, 300), (1500, 2048), (5, 3000)])
def test_min_sqdist(Na, Nb):
    g = torch.Generator().manual_seed(Na)
    a, b = torch.rand(3, Na, 3, generator=g).to(DEV), torch.rand(3, Nb, 3, generator=g).to(DEV)
    want = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1).min(dim=2)[0]
    assert torch.allclose(ops.min_sqdist(a, b), want, rtol=1e-6, atol=1e-9)
