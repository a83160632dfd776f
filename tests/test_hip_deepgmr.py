"""The DeepGMR baseline on the kernels of the main path, against the reference's baseline/deepgmr.py run on
CPU (tests/golden/make_golden_deepgmr.py)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from ogmm_amd import metric, synth
from ogmm_amd.deepgmr import DeepGMR, state_spec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deepgmr_b2_n512_j16.npz")


def test_state_dict_keys_match_reference():
    fx = np.load(GOLD)
    assert sorted(k for k, _ in state_spec(512, 16)) == sorted(str(k) for k in fx["keys"])
    m = DeepGMR(512, 16, Namespace(gnn_k=20, overlap_radius=0.035))
    assert sorted(m.state_dict().keys()) == sorted(str(k) for k in fx["keys"])
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 64), torch.zeros(1, 3, 64))          # CPU tensors: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_forward_matches_reference(precision):
    fx = np.load(GOLD)
    cfg = Namespace(gnn_k=20, overlap_radius=0.035, precision=precision)
    m = DeepGMR(512, 16, cfg)
    synth.fill_state_dict(m.state_dict())
    with torch.no_grad():
        m.state_dict()["cluster.net.6.weight"].mul_(float(fx["c6_scale"]))
    m = m.to("cuda:0").eval()
    src, tgt = torch.from_numpy(fx["src"]).cuda(), torch.from_numpy(fx["tgt"]).cuda()
    with torch.no_grad():
        R, second = m(src, tgt)
        R2, t2 = m(src, tgt, is_test=True)
    err = metric.rotation_error_rad(R.cpu(), torch.from_numpy(fx["R"])).max().item()
    print("DEEPGMR-PARITY %s R=%.2e" % (precision, err))
    assert err < 5e-5          # the 3x3 registration matrix has singular values 7e-4 / 6e-5 / 8e-6 here: conditioning ~1e5
    assert torch.equal(second.cpu(), torch.from_numpy(fx["second"]))        # tsfm[:, 3, 0:3] == 0 (sic)
    assert torch.isfinite(R2).all() and torch.isfinite(t2).all()
    assert (torch.det(R2) - 1).abs().max() < 1e-5
