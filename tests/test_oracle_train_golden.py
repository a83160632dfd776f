"""The oracle's TRAIN mode (batch-statistics BatchNorm, autograd, the loss of train.py:54-72) against fixtures produced
by running the reference's own training step (tests/golden/make_golden_train.py)."""
import numpy as np
import pytest
import torch

from ogmm_amd import gmmreg, synth
from oracle import ogmm_oracle as O
from train_util import TRAIN_CASES, TRAIN_CASES_ENGINE, check_grads, load_train_case, noise_tol, profile_of


@pytest.mark.parametrize("name", TRAIN_CASES + TRAIN_CASES_ENGINE)
def test_oracle_training_step_matches_reference(name):
    torch.set_num_threads(8)
    fx, cfg, (B, N, J, D, top_k) = load_train_case(name)
    P = {k: torch.zeros(s) if "num_batches" not in k else torch.zeros((), dtype=torch.long) for k, s in gmmreg.state_spec(D)}
    synth.fill_state_dict(P, profile=profile_of(fx))
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    src, tgt = torch.from_numpy(fx["src"]), torch.from_numpy(fx["tgt"])
    out = O.forward(P, cfg, src, tgt, torch.from_numpy(fx["fps_starts"]), train=True)
    loss = O.training_loss(out, src, tgt, torch.from_numpy(fx["T_gt"]), torch.from_numpy(fx["src_overlap"]),
                           torch.from_numpy(fx["tgt_overlap"]), 10.0, top_k)
    # bars: the base bar, or 3 x the reference's own train-mode noise on this fixture (recorded by the generator; it matters on the non-degenerate family)
    assert abs(loss.item() - float(fx["loss"])) <= noise_tol(fx, "loss", 2e-6 * abs(float(fx["loss"])))
    assert O.rotation_error_rad(out[0].detach(), torch.from_numpy(fx["R"])).max() < noise_tol(fx, "R", 5e-6)     # input layout changes torch kernel choices (see make_golden_train.py)
    np.testing.assert_allclose(out[2].detach().numpy(), fx["src_o"], atol=noise_tol(fx, "o", 5e-6))
    loss.backward()
    grads = {k: v.grad for k, v in P.items() if v.is_floating_point() and "running" not in k}
    worst = check_grads(fx, grads)
    print("TRAIN-ORACLE %s loss=%.8f worst_grad_err_over_allowed=%.2f" % (name, loss.item(), worst))
    for key in (f[len("stat/"):] for f in fx.files if f.startswith("stat/")):
        np.testing.assert_allclose(P[key].detach().numpy(), fx["stat/" + key], rtol=2e-6, atol=1e-7, err_msg=key)
