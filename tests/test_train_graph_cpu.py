"""The wiring of the training graph (ogmm_amd/train_graph.py + ogmm_amd/losses.py), run on CPU with the plain-PyTorch
operation set of tests/train_ref.py, against the fixtures of the reference's own training step."""
import numpy as np
import pytest
import torch

from ogmm_amd import losses, metric, train_graph
from train_ref import RefTrainOps, params_from_fixture_spec
from train_util import TRAIN_CASES, check_grads, load_train_case, noise_tol, profile_of


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_graph_matches_reference_training_step(name):
    torch.set_num_threads(8)
    fx, cfg, (B, N, J, D, top_k) = load_train_case(name)
    P = params_from_fixture_spec(D, profile=profile_of(fx))
    src, tgt = torch.from_numpy(fx["src"]), torch.from_numpy(fx["tgt"])
    out = train_graph.forward_train(RefTrainOps(), P, cfg, J, src, tgt, torch.from_numpy(fx["fps_starts"]))
    loss, parts = losses.training_loss(out, src, tgt, torch.from_numpy(fx["T_gt"]), torch.from_numpy(fx["src_overlap"]),
                                       torch.from_numpy(fx["tgt_overlap"]), 10.0, top_k)
    for kpart in ("dcp", "clu", "mse", "welsch"):
        tol = 1e-4 if kpart == "welsch" else 1e-5      # 2 - exp(-a) - exp(-b), a, b ~ 1e-6: fp32 cancellation (see test_hip_train.py)
        assert abs(parts[kpart].item() - float(fx["loss_" + kpart])) <= noise_tol(fx, "loss", tol * max(1.0, abs(float(fx["loss_" + kpart])))), kpart
    assert abs(loss.item() - float(fx["loss"])) <= noise_tol(fx, "loss", 1e-5 * abs(float(fx["loss"])))
    assert metric.rotation_error_rad(out[0].detach(), torch.from_numpy(fx["R"])).max() < noise_tol(fx, "R", 1e-5)
    assert metric.translation_error(out[1].detach(), torch.from_numpy(fx["t"])).max() < noise_tol(fx, "t", 1e-5)
    o_tol = noise_tol(fx, "o", 1e-5)
    np.testing.assert_allclose(out[2].detach().numpy(), fx["src_o"], atol=o_tol)
    np.testing.assert_allclose(out[3].detach().numpy(), fx["tgt_o"], atol=o_tol)
    loss.backward()
    grads = {k: v.grad for k, v in P.items() if v.is_floating_point() and "running" not in k}
    worst = check_grads(fx, grads)
    print("TRAIN-GRAPH-CPU %s loss=%.8f worst_grad_err_over_allowed=%.2f" % (name, loss.item(), worst))
    for key in (f[len("stat/"):] for f in fx.files if f.startswith("stat/")):
        np.testing.assert_allclose(P[key].detach().numpy(), fx["stat/" + key], rtol=1e-5, atol=1e-6, err_msg=key)


def test_deepgmr_graph_matches_reference_training_step():
    """The DeepGMR baseline's training graph (train_graph.deepgmr_forward_train) on the plain-PyTorch operation set against one training step of the
    reference's own baseline/deepgmr.py + train_base.py loss (tests/golden/make_golden_deepgmr_train.py)."""
    import os
    from train_ref import deepgmr_params_from_fixture_spec
    from train_util import GOLDEN
    from argparse import Namespace
    torch.set_num_threads(8)
    fx = np.load(os.path.join(GOLDEN, "deepgmr_train_b2_n512_j16.npz"))
    B, N, J, k, D = (int(v) for v in fx["meta"])
    P = deepgmr_params_from_fixture_spec(D, J, float(fx["c6_scale"]))
    src, tgt, T_gt = torch.from_numpy(fx["src"]), torch.from_numpy(fx["tgt"]), torch.from_numpy(fx["T_gt"])
    R, _ = train_graph.deepgmr_forward_train(RefTrainOps(), P, Namespace(gnn_k=k), J, src, tgt)
    loss = torch.nan_to_num(losses.dcp_loss(R, T_gt[:, :3, :3], torch.zeros(B, 3), T_gt[:, :3, 3]), nan=0.0)
    assert abs(loss.item() - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    assert metric.rotation_error_rad(R.detach(), torch.from_numpy(fx["R"])).max() < 3e-5          # (fp32 vs fp64 of the reference itself: 1.3e-5 on this case)
    loss.backward()
    grads = {k_: v.grad for k_, v in P.items() if v.is_floating_point() and "running" not in k_}
    worst = check_grads(fx, grads)
    print("DEEPGMR-TRAIN-GRAPH-CPU loss=%.8f worst_grad_err_over_allowed=%.2f" % (loss.item(), worst))
    for key in (f[len("stat/"):] for f in fx.files if f.startswith("stat/")):
        np.testing.assert_allclose(P[key].detach().numpy(), fx["stat/" + key], rtol=1e-5, atol=1e-6, err_msg=key)
