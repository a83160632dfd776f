"""GPU point-to-point ICP (`forward(is_test=True)`) against the numpy restatement of open3d's algorithm
(oracle/icp_oracle.py; parity with open3d itself is unpinned) and against the ground-truth motion."""
import numpy as np
import pytest
import torch

from ogmm_amd import metric, ops, synth
from oracle import icp_oracle
from test_icp_oracle import _perturbed

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("engine", ["chip", "multi"])
@pytest.mark.parametrize("kind,n,radius", [("clean", 400, 0.07), ("partial", 717, 0.07), ("partial", 1024, 0.075), ("room", 600, 0.1)])
def test_matches_oracle(kind, n, radius, engine):
    B = 4
    src, tgt, R, t = synth.make_batch(40, B, n, kind)
    T0 = np.stack([_perturbed(R[i].double().numpy(), t[i].double().numpy(), 0.04, 0.01, i) for i in range(B)])
    R0, t0 = torch.from_numpy(T0[:, :3, :3]).float(), torch.from_numpy(T0[:, :3, 3]).float()
    xs, xt = src.transpose(1, 2).contiguous().to(DEV), tgt.transpose(1, 2).contiguous().to(DEV)
    Rg, tg, fit, rmse, iters = ops.icp_point_to_point(xs, xt, R0.to(DEV), t0.to(DEV), radius, want_stats=True, engine=engine)   # one workgroup per pair / grid-wide sequence
    for i in range(B):
        Ti = np.eye(4)
        Ti[:3, :3], Ti[:3, 3] = R0[i].double().numpy(), t0[i].double().numpy()        # the float32 initial motion both sides start from
        To, fo, ro, io = icp_oracle.icp_point_to_point(src[i].T.numpy(), tgt[i].T.numpy(), Ti, radius)
        assert int(iters[i]) == io, (int(iters[i]), io)
        assert abs(float(fit[i]) - fo) < 1e-6 and abs(float(rmse[i]) - ro) < 1e-6
        assert np.abs(Rg[i].cpu().double().numpy() - To[:3, :3]).max() < 2e-6
        assert np.abs(tg[i].cpu().double().numpy() - To[:3, 3]).max() < 2e-6


@pytest.mark.parametrize("engine", ["chip", "multi"])
def test_recovers_ground_truth_and_handles_no_overlap(engine):
    B, n = 3, 512
    src, tgt, R, t = synth.make_batch(7, B, n, "clean")
    T0 = np.stack([_perturbed(R[i].double().numpy(), t[i].double().numpy(), 0.06, 0.02, 10 + i) for i in range(B)])
    R0, t0 = torch.from_numpy(T0[:, :3, :3]).float().to(DEV), torch.from_numpy(T0[:, :3, 3]).float().to(DEV)
    t0[2] += 40.0                                           # pair 2: nothing within the radius -> the initial motion comes back
    xs, xt = src.transpose(1, 2).contiguous().to(DEV), tgt.transpose(1, 2).contiguous().to(DEV)
    Rg, tg, fit, rmse, iters = ops.icp_point_to_point(xs, xt, R0, t0, 0.07, want_stats=True, engine=engine)
    assert metric.rotation_error_rad(Rg[:2].cpu(), R[:2]).max() < 1e-5 and metric.translation_error(tg[:2].cpu(), t[:2]).max() < 1e-5
    assert float(fit[0]) == 1.0 and float(fit[2]) == 0.0 and int(iters[2]) == 1
    assert torch.equal(Rg[2], R0[2]) and torch.equal(tg[2], t0[2])


def test_forward_is_test_refines_the_network_estimate(golden):
    from argparse import Namespace
    from ogmm_amd.gmmreg import GMMReg
    fx = golden("clean_b1_n1024_j16")
    B, N, J, k, M, D, H = (int(v) for v in fx["meta"])
    cfg = Namespace(gnn_k=k, num_heads=H, km_clusters=M, overlap_radius=0.035)
    model = GMMReg(D, J, cfg)
    synth.fill_state_dict(model.state_dict())
    model = model.to(DEV).eval()
    src, tgt = torch.from_numpy(fx["src"]).to(DEV), torch.from_numpy(fx["tgt"]).to(DEV)
    starts = torch.from_numpy(fx["fps_starts"])
    with torch.no_grad():
        R0, t0, so0, to0, l0 = model(src, tgt, fps_starts=starts)
        R1, t1, so1, to1, l1 = model(src, tgt, is_test=True, fps_starts=starts)
    assert torch.equal(so0, so1) and torch.equal(to0, to1) and torch.equal(l0, l1)
    # what the call must equal: ICP started from the network's own estimate with radius 2 * overlap_radius
    Ti = np.eye(4)
    Ti[:3, :3], Ti[:3, 3] = R0[0].cpu().double().numpy(), t0[0].cpu().double().numpy()
    To, _, _, _ = icp_oracle.icp_point_to_point(fx["src"][0].T, fx["tgt"][0].T, Ti, 0.07)
    assert np.abs(R1[0].cpu().double().numpy() - To[:3, :3]).max() < 2e-6 and np.abs(t1[0].cpu().double().numpy() - To[:3, 3]).max() < 2e-6
