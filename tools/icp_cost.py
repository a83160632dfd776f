"""Cost of the ICP refinement (is_test=True) on top of the forward, and what it does to the registration error, B=64 / N=1024 clean and partial pairs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth, metric, ops
from ogmm_amd.gmmreg import GMMReg
dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
m = GMMReg(512, 16, cfg); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
for kind in ("clean", "partial"):
    src, tgt, R, t = synth.make_batch(0, 64, 1024, kind)
    src, tgt = src.to(dev), tgt.to(dev)
    starts = synth.fps_starts_for(0, 64, 1024)
    res = {}
    with torch.no_grad():
        for flag in (False, True):
            for _ in range(3): out = m(src, tgt, is_test=flag, fps_starts=starts)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): out = m(src, tgt, is_test=flag, fps_starts=starts)
            torch.cuda.synchronize(); res[flag] = ((time.perf_counter() - t0) / 10 * 1e3, out)
        # ICP from a near-correct start (what a trained network would hand over): ground truth perturbed by 3 degrees / 0.02
        ang = torch.tensor(3.0 * 3.14159 / 180)
        K = torch.tensor([[0., -1, 0], [1, 0, 0], [0, 0, 0]])
        dR = torch.eye(3) + torch.sin(ang) * K + (1 - torch.cos(ang)) * K @ K
        R0, t0_ = (dR @ R).to(dev), (t + 0.02).to(dev)
        Ri, ti, fit, rmse, iters = ops.icp_point_to_point(src.transpose(1, 2).contiguous(), tgt.transpose(1, 2).contiguous(), R0, t0_, 0.07, want_stats=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(10): ops.icp_point_to_point(src.transpose(1, 2).contiguous(), tgt.transpose(1, 2).contiguous(), R0, t0_, 0.07)
        torch.cuda.synchronize(); icp_ms = (time.perf_counter() - t1) / 10 * 1e3
    print("%-8s forward %.2f ms, forward + ICP %.2f ms; ICP alone from a 3-degree start: %.2f ms, %.1f iterations on average, R error %.3f -> %.3f deg, fitness %.2f" % (
        kind, res[False][0], res[True][0], icp_ms, iters.float().mean(), metric.rotation_error(R0.cpu(), R).mean(), metric.rotation_error(Ri.cpu(), R).mean(), fit.mean()))
