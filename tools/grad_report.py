"""Per-parameter gradient report of one training fixture on the GPU box: the HIP training step's gradients (both engines) against the fixture's fp64 truth,
next to the reference's own fp32-vs-fp64 distance per parameter (tests/train_util.py check_grads, with the outlier allowance lifted so that every
parameter is listed).  Used to pick / reject seeds for training fixtures (tests/golden/make_golden_train.py).   usage: grad_report.py <fixture name>"""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from train_util import check_grads, load_train_case, profile_of
from ogmm_amd import losses, synth
from ogmm_amd.gmmreg import GMMReg
name = sys.argv[1]
fx, cfg, (B, N, J, D, top_k) = load_train_case(name)
live = sorted(float(fx[f]) for f in fx.files if f.startswith("gerr/") and float(fx["gnorm/" + f[5:]]) >= 1e-6 * float(fx["gnorm_total"]))
print("reference's own fp32-vs-fp64 gradient distance: median %.2e  p90 %.2e  max %.2e  (%d params)" % (np.median(live), live[int(0.9 * len(live))], live[-1], len(live)))
for precision in ("f16x3", "f32"):
    cfg.precision = precision
    model = GMMReg(D, J, cfg); synth.fill_state_dict(model.state_dict(), profile=profile_of(fx)); model = model.cuda().train()
    src, tgt = torch.from_numpy(fx["src"]).cuda(), torch.from_numpy(fx["tgt"]).cuda()
    out = model(src, tgt, fps_starts=torch.from_numpy(fx["fps_starts"]))
    loss, parts = losses.training_loss(out, src, tgt, torch.from_numpy(fx["T_gt"]).cuda(), torch.from_numpy(fx["src_overlap"]).cuda(), torch.from_numpy(fx["tgt_overlap"]).cuda(), 10.0, top_k)
    scale = 65536.0 if precision == "f16x3" else 1.0
    (loss * scale).backward()
    grads = {k: (p.grad / scale if p.grad is not None else None) for k, p in model.named_parameters()}
    rep = {}
    try:
        check_grads(fx, grads, report=rep, max_outlier_frac=1.0)
    except AssertionError as e:
        print("assert:", str(e)[:200])
    ratios = sorted(((e / a, k, e, a) for k, (e, a) in rep.items()), reverse=True)
    errs = sorted(e for e, a in rep.values())
    print(precision, "candidate distance from fp64 truth: median %.2e p90 %.2e max %.2e; beyond bound: %d of %d" % (np.median(errs), errs[int(0.9 * len(errs))], errs[-1], sum(1 for r in ratios if r[0] > 1), len(ratios)))
    for r in ratios[:12]: print("    %-28s err %.2e allowed %.2e ref %.2e" % (r[1], r[2], r[3], float(fx["gerr/" + r[1]])))
