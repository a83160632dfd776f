"""DeepGMR training step: per-parameter gradient distance from the fixture's fp64 evaluation next to the reference's own fp32 distance (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
from test_hip_deepgmr import _train_model, TRAIN_GOLD
from ogmm_amd import losses
from train_util import check_grads
fx = np.load(TRAIN_GOLD)
B = int(fx["meta"][0])
for precision, scale in (("f32", 1.0), ("f16x3", 2.0 ** 12), ("f16x3", 2.0 ** 16), ("f16x3", 2.0 ** 20), ("f16x3", 2.0 ** 24)):
    model = _train_model(fx, precision)
    src, tgt, T_gt = (torch.from_numpy(fx[k]).cuda() for k in ("src", "tgt", "T_gt"))
    R, second = model(src, tgt)
    loss = torch.nan_to_num(losses.dcp_loss(R, T_gt[:, :3, :3], second, T_gt[:, :3, 3].reshape(B, 3)), nan=0.0)
    (loss * scale).backward()
    grads = {k: (p.grad / scale if p.grad is not None else None) for k, p in model.named_parameters()}
    rep = {}
    try:
        check_grads(fx, grads, factor=1000.0, report=rep)
    except AssertionError as e:
        print("assert", str(e)[:200])
    print(precision, "loss scale 2^%d" % int(np.log2(scale)), "loss", loss.item(), float(fx["loss"]), "overflow flag", int(model.overflow_flag("cuda:0").item()))
    for k, (err, allowed) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:5]:
        print("   %-28s err %.2e  ref_err %.2e" % (k, err, float(fx["gerr/" + k])))
