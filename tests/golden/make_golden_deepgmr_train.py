"""Generates tests/golden/deepgmr_train_b2_n512_j16.npz: ONE training step of the reference's DeepGMR baseline
(baseline/deepgmr.py in `.train()`, loss of train_base.py:52-56: `dcp_loss` of the model's two outputs, NaN -> 0) run on
CPU with the closed-form weights of ogmm_amd/synth.py.  `gmm_register` hard-codes `.cuda()` (baseline/deepgmr.py:30-31);
`Tensor.cuda` is patched to the identity for the run.

Stored in the format of make_golden_train.py so that tests/train_util.check_grads applies: the loss, the rotation, every
parameter's gradient norm plus a strided sample, the BatchNorm running statistics after the step, and -- as the yard-stick
for the gradients -- the same step of the same reference code evaluated in fp64 (default dtype switched for that run)
together with the reference's own fp32 distance from it.

    python tests/golden/make_golden_deepgmr_train.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_harness import default_config, import_reference  # noqa: E402
from ogmm_amd import synth                                        # noqa: E402

SAMPLE = 97
C6_SCALE = 60.0          # sharper cluster logits, as in make_golden_deepgmr.py (otherwise the 3x3 matrix is set by its + 1e-4)


def sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, min(numel, SAMPLE)).astype(np.int64))


def one_step(ref, ref_loss, ref_se3, cfg, J, src, tgt, T_gt, dtype):
    """-> (loss, R, second output, gradients, state after the step) of the reference in `dtype`"""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        net = ref.DeepGMR(512, J, cfg).train()
        synth.fill_state_dict(net.state_dict())
        with torch.no_grad():
            net.state_dict()["cluster.net.6.weight"].mul_(C6_SCALE)
        B = src.shape[0]
        rot, trans = net(src.to(dtype), tgt.to(dtype))
        rot_gt, trans_gt = ref_se3.decompose_trans(T_gt.to(dtype))
        loss = torch.nan_to_num(ref_loss.dcp_loss(rot, rot_gt, trans, trans_gt.view(B, 3)), nan=0.0)
        loss.backward()
    finally:
        torch.Tensor.cuda = real_cuda
        torch.set_default_dtype(old)
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    return loss.detach(), rot.detach(), trans.detach(), grads, {k: v.clone() for k, v in net.state_dict().items()}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import_reference()
    import baseline.deepgmr as ref
    import lib.loss as ref_loss
    import lib.se3 as ref_se3
    B, N, J = 2, 512, 16
    cfg = default_config(n_clusters=J)
    src, tgt, T_gt, _, _ = synth.make_train_batch(810, B, N, "partial")
    loss, rot, second, grads, P1 = one_step(ref, ref_loss, ref_se3, cfg, J, src, tgt, T_gt, torch.float32)
    loss64, rot64, _, grads64, _ = one_step(ref, ref_loss, ref_se3, cfg, J, src, tgt, T_gt, torch.float64)
    print("loss fp32 %.8f  fp64 %.10f   R difference %.2e" % (loss.item(), loss64.item(), (rot.double() - rot64).abs().max().item()))
    assert float(second.abs().max()) == 0.0          # tsfm[:, 3, 0:3]: the bottom row
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None)))
    fx = dict(src=src.numpy(), tgt=tgt.numpy(), T_gt=T_gt.numpy(), meta=np.array([B, N, J, cfg.gnn_k, 512]), c6_scale=np.float32(C6_SCALE),
              loss=loss.numpy(), R=rot.numpy(), loss64=np.float64(loss64.item()), gnorm_total=np.float64(total))
    errs = []
    for key, g in grads.items():
        if g is None:
            fx["gnorm/" + key] = np.float32(-1.0)
            continue
        flat, flat64 = g.reshape(-1).numpy(), grads64[key].reshape(-1).numpy()
        ids = sample_idx(flat.size)
        fx["gnorm/" + key] = np.float32(np.linalg.norm(flat.astype(np.float64)))
        fx["gsamp/" + key] = flat[ids]
        fx["gnorm64/" + key] = np.float64(np.linalg.norm(flat64))
        fx["gsamp64/" + key] = flat64[ids]
        tn = np.linalg.norm(flat64)
        scale = max(np.linalg.norm(flat64[ids]), tn * np.sqrt(len(ids) / flat.size))
        err = max(np.linalg.norm(flat[ids] - flat64[ids]) / max(scale, 1e-300), abs(np.linalg.norm(flat.astype(np.float64)) - tn) / max(tn, 1e-300))
        fx["gerr/" + key] = np.float64(err)
        if np.linalg.norm(flat) >= 1e-6 * total:
            errs.append((err, key))
    for key, v in P1.items():
        if "running" in key or "num_batches" in key:
            fx["stat/" + key] = v.numpy()
    errs.sort()
    print("global gradient norm %.4e; reference fp32 vs fp64 gradient distance: median %.2e, max %.2e (%s)" % (
        total, errs[len(errs) // 2][0], errs[-1][0], errs[-1][1]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deepgmr_train_b2_n512_j16.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
