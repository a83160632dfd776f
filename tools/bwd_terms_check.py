"""What rounding the B operand of the BACKWARD GEMMs to binary16 (two matrix instructions per product instead of three: W^T in dX = dY W, X^T in dW = dY^T X;
train_ops.BWD_TERMS_DX / BWD_TERMS_DW = 2; BWD_WHICH=dx|dw|both picks which) does to the gradients at a size on which the LDS-DMA engines actually run (the reference-generated training fixtures are too small for
them): per parameter the relative distance between the two-term and the three-term gradients, next to the distance between the three-term gradients and the
exact-fp32 engine's -- the engine's own fp32-class noise on the same step -- and, where a fixture of the family exists, the reference's own fp32-vs-fp64
distance as the scale the parity tests use.
usage: python3 tools/bwd_terms_check.py [pairs] [profile]"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ogmm_amd import losses, synth, train_ops  # noqa: E402
from ogmm_amd.gmmreg import GMMReg  # noqa: E402


def grads_of(precision, terms, batch, starts, profile, B):
    which = os.environ.get("BWD_WHICH", "both")
    train_ops.BWD_TERMS_DX = terms if which in ("both", "dx") else 0
    train_ops.BWD_TERMS_DW = terms if which in ("both", "dw") else 0
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, precision=precision)
    model = GMMReg(512, 16, cfg)
    synth.fill_state_dict(model.state_dict(), profile=profile)
    model = model.cuda().train()
    src, tgt, T_gt, so, to = batch
    out = model(src, tgt, fps_starts=starts)
    loss, _ = losses.training_loss(out, src, tgt, T_gt, so, to, 10.0, 512)
    scale = 65536.0 if precision == "f16x3" else 1.0
    (loss * scale).backward()
    assert not model.fp16_overflowed()
    return {k: (p.grad.double() / scale) for k, p in model.named_parameters() if p.grad is not None}, float(loss)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    profile = sys.argv[2] if len(sys.argv) > 2 else "default"
    batch = [t.cuda() for t in synth.make_train_batch(6000, B, 1024, "partial")]
    starts = synth.fps_starts_for(6000, B, 1024)
    g3, l3 = grads_of("f16x3", 0, batch, starts, profile, B)
    g3b, _ = grads_of("f16x3", 0, batch, starts, profile, B)          # the same step again: atomics / run-to-run noise
    g2, l2 = grads_of("f16x3", 2, batch, starts, profile, B)
    gf, lf = grads_of("f32", 0, batch, starts, profile, B)
    total = float(torch.sqrt(sum((g ** 2).sum() for g in g3.values())))
    rows = []
    for k in g3:
        n = float(g3[k].norm())
        if n < 1e-6 * total:
            continue
        rel = lambda a, b: float((a[k] - b[k]).norm()) / n          # noqa: E731
        rows.append((k, rel(g2, g3), rel(g3b, g3), rel(gf, g3)))
    d2, dn, df = (np.array([r[i] for r in rows]) for i in (1, 2, 3))
    print("# %d pairs of 1024 points, weight family %s: loss %.8f (3 terms) %.8f (2-term backward) %.8f (f32 engine)" % (B, profile, l3, l2, lf))
    print("# relative distance per parameter (%d live parameters), median / p90 / max" % len(rows))
    print("two-term vs three-term backward      %.2e  %.2e  %.2e" % (np.median(d2), np.percentile(d2, 90), d2.max()))
    print("three-term, the same step twice      %.2e  %.2e  %.2e" % (np.median(dn), np.percentile(dn, 90), dn.max()))
    print("exact-fp32 engine vs three-term      %.2e  %.2e  %.2e" % (np.median(df), np.percentile(df, 90), df.max()))
    print("ratio two-term distance / f32-engine distance: median %.3f  max %.3f" % (np.median(d2 / np.maximum(df, 1e-30)), (d2 / np.maximum(df, 1e-30)).max()))
    for r in sorted(rows, key=lambda r: -r[1])[:8]:
        print("    %-30s  2t-3t %.2e   3t-3t %.2e   f32-3t %.2e" % r)


if __name__ == "__main__":
    main()
