"""How far the DeepGMR training fixture's gradients move when the forward is perturbed at the level of the fp16x3 engine's rounding (CPU).

The reference-equivalent plain-PyTorch graph (tests/train_ref.RefTrainOps) is run with every linear layer's output multiplied by (1 + r U(-1, 1)),
r = 0, 2^-23, 2^-22, 2^-21, and every parameter's gradient compared with the fixture's fp64 evaluation.  Result on deepgmr_train_b2_n512_j16: from r = 2^-22 on
the SAME distances appear whatever the random draw -- backbone.bn4.bias 1.5e-2, bn3.bias 2.5e-3, bn2.bias 2.1e-3, conv3.weight 1.6e-3 -- i.e. one
max-pool / ReLU unit of the per-edge maps sits within 2^-22 of its kink and lands on the other side; everything upstream of it shifts by a fixed
amount.  Those are the distances the HIP path shows with precision "f16x3" (22-bit products); with "f32" it stays at the reference's own level."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from argparse import Namespace  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ogmm_amd import losses, train_graph  # noqa: E402
from train_ref import RefTrainOps, deepgmr_params_from_fixture_spec  # noqa: E402
from train_util import GOLDEN, check_grads  # noqa: E402


class Noisy(RefTrainOps):
    def __init__(self, rel):
        self.rel = rel

    def _noise(self, y):
        return y + y.detach().abs() * self.rel * (2 * torch.rand_like(y) - 1) if self.rel else y

    def linear(self, x, W, b, x2=None):
        return self._noise(super().linear(x, W, b, x2))

    def linear_stats(self, x, W, b, x2=None, groups=1):
        y, st = super().linear_stats(x, W, b, x2, groups)
        return self._noise(y), st


def main():
    torch.set_num_threads(8)
    fx = np.load(os.path.join(GOLDEN, "deepgmr_train_b2_n512_j16.npz"))
    B, N, J, k, D = (int(v) for v in fx["meta"])
    src, tgt, T_gt = torch.from_numpy(fx["src"]), torch.from_numpy(fx["tgt"]), torch.from_numpy(fx["T_gt"])
    keys = ["backbone.bn%d.bias" % i for i in (1, 2, 3, 4, 5)] + ["backbone.conv3.weight"]
    for rel in (0.0, 2.0 ** -23, 2.0 ** -22, 2.0 ** -21):
        for seed in (1, 2, 3):
            torch.manual_seed(seed)
            P = deepgmr_params_from_fixture_spec(D, J, float(fx["c6_scale"]))
            R, _ = train_graph.deepgmr_forward_train(Noisy(rel), P, Namespace(gnn_k=k), J, src, tgt)
            loss = torch.nan_to_num(losses.dcp_loss(R, T_gt[:, :3, :3], torch.zeros(B, 3), T_gt[:, :3, 3]), nan=0.0)
            loss.backward()
            rep = {}
            check_grads(fx, {n: v.grad for n, v in P.items() if v.is_floating_point() and "running" not in n}, factor=1e6, report=rep)
            print("r = %-9.3g seed %d  " % (rel, seed) + "  ".join("%s %.2e" % (n[9:], rep[n][0]) for n in keys))


if __name__ == "__main__":
    main()
