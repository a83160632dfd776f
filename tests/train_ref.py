"""Plain-PyTorch (CPU-capable) statement of the training-graph operations: the numerical reference the tests hold the
HIP-backed `ogmm_amd.train_ops.TrainOps` against, and the backend that lets the graph wiring of
ogmm_amd/train_graph.py be checked against the reference's training-step fixtures without a GPU.

Selections come from the CPU oracle (test infrastructure).  Every method that the product implements with HIP kernels
is overridden here with its torch-op statement.
"""
import torch
import torch.nn.functional as F

from ogmm_amd.train_ops import TrainOps
from oracle import ogmm_oracle as O


class RefTrainOps(TrainOps):
    def knn(self, xyz, k):
        return O.knn_indices(xyz, k)

    def fps(self, xyz, npoint, starts):
        if starts is None:
            return O.fps(xyz, npoint, None)
        return torch.stack([O.fps(xyz, npoint, s) for s in starts])

    def linear(self, x, W, b, x2=None):
        if x2 is not None:
            x = torch.cat([x, x2], dim=1)
        y = x @ W.t()
        return y if b is None else y + b

    def batchnorm_act(self, y, weight, bias, running_mean, running_var, num_batches, groups, act):
        n = y.shape[0] // groups
        outs = []
        for g in range(groups):                       # one F.batch_norm call per call of the reference's shared layer, src first
            blk = y[g * n:(g + 1) * n].t()[None]      # [1, channels, rows]
            outs.append(F.batch_norm(blk, running_mean, running_var, weight, bias, True, 0.1, 1e-5)[0].t())
        num_batches += groups
        h = torch.cat(outs, dim=0)
        return F.relu(h) if act == "relu" else F.leaky_relu(h, 0.2)

    def instnorm_relu(self, z, C, N):
        zc = z.view(C, N, -1).transpose(1, 2)
        return F.relu(F.instance_norm(zc, eps=1e-5)).transpose(1, 2).reshape(C * N, -1)

    def maxpool_k(self, h, k):
        return h.view(-1, k, h.shape[1]).max(dim=1)[0]

    def gmm_em(self, xyz, o, ids_j):
        C, N, _ = xyz.shape
        gamma, pi, mu, _, ids = O.weighted_em(xyz, xyz.new_zeros(C, N, 1), o, ids_j.shape[1], iters=10, tau=1.0)
        assert torch.equal(ids, ids_j)
        return gamma, pi, mu


def params_from_fixture_spec(D, dtype=torch.float32):
    """closed-form weights (ogmm_amd/synth.py) as a name -> tensor dict with requires_grad on the parameters"""
    from ogmm_amd import gmmreg, synth
    P = {k: torch.zeros(s) if "num_batches" not in k else torch.zeros((), dtype=torch.long) for k, s in gmmreg.state_spec(D)}
    synth.fill_state_dict(P)
    P = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return P
