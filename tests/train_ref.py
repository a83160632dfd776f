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

    def _norm_linear_fusable(self, y, group_rows, W):
        return False                                  # the composed statement: normalise, then the layer

    def fanout(self, x, n):
        return (x,) * n

    def linear(self, x, W, b, x2=None):
        if x2 is not None:
            x = torch.cat([x, x2], dim=1)
        y = x @ W.t()
        return y if b is None else y + b

    def linear_stats(self, x, W, b, x2=None, groups=1):
        return self.linear(x, W, b, x2), None

    def batchnorm_act(self, y, weight, bias, running_mean, running_var, num_batches, groups, act, stats=None):
        n = y.shape[0] // groups
        outs = []
        for g in range(groups):                       # one F.batch_norm call per call of the reference's shared layer, src first
            blk = y[g * n:(g + 1) * n].t()[None]      # [1, channels, rows]
            outs.append(F.batch_norm(blk, running_mean, running_var, weight, bias, True, 0.1, 1e-5)[0].t())
        num_batches += groups
        h = torch.cat(outs, dim=0)
        return F.relu(h) if act == "relu" else F.leaky_relu(h, 0.2)

    def batchnorm_act_pool(self, y, weight, bias, running_mean, running_var, num_batches, groups, act, k, want_h, stats=None):
        h = self.batchnorm_act(y, weight, bias, running_mean, running_var, num_batches, groups, act)
        return (h if want_h else None), self.maxpool_k(h, k)

    def instnorm_relu(self, z, C, N, stats=None):
        zc = z.view(C, N, -1).transpose(1, 2)
        return F.relu(F.instance_norm(zc, eps=1e-5)).transpose(1, 2).reshape(C * N, -1)

    def maxpool_k(self, h, k):
        return h.view(-1, k, h.shape[1]).max(dim=1)[0]

    def nearest_point(self, xyz, mu):
        return torch.cdist(mu, xyz).argmin(dim=2)

    def edge_features(self, xyz, idx):
        C, N, k = idx.shape
        nb = torch.gather(xyz, 1, idx.reshape(C, N * k, 1).expand(-1, -1, 3)).view(C, N, k, 3)
        ctr = xyz[:, :, None, :].expand(-1, -1, k, -1)
        return torch.cat([nb - ctr, ctr], dim=3).reshape(C * N * k, 6)

    def pos_features(self, xyz, idx5):
        C, N, k = idx5.shape
        g = xyz - xyz.mean(dim=1, keepdim=True)
        d2 = (g * g).sum(dim=2).reshape(C * N, 1)
        nb = torch.gather(xyz, 1, idx5.reshape(C, N * k, 1).expand(-1, -1, 3)).view(C, N, k, 3)
        loc = F.normalize(nb - xyz[:, :, None, :], dim=3)
        alpha = (loc * F.normalize(g, dim=2)[:, :, None, :]).sum(dim=3)
        return d2, alpha.reshape(C * N * k, 1)

    def l2norm_rows(self, f):
        return F.normalize(f, dim=1)

    def attention(self, q, k, v, C, N, M, H):
        D = q.shape[1]
        dh = D // H
        qh = q.view(C, N, H, dh).transpose(1, 2)
        kh = k.view(C, M, H, dh).transpose(1, 2)
        vh = v.view(C, M, H, dh).transpose(1, 2)
        p = torch.softmax(qh @ kh.transpose(2, 3) / dh ** .5, dim=-1)
        return (p @ vh).transpose(1, 2).reshape(C * N, D)

    def overlap_cross(self, fn, ol, B, N):
        fs, ft = fn[:B * N].view(B, N, -1), fn[B * N:].view(B, N, -1)
        S = fs @ ft.transpose(1, 2)
        os_, ot = ol[:B * N].view(B, N), ol[B * N:].view(B, N)
        wo_s = (torch.softmax(S, dim=2) * os_[:, None, :]).sum(dim=2)
        wo_t = (torch.softmax(S, dim=1) * ot[:, :, None]).sum(dim=1)
        return torch.cat([wo_s.reshape(B * N, 1), wo_t.reshape(B * N, 1)], dim=0)

    def gmm_feat_mean(self, gamma, pi, f, C, N):
        return gamma.transpose(1, 2) @ f.view(C, N, -1) / (pi * N + 1e-5)[:, :, None]

    def kabsch(self, src, corr, w):
        ws = w.sum(dim=1, keepdim=True)
        c_s = (src * w[:, :, None]).sum(dim=1) / ws
        c_c = (corr * w[:, :, None]).sum(dim=1) / ws
        cov = ((src - c_s[:, None, :]) * w[:, :, None]).transpose(1, 2) @ (corr - c_c[:, None, :])
        cov = torch.nan_to_num(cov, nan=0.0) + 1e-5 * torch.eye(3, dtype=cov.dtype, device=cov.device)
        U, _, Vh = torch.linalg.svd(cov)
        V = Vh.transpose(1, 2)
        flip = torch.det(V @ U.transpose(1, 2)) <= 0
        V = torch.where(flip[:, None, None] & (torch.arange(3, device=V.device) == 2)[None, None, :], -V, V)
        R = V @ U.transpose(1, 2)
        t = c_c - (R @ c_s[:, :, None])[:, :, 0]
        return R, t

    def rotation_from_cov(self, M):
        U, _, Vh = torch.linalg.svd(M)                                  # baseline/deepgmr.py:28-34
        V = Vh.transpose(1, 2)
        S = torch.eye(3, dtype=M.dtype, device=M.device).repeat(M.shape[0], 1, 1)
        S[:, 2, 2] = torch.det(V @ U.transpose(1, 2))
        return V @ S @ U.transpose(1, 2)

    def gmm_em(self, xyz, o, ids_j):
        C, N, _ = xyz.shape
        B = C // 2          # src clouds | tgt clouds: two wkeans_plus calls (the Sinkhorn early exit averages over one call's clouds)
        outs = []
        for h in (slice(0, B), slice(B, C)):
            gamma, pi, mu, _, ids = O.weighted_em(xyz[h], xyz.new_zeros(B, N, 1), o[h], ids_j.shape[1], iters=10, tau=1.0)
            assert torch.equal(ids, ids_j[h])
            outs.append((gamma, pi, mu))
        return tuple(torch.cat(t) for t in zip(*outs))


def params_from_fixture_spec(D, dtype=torch.float32, profile="default"):
    """closed-form weights (ogmm_amd/synth.py) as a name -> tensor dict with requires_grad on the parameters"""
    from ogmm_amd import gmmreg, synth
    P = {k: torch.zeros(s) if "num_batches" not in k else torch.zeros((), dtype=torch.long) for k, s in gmmreg.state_spec(D)}
    synth.fill_state_dict(P, profile=profile)
    P = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return P


def deepgmr_params_from_fixture_spec(D, J, c6_scale, dtype=torch.float32):
    """the DeepGMR baseline's parameters with the closed-form fill and the sharpened cluster logits of its fixtures"""
    from ogmm_amd import deepgmr, synth
    P = {k: torch.zeros(s) if "num_batches" not in k else torch.zeros((), dtype=torch.long) for k, s in deepgmr.state_spec(D, J)}
    synth.fill_state_dict(P)
    P["cluster.net.6.weight"].mul_(c6_scale)
    P = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return P
