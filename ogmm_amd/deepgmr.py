"""Drop-in `DeepGMR` baseline (gfmei/ogmm baseline/deepgmr.py:64-79) on the kernels of the main path.

Same constructor `DeepGMR(emb_dims, n_clusters, config)` (config: gnn_k, overlap_radius), same call
`model(src, tgt, is_test=False)` with float32 [B,3,N] inputs and the same state_dict keys (`backbone.conv1..5`,
`backbone.bn1..5`, `cluster.net.{0,1,3,4,6}`); `.eval()` runs the fused inference kernels, `.train()` the training graph
(train_graph.deepgmr_forward_train; the loop of train_base.py:27-75 is trainer.BaselineTrainer).  Shares with GMMReg: kNN + fused EdgeConv + the dense GEMM engine
(backbone and cluster head), the softmax kernel, the GMM moment kernel, the in-register 3x3 solve and the ICP refinement.

Behaviour kept as is: without `is_test` the second return value is `tsfm[:, 3, 0:3]`, the bottom row of
the 4x4 motion, i.e. zeros -- not the translation (baseline/deepgmr.py:79); `gmm_register` centres BOTH mean sets with the
source weights and adds 1e-4 to every entry of the 3x3 matrix before the SVD (baseline/deepgmr.py:24-29).
"""
import torch
from torch import nn

from . import ops
from ._lib import OgmmError
from .gmmreg import BN_EPS, _add_splits, _build_tree, _default_init, _fold_bn, _w2d  # noqa: F401
from .ops import ACT_RELU


def state_spec(D=512, J=16):
    spec = []
    for i, (co, ci) in enumerate(((64, 6), (64, 64), (128, 64), (256, 128), (D, 512)), 1):      # models/dgcnn.py:121-125
        spec.append(("backbone.conv%d.weight" % i, (co, ci, 1, 1)))
    for i, c in enumerate((64, 64, 128, 256, D), 1):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            spec.append(("backbone.bn%d.%s" % (i, leaf), (c,)))
        spec.append(("backbone.bn%d.num_batches_tracked" % i, ()))
    hid = D // 2
    for idx, (co, ci) in (("0", (hid, D)), ("3", (hid, hid)), ("6", (J, hid))):                  # models/dgcnn.py:19-28
        spec.append(("cluster.net.%s.weight" % idx, (co, ci, 1)))
        spec.append(("cluster.net.%s.bias" % idx, (co,)))
        if idx != "6":
            bn = str(int(idx) + 1)
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                spec.append(("cluster.net.%s.%s" % (bn, leaf), (hid,)))
            spec.append(("cluster.net.%s.num_batches_tracked" % bn, ()))
    return spec


def pack_weights(sd):
    L = {}
    for i in range(1, 6):
        s, t = _fold_bn(sd, "backbone.bn%d" % i)
        L["emd%d" % i] = {"W": _w2d(sd, "backbone.conv%d" % i), "scale": s, "shift": t}
    for idx, bn in (("0", "1"), ("3", "4")):
        s, t = _fold_bn(sd, "cluster.net." + bn, sd["cluster.net.%s.bias" % idx])
        L["c" + idx] = {"W": _w2d(sd, "cluster.net." + idx), "scale": s, "shift": t}
    L["c6"] = {"W": _w2d(sd, "cluster.net.6"), "shift": sd["cluster.net.6.bias"].float().contiguous()}
    _add_splits(L)
    return L


class DeepGMR(nn.Module):
    def __init__(self, emb_dims, n_clusters, config):
        super().__init__()
        self.emb_dims, self.n_clusters, self.config = emb_dims, n_clusters, config
        spec = state_spec(emb_dims, n_clusters)
        _build_tree(self, spec)
        _default_init(self, spec)
        self.precision = getattr(config, "precision", "f16x3")
        self._packed = self._packed_key = None
        self._overflow = None

    def invalidate_packed(self):
        """Drop the packed weights (needed after in-place edits through `.data`, which do not bump the tensors' _version)."""
        self._packed = self._packed_key = None

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def _layers(self):
        sd = self.state_dict()
        key = tuple((t.data_ptr(), t._version) for t in sd.values())
        if self._packed is None or key != self._packed_key:
            self._packed, self._packed_key = pack_weights(sd), key
        return self._packed

    def forward(self, src, tgt, is_test=False):
        if not (isinstance(src, torch.Tensor) and src.is_cuda and tgt.is_cuda):
            raise OgmmError("DeepGMR.forward needs CUDA/ROCm tensors: the MI355X path has no CPU fallback")
        if src.dim() != 3 or src.shape[1] != 3 or src.shape != tgt.shape or src.dtype != torch.float32:
            raise OgmmError("src and tgt must be float32 [B,3,N] of equal shape")
        B, _, N = src.shape
        C, D, J, k = 2 * B, self.emb_dims, self.n_clusters, self.config.gnn_k
        dev = src.device
        if self._overflow is None or self._overflow.device != dev:
            self._overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.training:
            return self._forward_train(src, tgt, is_test)
        L = self._layers()
        eng = ops.Engine(self.precision, self._overflow)
        xyz = torch.cat([src, tgt], dim=0).transpose(1, 2).contiguous()                 # [C,N,3]
        idx = ops.knn(xyz, k)
        R_ = C * N
        xcat = torch.empty((R_, 512), dtype=torch.float32, device=dev)
        emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
        if eng.split and ops.edgeconv_fused_supported(k, emd):
            ops.edgeconv_fused(xyz, idx, emd, xcat)
        else:
            h = ops.edgeconv_first(xyz, idx, L["emd1"], xcat[:, 0:64])
            h = ops.edgeconv_layer(h, L["emd2"], k, xcat[:, 64:128], eng=eng)
            h = ops.edgeconv_layer(h, L["emd3"], k, xcat[:, 128:256], eng=eng)
            ops.edgeconv_layer(h, L["emd4"], k, xcat[:, 256:512], store=False, eng=eng)
        feats = ops.conv1x1(xcat, L["emd5"], ACT_RELU, eng=eng)                          # baseline/deepgmr.py:66-67
        h = ops.conv1x1(feats, L["c0"], ACT_RELU, eng=eng)                               # :69-70 (`CONV`, used='proj')
        h = ops.conv1x1(h, L["c3"], ACT_RELU, eng=eng)
        logits = ops.conv1x1(h, L["c6"], split=False, eng=eng) if J < 32 else ops.conv1x1(h, L["c6"], eng=eng)
        gamma = ops.softmax_rows_(logits).view(C, N, J)                                  # :71-72, softmax over the J clusters
        # gmm_params(..., return_sigma=True)  (lib/utils.py:130-148): pi, mu, isotropic sigma
        pi = gamma.mean(dim=1)
        mu = ops.gmm_feat_mean(gamma, pi, xyz.view(R_, 3), C, N)                         # [C,J,3]
        npi = pi * N + 1e-5
        d2 = ((xyz[:, :, None, :] - mu[:, None, :, :]) ** 2).sum(dim=-1)                 # [C,N,J]
        sigma = (d2 * gamma).sum(dim=1) / npi                                            # [C,J]
        # gmm_register(pi_s, mu_s, mu_t, sigma_t)  (baseline/deepgmr.py:17-37)
        pi_s, mu_s, mu_t, sig_t = pi[:B], mu[:B], mu[B:], sigma[B:]
        c_s = torch.bmm(pi_s[:, None, :], mu_s)                                          # [B,1,3]
        c_t = torch.bmm(pi_s[:, None, :], mu_t)
        Ms = torch.bmm((pi_s[:, :, None] * (mu_s - c_s)).transpose(1, 2), (mu_t - c_t) / sig_t[:, :, None])
        Rm = ops.rotation_from_cov(torch.nan_to_num(Ms, nan=0.0) + 1e-4)
        t = c_t.transpose(1, 2) - torch.bmm(Rm, c_s.transpose(1, 2))                     # [B,3,1]
        if is_test:                                                                      # :76-78
            return ops.icp_point_to_point(xyz[:B], xyz[B:], Rm, t[:, :, 0].contiguous(), 2.0 * self.config.overlap_radius)
        return Rm, torch.zeros((B, 3), dtype=torch.float32, device=dev)                  # tsfm[:, 3, 0:3]: the bottom row (sic)

    def overflow_flag(self, device):
        """the engine's device-side overflow word (int32[1]; ogmm_amd/trainer.py snapshots it around forward and backward)"""
        if self._overflow is None or self._overflow.device != torch.device(device):
            self._overflow = torch.zeros(1, dtype=torch.int32, device=device)
        return self._overflow

    def _forward_train(self, src, tgt, is_test=False):
        """`.train()`: batch-statistics BatchNorm with running-statistics updates and autograd through everything but the kNN
        (ogmm_amd/train_graph.deepgmr_forward_train over the kernels of ogmm_amd/train_ops.py); train_base.py:27-75 is the loop around it."""
        from . import train_graph, train_ops
        if self.precision not in ("f16x3", "f32"):
            raise OgmmError("training runs with precision 'f16x3' or 'f32' (the reduced 'f16' mode is inference only)")
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        R, t = train_graph.deepgmr_forward_train(train_ops.TrainOps(self.precision, self.overflow_flag(src.device)), P, self.config,
                                                 self.n_clusters, src, tgt)
        if is_test:                                                                      # baseline/deepgmr.py:76-78 (no gradient through open3d there either)
            return ops.icp_point_to_point(src.transpose(1, 2).contiguous(), tgt.transpose(1, 2).contiguous(), R.detach(), t.detach().contiguous(),
                                          2.0 * self.config.overlap_radius)
        # tsfm[:, 3, 0:3]: the bottom row of the 4x4 (sic) -- the baseline's loss sees the rotation only
        return R, torch.zeros((src.shape[0], 3), dtype=torch.float32, device=src.device)
