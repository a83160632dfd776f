"""Training-mode forward of GMMReg as a differentiable graph over `TrainOps` (ogmm_amd/train_ops.py).

Follows models/gmmreg.py:50-119 with the module in `.train()`: every BatchNorm normalises with the statistics of its
own call -- the reference calls each shared block once for src and once for tgt, so statistics are per HALF of the
stacked batch -- and updates its running statistics (momentum 0.1, unbiased variance) src call first
(models/dgcnn.py:126-130, :21-27; models/attn.py:34-57).  InstanceNorm is per cloud as in eval.  Gradients flow through
everything except the discrete selections (kNN, FPS, nearest point) and the E/M loop (lib/utils.py:275-288).

Layout as in the eval path: clouds stacked C = 2B (src clouds, then tgt clouds), feature maps point-major
[C*N, channels]; per-edge maps [C*N*k, channels].  `ops` supplies every tensor operation; this file only wires them.
"""
import torch

from . import losses

GROUPS = 2          # BatchNorm statistics groups: the src call and the tgt call of the reference


def _w(P, key):
    w = P[key + ".weight"]
    return w.reshape(w.shape[0], w.shape[1])


def _b(P, key):
    return P.get(key + ".bias")


def _bn(ops, P, name, y, act, stats=None):
    return ops.batchnorm_act(y, P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"], P[name + ".running_var"],
                             P[name + ".num_batches_tracked"], GROUPS, act, stats=stats)


def _conv_bn(ops, P, conv, bn, x, act, x2=None):
    """1x1 conv -> train-mode BatchNorm -> activation; the conv also delivers the column sums the BatchNorm needs"""
    y, st = ops.linear_stats(x, _w(P, conv), _b(P, conv), x2=x2, groups=GROUPS)
    return _bn(ops, P, bn, y, act, st)


def _conv_bn_pool(ops, P, conv, bn, x, act, k, want_h):
    """1x1 conv -> train-mode BatchNorm -> activation -> max over the k rows of every point, normalisation and pooling in one pass;
    -> (per-row map or None when nothing reads it, pooled map)"""
    y, st = ops.linear_stats(x, _w(P, conv), _b(P, conv), groups=GROUPS)
    return ops.batchnorm_act_pool(y, P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"], P[bn + ".running_var"],
                                  P[bn + ".num_batches_tracked"], GROUPS, act, k, want_h, stats=st)


def dgcnn(ops, P, xyz, idx):
    """models/dgcnn.py:133-154.  -> [C*N, D]"""
    C, N, k = idx.shape
    h = ops.edge_features(xyz, idx)                                   # [C*N*k, 6], constant
    pooled = []
    for l in (1, 2, 3, 4):
        # max over the k edges of a point, after the ReLU; the last layer's per-edge map is only pooled
        h, pl = _conv_bn_pool(ops, P, "emd.conv%d" % l, "emd.bn%d" % l, h, "relu", k, want_h=l < 4)
        pooled.append(pl)
    xcat = torch.cat(pooled, dim=1)
    return _conv_bn(ops, P, "emd.conv5", "emd.bn5", xcat, "relu")


def pos_encoding(ops, P, xyz, idx5):
    """models/attn.py:59-75 (`pos.conv` is never applied).  -> [C*N, D]"""
    d2, alpha = ops.pos_features(xyz, idx5)                              # [C*N, 1], [C*N*5, 1]: constants of the input
    h = _conv_bn(ops, P, "pos.conv_dis.0", "pos.conv_dis.1", d2, "leaky")
    dis = _conv_bn(ops, P, "pos.conv_dis.3", "pos.conv_dis.4", h, "leaky")
    _, a = _conv_bn_pool(ops, P, "pos.conv_ang1.0", "pos.conv_ang1.1", alpha, "leaky", idx5.shape[2], want_h=False)
    ang = _conv_bn(ops, P, "pos.conv_ang2.0", "pos.conv_ang2.1", a, "leaky")
    return torch.cat([dis, ang], dim=1)


def transformer(ops, P, name, x, anchors, C, N, M, H, res=None):
    """models/attn.py:78-111 without the residual.  x [C*N, D] or a pair of handles of it (query projection, MLP input: ops.fanout), anchors [C*M, D].  The reference splits heads as channel
    c = d*H + h (models/attn.py:96); the projections are re-ordered head-major (c' = h*dh + d) by permuting weight rows,
    and the merge convolution's input columns the same way, so that a head is a contiguous slice."""
    xq, x = x if isinstance(x, tuple) else (x, x)
    D = x.shape[1]
    dh = D // H
    # head-major re-ordering c' = h dh + d <- c = d H + h as a transposed VIEW of the parameter (one strided copy forwards, one backwards).  Until round 6 this
    # was advanced indexing with a permutation vector, whose backward is index_put_(accumulate=True): a radix sort per weight and step (25 of them in the profile).
    def rows_hm(w):          # [D, ...] -> rows in head-major order
        return w.view(dh, H, *w.shape[1:]).transpose(0, 1).reshape(w.shape)

    def cols_hm(w):          # [Dout, D] -> input columns in head-major order
        return w.view(w.shape[0], dh, H).transpose(1, 2).reshape(w.shape)
    q = ops.linear(xq, rows_hm(_w(P, name + ".attn.proj.0")), rows_hm(_b(P, name + ".attn.proj.0")))
    kk = ops.linear(anchors, rows_hm(_w(P, name + ".attn.proj.1")), rows_hm(_b(P, name + ".attn.proj.1")))
    vv = ops.linear(anchors, rows_hm(_w(P, name + ".attn.proj.2")), rows_hm(_b(P, name + ".attn.proj.2")))
    o = ops.attention(q, kk, vv, C, N, M, H)
    msg = ops.linear(o, cols_hm(_w(P, name + ".attn.merge")), _b(P, name + ".attn.merge"))
    z, st = ops.linear_stats(x, _w(P, name + ".mlp.0"), _b(P, name + ".mlp.0"), x2=msg, groups=C)
    return ops.instnorm_relu_linear(z, C, N, st, _w(P, name + ".mlp.3"), _b(P, name + ".mlp.3"), res=res)          # res: the caller's "+ x"


def conv_stack(ops, P, name, x, three, x2=None):
    """models/dgcnn.py:16-38"""
    def bn(key):
        return [P[name + key + s_] for s_ in (".weight", ".bias", ".running_mean", ".running_var", ".num_batches_tracked")]

    y, st = ops.linear_stats(x, _w(P, name + ".net.0"), _b(P, name + ".net.0"), x2=x2, groups=GROUPS)
    if not three:
        return ops.batchnorm_relu_linear(y, st, *bn(".net.1"), GROUPS, _w(P, name + ".net.3"), _b(P, name + ".net.3"))
    y, st = ops.batchnorm_relu_linear(y, st, *bn(".net.1"), GROUPS, _w(P, name + ".net.3"), _b(P, name + ".net.3"), want_stats=True)
    return ops.batchnorm_relu_linear(y, st, *bn(".net.4"), GROUPS, _w(P, name + ".net.6"), _b(P, name + ".net.6"))


def forward_train(ops, P, cfg, n_clusters, src, tgt, fps_starts, cap=None):
    """P: name -> tensor (parameters with requires_grad, buffers updated in place).  src, tgt [B,3,N];
    fps_starts int [6,B] in the reference's draw order.  Returns (R [B,3,3], t [B,3], src_o [B,N], tgt_o [B,N], clu_loss [])."""
    B, _, N = src.shape
    C, k, H, M, J = 2 * B, cfg.gnn_k, cfg.num_heads, cfg.km_clusters, n_clusters
    D = P["emd.conv5.weight"].shape[0]
    xyz = torch.cat([src, tgt], dim=0).transpose(1, 2).contiguous()              # [C,N,3]
    starts = fps_starts.reshape(3, 2 * B)                                         # [stage][src clouds | tgt clouds]
    idx, idx5 = ops.knn(xyz, k), ops.knn(xyz, 5)
    ids_a = ops.fps(xyz, M, starts)                                               # [3,C,M]
    ids_j = ops.fps(xyz, J, None)                                                 # [C,J]
    swap = torch.cat([torch.arange(B, C, device=xyz.device), torch.arange(0, B, device=xyz.device)])          # (made on the device: no host copy inside a recorded step)

    # a map with several consumers goes through ops.fanout: one handle per consumer, their gradients are added in one pass
    emb = dgcnn(ops, P, xyz, idx)
    emb_a, emb_x = ops.fanout(emb, 2)
    a0 = ops.gather_points(emb_a, C, N, ids_a[0])                                 # gmmreg.py:54-57
    x0 = emb_x + pos_encoding(ops, P, xyz, idx5)                                  # gmmreg.py:58-61
    x0_q, x0_m, x0_r = ops.fanout(x0, 3)
    ft = conv_stack(ops, P, "conv1", transformer(ops, P, "sattn1", (x0_q, x0_m), a0, C, N, M, H, res=x0_r), True)
    ft_a, ft_q, ft_m, ft_r = ops.fanout(ft, 4)
    a1 = ops.gather_points(ft_a, C, N, ids_a[1], cloud_map=swap)                  # the OTHER cloud's anchors (gmmreg.py:67-72)
    f = transformer(ops, P, "cattn", (ft_q, ft_m), a1, C, N, M, H, res=ft_r)
    f_n, f_p, f_c, f_a, f_q, f_m, f_r = ops.fanout(f, 7)

    fn = ops.l2norm_rows(f_n)                                                     # gmmreg.py:74
    ol = conv_stack(ops, P, "proj", f_p, False)                                   # [C*N, 1] overlap logits
    wo = ops.overlap_cross(fn, ol, B, N)                                          # [C*N, 1]  (gmmreg.py:75-80)
    fo = conv_stack(ops, P, "conv2", f_c, True, x2=torch.cat([wo, ol], dim=1))
    o = torch.sigmoid(conv_stack(ops, P, "overlap", fo, True)).view(C, N)         # gmmreg.py:85-89

    a2 = ops.gather_points(f_a, C, N, ids_a[2])
    f2 = transformer(ops, P, "sattn2", (f_q, f_m), a2, C, N, M, H, res=f_r)       # gmmreg.py:92-97

    gamma, pi, mu = ops.gmm_em(xyz, o.detach(), ids_j)                            # no gradient (lib/utils.py:275-288)
    f2_m, f2_g = ops.fanout(f2, 2)                                                # (round 5: the nearest-point gather below hands its 16 gradient rows per cloud to this sum)
    muf = ops.gmm_feat_mean(gamma, pi, f2_m, C, N)                                # [C,J,D], gradient to f2 only
    R, t = ops.match_kabsch(mu[:B], mu[B:], muf[:B], muf[B:], 0.05)               # gmmreg.py:102-103
    near = ops.nearest_point(xyz, mu)                                             # [C,J] (lib/utils.py:244-254)
    anchors_f = ops.gather_points(f2_g, C, N, near).view(C, J, D)
    clu = 0.5 * (losses.info_nce(anchors_f[:B], muf[:B], 0.1) + losses.info_nce(anchors_f[B:], muf[B:], 0.1))
    if cap is not None:
        cap.update(knn_idx=idx, fps_anchor=ids_a, fps_J=ids_j, emb=emb, x0=x0, ft=ft, f=f, f2=f2, o=o, gamma=gamma, pi=pi, mu=mu,
                   muf=muf, near=near)
    return R, t, o[:B], o[B:], clu


def deepgmr_forward_train(ops, P, cfg, n_clusters, src, tgt, cap=None):
    """The DeepGMR baseline in `.train()` (baseline/deepgmr.py:64-79; its loop is train_base.py:27-75): backbone and cluster head with batch-statistics
    BatchNorm per call (src call, then tgt call: two statistics groups, as in GMMReg), soft assignments, `gmm_params(..., return_sigma=True)`
    (lib/utils.py:130-148) and `gmm_register` (baseline/deepgmr.py:17-37) as differentiable tensor operations on the [C,N,J] / [C,J,*] maps.
    P: name -> tensor with the baseline's own keys (`backbone.*`, `cluster.*`).  Returns (R [B,3,3], t [B,3]) of `tsfm`; the module hands the caller
    what the reference does (ogmm_amd/deepgmr.py)."""
    B, _, N = src.shape
    C, k, J = 2 * B, cfg.gnn_k, n_clusters
    xyz = torch.cat([src, tgt], dim=0).transpose(1, 2).contiguous()              # [C,N,3]
    idx = ops.knn(xyz, k)
    Pb = {"emd." + key[len("backbone."):]: v for key, v in P.items() if key.startswith("backbone.")}
    feats = dgcnn(ops, Pb, xyz, idx)                                              # deepgmr.py:66-67
    logits = conv_stack(ops, P, "cluster", feats, True)                           # :69-70  [C*N, J]
    gamma = torch.softmax(logits, dim=1).view(C, N, J)                            # :71-72
    pi = gamma.mean(dim=1)                                                        # lib/utils.py:136-140
    npi = pi * N + 1e-5
    mu = torch.bmm(gamma.transpose(1, 2), xyz) / npi[:, :, None]                  # [C,J,3]
    d2 = ((xyz[:, :, None, :] - mu[:, None, :, :]) ** 2).sum(dim=-1)              # lib/utils.py:143-147: isotropic sigma
    sigma = (d2 * gamma).sum(dim=1) / npi                                         # [C,J]
    pi_s, mu_s, mu_t, sig_t = pi[:B], mu[:B], mu[B:], sigma[B:]
    c_s = torch.bmm(pi_s[:, None, :], mu_s)                                       # deepgmr.py:24-27: both centred with the SOURCE weights
    c_t = torch.bmm(pi_s[:, None, :], mu_t)
    Ms = torch.bmm((pi_s[:, :, None] * (mu_s - c_s)).transpose(1, 2), (mu_t - c_t) / sig_t[:, :, None])
    R = ops.rotation_from_cov(torch.nan_to_num(Ms, nan=0.0) + 1e-4)               # :28-34
    t = (c_t.transpose(1, 2) - torch.bmm(R, c_s.transpose(1, 2)))[:, :, 0]         # :35
    if cap is not None:
        cap.update(knn_idx=idx, feats=feats, gamma=gamma, pi=pi, mu=mu, sigma=sigma)
    return R, t
