"""CPU ORACLE for the OGMM registration hot path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the reported CPU baseline.  The product path
(`ogmm_amd/`) never imports it and has no CPU fallback.

What it is: a functional (no nn.Module) plain-PyTorch-CPU restatement of
`GMMReg.forward(src, tgt, is_test=False)` of gfmei/ogmm in eval mode, written against the
reference sources cited per function (paths relative to the reference root).  It runs in fp32
(the reference's arithmetic) or fp64 (pass double tensors: a "truth" to judge both fp32 paths).

Parity pinning: there are no tests / golden vectors in the reference, so the
oracle is pinned against outputs of the reference itself, run in the build container
(`tests/golden/make_golden.py` imports /root/reference and commits inputs + outputs as fixtures;
`tests/test_oracle_golden.py` checks this file against them).  `is_test=True` (open3d ICP) is out
of scope and unpinned.

Determinism notes that the HIP path relies on (probed against torch 2.10 CPU):
  * `matmul` with K=3 is bitwise a left-to-right fmaf chain  fma(a2,b2, fma(a1,b1, a0*b0));
    `sum(x**2, -1)` over 3 channels is ((x0^2 + x1^2) + x2^2) with individually rounded squares.
    Hence the kNN / FPS distance values, and so the discrete index choices, are reproducible
    bit-for-bit on the GPU.
  * torch's vectorised sqrt/exp/log are NOT correctly rounded, so continuous quantities agree
    to rounding only.
"""
import math

import torch
import torch.nn.functional as F

from . import split_emulation as _emu

BN_EPS = 1e-5  # nn.BatchNorm*/nn.InstanceNorm1d default eps (models/dgcnn.py:126-130, models/attn.py:24)


# --------------------------------------------------------------------------------------
# L1 functional ops (lib/utils.py, lib/se3.py)
# --------------------------------------------------------------------------------------
def sq_dist_expanded(a, b):
    """lib/utils.py:12-34 (`square_distance`, normalize=False): -2ab^T + |a|^2 + |b|^2, clamp 1e-12.
    a [B,N,3], b [B,M,3] -> [B,N,M].  The in-place add order (row norms first) is kept."""
    d = -2 * torch.matmul(a, b.transpose(1, 2))
    d = d + (a ** 2).sum(-1)[:, :, None]
    d = d + (b ** 2).sum(-1)[:, None, :]
    return d.clamp(min=1e-12)


def knn_indices(pts, k):
    """lib/utils.py:37-44 (`knn`): indices of the k smallest expanded distances, ascending.
    pts [B,N,3] -> [B,N,k] int64.  Self is rank 0 (its distance is clamped to 1e-12)."""
    return torch.topk(sq_dist_expanded(pts, pts), k, dim=-1, largest=False, sorted=True)[1]


def edge_features(x, idx):
    """lib/utils.py:47-66 (`get_graph_feature`) without its in-place mutation of `idx`.
    x [B,C,N], idx [B,N,k] -> [B,2C,N,k] = cat(x_j - x_i, x_i)."""
    B, C, N = x.shape
    k = idx.shape[-1]
    xt = x.transpose(1, 2)  # [B,N,C]
    nb = torch.gather(xt, 1, idx.reshape(B, N * k, 1).expand(-1, -1, C)).view(B, N, k, C)
    ctr = xt[:, :, None, :].expand(-1, -1, k, -1)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2)


def fps(xyz, npoint, start=None):
    """lib/utils.py:170-198 (`farthest_point_sample`).  xyz [B,N,3] -> ids [B,npoint] int64.
    start=None  -> the `is_center=True` branch (:183-188): the running-min array is seeded with
                   the squared distances to the centroid and the first pick is their argmax.
    start=[B]   -> the `is_center=False` branch with the `torch.randint` draw (:190) made an
                   explicit input.
    Distances are the direct form sum((p-c)^2) (:194); argmax = first maximal index."""
    B, N, _ = xyz.shape
    ids = torch.zeros(B, npoint, dtype=torch.long)
    running = torch.full((B, N), 1e10, dtype=xyz.dtype)
    ar = torch.arange(B)
    if start is None:
        c = xyz.mean(1).view(B, 1, 3)
        d = ((xyz - c) ** 2).sum(-1)
        running = torch.where(d < running, d, running)
        far = running.max(-1)[1]
    else:
        far = start.long()
    for i in range(npoint):
        ids[:, i] = far
        c = xyz[ar, far, :].view(B, 1, 3)
        d = ((xyz - c) ** 2).sum(-1)
        running = torch.where(d < running, d, running)
        far = running.max(-1)[1]
    return ids


def gather_rows(pts, ids):
    """lib/utils.py:111-127 (`index_points`): pts [B,N,C], ids [B,S] -> [B,S,C]."""
    return torch.gather(pts, 1, ids[:, :, None].expand(-1, -1, pts.shape[-1]))


def sinkhorn_log(cost, p, q=None, epsilon=1e-2, thresh=1e-2, max_iter=100, resid=None):
    """lib/utils.py:69-108 (`log_boltzmann_kernel` + `sinkhorn`), log-domain Sinkhorn.
    cost [B,N,J], p [B,N]; q=None -> uniform 1/J (the reference `.squeeze()`s it, which only
    changes shapes, not values).  Returns (gamma=exp(K) [B,N,J], iterations actually run).
    The early exit compares the BATCH-MEAN of sum|du|+sum|dv| with `thresh` (:99-102); `resid` (optional list) receives
    every sweep's per-cloud value of that sum [B]."""
    B, N, J = cost.shape
    if q is None:
        q = torch.full((B, J), 1.0 / J, dtype=torch.float).to(cost.dtype)
    u = torch.zeros_like(p)
    v = torch.zeros_like(q)
    logp = torch.log(p + 1e-8)
    logq = torch.log(q + 1e-8)
    iters = 0
    for _ in range(max_iter):
        iters += 1
        u0, v0 = u, v
        K = (-cost + u[:, :, None] + v[:, None, :]) / epsilon
        u = epsilon * (logp - torch.logsumexp(K, dim=-1)) + u
        Kt = ((-cost + u[:, :, None] + v[:, None, :]) / epsilon).transpose(1, 2)
        v = epsilon * (logq - torch.logsumexp(Kt, dim=-1)) + v
        diff = (u - u0).abs().sum(-1) + (v - v0).abs().sum(-1)
        if resid is not None:
            resid.append(diff.clone())
        if diff.mean().item() < thresh:
            break
    K = (-cost + u[:, :, None] + v[:, None, :]) / epsilon
    return _emu.ew(torch.exp(K), 'em.gamma'), iters


def gmm_moments(gamma, pts):
    """lib/utils.py:130-140 (`gmm_params`, return_sigma=False).
    gamma [B,N,J], pts [B,N,C] -> pi [B,J] = mean_n gamma, mu [B,J,C] = gamma^T pts / (pi*N + 1e-5)."""
    pi = gamma.mean(dim=1)
    npi = pi * gamma.shape[1] + 1e-5
    return pi, gamma.transpose(1, 2) @ pts / npi[:, :, None]


def weighted_em(xyz, feats, o_scores, n_clusters, iters=10, tau=1.0, stats=None, resid=None):
    """lib/utils.py:269-291 (`wkeans_plus`): overlap-weighted Sinkhorn k-means ("GMM E/M").
    xyz [B,N,3], feats [B,N,D], o_scores [B,N] -> gamma [B,N,J], pi [B,J], mu_xyz [B,J,3], mu_feat [B,J,D].
    `stats` (optional list) receives the Sinkhorn iteration count of every E-step, `resid` (optional list) every sweep's
    per-cloud residual sum|du| + sum|dv| (the early exit's quantity)."""
    ids = fps(xyz, n_clusters, None)
    mu = gather_rows(xyz, ids)
    o_scores = o_scores.detach()                              # lib/utils.py:275: no gradient reaches the overlap scores from here
    o = o_scores / o_scores.sum(-1, keepdim=True).clip(min=1e-4)
    gamma = None
    with torch.no_grad():                                     # lib/utils.py:278
        for _ in range(iters):
            cost = torch.cdist(xyz, mu).clip(min=0.0) / tau
            g, n_it = sinkhorn_log(cost, o, None, max_iter=10, resid=resid)
            if stats is not None:
                stats.append(n_it)
            g = torch.nan_to_num(g, nan=0.0)
            gamma = g / g.sum(-1, keepdim=True).clip(min=1e-3)
            pi, mu = gmm_moments(gamma, xyz)
    mu_feat = gmm_moments(gamma, feats)[1]                    # with grad w.r.t. feats (lib/utils.py:289-290)
    return gamma, pi, mu, mu_feat, ids


def kabsch(src, corr, w):
    """lib/se3.py:256-289 (`compute_rigid_transformation`).  src, corr [B,3,J], w [B,1,J]
    -> R [B,3,3], t [B,3,1].  cov + 1e-5 I, SVD, R = V U^T, reflection fixed by negating V[:,:,2]."""
    ws = w.sum(2, keepdim=True)
    c_s = (src * w).sum(2, keepdim=True) / ws
    c_c = (corr * w).sum(2, keepdim=True) / ws
    cov = torch.matmul((src - c_s) * w, (corr - c_c).transpose(1, 2))
    cov = torch.nan_to_num(cov, nan=0.0) + 1e-5 * torch.eye(3, dtype=src.dtype)[None]
    U, _, Vh = torch.linalg.svd(cov)
    V = Vh.transpose(1, 2)
    R_pos = V @ U.transpose(1, 2)
    Vn = V.clone()
    Vn[:, :, 2] *= -1
    R_neg = Vn @ U.transpose(1, 2)
    R = torch.where(torch.det(R_pos)[:, None, None] > 0, R_pos, R_neg)
    t = torch.matmul(-R, c_s) + c_c
    return R, t


def nearest_feats(xyz, mu, feats):
    """lib/utils.py:244-254 (`get_local_corrs`): feature of the point nearest (Euclidean cdist)
    to each mu.  xyz [B,N,3], mu [B,S,3], feats [B,N,D] -> ([B,S,D], idx [B,S])."""
    idx = torch.topk(torch.cdist(mu, xyz), k=1, dim=2, largest=False)[1]
    return torch.gather(feats, 1, idx.expand(-1, -1, feats.shape[-1])), idx[:, :, 0]


def info_nce(x, y, tau):
    """lib/loss.py:16-57 (`ConLoss.forward`, normalize=True).  x, y [B,n,D] -> scalar:
    cross-entropy (label 0) over B*2n rows of [pos | own-set negatives | cross-set negatives]."""
    B, n, _ = y.shape
    x = F.normalize(x, p=2, dim=-1)
    y = F.normalize(y, p=2, dim=-1)
    s_xy = torch.einsum('bmd,bnd->bmn', x, y) / tau
    s_yx = torch.einsum('bmd,bnd->bmn', y, x) / tau
    s_xx = torch.einsum('bmd,bnd->bmn', x, x) / tau
    s_yy = torch.einsum('bmd,bnd->bmn', y, y) / tau
    off = ~torch.eye(n, dtype=torch.bool)

    def offdiag(s):
        return s[:, off].reshape(B, n, n - 1)

    pos = torch.cat((torch.diagonal(s_xy, dim1=1, dim2=2), torch.diagonal(s_yx, dim1=1, dim2=2)), 1)[:, :, None]
    neg = torch.cat((torch.cat((offdiag(s_xx), offdiag(s_xy)), 2),
                     torch.cat((offdiag(s_yx), offdiag(s_yy)), 2)), 1)
    logits = torch.cat((pos, neg), 2).view(-1, 2 * n - 1)
    return F.cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.long))


# --------------------------------------------------------------------------------------
# L2 model blocks (models/dgcnn.py, models/attn.py, models/gmmreg.py), eval mode
# --------------------------------------------------------------------------------------
_BN_TRAINING = False   # set by forward(train=True): batch statistics + in-place running-stat updates (momentum 0.1)


def _bn(P, name, x):
    if _BN_TRAINING and (name + '.num_batches_tracked') in P:
        P[name + '.num_batches_tracked'] += 1                 # nn.BatchNorm bookkeeping (unused by the arithmetic: momentum is fixed)
    return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'],
                        P[name + '.weight'], P[name + '.bias'], _BN_TRAINING, 0.1, BN_EPS)


def _conv(P, name, x):
    w = P[name + '.weight']
    b = P.get(name + '.bias')
    mode = _emu.mode_of(name)                                  # oracle/split_emulation.py: operand rounding of the HIP engines, off by default
    if mode is not None:
        return _emu.conv(x, w, b, mode)
    return F.conv2d(x, w, b) if w.dim() == 4 else F.conv1d(x, w, b)


def _einsum(name, eq, a, b):
    mode = _emu.mode_of(name)
    return torch.einsum(eq, a, b) if mode is None else _emu.einsum(eq, a, b, mode)


def dgcnn_embed(P, x, k, idx=None, cap=None):
    """models/dgcnn.py:118-154 (`DGCNN.forward`): static-graph EdgeConv, x [B,3,N] -> [B,D,N]."""
    B, _, N = x.shape
    if idx is None:
        idx = knn_indices(x.transpose(1, 2), k)
    if cap is not None:
        cap['knn_idx'] = idx
    h = edge_features(x, idx)
    pooled = []
    for l in (1, 2, 3, 4):
        h = F.relu(_bn(P, 'emd.bn%d' % l, _conv(P, 'emd.conv%d' % l, h)))
        pooled.append(h.max(dim=-1, keepdim=True)[0])
    h = torch.cat(pooled, dim=1)
    return F.relu(_bn(P, 'emd.bn5', _conv(P, 'emd.conv5', h))).view(B, -1, N)


def pos_encoding(P, pts, k=5, idx=None):
    """models/attn.py:59-75 (`PositionEncoding.forward`): pts [B,3,N] -> [B,D,N]
    (distance-to-centroid channels | max-over-kNN angle channels).  `pos.conv` is never applied."""
    c = pts.mean(dim=-1, keepdim=True)
    g = pts - c
    d2 = (g * g).sum(dim=1, keepdim=True)
    h = F.leaky_relu(_bn(P, 'pos.conv_dis.1', _conv(P, 'pos.conv_dis.0', d2)), 0.2)
    dis = F.leaky_relu(_bn(P, 'pos.conv_dis.4', _conv(P, 'pos.conv_dis.3', h)), 0.2)
    if idx is None:
        idx = knn_indices(pts.transpose(1, 2), k)
    loc = edge_features(pts, idx)[:, :3]
    gn = F.normalize(g, dim=1)
    ln = F.normalize(loc, dim=1)
    alpha = torch.einsum('bdnk,bdn->bnk', ln, gn)[:, None]
    a = F.leaky_relu(_bn(P, 'pos.conv_ang1.1', _conv(P, 'pos.conv_ang1.0', alpha)), 0.2).max(dim=-1)[0]
    ang = F.leaky_relu(_bn(P, 'pos.conv_ang2.1', _conv(P, 'pos.conv_ang2.0', a)), 0.2)
    return torch.cat([dis, ang], dim=1)


def transformer(P, name, src, anchors, heads, cap=None):
    """models/attn.py:78-111 (`attention`, `MultiHeadAttention`, `MLP`, `Transformer`):
    src [B,D,N] attends to anchors [B,D,M]; returns mlp(cat[src, message]) [B,D,N] (no residual)."""
    B, D, _ = src.shape
    dh = D // heads
    q = _conv(P, name + '.attn.proj.0', src).view(B, dh, heads, -1)
    kk = _conv(P, name + '.attn.proj.1', anchors).view(B, dh, heads, -1)
    vv = _conv(P, name + '.attn.proj.2', anchors).view(B, dh, heads, -1)
    prob = _emu.ew(torch.softmax(_einsum(name + '.attn.qk', 'bdhn,bdhm->bhnm', q, kk) / dh ** .5, dim=-1), name + '.attn.softmax')
    if cap is not None:                                       # diagnostic only: how peaked the attention is (mean over queries of the largest probability; 1/M = uniform)
        cap.setdefault('attn_maxprob_list_' + name, []).append(prob.max(dim=-1)[0].mean().item())
        cap['attn_maxprob_' + name] = sum(cap['attn_maxprob_list_' + name]) / len(cap['attn_maxprob_list_' + name])
    msg = _einsum(name + '.attn.pv', 'bhnm,bdhm->bdhn', prob, vv).contiguous().view(B, D, -1)
    msg = _conv(P, name + '.attn.merge', msg)
    h = _conv(P, name + '.mlp.0', torch.cat([src, msg], dim=1))
    h = F.relu(F.instance_norm(h, eps=BN_EPS))
    return _conv(P, name + '.mlp.3', h)


def conv_stack(P, name, x, three):
    """models/dgcnn.py:16-38 (`CONV`): conv-BN-ReLU (x2 if `three`) then a last conv."""
    h = F.relu(_bn(P, name + '.net.1', _conv(P, name + '.net.0', x)))
    if three:
        h = F.relu(_bn(P, name + '.net.4', _conv(P, name + '.net.3', h)))
        return _conv(P, name + '.net.6', h)
    return _conv(P, name + '.net.3', h)


def match_and_solve(mu_s, mu_t, f_s, f_t):
    """models/dgcnn.py:96-115 (`GMMSVD.forward`, is_sk=False) + lib/utils.py:222-226.
    mu_* [B,J,3], f_* [B,J,D] -> R [B,3,3], t [B,3]."""
    sim = torch.einsum('bnd,bmd->bnm', F.normalize(f_s, dim=-1, p=2), F.normalize(f_t, dim=-1, p=2))
    sc = _emu.ew(torch.softmax(sim / 0.05, dim=2), 'match.softmax')
    corr = torch.einsum('bmd,bnm->bdn', mu_t, sc)
    w = sc.sum(dim=-1).unsqueeze(1)
    R, t = kabsch(mu_s.transpose(1, 2), corr, w)
    return R, t.view(-1, 3), sc


def draw_fps_starts(B, N):
    """The six `torch.randint(0, N, (B,))` draws of one forward (lib/utils.py:190), in the call
    order src,tgt,src,tgt,src,tgt of models/gmmreg.py:54,56,67,69,92,94 -> int64 [6,B]."""
    return torch.stack([torch.randint(0, N, (B,), dtype=torch.long) for _ in range(6)])


def forward(P, cfg, src, tgt, fps_starts=None, cap=None, inject=None, train=False):
    """models/gmmreg.py:50-119 (`GMMReg.forward`, is_test=False); eval mode, or with train=True the `.train()`
    behaviour: every BatchNorm normalises with the statistics of its own call (src and tgt are separate calls) and
    updates P's running statistics in place, src call first.  Autograd runs through everything except kNN / FPS / the
    E-M loop, as in the reference.

    P: the reference state_dict (153 keys); cfg: gnn_k, num_heads, km_clusters (+ n_clusters);
    src, tgt [B,3,N].  fps_starts [6,B] pins the random FPS starts (drawn like the reference
    when None).  `cap` (dict) receives intermediates; `inject` may carry 'knn_idx_src/tgt' to
    pin the kNN graph, and -- for stage-by-stage bisection of a second implementation (tools/tail_bisect.py) -- any of
    'emb_<s>', 'ft_<s>', 'f_<s>', 'f2_<s>' [B,D,N], 'o_<s>' [B,N], 'em_<s>' = (gamma [B,N,J], pi [B,J], mu [B,J,3]) and
    'muf_<s>' [B,J,D] (s = src / tgt): the stage's own result is replaced by the injected one and everything downstream
    is evaluated in this file's arithmetic.  Returns (R [B,3,3], t [B,3], src_o [B,N], tgt_o [B,N], loss [])."""
    global _BN_TRAINING
    _BN_TRAINING = bool(train)
    try:
        return _forward(P, cfg, src, tgt, fps_starts, cap, inject)
    finally:
        _BN_TRAINING = False


def _forward(P, cfg, src, tgt, fps_starts, cap, inject):
    B, _, N = src.shape
    k, H, M, J = cfg.gnn_k, cfg.num_heads, cfg.km_clusters, cfg.n_clusters
    if fps_starts is None:
        fps_starts = draw_fps_starts(B, N)
    cap = {} if cap is None else cap
    inject = inject or {}
    pts = {'src': src, 'tgt': tgt}
    xyz = {s: pts[s].transpose(1, 2).contiguous() for s in pts}
    other = {'src': 'tgt', 'tgt': 'src'}
    draw = {('src', 0): 0, ('tgt', 0): 1, ('src', 1): 2, ('tgt', 1): 3, ('src', 2): 4, ('tgt', 2): 5}

    def anchors(s, feats, stage):
        ids = fps(xyz[s], M, fps_starts[draw[(s, stage)]])
        cap['fps%d_%s' % (stage, s)] = ids
        return gather_rows(feats.transpose(1, 2), ids).transpose(1, 2)

    emb, a0, ft, a1, f, o_logit = {}, {}, {}, {}, {}, {}
    for s in pts:                                           # gmmreg.py:52-57
        c = {}
        emb[s] = dgcnn_embed(P, pts[s], k, inject.get('knn_idx_' + s), c)
        emb[s] = inject.get('emb_' + s, emb[s])
        cap['knn_idx_' + s] = c['knn_idx']
        cap['emb_' + s] = emb[s]
    for s in pts:
        a0[s] = anchors(s, emb[s], 0)
    for s in pts:                                           # gmmreg.py:58-63
        pos = pos_encoding(P, pts[s], 5, cap['knn_idx_' + s][:, :, :5] if k >= 5 and 'knn_idx_' + s in inject else None)
        cap['pos_' + s] = pos
        x = emb[s] + pos
        ft[s] = inject['ft_' + s] if 'ft_' + s in inject else conv_stack(P, 'conv1', transformer(P, 'sattn1', x, a0[s], H, cap) + x, True)
        cap['ft_' + s] = ft[s]
    for s in pts:                                           # gmmreg.py:67-70
        a1[s] = anchors(s, ft[s], 1)
    for s in pts:                                           # gmmreg.py:71-72
        f[s] = inject['f_' + s] if 'f_' + s in inject else transformer(P, 'cattn', ft[s], a1[other[s]], H, cap) + ft[s]
        cap['f_' + s] = f[s]
    fn = {s: F.normalize(f[s]) for s in pts}                # gmmreg.py:74-80
    sim = _einsum('similarity', 'bdm,bdn->bmn', fn['src'], fn['tgt'])
    for s in pts:
        o_logit[s] = conv_stack(P, 'proj', f[s], False)
    wo = {'src': torch.einsum('bmn,bdn->bdm', _emu.ew(torch.softmax(sim, dim=-1), 'overlap.softmax.rows'), o_logit['src']),
          'tgt': torch.einsum('bmn,bdm->bdn', _emu.ew(torch.softmax(sim, dim=1), 'overlap.softmax.cols'), o_logit['tgt'])}
    o = {}
    for s in pts:                                           # gmmreg.py:82-89
        fo = conv_stack(P, 'conv2', torch.cat([f[s], wo[s], o_logit[s]], dim=1), True)
        o[s] = inject['o_' + s] if 'o_' + s in inject else torch.sigmoid(conv_stack(P, 'overlap', fo, True)).view(B, -1)
        cap['wo_' + s] = wo[s]
        cap['o_' + s] = o[s]
    a2, f2 = {}, {}
    for s in pts:                                           # gmmreg.py:92-95
        a2[s] = anchors(s, f[s], 2)
    for s in pts:                                           # gmmreg.py:96-97
        f2[s] = inject['f2_' + s] if 'f2_' + s in inject else transformer(P, 'sattn2', f[s], a2[s], H, cap) + f[s]
        cap['f2_' + s] = f2[s]
    clu = {}
    for s in pts:                                           # gmmreg.py:100-101, :24-29
        st, rs = [], []
        clu[s] = weighted_em(xyz[s], f2[s].transpose(1, 2), o[s], J, iters=10, tau=1.0, stats=st, resid=rs)
        if 'em_' + s in inject:                                 # (gamma, pi, mu) of another implementation; the feature means follow from ITS gamma, in this arithmetic
            g_, pi_, mu_ = inject['em_' + s]
            clu[s] = (g_, pi_, mu_, gmm_moments(g_, f2[s].transpose(1, 2))[1], clu[s][4])
        if 'muf_' + s in inject:
            clu[s] = clu[s][:3] + (inject['muf_' + s], clu[s][4])
        cap['gamma_' + s], cap['pi_' + s], cap['mu_' + s], cap['muf_' + s], cap['fpsJ_' + s] = clu[s]
        cap['sk_iters_' + s] = st                               # sweeps every E-step ran (lib/utils.py:99-102: the batch-mean early exit)
        cap['sk_resid_' + s] = rs                               # every sweep's per-cloud residual [B], in execution order
    R, t, sc = match_and_solve(clu['src'][2], clu['tgt'][2], clu['src'][3], clu['tgt'][3])   # gmmreg.py:102-103
    cap['match_scores'] = sc
    loss = 0
    for s in pts:                                           # gmmreg.py:106-110, lib/loss.py:114-118
        gamma, _, mu, _, _ = clu[s]
        fT = f2[s].transpose(1, 2)
        positives = gmm_moments(gamma, fT)[1]
        anchors_f, near = nearest_feats(xyz[s], mu, fT)
        cap['near_' + s] = near
        loss = loss + info_nce(anchors_f, positives, 0.1)
    loss = 0.5 * loss
    return R, t, o['src'], o['tgt'], loss


# --------------------------------------------------------------------------------------
# training losses (lib/loss.py, train.py:54-74) -- SURVEY section 8 a22
# --------------------------------------------------------------------------------------
def dcp_loss(R, R_gt, t, t_gt):
    """lib/loss.py:121-126: mse(R^T R_gt, I) + mse(t, t_gt)."""
    B = t_gt.shape[0]
    eye = torch.eye(3, dtype=R.dtype)[None].repeat(B, 1, 1)
    return F.mse_loss(torch.matmul(R.transpose(2, 1), R_gt), eye) + F.mse_loss(t.view(B, 3), t_gt.view(B, 3))


def welsch_loss(src, tgt, T, src_o, tgt_o, alpha=10.0, top_k=512):
    """lib/loss.py:83-106 (`WelschLoss`): src, tgt [B,N,3]; T [B,4,4]; the top_k points by (ground-truth) overlap of
    each cloud are matched to their nearest neighbour in the other cloud after moving src by T."""
    moved = torch.matmul(src, T[:, :3, :3].transpose(-1, -2)) + T[:, :3, 3][:, None, :]      # lib/se3.py:105-109
    s_ids = torch.topk(src_o, k=top_k, dim=-1)[1][:, :, None].expand(-1, -1, 3)
    t_ids = torch.topk(tgt_o, k=top_k, dim=-1)[1][:, :, None].expand(-1, -1, 3)
    z1 = torch.cdist(torch.gather(moved, 1, s_ids), tgt).min(dim=-1)[0]
    z2 = torch.cdist(torch.gather(tgt, 1, t_ids), moved).min(dim=-1)[0]
    a2 = alpha * alpha
    return (2.0 - torch.exp(-0.5 * z1 ** 2.0 / a2) - torch.exp(-0.5 * z2 ** 2.0 / a2)).sum(dim=1).mean()


def training_loss(out, src, tgt, T_gt, src_overlap, tgt_overlap, alpha=10.0, top_k=512):
    """train.py:54-72: 10 dcp + clu + mse(overlap) + 0.01 welsch, nan -> 0.  `out` = forward(...)'s 5-tuple;
    src, tgt [B,3,N]; T_gt [B,4,4]; *_overlap [B,N] ground-truth labels."""
    R, t, so, to, clu = out
    B = T_gt.shape[0]
    R_gt, t_gt = T_gt[:, :3, :3], T_gt[:, :3, 3:4].reshape(B, 3)
    o_pred = torch.nan_to_num(torch.cat([so, to], dim=-1), nan=0.0).clip(min=0.0)
    o_gt = torch.nan_to_num(torch.cat([src_overlap, tgt_overlap], dim=-1), nan=0.0).clip(min=0.0)
    T_pred = torch.eye(4, dtype=R.dtype)[None].repeat(B, 1, 1)          # lib/se3.py:29-52 (in-place writes into a leaf-free eye)
    T_pred[:, :3, :3] = R
    T_pred[:, :3, 3:4] = t.view(-1, 3, 1)
    loss = 10 * dcp_loss(R, R_gt, t, t_gt) + clu + F.mse_loss(o_pred, o_gt) \
        + 0.01 * welsch_loss(src.transpose(1, 2), tgt.transpose(1, 2), T_pred, src_overlap, tgt_overlap, alpha, top_k)
    return torch.nan_to_num(loss, nan=0.0)


# --------------------------------------------------------------------------------------
# parity metrics (lib/metric.py:85-93 restated in fp64; SURVEY.md section 8d caveat: the reference's
# fp32 acos cannot resolve below ~3.5e-4 rad, so the geodesic angle is computed via asin)
# --------------------------------------------------------------------------------------
def rotation_error_rad(R1, R2):
    d = (R1.double() - R2.double()).flatten(1).norm(dim=1)
    return 2.0 * torch.asin((d / (2.0 * math.sqrt(2.0))).clamp(max=1.0))


def translation_error(t1, t2):
    return (t1.double() - t2.double()).norm(dim=1)
