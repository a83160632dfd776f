"""Losses of the training step: the clustering InfoNCE that `GMMReg.forward` itself returns
(lib/loss.py:16-57, :109-118) and the three terms train.py:69-71 adds -- all on small tensors ([B,J,D], [B,3,3], [B,2N],
[B,top_k,N]), written with differentiable torch tensor ops on whatever device the inputs live on.
"""
import torch
import torch.nn.functional as F


def _bmm(a, b):
    """a [B,R,m] b [B,m,D]: on the device the small-product kernel T10 (ogmm_small_bmm_nn, with autograd: train_ops._SmallNN) -- torch.bmm, i.e. a hipBLASLt
    kernel, until round 6; on the CPU (the gloo / wiring tests) plain torch"""
    if a.is_cuda:
        from .train_ops import small_bmm
        return small_bmm(a, b)
    return torch.bmm(a, b)


def _bmm_nt(a, b):
    """a [B,n,D] b [B,m,D]^T, as _bmm"""
    if a.is_cuda:
        from .train_ops import small_bmm_nt
        return small_bmm_nt(a, b)
    return torch.bmm(a, b.transpose(1, 2))


def info_nce(anchor, positive, tau):
    """`ConLoss.forward` (lib/loss.py:22-57) for one cloud set.  anchor, positive [B,n,D] -> scalar.
    Each of the B*2n rows is a softmax classification whose class 0 is the matching pair (x_i, y_i) and whose other
    2n-2 classes are the same-set and cross-set similarities with the diagonal removed, all divided by tau."""
    B, n, _ = anchor.shape
    x = F.normalize(anchor, p=2, dim=-1)
    y = F.normalize(positive, p=2, dim=-1)
    both = torch.cat([x, y], dim=1)                                   # [B,2n,D]
    sim = _bmm_nt(both, both) / tau                                   # blocks [[xx, xy], [yx, yy]]
    # [B,2n]: xy_ii for the x rows, yx_ii for the y rows -- the two off-diagonals as views (advanced indexing here meant an index_put with a sort in the backward)
    pos = torch.cat([torch.diagonal(sim, offset=n, dim1=1, dim2=2), torch.diagonal(sim, offset=-n, dim1=1, dim2=2)], dim=1)
    # the negatives are every entry of a row except the two "diagonals" (self-similarity and the positive): row r drops columns r and (r + n) mod 2n.
    # As a gather with computed column indices (ascending, as a boolean mask would select them): no index_put, no mask -> nonzero, i.e. no host
    # synchronisation -- the loss can sit inside a recorded (HIP graph) training step.
    r = torch.arange(2 * n, device=anchor.device)[:, None]
    lo, hi = torch.minimum(r, (r + n) % (2 * n)), torch.maximum(r, (r + n) % (2 * n))
    c = torch.arange(2 * n - 2, device=anchor.device)[None, :].expand(2 * n, -1)
    c = c + (c >= lo).long()
    c = c + (c >= hi).long()
    neg = torch.gather(sim, 2, c[None].expand(B, -1, -1))
    logits = torch.cat([pos[:, :, None], neg], dim=2).reshape(B * 2 * n, 2 * n - 1)
    return F.cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.long, device=anchor.device))


def dcp_loss(R, R_gt, t, t_gt):
    """lib/loss.py:121-126"""
    B = R.shape[0]
    eye = torch.eye(3, dtype=R.dtype, device=R.device).expand(B, 3, 3)
    return F.mse_loss(_bmm(R.transpose(1, 2), R_gt), eye) + F.mse_loss(t.reshape(B, 3), t_gt.reshape(B, 3))


def overlap_mse(src_o, tgt_o, src_overlap, tgt_overlap):
    """train.py:59-62 + lib/loss.py:137-138 (`get_weighted_bce_loss` is a plain MSE)"""
    pred = torch.nan_to_num(torch.cat([src_o, tgt_o], dim=-1), nan=0.0).clip(min=0.0)
    gt = torch.nan_to_num(torch.cat([src_overlap, tgt_overlap], dim=-1), nan=0.0).clip(min=0.0)
    return F.mse_loss(pred, gt)


def welsch_loss(src, tgt, R, t, src_overlap, tgt_overlap, alpha=10.0, top_k=512):
    """`WelschLoss.forward` (lib/loss.py:83-106) with the predicted motion given as (R, t) instead of the 4x4 the
    reference packs first (lib/se3.py:29-52).  src, tgt [B,N,3].

    Tie semantics of the label top-k (lib/loss.py:92, :95).  The labels are 0 / 1, so with more than top_k ones (or fewer: then zeros are drawn too) WHICH points
    `torch.topk` keeps is decided by its kernel's selection moves, not by the values.  Parity is defined against the reference's CPU evaluation -- the only one that
    can be pinned here: the goldens of tests/golden/make_golden_train.py come from it -- and on the device `ops.topk_rows` replays ATen's CPU selection move for move.
    The reference's own GPU runs use torch's CUDA top-k, whose choice among ties is unspecified and differs from both; the loss is a mean over whichever tied points
    are drawn (5e-4 apart in the Welsch term at N = 1024, top_k = 512).  Rows beyond `ops.TOPK_ROWS_MAX_N` points fall back to the device library's top-k."""
    moved = _bmm(src, R.transpose(1, 2)) + t.reshape(-1, 1, 3)
    if moved.is_cuda:
        from . import ops
        if src_overlap.shape[-1] <= ops.TOPK_ROWS_MAX_N:
            s_ids, t_ids = ops.topk_rows(src_overlap.float().contiguous(), top_k), ops.topk_rows(tgt_overlap.float().contiguous(), top_k)
        else:          # (the LDS-resident candidate list of ogmm_topk_rows ends there: the device library's own tie choice from here on)
            s_ids = torch.topk(src_overlap, k=top_k, dim=-1)[1]
            t_ids = torch.topk(tgt_overlap, k=top_k, dim=-1)[1]
    else:
        s_ids = torch.topk(src_overlap, k=top_k, dim=-1)[1]
        t_ids = torch.topk(tgt_overlap, k=top_k, dim=-1)[1]
    take = lambda p, ids: torch.gather(p, 1, ids[:, :, None].expand(-1, -1, 3))          # noqa: E731
    if moved.is_cuda:
        # min over a [B, top_k, N] distance tensor (268 MB per term at 128 x 512 x 1024, plus its backward) = the distance to the NEAREST point: the
        # index comes from the nearest-point kernel (cdist's matmul form, first minimum), the distance and its gradient -- (p - q*) / |p - q*|, the
        # sub-gradient torch's min picks -- from the two gathered points
        from . import ops

        def nearest_dist(a, b):          # a [B,K,3] (grad), b [B,N,3] (grad) -> [B,K]
            q = take(b, ops.nearest_point(b.detach().contiguous(), a.detach().contiguous()).long())
            return torch.linalg.vector_norm(a - q, dim=-1)
        z1 = nearest_dist(take(moved, s_ids), tgt)
        z2 = nearest_dist(take(tgt, t_ids), moved)
    else:
        z1 = torch.cdist(take(moved, s_ids), tgt).min(dim=-1)[0]
        z2 = torch.cdist(take(tgt, t_ids), moved).min(dim=-1)[0]
    a2 = alpha * alpha
    return (2.0 - torch.exp(-0.5 * z1 * z1 / a2) - torch.exp(-0.5 * z2 * z2 / a2)).sum(dim=1).mean()


def training_loss(out, src, tgt, transform_gt, src_overlap, tgt_overlap, alpha=10.0, top_k=512):
    """train.py:54-72: 10*dcp + clu + mse + 0.01*welsch with NaN -> 0.  out = GMMReg.forward's 5-tuple; src, tgt [B,3,N];
    transform_gt [B,4,4]; *_overlap [B,N].  Returns (loss, dict of the four parts)."""
    R, t, so, to, clu = out
    B = R.shape[0]
    R_gt, t_gt = transform_gt[:, :3, :3], transform_gt[:, :3, 3].reshape(B, 3)
    parts = {"dcp": dcp_loss(R, R_gt, t, t_gt), "clu": clu, "mse": overlap_mse(so, to, src_overlap, tgt_overlap),
             "welsch": welsch_loss(src.transpose(1, 2), tgt.transpose(1, 2), R, t, src_overlap, tgt_overlap, alpha, top_k)}
    loss = torch.nan_to_num(10 * parts["dcp"] + parts["clu"] + parts["mse"] + 0.01 * parts["welsch"], nan=0.0)
    return loss, parts
