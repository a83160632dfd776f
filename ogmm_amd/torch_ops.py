"""The operator surface of the path as PyTorch custom ops: `torch.ops.ogmm.*` (SURVEY.md section 8b's list).

Every op is registered with the dispatcher through `torch.library.custom_op` -- schema, a CUDA(=ROCm) implementation that enqueues the
kernels of libogmm_hip.so through the C ABI (ogmm_amd/ops.py -> include/ogmm_hip.h), a fake (meta) implementation so that the ops trace
under FakeTensor / torch.compile / torch.export, and an autograd formula where the reference differentiates through the op
(kabsch, match_kabsch, gmm_feat_mean; the selection ops and the E/M loop are not differentiated in the reference either:
lib/utils.py:275-288 runs under no_grad, kNN / FPS return indices).  There is no CPU implementation: CPU tensors raise (the product path
has no fallback).  `GMMReg.forward` (the reference's call site: train.py:57,137) reaches its kernels through these ops where an op is one
kernel group, and through the same `ops.*` helpers with pre-packed weights where the model fuses across op boundaries;
tests/test_torch_ops.py assembles a whole forward from `torch.ops.ogmm.*` alone and checks it against the model.

Packed weights: an op that owns layers takes `Tensor[] packed_w` = 5 tensors per layer in the order
    [W fp32 [Cout, Kpad], scale [Cout] or empty, shift [Cout] or empty, W_hi image or empty, W_lo image or empty]
(the images: `ops.split_f16(W, frag=True)`) plus `float[] inv_scales` (one per layer) and `int[] meta` = [variant, ldb_h] per layer;
`pack_layers` / `unpack_layers` below convert from / to the dicts of `gmmreg.pack_weights`.
`precision`: 0 = exact-fp32 engine, 1 = fp16x3 (default), 2 = reduced single-term fp16; `overflow`: int32[1] device flag or None.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import ops
from ._lib import OgmmError
from .ops import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SIGMOID  # noqa: F401

PRECISIONS = ("f32", "f16x3", "f16")
BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------ packed layers <-> tensor lists
def pack_layers(layers):
    """[layer dict, ...] (gmmreg.pack_weights entries) -> (packed_w, inv_scales, meta)"""
    tensors, inv, meta = [], [], []
    for L in layers:
        W = L["W"]
        e = W.new_empty(0)
        sp = L.get("split")
        tensors += [W, L.get("scale", e) if L.get("scale") is not None else e, L.get("shift", e) if L.get("shift") is not None else e,
                    sp["W_hi"] if sp else W.new_empty(0, dtype=torch.float16), sp["W_lo"] if sp else W.new_empty(0, dtype=torch.float16)]
        inv.append(float(sp["inv_scale"]) if sp else 1.0)
        meta += [int(sp.get("variant", ops.PREC_F16X3)) if sp else -1, int(sp.get("ldb_h", 0)) if sp else 0]
    return tensors, inv, meta


def unpack_layers(packed_w, inv_scales, meta):
    out = []
    for i in range(len(inv_scales)):
        W, sc, sh, hi, lo = packed_w[5 * i:5 * i + 5]
        L = {"W": W}
        if sc.numel():
            L["scale"] = sc
        if sh.numel():
            L["shift"] = sh
        if hi.numel():
            L["split"] = {"W_hi": hi, "W_lo": lo, "inv_scale": inv_scales[i], "variant": meta[2 * i], "ldb_h": meta[2 * i + 1]}
        out.append(L)
    return out


def _eng(precision, overflow):
    return ops.Engine(PRECISIONS[precision], overflow)


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise OgmmError("torch.ops.ogmm.* need CUDA/ROCm tensors: the MI355X path has no CPU fallback")


# ------------------------------------------------------------------------------------------ selection ops (no gradient in the reference)
@custom_op("ogmm::knn_idx", mutates_args=(), device_types="cuda")
def knn_idx(xyz: Tensor, k: int) -> Tensor:
    """lib/utils.py:37-44: xyz [C,N,3] -> indices [C,N,k] (int32) of the k nearest points, self first, torch.topk's choice at rank-k ties"""
    _cuda(xyz)
    return ops.knn(xyz.contiguous(), k)


@knn_idx.register_fake
def _(xyz, k):
    return xyz.new_empty((xyz.shape[0], xyz.shape[1], k), dtype=torch.int32)


@custom_op("ogmm::fps", mutates_args=(), device_types="cuda")
def fps(xyz: Tensor, n: int, start: Optional[Tensor], center_start: bool) -> Tensor:
    """lib/utils.py:170-198: xyz [C,N,3] -> ids int32.  center_start: the `is_center=True` branch -> [C,n]; otherwise `start` int32 [S,C] holds
    the `torch.randint` draws of S samplings (lib/utils.py:190) -> [S,C,n]"""
    _cuda(xyz, start)
    if center_start != (start is None):
        raise OgmmError("fps: give `start` [S,C] (random-start samplings) or center_start=True, not both / neither")
    return ops.fps(xyz.contiguous(), n, None if center_start else start.to(torch.int32).contiguous())


@fps.register_fake
def _(xyz, n, start, center_start):
    C = xyz.shape[0]
    return xyz.new_empty((C, n) if center_start else (start.shape[0], C, n), dtype=torch.int32)


@custom_op("ogmm::gmm_em", mutates_args=(), device_types="cuda")
def gmm_em(xyz: Tensor, o: Tensor, ids0: Tensor, iters: int, sk_iters: int, eps: float, thresh: float, tau: float,
           group_size: int) -> Tuple[Tensor, Tensor, Tensor]:
    """lib/utils.py:269-288 (+ :69-108, :130-140): xyz [C,N,3], o [C,N], ids0 int32 [C,J] (the centre-start FPS picks) -> gamma [C,N,J], pi [C,J],
    mu [C,J,3].  thresh / group_size: the Sinkhorn early exit per reference call of `group_size` clouds (0: all C clouds are one call)."""
    _cuda(xyz, o, ids0)
    return ops.gmm_em(xyz.contiguous(), o.contiguous(), ids0.to(torch.int32).contiguous(), iters=iters, sk_iters=sk_iters, epsilon=eps,
                      tau=tau, thresh=thresh, group_size=group_size if group_size > 0 else None)


@gmm_em.register_fake
def _(xyz, o, ids0, iters, sk_iters, eps, thresh, tau, group_size):
    C, N, _ = xyz.shape
    J = ids0.shape[1]
    return xyz.new_empty((C, N, J)), xyz.new_empty((C, J)), xyz.new_empty((C, J, 3))


gmm_em.register_autograd(lambda ctx, *g: (None,) * 9)          # lib/utils.py:275-288: detached scores, torch.no_grad() loop


@custom_op("ogmm::clu_infonce", mutates_args=(), device_types="cuda")
def clu_infonce(xyz: Tensor, mu: Tensor, feats: Tensor, mu_feat: Tensor, tau: float) -> Tuple[Tensor, Tensor]:
    """lib/loss.py:109-118 with :22-57 and lib/utils.py:244-254: xyz [C,N,3], mu [C,J,3], feats [C*N,D], mu_feat [C,J,D] -> (row_loss [C,2,J]: the
    cross-entropy rows of ConLoss, whose mean over a call's clouds is that call's loss; near int32 [C,J]: the point nearest to each mu)"""
    _cuda(xyz, mu, feats, mu_feat)
    C, N, _ = xyz.shape
    return ops.clu_infonce(xyz.contiguous(), mu.contiguous(), feats, mu_feat.contiguous(), C, N, tau)


@clu_infonce.register_fake
def _(xyz, mu, feats, mu_feat, tau):
    C, J = mu.shape[0], mu.shape[1]
    return xyz.new_empty((C, 2, J)), xyz.new_empty((C, J), dtype=torch.int32)


# ------------------------------------------------------------------------------------------ differentiable head
@custom_op("ogmm::gmm_feat_mean", mutates_args=(), device_types="cuda")
def gmm_feat_mean(gamma: Tensor, pi: Tensor, feats: Tensor) -> Tensor:
    """lib/utils.py:130-140 on features: gamma [C,N,J], pi [C,J], feats [C*N,D] (row-major, last stride 1) -> gamma^T feats / (N pi + 1e-5) [C,J,D]"""
    _cuda(gamma, pi, feats)
    C, N, _ = gamma.shape
    return ops.gmm_feat_mean(gamma.contiguous(), pi.contiguous(), feats, C, N)


@gmm_feat_mean.register_fake
def _(gamma, pi, feats):
    return gamma.new_empty((gamma.shape[0], gamma.shape[2], feats.shape[1]))


def _feat_mean_setup(ctx, inputs, output):
    gamma, pi, _ = inputs
    ctx.save_for_backward(gamma, pi)


def _feat_mean_bwd(ctx, dmu):          # gradient to the features only (gamma, pi leave the no-grad E/M loop)
    gamma, pi = ctx.saved_tensors
    N = gamma.shape[1]
    df = torch.bmm(gamma, dmu / (pi * N + 1e-5)[:, :, None])
    return None, None, df.reshape(-1, dmu.shape[2])


gmm_feat_mean.register_autograd(_feat_mean_bwd, setup_context=_feat_mean_setup)


@custom_op("ogmm::kabsch", mutates_args=(), device_types="cuda")
def kabsch(src: Tensor, corr: Tensor, w: Tensor) -> Tuple[Tensor, Tensor]:
    """lib/se3.py:256-289: src, corr [B,3,J], w [B,1,J] -> R [B,3,3], t [B,3,1]; fp64 3x3 SVD in registers, no host round trip"""
    _cuda(src, corr, w)
    return ops.kabsch(src, corr, w)


@kabsch.register_fake
def _(src, corr, w):
    B = src.shape[0]
    return src.new_empty((B, 3, 3)), src.new_empty((B, 3, 1))


@custom_op("ogmm::kabsch_bwd", mutates_args=(), device_types="cuda")
def kabsch_bwd(src: Tensor, corr: Tensor, w: Tensor, gR: Tensor, gt: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """closed-form backward of `kabsch` through the 3x3 SVD -> (g_src, g_corr [B,3,J], g_w [B,1,J])"""
    _cuda(src, corr, w, gR, gt)
    B, _, J = src.shape
    g_src, g_corr, g_w = ops.kabsch_bwd(src, corr, w, gR.contiguous(), gt.reshape(B, 3).contiguous())
    return g_src, g_corr, g_w.view(B, 1, J)


@kabsch_bwd.register_fake
def _(src, corr, w, gR, gt):
    return torch.empty_like(src), torch.empty_like(corr), src.new_empty((src.shape[0], 1, src.shape[2]))


def _kabsch_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _kabsch_bwd(ctx, gR, gt):
    src, corr, w = ctx.saved_tensors
    gR = torch.zeros((src.shape[0], 3, 3), dtype=src.dtype, device=src.device) if gR is None else gR
    gt = torch.zeros((src.shape[0], 3, 1), dtype=src.dtype, device=src.device) if gt is None else gt
    g_src, g_corr, g_w = torch.ops.ogmm.kabsch_bwd(src, corr, w, gR, gt)
    return g_src, g_corr, g_w.view_as(w)


kabsch.register_autograd(_kabsch_bwd, setup_context=_kabsch_setup)


@custom_op("ogmm::match_kabsch", mutates_args=(), device_types="cuda")
def match_kabsch(mu_s: Tensor, mu_t: Tensor, f_s: Tensor, f_t: Tensor, temp: float) -> Tuple[Tensor, Tensor]:
    """models/dgcnn.py:96-115 (GMMSVD, is_sk=False) + lib/se3.py:256-289: mu_* [B,J,3], f_* [B,J,D] -> R [B,3,3], t [B,3]"""
    _cuda(mu_s, mu_t, f_s, f_t)
    return ops.match_kabsch(mu_s.contiguous(), mu_t.contiguous(), f_s.contiguous(), f_t.contiguous(), temp)


@match_kabsch.register_fake
def _(mu_s, mu_t, f_s, f_t, temp):
    B = mu_s.shape[0]
    return mu_s.new_empty((B, 3, 3)), mu_s.new_empty((B, 3))


def _match_setup(ctx, inputs, output):
    mu_s, mu_t, f_s, f_t, temp = inputs
    ctx.save_for_backward(mu_s, mu_t, f_s, f_t)
    ctx.temp = temp


def _match_bwd(ctx, gR, gt):
    """the matching softmax re-formed with torch ops (J x J per pair: tiny), the rigid solve's gradient from kabsch_bwd"""
    mu_s, mu_t, f_s, f_t = ctx.saved_tensors
    B, J, _ = mu_s.shape
    ns, nt = f_s.norm(dim=-1, keepdim=True).clamp_min(1e-12), f_t.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    fs, ft = f_s / ns, f_t / nt
    sc = torch.softmax(fs @ ft.transpose(1, 2) / ctx.temp, dim=2)
    corr = sc @ mu_t
    w = sc.sum(dim=2)
    gR = torch.zeros((B, 3, 3), dtype=mu_s.dtype, device=mu_s.device) if gR is None else gR
    gt = torch.zeros((B, 3), dtype=mu_s.dtype, device=mu_s.device) if gt is None else gt
    g_src, g_corr, g_w = torch.ops.ogmm.kabsch_bwd(mu_s.transpose(1, 2).contiguous(), corr.transpose(1, 2).contiguous(), w.view(B, 1, J),
                                                   gR.contiguous(), gt.reshape(B, 3, 1))
    g_corr = g_corr.transpose(1, 2)                                   # [B,J,3]
    g_sc = g_corr @ mu_t.transpose(1, 2) + g_w.view(B, J, 1)
    g_mu_t = sc.transpose(1, 2) @ g_corr
    g_sim = sc * (g_sc - (g_sc * sc).sum(-1, keepdim=True)) / ctx.temp
    g_fs, g_ft = g_sim @ ft, g_sim.transpose(1, 2) @ fs
    g_f_s = (g_fs - fs * (fs * g_fs).sum(-1, keepdim=True)) / ns
    g_f_t = (g_ft - ft * (ft * g_ft).sum(-1, keepdim=True)) / nt
    return g_src.transpose(1, 2), g_mu_t, g_f_s, g_f_t, None


match_kabsch.register_autograd(_match_bwd, setup_context=_match_setup)


# ------------------------------------------------------------------------------------------ weight-bearing stages (eval mode: BatchNorm folded into packed_w)
@custom_op("ogmm::edgeconv_dgcnn", mutates_args=("overflow",), device_types="cuda")
def edgeconv_dgcnn(xyz: Tensor, idx: Tensor, packed_w: List[Tensor], inv_scales: List[float], meta: List[int], precision: int,
                   overflow: Optional[Tensor]) -> Tensor:
    """models/dgcnn.py:133-154: graph features + conv1..4 (+BN+ReLU, max over k) + conv5: xyz [C,N,3], idx int32 [C,N,k], the five packed layers
    emd1..emd5 -> feats [C*N, D].  The per-edge tensors never leave the chip (edgeconv_fused.hip)."""
    _cuda(xyz, idx)
    eng = _eng(precision, overflow)
    L = unpack_layers(packed_w, inv_scales, meta)
    C, N, k = idx.shape
    xcat = torch.empty((C * N, 512), dtype=torch.float32, device=xyz.device)
    if eng.split and ops.edgeconv_fused_supported(k, L[:4]):
        ops.edgeconv_fused(xyz.contiguous(), idx, L[:4], xcat)
    else:
        h = ops.edgeconv_first(xyz.contiguous(), idx, L[0], xcat[:, 0:64])
        h = ops.edgeconv_layer(h, L[1], k, xcat[:, 64:128], eng=eng)
        h = ops.edgeconv_layer(h, L[2], k, xcat[:, 128:256], eng=eng)
        ops.edgeconv_layer(h, L[3], k, xcat[:, 256:512], store=False, eng=eng)
    return ops.conv1x1(xcat, L[4], ACT_RELU, eng=eng)


@edgeconv_dgcnn.register_fake
def _(xyz, idx, packed_w, inv_scales, meta, precision, overflow):
    return xyz.new_empty((xyz.shape[0] * xyz.shape[1], packed_w[20].shape[0]))


@custom_op("ogmm::pos_encoding", mutates_args=("overflow",), device_types="cuda")
def pos_encoding(xyz: Tensor, idx5: Tensor, front: List[Tensor], packed_w: List[Tensor], inv_scales: List[float], meta: List[int], precision: int,
                 overflow: Optional[Tensor]) -> Tensor:
    """models/attn.py:59-75: xyz [C,N,3], idx5 = knn_idx(xyz, 5) -> [C*N, D] (distance channels | angle channels).  front = the six small tensors of
    the 1 -> 64 layers (w_dis, s_dis, t_dis, w_ang, s_ang, t_ang), packed_w = the two 64 -> D/2 layers."""
    _cuda(xyz, idx5)
    eng = _eng(precision, overflow)
    p = dict(zip(("w_dis", "s_dis", "t_dis", "w_ang", "s_ang", "t_ang"), front))
    dis2, ang2 = unpack_layers(packed_w, inv_scales, meta)
    hd, ha = ops.pos_hidden(xyz.contiguous(), idx5, 5, p)
    half = dis2["W"].shape[0]
    out = torch.empty((hd.shape[0], 2 * half), dtype=torch.float32, device=xyz.device)
    ops.conv1x1(hd, dis2, ACT_LEAKY02, out=out[:, :half], eng=eng)
    ops.conv1x1(ha, ang2, ACT_LEAKY02, out=out[:, half:], eng=eng)
    return out


@pos_encoding.register_fake
def _(xyz, idx5, front, packed_w, inv_scales, meta, precision, overflow):
    return xyz.new_empty((xyz.shape[0] * xyz.shape[1], 2 * packed_w[0].shape[0]))


@custom_op("ogmm::conv_mlp", mutates_args=("overflow",), device_types="cuda")
def conv_mlp(x: Tensor, x2: Optional[Tensor], packed_w: List[Tensor], inv_scales: List[float], meta: List[int], acts: List[int], precision: int,
             overflow: Optional[Tensor], res: Optional[Tensor]) -> Tensor:
    """models/dgcnn.py:16-38 (`CONV`): a chain of 1x1 convolutions with folded BatchNorm and the activation acts[i] after layer i, on point-major
    maps: (x | x2) [rows, K] -> [rows, Cout] (+ res)."""
    _cuda(x, x2, res)
    eng = _eng(precision, overflow)
    L = unpack_layers(packed_w, inv_scales, meta)
    h = x
    for i, layer in enumerate(L):
        last = i + 1 == len(L)
        h = ops.conv1x1(h, layer, acts[i], x2=x2 if i == 0 else None, res=res if last else None, eng=eng)
    return h


@conv_mlp.register_fake
def _(x, x2, packed_w, inv_scales, meta, acts, precision, overflow, res):
    return x.new_empty((x.shape[0], packed_w[5 * (len(acts) - 1)].shape[0]))


@custom_op("ogmm::anchor_transformer", mutates_args=("overflow",), device_types="cuda")
def anchor_transformer(x: Tensor, anchor_feats: Tensor, anchor_ids: Tensor, cloud_map: Optional[Tensor], n_points: int, heads: int,
                       packed_w: List[Tensor], inv_scales: List[float], meta: List[int], precision: int, overflow: Optional[Tensor]) -> Tensor:
    """models/attn.py:78-111 (`Transformer`): x [C*N, D] attends to the anchors = rows anchor_ids [C,M] of anchor_feats [C*N, D] (of cloud
    cloud_map[c] when given: the cross-attention of models/gmmreg.py:67-72) -> mlp(cat[x, merge(attention)]) + x.  packed_w: the layers
    q, kv (keys | values), mlp0 with the merge convolution folded in, mlp3 (head-major channel order: gmmreg.pack_weights)."""
    _cuda(x, anchor_feats, anchor_ids, cloud_map)
    eng = _eng(precision, overflow)
    q_l, kv_l, mlp0, mlp3 = unpack_layers(packed_w, inv_scales, meta)
    D = x.shape[1]
    N = n_points
    C = x.shape[0] // N
    M = anchor_ids.shape[1]
    if not ops.attention_supported(M, D // heads):
        raise OgmmError("anchor_transformer: the fused attention kernel takes head dim 128 and 32 / 64 / 128 anchors")
    q = ops.conv1x1(x, q_l, eng=eng)
    kv = ops.conv1x1_gathered(anchor_feats, C, N, anchor_ids, kv_l, cloud_map=cloud_map, eng=eng)
    o = ops.attention(q, kv[:, :D], kv[:, D:], C, N, M, heads)
    if eng.split and ops.instnorm_fusable(mlp0.get("split"), N):
        stats = torch.zeros((C, 2 * D, 2), dtype=torch.float64, device=x.device)
        z = ops.conv1x1(x, mlp0, x2=o, col_stats=stats, group_rows=N, eng=eng)
        a_sc, a_sh = ops.instnorm_finalize(stats, N, BN_EPS)
        return ops.conv1x1(z, mlp3, res=x, a_affine=(a_sc, a_sh, True), group_rows=N, eng=eng)
    z = ops.conv1x1(x, mlp0, x2=o, eng=eng)
    ops.instnorm_relu_(z, C, N, BN_EPS)
    return ops.conv1x1(z, mlp3, res=x, eng=eng)


@anchor_transformer.register_fake
def _(x, anchor_feats, anchor_ids, cloud_map, n_points, heads, packed_w, inv_scales, meta, precision, overflow):
    return torch.empty_like(x)


@custom_op("ogmm::overlap_cross", mutates_args=("overflow",), device_types="cuda")
def overlap_cross(fs: Tensor, ft: Tensor, o_s: Tensor, o_t: Tensor, precision: int, overflow: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """models/gmmreg.py:74-80: fs, ft [B,N,D] (un-normalised), o_s, o_t [B,N] overlap logits -> (wo_s, wo_t) [B,N]: the row- / column-softmax of the
    N x N cosine similarity applied to the logits.  The similarity lives only in the GEMM's accumulators where the engine takes the shape."""
    _cuda(fs, ft, o_s, o_t)
    eng = _eng(precision, overflow)
    B, N, D = fs.shape
    f_s, f_t = fs.reshape(B * N, D), ft.reshape(B * N, D)
    os_, ot_ = o_s.contiguous(), o_t.contiguous()
    wo_s, wo_t = torch.empty_like(os_), torch.empty_like(ot_)
    if D % 64 == 0 and ops.overlap_fusable(B, N, D, eng):
        img = ops.l2norm_pack_frag_batched(f_t, B, N)
        ops.overlap_fused(f_s, img, B, N, D, os_.view(-1), ot_.view(-1), 1, wo_s.view(-1), wo_t.view(-1), 1, overflow=overflow)
        return wo_s, wo_t
    S = torch.empty((B, N, N), dtype=torch.float32, device=fs.device)
    if eng.split and D % 64 == 0:
        fn_src = ops.l2norm_rows(f_s)
        img = ops.l2norm_pack_frag_batched(f_t, B, N)
        ops.gemm_nt(fn_src, D, D, None, D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sC=(N * N, 0), split=img, overflow=overflow,
                    single_term=eng.single_term)
    else:
        fn_s, fn_t = ops.l2norm_rows(f_s), ops.l2norm_rows(f_t)
        ops.gemm_nt(fn_s, D, D, fn_t, D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sB=(N * D, 0), sC=(N * N, 0))
    ops.overlap_cross(S, os_.view(-1), ot_.view(-1), 1, wo_s.view(-1), wo_t.view(-1), 1)
    return wo_s, wo_t


@overlap_cross.register_fake
def _(fs, ft, o_s, o_t, precision, overflow):
    return torch.empty_like(o_s), torch.empty_like(o_t)


# SURVEY.md section 8b's operator list -> the registered schema (checked by tests/test_torch_ops.py).  `overflow` is declared as MUTATED (Tensor(a!)?):
# the engines raise that device flag; functionalisation under torch.compile / export must not drop or reorder the write.
SCHEMAS = {
    "knn_idx": "ogmm::knn_idx(Tensor xyz, SymInt k) -> Tensor",
    "edgeconv_dgcnn": "ogmm::edgeconv_dgcnn(Tensor xyz, Tensor idx, Tensor[] packed_w, float[] inv_scales, SymInt[] meta, SymInt precision, Tensor(a6!)? overflow) -> Tensor",
    "fps": "ogmm::fps(Tensor xyz, SymInt n, Tensor? start, bool center_start) -> Tensor",
    "pos_encoding": "ogmm::pos_encoding(Tensor xyz, Tensor idx5, Tensor[] front, Tensor[] packed_w, float[] inv_scales, SymInt[] meta, SymInt precision, Tensor(a7!)? overflow) -> Tensor",
    "anchor_transformer": "ogmm::anchor_transformer(Tensor x, Tensor anchor_feats, Tensor anchor_ids, Tensor? cloud_map, SymInt n_points, SymInt heads, Tensor[] packed_w, float[] inv_scales, SymInt[] meta, SymInt precision, Tensor(a10!)? overflow) -> Tensor",
    "conv_mlp": "ogmm::conv_mlp(Tensor x, Tensor? x2, Tensor[] packed_w, float[] inv_scales, SymInt[] meta, SymInt[] acts, SymInt precision, Tensor(a7!)? overflow, Tensor? res) -> Tensor",
    "overlap_cross": "ogmm::overlap_cross(Tensor fs, Tensor ft, Tensor o_s, Tensor o_t, SymInt precision, Tensor(a5!)? overflow) -> (Tensor, Tensor)",
    "gmm_em": "ogmm::gmm_em(Tensor xyz, Tensor o, Tensor ids0, SymInt iters, SymInt sk_iters, float eps, float thresh, float tau, SymInt group_size) -> (Tensor, Tensor, Tensor)",
    "gmm_feat_mean": "ogmm::gmm_feat_mean(Tensor gamma, Tensor pi, Tensor feats) -> Tensor",
    "match_kabsch": "ogmm::match_kabsch(Tensor mu_s, Tensor mu_t, Tensor f_s, Tensor f_t, float temp) -> (Tensor, Tensor)",
    "kabsch": "ogmm::kabsch(Tensor src, Tensor corr, Tensor w) -> (Tensor, Tensor)",
    "clu_infonce": "ogmm::clu_infonce(Tensor xyz, Tensor mu, Tensor feats, Tensor mu_feat, float tau) -> (Tensor, Tensor)",
}
