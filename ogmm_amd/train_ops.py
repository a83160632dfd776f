"""Operations of the training graph (ogmm_amd/train_graph.py), each differentiable where the reference is.

`TrainOps` is the product implementation: discrete selections and the E/M loop run on the HIP kernels of
libogmm_hip.so (no gradient flows through them in the reference either); dense layers run forward, dX = dY W and (wide
layers) dW = dY^T X on the GEMM engine, thin-layer dW on an exact-fp32 MFMA reduction kernel; normalisation (+ pooling), the
overlap block, the rigid solve, L2 normalisation and the cluster means have hand-written forward AND backward kernels wrapped
in `torch.autograd.Function`.  Still library calls: the attention backward for M != 128 anchors, dfn = dS fn of the overlap block, thin-layer
forwards (batched / plain hipBLASLt GEMMs through torch.matmul).  CPU tensors are rejected by the kernels' wrappers: there is
no CPU path here.  The plain-PyTorch statement of the same
operations that the tests use as the numerical reference lives in tests/train_ref.py.
"""

import torch
import torch.nn.functional as F

from . import ops
from ._lib import OgmmError

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _rm(t):
    """t as the kernels take it: a row-major 2-D map with unit column stride and 16-byte aligned rows -- a column slice of a wider buffer
    (one half of a two-piece layer's input gradient) qualifies and is NOT copied"""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


_ACT = {"relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY02, "none": ops.ACT_NONE}


class _NormAct(torch.autograd.Function):
    """act((y - mean_g) * rstd_g * weight + bias) over row groups: kernels T1 of include/ogmm_hip.h.
    Also returns the fp64 group means / biased variances for the running-statistics update."""

    @staticmethod
    def forward(ctx, y, weight, bias, group_rows, act, st=None):
        y = y.contiguous()
        if st is None:                     # otherwise: column sums that the producing GEMM's epilogue already accumulated
            st = ops.colstats(y, group_rows)
        scale, shift, mean, rstd, mean64, var64 = ops.norm_finalize(st, group_rows, weight, bias, BN_EPS)
        h = ops.affine_act(y, group_rows, scale, shift, act)
        ctx.save_for_backward(y, scale, shift, mean, rstd)
        ctx.group_rows, ctx.act, ctx.affine = group_rows, act, weight is not None
        ctx.mark_non_differentiable(mean64, var64)
        return h, mean64, var64

    @staticmethod
    def backward(ctx, dh, _dm, _dv):
        y, scale, shift, mean, rstd = ctx.saved_tensors
        dy, sums = ops.norm_bwd(y, _rm(dh), ctx.group_rows, scale, shift, mean, rstd, ctx.act)
        if not ctx.affine:
            return dy, None, None, None, None, None
        dg, dbeta = ops.norm_param_grads(sums)
        return dy, dg, dbeta, None, None, None


class _NormActPool(torch.autograd.Function):
    """_NormAct followed by the max over the k rows of every point in ONE pass (ogmm_affine_act_pool); the normalised per-edge
    map itself is written only when a later layer reads it.  Backward: the pooled gradient is routed to the winning rows
    inside the normalisation's backward kernels (no scattered per-edge gradient map, no separate add)."""

    @staticmethod
    def forward(ctx, y, weight, bias, group_rows, act, k, want_h, st=None):
        y = y.contiguous()
        if st is None:
            st = ops.colstats(y, group_rows)
        scale, shift, mean, rstd, mean64, var64 = ops.norm_finalize(st, group_rows, weight, bias, BN_EPS)
        h, pooled, arg = ops.affine_act_pool(y, k, group_rows, scale, shift, act, want_y=want_h)
        ctx.save_for_backward(y, scale, shift, mean, rstd, arg)
        ctx.group_rows, ctx.act, ctx.affine, ctx.k = group_rows, act, weight is not None, k
        ctx.set_materialize_grads(False)
        if h is None:
            h = y.new_empty(0)
        ctx.mark_non_differentiable(mean64, var64)
        return h, pooled, mean64, var64

    @staticmethod
    def backward(ctx, dh, dpooled, _dm, _dv):
        y, scale, shift, mean, rstd, arg = ctx.saved_tensors
        if dh is not None and dh.numel() == 0:
            dh = None
        if dh is None and dpooled is None:
            return (None,) * 8
        dy, sums = ops.norm_bwd(y, None if dh is None else _rm(dh), ctx.group_rows, scale, shift, mean, rstd, ctx.act,
                                dpool=None if dpooled is None else dpooled.contiguous(), arg=arg, k=ctx.k)
        if not ctx.affine:
            return (dy,) + (None,) * 7
        return (dy,) + ops.norm_param_grads(sums) + (None,) * 5


class _MaxPoolK(torch.autograd.Function):
    """kernels T2 of include/ogmm_hip.h"""

    @staticmethod
    def forward(ctx, h, k):
        out, arg = ops.maxpool_k(h.contiguous(), k)
        ctx.save_for_backward(arg)
        ctx.k = k
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        return ops.maxpool_k_bwd(dout.contiguous(), arg, ctx.k), None


class _SmallNN(torch.autograd.Function):
    """out[b] = S[b] X[b]: S [B, R, m], X [B, m, D], contraction m <= 1024 (ops.small_bmm_nn, kernel T10).  Backward on the same two kernels:
    dS = dOut X^T (both tiny: the nt form; a thin D: the nn form over D), dX = S^T dOut (the nn form over R <= 128)."""

    @staticmethod
    def forward(ctx, S, X):
        ctx.save_for_backward(S, X)
        return ops.small_bmm_nn(S, X)

    @staticmethod
    def backward(ctx, g):
        S, X = ctx.saved_tensors
        g = g.contiguous()
        dS = dX = None
        R, m, D = S.shape[1], S.shape[2], X.shape[2]
        if ctx.needs_input_grad[0]:
            if R <= 256 and m <= 256:
                dS = ops.small_bmm_nt(g, X.contiguous())
            elif D <= 128:
                dS = ops.small_bmm_nn(g, X.transpose(1, 2))
            else:
                raise OgmmError("_SmallNN: no kernel for dS at R=%d m=%d D=%d" % (R, m, D))
        if ctx.needs_input_grad[1]:
            # a contraction over the rows: the nn form up to 128 of them, the nt form (both operands as [., R] rows) beyond
            dX = ops.small_bmm_nn(S.transpose(1, 2), g) if R <= 128 else ops.small_bmm_nt(S.transpose(1, 2).contiguous(), g.transpose(1, 2).contiguous())
        return dS, dX


class _SmallNT(torch.autograd.Function):
    """out[b] = alpha A[b] B[b]^T: A [B, n, D], B [B, m, D] (ops.small_bmm_nt, kernel T10); dA = alpha dOut B, dB = alpha dOut^T A (nn form)."""

    @staticmethod
    def forward(ctx, A, Bm, alpha):
        A, Bm = A.contiguous(), Bm.contiguous()
        ctx.save_for_backward(A, Bm)
        ctx.alpha = alpha
        return ops.small_bmm_nt(A, Bm, alpha)

    @staticmethod
    def backward(ctx, g):
        A, Bm = ctx.saved_tensors
        g = g.contiguous() * ctx.alpha if ctx.alpha != 1.0 else g.contiguous()
        dA = ops.small_bmm_nn(g, Bm) if ctx.needs_input_grad[0] else None
        dB = ops.small_bmm_nn(g.transpose(1, 2), A) if ctx.needs_input_grad[1] else None
        return dA, dB, None


def small_bmm(a, b):
    """torch.bmm(a, b) for the training step's small products: a [B, R, m] x b [B, m, D], m <= 128, on kernel T10 with autograd"""
    return _SmallNN.apply(a, b)


def small_bmm_nt(a, b, alpha=1.0):
    """alpha * torch.bmm(a, b^T) for a [B, n, D], b [B, m, D]"""
    return _SmallNT.apply(a, b, alpha)


class _ThinLinear(torch.autograd.Function):
    """y = x W^T + b for layers too thin for a matrix-core tile in the forward direction (the 6 -> 64 edge layer, the 1 -> 64
    positional layers): the forward is an HBM-bound outer product and dX = dY W a 64-term row reduction (kernel T10: ops.small_bmm_nn, exact fp32 --
    library GEMMs until round 6), the weight gradient the thin reduction kernel T9."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        if not x.is_cuda:          # (CPU tensors only reach this through the tests' wiring seam)
            y = x @ W.t()
            return y if b is None else y + b
        return ops.small_bmm_nn(x[None], W.detach().t()[None], bias=None if b is None else b.detach())[0]

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = dW = db = None
        dyc = dy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = ops.small_bmm_nn(dyc[None], W.detach()[None])[0] if dyc.is_cuda and W.shape[0] <= 1024 else dyc @ W
        if ctx.needs_input_grad[1]:
            xc = x.contiguous()
            dW = ops.weight_grad_thin(dyc, xc) if ops.weight_grad_thin_supported(dyc, xc) else dyc.t() @ xc
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dyc.sum(dim=0)
        return dx, dW, db


def _transposed_weight_layer(W):
    """The packed layer of dX = dY W for the engine: the split fragment image of W^T [K, Cout] written straight from W (ops.split_f16_training(transpose=True):
    two launches, no transposed copy); K is rounded up to a multiple of 4 (16-byte output rows), the extra output columns are zero.  `W` of the layer is a
    shape-only stand-in: the fragment engines never read the fp32 operand."""
    Wd = W.detach()
    cout, k = Wd.shape
    k4 = (k + 3) // 4 * 4
    sp = ops.split_f16_training(Wd, k4, transpose=True, frag=True)
    return {"W": Wd.new_empty(1).expand(k4, cout), "split": sp, "scale": sp["col_scale"]}


class _Linear(torch.autograd.Function):
    """y = [x | x2] W^T + b.  Forward and dX = dY W on the GEMM engine (fp16x3 split of the CURRENT weights, or exact fp32
    forward + library dX with precision "f32"); dW = dY^T X runs on the engine as a split-K GEMM over transposed operands
    (kernels T3) for the wide layers and as a plain library GEMM (hipBLASLt through torch.matmul) for the thin ones; db is a
    column sum."""

    @staticmethod
    def forward(ctx, x, x2, W, b, precision, overflow, stats_rows=0):
        K1 = x.shape[1]
        K2 = 0 if x2 is None else x2.shape[1]
        Wd = W.detach()
        if K2 % 4:                                        # 16-byte rows for the second piece (conv2.net.0: 512 + 2 channels)
            pad = 4 - K2 % 4
            x2p = torch.cat([x2, x2.new_zeros(x2.shape[0], pad)], dim=1)
            Wd = torch.cat([Wd, Wd.new_zeros(Wd.shape[0], pad)], dim=1)
        else:
            x2p = x2
        layer = {"W": Wd.contiguous()}
        if b is not None:
            layer["shift"] = b.detach().contiguous()
        if precision == "f16x3":
            layer["split"] = ops.split_f16_training(layer["W"], Wd.shape[0], frag=True, k1=K1)
            layer["scale"] = layer["split"]["col_scale"]
        stats = None
        if stats_rows and precision == "f16x3" and stats_rows % 256 == 0 and W.shape[0] % 4 == 0 and (stats_rows <= 131072 or W.shape[0] >= 256):
            # the normalisation that follows needs sum / sum of squares per (row group, column): the engine's epilogue adds them up
            # (one fp64 atomic per tile and column: fine for <= 512 row tiles per group; on the 2.6 M-row per-edge maps, whose
            # 10240 tiles per group would all hit the same few addresses -- measured 12x slower than a separate pass -- the tiles
            # are dealt over 64 copies of the table, summed afterwards.  Only for >= 256 output channels, where the LDS-DMA engines
            # issue one atomic per tile and column: the 64- and 128-channel maps run on the 128 x 128-tile kernel, whose epilogue
            # issues them per wave -- 10 M fp64 atomics per map cost it 0.4-0.55 ms, more than the separate pass (0.2-0.55 ms))
            slots = 1 if stats_rows <= 131072 else 64
            stats = torch.zeros((slots, x.shape[0] // stats_rows, W.shape[0], 2)[0 if slots > 1 else 1:], dtype=torch.float64, device=x.device)
        y = ops.conv1x1(_rm(x), layer, ops.ACT_NONE, x2=None if x2p is None else _rm(x2p),
                        split=precision == "f16x3", overflow=overflow, col_stats=stats, group_rows=stats_rows if stats is not None else 0)
        ctx.save_for_backward(x, x2, W)
        ctx.has_bias, ctx.precision, ctx.overflow = b is not None, precision, overflow
        # a bias in front of a normalisation has an exactly-zero gradient (the normalisation's backward output sums to zero over
        # every column of a group); the reference's autograd returns rounding noise of order 1e-8 there.  Skip the column sums.
        ctx.bias_grad_is_zero = bool(stats_rows)
        if stats_rows:
            if stats is None:
                stats = ops.colstats(y, stats_rows)
            elif stats.dim() == 4:
                stats = stats.sum(dim=0)
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        x, x2, W = ctx.saved_tensors
        K1 = x.shape[1]
        dx = dx2 = dW = db = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            if ctx.precision == "f16x3":
                # dX = dY W on the engine: the weight operand is W^T [K, Cout], split per step.  Activation gradients sit far
                # below binary16's normal range (max 1e-6 .. 0.2 per layer at loss scale 1); the trainer's power-of-two loss
                # scale (exact in fp32) lifts them into it, the overflow flag reports a scale that is too large.
                layer = _transposed_weight_layer(W)
                dall = ops.conv1x1(_rm(dy), layer, ops.ACT_NONE, split=True, overflow=ctx.overflow, terms=BWD_TERMS_DX)
            else:
                dall = dy @ W
            dx = dall[:, :K1]
            dx2 = dall[:, K1:K1 + x2.shape[1]] if x2 is not None else None
        if ctx.needs_input_grad[2]:
            # kernels T3 for every piece with >= 256 input channels when the layer has >= 256 outputs.  Measured at R = 131072
            # (tools/dw_bench.py, ms engine incl. relayouts vs library): 1024x1024 1.61 vs 2.73, 512x1024 0.95 vs 1.09,
            # 1024x512 0.98 vs 1.09, 512x512 0.62 vs 0.73, 256x512 0.39 vs 1.12.  Thin pieces (the per-edge maps with <= 128
            # channels, the 2-channel overlap input) are HBM-bound row sums: library GEMM.
            pieces = [x] if x2 is None else [x, x2]
            wide = [ctx.precision == "f16x3" and W.shape[0] >= 256 and p_.shape[1] >= 256 for p_ in pieces]
            parts = [None] * len(pieces)
            want_db = ctx.has_bias and ctx.needs_input_grad[3] and not ctx.bias_grad_is_zero
            if any(wide):
                got = ops.weight_grad(_rm(dy), [_rm(p_) for p_, w_ in zip(pieces, wide) if w_], ctx.overflow, colsum=want_db, terms=BWD_TERMS_DW)
                if want_db:
                    got, db = got
                off = 0
                for i_, (p_, w_) in enumerate(zip(pieces, wide)):
                    if w_:
                        parts[i_] = got[:, off:off + p_.shape[1]]
                        off += p_.shape[1]
            if not all(wide):
                dyc = dy.contiguous()
                for i_, (p_, w_) in enumerate(zip(pieces, wide)):
                    if not w_:
                        # (round 5: on the engines' fp16x3 arithmetic in the fp16x3 step -- the exact-fp32 reduction is matrix-bound on the 256 x 128 per-edge layer)
                        parts[i_] = (ops.weight_grad_thin(dyc, p_, split=ctx.precision == "f16x3", overflow=ctx.overflow)
                                     if ops.weight_grad_thin_supported(dyc, p_) else dyc.t() @ p_)
            dW = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        if ctx.has_bias and ctx.needs_input_grad[3] and db is None:
            db = torch.zeros(dy.shape[1], dtype=dy.dtype, device=dy.device) if ctx.bias_grad_is_zero else dy.sum(dim=0)
        return dx, dx2, dW, db, None, None, None


# Term budget of the BACKWARD GEMMs (struct ogmm_gemm.terms; 0 / 3 = three binary16 products per fp32 product, 2 = the B operand rounded to binary16).  Default:
# THREE terms everywhere.  Both reduced forms were built and measured this round (tools/bwd_terms_check.py at 32 pairs of 1024 points, both weight families;
# HISTORY.md section 7) and are opt-in switches, not defaults:
#   BWD_TERMS_DW = 2     dW = dY^T X with the fragment image of X^T (activations) rounded: -3.5 ms per 128-pair step (115.3 -> 111.6).  The weight gradients of the
#     wide layers move by up to 1.0e-4 relative (median over all parameters 1.6e-7, p90 5e-5) -- a third of the reference's own fp32-vs-fp64 distance (2.5e-4 ...
#     3.2e-4 median) and a sixth of the distance between two fp32-class evaluations of the same step (split engine vs exact-fp32 engine: 5e-4 ... 6e-4); on the
#     reference-generated fixtures of that size the distance to the fp64 truth is unchanged to two digits for most parameters and grows by < 1e-4 for all (one
#     whose three-term gradient is unusually accurate goes 4.9e-5 -> 7.5e-5; test_two_term_weight_gradient_stays_at_the_three_term_distance_from_the_truth).  But a single layer's dW is then 1e-4 from its fp64 value where three terms
#     give 2e-6 (the layer tests' 2e-5 bar): an fp32-class engine by default, a labelled option for who wants the 3 %.
#   BWD_TERMS_DX = 2     dX = dY W with W^T rounded: a rounded WEIGHT is a fixed perturbation of every row's gradient -- 2.6e-4 relative on ALL parameters, the size
#     of the reference's own distance; -2.1 ms.  Not recommended.
# (module attributes: tools/bwd_terms_check.py and the tests assign them; nothing reads the environment)
BWD_TERMS_DX = 0
BWD_TERMS_DW = 0
FUSE_NORM_LINEAR = True      # 0: write the normalised maps (A/B timing)


class _NormLinear(torch.autograd.Function):
    """relu((y - mean_g) rstd_g weight + bias) W^T + b without ever writing the normalised map: the forward GEMM reads y through
    struct ogmm_gemm.a_scale (relu(y * scale_g + shift_g) while the operand is split), the weight gradient's operand image is built
    from y the same way (ogmm_pack_frag_t a_scale), and the normalisation's own backward is _NormAct's (it recomputes the activation
    mask from y).  One fewer 2-byte-per-byte pass over every hidden map, and the map is not kept for the backward either.
    Conditions (TrainOps._norm_linear_fusable): fp16x3 engine, ReLU, group_rows a multiple of the engine's 256-row tile, a wide layer."""

    @staticmethod
    def forward(ctx, y, st, nweight, nbias, group_rows, W, b, overflow, stats_rows, res=None):
        y = y.contiguous()
        scale, shift, mean, rstd, mean64, var64 = ops.norm_finalize(st, group_rows, nweight, nbias, BN_EPS)
        Wd = W.detach().contiguous()
        sp = ops.split_f16_training(Wd, Wd.shape[0], frag=True, k1=Wd.shape[1])
        layer = {"W": Wd, "split": sp, "scale": sp["col_scale"]}
        if b is not None:
            layer["shift"] = b.detach().contiguous()
        stats = None
        if stats_rows and stats_rows == group_rows and W.shape[0] % 4 == 0 and stats_rows <= 131072:
            stats = torch.zeros((y.shape[0] // stats_rows, W.shape[0], 2), dtype=torch.float64, device=y.device)
        # res: the block's residual (models/gmmreg.py:62, 73, 97: x + transformer(x)) rides in the GEMM's epilogue; its gradient is dout itself
        out = ops.conv1x1(y, layer, ops.ACT_NONE, split=True, overflow=overflow, col_stats=stats, a_affine=(scale, shift, True), group_rows=group_rows,
                          res=None if res is None else _rm(res))
        ctx.save_for_backward(y, scale, shift, mean, rstd, W)
        ctx.has_res = res is not None
        ctx.group_rows, ctx.affine, ctx.has_bias, ctx.overflow = group_rows, nweight is not None, b is not None, overflow
        ctx.bias_grad_is_zero = bool(stats_rows)          # see _Linear
        ctx.mark_non_differentiable(mean64, var64)
        if stats_rows:
            if stats is None:
                stats = ops.colstats(out, stats_rows)
            ctx.mark_non_differentiable(stats)
            return out, mean64, var64, stats
        return out, mean64, var64

    @staticmethod
    def backward(ctx, dout, _dm, _dv, _dst=None):
        y, scale, shift, mean, rstd, W = ctx.saved_tensors
        dout = _rm(dout)
        layer = _transposed_weight_layer(W)
        rows, cin = y.shape
        fused = ops.norm_bwd_fusable(rows, cin, dout.shape[1], ctx.group_rows)
        if fused:
            # the normalisation backward's reduction pass (sum dz, sum dz xhat per group and channel: a read of y and of dh, 2 x 1 GB per 1024-wide map) rides in
            # the epilogue of the GEMM that produces dh: it reads y there, stores dz = dh * relu'(.) instead of dh and leaves the two sums in col_stats
            sums = torch.zeros((rows // ctx.group_rows, cin, 2), dtype=torch.float64, device=y.device)
            dz = ops.conv1x1(dout, layer, ops.ACT_NONE, split=True, overflow=ctx.overflow, col_stats=sums, group_rows=ctx.group_rows,
                             norm_bwd=(y, mean, rstd, scale, shift, ops.ACT_RELU))
        else:
            dh = ops.conv1x1(dout, layer, ops.ACT_NONE, split=True, overflow=ctx.overflow, terms=BWD_TERMS_DX)
        dW = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[6] and not ctx.bias_grad_is_zero
        if ctx.needs_input_grad[5]:
            dW = ops.weight_grad(dout, [y], ctx.overflow, x_affine=(scale, shift, True, ctx.group_rows), colsum=want_db, terms=BWD_TERMS_DW)
            if want_db:
                dW, db = dW
        if ctx.has_bias and ctx.needs_input_grad[6] and db is None:
            db = torch.zeros(dout.shape[1], dtype=dout.dtype, device=dout.device) if ctx.bias_grad_is_zero else dout.sum(dim=0)
        if fused:
            dy = ops.norm_bwd_apply(y, dz, ctx.group_rows, scale, shift, mean, rstd, sums)
        else:
            dy, sums = ops.norm_bwd(y, dh, ctx.group_rows, scale, shift, mean, rstd, ops.ACT_RELU)
        dg, dbeta = ops.norm_param_grads(sums) if ctx.affine else (None, None)
        return dy, None, dg, dbeta, None, dW, db, None, None, (dout if ctx.has_res else None)


GATHER_SPARSE = True      # 0: the anchor gathers' gradients as dense zero-filled maps through ogmm_add_n (A/B)


class _Fanout(torch.autograd.Function):
    """n handles of one feature map for its n consumers; the backward adds their gradients in ONE pass (ogmm_add_n) instead of autograd's
    n - 1 pairwise accumulations (each 2 reads + 1 write of the map).  `stash` (round 5): a consumer that only GATHERS rows of its handle
    (_GatherRows: the anchors, 128 of a cloud's 1024 rows) does not return a dense gradient map -- zero-filled, scattered into and read again by
    the sum, 1.1 GB of traffic per gather at 128 pairs -- but leaves (rows, gradient rows) here, and the sum gets them by one index_add_ on 1 / 8 of its rows."""

    @staticmethod
    def forward(ctx, x, n, stash=None):
        ctx.stash, ctx.x_shape = stash, tuple(x.shape)
        ctx.set_materialize_grads(False)          # a handle whose consumer left its gradient in the stash arrives as None, not as a zero map
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g_ for g_ in grads if g_ is not None]
        sparse = list(ctx.stash) if ctx.stash else []
        if ctx.stash:
            ctx.stash.clear()
        if not gs and not sparse:
            return None, None, None
        if not gs:
            out = torch.zeros(ctx.x_shape, dtype=sparse[0][1].dtype, device=sparse[0][1].device)
        elif len(gs) == 1:
            out = gs[0].clone() if sparse else gs[0]          # (never scatter into a tensor another node may hold)
        elif gs:
            gs = [_rm(g_) for g_ in gs]
            out = ops.add_n(gs[:8])
            for i in range(8, len(gs), 7):
                out = ops.add_n([out] + gs[i:i + 7])
        for rows, g in sparse:
            if out.is_cuda:
                ops.scatter_add_rows_(out, rows.contiguous(), g)          # (kernel T10; torch's index_add_ until round 6)
            else:
                out.index_add_(0, rows, g)
        return out, None, None


def _select_rows(feats, rows):
    """feats[rows] for int64 row numbers: the library's index_select on the CPU seam, on the device the row-gather kernel K6 (one cloud of R rows: ids = rows)"""
    if not feats.is_cuda or feats.stride(1) != 1:
        return feats.index_select(0, rows)
    return ops.gather_rows(feats, feats.stride(0), 1, feats.shape[0], feats.shape[1], rows.to(torch.int32)[None].contiguous())[0]


class _SelectRows(torch.autograd.Function):
    """feats[rows] with a dense gradient map (a gather whose source is no fan-out handle): zero map + the row scatter-add kernel"""

    @staticmethod
    def forward(ctx, feats, rows):
        ctx.rows, ctx.shape = rows, tuple(feats.shape)
        return _select_rows(feats, rows)

    @staticmethod
    def backward(ctx, g):
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        if out.is_cuda:
            ops.scatter_add_rows_(out, ctx.rows.contiguous(), g.contiguous())
        else:
            out.index_add_(0, ctx.rows, g)
        return out, None


class _GatherRows(torch.autograd.Function):
    """feats.index_select(0, rows) for a handle of a _Fanout: the backward hands (rows, gradient rows) to the fan-out's sum instead of a dense map."""

    @staticmethod
    def forward(ctx, feats, rows, stash):
        ctx.rows, ctx.stash = rows, stash
        return _select_rows(feats, rows)

    @staticmethod
    def backward(ctx, g):
        ctx.stash.append((ctx.rows, g.contiguous()))
        return None, None, None


class _L2Norm(torch.autograd.Function):
    """F.normalize over channels (models/gmmreg.py:74): ogmm_l2norm_rows / ogmm_l2norm_rows_bwd"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.l2norm_rows(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.l2norm_rows_bwd(x, g.contiguous())


class _FeatMean(torch.autograd.Function):
    """mu_f = gamma^T f / (pi N + 1e-5) (lib/utils.py:130-140): ogmm_gmm_feat_mean forward; the gradient reaches f only
    (gamma, pi come out of the no-grad E/M loop): df = gamma (dmu / (pi N + 1e-5)), J-term rows on kernel T10."""

    @staticmethod
    def forward(ctx, gamma, pi, f, C, N):
        ctx.save_for_backward(gamma, pi)
        ctx.N = N
        return ops.gmm_feat_mean(gamma.contiguous(), pi.contiguous(), f.contiguous(), C, N)

    @staticmethod
    def backward(ctx, dmu):
        gamma, pi = ctx.saved_tensors
        df = ops.small_bmm_nn(gamma, (dmu / (pi * ctx.N + 1e-5)[:, :, None]).contiguous())          # (kernel T10: J-term rows; torch.bmm until round 6)
        return None, None, df.reshape(-1, dmu.shape[2]), None, None


def _attention_torch(q, k, v, C, N, M, H):
    D = q.shape[1]
    dh = D // H
    qh = q.view(C, N, H, dh).transpose(1, 2)
    kh = k.view(C, M, H, dh).transpose(1, 2)
    vh = v.view(C, M, H, dh).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(2, 3) / dh ** .5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(C * N, D)


FUSED_ATTENTION_BWD = True      # 0: the library path, for A/B timing


class _Attention(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v (models/attn.py:78-82): forward on the fused attention kernel, backward on ogmm_attention_bwd
    (kernel T11: the scores are re-formed per query tile on chip in both directions and never reach HBM).  Shapes the backward
    kernel is not built for (M != 128) re-form the scores with batched library GEMMs + softmax and differentiate those."""

    @staticmethod
    def forward(ctx, q, k, v, C, N, M, H, precision="f32", overflow=None):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.save_for_backward(q, k, v)
        ctx.dims = (C, N, M, H)
        ctx.split, ctx.overflow = precision == "f16x3", overflow
        return ops.attention(q, k, v, C, N, M, H)

    @staticmethod
    def backward(ctx, g):
        C, N, M, H = ctx.dims
        if FUSED_ATTENTION_BWD and ops.attention_bwd_supported(M, ctx.saved_tensors[0].shape[1] // H):
            q, k, v = ctx.saved_tensors
            # (round 5: in the fp16x3 step the two products over the head dimension run on the engines' arithmetic, ogmm_attention_bwd_f16x3)
            return ops.attention_bwd(q, k, v, _rm(g), C, N, M, H, split=ctx.split, overflow=ctx.overflow) + (None,) * 6
        q, k, v = (t_.detach().requires_grad_(True) for t_ in ctx.saved_tensors)
        with torch.enable_grad():
            o = _attention_torch(q, k, v, *ctx.dims)
        gq, gk, gv = torch.autograd.grad(o, (q, k, v), g)
        return gq, gk, gv, None, None, None, None, None, None


class _OverlapCross(torch.autograd.Function):
    """models/gmmreg.py:75-80: N x N similarity on the GEMM engine, row / column softmax-dots (kernels K13 + T8); backward: one
    pass over S gives dS and the logit gradients, dL/dfn = (dS fn_tgt, dS^T fn_src) are two batched library GEMMs."""

    @staticmethod
    def forward(ctx, fn, ol, B, N, precision, overflow):
        fn, ol = fn.contiguous(), ol.contiguous()
        S = ops.similarity(fn, B, N, split=precision == "f16x3", overflow=overflow)
        wo, stats = ops.overlap_cross_train(S, ol)
        ctx.save_for_backward(S, fn, ol, wo, stats)
        ctx.B, ctx.N = B, N
        ctx.engine, ctx.overflow = precision == "f16x3", overflow
        return wo

    @staticmethod
    def backward(ctx, g_wo):
        S, fn, ol, wo, stats = ctx.saved_tensors
        B, N = ctx.B, ctx.N
        D = fn.shape[1]
        if ctx.engine and N % 64 == 0 and D % 64 == 0:
            # both products on the fp16x3 engine (round 4; two batched fp32 library GEMMs before: 2.2 ms of the 128-pair step): dS[b] fn_tgt[b] reads dS as
            # it lies against a per-batch image of fn_tgt^T; dS[b]^T fn_src[b] is the weight gradient's dY^T X form with one row chunk per pair, un-summed.
            # dS is the engine's fp32 operand here, split into binary16 terms on the fly, and its entries are softmax gradients of order |g| / N: far into
            # binary16's subnormals unless scaled.  dS is linear in g_wo, so g_wo is multiplied by a power of two (|dS| <= 2 max|g_wo| -> at most 2^14; found on
            # the device, no host round trip) and the products and the logit gradient are multiplied back by its inverse: exact either way.
            k2 = torch.floor(13.0 - torch.log2(g_wo.detach().abs().amax().clamp_min(1e-30).float())).clamp_(-40.0, 40.0)
            up, down = torch.exp2(k2), torch.exp2(-k2)
            dS, g_ol = ops.overlap_cross_bwd(S, ol, wo, stats, g_wo * up)
            inv = down.expand(D).contiguous()
            g_fn = torch.empty((2 * B * N, D), dtype=torch.float32, device=fn.device)          # both halves written in place: no concatenation pass
            ops.batched_a_times_x(dS, fn[B * N:], ctx.overflow, out_scale=inv, out=g_fn[:B * N])
            ops.weight_grad(dS.view(B * N, N), [fn[:B * N]], ctx.overflow, chunk_rows=N, keep_parts=True, out_scale=inv, parts_out=g_fn[B * N:].view(B, N, D))
            return g_fn, g_ol * down, None, None, None, None
        dS, g_ol = ops.overlap_cross_bwd(S, ol, wo, stats, g_wo)
        fs, ft = fn[:B * N].view(B, N, -1), fn[B * N:].view(B, N, -1)
        g_fn = torch.cat([torch.bmm(dS, ft), torch.bmm(dS.transpose(1, 2), fs)], dim=0).view(2 * B * N, -1)
        return g_fn, g_ol, None, None, None, None


class _Kabsch(torch.autograd.Function):
    """lib/se3.py:256-289 on the fp64 in-register 3x3 solver (ogmm_kabsch) with its closed-form backward (ogmm_kabsch_bwd).
    Tensors in the kernels' layout: src, corr [B,3,J], w [B,J]."""

    @staticmethod
    def forward(ctx, src, corr, w):
        ctx.save_for_backward(src, corr, w)
        R, t = ops.kabsch(src, corr, w)
        return R, t.reshape(-1, 3)

    @staticmethod
    def backward(ctx, gR, gt):
        src, corr, w = ctx.saved_tensors
        return ops.kabsch_bwd(src, corr, w, gR, gt)


class _RotationFromCov(torch.autograd.Function):
    """R = V diag(1, 1, det(V U^T)) U^T of M = U S V^T (baseline/deepgmr.py:28-34) on ogmm_rotation_from_cov.  Backward: M is the covariance of a
    four-point fit -- src = (e0, e1, e2, -(e0+e1+e2)) has centroid 0, so sum_j (src_j - 0)(corr_j - c)^T = sum_j src_j corr_j^T = M for corr_j = M[j, :]
    (j < 3), corr_3 = 0 -- so dL/dM[j, :] is ogmm_kabsch_bwd's gradient for corr_j: the closed-form derivative through the 3x3 SVD (fp64, finite for equal
    singular values), shared with the main head.  (That kernel's covariance carries the reference's + 1e-5 I of lib/se3.py:281; it is taken off M first.)"""

    @staticmethod
    def forward(ctx, M):
        ctx.save_for_backward(M)
        return ops.rotation_from_cov(M)

    @staticmethod
    def backward(ctx, gR):
        (M,) = ctx.saved_tensors
        B = M.shape[0]
        eye = torch.eye(3, dtype=M.dtype, device=M.device)
        src = torch.cat([eye, -torch.ones(3, 1, dtype=M.dtype, device=M.device)], dim=1).expand(B, 3, 4).contiguous()       # [B,3,J=4], points as columns
        corr = torch.cat([(M - 1e-5 * eye).transpose(1, 2), M.new_zeros(B, 3, 1)], dim=2).contiguous()
        w = M.new_ones(B, 4)
        _, g_corr, _ = ops.kabsch_bwd(src, corr, w, gR.contiguous(), torch.zeros(B, 3, dtype=M.dtype, device=M.device))
        return g_corr[:, :, :3].transpose(1, 2).contiguous()


class TrainOps:
    """precision: "f16x3" (dense forward layers on the split-binary16 matrix-core engine) or "f32" (exact-fp32 engine)."""

    def __init__(self, precision="f16x3", overflow=None, status=None):
        if precision not in ("f16x3", "f32"):
            raise ValueError("precision must be 'f16x3' or 'f32'")
        self.precision, self.overflow, self.status = precision, overflow, status          # status: the model's protocol-error word (gmmreg._flags[1])

    # ------------------------------------------------------------------ selections (no gradient)
    def knn(self, xyz, k):
        return ops.knn(xyz, k).long()

    def fps(self, xyz, npoint, starts):
        if starts is None:
            return ops.fps(xyz, npoint, None).long()
        return ops.fps(xyz, npoint, starts.to(device=xyz.device, dtype=torch.int32).contiguous()).long()

    def gmm_em(self, xyz, o, ids_j):
        # (src clouds | tgt clouds: two wkeans_plus calls in the reference, i.e. two call groups of the Sinkhorn early exit)
        return ops.gmm_em(xyz, o.contiguous(), ids_j.to(torch.int32).contiguous(), iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=1e-2,
                          group_size=xyz.shape[0] // 2, status=self.status)

    def nearest_point(self, xyz, mu):
        """index of the point nearest to each mu (lib/utils.py:244-254) -> [C,J] int64"""
        return ops.nearest_point(xyz, mu).long()

    # ------------------------------------------------------------------ constants of the input
    def edge_features(self, xyz, idx):
        """lib/utils.py:47-66: [x_j - x_i ; x_i] per edge -> [C*N*k, 6]"""
        return ops.edge_features(xyz, idx.to(torch.int32).contiguous())

    def pos_features(self, xyz, idx5):
        """models/attn.py:60-70: squared distance to the cloud centroid [C*N,1]; cosine between each 5-NN offset and the
        centroid offset [C*N*5,1] (the self neighbour gives a zero vector, hence cosine 0)."""
        return ops.pos_features(xyz, idx5.to(torch.int32).contiguous(), xyz.mean(dim=1))

    # ------------------------------------------------------------------ dense layers
    def linear(self, x, W, b, x2=None):
        """y = [x | x2] W^T + b.   x [R,K1], x2 [R,K2] or None, W [Cout, K1+K2].  Layers too thin for a matrix-core tile
        (the 6->64 edge layer, the 1->64 positional layers, the Cout=1 heads) are HBM-bound outer products / row dots."""
        K1, Cout = x.shape[1], W.shape[0]
        if K1 < 32 or Cout < 32 or (K1 % 64 and x2 is not None):
            return _ThinLinear.apply(x if x2 is None else torch.cat([x, x2], dim=1), W, b)
        return _Linear.apply(x, x2, W, b, self.precision, self.overflow)

    def linear_stats(self, x, W, b, x2=None, groups=1):
        """linear() for a layer that feeds a normalisation over `groups` equal row blocks: also returns the fp64 column sums
        [groups, Cout, 2] = {sum, sum of squares} (accumulated by the GEMM epilogue when the shape allows it)."""
        K1, Cout = x.shape[1], W.shape[0]
        rows = x.shape[0] // groups
        if K1 < 32 or Cout < 32 or (K1 % 64 and x2 is not None):
            y = self.linear(x, W, b, x2)
            return y, ops.colstats(y.contiguous(), rows)
        return _Linear.apply(x, x2, W, b, self.precision, self.overflow, rows)

    def batchnorm_act(self, y, weight, bias, running_mean, running_var, num_batches, groups, act, stats=None):
        """Train-mode BatchNorm over each of the `groups` equal row blocks of y (one block per call of the reference's
        shared layer), then ReLU or LeakyReLU(0.2).  Running statistics are updated in place, block 0 first."""
        n = y.shape[0] // groups
        h, mean64, var64 = _NormAct.apply(y, weight, bias, n, _ACT[act], stats)
        self._update_running(running_mean, running_var, num_batches, mean64, var64, n, groups)
        return h

    def batchnorm_act_pool(self, y, weight, bias, running_mean, running_var, num_batches, groups, act, k, want_h, stats=None):
        """batchnorm_act + max over the k rows of every point: -> (h or None, pooled [rows/k, c])"""
        n = y.shape[0] // groups
        if y.shape[1] % 4:
            h = self.batchnorm_act(y, weight, bias, running_mean, running_var, num_batches, groups, act, stats)
            return (h if want_h else None), self.maxpool_k(h, k)
        h, pooled, mean64, var64 = _NormActPool.apply(y, weight, bias, n, _ACT[act], k, want_h, stats)
        self._update_running(running_mean, running_var, num_batches, mean64, var64, n, groups)
        return (h if want_h else None), pooled

    def _norm_linear_fusable(self, y, group_rows, W):
        K, Cout = y.shape[1], W.shape[0]
        return (FUSE_NORM_LINEAR and self.precision == "f16x3" and group_rows % 256 == 0 and K % 32 == 0 and 256 <= K <= 4096 and Cout >= 256
                and Cout % 4 == 0)

    def _update_running(self, running_mean, running_var, num_batches, mean64, var64, n, groups):
        """torch.nn.BatchNorm1d's buffer update for the `groups` sequential calls of the shared layer (models/dgcnn.py:126-130): one launch"""
        assert mean64.shape[0] == groups
        with torch.no_grad():
            if (running_mean.dtype == torch.float32 and running_mean.is_contiguous() and running_var.is_contiguous() and mean64.is_contiguous()
                    and var64.is_contiguous() and num_batches.dtype == torch.int64):
                ops.bn_update_running(mean64, var64, n, BN_MOMENTUM, running_mean, running_var, num_batches)
                return
            unbiased = var64 * (n / max(n - 1, 1))
            for g in range(groups):
                running_mean.mul_(1 - BN_MOMENTUM).add_(mean64[g].to(running_mean.dtype), alpha=BN_MOMENTUM)
                running_var.mul_(1 - BN_MOMENTUM).add_(unbiased[g].to(running_var.dtype), alpha=BN_MOMENTUM)
            num_batches += groups

    def batchnorm_relu_linear(self, y, stats, weight, bias, running_mean, running_var, num_batches, groups, W, b, want_stats=False):
        """linear(batchnorm_act(y, ..., "relu"), W, b) [-> (out, column sums of out) with want_stats]; where the engine takes it the normalised
        map is never written (_NormLinear)."""
        n = y.shape[0] // groups
        if self._norm_linear_fusable(y, n, W):
            res = _NormLinear.apply(y, stats, weight, bias, n, W, b, self.overflow, n if want_stats else 0)
            self._update_running(running_mean, running_var, num_batches, res[1], res[2], n, groups)
            return (res[0], res[3]) if want_stats else res[0]
        h = self.batchnorm_act(y, weight, bias, running_mean, running_var, num_batches, groups, "relu", stats=stats)
        return self.linear_stats(h, W, b, groups=groups) if want_stats else self.linear(h, W, b)

    def instnorm_relu_linear(self, z, C, N, stats, W, b, res=None):
        """linear(instnorm_relu(z), W, b) [+ res]"""
        if self._norm_linear_fusable(z, N, W):
            return _NormLinear.apply(z, stats, None, None, N, W, b, self.overflow, 0, res)[0]
        out = self.linear(self.instnorm_relu(z, C, N, stats=stats), W, b)
        return out if res is None else out + res

    def fanout(self, x, n):
        """n handles of x, one per consumer (see _Fanout)"""
        if not x.requires_grad:
            return (x,) * n
        stash = [] if GATHER_SPARSE else None
        outs = _Fanout.apply(x, n, stash)
        if stash is not None:
            for o in outs:
                o._ogmm_stash = stash          # gather_points() on a handle leaves its gradient rows here
        return outs

    def instnorm_relu(self, z, C, N, stats=None):
        """InstanceNorm1d (no affine, biased variance, eps 1e-5) over the N points of each cloud, then ReLU"""
        return _NormAct.apply(z, None, None, N, ops.ACT_RELU, stats)[0]

    def maxpool_k(self, h, k):
        """max over the k consecutive rows of each point: [P*k, c] -> [P, c]"""
        return _MaxPoolK.apply(h, k)

    def gather_points(self, feats, C, N, ids, cloud_map=None):
        """rows ids[c, s] of cloud map(c) -> [C*S, D]  (lib/utils.py:111-127)"""
        S = ids.shape[1]
        clouds = torch.arange(C, device=feats.device) if cloud_map is None else cloud_map
        rows = (clouds[:, None] * N + ids[clouds]).reshape(C * S)
        stash = getattr(feats, "_ogmm_stash", None)
        if stash is not None and feats.requires_grad:
            return _GatherRows.apply(feats, rows, stash)
        return _SelectRows.apply(feats, rows) if feats.requires_grad else _select_rows(feats, rows)

    def attention(self, q, k, v, C, N, M, H):
        """softmax(q k^T / sqrt(dh)) v per cloud and head; head-major channels.  q [C*N,D], k, v [C*M,D] -> [C*N,D]"""
        if ops.attention_supported(M, q.shape[1] // H):
            return _Attention.apply(q, k, v, C, N, M, H, self.precision, self.overflow)
        return _attention_torch(q, k, v, C, N, M, H)

    def l2norm_rows(self, f):
        return _L2Norm.apply(f)

    def overlap_cross(self, fn, ol, B, N):
        """models/gmmreg.py:75-80, literally: with S[b,m,n] = <fn_src[b,m], fn_tgt[b,n]>,
        wo_src[b,m] = sum_n softmax_n(S[b,m,:])[n] * ol_src[b,n]   (the src logits indexed along the tgt axis) and
        wo_tgt[b,n] = sum_m softmax_m(S[b,:,n])[m] * ol_tgt[b,m].   fn [2B*N, D], ol [2B*N, 1] -> [2B*N, 1]"""
        return _OverlapCross.apply(fn, ol, B, N, self.precision, self.overflow)

    # ------------------------------------------------------------------ GMM head
    def gmm_feat_mean(self, gamma, pi, f, C, N):
        """lib/utils.py:130-140 on features: mu_f = gamma^T f / (pi N + 1e-5)  -> [C,J,D]"""
        return _FeatMean.apply(gamma, pi, f, C, N)

    def match_kabsch(self, mu_s, mu_t, f_s, f_t, temperature):
        """models/dgcnn.py:96-115 + lib/se3.py:256-289"""
        if f_s.is_cuda and f_s.shape[1] <= 128:          # kernel T10 (torch.matmul until round 6)
            sim = small_bmm_nt(F.normalize(f_s, dim=-1), F.normalize(f_t, dim=-1))
            sc = torch.softmax(sim / temperature, dim=2)                # [B,J,J]
            corr = small_bmm(sc, mu_t.contiguous())                     # [B,J,3]
        else:
            sim = F.normalize(f_s, dim=-1) @ F.normalize(f_t, dim=-1).transpose(1, 2)
            sc = torch.softmax(sim / temperature, dim=2)
            corr = sc @ mu_t
        w = sc.sum(dim=2)                                               # == 1 up to rounding
        return self.kabsch(mu_s, corr, w)

    def kabsch(self, src, corr, w):
        """weighted rigid fit src -> corr, points as rows: src, corr [B,J,3], w [B,J] -> R [B,3,3], t [B,3]"""
        return _Kabsch.apply(src.transpose(1, 2).contiguous(), corr.transpose(1, 2).contiguous(), w.contiguous())

    def rotation_from_cov(self, M):
        """baseline/deepgmr.py:28-34: M [B,3,3] -> R [B,3,3], differentiable"""
        return _RotationFromCov.apply(M.contiguous())
