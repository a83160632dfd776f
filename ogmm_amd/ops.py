"""Thin torch-tensor wrappers over the C ABI (include/ogmm_hip.h).  PyTorch is plumbing here: it owns the
device memory and the stream; every function enqueues HIP kernels of libogmm_hip.so on torch's current
stream and returns without synchronising.  All tensors must be CUDA (ROCm) tensors -- there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SIGMOID, PREC_F16_FRAG, PREC_F16X3, PREC_F16X3_FRAG, PREC_F32, GemmDesc  # noqa: F401


# bench.py sets this to a list to time the GEMM-engine launches with events on the launch stream:
# entries are (start_event, end_event, algorithmic_flops)
GEMM_TIMELINE = None
# Alternatives kept for A/B measurements (tools/*.py and the tests assign these module attributes; nothing reads the environment): every default below is the
# product path, every other value the form it replaced -- bit-identical where the comment says so.
NORM_BWD_FUSED = True          # training: the normalisation backward's reduction in the dh GEMM's epilogue (A/B switch)
FUSE_GATHER = True      # anchor rows gathered by the consuming GEMM's operand DMA (conv1x1_gathered)
EDGECONV_PC = True      # the EdgeConv chain as a producer / consumer pipeline (k = 20); False: the barrier-phased kernel
FUSE_HEAD = True      # Cout = 1 heads in the producing layer's epilogue (conv1x1_head)
GEMM_TIMELINE_ONLY = None      # optional set of variant tags: only those launches are bracketed by events (bench.py: the dominant engine only)
KERNEL_TIMELINE = None         # bench.py: a list -> the EdgeConv and attention launches are bracketed too: (start_event, end_event, name, algorithmic flops, algorithmic bytes)
_EVENT_POOL = []               # timing events are recycled: creating two torch events per launch costs more host time than the launch itself


def _timing_event():
    return _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)


def recycle_timing_events(timeline):
    """hand the events of a consumed GEMM_TIMELINE back to the pool"""
    for e0, e1, *_ in timeline:
        _EVENT_POOL.append(e0); _EVENT_POOL.append(e1)


def _timed_call(name, flops, nbytes, fn_name, *args):
    """_lib.call, bracketed by events when bench.py collects KERNEL_TIMELINE"""
    if KERNEL_TIMELINE is None:
        _lib.call(fn_name, *args)
        return
    e0, e1 = _timing_event(), _timing_event()
    e0.record()
    _lib.call(fn_name, *args)
    e1.record()
    KERNEL_TIMELINE.append((e0, e1, name, flops, nbytes))

class Engine:
    """Which GEMM engine the layers of ONE model run on, and where that model's binary16-overflow flag lives.  Every model owns one and passes it
    down explicitly (`eng=`): two models with different `precision` in one process -- or on two threads -- do not see each other's choice.
      split        layers that carry pre-split weights run on the fp16x3 matrix-core engine (False: exact-fp32 engine)
      single_term  precision "f16": the large-shape engine multiplies only the leading binary16 terms (REDUCED precision)
      overflow     device int32[1] or None: set non-zero by the fp16 engines when |activation| > 65504 was clamped"""
    __slots__ = ("split", "single_term", "overflow")

    def __init__(self, precision="f16x3", overflow=None):
        if precision not in ("f16x3", "f32", "f16"):
            raise _lib.OgmmError("precision must be 'f16x3', 'f32' or 'f16' (reduced: single binary16 term in the large GEMMs)")
        self.split = precision in ("f16x3", "f16")
        self.single_term = precision == "f16"
        self.overflow = overflow


DEFAULT_ENGINE = Engine()      # for direct calls of the functions below (tests, tools); models never modify it


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.OgmmError("%s must be a CUDA/ROCm tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise _lib.OgmmError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _f32(t, name):
    return _chk(t, torch.float32, name)


def _i32(t, name):
    return _chk(t, torch.int32, name)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---------------------------------------------------------------------------------------------- selection
KNN_HEAD = True          # A/B switch: 0 = ogmm_knn (k = 20), ogmm_knn (k = 5) and ogmm_pos_hidden as three launches


def pack_clouds(src, tgt):
    """src, tgt [B,3,N] (models/gmmreg.py:50) -> xyz [2B,N,3] (src clouds, then tgt clouds): one launch instead of torch's cat + transpose copy."""
    B, _, N = src.shape
    src, tgt = _f32(src, "src").contiguous(), _f32(tgt, "tgt").contiguous()
    xyz = torch.empty((2 * B, N, 3), dtype=torch.float32, device=src.device)
    _lib.call("ogmm_pack_clouds", _p(src), _p(tgt), B, N, _p(xyz), _stream())
    return xyz


def knn(xyz, k):
    """xyz [C,N,3] -> idx [C,N,k] int32   (lib/utils.py:37-44)."""
    xyz = _f32(xyz, "xyz")
    assert xyz.is_contiguous() and xyz.dim() == 3 and xyz.shape[2] == 3
    C, N, _ = xyz.shape
    idx = torch.empty((C, N, k), dtype=torch.int32, device=xyz.device)
    _lib.call("ogmm_knn", _p(xyz), C, N, k, _p(idx), _stream())
    return idx


def knn_pos_head_supported(N, k):
    return KNN_HEAD and _lib.load().ogmm_knn_pos_head_supported(N, k) == 1


def knn_pos_head(xyz, k, pos=None):
    """The forward's head in one launch (include/ogmm_hip.h: ogmm_knn_pos_head): idx [C,N,k] as knn(xyz, k) and -- with pos = the packed "pos" layer
    (w_dis, s_dis, t_dis, w_ang, s_ang, t_ang) -- idx5 [C,N,5] as knn(xyz, 5) and (hid_dis, hid_ang) [C*N,64] as pos_hidden(xyz, idx5, 5, pos).
    Returns idx or (idx, idx5, hid_dis, hid_ang)."""
    xyz = _f32(xyz, "xyz")
    assert xyz.is_contiguous() and xyz.dim() == 3 and xyz.shape[2] == 3
    C, N, _ = xyz.shape
    dev = xyz.device
    idx = torch.empty((C, N, k), dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.load().ogmm_knn_pos_head_workspace_bytes(C, N), dtype=torch.uint8, device=dev)
    if pos is None:
        _lib.call("ogmm_knn_pos_head", _p(xyz), C, N, k, _p(idx), None, None, None, None, None, None, None, None, None, _p(ws), _stream())
        return idx
    idx5 = torch.empty((C, N, 5), dtype=torch.int32, device=dev)
    hd = torch.empty((C * N, 64), dtype=torch.float32, device=dev)
    ha = torch.empty_like(hd)
    _lib.call("ogmm_knn_pos_head", _p(xyz), C, N, k, _p(idx), _p(idx5), _p(pos["w_dis"]), _p(pos["s_dis"]), _p(pos["t_dis"]), _p(pos["w_ang"]), _p(pos["s_ang"]),
              _p(pos["t_ang"]), _p(hd), _p(ha), _p(ws), _stream())
    return idx, idx5, hd, ha


def fps(xyz, npoint, start=None):
    """xyz [C,N,3]; start None (centre start) -> ids [C,npoint]; start [S,C] int32 -> ids [S,C,npoint]
    (lib/utils.py:170-198)."""
    xyz = _f32(xyz, "xyz")
    assert xyz.is_contiguous()
    C, N, _ = xyz.shape
    if start is None:
        ids = torch.empty((C, npoint), dtype=torch.int32, device=xyz.device)
        _lib.call("ogmm_fps", _p(xyz), C, N, npoint, 1, None, _p(ids), _stream())
        return ids
    start = _i32(start, "start")
    assert start.is_contiguous() and start.dim() == 2 and start.shape[1] == C
    S = start.shape[0]
    ids = torch.empty((S, C, npoint), dtype=torch.int32, device=xyz.device)
    _lib.call("ogmm_fps", _p(xyz), C, N, npoint, S, _p(start), _p(ids), _stream())
    return ids


def gather_rows(feats, ld, C, N, D, ids, cloud_map=None):
    """feats rows [(C*N), ld] -> [C,S,D] with out[c,s] = feats[map(c)*N + ids[map(c),s]]   (lib/utils.py:111-127)."""
    feats, ids = _f32(feats, "feats"), _i32(ids, "ids")
    assert ids.is_contiguous() and ids.shape[0] == C
    S = ids.shape[1]
    out = torch.empty((C, S, D), dtype=torch.float32, device=feats.device)
    if cloud_map is not None:
        cloud_map = _i32(cloud_map, "cloud_map")
    _lib.call("ogmm_gather_rows", _p(feats), ld, C, N, D, _p(ids), S, _p(cloud_map), _p(out), _stream())
    return out


# ---------------------------------------------------------------------------------------------- GEMM engine
def split_f16(W, pad_to=8, frag=False, k1=None, exp=None, scale_t=None):
    """fp32 [N,K] -> dict(W_hi, W_lo binary16, inv_scale): W * 2^e = hi + lo with the power of two chosen so
    that max|W| * 2^e is in [2^11, 2^12) (keeps `lo` a normal binary16 number); inv_scale = 2^-e goes into alpha.
    frag=False: row-major [N, Kpad8] planes (OGMM_PREC_F16X3).  frag=True: the fragment-major image of
    OGMM_PREC_F16X3_FRAG ([Npad256/32][Kpad32/16][64][8]); k1 = length of the first A piece when the input is a
    channel concatenation (the second piece is then moved to start at a multiple of 32)."""
    import math
    W = W.float()
    if frag:
        N, K = W.shape
        k1 = K if k1 is None else k1
        assert k1 % 64 == 0 or k1 == K
        k2 = K - k1
        k1p, k2p = (k1 + 63) // 64 * 64, (k2 + 63) // 64 * 64      # K tiles of 64 (v3 engine); v2 walks them as 2 x 32
        if N % 256 == 0 and k2 == 0 and K == k1p:
            Wp = W          # already whole tiles: no padded copy (two launches per split less: the training step re-splits ~120 weights)
        else:
            Wp = W.new_zeros((N + 255) // 256 * 256, k1p + k2p)
            Wp[:N, :k1] = W[:, :k1]
            if k2:
                Wp[:N, k1p:k1p + k2] = W[:, k1:]
        planes = split_f16(Wp, pad_to=64, exp=exp, scale_t=scale_t)
        Np, Kp = Wp.shape

        def image(P):      # [Np, Kp] -> [Np/32][Kp/16][lane = g*32 + r][8]
            return P.view(Np // 32, 32, Kp // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
        return {"W_hi": image(planes["W_hi"]), "W_lo": image(planes["W_lo"]), "inv_scale": planes["inv_scale"],
                "variant": PREC_F16X3_FRAG, "ldb_h": Kp}
    # power of two on the DEVICE (no .item(): the training path splits ~60 weights per step): e = 11 - floor(log2(max|W|)), clamped to +-24;
    # the scale itself stays a device scalar there, and `inv_scale` is a python float only for the pack-once inference path (one sync per pack)
    if exp is not None or scale_t is not None:          # caller-supplied scale: a python exponent, or a device scalar 2^e (split_f16_training: no host sync)
        K = W.shape[1]
        Kp = (K + pad_to - 1) // pad_to * pad_to
        exp = 0 if exp is None else exp
        Ws = W * scale_t if scale_t is not None else W * (2.0 ** exp)
        if Kp != K:
            Ws = torch.cat([Ws, Ws.new_zeros(W.shape[0], Kp - K)], dim=1)
        hi = Ws.half()
        lo = (Ws - hi.float()).half()
        return {"W_hi": hi.contiguous(), "W_lo": lo.contiguous(), "inv_scale": 2.0 ** (-exp)}
    amax = W.abs().max()
    ok = torch.isfinite(amax) & (amax > 0)
    e_t = torch.where(ok, 11.0 - torch.floor(torch.log2(torch.where(ok, amax, torch.ones_like(amax)))), torch.zeros_like(amax)).clamp_(-24.0, 24.0)
    scale_t = torch.exp2(e_t)
    Ws = W * scale_t
    K = W.shape[1]
    Kp = (K + pad_to - 1) // pad_to * pad_to
    if Kp != K:
        Ws = torch.cat([Ws, Ws.new_zeros(W.shape[0], Kp - K)], dim=1)
    hi = Ws.half()
    lo = (Ws - hi.float()).half()
    e = float(e_t)
    return {"W_hi": hi.contiguous(), "W_lo": lo.contiguous(), "inv_scale": 2.0 ** (-e)}


_SPLIT_SLOTS = {}          # device -> [pool float32 [4096, 4] (zero), next slot]: the scale kernel's scratch, left zero by every call (ogmm_split_weight)
SPLIT_WEIGHT_FUSED = True      # 0: rounds 1-3's path (tensor expressions + ogmm_pack_frag), for A/B timing


def _split_slot(device):
    ent = _SPLIT_SLOTS.get(device)
    if ent is None:
        ent = _SPLIT_SLOTS[device] = [torch.zeros((4096, 4), dtype=torch.float32, device=device), 0]
    slot = ent[0][ent[1] % 4096]
    ent[1] += 1
    return slot


def split_f16_training(W, cout, transpose=False, **kw):
    """split_f16(frag=True) for weights that change every step (the trainer re-splits ~120 of them per step): the power-of-two scale is found on the DEVICE
    (no host synchronisation, nothing cached that could go stale or be keyed on a recycled address), leaving one binade of headroom, and the fragment image
    is written straight from the weight -- two launches per split (ogmm_split_weight).  transpose=True: the image of W^T (the operand of dX = dY W) from W as it
    lies; `cout` is then the number of the GEMM's output columns = W.shape[1] (rounded up by the caller if it pads).  The inverse scale cannot ride in the
    host-side `alpha`, so it comes back as `col_scale` [cout], the GEMM's per-column scale (struct ogmm_gemm.scale); `inv_scale` is 1."""
    W = W.float().contiguous()
    k1 = kw.get("k1")
    if SPLIT_WEIGHT_FUSED and kw.get("frag"):
        rows, cols = W.shape
        N, K = (cols, rows) if transpose else (rows, cols)
        k1 = K if (transpose or k1 is None) else k1
        # (split_weight_kernel pads piece one to a multiple of 64 itself -- k1p -- so a two-piece layer whose first piece is not one needs no other path)
        ldb_h = (k1 + 63) // 64 * 64 + (K - k1 + 63) // 64 * 64
        n_pad = (N + 255) // 256 * 256
        hi = torch.empty(n_pad * ldb_h, dtype=torch.float16, device=W.device)
        lo = torch.empty_like(hi)
        inv = torch.empty(cout, dtype=torch.float32, device=W.device)
        _lib.call("ogmm_split_weight", _p(W), cols, rows, cols, 1 if transpose else 0, k1, _p(_split_slot(W.device)), _p(inv), cout, _p(hi), _p(lo), ldb_h, n_pad,
                  _stream())
        return {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": PREC_F16X3_FRAG, "ldb_h": ldb_h, "col_scale": inv}
    if transpose:
        W = W.t().contiguous()
    sc = torch.zeros(4, dtype=torch.float32, device=W.device)          # [0] the scale; [2], [3] the grid-wide reduction's scratch (zero on entry)
    inv = torch.empty(cout, dtype=torch.float32, device=W.device)
    _lib.call("ogmm_pow2_scale", _p(W), W.numel(), 10, _p(sc), _p(inv), cout, _stream())
    N, K = W.shape
    if kw.get("frag") and N % 256 == 0 and K % 64 == 0 and (k1 is None or k1 == K or k1 % 64 == 0):
        # whole tiles and no padding between the A pieces: the split fragment images straight from the scaled weight with the activation packer
        Ws = W * sc[0:1]
        hi = torch.empty(N * K, dtype=torch.float16, device=W.device)
        lo = torch.empty_like(hi)
        _lib.call("ogmm_pack_frag", _p(Ws), K, N, K, _p(hi), _p(lo), _stream())
        return {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": PREC_F16X3_FRAG, "ldb_h": K, "col_scale": inv}
    sp = split_f16(W, scale_t=sc[0:1], **kw)
    sp["col_scale"] = inv
    return sp


def gemm_nt(A, lda, K1, B, ldb, M, N, C=None, ldc=0, A2=None, lda2=0, K2=0, scale=None, shift=None, row_affine=False,
            alpha=1.0, act=ACT_NONE, res=None, ldr=0, batch=(1, 1), sA=(0, 0), sA2=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0),
            pool_k=0, pool_out=None, ldp=0, store_c=True, split=None, overflow=None, col_stats=None, a_affine=None, group_rows=0,
            overlap=None, row_rscale=None, head=None, a_gather=None, single_term=False, terms=0, norm_bwd=None, a_trans=False, a_colsum=None):
    """Raw descriptor call; A, B, ... are tensors (only their data_ptr is used) -- see `struct ogmm_gemm`.
    split = dict from split_f16(B) selects the fp16x3 engine (B itself may then be None)."""
    d = GemmDesc()
    d.A, d.lda, d.K1 = A.data_ptr(), lda, K1
    d.A2, d.lda2, d.K2 = (A2.data_ptr() if A2 is not None else None), lda2, K2
    d.B, d.ldb = (B.data_ptr() if B is not None else None), ldb
    if split is not None:
        d.precision = split.get("variant", PREC_F16X3)
        if single_term and d.precision == PREC_F16X3_FRAG:
            d.precision = PREC_F16_FRAG
        d.B_hi, d.B_lo, d.ldb_h = split["W_hi"].data_ptr(), split["W_lo"].data_ptr(), split.get("ldb_h", split["W_hi"].shape[-1])
        d.overflow = overflow.data_ptr() if overflow is not None else None
        alpha = alpha * split["inv_scale"]
        if "sB" in split:
            sB = (split["sB"], 0)
    d.C, d.ldc = (C.data_ptr() if C is not None else None), ldc
    d.Res, d.ldr = (res.data_ptr() if res is not None else None), ldr
    d.M, d.N = M, N
    d.batch_outer, d.batch_inner = batch
    d.sA_o, d.sA_i = sA
    d.sA2_o, d.sA2_i = sA2
    d.sB_o, d.sB_i = sB
    d.sC_o, d.sC_i = sC
    d.sR_o, d.sR_i = sR
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.row_affine = 1 if row_affine else 0
    d.alpha = alpha
    d.act = act
    d.pool_k, d.pool_out, d.ldp, d.store_c = pool_k, (pool_out.data_ptr() if pool_out is not None else None), ldp, 1 if store_c else 0
    d.group_rows = group_rows
    d.a_trans = 1 if a_trans else 0          # A_mem is [K][M] row-major (struct ogmm_gemm.a_trans): the weight gradient's dY read as it lies
    if a_colsum is not None:                 # float64 [M], zeroed by the caller: += the column sums of A_mem (struct ogmm_gemm.a_colsum)
        assert a_trans and a_colsum.dtype == torch.float64 and a_colsum.is_contiguous() and a_colsum.numel() == M
        d.a_colsum = a_colsum.data_ptr()
    d.terms = terms          # per-layer term budget (struct ogmm_gemm.terms): 2 = the weight operand rounded to binary16 where the engine has the form
    if col_stats is not None:          # [G, N, 2] or, spread over 2^n copies that the caller sums, [2^n, G, N, 2] (struct ogmm_gemm.col_stats_slot_mask)
        d.col_stats = col_stats.data_ptr()
        if col_stats.dim() == 4:
            assert col_stats.is_contiguous() and col_stats.shape[0] & (col_stats.shape[0] - 1) == 0
            d.col_stats_slot_mask, d.col_stats_slot_stride = col_stats.shape[0] - 1, col_stats.stride(0)
    if a_affine is not None:
        d.a_scale, d.a_shift, d.a_relu = a_affine[0].data_ptr(), a_affine[1].data_ptr(), 1 if a_affine[2] else 0
    if norm_bwd is not None:          # (x, mean, rstd, scale, shift, act): struct ogmm_gemm.nb_* -- x travels as Res, the sums come out of col_stats
        xnb = norm_bwd[0]
        assert res is None and col_stats is not None and xnb.stride(1) == 1 and tuple(xnb.shape) == (M, N)
        d.Res, d.ldr = xnb.data_ptr(), xnb.stride(0)
        d.nb_mean, d.nb_rstd, d.nb_scale, d.nb_shift, d.nb_act = (norm_bwd[1].data_ptr(), norm_bwd[2].data_ptr(), norm_bwd[3].data_ptr(),
                                                                 norm_bwd[4].data_ptr(), norm_bwd[5])
    if overlap is not None:          # (o_row, o_col, ld, rowpart, colpart): the fused overlap block, S is not stored (struct ogmm_gemm)
        d.ovl_orow, d.ovl_ocol, d.ovl_ld = overlap[0].data_ptr(), overlap[1].data_ptr(), overlap[2]
        d.ovl_rowpart, d.ovl_colpart = overlap[3].data_ptr(), overlap[4].data_ptr()
        store_c = False
    if row_rscale is not None:
        d.row_rscale = row_rscale.data_ptr()
    if a_gather is not None:          # (ids int32 [C, S], cloud_map int32 [C] or None, N points per cloud, rows of A): struct ogmm_gemm.a_gather_*
        d.a_gather_ids, d.a_gather_map = a_gather[0].data_ptr(), (a_gather[1].data_ptr() if a_gather[1] is not None else None)
        d.a_gather_S, d.a_gather_N, d.a_gather_rows = a_gather[0].shape[1], a_gather[2], a_gather[3]
    if head is not None:          # (w [N], b [1] or None, act, out, ld): a Cout = 1 convolution behind this layer (struct ogmm_gemm.rd_*)
        d.rd_w, d.rd_b, d.rd_act = head[0].data_ptr(), (head[1].data_ptr() if head[1] is not None else None), head[2]
        d.rd_out, d.rd_ld = head[3].data_ptr(), head[4]
        if C is None:
            store_c = False
    variant = None
    if GEMM_TIMELINE is not None:
        variant = ("f16x3" if split is not None else "f32") + ("_pool" if pool_k else "") + ("_n64" if N <= 64 else "")
        if split is not None and split.get("variant", 1) != PREC_F16X3_FRAG and batch == (1, 1) and not pool_k:
            variant += "_rowmajor"
    if GEMM_TIMELINE is None or (GEMM_TIMELINE_ONLY is not None and variant not in GEMM_TIMELINE_ONLY):
        _lib.call("ogmm_gemm_nt", ctypes.byref(d), _stream())
        return
    e0, e1 = _timing_event(), _timing_event()
    e0.record()
    _lib.call("ogmm_gemm_nt", ctypes.byref(d), _stream())
    e1.record()
    # flops, then the launch's algorithmic HBM bytes: A (and A2) read once, C written once, residual read once, weights once
    nb = batch[0] * batch[1]
    abytes = 4.0 * nb * (M * (K1 + K2) + (M * N if store_c else 0) + (M * N if res is not None else 0)) + 4.0 * N * (K1 + K2) * (nb if batch != (1, 1) else 1)
    # matrix instructions issued per algorithmic product: 3 (split engines), 2 where the two-term form runs (fragment-major image, no A transform), 1 (fp32 engine, reduced mode)
    issued = 1 if split is None or (single_term and d.precision == PREC_F16_FRAG) else (
        terms if terms in (1, 2) and split.get("variant") == PREC_F16X3_FRAG and a_affine is None and pool_k == 0 and N >= (512 if terms == 1 else 256) else 3)
    GEMM_TIMELINE.append((e0, e1, 2.0 * M * N * (K1 + K2) * nb, variant, abytes, issued))


def instnorm_fusable(layer_split, N):
    """The fused InstanceNorm path needs the fragment-major fp16x3 engine and clouds that are whole row tiles."""
    return layer_split is not None and layer_split.get("variant") == PREC_F16X3_FRAG and N % 256 == 0


def instnorm_finalize(col_stats, rows, eps=1e-5, clear=False):
    """col_stats [G, cols, 2] float64 -> (scale, shift) float32 [G, cols].  clear: the statistics are zeroed behind the read (a persistent buffer is then
    ready for the next forward's accumulation without a fill)."""
    G, cols, _ = col_stats.shape
    assert col_stats.is_contiguous() and col_stats.dtype == torch.float64
    scale = torch.empty((G, cols), dtype=torch.float32, device=col_stats.device)
    shift = torch.empty_like(scale)
    _lib.call("ogmm_instnorm_finalize", _p(col_stats), G * cols, rows, eps, _p(scale), _p(shift), 1 if clear else 0, _stream())
    return scale, shift


def conv1x1(x, layer, act=ACT_NONE, out=None, x2=None, res=None, split=None, overflow=None, col_stats=None, a_affine=None, group_rows=0, head=None, store=True,
            eng=None, terms=0, norm_bwd=None):
    """y[rows, Cout] = act((x | x2)[rows, K] @ W^T * scale + shift) + res for a packed layer
    (dict with W [Cout, Kpad], scale, shift -- see gmmreg.pack_*).  x, x2, res, out may be column views
    of wider row-major buffers (last stride 1).  split=True uses the layer's pre-split weights (fp16x3 engine)
    when the layer carries them; eng: the calling model's Engine (split / overflow default to its settings)."""
    x = _f32(x, "x")
    rows, K1 = x.shape
    assert x.stride(1) == 1
    eng = eng or DEFAULT_ENGINE
    split = eng.split if split is None else split
    overflow = eng.overflow if overflow is None else overflow
    W = layer["W"]
    Cout, Kp = W.shape
    K2 = 0
    if x2 is not None:
        assert x2.stride(1) == 1 and x2.shape[0] == rows
        K2 = x2.shape[1]
    assert K1 + K2 == Kp, "conv1x1: input channels %d+%d != packed K %d" % (K1, K2, Kp)
    if out is None and store:
        out = torch.empty((rows, Cout), dtype=torch.float32, device=x.device)
    assert out is None or (out.stride(1) == 1 and out.shape == (rows, Cout))
    if res is not None:
        assert res.stride(1) == 1 and res.shape == (rows, Cout)
    gemm_nt(x, x.stride(0), K1, W, Kp, rows, Cout, C=out, ldc=(out.stride(0) if out is not None else 0), head=head,
            A2=x2, lda2=(x2.stride(0) if x2 is not None else 0), K2=K2,
            scale=layer.get("scale"), shift=layer.get("shift"), act=act,
            res=res, ldr=(res.stride(0) if res is not None else 0),
            split=(layer.get("split") if split else None), overflow=overflow, col_stats=col_stats, a_affine=a_affine, group_rows=group_rows,
            single_term=eng.single_term, terms=terms, norm_bwd=norm_bwd)
    return out


def edgeconv_first(xyz, idx, layer, pool_out):
    """-> h1 [C*N*k, 64]; pooled max written into pool_out[:, :64]   (models/dgcnn.py:137-139)."""
    C, N, k = idx.shape
    h1 = torch.empty((C * N * k, 64), dtype=torch.float32, device=xyz.device)
    _lib.call("ogmm_edgeconv_first", _p(_f32(xyz, "xyz")), _p(_i32(idx, "idx")), C, N, k, _p(layer["W"]), _p(layer["scale"]),
              _p(layer["shift"]), _p(h1), _p(pool_out), pool_out.stride(0), _stream())
    return h1


def edgeconv_fused_supported(k, layers):
    return 7 <= k <= 32 and all(l.get("split") is not None and l["split"].get("variant") == PREC_F16X3_FRAG for l in layers[1:])


def edgeconv_fused(xyz, idx, layers, xcat, status=None):
    """The whole EdgeConv chain in one kernel: layers = [emd1, emd2, emd3, emd4] packed dicts; fills xcat[:, :512].
    status: device int32[1] (or None) that the producer / consumer kernel ORs STATUS_EDGECONV_PROTOCOL into when one of its bounded waits times out."""
    C, N, k = idx.shape
    args = [_p(_f32(xyz, "xyz")), _p(_i32(idx, "idx")), C, N, k, _p(layers[0]["W"]), _p(layers[0]["scale"]), _p(layers[0]["shift"])]
    for l in layers[1:]:
        sp = l["split"]
        args += [_p(sp["W_hi"]), _p(sp["W_lo"]), _p(l["scale"]), _p(l["shift"]), sp["inv_scale"]]
    # algorithmic work: the four 1x1 convolutions over C*N*k edges (models/dgcnn.py:121-124); bytes: xyz + idx in, xcat out
    E = float(C) * N * k
    # k = 20 (the reference's gnn_k): the producer / consumer pipeline (edgeconv_pc.hip, bit-identical); ops.EDGECONV_PC = False: the barrier-phased kernel
    fn = "ogmm_edgeconv_pc" if k == 20 and EDGECONV_PC else "ogmm_edgeconv_fused"
    tail = (_p(status),) if fn == "ogmm_edgeconv_pc" else ()
    _timed_call("edgeconv_fused_kernel", 2.0 * E * (6 * 64 + 64 * 64 + 64 * 128 + 128 * 256), 4.0 * (3 * C * N + E + 512.0 * C * N),
                fn, *args, _p(xcat), xcat.stride(0), *tail, _stream())
    return xcat


def edgeconv_layer(h, layer, k, pool_out, store=True, split=None, overflow=None, eng=None):
    """conv + BN + ReLU on the per-edge tensor h [E, Cin] with max over each point's k edges fused in
    (models/dgcnn.py:141-148).  Returns the un-pooled [E, Cout] (None when store=False)."""
    E, Cin = h.shape
    eng = eng or DEFAULT_ENGINE
    split = eng.split if split is None else split
    overflow = eng.overflow if overflow is None else overflow
    W = layer["W"]
    Cout = W.shape[0]
    out = torch.empty((E, Cout), dtype=torch.float32, device=h.device) if store else None
    gemm_nt(h, h.stride(0), Cin, W, W.shape[1], E, Cout, C=out, ldc=Cout, scale=layer["scale"], shift=layer["shift"],
            act=ACT_RELU, pool_k=k, pool_out=pool_out, ldp=pool_out.stride(0), store_c=store,
            split=(layer.get("split") if split else None), overflow=overflow)
    return out


def pos_hidden(xyz, idx, k_pos, p):
    C, N, _ = xyz.shape
    hd = torch.empty((C * N, 64), dtype=torch.float32, device=xyz.device)
    ha = torch.empty_like(hd)
    _lib.call("ogmm_pos_hidden", _p(xyz), _p(_i32(idx, "idx")), idx.shape[2], k_pos, C, N, _p(p["w_dis"]), _p(p["s_dis"]), _p(p["t_dis"]),
              _p(p["w_ang"]), _p(p["s_ang"]), _p(p["t_ang"]), _p(hd), _p(ha), _stream())
    return hd, ha


def pack_frag_batched(x, batch, rows):
    """x [(batch*rows), K] fp32 -> split dict for gemm_nt(batch=(batch,1)): per-batch fragment images of `rows` rows (zero-padded to a
    multiple of 256), laid out back to back; `sB` (halfs) is the stride between them."""
    K = x.shape[1]
    assert x.stride(1) == 1 and x.shape[0] == batch * rows and K % 64 == 0
    rp = (rows + 255) // 256 * 256
    hi = torch.zeros((batch, rp * K), dtype=torch.float16, device=x.device) if rp != rows else torch.empty((batch, rp * K), dtype=torch.float16, device=x.device)
    lo = torch.zeros_like(hi) if rp != rows else torch.empty_like(hi)
    for b in range(batch) if rp != rows else ():
        _lib.call("ogmm_pack_frag", _p(x[b * rows:]), x.stride(0), rows, K, _p(hi[b]), _p(lo[b]), _stream())
    if rp == rows:      # images are contiguous row blocks: one launch packs every batch element
        _lib.call("ogmm_pack_frag", _p(x), x.stride(0), batch * rows, K, _p(hi), _p(lo), _stream())
    return {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": PREC_F16X3_FRAG, "ldb_h": K, "sB": rp * K}


def l2norm_pack_frag_batched(x, batch, rows, rnorm_of=None):
    """pack_frag_batched(l2norm_rows(x), batch, rows) in one kernel per image set: the normalised rows exist only as split images.
    rnorm_of [rows', K] (optional): the same launch also computes row_rnorm(rnorm_of) (the src half's row scale of the similarity GEMM), returned as
    the image dict's "rnorm" entry."""
    K = x.shape[1]
    assert x.stride(1) == 1 and x.shape[0] == batch * rows and K % 64 == 0
    rp = (rows + 255) // 256 * 256
    hi = torch.zeros((batch, rp * K), dtype=torch.float16, device=x.device) if rp != rows else torch.empty((batch, rp * K), dtype=torch.float16, device=x.device)
    lo = torch.zeros_like(hi) if rp != rows else torch.empty_like(hi)
    for b in range(batch) if rp != rows else ():
        _lib.call("ogmm_l2norm_pack_frag", _p(x[b * rows:]), x.stride(0), rows, K, _p(hi[b]), _p(lo[b]), _stream())
    rnorm = None
    if rp == rows and rnorm_of is not None and rnorm_of.shape[1] == K and rnorm_of.stride(1) == 1:
        rnorm = torch.empty((rnorm_of.shape[0],), dtype=torch.float32, device=x.device)
        _lib.call("ogmm_l2norm_pack_frag_rnorm", _p(x), x.stride(0), batch * rows, K, _p(hi), _p(lo), _p(_f32(rnorm_of, "rnorm_of")), rnorm_of.stride(0),
                  rnorm_of.shape[0], _p(rnorm), _stream())
    elif rp == rows:
        _lib.call("ogmm_l2norm_pack_frag", _p(x), x.stride(0), batch * rows, K, _p(hi), _p(lo), _stream())
    if rnorm is None and rnorm_of is not None:
        rnorm = row_rnorm(rnorm_of)
    return {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": PREC_F16X3_FRAG, "ldb_h": K, "sB": rp * K, "rnorm": rnorm}


def attention(q, k, v, C, N, M, H, out=None, use_workspace=True, qk_terms=0):
    """Fused anchor attention (models/attn.py:78-82).  q [C*N, D], k, v [C*M, D] (row-major views, last stride 1), head-major
    channels; returns [C*N, D].  qk_terms = 1: the score product with both operands rounded to binary16 (the term budget's entry "<transformer>.qk")."""
    D = q.shape[1]
    dh = D // H
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1 and q.shape[0] == C * N and k.shape[0] == C * M
    if out is None:
        out = torch.empty((C * N, D), dtype=torch.float32, device=q.device)
    ws = None
    if use_workspace:
        nbytes = _lib.load().ogmm_attention_workspace_bytes(C, M, H, dh)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    # algorithmic work: Q K^T and P V per head; bytes: Q in, O out, K and V in
    _timed_call("attention_t_kernel", 4.0 * C * N * M * D, 4.0 * (2.0 * C * N * D + 2.0 * C * M * D),
                "ogmm_attention_terms", _p(_f32(q, "q")), q.stride(0), _p(_f32(k, "k")), k.stride(0), _p(_f32(v, "v")), v.stride(0), C, N, M, H, dh,
                1.0 / dh ** .5, _p(out), out.stride(0), int(qk_terms), _p(ws), _stream())
    return out


def attention_supported(M, dh):
    return dh == 128 and M in (32, 64, 128)


def add_n(maps):
    """sum of up to 8 [R, C] maps (last stride 1, own row pitches) in one pass (kernel T12)"""
    n = len(maps)
    R, Cc = maps[0].shape
    assert 1 <= n <= 8 and all(m.shape == (R, Cc) and m.stride(1) == 1 and m.dtype == torch.float32 for m in maps)
    out = torch.empty((R, Cc), dtype=torch.float32, device=maps[0].device)
    ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in maps])
    lds = (ctypes.c_int64 * n)(*[m.stride(0) for m in maps])
    _lib.call("ogmm_add_n", n, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(lds, ctypes.c_void_p), R, Cc, _p(out), out.stride(0), _stream())
    return out


def attention_bwd_supported(M, dh):
    return bool(_lib.load().ogmm_attention_bwd_supported(M, dh))


ATTN_BWD_F16X3 = 2      # fp16x3 step: 0 = exact-fp32 attention backward, 1 = S and dP on the fp16x3 arithmetic, 2 = all five products (A/B)


def attention_bwd(q, k, v, dout, C, N, M, H, split=False, overflow=None):
    """Backward of attention(): (dq [C*N, D], dk [C*M, D], dv [C*M, D]) from dout = dL/dO; scores re-formed on chip (kernel T11).
    split: False = exact fp32; True = the fp16x3 training step's form (ops.ATTN_BWD_F16X3, default 2); 1 = S = Q K^T and dP = dO V^T on the engines'
    fp16x3 arithmetic; 2 = all five products (csrc/train_attn_bwd16.hip).  `overflow` reports operands beyond binary16."""
    D = q.shape[1]
    dh = D // H
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1 and dout.stride(1) == 1
    assert q.shape[0] == C * N and dout.shape == q.shape and k.shape[0] == C * M and v.shape == k.shape
    dq = torch.empty((C * N, D), dtype=torch.float32, device=q.device)
    dk = torch.empty((C * M, D), dtype=torch.float32, device=q.device)
    dv = torch.empty((C * M, D), dtype=torch.float32, device=q.device)
    level = ATTN_BWD_F16X3 if split is True else int(split)
    if level:
        _lib.call("ogmm_attention_bwd_f16x3", _p(_f32(q, "q")), q.stride(0), _p(_f32(k, "k")), k.stride(0), _p(_f32(v, "v")), v.stride(0),
                  _p(_f32(dout, "dout")), dout.stride(0), C, N, M, H, dh, 1.0 / dh ** .5, _p(dq), dq.stride(0), _p(dk), dk.stride(0),
                  _p(dv), dv.stride(0), 1 if level >= 2 else 0, _p(overflow), _stream())
        return dq, dk, dv
    _lib.call("ogmm_attention_bwd", _p(_f32(q, "q")), q.stride(0), _p(_f32(k, "k")), k.stride(0), _p(_f32(v, "v")), v.stride(0),
              _p(_f32(dout, "dout")), dout.stride(0), C, N, M, H, dh, 1.0 / dh ** .5, _p(dq), dq.stride(0), _p(dk), dk.stride(0),
              _p(dv), dv.stride(0), _stream())
    return dq, dk, dv


# ---------------------------------------------------------------------------------------------- row / column kernels
def softmax_rows_(x2d):
    assert x2d.stride(1) == 1
    _lib.call("ogmm_softmax_rows", _p(_f32(x2d, "x")), x2d.shape[0], x2d.shape[1], x2d.stride(0), _stream())
    return x2d


def instnorm_relu_(x, C, N, eps=1e-5):
    """x [(C*N), D] row-major (stride(0) = ld): InstanceNorm over the N rows of each cloud + ReLU, in place."""
    assert x.stride(1) == 1 and x.shape[0] == C * N
    _lib.call("ogmm_instnorm_relu", _p(_f32(x, "x")), x.stride(0), C, N, x.shape[1], eps, _stream())
    return x


def l2norm_rows(x, out=None):
    assert x.stride(1) == 1
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.call("ogmm_l2norm_rows", _p(_f32(x, "x")), x.stride(0), x.shape[0], x.shape[1], _p(out), out.stride(0), _stream())
    return out


def conv1x1_gathered(feats, C, N, ids, layer, act=ACT_NONE, cloud_map=None, eng=None, terms=0):
    """conv1x1(gather_rows(feats, ids, cloud_map), layer): a convolution over the anchor rows of every cloud (models/gmmreg.py:54, 67-68).  Where the
    engine takes it, the rows are gathered by the GEMM's own operand DMA (struct ogmm_gemm.a_gather_*) and the anchor tensor is never written."""
    feats, ids = _f32(feats, "feats"), _i32(ids, "ids")
    rows, D = feats.shape
    S = ids.shape[1]
    Cout = layer["W"].shape[0]
    eng = eng or DEFAULT_ENGINE
    sp = layer.get("split") if eng.split else None
    if (FUSE_GATHER and sp is not None and sp.get("variant") == PREC_F16X3_FRAG and not eng.single_term and feats.stride(1) == 1 and ids.is_contiguous()
            and _lib.load().ogmm_gemm_gather_fusable(C * S, Cout, D, rows) == 1):
        out = torch.empty((C * S, Cout), dtype=torch.float32, device=feats.device)
        cm = _i32(cloud_map, "cloud_map") if cloud_map is not None else None
        gemm_nt(feats, feats.stride(0), D, layer["W"], D, C * S, Cout, C=out, ldc=Cout, scale=layer.get("scale"), shift=layer.get("shift"), act=act,
                split=sp, overflow=eng.overflow, a_gather=(ids, cm, N, rows), terms=terms)
        return out
    return conv1x1(gather_rows(feats, feats.stride(0), C, N, D, ids, cloud_map=cloud_map).view(C * S, D), layer, act, eng=eng, terms=terms)


def conv1x1_head(x, layer, act, w, b, head_act, out, ldy=1, x2=None, eng=None, terms=0):
    """out[row * ldy] = head_act(act(conv1x1(x, layer))[row] . w + b): a layer followed by a Cout = 1 convolution (models/gmmreg.py:30-47: proj, overlap).
    Where the engine takes it (N = 256, whole row tiles) the head runs in the layer's epilogue and the 256-wide map is never written."""
    rows, K1 = x.shape
    Cout = layer["W"].shape[0]
    K2 = x2.shape[1] if x2 is not None else 0
    eng = eng or DEFAULT_ENGINE
    sp = layer.get("split") if eng.split else None
    if (FUSE_HEAD and sp is not None and sp.get("variant") == PREC_F16X3_FRAG and not eng.single_term and _lib.load().ogmm_gemm_rowdot_fusable(rows, Cout, K1, K2) == 1):
        conv1x1(x, layer, act, x2=x2, head=(w, b, head_act, out, ldy), store=False, eng=eng, terms=terms)
        return
    rowdot(conv1x1(x, layer, act, x2=x2, eng=eng, terms=terms), w, b, head_act, out, ldy=ldy)


def rowdot(x, w, b, act, out, ldy=1):
    """out[m*ldy] = act(x[m,:] . w + b)."""
    assert x.stride(1) == 1
    _lib.call("ogmm_rowdot", _p(_f32(x, "x")), x.stride(0), x.shape[0], x.shape[1], _p(w), _p(b), act, _p(out), ldy, _stream())
    return out


def _overlap_ws(B, N, device):
    ws = torch.empty(_lib.load().ogmm_overlap_cross_workspace_bytes(B, N), dtype=torch.uint8, device=device)
    ws.record_stream(torch.cuda.current_stream())
    return ws


def overlap_fusable(B, N, D, eng=None):
    """True if the similarity GEMM can run the overlap block's softmax-dots in its epilogue (struct ogmm_gemm.ovl_rowpart)."""
    eng = eng or DEFAULT_ENGINE
    return bool(eng.split) and not eng.single_term and _lib.load().ogmm_gemm_overlap_fusable(B, N, D) == 1


def row_rnorm(x):
    """1 / max(|row|_2, 1e-12) per row of x [rows, D] (row stride = x.stride(0)): F.normalize's divisor as a GEMM row scale."""
    rows, D = x.shape
    out = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.call("ogmm_row_rnorm", _p(_f32(x, "x")), x.stride(0), rows, D, _p(out), _stream())
    return out


def overlap_fused(f_src, tgt_img, B, N, D, o_src, o_tgt, ldo_in, wo_src, wo_tgt, ldo, overflow=None, terms=0):
    """models/gmmreg.py:75-80 without the similarity matrix: S = normalize(f_src) normalize(f_tgt)^T lives only in the GEMM's accumulators.
    f_src [B*N, D] un-normalised (its 1/|row| is a row scale), tgt_img = l2norm_pack_frag_batched(f_tgt); o_* / wo_* as overlap_cross."""
    nt = N // 256
    rowpart = torch.empty((B, nt, N, 3), dtype=torch.float32, device=f_src.device)
    colpart = torch.empty((B, nt, N, 3), dtype=torch.float32, device=f_src.device)
    rinv = tgt_img.get("rnorm")          # (l2norm_pack_frag_batched(..., rnorm_of=f_src): computed by the launch that made the image)
    if rinv is None:
        rinv = row_rnorm(f_src)
    gemm_nt(f_src, f_src.stride(0), D, None, D, N, N, batch=(B, 1), sA=(N * f_src.stride(0), 0), split=tgt_img, overflow=overflow,
            overlap=(o_tgt, o_src, ldo_in, rowpart, colpart), row_rscale=rinv, terms=terms)          # the reference weights the ROW softmax with src_o, indexed by column
    _lib.call("ogmm_overlap_finalize", _p(rowpart), _p(colpart), B, N, _p(wo_src), _p(wo_tgt), ldo, _stream())
    for t in (rowpart, colpart, rinv):
        t.record_stream(torch.cuda.current_stream())


def overlap_cross(S, o_src, o_tgt, ldo_in, wo_src, wo_tgt, ldo, two_pass=False):
    """models/gmmreg.py:79-80 on S [B,N,N]; one pass over S (two_pass: the older row kernel + column kernel)."""
    B, N, _ = S.shape
    if two_pass:
        _lib.call("ogmm_overlap_cross", _p(_f32(S, "S")), B, N, _p(o_src), _p(o_tgt), ldo_in, _p(wo_src), _p(wo_tgt), ldo, _stream())
        return
    ws = _overlap_ws(B, N, S.device)
    _lib.call("ogmm_overlap_cross_ws", _p(_f32(S, "S")), B, N, _p(o_src), _p(o_tgt), ldo_in, _p(wo_src), _p(wo_tgt), ldo, None, _p(ws), _stream())


# ---------------------------------------------------------------------------------------------- GMM head
def gmm_em(xyz, o, ids0, iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=1e-2, group_size=None, engine=None, return_resid=False,
           return_sweeps=False, status=None):
    """-> gamma [C,N,J], pi [C,J], mu [C,J,3] (, resid [C,iters,sk_iters]) (, sweeps int32 [C/group_size, iters])   (lib/utils.py:269-288).
    thresh / group_size: the reference's Sinkhorn early exit (lib/utils.py:99-102): an E-step's sweeps end after the first sweep whose residual,
    averaged over the `group_size` clouds of one reference call (None: all C clouds are one call), is below thresh; thresh <= 0 runs every sweep.
    resid: every sweep's sum|u - u0| + sum|v - v0| per cloud, NaN for sweeps that did not run; sweeps: the sweeps every E-step ran per call group.
    status: device int32[1] (or None): with the exit on, the call's protocol-error word (a bounded wait between the clouds of a group timed out: pi / mu
    are NaN-poisoned) is ORed into it as STATUS_EM_EXIT_PROTOCOL behind the kernels -- two tiny device ops, no host synchronisation."""
    C, N, _ = xyz.shape
    J = ids0.shape[1]
    assert o.is_contiguous() and o.shape == (C, N) and ids0.is_contiguous()
    G = C if group_size is None else int(group_size)
    if G <= 0 or C % G != 0:
        raise _lib.OgmmError("gmm_em: %d clouds are not whole call groups of %d" % (C, G))
    lib = _lib.load()
    dev = xyz.device
    gamma = torch.empty((C, N, J), dtype=torch.float32, device=dev)
    pi = torch.empty((C, J), dtype=torch.float32, device=dev)
    mu = torch.empty((C, J, 3), dtype=torch.float32, device=dev)
    exit_on = thresh is not None and thresh > 0 and sk_iters > 1
    if engine is None:       # the on-chip loop while the N x J cost matrix fits one CU's LDS, the grid-wide sequence beyond
        engine = "chip" if lib.ogmm_gmm_em_chip_cached(N, J) == 1 or J > 128 else "multi"          # (the grid-wide kernels keep a row of J <= 128 exponents in registers)
        if engine == "chip" and exit_on and G > lib.ogmm_gmm_em_chip_max_group(N, J) and J <= 128:
            engine = "multi"          # the clouds of a call group wait for each other on chip: a group beyond one resident round takes the launch sequence
    resid = torch.empty((C, iters, sk_iters), dtype=torch.float32, device=dev) if return_resid else None
    sweeps = torch.empty((C // G, iters), dtype=torch.int32, device=dev) if return_sweeps else None
    xws = None
    if exit_on or return_resid:
        xws = torch.empty(lib.ogmm_gmm_em_exit_workspace_bytes(C, N, iters, sk_iters, G), dtype=torch.uint8, device=dev)
        xws.record_stream(torch.cuda.current_stream())
    head = (_p(_f32(xyz, "xyz")), _p(_f32(o, "o")), _p(_i32(ids0, "ids0")), C, N, J, iters, sk_iters, epsilon, tau, float(thresh or 0.0), G,
            _p(gamma), _p(pi), _p(mu), _p(resid), _p(sweeps), _p(xws))
    if engine == "multi":
        ws = torch.empty(lib.ogmm_gmm_em_workspace_bytes(C, N, J), dtype=torch.uint8, device=dev)
        _lib.call("ogmm_gmm_em_multi", *head, _p(ws), _stream())
        ws.record_stream(torch.cuda.current_stream())
    else:
        _lib.call("ogmm_gmm_em", *head, _stream())
    if status is not None and exit_on and xws is not None:
        err = xws[4:8].view(torch.int32)          # word 1 of the exit workspace (include/ogmm_hip.h)
        torch.bitwise_or(status, err * _lib.STATUS_EM_EXIT_PROTOCOL, out=status)
    out = (gamma, pi, mu)
    if return_resid:
        out += (resid,)
    if return_sweeps:
        out += (sweeps,)
    return out


def gmm_feat_mean(gamma, pi, feats, C, N):
    J, D = gamma.shape[2], feats.shape[1]
    assert feats.stride(1) == 1 and feats.shape[0] == C * N
    out = torch.empty((C, J, D), dtype=torch.float32, device=feats.device)
    _lib.call("ogmm_gmm_feat_mean", _p(_f32(gamma, "gamma")), _p(_f32(pi, "pi")), _p(_f32(feats, "feats")), feats.stride(0), C, N, J, D,
              _p(out), _stream())
    return out


def match_kabsch(mu_s, mu_t, f_s, f_t, temperature=0.05, want_scores=False):
    """mu_* [B,J,3], f_* [B,J,D] (contiguous) -> R [B,3,3], t [B,3] (, scores [B,J,J])   (models/dgcnn.py:96-115)."""
    B, J, D = f_s.shape
    for t_ in (mu_s, mu_t, f_s, f_t):
        assert _f32(t_, "match input").is_contiguous()
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=f_s.device)
    t = torch.empty((B, 3), dtype=torch.float32, device=f_s.device)
    sc = torch.empty((B, J, J), dtype=torch.float32, device=f_s.device) if want_scores else None
    _lib.call("ogmm_match_kabsch", _p(mu_s), _p(mu_t), _p(f_s), _p(f_t), B, J, D, temperature, _p(R), _p(t), _p(sc), _stream())
    return (R, t, sc) if want_scores else (R, t)


def kabsch(src, corr, w):
    """src, corr [B,3,J], w [B,1,J] or [B,J] -> R [B,3,3], t [B,3,1]   (lib/se3.py:256-289)."""
    src, corr = _f32(src, "src").contiguous(), _f32(corr, "corr").contiguous()
    B, _, J = src.shape
    w = _f32(w, "w").reshape(B, J).contiguous()
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=src.device)
    t = torch.empty((B, 3, 1), dtype=torch.float32, device=src.device)
    _lib.call("ogmm_kabsch", _p(src), _p(corr), _p(w), B, J, _p(R), _p(t), _stream())
    return R, t


def clu_infonce(xyz, mu, feats, mu_feat, C, N, tau=0.1):
    """-> (row_loss [C,2,J], near [C,J] int32)   (lib/loss.py:109-118, :22-57; lib/utils.py:244-254)."""
    J, D = mu_feat.shape[1], mu_feat.shape[2]
    assert feats.stride(1) == 1 and mu.is_contiguous() and mu_feat.is_contiguous()
    row_loss = torch.empty((C, 2, J), dtype=torch.float32, device=xyz.device)
    near = torch.empty((C, J), dtype=torch.int32, device=xyz.device)
    _lib.call("ogmm_clu_infonce", _p(_f32(xyz, "xyz")), _p(_f32(mu, "mu")), _p(_f32(feats, "feats")), feats.stride(0), _p(mu_feat), C, N, J, D,
              tau, _p(row_loss), _p(near), _stream())
    return row_loss, near


def icp_point_to_point(src, tgt, R0, t0, max_corr_dist, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, want_stats=False, engine=None):
    """src [B,N,3], tgt [B,Nt,3], R0 [B,3,3], t0 [B,3] -> R [B,3,3], t [B,3] (, fitness, rmse float32 [B], iters int32 [B])
    (lib/o3dutils.py:172-214)."""
    src, tgt = _f32(src, "src").contiguous(), _f32(tgt, "tgt").contiguous()
    B, N, _ = src.shape
    Nt = tgt.shape[1]
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=src.device)
    t = torch.empty((B, 3), dtype=torch.float32, device=src.device)
    fit = torch.empty(B, dtype=torch.float32, device=src.device) if want_stats else None
    rmse = torch.empty(B, dtype=torch.float32, device=src.device) if want_stats else None
    iters = torch.empty(B, dtype=torch.int32, device=src.device) if want_stats else None
    # contiguous copies are bound to names that live until after the launch: a temporary's block would go back to the caching allocator at once
    # and could be handed to the `ws` / output allocation below, i.e. be overwritten by the kernel that still reads it
    R0c = None if R0 is None else _f32(R0, "R0").contiguous()
    t0c = None if t0 is None else _f32(t0, "t0").contiguous()
    args = (_p(src), _p(tgt), B, N, Nt, _p(R0c),
            _p(t0c), float(max_corr_dist), int(max_iter), float(rel_fitness), float(rel_rmse),
            _p(R), _p(t), _p(fit), _p(rmse), _p(iters))
    if engine is None:          # one workgroup per pair once there are enough pairs to fill the chip, the grid-wide sequence below that
        engine = "chip" if B >= 256 else "multi"
    if engine == "multi":
        ws = torch.empty(_lib.load().ogmm_icp_workspace_bytes(B, N), dtype=torch.uint8, device=src.device)
        _lib.call("ogmm_icp_point_to_point_ws", *args, _p(ws), _stream())
    else:
        _lib.call("ogmm_icp_point_to_point", *args, _stream())
    return (R, t, fit, rmse, iters) if want_stats else (R, t)


def rotation_from_cov(M):
    """M [B,3,3] -> R = V diag(1,1,det(V U^T)) U^T of its SVD (baseline/deepgmr.py:28-34)"""
    M = _f32(M, "M").contiguous()
    R = torch.empty_like(M)
    _lib.call("ogmm_rotation_from_cov", _p(M), M.shape[0], _p(R), _stream())
    return R


def min_sqdist(a, b):
    """a [B,Na,3], b [B,Nb,3] -> [B,Na] squared distance to the nearest point of b (lib/metric.py:193-194 + min)"""
    a, b = _f32(a, "a").contiguous(), _f32(b, "b").contiguous()
    out = torch.empty(a.shape[:2], dtype=torch.float32, device=a.device)
    _lib.call("ogmm_min_sqdist", _p(a), _p(b), a.shape[0], a.shape[1], b.shape[1], _p(out), _stream())
    return out


# ---------------------------------------------------------------------------------------------- training mode
def colstats(x, group_rows):
    """x [rows, cols] (last stride 1) -> float64 [G, cols, 2] = {sum, sum of squares} per row group"""
    assert x.stride(1) == 1
    rows, cols = x.shape
    st = torch.empty((rows // group_rows, cols, 2), dtype=torch.float64, device=x.device)
    _lib.call("ogmm_colstats", _p(_f32(x, "x")), x.stride(0), rows, cols, group_rows, _p(st), _stream())
    return st


def norm_finalize(st, group_rows, weight, bias, eps):
    """st float64 [G, cols, 2] = {sum, sum of squares} over group_rows rows -> (scale, shift, mean, rstd float32 [G, cols], mean64, var64 float64 [G, cols]):
    the constants of one normalisation layer in one launch (scale = gamma rstd, shift = beta - mean scale; var biased, clamped at 0)"""
    assert st.dtype == torch.float64 and st.dim() == 3 and st.shape[2] == 2
    st = st.contiguous()
    G, cols = st.shape[0], st.shape[1]
    f32 = torch.empty((4, G, cols), dtype=torch.float32, device=st.device)
    f64 = torch.empty((2, G, cols), dtype=torch.float64, device=st.device)
    w = None if weight is None else _f32(weight.detach(), "weight").contiguous()
    b = None if bias is None else _f32(bias.detach(), "bias").contiguous()
    _lib.call("ogmm_norm_finalize", _p(st), G, cols, group_rows, float(eps), _p(w), _p(b), _p(f32[0]), _p(f32[1]), _p(f32[2]), _p(f32[3]), _p(f64[0]), _p(f64[1]),
              _stream())
    return f32[0], f32[1], f32[2], f32[3], f64[0], f64[1]


def norm_param_grads(sums):
    """sums float64 [G, cols, 2] = {sum dz, sum dz xhat} -> (dgamma, dbeta) float32 [cols]: the sums over the groups, one launch"""
    sums = sums.contiguous()
    G, cols = sums.shape[0], sums.shape[1]
    out = torch.empty((2, cols), dtype=torch.float32, device=sums.device)
    _lib.call("ogmm_norm_param_grads", _p(sums), G, cols, _p(out[0]), _p(out[1]), _stream())
    return out[0], out[1]


def bn_update_running(mean64, var64, group_rows, momentum, running_mean, running_var, num_batches):
    """torch.nn.BatchNorm1d's running-statistics update for the G sequential calls whose batch statistics are mean64 / var64 [G, cols] (biased variance), in
    place, one launch; num_batches (int64 scalar tensor) += G"""
    G, cols = mean64.shape
    assert (mean64.dtype == torch.float64 and var64.dtype == torch.float64 and mean64.is_contiguous() and var64.is_contiguous() and
            running_mean.dtype == torch.float32 and running_var.dtype == torch.float32 and running_mean.is_contiguous() and running_var.is_contiguous() and
            running_mean.numel() == cols and running_var.numel() == cols and (num_batches is None or num_batches.dtype == torch.int64))
    _lib.call("ogmm_bn_update_running", _p(mean64), _p(var64), G, cols, group_rows, float(momentum), _p(running_mean), _p(running_var), _p(num_batches), _stream())


def affine_act(x, group_rows, scale, shift, act, out=None):
    """act(x * scale[g] + shift[g]) with per-(group, column) float32 scale / shift [G, cols]"""
    assert x.stride(1) == 1 and scale.is_contiguous() and shift.is_contiguous()
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    _lib.call("ogmm_affine_act", _p(_f32(x, "x")), x.stride(0), rows, cols, group_rows, _p(_f32(scale, "scale")), _p(_f32(shift, "shift")), act,
              _p(out), out.stride(0), _stream())
    return out


def norm_bwd(x, dy, group_rows, scale, shift, mean, rstd, act, dpool=None, arg=None, k=0):
    """backward of y = act(x * scale + shift), scale = gamma * rstd, shift = beta - mean * scale; the upstream gradient is dy
    (may be None) plus, for a map that was max-pooled over k rows, dpool routed to the rows `arg`
    -> (dx, sums float64 [G, cols, 2] = {sum dz, sum dz*xhat})"""
    rows, cols = x.shape
    assert x.stride(1) == 1 and (dy is None or dy.stride(1) == 1) and (dpool is None or dpool.stride(1) == 1)
    G = rows // group_rows
    sums = torch.empty((G, cols, 2), dtype=torch.float64, device=x.device)
    up = (_p(None if dy is None else _f32(dy, "dy")), 0 if dy is None else dy.stride(0),
          _p(None if dpool is None else _f32(dpool, "dpool")), 0 if dpool is None else dpool.stride(0), _p(arg), k)
    tail = (_p(_f32(scale, "scale")), _p(_f32(shift, "shift")), _p(_f32(mean, "mean")), _p(_f32(rstd, "rstd")), act)
    _lib.call("ogmm_norm_bwd_reduce", _p(_f32(x, "x")), x.stride(0), *up, rows, cols, group_rows, *tail, _p(sums), _stream())
    dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    _lib.call("ogmm_norm_bwd_apply", _p(x), x.stride(0), *up, rows, cols, group_rows, *tail, _p(sums), _p(dx), dx.stride(0), _stream())
    return dx, sums


def norm_bwd_fusable(rows, cols, k, group_rows):
    """would the engine take the normalisation backward's reduction into the epilogue of the [rows, k] x [cols, k]^T GEMM that produces dh (ogmm_gemm.nb_*)?"""
    return NORM_BWD_FUSED and _lib.load().ogmm_gemm_normbwd_fusable(rows, cols, k, group_rows) == 1


def norm_bwd_apply(x, dz, group_rows, scale, shift, mean, rstd, sums):
    """second half of norm_bwd when its reduction came out of the producing GEMM's epilogue: dz = dy * act'(.) is stored already, sums = {sum dz, sum dz xhat}"""
    rows, cols = x.shape
    dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    _lib.call("ogmm_norm_bwd_apply", _p(_f32(x, "x")), x.stride(0), _p(_f32(dz, "dz")), dz.stride(0), _p(None), 0, _p(None), 0, rows, cols, group_rows,
              _p(_f32(scale, "scale")), _p(_f32(shift, "shift")), _p(_f32(mean, "mean")), _p(_f32(rstd, "rstd")), ACT_NONE, _p(sums), _p(dx), dx.stride(0), _stream())
    return dx


def affine_act_pool(x, k, group_rows, scale, shift, act, want_y=True):
    """-> (y or None, pooled [P, cols], arg uint8 [P, cols]) with P = rows / k"""
    rows, cols = x.shape
    assert x.stride(1) == 1 and rows % k == 0 and group_rows % k == 0
    P = rows // k
    y = torch.empty((rows, cols), dtype=torch.float32, device=x.device) if want_y else None
    pooled = torch.empty((P, cols), dtype=torch.float32, device=x.device)
    arg = torch.empty((P, cols), dtype=torch.uint8, device=x.device)
    _lib.call("ogmm_affine_act_pool", _p(_f32(x, "x")), x.stride(0), P, k, cols, group_rows // k, _p(_f32(scale, "scale")), _p(_f32(shift, "shift")), act,
              _p(y), cols, _p(pooled), cols, _p(arg), _stream())
    return y, pooled, arg


def maxpool_k(h, k):
    """-> (out [P, cols], arg uint8 [P, cols])"""
    assert h.stride(1) == 1 and h.shape[0] % k == 0
    P, cols = h.shape[0] // k, h.shape[1]
    out = torch.empty((P, cols), dtype=torch.float32, device=h.device)
    arg = torch.empty((P, cols), dtype=torch.uint8, device=h.device)
    _lib.call("ogmm_maxpool_k", _p(_f32(h, "h")), h.stride(0), P, k, cols, _p(out), out.stride(0), _p(arg), _stream())
    return out, arg


def maxpool_k_bwd(dout, arg, k):
    assert dout.stride(1) == 1 and arg.is_contiguous()
    P, cols = dout.shape
    dh = torch.empty((P * k, cols), dtype=torch.float32, device=dout.device)
    _lib.call("ogmm_maxpool_k_bwd", _p(_f32(dout, "dout")), dout.stride(0), _p(arg), P, k, cols, _p(dh), dh.stride(0), _stream())
    return dh


DW_MIN_TILES = 0          # 0: 256 / 512 by shape (weight_grad); a number: that many tiles at least (A/B timing)
DW_TRANSPOSED_A = True      # 0: materialise dY^T (ogmm_transpose_pad) as rounds 1-3 did (A/B timing, bit-identical)


def weight_grad(dy, xs, overflow=None, x_affine=None, colsum=False, chunk_rows=None, keep_parts=False, out_scale=None, parts_out=None, terms=0):
    """dW = dY^T [x_0 | x_1 | ...] on the fp16x3 engine: dy [R, n], xs = list of [R, k_i] (last stride 1) -> [n, sum k_i].
    x_affine (one x only): (scale [G, k], shift [G, k], relu, group_rows) -- x is a pre-normalisation map, X = relu(x * scale + shift).
    colsum=True: -> (dW, db) with db = dy.sum(0) gathered while dY^T is written (fp64 partial sums), no extra pass over dy.
    dY^T is materialised once (fp32, chunk-major), every x_i becomes per-chunk split fragment images of x_i^T, the contraction
    over r runs as split-K batches of the engine and the partial products are summed (kernels T3 of include/ogmm_hip.h).
    chunk_rows / keep_parts: the row chunks are given (a multiple of 64 that divides R) and the per-chunk products are RETURNED, not summed:
    [R / chunk_rows, n, k] = a batch of independent X_b^T-style products dY_b^T X_b (the overlap block's backward: dS[b]^T fn_src[b])."""
    R, n = dy.shape
    assert dy.stride(1) == 1
    tiles_mn = ((n + 255) // 256) * max((max(x.shape[1] for x in xs) + 255) // 256, 1)
    # row chunks (split K): exactly one 256 x 256 tile per CU where the tile count of dW divides 256 (longer K loops per tile, half the partial products to sum:
    # 115.7 against 117.5 ms per training step), else >= 512 tiles so that a ragged second round costs little (384: 125 ms)
    want_tiles = DW_MIN_TILES if DW_MIN_TILES else (256 if 256 % tiles_mn == 0 else 512)
    S = max(1, min((want_tiles + tiles_mn - 1) // tiles_mn, (R + 255) // 256))
    chunk = ((R + S - 1) // S + 63) // 64 * 64
    if chunk_rows is not None:
        assert chunk_rows % 64 == 0 and R % chunk_rows == 0 and len(xs) == 1 and not colsum
        chunk = chunk_rows
    S = (R + chunk - 1) // chunk
    pitch = chunk + 64
    # round 4: the engine reads dY as it lies (struct ogmm_gemm.a_trans: transposing fragment reads) where every chunk is whole and the shapes fit its
    # 256 x 256 tiles; the bias gradient's column sums then ride on the engine's own operand fragments (struct ogmm_gemm.a_colsum) instead of the transposed copy
    ldy = dy.stride(0)
    direct = (DW_TRANSPOSED_A and R % chunk == 0 and chunk % 32 == 0 and ldy % 4 == 0 and dy.data_ptr() % 16 == 0 and
              all(_lib.load().ogmm_gemm_atrans_supported(n, x.shape[1], chunk, ldy, S) for x in xs))
    csum = None
    if direct:
        csum = torch.zeros(n, dtype=torch.float64, device=dy.device) if colsum else None
    else:
        dyt = torch.empty((S, n, pitch), dtype=torch.float32, device=dy.device)
        csum = torch.empty((16, n), dtype=torch.float64, device=dy.device) if colsum else None
        _lib.call("ogmm_transpose_pad", _p(_f32(dy, "dy")), dy.stride(0), R, n, chunk, pitch, S, _p(dyt), _p(csum), 16, _stream())
    outs = []
    for x in xs:
        assert x.stride(1) == 1 and x.shape[0] == R
        k = x.shape[1]
        n_pad = (k + 255) // 256 * 256
        hi = torch.empty(S * n_pad * pitch, dtype=torch.float16, device=dy.device)
        lo = torch.empty_like(hi)
        aff = x_affine if x_affine is not None else (None, None, False, 0)
        assert x_affine is None or (len(xs) == 1 and aff[0].shape[-1] == k and aff[0].is_contiguous() and aff[1].is_contiguous())
        _lib.call("ogmm_pack_frag_t", _p(_f32(x, "x")), x.stride(0), R, k, chunk, pitch, S, n_pad, _p(hi), _p(lo), _p(overflow), _p(aff[0]), _p(aff[1]),
                  1 if aff[2] else 0, aff[3], _stream())
        part = torch.empty((S, n, k), dtype=torch.float32, device=dy.device) if parts_out is None else parts_out
        assert tuple(part.shape) == (S, n, k) and part.is_contiguous()
        split = {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": PREC_F16X3_FRAG, "ldb_h": pitch, "sB": n_pad * pitch}
        if direct:
            gemm_nt(dy, ldy, chunk, None, 0, n, k, C=part, ldc=k, batch=(S, 1), sA=(chunk * ldy, 0), sC=(n * k, 0), split=split, overflow=overflow,
                    scale=out_scale, a_trans=True, a_colsum=csum if x is xs[0] else None, terms=terms)
        else:
            gemm_nt(dyt, pitch, chunk, None, 0, n, k, C=part, ldc=k, batch=(S, 1), sA=(n * pitch, 0), sC=(n * k, 0), split=split, overflow=overflow,
                    scale=out_scale, terms=terms)          # (out_scale [k]: a per-column factor on the products, e.g. the inverse of a power of two dy was scaled by)
        if keep_parts:
            return part
        outs.append(part.sum(dim=0) if S > 1 else part[0])
    dW = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    if colsum:
        return dW, (csum.float() if direct else csum.sum(dim=0).float())
    return dW


def batched_a_times_x(A, x, overflow=None, out_scale=None, out=None):
    """A [B, N, N] (row-major), x [B*N, D] -> [B*N, D] with out[b] = A[b] x[b], on the fp16x3 engine: x[b]^T becomes a split fragment image per batch
    (ogmm_pack_frag_t with one chunk per batch: the weight gradient's operand packer), A is read as it lies.  N % 64 == 0."""
    B, N, _ = A.shape
    D = x.shape[1]
    assert A.is_contiguous() and x.stride(1) == 1 and x.shape[0] == B * N and N % 64 == 0
    pitch = N + 64
    n_pad = (D + 255) // 256 * 256
    hi = torch.empty(B * n_pad * pitch, dtype=torch.float16, device=A.device)
    lo = torch.empty_like(hi)
    _lib.call("ogmm_pack_frag_t", _p(_f32(x, "x")), x.stride(0), B * N, D, N, pitch, B, n_pad, _p(hi), _p(lo), _p(overflow), _p(None), _p(None), 0, 0, _stream())
    if out is None:
        out = torch.empty((B * N, D), dtype=torch.float32, device=A.device)
    assert tuple(out.shape) == (B * N, D) and out.is_contiguous()
    split = {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": PREC_F16X3_FRAG, "ldb_h": pitch, "sB": n_pad * pitch}
    gemm_nt(A, N, N, None, 0, N, D, C=out, ldc=D, batch=(B, 1), sA=(N * N, 0), sC=(N * D, 0), split=split, overflow=overflow, scale=out_scale)
    return out


def kabsch_bwd(src, corr, w, gR, gt):
    """backward of `kabsch` (layout [B,3,J]): -> (g_src, g_corr [B,3,J], g_w [B,J])"""
    B, _, J = src.shape
    src, corr, w = _f32(src, "src").contiguous(), _f32(corr, "corr").contiguous(), _f32(w, "w").reshape(B, J).contiguous()
    g_src, g_corr = torch.empty_like(src), torch.empty_like(corr)
    g_w = torch.empty((B, J), dtype=torch.float32, device=src.device)
    gRc = None if gR is None else _f32(gR, "gR").contiguous()          # named: must outlive the launch (see icp_point_to_point)
    gtc = None if gt is None else _f32(gt, "gt").contiguous()
    _lib.call("ogmm_kabsch_bwd", _p(src), _p(corr), _p(w), B, J, _p(gRc), _p(gtc), _p(g_src), _p(g_corr), _p(g_w), _stream())
    return g_src, g_corr, g_w


TOPK_ROWS_MAX_N = 20224          # ogmm_topk_rows: the row's candidates live in LDS (158 KiB / 8 B)


def topk_rows(v, k, largest=True):
    """torch.topk(v, k, dim=-1, largest)[1] for v [rows, n] with the reference CPU kernel's choice among tied values (ogmm_topk_rows) -> int64 [rows, k]"""
    assert v.dim() == 2 and v.stride(1) == 1
    rows, n = v.shape
    idx = torch.empty((rows, k), dtype=torch.int32, device=v.device)
    _lib.call("ogmm_topk_rows", _p(_f32(v, "v")), v.stride(0), rows, n, k, 1 if largest else 0, _p(idx), _stream())
    return idx.long()


def nearest_point(xyz, mu):
    C, N, _ = xyz.shape
    J = mu.shape[1]
    near = torch.empty((C, J), dtype=torch.int32, device=xyz.device)
    xyz, mu = _f32(xyz, "xyz"), _f32(mu, "mu").contiguous()
    _lib.call("ogmm_nearest_point", _p(xyz), _p(mu), C, N, J, _p(near), _stream())
    return near


def edge_features(xyz, idx):
    C, N, k = idx.shape
    out = torch.empty((C * N * k, 6), dtype=torch.float32, device=xyz.device)
    _lib.call("ogmm_edge_features", _p(_f32(xyz, "xyz")), _p(_i32(idx, "idx")), C, N, k, _p(out), _stream())
    return out


def pos_features(xyz, idx, centroid):
    C, N, k = idx.shape
    d2 = torch.empty((C * N, 1), dtype=torch.float32, device=xyz.device)
    alpha = torch.empty((C * N * k, 1), dtype=torch.float32, device=xyz.device)
    xyz, idx, centroid = _f32(xyz, "xyz"), _i32(idx, "idx"), _f32(centroid, "centroid").contiguous()
    _lib.call("ogmm_pos_features", _p(xyz), _p(idx), C, N, k, _p(centroid), _p(d2), _p(alpha), _stream())
    return d2, alpha


def scatter_add_rows_(out, rows, g):
    """out[rows[i]] += g[i] in place (kernel T10; rows int64 [n], g [n, D] rows contiguous, out [R, D])"""
    assert out.stride(1) == 1 and g.stride(1) == 1 and rows.dtype == torch.int64 and rows.is_contiguous() and g.shape[0] == rows.shape[0]
    _lib.call("ogmm_scatter_add_rows", _p(_f32(out, "out")), out.stride(0), out.shape[0], _p(rows), _p(_f32(g, "g")), g.stride(0), rows.shape[0], out.shape[1], _stream())
    return out


def small_bmm_nn(S, X, bias=None, out=None):
    """out[b] = S[b] X[b] (+ bias): S [B, R, m] and X [B, m, D] as ANY strided views (a transposed view costs nothing), contraction m <= 1024  -> [B, R, D]
    (kernel T10, exact fp32 fmaf chains; the thin / tiny products the training step used to hand to torch.matmul: include/ogmm_hip.h)."""
    B, R, m = S.shape
    D = X.shape[2]
    assert X.shape[0] == B and X.shape[1] == m and _f32(S, "S") is S and _f32(X, "X") is X
    if out is None:
        out = torch.empty((B, R, D), dtype=torch.float32, device=S.device)
    assert out.stride(2) == 1 and out.shape == (B, R, D)
    if bias is not None:
        bias = _f32(bias, "bias").contiguous()
    _lib.call("ogmm_small_bmm_nn", _p(S), S.stride(0), S.stride(1), S.stride(2), _p(X), X.stride(0), X.stride(1), X.stride(2), _p(bias), B, R, m, D,
              _p(out), out.stride(0), out.stride(1), _stream())
    return out


def small_bmm_nt(A, Bm, alpha=1.0):
    """out[b] = alpha A[b] Bm[b]^T: A [B, n, D], Bm [B, m, D] (rows contiguous) -> [B, n, m]   (kernel T10)"""
    B, n, D = A.shape
    m = Bm.shape[1]
    assert Bm.shape[0] == B and Bm.shape[2] == D and A.stride(2) == 1 and Bm.stride(2) == 1
    out = torch.empty((B, n, m), dtype=torch.float32, device=A.device)
    _lib.call("ogmm_small_bmm_nt", _p(_f32(A, "A")), A.stride(0), A.stride(1), _p(_f32(Bm, "B")), Bm.stride(0), Bm.stride(1), B, n, m, D, float(alpha),
              _p(out), out.stride(0), out.stride(1), _stream())
    return out


def l2norm_rows_bwd(x, g):
    assert x.stride(1) == 1 and g.stride(1) == 1
    dx = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.call("ogmm_l2norm_rows_bwd", _p(_f32(x, "x")), x.stride(0), _p(_f32(g, "g")), g.stride(0), x.shape[0], x.shape[1], _p(dx), dx.stride(0), _stream())
    return dx


def similarity(fn, B, N, split=True, overflow=None):
    """S[b] = fn_src[b] fn_tgt[b]^T for the stacked, channel-normalised map fn [(2B*N), D] -> [B,N,N]  (models/gmmreg.py:75)"""
    D = fn.shape[1]
    S = torch.empty((B, N, N), dtype=torch.float32, device=fn.device)
    if split and D % 64 == 0:
        img = pack_frag_batched(fn[B * N:], B, N)
        gemm_nt(fn, D, D, None, D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sC=(N * N, 0), split=img, overflow=overflow)
    else:
        gemm_nt(fn, D, D, fn[B * N:], D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sB=(N * D, 0), sC=(N * N, 0))
    return S


def overlap_cross_train(S, ol):
    """ol [(2B*N), 1] logits -> (wo [(2B*N), 1], stats [B,4,N])"""
    B, N, _ = S.shape
    wo = torch.empty((2 * B * N, 1), dtype=torch.float32, device=S.device)
    stats = torch.empty((B, 4, N), dtype=torch.float32, device=S.device)
    ws = _overlap_ws(B, N, S.device)
    _lib.call("ogmm_overlap_cross_ws", _p(_f32(S, "S")), B, N, _p(_f32(ol, "ol")), _p(ol[B * N:]), 1, _p(wo), _p(wo[B * N:]), 1, _p(stats), _p(ws), _stream())
    return wo, stats


def overlap_cross_bwd(S, ol, wo, stats, g_wo):
    """-> (dS [B,N,N], g_ol [(2B*N), 1])"""
    B, N, _ = S.shape
    dS = torch.empty_like(S)
    g_ol = torch.empty((2 * B * N, 1), dtype=torch.float32, device=S.device)
    g_wo = _f32(g_wo, "g_wo").contiguous()
    _lib.call("ogmm_overlap_cross_bwd", _p(S), B, N, _p(ol), _p(ol[B * N:]), 1, _p(wo), _p(wo[B * N:]), 1, _p(stats), _p(g_wo), _p(g_wo[B * N:]), 1,
              _p(dS), _p(g_ol), _p(g_ol[B * N:]), 1, _stream())
    return dS, g_ol


def weight_grad_thin_supported(dy, x):
    n, k = dy.shape[1], x.shape[1]
    if n > 256 and k <= 64 and n % 256 == 0:
        return weight_grad_thin_supported(dy[:, :256], x)
    nv, kv = (4 if n > 64 else (2 if n > 32 else 1)), (2 if k > 32 else 1)
    return (_lib.load().ogmm_weight_grad_thin_streams(n, k) > 0 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % nv == 0
            and x.stride(0) % kv == 0 and dy.data_ptr() % (4 * nv) == 0 and x.data_ptr() % (4 * kv) == 0)


DW_THIN_F16X3 = True      # 0: the exact-fp32 thin reduction in the fp16x3 training step too (A/B)


def weight_grad_thin(dy, x, split=False, overflow=None):
    """dW = dY^T X for thin layers (kernel T9): dy [R, n], x [R, k] -> [n, k].  split=False: exact fp32; split=True: the engines' fp16x3 arithmetic
    (the fp16x3 training step: dy carries the trainer's power-of-two loss scale, `overflow` reports operands beyond binary16's range)."""
    R, n = dy.shape
    k = x.shape[1]
    if n > 256 and k <= 64 and n % 256 == 0:          # a wide layer's few-channel input piece (conv2.net.0's two overlap channels): 256 outputs at a time
        return torch.cat([weight_grad_thin(dy[:, c0:c0 + 256], x, split, overflow) for c0 in range(0, n, 256)], dim=0)
    streams = _lib.load().ogmm_weight_grad_thin_streams(n, k)
    part = torch.empty((streams, n, k), dtype=torch.float32, device=dy.device)
    if split and DW_THIN_F16X3:
        _lib.call("ogmm_weight_grad_thin_f16x3", _p(_f32(dy, "dy")), dy.stride(0), _p(_f32(x, "x")), x.stride(0), R, n, k, _p(part), _p(overflow), _stream())
    else:
        _lib.call("ogmm_weight_grad_thin", _p(_f32(dy, "dy")), dy.stride(0), _p(_f32(x, "x")), x.stride(0), R, n, k, _p(part), _stream())
    return part.sum(dim=0)
