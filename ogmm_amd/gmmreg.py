"""Drop-in `GMMReg` for gfmei/ogmm's models/gmmreg.py:32-119, running on the HIP kernels of libogmm_hip.so.

Same constructor (`GMMReg(emb_dims, n_clusters, config)`, config needs gnn_k / num_heads / km_clusters /
overlap_radius), same call (`model(src, tgt, is_test=False)` with src, tgt float32 [B,3,N]), same 5-tuple
`(rot [B,3,3], trans [B,3], src_o [B,N], tgt_o [B,N], loss [])`, same 153 state_dict keys so the reference's
`optim_model.pt` / `model_XXXX.pt` load with `load_state_dict`.

What runs where: this file is host orchestration only (the reference's forward is Python too).  Every tensor
operation of the forward is a kernel of libogmm_hip.so reached through ogmm_amd/ops.py; PyTorch provides
device buffers and the stream.  There is no CPU fallback: CPU tensors raise.

Scope: `model.eval()` runs the fused inference kernels; `model.train()` runs the training graph (batch-statistics
BatchNorm, autograd; ogmm_amd/train_graph.py).  `is_test=True` adds the point-to-point ICP refinement of
models/gmmreg.py:115-117 as a GPU kernel (open3d's published algorithm; parity unpinned, see oracle/icp_oracle.py).

Layout: clouds are stacked as C = 2B (src clouds, then tgt clouds: in eval mode the shared-weight src/tgt calls of
the reference are independent, models/gmmreg.py:52-53) and feature maps are point-major [C*N, channels].
"""
import math

import torch
from torch import nn

from . import ops
from ._lib import OgmmError
from .ops import ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SIGMOID

BN_EPS = 1e-5
# Measured budget (HISTORY.md section 4 "Per-layer term budget").  Round 4: an entry stays only if the layer's rounding holds the 1e-5 bar on BOTH weight
# families of the parity suite -- the closed-form default fill AND synth.fill_state_dict(profile="sharp") (peaked attention, saturated overlap scores).
# Round 3's entries for conv2.0 / conv2.3 (weight rounded), the three Q projections and the attention's score product (both rounded) were measured on
# the default fill only, where the attention is uniform to 1e-4 and every overlap score is 0.496 +- 0.003: on the sharp family each of them alone moves
# R by 1e-5 ... 5e-4 (tools/term_budget.py --profile sharp; profiles/round4_term_budget.txt).  What survives is the N x N cosine similarity: its
# entries lie in [-1, 1] and pass an un-tempered softmax, on either family one binary16 term moves R by ~1e-6 (the noise level of three terms).
TERM_BUDGET = {"similarity": 1}
TERM_BUDGET_ROUND3 = {"conv2.0": 2, "conv2.3": 2, "similarity": 1,
                      **{"%s.%s" % (t_, l_): 1 for t_ in ("sattn1", "cattn", "sattn2") for l_ in ("q", "qk")}}          # kept for A/B records only: NOT parity-safe


# ------------------------------------------------------------------------------------------ parameters
def state_spec(D=512):
    """(key, shape) for the reference's 153 state_dict entries, in its registration order."""
    spec = []

    def conv(name, cout, cin, nd, bias):
        spec.append((name + ".weight", (cout, cin) + (1,) * nd))
        if bias:
            spec.append((name + ".bias", (cout,)))

    def bn(name, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            spec.append((name + "." + leaf, (c,)))
        spec.append((name + ".num_batches_tracked", ()))

    for i, (co, ci) in enumerate(((64, 6), (64, 64), (128, 64), (256, 128), (D, 512)), 1):   # models/dgcnn.py:121-125
        conv("emd.conv%d" % i, co, ci, 2, False)
    for i, c in enumerate((64, 64, 128, 256, D), 1):                                            # models/dgcnn.py:126-130
        bn("emd.bn%d" % i, c)
    conv("proj.net.0", D // 2, D, 1, True); bn("proj.net.1", D // 2); conv("proj.net.3", 1, D // 2, 1, True)   # gmmreg.py:36
    for name, cin, hid, cout in (("overlap", D, D // 2, 1), ("conv1", D, 2 * D, D), ("conv2", D + 2, 2 * D, D)):  # gmmreg.py:37-39
        conv(name + ".net.0", hid, cin, 1, True); bn(name + ".net.1", hid)
        conv(name + ".net.3", hid, hid, 1, True); bn(name + ".net.4", hid)
        conv(name + ".net.6", cout, hid, 1, True)
    conv("pos.conv_dis.0", 64, 1, 1, False); bn("pos.conv_dis.1", 64)                          # models/attn.py:34-57
    conv("pos.conv_dis.3", D // 2, 64, 1, False); bn("pos.conv_dis.4", D // 2)
    conv("pos.conv_ang1.0", 64, 1, 2, False); bn("pos.conv_ang1.1", 64)
    conv("pos.conv_ang2.0", D // 2, 64, 1, False); bn("pos.conv_ang2.1", D // 2)
    conv("pos.conv.0", D, D, 1, False); bn("pos.conv.1", D)                                     # defined, never applied
    for name in ("sattn1", "cattn", "sattn2"):                                                  # models/attn.py:85-107
        conv(name + ".attn.merge", D, D, 1, True)
        for i in range(3):
            conv(name + ".attn.proj.%d" % i, D, D, 1, True)
        conv(name + ".mlp.0", 2 * D, 2 * D, 1, True)
        conv(name + ".mlp.3", D, 2 * D, 1, True)
    return spec


class _Node(nn.Module):
    """Name-only container so that parameter paths reproduce the reference's state_dict keys."""


def _build_tree(root, spec):
    for key, shape in spec:
        *path, leaf = key.split(".")
        node = root
        for part in path:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        if leaf in ("running_mean", "running_var"):
            node.register_buffer(leaf, torch.zeros(shape) if leaf == "running_mean" else torch.ones(shape))
        elif leaf == "num_batches_tracked":
            node.register_buffer(leaf, torch.zeros((), dtype=torch.long))
        else:
            node.register_parameter(leaf, nn.Parameter(torch.empty(shape)))


def _default_init(module, spec):
    """PyTorch's own defaults for Conv (kaiming-uniform a=sqrt(5) <=> U(+-1/sqrt(fan_in))) and BatchNorm (1, 0),
    plus the zero bias of models/attn.py:107."""
    sd = dict(module.named_parameters())
    shapes = dict(spec)
    with torch.no_grad():
        for key, p in sd.items():
            base, leaf = key.rsplit(".", 1)
            is_bn = (base + ".running_var") in shapes
            if is_bn:
                p.fill_(1.0 if leaf == "weight" else 0.0)
            else:
                w_shape = shapes[base + ".weight"]
                bound = 1.0 / math.sqrt(max(1, int(torch.tensor(w_shape[1:]).prod())))
                p.uniform_(-bound, bound)
        for name in ("sattn1", "cattn", "sattn2"):
            if name + ".mlp.3.bias" in sd:
                sd[name + ".mlp.3.bias"].zero_()


# ------------------------------------------------------------------------------------------ weight packing
def _fold_bn(sd, bn, conv_bias=None):
    """eval BatchNorm (+ preceding conv bias) as y = acc * scale + shift, computed in fp64 and rounded once."""
    w, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
    mean, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    s = w / torch.sqrt(var + BN_EPS)
    t = b - mean * s
    if conv_bias is not None:
        t = t + conv_bias.double() * s
    return s.float().contiguous(), t.float().contiguous()


def _w2d(sd, key, pad_to=None):
    w = sd[key + ".weight"]
    w = w.reshape(w.shape[0], w.shape[1]).float()
    if pad_to is not None and w.shape[1] < pad_to:
        w = torch.cat([w, w.new_zeros(w.shape[0], pad_to - w.shape[1])], dim=1)
    return w.contiguous()


def _add_splits(L):
    """Adds the pre-split binary16 weights (ops.split_f16) to every layer that owns a 2-D `W` with K >= 32."""
    for v in L.values():
        if isinstance(v, dict):
            if "W" in v and torch.is_tensor(v["W"]) and v["W"].dim() == 2 and v["W"].shape[1] >= 32:
                K = v["W"].shape[1]
                v["split"] = ops.split_f16(v["W"], frag=True, k1=v.get("k1", K))
            else:
                _add_splits(v)


def pack_weights(sd, D, H):
    """state_dict (tensors already on the target device) -> dict of kernel-ready layers.  Done once per load:
    BN folded to scale/shift, conv2.net.0 zero-padded from 514 to 516 input channels (16-byte rows), attention
    projections re-ordered head-major (reference channel c = d*H + h, models/attn.py:96 -> c' = h*dh + d) with the
    merge convolution's input columns permuted to match."""
    L = {}
    dh = D // H

    def conv_bn(key, bn, has_bias, pad_to=None):
        s, t = _fold_bn(sd, bn, sd[key + ".bias"] if has_bias else None)
        return {"W": _w2d(sd, key, pad_to), "scale": s, "shift": t}

    def conv_bias(key):
        return {"W": _w2d(sd, key), "shift": sd[key + ".bias"].float().contiguous()}

    for i in range(1, 6):
        L["emd%d" % i] = conv_bn("emd.conv%d" % i, "emd.bn%d" % i, False)
    s, t = _fold_bn(sd, "pos.conv_dis.1")
    sa, ta = _fold_bn(sd, "pos.conv_ang1.1")
    L["pos"] = {"w_dis": sd["pos.conv_dis.0.weight"].reshape(-1).float().contiguous(), "s_dis": s, "t_dis": t,
                "w_ang": sd["pos.conv_ang1.0.weight"].reshape(-1).float().contiguous(), "s_ang": sa, "t_ang": ta}
    L["pos_dis2"] = conv_bn("pos.conv_dis.3", "pos.conv_dis.4", False)
    L["pos_ang2"] = conv_bn("pos.conv_ang2.0", "pos.conv_ang2.1", False)
    cprime = torch.arange(D, device=sd["emd.conv1.weight"].device)
    old_of_new = (cprime % dh) * H + (cprime // dh)
    for name in ("sattn1", "cattn", "sattn2"):
        T = {}
        for i, tag in enumerate(("q", "k", "v")):
            w = _w2d(sd, "%s.attn.proj.%d" % (name, i))
            b = sd["%s.attn.proj.%d.bias" % (name, i)].float()
            T[tag] = {"W": w[old_of_new].contiguous(), "shift": b[old_of_new].contiguous()}
        T["kv"] = {"W": torch.cat([T["k"]["W"], T["v"]["W"]], 0).contiguous(), "shift": torch.cat([T["k"]["shift"], T["v"]["shift"]]).contiguous()}
        wm = _w2d(sd, name + ".attn.merge")
        T["merge"] = {"W": wm[:, old_of_new].contiguous(), "shift": sd[name + ".attn.merge.bias"].float().contiguous()}
        T["mlp0"] = conv_bias(name + ".mlp.0")
        # merge conv folded into the MLP's first conv (SURVEY.md section 7 "legal flop savings"): with msg = o Wm^T + bm,
        #   mlp0([x | msg]) = x W0a^T + o (W0b Wm)^T + (W0b bm + b0); the product of the two weight matrices is formed in
        # fp64 and rounded once, so the result differs from the two-step evaluation only by fp32 rounding (parity-tested).
        w0 = T["mlp0"]["W"].double()
        T["mlp0_folded"] = {"W": torch.cat([w0[:, :D], w0[:, D:] @ T["merge"]["W"].double()], 1).float().contiguous(),
                            "shift": (T["mlp0"]["shift"].double() + w0[:, D:] @ T["merge"]["shift"].double()).float().contiguous()}
        T["mlp3"] = conv_bias(name + ".mlp.3")
        L[name] = T
    for name in ("conv1", "conv2", "overlap"):
        cin = sd[name + ".net.0.weight"].shape[1]
        # conv2.net.0 has 514 input channels: [f | wo, o].  The two extra ones travel as a second A piece of 32 columns (30 of them zero, the
        # weights zero-padded to match): the LDS-DMA GEMM engine walks K in steps of 32, and a 16 MB all-but-two-columns-zero operand is cheaper than
        # a slower engine for a 1024 x 514 layer.
        S = {"0": conv_bn(name + ".net.0", name + ".net.1", True, pad_to=(cin if cin % 32 == 0 else cin // 32 * 32 + 32)),
             "3": conv_bn(name + ".net.3", name + ".net.4", True)}
        if sd[name + ".net.6.weight"].shape[0] == 1:
            S["6"] = {"w": sd[name + ".net.6.weight"].reshape(-1).float().contiguous(), "b": sd[name + ".net.6.bias"].float().contiguous()}
        else:
            S["6"] = conv_bias(name + ".net.6")
        L[name] = S
    # conv2.net.6 (1024 -> 512, no activation) feeds nothing but overlap.net.0 (512 -> 256) (models/gmmreg.py:83-85: `fo` has no other use), so the two
    # linear maps are one 1024 -> 256 layer: W = W_o0 W_c6, b = W_o0 b_c6 + b_o0, formed in fp64 and rounded once (0.40 of 11.2 GMAC per cloud less;
    # the result differs from the two-step evaluation by fp32 rounding only, as with the folded merge convolution).
    w46 = _w2d(sd, "overlap.net.0").double() @ _w2d(sd, "conv2.net.6").double()
    b46 = _w2d(sd, "overlap.net.0").double() @ sd["conv2.net.6.bias"].double() + sd["overlap.net.0.bias"].double()
    s46, t46 = _fold_bn(sd, "overlap.net.1", b46)
    L["conv2_6_overlap_0"] = {"W": w46.float().contiguous(), "scale": s46, "shift": t46}
    L["proj"] = {"0": conv_bn("proj.net.0", "proj.net.1", True),
                 "3": {"w": sd["proj.net.3.weight"].reshape(-1).float().contiguous(), "b": sd["proj.net.3.bias"].float().contiguous()}}
    _add_splits(L)
    return L


# ------------------------------------------------------------------------------------------ the module
class GMMReg(nn.Module):
    def __init__(self, emb_dims, n_clusters, config):
        super().__init__()
        if emb_dims % config.num_heads != 0 or (emb_dims // config.num_heads) % 4 != 0 or emb_dims % 64 != 0:
            raise ValueError("emb_dims must be a multiple of 64 and of 4*num_heads")
        self.emb_dims, self.n_clusters, self.config = emb_dims, n_clusters, config
        spec = state_spec(emb_dims)
        _build_tree(self, spec)
        _default_init(self, spec)
        self._packed = None
        self._packed_key = None
        self.last_intermediates = None
        # "f16x3": weight GEMMs on the binary16 matrix cores with two-term operand splitting (fp32-class accuracy);
        # "f32": everything on the exact-fp32 MFMA engine;
        # "f16": REDUCED precision for BASELINE configs[2] (quoted in bf16): the large GEMMs multiply only the leading binary16
        #        terms (11-bit mantissa >= bf16's 8, fp32 accumulate); R / t then agree with the reference to ~1e-4, not 1e-5.
        self.precision = getattr(config, "precision", "f16x3")
        # Per-layer term budget of the fp16 split engine (struct ogmm_gemm.terms): layer -> 2 runs that layer with the WEIGHT operand rounded to
        # binary16 ((a_hi + a_lo) w_hi: two matrix instructions per product block instead of three, x0.74-0.81 of the layer's time), 1 with both
        # operands rounded (a_hi w_hi: x0.5-0.63).  Only layers whose
        # rounding was measured to leave (R, t) within the parity bar are listed -- on the CPU oracle with the same rounding (tools/term_budget.py)
        # and on the GPU's parity distribution (tools/parity_distribution.py); HISTORY.md section 4 has the table.  {} = three terms everywhere.
        self.term_budget = dict(TERM_BUDGET)
        self.sinkhorn_thresh = 1e-2      # lib/utils.py:73 (`thresh` default, which wkeans_plus :281 does not override); <= 0 runs every sweep
        self.fold_merge = True      # evaluate merge(attn) inside mlp.0 (one GEMM less per transformer)
        self.fold_conv2_overlap = True      # conv2.net.6 and overlap.net.0 (two linear maps in a row) as one 1024 -> 256 layer
        self.fuse_overlap = True            # overlap block's softmax-dots in the similarity GEMM's epilogue where the engine takes it (ops.overlap_fusable)
        self._overflow = None          # device int32[1]: the engines' fp16 range flag ...
        self._status = None            # ... and its neighbour word: protocol errors of kernels with bounded on-chip / cross-workgroup waits (_lib.STATUS_*)
        self._overflow_host = self._overflow_event = None
        self._overflow_pending = False
        self.overflow_policy = "deferred"          # what an fp16 range overflow in an eval forward does: _post_overflow_check
        self._side = None
        self._side2 = None
        self._ws = None             # persistent zero-initialised buffers of the eval forward (_workspace)
        self._capture_ws = None     # ... and the set capture_graph() prepares for the forward it records
        self._swap = None           # cloud map of the cross-attention (src <-> tgt), per batch size
        self._head = None
        # Opt-in for serving loops: the head of a forward (cloud stacking, kNN graph + positional front end, FPS chains: everything that depends on the
        # inputs alone) is queued on its own streams WITHOUT waiting for what the current stream still has to run, so that it overlaps the latency-bound tail
        # of the previous forward.  Contract when True: `src` / `tgt` (and `fps_starts`) must be COMPLETE when forward() is called -- not pending on the
        # current stream -- e.g. inputs resident from an earlier synchronisation, or produced on another stream the caller has waited on.
        # (First built and withdrawn in the first half of round 5: the FPS chains were not reproducible beside the previous forward's GEMMs.  That was the
        # packed-fp32 hazard -- HISTORY.md section 4 -- and is gone with it: tests/test_hip_forward.py::test_pipelined_head_...)
        self.pipeline_head = False
        self._train_ops = None      # tests inject the plain-PyTorch operation set (tests/train_ref.py) to check the graph wiring on CPU

    # -- packed-weight cache: rebuilt when any parameter/buffer was modified or moved.  (data_ptr, _version) catches optimizer steps,
    # load_state_dict, .to(); in-place edits through `.data` (EMA swaps, manual copies) do NOT bump _version, so the key also carries a
    # content fingerprint -- one device-side reduction over all tensors, re-checked every `fingerprint_every` forwards -- and train() / eval() /
    # load_state_dict() invalidate explicitly.  Call `invalidate_packed()` after editing weights through `.data` when the next forward must see it.
    fingerprint_every = 64

    def invalidate_packed(self):
        self._packed = None
        self._packed_key = None

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    @staticmethod
    def _fingerprint(tensors):
        """Two multi-tensor launches (L1 and L2 norms of every parameter / buffer) + a handful of small ones: a per-tensor python loop costs ~700
        launches (2.5 ms) here, which showed up in the step time."""
        with torch.no_grad():
            ts = [t.detach() for t in tensors if t.is_floating_point()]
            n1 = torch.stack(torch._foreach_norm(ts, 1)).double()
            n2 = torch.stack(torch._foreach_norm(ts, 2)).double()
            w = torch.arange(1, len(ts) + 1, dtype=torch.float64, device=n1.device)          # position weights: swapping two tensors' contents is seen too
            return torch.stack([(n1 * w).sum(), (n2 * w).sum()])

    def _layers(self):
        sd = self.state_dict()
        key = tuple((t.data_ptr(), t._version) for t in sd.values())
        self._fp_calls = getattr(self, "_fp_calls", 0) + 1
        stale = self._packed is None or key != self._packed_key
        if not stale and self.fingerprint_every and self._fp_calls % self.fingerprint_every == 0:
            stale = not torch.equal(self._fingerprint(list(sd.values())), self._packed_fp)          # one host sync every `fingerprint_every` forwards
        if stale:
            self._packed = pack_weights(sd, self.emb_dims, self.config.num_heads)
            self._packed_key = key
            self._packed_fp = self._fingerprint(list(sd.values()))
        return self._packed

    def _workspace(self, dev, stream, C, N, D, XW):
        """Persistent zero-initialised buffers of the eval forward, one set per (stream, shape) -- a model that is driven from two streams at once gets
        two sets -- at most four sets are kept.  `clean` is False while a forward is between its first accumulation and its last finalize: a forward that
        raised in between leaves statistics behind, and the next one re-zeroes them."""
        if torch.cuda.is_current_stream_capturing():
            # Under stream capture nothing may be created or trusted here: a torch.zeros would be RECORDED (and its 16.8 MB fill replayed with every graph
            # launch), and a `clean` flag set by an earlier capture on the same capture stream says nothing about memory the graph has not run on yet.
            # capture_graph() hands in a set it created and zeroed eagerly, owns for the graph's lifetime and never shares.
            ws = self._capture_ws
            if ws is None or ws["key"] != (str(dev), C, N, D, XW):
                raise OgmmError("eval forward under stream capture without a prepared workspace: use GMMReg.capture_graph()")
            return ws
        key = (str(dev), stream.cuda_stream, C, N, D, XW)
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 4:
                self._ws.pop(next(iter(self._ws)))
            ws = self._ws[key] = {"stats3": torch.zeros((3, C, 2 * D, 2), dtype=torch.float64, device=dev),
                                  "extra": torch.zeros((C * N, XW), dtype=torch.float32, device=dev), "clean": True}
        elif not ws["clean"]:
            ws["stats3"].zero_()
            ws["extra"][:, 2:].zero_()
        ws["clean"] = False
        return ws

    def _transformer(self, eng, L, x, anchor_feats, anchor_ids, C, N, res, cloud_map=None, stats=None, q_terms=0, kv_terms=0, qk_terms=0):
        """models/attn.py:78-111: mlp(cat[x, merge(softmax(q k^T / sqrt(dh)) v)]) (+ res).  x [C*N, D]; the anchors [C, M, D] are rows
        anchor_ids [C, M] of anchor_feats [C*N, D] (of the cloud cloud_map[c], if given): lib/utils.py:111-127."""
        D, H = self.emb_dims, self.config.num_heads
        dh, M = D // H, anchor_ids.shape[1]
        dev = x.device
        q = ops.conv1x1(x, L["q"], eng=eng, terms=q_terms)
        if ops.attention_supported(M, dh):
            kv = ops.conv1x1_gathered(anchor_feats, C, N, anchor_ids, L["kv"], cloud_map=cloud_map, eng=eng, terms=kv_terms)      # keys | values in one GEMM, rows gathered by its DMA
            o = ops.attention(q, kv[:, :D], kv[:, D:], C, N, M, H, qk_terms=qk_terms if eng.split else 0)
            if self.fold_merge:
                mlp0, msg = L["mlp0_folded"], o                       # merge conv folded into mlp0's weights
            else:
                mlp0, msg = L["mlp0"], ops.conv1x1(o, L["merge"], eng=eng)
            if eng.split and ops.instnorm_fusable(mlp0.get("split"), N):
                # InstanceNorm fused: statistics in mlp0's epilogue, normalise + ReLU while mlp3 stages its A operand
                own = stats is None          # (the forward hands in a slice of its persistent, self-cleaning workspace: _workspace)
                if own:
                    stats = torch.zeros((C, 2 * D, 2), dtype=torch.float64, device=dev)
                z = ops.conv1x1(x, mlp0, x2=msg, col_stats=stats, group_rows=N, eng=eng)
                a_sc, a_sh = ops.instnorm_finalize(stats, N, BN_EPS, clear=not own)
                return ops.conv1x1(z, L["mlp3"], res=res, a_affine=(a_sc, a_sh, True), group_rows=N, eng=eng)
            z = ops.conv1x1(x, mlp0, x2=msg, eng=eng)
            ops.instnorm_relu_(z, C, N, BN_EPS)
            return ops.conv1x1(z, L["mlp3"], res=res, eng=eng)
        anchors = ops.gather_rows(anchor_feats, D, C, N, D, anchor_ids, cloud_map=cloud_map)
        kk = ops.conv1x1(anchors.view(C * M, D), L["k"], eng=eng)
        vT = torch.empty((C, D, M), dtype=torch.float32, device=dev)            # V^T per cloud: rows = head-major channels
        ops.gemm_nt(L["v"]["W"], D, D, anchors, D, D, M, C=vT, ldc=M, shift=L["v"]["shift"], row_affine=True,
                    batch=(C, 1), sB=(M * D, 0), sC=(D * M, 0))
        S = torch.empty((C, H, N, M), dtype=torch.float32, device=dev)
        ops.gemm_nt(q, D, dh, kk, D, N, M, C=S, ldc=M, alpha=1.0 / dh ** .5, batch=(C, H),
                    sA=(N * D, dh), sB=(M * D, dh), sC=(H * N * M, N * M))
        ops.softmax_rows_(S.view(C * H * N, M))
        o = torch.empty((C * N, D), dtype=torch.float32, device=dev)
        ops.gemm_nt(S, M, M, vT, M, N, dh, C=o, ldc=D, batch=(C, H), sA=(H * N * M, N * M), sB=(D * M, dh * M), sC=(N * D, dh))
        msg = ops.conv1x1(o, L["merge"], eng=eng)
        z = ops.conv1x1(x, L["mlp0"], x2=msg, eng=eng)
        ops.instnorm_relu_(z, C, N, BN_EPS)
        return ops.conv1x1(z, L["mlp3"], res=res, eng=eng)

    @staticmethod
    def _stack3(eng, S, x, x2=None):
        """models/dgcnn.py:19-28 (`CONV`, used='proj') with a Cout > 1 last layer."""
        h = ops.conv1x1(x, S["0"], ACT_RELU, x2=x2, eng=eng)
        h = ops.conv1x1(h, S["3"], ACT_RELU, eng=eng)
        return ops.conv1x1(h, S["6"], eng=eng)

    def forward(self, src, tgt, is_test=False, fps_starts=None, capture=False):
        """models/gmmreg.py:50-119.  `fps_starts` (int64/int32 [6,B], optional) pins the six `torch.randint` draws of
        lib/utils.py:190; when None they are drawn from torch's global CPU generator in the reference's call order,
        so the same `torch.manual_seed` gives the same anchors as the reference."""
        if not (isinstance(src, torch.Tensor) and isinstance(tgt, torch.Tensor)):
            raise OgmmError("src and tgt must be tensors")
        if not (src.is_cuda and tgt.is_cuda) and not (self.training and self._train_ops is not None):
            raise OgmmError("GMMReg.forward needs CUDA/ROCm tensors: the MI355X path has no CPU fallback")
        if src.dim() != 3 or src.shape[1] != 3 or src.shape != tgt.shape or src.dtype != torch.float32 or tgt.dtype != torch.float32:
            raise OgmmError("src and tgt must be float32 [B,3,N] of equal shape (models/gmmreg.py:79-80 needs N_src == N_tgt)")
        cfg = self.config
        B, _, N = src.shape
        C, D, H, k, M, J = 2 * B, self.emb_dims, cfg.num_heads, cfg.gnn_k, cfg.km_clusters, self.n_clusters
        if M % 4 != 0 or M > N or J > N or k > N:
            raise OgmmError("km_clusters must be a multiple of 4 and km_clusters, n_clusters, gnn_k <= N")
        if not (4 <= k <= 32) or J > 128:
            # the kernels' documented ceilings, stated here instead of surfacing as a launch error from deep inside the forward: the kNN lists and the per-edge
            # pooling are built for 4 ... 32 neighbours (the reference uses 20), the matching kernel for at most 128 components per cloud (the reference 16 ... 128)
            raise OgmmError("this build supports 4 <= gnn_k <= 32 and n_clusters <= 128 (got gnn_k=%d, n_clusters=%d)" % (k, J))
        dev = src.device
        if self.precision not in ("f16x3", "f32", "f16"):
            raise OgmmError("precision must be 'f16x3', 'f32' or 'f16' (reduced: single binary16 term in the large GEMMs)")
        if dev.type == "cuda":
            self.overflow_flag(dev)
        if self.emd.conv1.weight.device != dev:
            raise OgmmError("GMMReg parameters live on %s but the inputs on %s: call model.to(device) first" % (self.emd.conv1.weight.device, dev))
        if self.training:
            return self._forward_train(src, tgt, fps_starts, capture, is_test)
        self._raise_pending_overflow()
        L = self._layers()
        cap = {} if capture else None
        eng = ops.Engine(self.precision, self._overflow)          # this model's engine choice travels with every call: no process-wide switch
        tb = self.term_budget if self.precision == "f16x3" else {}

        if fps_starts is None:
            fps_starts = torch.stack([torch.randint(0, N, (B,), dtype=torch.long) for _ in range(6)])
        main = torch.cuda.current_stream()
        fused_head = ops.knn_pos_head_supported(N, k)
        # `pipeline_head` (opt-in, see __init__): the head of THIS forward -- anchor draws, cloud stacking, kNN + positional front end, FPS chains; all of it
        # depends on nothing but the inputs -- goes to its own streams WITHOUT waiting for the work the current stream still holds, so in a loop of forwards
        # it runs under the tail of the previous one (cluster means, matching, loss: ~0.2 ms on a mostly idle chip) and beside its last GEMMs.
        pipelined = bool(self.pipeline_head) and fused_head and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing()
        if pipelined:
            if self._head is None or self._head[0].device != dev:
                self._head = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            hs, hs_fps = self._head
        else:
            hs = hs_fps = main
        with torch.cuda.stream(hs):
            # [stage][src clouds | tgt clouds] on the device.  Host draws travel through PINNED memory with a non-blocking copy: `.to(device)` of a pageable
            # tensor makes the host wait for the copy, which is queued behind everything the stream still has to run (the caching host allocator keeps the
            # pinned block alive until the copy has run).
            if fps_starts.device.type == "cpu" and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                fps_starts = fps_starts.reshape(3, 2 * B).to(torch.int32).pin_memory().to(dev, non_blocking=True)
            else:
                fps_starts = fps_starts.reshape(3, 2 * B).to(device=dev, dtype=torch.int32).contiguous()
            xyz = ops.pack_clouds(src, tgt)          # [C,N,3] (src clouds, then tgt clouds): one launch
            inputs_ready = torch.cuda.Event()          # what the FPS chains wait for: the stacked clouds and the anchor draws, NOT the kNN kernel behind them
            inputs_ready.record(hs)
            # The head (round 5): the 20-NN graph, the positional encoding's 5-NN graph and its hidden maps come out of ONE launch (ops.knn_pos_head: the
            # 5-NN set is the head of the sorted 20-NN list, with its own rank-5 tie resolution, and the front end needs only the cloud the kernel already
            # holds) -- two kernels less that had to be on the chip before the persistent EdgeConv kernel starts.  Only the FPS chains stay on a side stream.
            if fused_head:
                idx, idx5, hd, ha = ops.knn_pos_head(xyz, k, L["pos"])
        if self._swap is None or self._swap.device != dev or self._swap.numel() != C:          # (cached per batch size: four tiny launches per forward otherwise)
            self._swap = torch.cat([torch.arange(B, C, device=dev), torch.arange(0, B, device=dev)]).to(torch.int32)      # built on the device: capturable
        swap = self._swap
        # Latency-bound selection kernels (one workgroup per cloud: FPS chains, later the E/M loop) run on a side stream next to the GEMM-bound main
        # stream: they occupy <= C of the 256 CUs.  Every tensor they touch stays referenced until the streams are joined again.  Everything queued here
        # has to be ON the chip before the persistent EdgeConv kernel starts (~0.25 ms into the forward): that kernel holds every CU's registers and LDS
        # for ~0.7 ms, and a selection kernel that is still queued then runs after it, next to -- and slowing -- the first GEMM.
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        side = self._side
        side.wait_event(inputs_ready)
        if self._side2 is None or self._side2.device != dev:
            self._side2 = torch.cuda.Stream(device=dev)
        side2 = self._side2
        side2.wait_event(inputs_ready)
        R = C * N
        XW = L["conv2"]["0"]["W"].shape[1] - D                                   # conv2 input channels 512 (wo), 513 (o), zero pad to the packed width
        # the three transformers' InstanceNorm statistics and the [wo | o | pad] piece of conv2.net.0: persistent per (stream, shape), zeroed when created.
        # Round 4 re-allocated and re-zeroed both per forward: the 16.8 MB fill of `extra` (30 of its 32 columns are constant zero) ran for 660 us beside
        # the EdgeConv kernel.  Now `extra`'s pad columns are zeroed once (the forward writes columns 0 and 1 only) and ogmm_instnorm_finalize zeroes the
        # statistics behind its read.
        ws = self._workspace(dev, main, C, N, D, XW)
        stats3, extra = ws["stats3"], ws["extra"]
        with torch.cuda.stream(side2):
            ids_a = ops.fps(xyz, M, fps_starts)                                   # [3,C,M]: all three random-start samplings at once
        # (beside the anchor chains, not behind them: the EdgeConv kernel waits for both.  Pipelined: on a stream of its own -- `side` still holds the previous
        #  forward's E/M and clustering loss, behind which the sampling would not run ahead)
        if pipelined:
            hs_fps.wait_event(inputs_ready)
        with torch.cuda.stream(hs_fps if pipelined else (side if fused_head else side2)):
            ids_j = ops.fps(xyz, J, None)                                         # centre-start sampling for the GMM init
        if not fused_head:
            with torch.cuda.stream(side):
                idx5 = ops.knn(xyz, 5)        # its own top-k call in the reference (lib/utils.py:52): ties at rank 5 resolve independently
                hd, ha = ops.pos_hidden(xyz, idx5, 5, L["pos"])      # positional front end (models/attn.py:65-73): needs only xyz and the 5-NN graph
        fps_done, side_done = torch.cuda.Event(), torch.cuda.Event()
        fps_done.record(side2)
        side_done.record(hs_fps if pipelined else side)
        if not fused_head:
            idx = ops.knn(xyz, k)
        # The FPS chains run BESIDE the kNN kernel and must be through before the persistent EdgeConv kernel takes every CU (queued behind the kNN kernel they ran
        # next to EdgeConv instead: slower for both).  The wait costs nothing: they finish with the kNN kernel.
        main.wait_event(fps_done)          # (each side stream directly: a hand-over through a second stream costs another ~15 us)
        main.wait_event(side_done)
        xyz.record_stream(side2)
        xyz.record_stream(side)
        fps_starts.record_stream(side2)
        for t_ in (ids_a, ids_j) + (() if fused_head else (idx5, hd, ha)):
            t_.record_stream(main)
        if pipelined:
            main.wait_stream(hs)
            xyz.record_stream(hs_fps)
            for t_ in (xyz, idx, idx5, hd, ha):
                t_.record_stream(main)
            src.record_stream(hs)
            tgt.record_stream(hs)

        # ---- DGCNN (models/dgcnn.py:133-154)
        R = C * N
        xcat = torch.empty((R, 512), dtype=torch.float32, device=dev)
        emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
        if eng.split and ops.edgeconv_fused_supported(k, emd):
            ops.edgeconv_fused(xyz, idx, emd, xcat, status=self._status)                    # per-edge tensors stay on chip
        else:
            h = ops.edgeconv_first(xyz, idx, L["emd1"], xcat[:, 0:64])
            h = ops.edgeconv_layer(h, L["emd2"], k, xcat[:, 64:128], eng=eng)
            h = ops.edgeconv_layer(h, L["emd3"], k, xcat[:, 128:256], eng=eng)
            ops.edgeconv_layer(h, L["emd4"], k, xcat[:, 256:512], store=False, eng=eng)
            del h
        emb = ops.conv1x1(xcat, L["emd5"], ACT_RELU, eng=eng)

        # ---- positional encoding added to the embedding (models/attn.py:59-75, gmmreg.py:58-61)
        x0 = torch.empty((R, D), dtype=torch.float32, device=dev)
        ops.conv1x1(hd, L["pos_dis2"], ACT_LEAKY02, out=x0[:, :D // 2], res=emb[:, :D // 2], eng=eng)
        ops.conv1x1(ha, L["pos_ang2"], ACT_LEAKY02, out=x0[:, D // 2:], res=emb[:, D // 2:], eng=eng)

        # ---- self-attention 1 + conv1 (gmmreg.py:54-57, 62-63)
        t1 = self._transformer(eng, L["sattn1"], x0, emb, ids_a[0], C, N, res=x0, stats=stats3[0], q_terms=tb.get("sattn1.q", 0), kv_terms=tb.get("sattn1.kv", 0), qk_terms=tb.get("sattn1.qk", 0))
        ft = self._stack3(eng, L["conv1"], t1)
        # ---- cross-attention: keys/values are the OTHER cloud's anchors (gmmreg.py:67-72)
        f = self._transformer(eng, L["cattn"], ft, ft, ids_a[1], C, N, res=ft, cloud_map=swap, stats=stats3[1], q_terms=tb.get("cattn.q", 0), kv_terms=tb.get("cattn.kv", 0), qk_terms=tb.get("cattn.qk", 0))

        # ---- overlap scores (gmmreg.py:74-89)
        ops.conv1x1_head(f, L["proj"]["0"], ACT_RELU, L["proj"]["3"]["w"], L["proj"]["3"]["b"], ACT_NONE, extra[:, 1], ldy=XW, eng=eng,
                         terms=tb.get("proj.0", 0))      # proj.0 + proj.3
        if self.fuse_overlap and D % 64 == 0 and ops.overlap_fusable(B, N, D, eng):
            # the N x N similarity never leaves the GEMM's accumulators: its epilogue forms the partial softmax-dots (struct ogmm_gemm.ovl_rowpart)
            tgt_img = ops.l2norm_pack_frag_batched(f[B * N:], B, N, rnorm_of=f[:B * N])        # B operand: the tgt half, normalised and split in one pass; the same launch leaves the src half's 1 / |row|
            ops.overlap_fused(f[:B * N], tgt_img, B, N, D, extra[:B * N, 1], extra[B * N:, 1], XW, extra[:B * N, 0], extra[B * N:, 0], XW,
                              overflow=self._overflow, terms=tb.get("similarity", 0))
        else:
            S = torch.empty((B, N, N), dtype=torch.float32, device=dev)
            if eng.split and D % 64 == 0:
                fn_src = ops.l2norm_rows(f[:B * N])                            # A operand: the src half, normalised
                tgt_img = ops.l2norm_pack_frag_batched(f[B * N:], B, N)
                ops.gemm_nt(fn_src, D, D, None, D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sC=(N * N, 0), split=tgt_img, overflow=self._overflow,
                            single_term=eng.single_term)
            else:
                fn = ops.l2norm_rows(f)
                ops.gemm_nt(fn, D, D, fn[B * N:], D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sB=(N * D, 0), sC=(N * N, 0))
            ops.overlap_cross(S, extra[:B * N, 1], extra[B * N:, 1], XW, extra[:B * N, 0], extra[B * N:, 0], XW)
            del S
        if self.fold_conv2_overlap:
            h2 = ops.conv1x1(ops.conv1x1(f, L["conv2"]["0"], ACT_RELU, x2=extra, eng=eng, terms=tb.get("conv2.0", 0)), L["conv2"]["3"], ACT_RELU, eng=eng,
                             terms=tb.get("conv2.3", 0))
            g = ops.conv1x1(h2, L["conv2_6_overlap_0"], ACT_RELU, eng=eng, terms=tb.get("conv2.6+overlap.0", 0))            # conv2.net.6 and overlap.net.0 as one layer (pack_weights)
        else:
            fo = self._stack3(eng, L["conv2"], f, x2=extra)
            g = ops.conv1x1(fo, L["overlap"]["0"], ACT_RELU, eng=eng)
        o = torch.empty((C, N), dtype=torch.float32, device=dev)
        ops.conv1x1_head(g, L["overlap"]["3"], ACT_RELU, L["overlap"]["6"]["w"], L["overlap"]["6"]["b"], ACT_SIGMOID, o, ldy=1, eng=eng,
                         terms=tb.get("overlap.3", 0))      # overlap.3 + overlap.6

        # ---- GMM E/M (needs only xyz and the overlap scores) on the side stream, next to self-attention 2 (gmmreg.py:92-101)
        def run_em():
            # thresh / group_size: the reference's Sinkhorn early exit (lib/utils.py:99-102), per call batch -- the B src clouds and the B tgt
            # clouds are separate wkeans_plus calls (models/gmmreg.py:100-101).  capture=True also records every sweep's residual and the
            # number of sweeps every E-step ran: see sinkhorn_exit_margin()
            return ops.gmm_em(xyz, o, ids_j, iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=self.sinkhorn_thresh, group_size=B,
                              return_resid=capture, return_sweeps=capture, status=self._status)
        # beside the whole last transformer (measured alternatives -- serial behind it, a high-priority stream, queued behind the attention kernel -- were all slower
        # or equal: HISTORY.md "Where the E/M runs")
        side.wait_stream(main)
        with torch.cuda.stream(side):
            em = run_em()
            em_done = torch.cuda.Event()
            em_done.record(side)
        o.record_stream(side)
        for t_ in em[:3]:
            t_.record_stream(main)
        gamma, pi, mu = em[:3]
        f2 = self._transformer(eng, L["sattn2"], f, f, ids_a[2], C, N, res=f, stats=stats3[2], q_terms=tb.get("sattn2.q", 0), kv_terms=tb.get("sattn2.kv", 0), qk_terms=tb.get("sattn2.qk", 0))
        main.wait_event(em_done)

        # ---- cluster features, matching, rigid solve, clustering loss (gmmreg.py:100-114)
        muf = ops.gmm_feat_mean(gamma, pi, f2, C, N)
        # the clustering loss (nearest point, InfoNCE rows, mean: three small launches) next to the matching / rigid solve (one 64-workgroup
        # launch): both are latency-bound tails on an otherwise idle chip
        side.wait_stream(main)
        with torch.cuda.stream(side):
            row_loss, near = ops.clu_infonce(xyz, mu, f2, muf, C, N, 0.1)
            loss = row_loss.mean()          # 0.5 * (mean over src rows + mean over tgt rows), gmmreg.py:110
            clu_done = torch.cuda.Event()
            clu_done.record(side)
        for t_ in (f2, muf, mu):
            t_.record_stream(side)
        for t_ in (row_loss, near, loss):
            t_.record_stream(main)
        rot, trans = ops.match_kabsch(mu[:B].contiguous(), mu[B:].contiguous(), muf[:B].contiguous(), muf[B:].contiguous(), 0.05)
        main.wait_event(clu_done)

        if capture:
            cap.update(knn_idx=idx, fps_anchor=ids_a, fps_J=ids_j, emb=emb, x0=x0, ft=ft, f=f, f2=f2, wo=extra[:, 0].clone(), o_logit=extra[:, 1].clone(),          # (clones: `extra` is the persistent workspace, the next forward overwrites it)
                      
                       o=o, gamma=gamma, pi=pi, mu=mu, muf=muf, near=near, row_loss=row_loss, sinkhorn_resid=em[3], sinkhorn_sweeps=em[4])
            self.last_intermediates = cap
        ws["clean"] = True
        if is_test:
            # models/gmmreg.py:115-117: point-to-point ICP from the network's motion, correspondence radius 2 * overlap_radius
            # (lib/o3dutils.py:176).  The reference hands every pair to open3d on the CPU; here the whole batch stays on the GPU.
            rot, trans = ops.icp_point_to_point(xyz[:B], xyz[B:], rot, trans, 2.0 * cfg.overlap_radius)
        self._post_overflow_check(dev)
        return rot, trans, o[:B], o[B:], loss

    # ---- binary16 range: an activation beyond +-65504 is clamped by the fp16x3 engines and raises a device-side flag.  Results built on a clamped value
    # must not pass silently (a trained checkpoint with large activations): `overflow_policy`
    #   "deferred" (default)  the flag travels to pinned host memory behind the forward (asynchronous: no host synchronisation in a serving loop) and is
    #                         looked at when it has arrived -- at the latest at the start of the NEXT forward or in fp16_overflowed() -- and raises then;
    #   "sync"                every forward waits for its own flag and raises itself (one host synchronisation per call);
    #   "ignore"              poll fp16_overflowed() yourself.
    def _post_overflow_check(self, dev):
        if self._overflow is None or self.overflow_policy == "ignore" or torch.cuda.is_current_stream_capturing():
            return
        if self.overflow_policy == "sync":
            if self.fp16_overflowed():
                raise OgmmError(self._OVERFLOW_TEXT % "this")
            return
        if self._overflow_host is None:
            self._overflow_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._overflow_event = torch.cuda.Event()
        elif self._overflow_pending and not self._overflow_event.query():
            self._overflow_event.synchronize()          # (two forwards in flight share one host buffer: the older one first)
        self._raise_pending_overflow()
        self._overflow_host.copy_(self._flags, non_blocking=True)          # range flag and protocol status word in one copy
        self._flags.zero_()
        self._overflow_event.record()
        self._overflow_pending = True

    _STATUS_TEXT = ("kernel protocol error in %s forward (status word %d: 2 = EdgeConv producer / consumer hand-over, 4 = E/M early-exit group wait): a bounded "
                    "wait between waves / workgroups ran into its limit -- the outputs of that forward are NaN-poisoned, not the reference's.")
    _OVERFLOW_TEXT = ("fp16x3 engine: an activation beyond +-65504 was clamped in %s forward -- its results are not the reference's.  "
                      "Run this model with precision='f32' (exact fp32 engine) or rescale the inputs; overflow_policy='ignore' restores polling.")

    def _raise_pending_overflow(self, wait=False):
        if not self._overflow_pending or torch.cuda.is_current_stream_capturing():          # (an event query would invalidate a stream capture)
            return
        if wait:
            self._overflow_event.synchronize()
        if self._overflow_event.query():
            self._overflow_pending = False
            ovf, st = int(self._overflow_host[0]), int(self._overflow_host[1])
            if ovf or st:
                self._overflow_host.zero_()
                raise OgmmError(self._STATUS_TEXT % ("an earlier", st) if st else self._OVERFLOW_TEXT % "an earlier")

    def sinkhorn_exit_margin(self):
        """After forward(..., capture=True): the smallest batch-mean Sinkhorn residual of the call, per call group (src clouds, tgt clouds) as the
        reference forms it (lib/utils.py:99-101: mean over the clouds of sum |u - u0| + sum |v - v0|), divided by its exit threshold.
        <= 1: some E-step left its sweeps early (as the reference does: `last_intermediates["sinkhorn_sweeps"]` has the counts, [2, iters]);
        close to 1: the decision is a knife edge, and rounding differences between this path and the reference may flip it."""
        r = self.last_intermediates["sinkhorn_resid"]          # [2B, iters, sweeps], NaN for sweeps that did not run
        B = r.shape[0] // 2
        means = torch.stack([r[:B].mean(0), r[B:].mean(0)])    # [2, iters, sweeps]
        means = torch.where(torch.isnan(means), torch.full_like(means, float("inf")), means)
        return float(means.min().item()) / (self.sinkhorn_thresh if self.sinkhorn_thresh and self.sinkhorn_thresh > 0 else 1e-2)

    def _forward_train(self, src, tgt, fps_starts, capture, is_test=False):
        """`.train()` mode: batch-statistics BatchNorm with running-stat updates and autograd through every differentiable
        stage (ogmm_amd/train_graph.py over the kernels of ogmm_amd/train_ops.py)."""
        from . import train_graph, train_ops
        B, _, N = src.shape
        if fps_starts is None:
            fps_starts = torch.stack([torch.randint(0, N, (B,), dtype=torch.long) for _ in range(6)])
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        cap = {} if capture else None
        if self.precision not in ("f16x3", "f32"):
            raise OgmmError("training runs with precision 'f16x3' or 'f32' (the reduced 'f16' mode is inference only)")
        backend = self._train_ops if self._train_ops is not None else train_ops.TrainOps(self.precision, self._overflow, self._status)
        out = train_graph.forward_train(backend, P, self.config, self.n_clusters, src, tgt, fps_starts.to(src.device), cap)
        if capture:
            self.last_intermediates = cap
        if is_test:                                            # models/gmmreg.py:115-117 (no gradient through open3d there either)
            rot, trans = ops.icp_point_to_point(src.transpose(1, 2).contiguous(), tgt.transpose(1, 2).contiguous(),
                                                out[0].detach(), out[1].detach(), 2.0 * self.config.overlap_radius)
            out = (rot, trans) + tuple(out[2:])
        return out

    def capture_graph(self, batch, n_points, device=None):
        """HIP-graph version of the eval forward for one (batch, n_points) shape: every kernel launch of the forward -- about 140,
        on two streams -- is recorded once and replayed with a single hipGraphLaunch, which removes the host-side launch cost that
        dominates small batches (B = 1: the GPU work is < 1 ms).  Returns `run(src, tgt, fps_starts=None) -> the usual 5-tuple`;
        the outputs are views of the graph's static buffers and are overwritten by the next call."""
        if self.training:
            raise OgmmError("capture_graph is for eval mode")
        from . import graph_replay_safe
        if not graph_replay_safe():
            raise OgmmError("capture_graph: the HIP runtime was initialised before ogmm_amd could set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (replays mixed with "
                            "other launches fault on this ROCm with the runtime's graph packet capture on): export it, or import ogmm_amd before touching the GPU")
        dev = torch.device(device) if device is not None else self.emd.conv1.weight.device
        s_src = torch.zeros((batch, 3, n_points), dtype=torch.float32, device=dev)
        s_tgt = torch.zeros_like(s_src)
        s_starts = torch.zeros((6, batch), dtype=torch.long, device=dev)
        src0, tgt0, _, _ = __import__("ogmm_amd.synth", fromlist=["make_batch"]).make_batch(0, batch, n_points, "partial")
        s_src.copy_(src0)
        s_tgt.copy_(tgt0)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm), torch.no_grad():          # warm-up on a side stream: lazy one-time setup (function attributes,
            for _ in range(2):                                  # packed weights, allocator pools) must not happen during capture
                self.forward(s_src, s_tgt, fps_starts=s_starts)
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        self._raise_pending_overflow(wait=True)          # the warm-up forwards' range check, before anything is recorded
        # the recorded forward's persistent buffers: created and zeroed HERE (eagerly, before anything is recorded), owned by the returned closure -- not an entry of
        # the per-stream cache, so no other forward, capture or eviction ever touches memory a live graph addresses
        D, C = self.emb_dims, 2 * batch
        XW = self._layers()["conv2"]["0"]["W"].shape[1] - D
        ws = {"stats3": torch.zeros((3, C, 2 * D, 2), dtype=torch.float64, device=dev),
              "extra": torch.zeros((C * n_points, XW), dtype=torch.float32, device=dev), "clean": True, "key": (str(dev), C, n_points, D, XW)}
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        self._capture_ws = ws
        try:
            with torch.cuda.graph(graph), torch.no_grad():
                outs = self.forward(s_src, s_tgt, fps_starts=s_starts)
        finally:
            self._capture_ws = None

        def run(src, tgt, fps_starts=None):
            if tuple(src.shape) != (batch, 3, n_points) or tuple(tgt.shape) != (batch, 3, n_points):
                raise OgmmError("graph captured for [%d,3,%d] inputs, got %s" % (batch, n_points, tuple(src.shape)))
            if fps_starts is None:
                fps_starts = torch.stack([torch.randint(0, n_points, (batch,), dtype=torch.long) for _ in range(6)])
            s_src.copy_(src, non_blocking=True)
            s_tgt.copy_(tgt, non_blocking=True)
            s_starts.copy_(fps_starts.reshape(6, batch), non_blocking=True)
            graph.replay()
            return outs
        run.graph = graph
        run.workspace = ws          # (keeps the graph's statistics / conv2 side-input buffers alive as long as the closure)
        return run

    def overflow_flag(self, device=None):
        """the device int32[1] the fp16 engines raise when |value| > 65504 was clamped (created on first use); callers that want to read it without a
        host round trip per query (the trainer) snapshot it with device ops"""
        dev = torch.device(device) if device is not None else self.emd.conv1.weight.device
        if dev.type == "cuda" and (self._overflow is None or self._overflow.device != dev):
            self._flags = torch.zeros(2, dtype=torch.int32, device=dev)          # one buffer: both words travel to the host in one copy
            self._overflow, self._status = self._flags[0:1], self._flags[1:2]
        return self._overflow

    def fp16_overflowed(self):
        """True if any fp16x3 GEMM since the last call clamped an activation beyond +-65504 (synchronises)."""
        if self._overflow is None:
            return False
        ovf, st = self._flags.tolist()
        self._flags.zero_()
        if self._overflow_pending:          # a deferred check that has not been looked at yet
            self._overflow_event.synchronize()
            self._overflow_pending = False
            ovf, st = ovf or int(self._overflow_host[0]), st or int(self._overflow_host[1])
            self._overflow_host.zero_()
        if st:
            raise OgmmError(self._STATUS_TEXT % ("a", st))
        return bool(ovf)
