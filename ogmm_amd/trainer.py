"""One optimisation step of train.py:36-74, one process per GPU.

Reference semantics reproduced (train.py:185-200, nn.DataParallel):
  * Adam(lr, weight_decay=1e-4) on all parameters (`pos.conv.*` never receive a gradient and are skipped, as there);
  * every replica normalises BatchNorm with the statistics of ITS shard (no SyncBN) and the running statistics of
    replica 0 are the ones that survive the step -> rank 0's buffers are broadcast after each step;
  * the loss is built from the gathered outputs, i.e. dcp / overlap-MSE / Welsch are means over the GLOBAL batch, while
    the clustering loss is the SUM of the per-replica means (`clu_loss.sum()`, train.py:63-64).  With equal shards the
    global loss is (1/W) sum_r [10 dcp_r + mse_r + 0.01 welsch_r] + sum_r clu_r, so each rank back-propagates
    rest_r / W + clu_r and the gradients are SUM-all-reduced.
Gradients travel as ONE flat fp32 bucket (13.0 M elements, 52 MB) per step: a single RCCL all-reduce over xGMI, which is
per-link bound (a ring moves 2(W-1)/W x 52 MB per GPU), instead of one small collective per parameter.
"""
import torch

from . import losses, metric


def flatten_grads(params):
    """-> (flat fp32 bucket, the parameters that own a gradient, in order)"""
    owners = [p for p in params if p.grad is not None]
    if not owners:
        return None, owners
    return torch.cat([p.grad.reshape(-1) for p in owners]), owners


def unflatten_into_grads(flat, owners):
    off = 0
    for p in owners:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def allreduce_gradients(model, dist, world):
    """SUM all-reduce of every gradient as one bucket.  Every rank must own gradients for the same parameters."""
    if dist is None:          # (a process group handed in at world 1 -- dist.init(..., force=True) -- runs the collective: the one-GPU test of this path)
        return 0
    flat, owners = flatten_grads(list(model.parameters()))
    if flat is None:
        return 0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    unflatten_into_grads(flat, owners)
    return flat.numel()


def broadcast_buffers(model, dist, world, src=0):
    """BatchNorm running statistics of rank `src` win (DataParallel re-broadcasts replica 0 every step)."""
    if dist is None:
        return
    bufs = [b for b in model.buffers() if b.is_floating_point()]
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.broadcast(flat, src=src)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()
    for b in model.buffers():
        if not b.is_floating_point():
            dist.broadcast(b, src=src)


class Trainer:
    def __init__(self, model, lr=1e-4, weight_decay=1e-4, welsch_alpha=10.0, welsch_top_k=512, dist=None, world=1, loss_scale=None, graph=False):
        """lr / alpha / top_k defaults: configs/cfgs.py:55,41,44; weight decay: train.py:199.
        graph=True: forward + loss + backward of a step -- about 2500 kernel launches, whose enqueueing on the host takes as long as the GPU needs to run
        them -- are recorded ONCE per input shape into a HIP graph (after `graph_warmup` eager steps) and replayed with one launch per step; the inputs
        travel through static buffers, the overflow flags, loss parts and outputs come back as static tensors, and everything the host decides (skip /
        loss-scale policy, gradient all-reduce, un-scaling, Adam) stays eager behind the replay.  Same kernels, same order: bit-identical steps.
        loss_scale: power of two the loss is multiplied by before backward (gradients are divided by it again before the
        optimizer; both exact in fp32).  It exists for the fp16x3 engine's backward GEMMs, whose operands are split into
        binary16 terms: unscaled activation gradients (1e-6 .. 0.2) would fall into binary16's subnormal range.  Default
        2^16 with precision "f16x3", 1 with "f32".  If a scaled gradient exceeds 65504 the engine raises its overflow flag:
        the step is skipped and the scale halved."""
        self.model, self.dist, self.world = model, dist, world
        self.alpha, self.top_k = welsch_alpha, welsch_top_k
        self.optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
        if loss_scale is None:
            loss_scale = 65536.0 if getattr(model, "precision", "f32") == "f16x3" else 1.0
        self.loss_scale = float(loss_scale)
        self.initial_loss_scale = self.loss_scale
        self.min_loss_scale = 1.0            # floor: halving below 1 cannot help (the scale only protects small gradients)
        self.growth_interval = 200           # clean steps after which a reduced scale doubles again (as torch's GradScaler)
        self._clean_steps = 0
        self.forward_overflows = 0           # CONSECUTIVE steps whose forward clamped an activation (reset by a clean step)
        self.max_forward_overflows = 8
        self.skipped_steps = 0
        if graph:
            from . import graph_replay_safe
            if not graph_replay_safe():
                raise RuntimeError("Trainer(graph=True): the HIP runtime of this process was initialised before ogmm_amd could set "
                                   "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; with the runtime's graph packet capture on, replays mixed with other launches end in a "
                                   "GPU memory fault on this ROCm.  Export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (or import ogmm_amd before touching the GPU).")
        self.graph, self.graph_warmup = bool(graph), 2
        self._g = None                       # the captured step: dict(key, graph, static inputs / outputs)
        self._eager_steps = 0

    def _overflow_hits(self, fwd, bwd):
        """(forward flag, backward flag) -> two bools, the same on every rank: ONE host read (and one MAX all-reduce) per step for both"""
        if fwd is None:
            return False, False
        both = torch.cat([fwd, bwd]).to(torch.float32)
        if self.dist is not None:
            self.dist.all_reduce(both, op=self.dist.ReduceOp.MAX)
        f, b, *st = both.tolist()
        if st and st[0] > 0:
            # the model's protocol status word (gmmreg._flags[1]: a bounded wait of the EdgeConv hand-over or of the E/M's early-exit group ran into its
            # limit): that step's pi / mu are NaN-poisoned and training_loss's nan_to_num would turn them into a silent zero-loss step -- raise instead
            raise FloatingPointError("kernel protocol error in a training step (status word %d: 2 = EdgeConv hand-over, 4 = E/M early-exit group wait); "
                                     "the step's outputs are NaN-poisoned" % int(st[0]))
        return f > 0, b > 0

    def _forward(self, src, tgt, fps_starts):
        return self.model(src, tgt, fps_starts=fps_starts)

    def _drop_grads(self):
        """a skipped step's gradients are not used.  Eager: dropped.  Recorded step: they are the graph's static tensors and must stay allocated (the next
        replay overwrites them) -- they are zeroed instead, so that nobody inspecting p.grad after a skipped step sees the clamped values"""
        if self._g is None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            grads = [p.grad for p in self.model.parameters() if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)

    def local_loss(self, out, src, tgt, transform_gt, src_overlap, tgt_overlap):
        loss, parts = losses.training_loss(out, src, tgt, transform_gt, src_overlap, tgt_overlap, self.alpha, self.top_k)
        if self.world > 1:      # this rank's share of the DataParallel loss (see the module docstring)
            loss = torch.nan_to_num((10 * parts["dcp"] + parts["mse"] + 0.01 * parts["welsch"]) / self.world + parts["clu"], nan=0.0)
        return loss, parts

    def _traced(self, src, tgt, transform_gt, src_overlap, tgt_overlap, fps_starts, scale):
        """forward + loss + backward: everything of a step that is pure device work (what graph=True records).  scale: python float or device scalar."""
        # The engine's overflow flag (device int32[1]) is shared by forward activations and backward gradients.  It is snapshotted with device ops
        # after the forward and after the backward and read ONCE, behind the backward: no extra host round trip between the two.
        flag = self.model.overflow_flag(src.device) if hasattr(self.model, "overflow_flag") and src.is_cuda else None
        flags = getattr(self.model, "_flags", None) if flag is not None else None          # [range flag, protocol status word]: both travel in the one read
        if flag is not None:
            (flags if flags is not None else flag).zero_()
        out = self._forward(src, tgt, fps_starts)
        fwd_flag = None
        if flag is not None:
            fwd_flag = flag.clone()
            flag.zero_()
        loss, parts = self.local_loss(out, src, tgt, transform_gt, src_overlap, tgt_overlap)
        (loss * scale).backward()
        # Detached: nothing a caller keeps from one step may hold that step's autograd graph alive into the next one.  (Not only tidiness: with the
        # previous step's outputs -- and through their grad_fn its graph -- still alive, ending the capture of a recorded step crashed inside the HIP
        # runtime on this ROCm; tools/train_capture_debug.py reproduces it.)
        return (tuple(o.detach() for o in out), loss.detach(), {k: v.detach() for k, v in parts.items()}, fwd_flag,
                ((flags if flags is not None else flag).clone() if flag is not None else None))

    def _replay(self, src, tgt, transform_gt, src_overlap, tgt_overlap, fps_starts):
        """graph=True: the recorded step on this batch (captured on first use per shape)"""
        B, _, N = src.shape
        if fps_starts is None:          # the reference's six torch.randint draws (lib/utils.py:190), on the host as the eager path makes them
            fps_starts = torch.stack([torch.randint(0, N, (B,), dtype=torch.long) for _ in range(6)])
        given = [src, tgt, transform_gt, src_overlap, tgt_overlap, fps_starts]
        key = tuple((tuple(t.shape), t.dtype) for t in given if t is not None)
        g = self._g
        if g is None or g["key"] != key:
            dev = src.device
            static = [None if t is None else torch.empty(t.shape, dtype=t.dtype, device=dev) for t in given]
            for st, t in zip(static, given):
                if st is not None:
                    st.copy_(t)
            scale_t = torch.full((), self.loss_scale, dtype=torch.float32, device=dev)
            self.optimizer.zero_grad(set_to_none=True)          # the recorded backward creates the gradients: static from here on
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                res = self._traced(*static, scale_t)
            g = self._g = {"key": key, "graph": graph, "static": static, "scale": scale_t, "res": res}
        for st, t in zip(g["static"], given):
            if st is not None:
                st.copy_(t, non_blocking=True)
        g["scale"].fill_(self.loss_scale)
        g["graph"].replay()
        # what the caller gets must not alias the graph's static tensors (the next replay overwrites them: a loop that collects info["loss"] per step
        # would read the LAST step's value for every entry); a handful of small clones, one tiny launch each
        out, loss, parts, fwd_flag, flag = g["res"]
        return tuple(o.clone() for o in out), loss.clone(), {k: v.clone() for k, v in parts.items()}, fwd_flag, flag

    def step(self, src, tgt, transform_gt, src_overlap, tgt_overlap, fps_starts=None):
        """train.py:53-74 for this rank's shard.  Returns loss (local share), the four parts, mean R / t errors."""
        self.model.train()
        self._backward_scale = self.loss_scale          # the scale this step's gradients carry (self.loss_scale may grow below)
        use_graph = self.graph and src.is_cuda and self._eager_steps >= self.graph_warmup
        if use_graph:
            out, loss, parts, fwd_flag, flag = self._replay(src, tgt, transform_gt, src_overlap, tgt_overlap, fps_starts)
        else:
            self._eager_steps += 1
            self._g = None
            self.optimizer.zero_grad(set_to_none=True)
            out, loss, parts, fwd_flag, flag = self._traced(src, tgt, transform_gt, src_overlap, tgt_overlap, fps_starts, self.loss_scale)
        fwd_hit, bwd_hit = self._overflow_hits(fwd_flag, flag)
        overflowed = fwd_hit or bwd_hit
        if fwd_hit:
            # |activation| > 65504 in the FORWARD: its outputs and every gradient derived from them are built on clamped values.  The loss scale cannot
            # cure that (it only lifts small gradients): skip the step without touching the scale, and give up when it persists.
            self.forward_overflows += 1
            self.skipped_steps += 1
            self._drop_grads()
            if self.forward_overflows > self.max_forward_overflows:
                raise FloatingPointError("fp16x3 engine: forward activations beyond +-65504 in %d consecutive steps -- the loss scale cannot fix this; "
                                         "use model.precision = 'f32' or rescale the inputs" % self.forward_overflows)
        elif bwd_hit:
            # a scaled gradient left binary16's range: skip, halve the scale (at scale 1 the step is still skipped: the gradient was clamped)
            self.forward_overflows = 0
            self.skipped_steps += 1
            self.loss_scale = max(self.min_loss_scale, self.loss_scale * 0.5)
            self._clean_steps = 0
            self._drop_grads()
        else:
            self.forward_overflows = 0
            self._clean_steps += 1
            if self.loss_scale < self.initial_loss_scale and self._clean_steps >= self.growth_interval:
                self.loss_scale = min(self.initial_loss_scale, self.loss_scale * 2.0)
                self._clean_steps = 0
            allreduce_gradients(self.model, self.dist, self.world)
            if self._backward_scale != 1.0:
                grads = [p.grad for p in self.model.parameters() if p.grad is not None]
                torch._foreach_mul_(grads, 1.0 / self._backward_scale)          # one multi-tensor launch instead of 150 small ones
            self.optimizer.step()
        broadcast_buffers(self.model, self.dist, self.world)
        with torch.no_grad():
            B = src.shape[0]
            r_err = metric.rotation_error(out[0], transform_gt[:, :3, :3]).mean()
            t_err = metric.translation_error(out[1], transform_gt[:, :3, 3].reshape(B, 3)).mean()
        return {"loss": loss, "skipped": overflowed, "parts": parts, "r_err_deg": r_err, "t_err": t_err, "out": out}


class BaselineTrainer(Trainer):
    """One optimisation step of the DeepGMR baseline's loop (train_base.py:27-75, :168): Adam(lr, weight_decay=1e-4), loss = `dcp_loss` of the model's two
    outputs with NaN -> 0 -- nothing else (no overlap, clustering or Welsch term).  Under DataParallel the loss is a mean over the gathered batch: each
    rank back-propagates its share / W and the gradients are SUM-all-reduced like the main model's; loss scaling and overflow handling as in `Trainer`."""

    def local_loss(self, out, src, tgt, transform_gt, src_overlap=None, tgt_overlap=None):
        B = out[0].shape[0]
        dcp = losses.dcp_loss(out[0], transform_gt[:, :3, :3], out[1], transform_gt[:, :3, 3].reshape(B, 3))
        return torch.nan_to_num(dcp / self.world, nan=0.0), {"dcp": dcp}

    def _forward(self, src, tgt, fps_starts):
        return self.model(src, tgt)

    def step(self, src, tgt, transform_gt, src_overlap=None, tgt_overlap=None, fps_starts=None):
        return super().step(src, tgt, transform_gt, src_overlap, tgt_overlap)
