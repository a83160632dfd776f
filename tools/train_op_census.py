"""Which tensor expressions of one EAGER training step launch the small library kernels: every aten op dispatched during Trainer.step, counted per
(op, first calling frame inside ogmm_amd/).  Ops on tensors below 64 K elements are the launch-bound ones (a few microseconds of GPU time each, ~7 us of
stream time); the census names the call sites worth a fused kernel.
usage: python3 tools/train_op_census.py [pairs]"""
import collections
import os
import sys
import traceback
from argparse import Namespace

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in ("view", "detach", "alias", "as_strided", "expand", "slice", "select", "t.default", "transpose", "unsqueeze", "squeeze", "permute",
                                   "reshape", "empty", "_unsafe_view", "stride", "size", "is_", "record_stream", "unbind", "split")):
            return out
        numel = max([a.numel() for a in list(args) + ([out] if torch.is_tensor(out) else []) if torch.is_tensor(a)] or [0])
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "/ogmm_amd/" in fr.filename and "train_op_census" not in fr.filename:
                site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                break
        self.count[(name, "small" if numel < 65536 else "large", site)] += 1
        return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    from ogmm_amd import synth
    from ogmm_amd.gmmreg import GMMReg
    from ogmm_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
    model = GMMReg(512, 16, cfg).to(dev).train()
    synth.fill_state_dict(model.state_dict())
    batch = [t.to(dev) for t in synth.make_train_batch(4000, B, 1024, "partial")]
    starts = synth.fps_starts_for(4000, B, 1024)
    trainer = Trainer(model, dist=None, world=1, graph=False)
    for _ in range(2):
        trainer.step(*batch, fps_starts=starts)
    torch.cuda.synchronize()
    with Census() as c:
        trainer.step(*batch, fps_starts=starts)
    torch.cuda.synchronize()
    tot = sum(c.count.values())
    print("# %d aten ops with a launch in one eager step of %d pairs (views / allocations not counted); by (op, size class, call site)" % (tot, B))
    by_site = collections.Counter()
    for (name, size, site), n in c.count.items():
        by_site[(site, size)] += n
    print("## by call site")
    for (site, size), n in by_site.most_common(60):
        print("%5d  %-5s  %s" % (n, size, site))
    print("## by op")
    by_op = collections.Counter()
    for (name, size, site), n in c.count.items():
        by_op[(name, size)] += n
    for (name, size), n in by_op.most_common(40):
        print("%5d  %-5s  %s" % (n, size, name))


if __name__ == "__main__":
    main()
