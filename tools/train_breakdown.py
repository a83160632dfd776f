"""Times the pieces of one training step (forward and backward of every TrainOps method, dX / dW of the dense layers) with events."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth, train_ops, ops
from ogmm_amd.gmmreg import GMMReg
from ogmm_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
model = GMMReg(512, 16, cfg); synth.fill_state_dict(model.state_dict()); model = model.to(dev)
batch = [t.to(dev) for t in synth.make_train_batch(0, B, 1024)]
starts = synth.fps_starts_for(0, B, 1024)
tr = Trainer(model)
events = []

def timed(tag, fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record(); events.append((tag, e0, e1)); return r
    return w

for cls in (train_ops._Linear, train_ops._NormAct, train_ops._NormActPool, train_ops._NormLinear, train_ops._Attention, train_ops._MaxPoolK):
    f, b = cls.forward, cls.backward
    def mk(cls, f, b):
        def fw(ctx, *a):
            shp = tuple(a[0].shape)
            return timed("%s.fwd %s" % (cls.__name__, shp), f)(ctx, *a)
        def bw(ctx, *g):
            shp = next((tuple(t_.shape) for t_ in g if t_ is not None), ())
            return timed("%s.bwd %s" % (cls.__name__, shp), b)(ctx, *g)
        cls.forward, cls.backward = staticmethod(fw), staticmethod(bw)
    mk(cls, f, b)
for name in ("knn", "fps", "gmm_em", "nearest_point", "edge_features", "pos_features", "attention", "l2norm_rows", "overlap_cross",
             "gmm_feat_mean", "match_kabsch", "gather_points"):
    setattr(train_ops.TrainOps, name, timed(name + ".fwd", getattr(train_ops.TrainOps, name)))

for it in range(3):
    events.clear()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); tr.step(*batch, fps_starts=starts); t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for tag, e0, e1 in events:
    a = agg.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = t0.elapsed_time(t1)
print("step %.2f ms; instrumented %.2f ms" % (tot, sum(a[1] for a in agg.values())))
for tag, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-44s x%-3d %8.3f ms" % (tag, n, ms))
