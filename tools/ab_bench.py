"""Same-box A/B of the eval forward (B = 64, N = 1024, J = 16): pairs/s for a list of model settings, interleaved, several rounds.
usage: ab_bench.py name=python-dict-of-attributes ...   e.g.  ab_bench.py base="{'term_budget': {}}" budget="{}" """
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg

B, N, J = 64, 1024, 16
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
settings = [(a.split("=", 1)[0], eval(a.split("=", 1)[1])) for a in sys.argv[1:]] or [("default", {})]
models = []
for name, attrs in settings:
    m = GMMReg(512, J, cfg)
    synth.fill_state_dict(m.state_dict())
    m = m.cuda().eval()
    m._ops_attrs = {k[4:]: v for k, v in attrs.items() if k.startswith("ops.")}          # module-level switches of ogmm_amd.ops, set before each of this model's runs
    for k, v in attrs.items():
        if not k.startswith("ops."):
            setattr(m, k, v)
    models.append((name, m))
src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
starts = synth.fps_starts_for(0, B, N)
src, tgt = src.cuda(), tgt.cuda()
res = {name: [] for name, _ in models}
with torch.no_grad():
    def switch(m):
        for k, v in m._ops_attrs.items():
            setattr(ops, k, v)
    for name, m in models:
        switch(m)
        for _ in range(3):
            m(src, tgt, fps_starts=starts)
    for rnd in range(5):
        for name, m in models:
            switch(m)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                m(src, tgt, fps_starts=starts)
            torch.cuda.synchronize()
            res[name].append(B * 20 / (time.perf_counter() - t0))
for name, v in res.items():
    print("%-24s pairs/s: %s   median %.0f" % (name, " ".join("%.0f" % x for x in v), sorted(v)[len(v) // 2]))
