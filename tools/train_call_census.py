"""Which C-ABI entry points one eager training step calls, how often and on what shapes (the integer arguments of every call): names the launches a fused
kernel would remove.  usage (GPU box): python3 tools/train_call_census.py [pairs] [name-filter ...]"""
import sys, os, collections; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import _lib, synth
from ogmm_amd.gmmreg import GMMReg
from ogmm_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flt = sys.argv[2:] or ["norm_bwd", "add_n", "pack_frag_t", "weight_grad_thin"]
dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.make_cfg(16)); synth.fill_state_dict(model.state_dict()); model = model.to(dev); model.precision = "f16x3"
batch = [t.to(dev) for t in synth.make_train_batch(0, B, 1024, "partial")]
starts = synth.fps_starts_for(0, B, 1024)
tr = Trainer(model, graph=False)
for _ in range(2):
    tr.step(*batch, fps_starts=starts)
count = collections.Counter(); names = collections.Counter()
real = _lib.call
def spy(name, *a):
    names[name] += 1
    if any(f in name for f in flt):
        count[(name, tuple(int(x) for x in a if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 31)))] += 1
    return real(name, *a)
_lib.call = spy
import ogmm_amd.ops as ops_mod
tr.step(*batch, fps_starts=starts)
torch.cuda.synchronize()
_lib.call = real
print("## entry points per step")
for k, v in names.most_common(): print("%5d  %s" % (v, k))
print("## shapes (integer arguments below 2^31)")
for (n, a), v in sorted(count.items(), key=lambda kv: (kv[0][0], -kv[1])): print("%3d  %-28s %s" % (v, n, a))
