import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import pack_weights, state_spec
sd = {k: torch.zeros(s, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32) for k, s in state_spec(512)}
synth.fill_state_dict(sd)
L = pack_weights({k: v.cuda() for k, v in sd.items()}, 512, 4)
C, N, k = 128, 1024, 20
src, tgt, _, _ = synth.make_batch(0, C // 2, N, "partial")
xyz = torch.cat([src, tgt], 0).transpose(1, 2).contiguous().cuda()
idx = ops.knn(xyz, k)
xcat = torch.empty((C * N, 512), device="cuda")
emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
for _ in range(2): ops.edgeconv_fused(xyz, idx, emd, xcat)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.edgeconv_fused(xyz, idx, emd, xcat)
e1.record(); torch.cuda.synchronize()
print("OGMM_EC_DBG=%s  %.3f ms" % (os.environ.get("OGMM_EC_DBG", "0"), e0.elapsed_time(e1) / 5))
