"""kNN graph kernel alone: 128 clouds x 1024 points, k = 20 and k = 5 (includes the tie-resolution pass)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, synth
src, tgt, _, _ = synth.make_batch(0, 64, 1024, "partial")
xyz = torch.cat([src, tgt], 0).transpose(1, 2).contiguous().cuda()
for k in (20, 5):
    for _ in range(2): ops.knn(xyz, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.knn(xyz, k)
    e1.record(); torch.cuda.synchronize()
    print("k=%d  %.1f us" % (k, e0.elapsed_time(e1) / 5 * 1e3))
