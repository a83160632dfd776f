"""The kNN kernels alone at the headline shape (GPU box): ogmm_knn (k = 20), ogmm_knn (k = 5) and ogmm_pos_hidden as three launches against the fused head
kernel ogmm_knn_pos_head (scan B over scan A's marks; 5-NN graph and positional front end folded in).  usage: python3 tools/knn_time.py [C] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
src, tgt, _, _ = synth.make_batch(0, C // 2, N, "partial")
xyz = ops.pack_clouds(src.cuda(), tgt.cuda())
g = torch.Generator().manual_seed(0)
pos = {key: (torch.rand(64, generator=g) * 2 - 0.5).cuda() for key in ("w_dis", "s_dis", "t_dis", "w_ang", "s_ang", "t_ang")}


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


idx5 = ops.knn(xyz, 5)
t20, t5, tp = timed(lambda: ops.knn(xyz, 20)), timed(lambda: ops.knn(xyz, 5)), timed(lambda: ops.pos_hidden(xyz, idx5, 5, pos))
print("C=%d N=%d: ogmm_knn k=20 %.1f us + k=5 %.1f us + ogmm_pos_hidden %.1f us = %.1f us in three launches" % (C, N, t20, t5, tp, t20 + t5 + tp))
if ops.knn_pos_head_supported(N, 20):
    a, b5, hd, ha = ops.knn_pos_head(xyz, 20, pos)
    same = torch.equal(a, ops.knn(xyz, 20)) and torch.equal(b5, idx5) and all(torch.equal(x, y) for x, y in zip((hd, ha), ops.pos_hidden(xyz, idx5, 5, pos)))
    print("          ogmm_knn_pos_head: graph only %.1f us, with the 5-NN graph and the positional front end %.1f us; outputs identical: %s" % (
        timed(lambda: ops.knn_pos_head(xyz, 20)), timed(lambda: ops.knn_pos_head(xyz, 20, pos)), same))
