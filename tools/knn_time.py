"""The two kNN kernels alone at the headline shape (GPU box): the LDS-broadcast one (ogmm_knn) against the scalar-load / packed-fp32 one (ogmm_knn_packed).
usage: python3 tools/knn_time.py [C] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
src, tgt, _, _ = synth.make_batch(0, C // 2, N, "partial")
src, tgt = src.cuda(), tgt.cuda()
xyz, packed = ops.pack_clouds(src, tgt)


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


for k in (20, 5):
    a = ops.knn(xyz, k)
    b = ops.knn(xyz, k, packed=packed)
    print("C=%d N=%d k=%2d: LDS kernel %.1f us, packed kernel %.1f us, identical %s" % (
        C, N, k, timed(lambda: ops.knn(xyz, k)), timed(lambda: ops.knn(xyz, k, packed=packed)), torch.equal(a, b)))
print("pack_clouds %.1f us" % timed(lambda: ops.pack_clouds(src, tgt)))
