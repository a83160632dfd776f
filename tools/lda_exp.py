"""Does a power-of-two row stride of A (4 KB rows) hot-spot HBM channels?  Compare lda = K with lda = K + pad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
M = 131072
torch.manual_seed(0)
for (n, k) in ((1024, 1024), (512, 1024), (1024, 512), (512, 512)):
    W = torch.randn(n, k, device="cuda") * 0.03
    split = ops.split_f16(W, frag=True)
    row = "N=%4d K=%4d" % (n, k)
    for pad_a, pad_c in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (0, 0)):
        Abuf = torch.randn(M, k + pad_a, device="cuda")
        A = Abuf[:, :k]
        Cbuf = torch.empty(M, n + pad_c, device="cuda")
        C = Cbuf[:, :n]
        def run():
            ops.gemm_nt(A, A.stride(0), k, None, k, M, n, C=C, ldc=C.stride(0), split=split)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row += "  padA%2d/C%2d %6.1f TF" % (pad_a, pad_c, 2.0 * M * n * k / best / 1e9)
    print(row)
