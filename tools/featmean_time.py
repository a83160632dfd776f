"""gmm_feat_mean alone at B=64 (128 clouds, N=1024, J=16, D=512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
C, N, J, D = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (128, 1024, 16, 512)
g = torch.softmax(torch.randn(C, N, J, device="cuda"), -1); pi = g.mean(1); f = torch.randn(C * N, D, device="cuda")
for _ in range(3): out = ops.gmm_feat_mean(g, pi, f, C, N)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.gmm_feat_mean(g, pi, f, C, N)
e1.record(); torch.cuda.synchronize()
print("feat_mean %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3))
