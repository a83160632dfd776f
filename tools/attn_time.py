import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
C, N, M, H, D = 128, 1024, 128, 4, 512
q = torch.randn(C * N, D, device="cuda"); kv = torch.randn(C * M, 2 * D, device="cuda")
ops.attention(q, kv[:, :D], kv[:, D:], C, N, M, H); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attention(q, kv[:, :D], kv[:, D:], C, N, M, H)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print("attention C=%d N=%d M=%d H=%d: %.1f us" % (C, N, M, H, best * 1e3))
