import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ogmm_amd import ops
torch.manual_seed(0)
C, N, M, H, D = 128, 1024, 128, 4, 512
q = torch.randn(C * N, D, device="cuda"); k = torch.randn(C * M, D, device="cuda"); v = torch.randn(C * M, D, device="cuda")
out = torch.empty_like(q)
for _ in range(3): ops.attention(q, k, v, C, N, M, H, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.attention(q, k, v, C, N, M, H, out=out)
e1.record(); torch.cuda.synchronize()
print("gx=%s  %.1f us (pack + attention)" % (os.environ.get("OGMM_ATTN_GX", "auto"), e0.elapsed_time(e1) / 10 * 1e3))
