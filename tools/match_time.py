"""match_kabsch alone: B pairs, J clusters, D channels (default 64, 16, 512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
B, J, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 16, 512)
torch.manual_seed(0)
mu_s, mu_t = torch.randn(B, J, 3, device="cuda"), torch.randn(B, J, 3, device="cuda")
f_s, f_t = torch.randn(B, J, D, device="cuda"), torch.randn(B, J, D, device="cuda")
for _ in range(3): ops.match_kabsch(mu_s, mu_t, f_s, f_t)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.match_kabsch(mu_s, mu_t, f_s, f_t)
e1.record(); torch.cuda.synchronize()
print("match_kabsch B=%d J=%d D=%d  %.1f us" % (B, J, D, e0.elapsed_time(e1) / 20 * 1e3))
